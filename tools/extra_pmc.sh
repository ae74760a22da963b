#!/bin/bash
# Extra PMC summaries beyond the headline command: B=1 denoise, 5B SFT step, full-size VAE decode.
cd /root/repo
bash tools/pmc_bench.sh b1 --batch 1 --no-vae > /dev/null 2>&1
bash tools/pmc_bench.sh tr5b --mode train --model 5b > /dev/null 2>&1
PMC_CMD="python /root/repo/tools/vae_bench.py 1 2" bash tools/pmc_bench.sh vae > /dev/null 2>&1
mkdir -p gpurun_out/extra_pmc
for t in b1 tr5b vae; do cp gpurun_out/pmc_bench_$t/summary.txt gpurun_out/extra_pmc/pmc_summary_$t.txt; done
find gpurun_out -name "*.csv" -size +1M -delete
head -8 gpurun_out/extra_pmc/*.txt | cut -c1-200
