#!/bin/bash
# headline step with the batch cut into ORV_CHAINS chains inside the captured graph (same box, interleaved): chains_ab.sh [values...]
cd /root/repo; mkdir -p gpurun_out
VALS=${@:-1 2 4}
{
for r in 1 2 3; do for f in $VALS; do echo -n "ORV_CHAINS=$f : "; env ORV_CHAINS=$f python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>gpurun_out/chains_err_$f.txt | python tools/bench_line_brief.py; done; done
} > gpurun_out/chains_ab.txt 2>&1
cat gpurun_out/chains_ab.txt; tail -3 gpurun_out/chains_err_2.txt
