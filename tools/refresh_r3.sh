#!/bin/bash
# round-3 profile refresh (GPU box): rocprofv3 kernel-trace stats of the headline / train commands + PMC passes -> gpurun_out/refresh3/
R=/root/repo; O=$R/gpurun_out/refresh3; rm -rf $O; mkdir -p $O
cd $R
bash tools/profile_bench.sh r3nv --no-vae --no-legs > $O/profile_bench_nv.log 2>&1 < /dev/null
cp gpurun_out/prof_r3nv/r3nv_kernel_stats_summary.txt $O/bench_novae_kernel_stats_summary.txt; grep '^{' gpurun_out/prof_r3nv/bench_stdout.log | tail -1 > $O/bench_line_under_rocprof.json
bash tools/profile_bench.sh r3tr --mode train --steps 4 --warmup 1 > $O/profile_train.log 2>&1 < /dev/null
cp gpurun_out/prof_r3tr/r3tr_kernel_stats_summary.txt $O/train_kernel_stats_summary.txt
bash tools/pmc_bench.sh r3 --no-vae --no-legs --eager > $O/pmc_bench.log 2>&1 < /dev/null
cp gpurun_out/pmc_bench_r3/summary.txt $O/pmc_summary_inference.txt; cp gpurun_out/pmc_bench_r3/hbm_traffic.json $O/hbm_traffic.json
find gpurun_out -name "*.csv" -size +1M -delete
ls -la $O
