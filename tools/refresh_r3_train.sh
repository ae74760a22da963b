#!/bin/bash
# round-3 profile refresh, training part only (GPU box) -> gpurun_out/refresh3t/
R=/root/repo; O=$R/gpurun_out/refresh3t; rm -rf $O; mkdir -p $O
cd $R
python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/train_2b_line.json
python bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 --grad-ckpt 2>/dev/null | tail -1 > $O/train_2b_ckpt_line.json
python bench.py --mode train --model 5b --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/train_5b_line.json
python bench.py --mode train --model 5b --no-cpu-baseline --steps 6 --warmup 2 --grad-ckpt 2>/dev/null | tail -1 > $O/train_5b_ckpt_line.json
bash tools/profile_bench.sh r3tr --mode train --steps 4 --warmup 1 > $O/profile_train.log 2>&1 < /dev/null
cp gpurun_out/prof_r3tr/r3tr_kernel_stats_summary.txt $O/train_kernel_stats_summary.txt
bash tools/pmc_bench.sh r3tr --mode train > $O/pmc_bench_train.log 2>&1 < /dev/null
cp gpurun_out/pmc_bench_r3tr/summary.txt $O/pmc_summary_train.txt; cp gpurun_out/pmc_bench_r3tr/hbm_traffic.json $O/hbm_traffic_train.json
find gpurun_out -name "*.csv" -size +200k -delete
rm -rf gpurun_out/prof_* gpurun_out/pmc_bench_*
ls -la $O
