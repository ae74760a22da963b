#!/bin/bash
# round-3 profile refresh (GPU box) -> gpurun_out/refresh3/ ; the developer copies the summaries into profiles/r3_*
R=/root/repo; O=$R/gpurun_out/refresh3; rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_line_default.json 2> $O/bench_default.err
python bench.py --eager --no-vae --no-cpu-baseline --no-legs 2>/dev/null | tail -1 > $O/bench_line_eager.json
python bench.py --batch 1 --no-vae --no-cpu-baseline --no-legs 2>/dev/null | tail -1 > $O/bench_line_b1.json
python bench.py --batch 2 --no-vae --no-cpu-baseline --no-legs 2>/dev/null | tail -1 > $O/bench_line_b2.json
python bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/train_2b_line.json
python bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 --grad-ckpt 2>/dev/null | tail -1 > $O/train_2b_ckpt_line.json
python bench.py --mode train --model 5b --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/train_5b_line.json
python bench.py --mode train --model 5b --no-cpu-baseline --steps 6 --warmup 2 --grad-ckpt 2>/dev/null | tail -1 > $O/train_5b_ckpt_line.json
bash tools/profile_bench.sh r3nv --no-vae --no-legs > $O/profile_bench_nv.log 2>&1 < /dev/null
cp gpurun_out/prof_r3nv/r3nv_kernel_stats_summary.txt $O/bench_novae_kernel_stats_summary.txt; grep '^{' gpurun_out/prof_r3nv/bench_stdout.log | tail -1 > $O/bench_line_under_rocprof.json
bash tools/profile_bench.sh r3tr --mode train --steps 4 --warmup 1 > $O/profile_train.log 2>&1 < /dev/null
cp gpurun_out/prof_r3tr/r3tr_kernel_stats_summary.txt $O/train_kernel_stats_summary.txt
bash tools/profile_vae.sh r3 1 3 > $O/profile_vae.log 2>&1 < /dev/null
cp gpurun_out/prof_vae_r3/r3_vae_kernel_stats_summary.txt $O/vae_kernel_stats_summary.txt 2>/dev/null
bash tools/pmc_bench.sh r3 --no-vae --no-legs --eager > $O/pmc_bench.log 2>&1 < /dev/null
cp gpurun_out/pmc_bench_r3/summary.txt $O/pmc_summary_inference.txt; cp gpurun_out/pmc_bench_r3/hbm_traffic.json $O/hbm_traffic.json
bash tools/pmc_bench.sh r3tr --mode train > $O/pmc_bench_train.log 2>&1 < /dev/null
cp gpurun_out/pmc_bench_r3tr/summary.txt $O/pmc_summary_train.txt; cp gpurun_out/pmc_bench_r3tr/hbm_traffic.json $O/hbm_traffic_train.json
find gpurun_out -name "*.csv" -size +200k -delete
rm -rf gpurun_out/prof_* gpurun_out/pmc_bench_*
ls -la $O
