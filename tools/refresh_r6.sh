#!/bin/bash
# round-6 profile refresh (GPU box) -> gpurun_out/refresh6/ ; the developer copies the summaries into profiles/r6_*
R=/root/repo; O=$R/gpurun_out/refresh6; rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | tail -1 > $O/gpu_test_count.txt
python -m pytest tests -m "not gpu" --collect-only -q 2>/dev/null | tail -1 >> $O/gpu_test_count.txt
python bench.py > $O/bench_line_default.json 2> $O/bench_default.err
python bench.py --eager --no-vae --no-cpu-baseline --no-legs --no-pmc 2>/dev/null | tail -1 > $O/bench_line_eager.json
python bench.py --batch 1 --no-vae --no-cpu-baseline --no-legs --no-pmc 2>/dev/null | tail -1 > $O/bench_line_b1.json
python bench.py --batch 2 --no-vae --no-cpu-baseline --no-legs --no-pmc 2>/dev/null | tail -1 > $O/bench_line_b2.json
python bench.py --mode train --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 > $O/train_2b_line.json
python bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 --grad-ckpt 2>/dev/null | tail -1 > $O/train_2b_ckpt_line.json
python bench.py --mode train --model 5b --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/train_5b_line.json
python bench.py --mode train --model 5b --no-cpu-baseline --steps 6 --warmup 2 --grad-ckpt 2>/dev/null | tail -1 > $O/train_5b_ckpt_line.json
bash tools/profile_bench.sh r6nv --no-vae --no-legs --no-pmc > $O/profile_bench_nv.log 2>&1 < /dev/null
cp gpurun_out/prof_r6nv/r6nv_kernel_stats_summary.txt $O/bench_novae_kernel_stats_summary.txt; grep '^{' gpurun_out/prof_r6nv/bench_stdout.log | tail -1 > $O/bench_line_under_rocprof.json
bash tools/profile_bench.sh r6b1 --batch 1 --no-vae --no-legs --no-pmc > $O/profile_bench_b1.log 2>&1 < /dev/null
cp gpurun_out/prof_r6b1/r6b1_kernel_stats_summary.txt $O/bench_b1_kernel_stats_summary.txt
bash tools/profile_bench.sh r6tr --mode train --steps 4 --warmup 1 > $O/profile_train.log 2>&1 < /dev/null
cp gpurun_out/prof_r6tr/r6tr_kernel_stats_summary.txt $O/train_kernel_stats_summary.txt
bash tools/profile_bench.sh r6t5 --mode train --model 5b --grad-ckpt --steps 3 --warmup 2 > $O/profile_train_5b.log 2>&1 < /dev/null
cp gpurun_out/prof_r6t5/r6t5_kernel_stats_summary.txt $O/train_5b_ckpt_kernel_stats_summary.txt
bash tools/profile_vae.sh r6 1 3 > $O/profile_vae.log 2>&1 < /dev/null
cp gpurun_out/prof_vae_r6/r6_vae_kernel_stats_summary.txt $O/vae_kernel_stats_summary.txt 2>/dev/null
bash tools/pmc_bench.sh r6 --no-vae --no-legs --eager --no-pmc > $O/pmc_bench.log 2>&1 < /dev/null
cp gpurun_out/pmc_bench_r6/summary.txt $O/pmc_summary_inference.txt; cp gpurun_out/pmc_bench_r6/hbm_traffic.json $O/hbm_traffic.json
bash tools/pmc_bench.sh r6tr --mode train > $O/pmc_bench_train.log 2>&1 < /dev/null
cp gpurun_out/pmc_bench_r6tr/summary.txt $O/pmc_summary_train.txt; cp gpurun_out/pmc_bench_r6tr/hbm_traffic.json $O/hbm_traffic_train.json
find gpurun_out -name "*.csv" -size +200k -delete
rm -rf gpurun_out/prof_* gpurun_out/pmc_bench_*
ls -la $O
