#!/bin/bash
# Regenerates every artefact under profiles/ that quotes the shipped kernels (run on the GPU box through gpurun; outputs land
# in gpurun_out/refresh/, the developer copies them into profiles/).
R=/root/repo; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_line_default.json 2> $O/bench_default.err
python bench.py --batch 1 --no-vae --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_b1.json
python bench.py --batch 1 --no-vae --no-cpu-baseline --graph 2>/dev/null | tail -1 > $O/bench_line_b1_graph.json
python bench.py --batch 2 --no-vae --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line_b2.json
python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 > $O/train_2b_line.json
python bench.py --mode train --model 5b --no-cpu-baseline 2>/dev/null | tail -1 > $O/train_5b_line.json
python bench.py --mode train --model 5b --batch 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/train_5b_b1_line.json
bash tools/profile_bench.sh r2 > $O/profile_bench.log 2>&1 < /dev/null
cp gpurun_out/prof_r2/r2_kernel_stats_summary.txt $O/bench_kernel_stats_summary.txt; grep '^{' gpurun_out/prof_r2/bench_stdout.log | tail -1 > $O/bench_line_under_rocprof.json
bash tools/profile_bench.sh r2nv --no-vae > $O/profile_bench_nv.log 2>&1 < /dev/null
cp gpurun_out/prof_r2nv/r2nv_kernel_stats_summary.txt $O/bench_novae_kernel_stats_summary.txt
bash tools/profile_bench.sh r2tr --mode train --steps 4 --warmup 1 > $O/profile_train.log 2>&1 < /dev/null
cp gpurun_out/prof_r2tr/r2tr_kernel_stats_summary.txt $O/train_kernel_stats_summary.txt
bash tools/profile_vae.sh r2 1 3 > $O/profile_vae.log 2>&1 < /dev/null
cp gpurun_out/prof_vae_r2/r2_vae_kernel_stats_summary.txt $O/vae_kernel_stats_summary.txt
python tools/vae_bench.py 1 5 2>/dev/null | tail -1 > $O/vae_bench.txt; python tools/vae_bench.py 4 3 2>/dev/null | tail -1 >> $O/vae_bench.txt
bash tools/pmc_bench.sh r2 --no-vae > $O/pmc_bench.log 2>&1 < /dev/null
cp gpurun_out/pmc_bench_r2/summary.txt $O/pmc_summary_inference.txt; cp gpurun_out/pmc_bench_r2/hbm_traffic.json $O/hbm_traffic.json
bash tools/pmc_bench.sh r2tr --mode train > $O/pmc_bench_train.log 2>&1 < /dev/null
cp gpurun_out/pmc_bench_r2tr/summary.txt $O/pmc_summary_train.txt; cp gpurun_out/pmc_bench_r2tr/hbm_traffic.json $O/hbm_traffic_train.json
find gpurun_out -name "*.csv" -size +1M -delete
ls -la $O
