#!/bin/bash
# round 6, second GPU call: the whole GPU suite (no -x: every failure listed), then kernel-trace stats of the configs[4] training leg
# (CogVideoX1.5-5B, activation checkpointing) and of the 2B training step
cd /root/repo; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|ERROR|rel-L2|census|full-depth|bf16 oracle|max \|diff\|" | tail -40 > gpurun_out/pytest_gpu.txt; cat gpurun_out/pytest_gpu.txt
bash tools/profile_bench.sh r6_5b --mode train --model 5b --grad-ckpt --steps 3 --warmup 2 --batch 4 > gpurun_out/r6_5b_prof.txt 2>&1; tail -30 gpurun_out/r6_5b_prof.txt | cut -c1-220
cp gpurun_out/prof_r6_5b/r6_5b_kernel_stats_summary.txt gpurun_out/r6_train_5b_ckpt_kernel_stats_summary.txt
bash tools/profile_bench.sh r6_tr --mode train --steps 5 --warmup 3 --batch 4 > gpurun_out/r6_tr_prof.txt 2>&1; tail -30 gpurun_out/r6_tr_prof.txt | cut -c1-220
cp gpurun_out/prof_r6_tr/r6_tr_kernel_stats_summary.txt gpurun_out/r6_train_kernel_stats_summary.txt
