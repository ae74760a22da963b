#!/bin/bash
# functional check of the N > 1 paths on ONE GPU: two ranks share it, gloo carries the collectives (ORV_DIST_BACKEND=gloo)
cd /root/repo; mkdir -p gpurun_out
{
echo "== denoise, 2 ranks"; ORV_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-vae 2>&1 | tail -2 | cut -c1-600
echo "== train, 2 ranks (overlapped gradient exchange)"; ORV_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --mode train --steps 3 --warmup 1 2>&1 | tail -2 | cut -c1-600
echo "== train, 2 ranks on ONE GPU over the nccl backend (= RCCL; it may refuse two ranks on one device - recorded either way)"; HSA_ENABLE_IPC_MODE_LEGACY=0 ORV_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --mode train --steps 2 --warmup 1 --layers 4 2>&1 | tail -3 | cut -c1-700
echo "== train, 1 rank, same seeds (loss should match rank-0-only semantics loosely)"; timeout 600 python bench.py --mode train --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-400
echo "== odd flags"; timeout 600 python bench.py --steps 1 --warmup 0 --no-legs --no-vae --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
} > gpurun_out/multi_rank_check.txt 2>&1
cat gpurun_out/multi_rank_check.txt
