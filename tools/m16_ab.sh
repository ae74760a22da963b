#!/bin/bash
# 16x16x32 attention forward (ORV_ATTN_M16=1) against the 32x32x16 ping-pong kernel: parity tests, standalone A/B, in-model A/B
cd /root/repo; mkdir -p gpurun_out
{
ORV_ATTN_M16=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or attn" 2>&1 | tail -3
cd tools/bin; export LD_LIBRARY_PATH=/root/repo/orv_amd
for r in 1 2 3; do for v in 0 1; do echo -n "M16=$v : "; ORV_ATTN_M16=$v FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4; done; done
for v in 0 1; do echo -n "B=1 M16=$v : "; ORV_ATTN_M16=$v FUSED=1 BOUND=12 ITERS=100 ./kbench_attn 1; done
} > gpurun_out/m16_ab.txt 2>&1
cat gpurun_out/m16_ab.txt
