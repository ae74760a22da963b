#!/bin/bash
# 64-rows-per-wave attention kernel (attn_fwd_w64_kernel, ORV_ATTN_W64=1) against the 8-wave ping-pong kernel: parity tests with the switch
# on, then standalone interleaved timing at the headline shape and at B = 1
cd /root/repo; mkdir -p gpurun_out
{
ORV_ATTN_W64=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -4
cd tools/bin
for r in 1 2 3; do for w in 0 1; do
  echo -n "W64=$w B=4: "; ORV_ATTN_W64=$w LD_LIBRARY_PATH=/root/repo/orv_amd FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4
  echo -n "W64=$w B=1: "; ORV_ATTN_W64=$w LD_LIBRARY_PATH=/root/repo/orv_amd FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 1
done; done
} > gpurun_out/w64_ab.txt 2>&1
cat gpurun_out/w64_ab.txt
