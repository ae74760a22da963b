#!/bin/bash
# persistent attention (per-XCD work queues, split tail items): parity tests, then A/B against the dispatcher-scheduled kernel
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or attn" 2>&1 | tail -3
cd tools/bin; export LD_LIBRARY_PATH=/root/repo/orv_amd
{
for r in 1 2; do
  echo -n "B=4 dispatcher : "; ORV_ATTN_PS=0 FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4
  for sp in 0 8 17 25 35 50 100; do echo -n "B=4 queues split $sp% : "; ORV_ATTN_PS_SPLIT=$sp FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4; done
done
echo -n "B=1 dispatcher : "; ORV_ATTN_PS=0 FUSED=1 BOUND=12 ITERS=100 ./kbench_attn 1
for sp in 0 25 50 100; do echo -n "B=1 queues split $sp% : "; ORV_ATTN_PS_SPLIT=$sp FUSED=1 BOUND=12 ITERS=100 ./kbench_attn 1; done
echo -n "B=2 dispatcher : "; ORV_ATTN_PS=0 FUSED=1 BOUND=12 ITERS=60 ./kbench_attn 2
for sp in 17 35 100; do echo -n "B=2 queues split $sp% : "; ORV_ATTN_PS_SPLIT=$sp FUSED=1 BOUND=12 ITERS=60 ./kbench_attn 2; done
} > ../../gpurun_out/attn_ps.txt 2>&1
cat ../../gpurun_out/attn_ps.txt
