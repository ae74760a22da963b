#!/bin/bash
# variants / ablations of the 16x16x32 attention kernel (variant builds in tools/bin/av_*), standalone, interleaved
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do
  echo -n "pp32        : "; ORV_ATTN_M16=0 LD_LIBRARY_PATH=av_base FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4
  for v in base m16vup; do echo -n "m16 $v : "; ORV_ATTN_M16=1 LD_LIBRARY_PATH=av_$v FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4; done
done
} > ../../gpurun_out/m16_abl.txt 2>&1
cat ../../gpurun_out/m16_abl.txt
