#!/bin/bash
# standalone A/B of the attention variant libraries (interleaved rounds) + PMC of the base ping-pong kernel
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in $(ls -d av_* | sed 's/av_//'); do echo -n "$v: "; LD_LIBRARY_PATH=av_$v FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4; done; done
echo -n "v2 static: "; ORV_ATTN_PP=0 LD_LIBRARY_PATH=av_base FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4
echo -n "v2 online: "; ORV_ATTN_PP=0 LD_LIBRARY_PATH=av_base FUSED=1 ITERS=40 ./kbench_attn 4
} > ../../gpurun_out/attn_variants.txt 2>&1
cat ../../gpurun_out/attn_variants.txt
cd /root/repo
FUSED=1 BOUND=12 ITERS=5 LD_LIBRARY_PATH=/root/repo/tools/bin/av_${PMCV:-base} bash tools/pmc_kernel.sh attn_pp attn_fwd_pp -- ./kbench_attn 4 > /dev/null 2>&1
cat gpurun_out/pmc_attn_pp/summary.txt
