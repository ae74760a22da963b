#!/bin/bash
# attention variants A/B (ablations of the vector segment) + a per-workgroup timeline of the ping-pong kernel
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in base pksum nosum noexp; do echo -n "$v: "; LD_LIBRARY_PATH=av_$v FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4; done; done
} > ../../gpurun_out/attn_msum.txt 2>&1
cat ../../gpurun_out/attn_msum.txt
LD_LIBRARY_PATH=av_trace FUSED=1 BOUND=12 ITERS=10 TRACE=../../gpurun_out/attn_trace.txt ./kbench_attn 4
LD_LIBRARY_PATH=av_trace FUSED=1 BOUND=12 ITERS=10 TRACE=../../gpurun_out/attn_trace_3r.txt ./kbench_attn 4 3072 32
