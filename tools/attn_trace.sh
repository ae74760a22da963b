#!/bin/bash
# per-workgroup timeline of the ping-pong attention kernel: build the library with -DORV_PP_TRACE (EXTRA_VARIANTS='trace:-DORV_PP_TRACE'
# bash tools/attn_variants.sh), run this on the GPU box, read gpurun_out/attn_trace*.txt (index start end hwid|xcc<<32 item; 100 MHz ticks)
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
LD_LIBRARY_PATH=av_trace FUSED=1 BOUND=12 ITERS=10 TRACE=../../gpurun_out/attn_trace.txt ./kbench_attn 4
LD_LIBRARY_PATH=av_trace FUSED=1 BOUND=12 ITERS=10 TRACE=../../gpurun_out/attn_trace_3r.txt ./kbench_attn 4 3072 32
