#!/bin/bash
# per-workgroup timeline of the ping-pong attention kernels: build the library with -DORV_PP_TRACE (EXTRA_VARIANTS='trace:-DORV_PP_TRACE'
# bash tools/attn_variants.sh), run this on the GPU box, read gpurun_out/attn_trace*.txt (index start end hwid|xcc<<32 item; 100 MHz ticks)
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
for m in 0 1; do
LD_LIBRARY_PATH=av_trace ORV_ATTN_M16=$m FUSED=1 BOUND=12 ITERS=10 TRACE=../../gpurun_out/attn_trace_m$m.txt ./kbench_attn 4
done
python3 - <<'PY'
import numpy as np
for m in (0, 1):
    a = np.loadtxt(f"/root/repo/gpurun_out/attn_trace_m{m}.txt", dtype=np.uint64)
    st = a[:, 1].astype(np.int64); en = a[:, 2].astype(np.int64)
    t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0
    life = en - st
    print(f"M16={m}: span {en.max():.1f} us, lifetime mean {life.mean():.1f} min {life.min():.1f} max {life.max():.1f} p10 {np.percentile(life,10):.1f} p90 {np.percentile(life,90):.1f}; span / mean lifetime {en.max()/life.mean():.2f}")
PY
