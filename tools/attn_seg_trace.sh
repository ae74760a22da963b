#!/bin/bash
# per-wave segment times of the two ping-pong attention kernels (build: EXTRA_VARIANTS='seg:-DORV_SEG_TRACE' bash tools/attn_variants.sh)
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
for m in 0 1; do
  ORV_ATTN_M16=$m LD_LIBRARY_PATH=av_seg FUSED=1 BOUND=12 ITERS=10 TRACE=../../gpurun_out/seg_m$m.txt TRACE_WORDS=32 ./kbench_attn 4
done
python3 - <<'PY'
import numpy as np
for m, name in ((0, "32x32x16 (attn_fwd_pp_kernel)"), (1, "16x16x32 (attn_fwd_m16_kernel)")):
    a = np.loadtxt(f"/root/repo/gpurun_out/seg_m{m}.txt", dtype=np.float64)[:, 1:].reshape(-1, 8, 4) * 10.0   # ns per wave, summed over tiles
    a = a[a[:, :, 0].min(axis=1) > 0]                      # workgroups whose waves were all active
    for half, sl in (("first half (waves 0-3)", slice(0, 4)), ("second half (waves 4-7)", slice(4, 8))):
        v = a[:, sl, :].mean(axis=(0, 1)) / 51.0
        print(f"{name:34s} {half:24s}: per tile  matrix segment {v[0]:7.1f} ns | barrier {v[1]:6.1f} | vector segment {v[2]:7.1f} | barrier {v[3]:6.1f} | sum {v.sum():7.1f}")
PY
