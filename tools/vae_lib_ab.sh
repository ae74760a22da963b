#!/bin/bash
# VAE decode: library variants (tools/bin/gv_<name>) against the in-tree build, same box, interleaved; parity tests first
cd /root/repo; mkdir -p gpurun_out
fmt='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d["vae_decode"]; print(v["ms_per_clip"], v["ms_per_clip_untiled"])'
{
timeout 1200 python -m pytest tests/test_gpu_vae.py tests/test_vae_naive_golden.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do for n in "$@"; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "vae $n (tiled, untiled ms per clip): "; ORV_LIB=$L python bench.py --no-legs --no-cpu-baseline --no-pmc --steps 3 --warmup 1 2>/dev/null | python -c "$fmt"
done; done
} > gpurun_out/vae_lib_ab.txt 2>&1
cat gpurun_out/vae_lib_ab.txt
