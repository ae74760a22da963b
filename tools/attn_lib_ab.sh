#!/bin/bash
# attention forward: library variants (tools/bin/gv_<name>) against the in-tree build - parity tests on the in-tree build, standalone A/B, in-model A/B
cd /root/repo; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention or attn" 2>&1 | tail -2
cd tools/bin
for r in 1 2 3; do for n in "$@"; do
  L=/root/repo/tools/bin/gv_$n; [ "$n" = base ] && L=/root/repo/orv_amd
  echo -n "$n : "; LD_LIBRARY_PATH=$L FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4
done; done
cd /root/repo
fmt='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], " ".join("%s %.4f" % (k["kernel"].split("(")[0][-24:], k["avg_ms"]) for k in d["kernels"][:2]))'
for r in 1 2 3; do for n in "$@"; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "model $n : "; ORV_LIB=$L python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"
done; done
} > gpurun_out/attn_lib_ab.txt 2>&1
cat gpurun_out/attn_lib_ab.txt
