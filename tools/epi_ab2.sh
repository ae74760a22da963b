#!/bin/bash
# in-model A/B of library variants (tools/bin/gv_<name>) against the in-tree build; 3 interleaved rounds
cd /root/repo; mkdir -p gpurun_out
fmt='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], " ".join("%s %.4f" % (k["kernel"].split("(")[0][-24:], k["avg_ms"]) for k in d["kernels"][:7]))'
{
for r in 1 2 3; do for n in "$@"; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$n : "; ORV_LIB=$L python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"
done; done
} > gpurun_out/epi_ab2.txt 2>&1
cat gpurun_out/epi_ab2.txt
