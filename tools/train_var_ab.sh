#!/bin/bash
# training-step A/B of library variants (tools/bin/gv_<name>) against the in-tree build, same box, interleaved: bash tools/train_var_ab.sh head flall flnone base
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2; do for n in "$@"; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "train $n : "; ORV_LIB=$L python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
} > gpurun_out/train_var_ab.txt 2>&1
cat gpurun_out/train_var_ab.txt
