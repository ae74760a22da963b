#!/bin/bash
# headline step with variant builds of the library (ORV_LIB), interleaved on one box: lib_ab.sh name1 name2 ... (tools/bin/gv_<name>; "base" = in-tree)
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2 3; do for n in "$@"; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$n : "; ORV_LIB=$L python bench.py --no-legs --no-vae --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], ' '.join('%s %.4f' % (k['kernel'].split('(')[0][-22:], k['avg_ms']) for k in d['kernels'][:5]))"; done; done
} > gpurun_out/lib_ab.txt 2>&1
cat gpurun_out/lib_ab.txt
