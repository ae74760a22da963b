#!/bin/bash
# attention backward: library variants (tools/bin/gv_<name>) against the in-tree build: gradient tests with the variant, then the pair timed
# forked (default) and back to back (ORV_ATTN_BWD_FORK=0), same box, interleaved x 3
cd /root/repo; mkdir -p gpurun_out; export PYTHONPATH=/root/repo
{
for n in "$@"; do
  [ "$n" = base ] && continue
  echo -n "tests with $n: "; ORV_LIB=/root/repo/tools/bin/gv_$n/liborv_mi355.so timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -1
done
for r in 1 2 3; do for n in "$@"; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$n forked: "; ORV_LIB=$L timeout 120 python tools/time_attn_bwd.py
  echo -n "$n serial: "; ORV_ATTN_BWD_FORK=0 ORV_LIB=$L timeout 120 python tools/time_attn_bwd.py
done; done
} > gpurun_out/bwd_var_ab.txt 2>&1
cat gpurun_out/bwd_var_ab.txt
