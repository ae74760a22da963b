#!/bin/bash
# what do LDS fragment reads cost in energy?  d8 with every fragment read twice (right results) against the shipped kernel, zero and random operands,
# package power / clock sampled (tools/power_gemm.py)
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2; do for v in base read2; do
  L=/root/repo/tools/bin/dv_$v/liborv_mi355.so; [ $v = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo "== $v"; PG_D8_ONLY=1 ORV_LIB=$L python tools/power_gemm.py 2>&1 | grep "d8:"
done; done
} > gpurun_out/power_abl.txt 2>&1
cat gpurun_out/power_abl.txt
