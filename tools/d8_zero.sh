#!/bin/bash
# d8 ablations on ZERO operands (KB_ZERO=1): separates the structural cost of a stream from the DVFS effect of the data it stops toggling
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2; do for v in base noread nodma noa nowait; do
  L=/root/repo/tools/bin/dv_$v; [ $v = base ] && L=/root/repo/orv_amd
  for z in 0 1; do
  if [ $z = 1 ]; then export KB_ZERO=1; else unset KB_ZERO; fi
  echo -n "$v zero=$z FFN1: "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 7680 1920 0 3 256 | grep "t8 row\|d8 packed" | sed 's/.*\(t8\|d8\)[^:]*: median \([0-9]*\).*(\(.*\) ms)/\1 \2 TF \3 ms;/' | tr "\n" " "; echo
  echo -n "$v zero=$z FFN2: "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 7680 0 3 192 | grep "t8 row\|d8 packed" | sed 's/.*\(t8\|d8\)[^:]*: median \([0-9]*\).*(\(.*\) ms)/\1 \2 TF \3 ms;/' | tr "\n" " "; echo
  done
done; done
} > ../../gpurun_out/d8_zero.txt 2>&1
cat ../../gpurun_out/d8_zero.txt
