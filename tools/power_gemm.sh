#!/bin/bash
# Socket power and shader clock (rocm-smi, 1 s samples) while ONE GEMM kernel loops: the FFN2 shape on the t8 and the d8 kernel, random against
# zero operands.  Evidence for DESIGN.md 4.1 finding 4 (the GEMMs run at the power-limited clock on random data).
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
export LD_LIBRARY_PATH=/root/repo/orv_amd
{
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
for z in 0 1; do
  if [ $z = 1 ]; then export KB_ZERO=1; else unset KB_ZERO; fi
  echo "== FFN2 shape (M=12904 N=1920 K=7680, epi 2), zero operands = $z: t8 and d8 alternate, ~25 s"
  ( for i in $(seq 1 60); do ./kbench_gemm abp 12904 1920 7680 2 12 192 > /tmp/pg_$z.log 2>&1; done ) &
  BP=$!
  sleep 6
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' | sed 's/=\+//g' | cut -c1-260; echo
    sleep 1
  done
  kill $BP 2>/dev/null; wait $BP 2>/dev/null
  grep "t8 row\|d8 packed" /tmp/pg_$z.log | sed 's/abp M=/M=/' | cut -c1-130
done
} > ../../gpurun_out/power_gemm.txt 2>&1
cat ../../gpurun_out/power_gemm.txt
