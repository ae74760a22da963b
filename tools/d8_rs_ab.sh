#!/bin/bash
# d8 epilogue 2: one residual register set (no spills, loads exposed per 64-column group) against two (one group ahead, 10 VGPR spills): standalone, interleaved
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in base rs0; do
  L=/root/repo/tools/bin/dv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v FFN2 : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 7680 2 3 192 | grep "d8 packed"
  echo -n "$v out  : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 1920 2 3 192 | grep "d8 packed"
done; done
} > ../../gpurun_out/d8_rs_ab.txt 2>&1
cat ../../gpurun_out/d8_rs_ab.txt
