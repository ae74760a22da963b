#!/bin/bash
# B = 1 shapes (M = 3226): d8 tiles on a packed A against whatever the row-major cost model picks (single-round simple kernels); + the rs1 epilogue variant at B = 4
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
export LD_LIBRARY_PATH=/root/repo/orv_amd KB_RM_FREE=1
{
for bn in 128 192; do
  timeout 120 ./kbench_gemm abp 3226 1920 7680 2 3 $bn | grep -v differ
  timeout 120 ./kbench_gemm abp 3226 1920 1920 2 3 $bn | grep -v differ
  timeout 120 ./kbench_gemm abp 3226 5760 1920 0 3 $bn | grep -v differ
done
timeout 120 ./kbench_gemm abp 3226 7680 1920 1 3 256 | grep -v differ
timeout 120 ./kbench_gemm abp 6452 1920 7680 2 3 128 | grep -v differ
timeout 120 ./kbench_gemm abp 6452 1920 7680 2 3 192 | grep -v differ
timeout 120 ./kbench_gemm abp 6452 1920 1920 2 3 128 | grep -v differ
unset KB_RM_FREE
bash /root/repo/tools/d8_rs_ab.sh
} > ../../gpurun_out/d8_b1.txt 2>&1
cat ../../gpurun_out/d8_b1.txt
