#!/bin/bash
# d8 kernel (gemm_d8.hip: packed A straight to registers, W through four LDS buffers) against the t8 kernel (row-major operands), same process,
# interleaved, outputs compared bit for bit: the CogVideoX-2B GEMM shapes at B = 4, ragged / small shapes, packed C.  usage: bash tools/d8_ab.sh [rounds]
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
export LD_LIBRARY_PATH=/root/repo/orv_amd
R=${1:-3}
{
for bn in 256 192; do for epi in 0 1 2; do
  timeout 120 ./kbench_gemm abp 300 $((bn*3)) 192 $epi 1 $bn
  timeout 120 ./kbench_gemm abp 3226 $((bn*3)) 384 $epi 1 $bn
done; done
timeout 120 ./kbench_gemm abp 3226 768 384 1 1 256 1
timeout 120 ./kbench_gemm abp 3226 576 384 0 1 192 1
timeout 300 ./kbench_gemm abp 12904 7680 1920 1 $R 256
timeout 300 ./kbench_gemm abp 12904 7680 1920 1 $R 256 1
timeout 300 ./kbench_gemm abp 12904 3840 1920 0 $R 256
timeout 300 ./kbench_gemm abp 12904 1920 7680 2 $R 192
timeout 300 ./kbench_gemm abp 12904 1920 1920 2 $R 192
timeout 300 ./kbench_gemm abp 12904 1920 1920 0 $R 192
timeout 300 ./kbench_gemm abp 12904 5760 1920 0 $R 192
timeout 300 ./kbench_gemm abp 8192 8192 7680 0 $R 256
timeout 300 ./kbench_gemm abp 3226 7680 1920 1 $R 256
} > ../../gpurun_out/d8_ab.txt 2>&1
cat ../../gpurun_out/d8_ab.txt
