#!/bin/bash
# four-wave 256 x 256 experiment (ORV_GEMM_TILE=4,256,256) against gemm_t8_kernel<256,*>: parity, then standalone A/B
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out; export LD_LIBRARY_PATH=/root/repo/orv_amd
{
for s in "4096 7680 4096 0 4096 0 0" "700 768 512 2 350 30 64" "3226 7680 1920 1 3226 226 600" "12904 7680 1920 1 3226 226 600" "512 256 128 0 512 0 0"; do
  echo -n "check t4 $s: "; ORV_GEMM_TILE=4,256,256 timeout 120 ./kbench_gemm check $s < /dev/null | tail -1
done
for r in 1 2 3; do for t in 3,256,256 4,256,256; do
  echo -n "tile $t 8192^3 : "; timeout 100 ./kbench_gemm ab 8192 8192 8192 0 3 $t | tail -1
  echo -n "tile $t FFN1   : "; timeout 100 ./kbench_gemm ab 12904 7680 1920 1 3 $t | tail -1
  echo -n "tile $t 4096^3 : "; timeout 100 ./kbench_gemm ab 4096 4096 4096 0 3 $t | tail -1
done; done
} > ../../gpurun_out/t4_ab.txt 2>&1
cat ../../gpurun_out/t4_ab.txt
