#!/bin/bash
# t4 variants (library dirs) vs gemm_t8_kernel<256,*>, standalone, same box, interleaved: bash tools/t4_ab2.sh base t4v1 ...
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
echo -n "check t4 (in-tree) FFN1: "; LD_LIBRARY_PATH=/root/repo/orv_amd ORV_GEMM_TILE=4,256,256 timeout 120 ./kbench_gemm check 3226 7680 1920 1 3226 226 600 < /dev/null | tail -1
echo -n "check t4 (in-tree) gated: "; LD_LIBRARY_PATH=/root/repo/orv_amd ORV_GEMM_TILE=4,256,256 timeout 120 ./kbench_gemm check 700 768 512 2 350 30 64 < /dev/null | tail -1
for r in 1 2 3; do
  echo -n "t8            8192^3 : "; LD_LIBRARY_PATH=/root/repo/orv_amd timeout 100 ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
  echo -n "t8            FFN1   : "; LD_LIBRARY_PATH=/root/repo/orv_amd timeout 100 ./kbench_gemm ab 12904 7680 1920 1 3 3,256,256 | tail -1
  for n in "$@"; do
    L=/root/repo/tools/bin/gv_$n; [ "$n" = base ] && L=/root/repo/orv_amd
    echo -n "t4 $n 8192^3 : "; LD_LIBRARY_PATH=$L timeout 100 ./kbench_gemm ab 8192 8192 8192 0 3 4,256,256 | tail -1
    echo -n "t4 $n FFN1   : "; LD_LIBRARY_PATH=$L timeout 100 ./kbench_gemm ab 12904 7680 1920 1 3 4,256,256 | tail -1
  done
done
} > ../../gpurun_out/t4_ab2.txt 2>&1
cat ../../gpurun_out/t4_ab2.txt
