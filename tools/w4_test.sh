#!/bin/bash
# 4-wave (one wave per SIMD, 128-row wave tiles) experiments vs the phased and ring kernels, same box, interleaved
cd tools/bin
export LD_LIBRARY_PATH=../../orv_amd
for t in 4,256,256 4,256,128 4,256,384; do
for s in "4096 4608 4096 0 4096 0 0" "700 768 512 2 350 30 64" "3226 7680 1920 1 3226 226 600" "12904 1536 1920 3 3226 226 600"; do echo -n "$t "; ORV_GEMM_TILE=$t timeout 120 ./kbench_gemm check $s < /dev/null; done; done
for r in 1 2; do
  for s in "4096 4096 4096 0" "8192 8192 8192 0" "12904 7680 1920 1"; do
    for t in 4,256,256 3,256,256 2,256,256 1,256,256; do
      echo -n "tile $t: "; ORV_GEMM_TILE=$t timeout 60 ./kbench_gemm bench $s 30 < /dev/null
    done
    echo -n "tile 4,256,256 4 slots: "; ORV_GEMM_P4_SLOTS=4 ORV_GEMM_TILE=4,256,256 timeout 60 ./kbench_gemm bench $s 30 < /dev/null
  done
  for s in "12904 5760 1920 0" "12904 1920 7680 2"; do
    for t in 4,256,384 4,256,128 1,256,384 1,256,192; do
      echo -n "tile $t: "; ORV_GEMM_TILE=$t timeout 60 ./kbench_gemm bench $s 30 < /dev/null
    done
  done
done
