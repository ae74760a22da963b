#!/bin/bash
# B=1 (M=3226) GEMM shapes: old library (tools/bin/abl_ph/OLD) vs current, default tile choice and forced 192x128 at 2/3/4 stages
cd tools/bin
for s in "3226 1920 7680 2" "3226 1920 1920 2" "3226 7680 1920 1" "3226 5760 1920 0" "12904 1920 7680 2" "12904 7680 1920 1" "6452 1920 7680 2"; do
  echo "== $s"
  for r in 1 2; do
    echo -n "OLD default: "; LD_LIBRARY_PATH=abl_ph/OLD timeout 60 ./kbench_gemm bench $s 50 < /dev/null
    for ns in 2 3 4; do echo -n "192x128 NS=$ns: "; ORV_GEMM_NS=$ns ORV_GEMM_TILE=0,192,128 LD_LIBRARY_PATH=../../orv_amd timeout 60 ./kbench_gemm bench $s 50 < /dev/null; done
  done
done
for s in "3226 1920 7680 2" "3226 1920 1920 0" "3226 7680 1920 1" "3226 1920 64 2" "3226 1920 128 2"; do echo -n "check $s: "; ORV_GEMM_TILE=0,192,128 LD_LIBRARY_PATH=../../orv_amd timeout 120 ./kbench_gemm check $s 3226 226 600 < /dev/null; done
