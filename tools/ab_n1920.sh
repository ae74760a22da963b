#!/bin/bash
# same-box interleaved A/B of tile choices for the N = 1920 GEMMs (out-proj K=1920, FFN2 K=7680) and QKV / FFN1
cd tools/bin
for r in 1 2 3; do
for t in "1,256,192" "1,256,384" "2,256,128" "0,256,192"; do
  for s in "12904 1920 1920 2" "12904 1920 7680 2"; do echo -n "round $r tile $t: "; ORV_GEMM_TILE=$t timeout 60 ./kbench_gemm bench $s 30; done
done
for t in "1,256,384" "2,256,256" "2,256,128"; do
  for s in "12904 5760 1920 0" "12904 7680 1920 1"; do echo -n "round $r tile $t: "; ORV_GEMM_TILE=$t timeout 60 ./kbench_gemm bench $s 30 2>&1 | tail -1; done
done
done
