#!/bin/bash
# correctness + timing of the phased GEMM (gemm_ph_kernel) against the ring kernel; run from the repo root on the GPU box
cd tools/bin
for t in "2,256,256" "2,256,128" "2,256,384"; do
  echo "== tile $t"
  export ORV_GEMM_TILE=$t
  bn=${t##*,}
  timeout 120 ./kbench_gemm check 700 $((bn*2)) 512 2 350 30 64
  timeout 120 ./kbench_gemm check 3226 $((bn*5)) 1920 1 3226 226 600
  timeout 120 ./kbench_gemm check 12904 $((bn*3)) 256 0 3226 226 600
  timeout 120 ./kbench_gemm check 13000 $((bn*4)) 128 2 3250 250 600
done
for t in "1,256,256" "2,256,256"; do
  export ORV_GEMM_TILE=$t; echo "== bench tile $t"
  for s in "4096 4096 4096 0" "8192 8192 8192 0" "12904 7680 1920 1" "12904 7680 1920 0" "3226 7680 1920 1"; do timeout 60 ./kbench_gemm bench $s 20; done
done
for t in "1,256,384" "2,256,384"; do
  export ORV_GEMM_TILE=$t; echo "== bench tile $t"
  for s in "12904 5760 1920 0" "12904 1920 1920 2" "12904 7680 1920 1" "12904 1920 7680 2"; do timeout 60 ./kbench_gemm bench $s 20; done
done
