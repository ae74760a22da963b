#!/bin/bash
# same-box interleaved A/B: previous build (tools/bin/abl_ph/OLD) vs current library on the phased kernel
cd tools/bin
export ORV_GEMM_TILE=2,256,256
for r in 1 2 3; do for v in OLD NEW; do
  if [ $v = NEW ]; then L=../../orv_amd; else L=abl_ph/OLD; fi
  for s in "4096 4096 4096 0" "12904 7680 1920 1"; do echo -n "$v: "; LD_LIBRARY_PATH=$L timeout 60 ./kbench_gemm bench $s 30; done
done; done
