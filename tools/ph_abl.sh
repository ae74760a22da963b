#!/bin/bash
# phased GEMM ablation: FULL vs NODMA vs NOMFMA builds (tools/ph_abl_build.sh), same shapes
cd tools/bin
export ORV_GEMM_TILE=${1:-2,256,256}
for v in FULL NODMA NOMFMA FULL; do
  echo "== $v tile $ORV_GEMM_TILE"
  if [ $v = FULL ]; then L=../../orv_amd; else L=abl_ph/$v; fi
  for s in "4096 4096 4096 0" "8192 8192 8192 0" "12904 7680 1920 0"; do LD_LIBRARY_PATH=$L timeout 60 ./kbench_gemm bench $s 20; done
done
