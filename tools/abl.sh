#!/bin/bash
# A/B ablation of the ping-pong GEMM: full / no main-loop DMA / no MFMA (separate builds of the same source)
cd /root/repo/tools/bin
for v in FULL NOLOAD NOMFMA; do
  mkdir -p /tmp/abl_$v; 
  if [ $v = FULL ]; then cp /root/repo/orv_amd/liborv_mi355.so /tmp/abl_$v/liborv_mi355.so; else cp abl/lib_$v.so /tmp/abl_$v/liborv_mi355.so; fi
  echo "== $v"; LD_LIBRARY_PATH=/tmp/abl_$v ./kbench_gemm bench 12904 7680 1920 1 10; LD_LIBRARY_PATH=/tmp/abl_$v ./kbench_gemm bench 12904 1920 7680 2 10; LD_LIBRARY_PATH=/tmp/abl_$v ./kbench_gemm bench 8192 8192 8192 0 5
done
