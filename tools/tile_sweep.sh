#!/bin/bash
# GEMM tile candidates (ORV_GEMM_TILE="ring,bm,bn") against the cost-model choice in gemm.hip, in ONE run (box-to-box noise)
cd /root/repo/tools/bin
if [ "$1" = rates ]; then
  for t in 1,256,256 1,256,192 1,256,128 0,256,192 0,256,128 0,256,64 0,128,192 0,128,128 0,128,64; do
    echo -n "tile $t: "; ORV_GEMM_TILE=$t ./kbench_gemm bench 12904 7680 1920 0 10 | awk '{print $(NF-1), "TF"}'
  done
fi
for M in 3226 6452; do for s in "5760 1920 0" "1920 1920 2" "7680 1920 1" "1920 7680 2"; do
  echo "== M=$M N K epi = $s"
  echo -n "   chooser      : "; ./kbench_gemm bench $M $s 20 | awk '{print $(NF-3), "ms"}'
  for t in 1,256,256 1,256,192 0,256,192 0,256,128 0,128,192 0,128,128 0,128,64; do
    r=$(ORV_GEMM_TILE=$t ./kbench_gemm bench $M $s 20 2>/dev/null | grep bench | awk '{print $(NF-3)}'); [ -n "$r" ] && echo -n "   $t: $r ms;"
  done; echo
done; done
