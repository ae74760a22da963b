#!/bin/bash
# two-phase K-tile schedule (-DORV_T8_SCHED2) vs the shipped four-phase one; standalone, interleaved, random operands, 3 rounds per call
# needs: VARIANTS="sched2:-DORV_T8_SCHED2 epidirect:-DORV_T8_EPI_DIRECT sched2d:-DORV_T8_SCHED2,-DORV_T8_EPI_DIRECT" bash tools/t8_variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
echo "== correctness of the sched2 build"
LD_LIBRARY_PATH=/root/repo/tools/bin/gv_sched2 ./kbench_gemm 2>&1 | grep -E "check|FAIL"
for t in 3,256,256 3,256,192; do LD_LIBRARY_PATH=/root/repo/tools/bin/gv_sched2 ORV_GEMM_TILE=$t ./kbench_gemm check 3226 3840 1920 0 3226 226 600; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_sched2 ORV_GEMM_TILE=$t ./kbench_gemm check 12904 1920 7680 2 3226 226 600; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_sched2 ORV_GEMM_TILE=$t ./kbench_gemm check 1000 768 256 1 500 20 100; done
for r in 1 2; do for v in base sched2 epidirect sched2d; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v FFN1 gelu : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 1 3 3,256,256 | tail -1
  echo -n "$v qk  plain : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 3840 1920 0 3 3,256,256 | tail -1
  echo -n "$v 4096^3    : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 4096 4096 4096 0 3 3,256,256 | tail -1
  echo -n "$v v   plain : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 1920 0 3 3,256,192 | tail -1
  echo -n "$v outproj   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 1920 2 3 3,256,192 | tail -1
  echo -n "$v FFN2      : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 7680 2 3 3,256,192 | tail -1
done; done
} > ../../gpurun_out/t8_sched_ab.txt 2>&1
cat ../../gpurun_out/t8_sched_ab.txt
