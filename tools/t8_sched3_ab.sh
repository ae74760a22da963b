#!/bin/bash
# balanced four-phase schedule (-DORV_T8_SCHED3, BN = 256) vs the shipped one: correctness of the variant build, then standalone A/B
# needs: VARIANTS="sched3:-DORV_T8_SCHED3" bash tools/t8_variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
echo "== correctness of the sched3 build"
LD_LIBRARY_PATH=/root/repo/tools/bin/gv_sched3 ./kbench_gemm 2>&1 | grep -E "check|FAIL"
for s in "3226 3840 1920 0" "12904 7680 1920 1" "1000 768 256 1" "5000 512 4096 0" "777 1024 128 0"; do LD_LIBRARY_PATH=/root/repo/tools/bin/gv_sched3 ORV_GEMM_TILE=3,256,256 ./kbench_gemm check $s 3226 226 600; done
for r in 1 2; do for v in base sched3; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v FFN1 gelu : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 1 3 3,256,256 | tail -1
  echo -n "$v qk  plain : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 3840 1920 0 3 3,256,256 | tail -1
  echo -n "$v 4096^3    : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 4096 4096 4096 0 3 3,256,256 | tail -1
  echo -n "$v 8192^3    : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
done; done
} > ../../gpurun_out/t8_sched3_ab.txt 2>&1
cat ../../gpurun_out/t8_sched3_ab.txt
