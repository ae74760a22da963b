#!/bin/bash
# LDS-transposed epilogue (shipped) vs the register-layout epilogue (-DORV_T8_EPI_DIRECT) vs no stores / no epilogue, standalone,
# interleaved, random operands.  needs: VARIANTS="epidirect:-DORV_T8_EPI_DIRECT nostore:-DORV_T8_ABL_NOSTORE noepi:-DORV_T8_ABL_NOEPI" bash tools/t8_variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in base epidirect nostore noepi; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v FFN1 gelu : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 1 1 3,256,256 | tail -1
  echo -n "$v qk  plain : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 3840 1920 0 1 3,256,256 | tail -1
  echo -n "$v v   plain : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 1920 0 1 3,256,192 | tail -1
  echo -n "$v outproj   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 1920 2 1 3,256,192 | tail -1
  echo -n "$v FFN2      : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 7680 2 1 3,256,192 | tail -1
done; done
for st in 0 "4,1200" "4,2000" "8,600"; do
  echo -n "base stagger=$st FFN1 gelu : "; LD_LIBRARY_PATH=/root/repo/orv_amd ORV_T8_STAGGER=$st ./kbench_gemm ab 12904 7680 1920 1 3 3,256,256 | tail -1
  echo -n "base stagger=$st qk        : "; LD_LIBRARY_PATH=/root/repo/orv_amd ORV_T8_STAGGER=$st ./kbench_gemm ab 12904 3840 1920 0 3 3,256,256 | tail -1
done
LD_LIBRARY_PATH=/root/repo/orv_amd ./kbench_gemm 2>&1 | grep -E "check|FAIL"
} > ../../gpurun_out/t8_epi_ab.txt 2>&1
cat ../../gpurun_out/t8_epi_ab.txt
