#!/bin/bash
# build first: VARIANTS="nostore:-DORV_T8_ABL_NOSTORE noepi:-DORV_T8_ABL_NOEPI" bash tools/t8_variants.sh
# what does a tile's epilogue cost the t8 kernel?  ablation builds (wrong results): every store predicated off / no epilogue at all;
# plus a K sweep of the shipped kernel (time = tiles x (a + b K): a = per-tile fixed cost).  Standalone, interleaved, random operands.
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in base nostore noepi; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v FFN1 gelu : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 1 1 3,256,256 | tail -1
  echo -n "$v FFN1 plain: "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 0 1 3,256,256 | tail -1
  echo -n "$v outproj   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 1920 2 1 3,256,192 | tail -1
  echo -n "$v FFN2      : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 7680 2 1 3,256,192 | tail -1
done; done
for K in 640 1280 1920 3840 7680; do
  echo -n "base K sweep N=7680 K=$K: "; LD_LIBRARY_PATH=/root/repo/orv_amd ./kbench_gemm ab 12904 7680 $K 0 3 3,256,256 | tail -1
  echo -n "base K sweep N=1920 K=$K: "; LD_LIBRARY_PATH=/root/repo/orv_amd ./kbench_gemm ab 12904 1920 $K 2 3 3,256,192 | tail -1
done
} > ../../gpurun_out/t8_epi_abl.txt 2>&1
cat ../../gpurun_out/t8_epi_abl.txt
