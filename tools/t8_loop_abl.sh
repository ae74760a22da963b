#!/bin/bash
# K-loop ablation of gemm_t8_kernel (wrong results): no LDS-DMA / no fragment reads / no MFMAs / MFMAs + barriers only.  Standalone, random operands.
# needs: VARIANTS="nodma:-DORV_T8_ABL_NODMA noread:-DORV_T8_ABL_NOREAD nomfma:-DORV_T8_ABL_NOMFMA nodmaread:-DORV_T8_ABL_NODMA,-DORV_T8_ABL_NOREAD mfmaonly:-DORV_T8_ABL_NODMA,-DORV_T8_ABL_NOREAD,-DORV_T8_ABL_NOEPI" bash tools/t8_variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2; do for v in base nodma noread nomfma nodmaread mfmaonly; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v 8192^3 : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
  echo -n "$v FFN1   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 0 3 3,256,256 | tail -1
  echo -n "$v FFN2   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 7680 2 3 3,256,192 | tail -1
done; done
} > ../../gpurun_out/t8_loop_abl.txt 2>&1
cat ../../gpurun_out/t8_loop_abl.txt
