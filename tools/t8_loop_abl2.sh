#!/bin/bash
# load-side ablation of gemm_t8_kernel without MFMAs (wrong results): fragment reads alone, LDS-DMA alone, both.  Standalone, random operands.
# needs: FILE=gemm_t8.hip VARIANTS="nomfma:-DORV_T8_ABL_NOMFMA nomfma_nodma:-DORV_T8_ABL_NOMFMA,-DORV_T8_ABL_NODMA nomfma_noread:-DORV_T8_ABL_NOMFMA,-DORV_T8_ABL_NOREAD" bash tools/variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2; do for v in base nomfma nomfma_nodma nomfma_noread; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v 8192^3 : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
  echo -n "$v FFN2   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 7680 2 3 3,256,192 | tail -1
done; done
} > ../../gpurun_out/t8_loop_abl2.txt 2>&1
cat ../../gpurun_out/t8_loop_abl2.txt
