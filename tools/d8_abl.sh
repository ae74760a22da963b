#!/bin/bash
# K-loop ablation of gemm_d8_kernel (wrong results): no fragment reads / no LDS-DMA / no A loads / no MFMAs.  Standalone, random operands, packed A.
# needs: VARIANTS="noread:-DORV_D8_ABL_NOREAD nodma:-DORV_D8_ABL_NODMA noa:-DORV_D8_ABL_NOA nomfma:-DORV_D8_ABL_NOMFMA" bash tools/d8_variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2; do for v in base noread nodma noa ${EXTRA}; do
  L=/root/repo/tools/bin/dv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v FFN1  : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 7680 1920 0 3 256 | grep "d8 packed\|MISMATCH" | tr "\n" " "; echo
  echo -n "$v FFN2  : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 7680 0 3 192 | grep "d8 packed\|MISMATCH" | tr "\n" " "; echo
done; done
} > ../../gpurun_out/d8_abl.txt 2>&1
cat ../../gpurun_out/d8_abl.txt
