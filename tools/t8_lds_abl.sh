#!/bin/bash
# build first: VARIANTS="nob23:-DORV_T8_ABL_NOB23 noa1:-DORV_T8_ABL_NOA1 noboth:-DORV_T8_ABL_NOB23,-DORV_T8_ABL_NOA1" bash tools/t8_variants.sh
# is the 192-wide t8 kernel bound by its LDS fragment reads?  ablation builds (wrong results): no B-block-2 read (22 -> 20 reads per
# K-tile and wave), no second A-half read (22 -> 14), both (12); standalone, interleaved
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in base nob23 noa1 noboth; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 7680 2 1 3,256,192 | tail -1
  echo -n "$v : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 1920 2 1 3,256,192 | tail -1
  echo -n "$v : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 1 1 3,256,256 | tail -1
done; done
} > ../../gpurun_out/t8_lds_abl.txt 2>&1
cat ../../gpurun_out/t8_lds_abl.txt
