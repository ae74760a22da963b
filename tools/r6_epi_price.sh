#!/bin/bash
# VERDICT r5 #1 priced: (a) the d8 kernel WITHOUT its epilogue (tools/bin/dv_noepi, wrong results) = the ceiling of any scheme that hides the
# epilogue perfectly; (b) the K loop of the narrower tiles a two-workgroups-per-CU scheme would have to use (256 x 128 instead of 256 x 192: four W
# buffers of 192 columns + scratch leave no room for a second workgroup's LDS) - forced tiles, same process.  Standalone, random operands.
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do
  for v in base noepi; do
    L=/root/repo/tools/bin/dv_$v; [ $v = base ] && L=/root/repo/orv_amd
    for shape in "1920 7680 2 192" "1920 1920 2 192" "1920 7680 2 128" "1920 1920 2 128"; do
      set -- $shape
      echo -n "$v N=$1 K=$2 epi=$3 bn=$4 : "; KB_RM_FREE=1 LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 $1 $2 $3 3 $4 | grep "d8 packed" | sed 's/.*: median/median/'; 
    done
  done
done
} > ../../gpurun_out/r6_epi_price.txt 2>&1
cat ../../gpurun_out/r6_epi_price.txt
