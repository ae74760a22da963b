#!/bin/bash
# Are the t8 kernel's epilogue stores slow because all 256 persistent workgroups store at the same moment?  (a) start stagger of
# workgroup groups (ORV_T8_STAGGER=groups,ns), (b) half the workgroups (ORV_T8_GRID=128) for the shipped and the no-store build.
# needs: VARIANTS="nostore:-DORV_T8_ABL_NOSTORE" bash tools/t8_variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
run() { echo -n "$1 : "; shift; env "$@" ./kbench_gemm ab $SHAPE 1 $TILE | tail -1; }
{
for r in 1 2 3; do
  for sh in "12904 7680 1920 1|3,256,256" "12904 3840 1920 0|3,256,256" "12904 1920 1920 2|3,256,192" "12904 1920 7680 2|3,256,192"; do
    SHAPE=${sh%%|*}; TILE=${sh##*|}
    echo "== $SHAPE"
    run "base          " LD_LIBRARY_PATH=/root/repo/orv_amd
    run "stagger 2x3000" LD_LIBRARY_PATH=/root/repo/orv_amd ORV_T8_STAGGER=2,3000
    run "stagger 4x1500" LD_LIBRARY_PATH=/root/repo/orv_amd ORV_T8_STAGGER=4,1500
    run "stagger 4x2500" LD_LIBRARY_PATH=/root/repo/orv_amd ORV_T8_STAGGER=4,2500
    run "stagger 8x1000" LD_LIBRARY_PATH=/root/repo/orv_amd ORV_T8_STAGGER=8,1000
    run "stagger 8x2000" LD_LIBRARY_PATH=/root/repo/orv_amd ORV_T8_STAGGER=8,2000
    run "nostore       " LD_LIBRARY_PATH=/root/repo/tools/bin/gv_nostore
    run "grid128 base  " LD_LIBRARY_PATH=/root/repo/orv_amd ORV_T8_GRID=128
    run "grid128 nostor" LD_LIBRARY_PATH=/root/repo/tools/bin/gv_nostore ORV_T8_GRID=128
  done
done
} > ../../gpurun_out/t8_stagger.txt 2>&1
cat ../../gpurun_out/t8_stagger.txt
