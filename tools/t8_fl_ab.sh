#!/bin/bash
# full-line LDS-DMA (8 rows x 128 B per instruction, -DORV_T8_FULLLINE) vs the st_16x32 one (16 rows x 64 B): parity, then standalone A/B
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for t in 3,256,256 3,256,192; do
for s in "4096 7680 4096 0 4096 0 0" "700 768 512 2 350 30 64" "3226 7680 1920 1 3226 226 600" "12904 1920 1920 2 3226 226 600" "12904 5760 1920 0 3226 226 600" "6452 3840 1920 3 3226 226 600"; do
  echo -n "check base $t $s: "; LD_LIBRARY_PATH=/root/repo/orv_amd ORV_GEMM_TILE=$t timeout 120 ./kbench_gemm check $s < /dev/null | tail -1
done; done
for r in 1 2; do for v in head base flnone flall; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v 8192^3 : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
  echo -n "$v FFN1   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 1 3 3,256,256 | tail -1
  echo -n "$v FFN2   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 7680 2 3 3,256,192 | tail -1
  echo -n "$v QKV    : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 5760 1920 0 3 3,256,192 | tail -1
  echo -n "$v out    : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 1920 2 3 3,256,192 | tail -1
done; done
} > ../../gpurun_out/t8_fl_ab.txt 2>&1
cat ../../gpurun_out/t8_fl_ab.txt
