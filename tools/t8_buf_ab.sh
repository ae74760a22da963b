#!/bin/bash
# buffer-form LDS-DMA (-DORV_T8_BUFLDS) vs the global form in gemm_t8_kernel: parity (kbench check), standalone A/B on the model shapes, the
# DMA-only ablation of both.  needs: FILE=gemm_t8.hip VARIANTS="buf:-DORV_T8_BUFLDS buf_dmaonly:-DORV_T8_BUFLDS,-DORV_T8_ABL_NOMFMA,-DORV_T8_ABL_NOREAD
#   nomfma_noread:-DORV_T8_ABL_NOMFMA,-DORV_T8_ABL_NOREAD" bash tools/variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for t in 3,256,256 3,256,192; do
for s in "4096 7680 4096 0 4096 0 0" "700 768 512 2 350 30 64" "3226 7680 1920 1 3226 226 600" "12904 1920 1920 2 3226 226 600" "12904 5760 1920 0 3226 226 600"; do
  echo -n "check buf $t $s: "; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_buf ORV_GEMM_TILE=$t timeout 120 ./kbench_gemm check $s < /dev/null | tail -1
done; done
for r in 1 2 3; do for v in base buf; do
  L=/root/repo/tools/bin/gv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v 8192^3 : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
  echo -n "$v FFN1   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 7680 1920 1 3 3,256,256 | tail -1
  echo -n "$v FFN2   : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 7680 2 3 3,256,192 | tail -1
  echo -n "$v QKV    : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 5760 1920 0 3 3,256,192 | tail -1
  echo -n "$v out    : "; LD_LIBRARY_PATH=$L ./kbench_gemm ab 12904 1920 1920 2 3 3,256,192 | tail -1
done; done
for v in nomfma_noread buf_dmaonly nomfma_noread buf_dmaonly; do
  echo -n "$v 8192^3 : "; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
  echo -n "$v FFN2   : "; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 12904 1920 7680 2 3 3,256,192 | tail -1
done
} > ../../gpurun_out/t8_buf_ab.txt 2>&1
cat ../../gpurun_out/t8_buf_ab.txt
