#!/bin/bash
# timing probe for a TN main loop: fragment reads as 2 x ds_read_b64_tr_b16 + asm-issued LDS-DMA (-DORV_T8_TRPROBE, wrong results) vs ds_read_b128
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in flnone trprobe; do
  echo -n "$v 8192^3 : "; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
  echo -n "$v wgrad FFN2-like M=1920 N=7680 K=12928 : "; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 1920 7680 12928 0 3 3,256,256 | tail -1
  echo -n "$v wgrad QKV-like M=5760 N=1920 K=12928 : "; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 5760 1920 12928 0 3 3,256,192 | tail -1
  echo -n "$v wgrad FFN1-like M=7680 N=1920 K=12928 : "; LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 7680 1920 12928 0 3 3,256,192 | tail -1
done; done
} > ../../gpurun_out/t8_tr_probe.txt 2>&1
cat ../../gpurun_out/t8_tr_probe.txt
