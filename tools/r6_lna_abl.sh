#!/bin/bash
# VERDICT r5 #3 priced: the d8 K loop with the LayerNorm-modulate arithmetic applied to every A element on its way into the MFMA
# (tools/bin/dv_lna = -DORV_D8_ABL_LNA, identity constants: right results) against the shipped kernel, standalone, interleaved, random operands.
# Shapes: q | k | v (N = 5760) and FFN1 (N = 7680) - the two GEMMs that would consume the un-normalised residual stream - and FFN2 for scale.
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in base lna; do
  L=/root/repo/tools/bin/dv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v qkv   : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 5760 1920 0 3 192 | grep "d8 packed\|MISMATCH" | tr "\n" " "; echo
  echo -n "$v FFN1  : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 7680 1920 1 3 256 | grep "d8 packed\|MISMATCH" | tr "\n" " "; echo
  echo -n "$v FFN2  : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 7680 0 3 192 | grep "d8 packed\|MISMATCH" | tr "\n" " "; echo
done; done
} > ../../gpurun_out/r6_lna_abl.txt 2>&1
cat ../../gpurun_out/r6_lna_abl.txt
