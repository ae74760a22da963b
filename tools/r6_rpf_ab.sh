#!/bin/bash
# residual-row prefetch of the d8 gated-residual epilogue (tools/bin/dv_rpf<n> = -DORV_D8_RPF=<n>: dummy loads n trips of three K-tiles before the
# K loop ends) against the shipped kernel: standalone (FFN2 / out-projection, epilogue 2), then in the headline step.  interleaved, one box.
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2 3; do for v in base rpf1 rpf2 rpf4; do
  L=/root/repo/tools/bin/dv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v FFN2 : "; KB_RM_FREE=1 LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 7680 2 3 192 | grep "d8 packed\|MISMATCH" | sed 's/.*: median/median/' | tr "\n" " "; echo
  echo -n "$v out  : "; KB_RM_FREE=1 LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 1920 2 3 192 | grep "d8 packed\|MISMATCH" | sed 's/.*: median/median/' | tr "\n" " "; echo
done; done
cd /root/repo
for r in 1 2 3; do for v in base rpf1 rpf2 rpf4; do
  L=/root/repo/tools/bin/dv_$v/liborv_mi355.so; [ $v = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$v : "; ORV_LIB=$L python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python tools/bench_line_brief.py
done; done
} > /root/repo/gpurun_out/r6_rpf_ab.txt 2>&1
cat /root/repo/gpurun_out/r6_rpf_ab.txt
