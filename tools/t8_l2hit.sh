#!/bin/bash
# is the t8 K-loop bound by L2 misses?  ORV_GEMM_DBG=77: every workgroup streams the operand panels of tile (0, 0) (all L2 hits after the
# first touch; wrong results) against the shipped walk; standalone, 3 rounds each
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for r in 1 2; do for d in 0 77; do
  echo -n "DBG=$d FFN1  : "; LD_LIBRARY_PATH=/root/repo/orv_amd ORV_GEMM_DBG=$d ./kbench_gemm ab 12904 7680 1920 0 3 3,256,256 | tail -1
  echo -n "DBG=$d 8192^3: "; LD_LIBRARY_PATH=/root/repo/orv_amd ORV_GEMM_DBG=$d ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
  echo -n "DBG=$d FFN2  : "; LD_LIBRARY_PATH=/root/repo/orv_amd ORV_GEMM_DBG=$d ./kbench_gemm ab 12904 1920 7680 2 3 3,256,192 | tail -1
done; done
} > ../../gpurun_out/t8_l2hit.txt 2>&1
cat ../../gpurun_out/t8_l2hit.txt
