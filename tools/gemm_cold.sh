#!/bin/bash
# warm vs cold operands for the per-block GEMM shapes (standalone); ORV_GEMM_DBG=64 walks the tile list backwards
cd /root/repo/tools/bin; export LD_LIBRARY_PATH=/root/repo/orv_amd; mkdir -p ../../gpurun_out
{ for r in 1 2; do for d in 0 64; do echo "ORV_GEMM_DBG=$d"; ORV_GEMM_DBG=$d ./kbench_gemm cold 12904 1920 7680 2 32 | grep "one operand\|rewritten\|only A" | tail -3; done; done; } > ../../gpurun_out/gemm_cold.txt 2>&1
cat ../../gpurun_out/gemm_cold.txt
