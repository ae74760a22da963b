#!/bin/bash
# per-tile timeline of the t8 kernel: builds (locally, before gpurun) via
#   VARIANTS="trace1:-DORV_T8_TRACE=1 trace2:-DORV_T8_TRACE=2 trace1d:-DORV_T8_TRACE=1,-DORV_T8_EPI_DIRECT" bash tools/t8_variants.sh
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for v in trace1 trace2 trace1d; do
  echo "=== $v"
  LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ORV_GEMM_TILE=3,256,256 ./trace_t8 12904 7680 1920 1
  LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ORV_GEMM_TILE=3,256,256 ./trace_t8 12904 7680 1920 0 | head -12
done
} > ../../gpurun_out/t8_trace.txt 2>&1
cat ../../gpurun_out/t8_trace.txt
