#!/bin/bash
# what bounds the LDS-DMA stream of gemm_t8_kernel: DMA-only builds (no MFMAs, no fragment reads; wrong results), full-line vs st_16x32 pieces,
# all workgroups on one L2-resident panel pair (ORV_GEMM_DBG=77), and a reduced number of persistent workgroups (ORV_T8_GRID)
cd /root/repo/tools/bin; mkdir -p ../../gpurun_out
{
for v in fl_dmaonly st_dmaonly; do for dbg in 0 77; do for grid in 0 128 64 32 8; do
  echo -n "$v dbg=$dbg grid=$grid 8192^3 : "; ORV_GEMM_DBG=$dbg ORV_T8_GRID=$grid LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 8192 8192 8192 0 3 3,256,256 | tail -1
done; done; done
for v in fl_dmaonly st_dmaonly; do for dbg in 0 77; do
  echo -n "$v dbg=$dbg FFN2 : "; ORV_GEMM_DBG=$dbg LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 12904 1920 7680 2 3 3,256,192 | tail -1
  echo -n "$v dbg=$dbg FFN1 : "; ORV_GEMM_DBG=$dbg LD_LIBRARY_PATH=/root/repo/tools/bin/gv_$v ./kbench_gemm ab 12904 7680 1920 1 3 3,256,256 | tail -1
done; done
} > ../../gpurun_out/t8_dma_probe.txt 2>&1
cat ../../gpurun_out/t8_dma_probe.txt
