#!/bin/bash
# d8 kernel: the tree's build against the previous commit's gemm_d8.hip (tools/bin/dv_prev), epilogue-2 shapes, standalone, interleaved; parity first
cd /root/repo; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "d8 or packed" 2>&1 | tail -2
cd tools/bin
for r in 1 2 3; do for v in prev base; do
  L=/root/repo/tools/bin/dv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v FFN2 : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 7680 2 3 192 | grep "d8 packed"
  echo -n "$v out  : "; LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 12904 1920 1920 2 3 192 | grep "d8 packed"
  echo -n "$v out1 : "; KB_RM_FREE=1 LD_LIBRARY_PATH=$L timeout 120 ./kbench_gemm abp 3226 1920 1920 2 3 128 | grep "d8 packed"
done; done
} > gpurun_out/d8_prev_ab.txt 2>&1
cat gpurun_out/d8_prev_ab.txt
