#!/bin/bash
# t8 kernel: correctness (pytest), then same-process interleaved A/B on the CogVideoX-2B GEMM shapes (B = 4 and B = 1)
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "t8 or (forced and 3_256)" 2>&1 | tail -5 > gpurun_out/t8_pytest.txt; cat gpurun_out/t8_pytest.txt
python -m pytest tests/test_gpu_kernels.py -x -q -k "forced" 2>&1 | tail -3 >> gpurun_out/t8_pytest.txt; tail -3 gpurun_out/t8_pytest.txt
cd tools/bin
{
./kbench_gemm ab 12904 7680 1920 1 ${1:-5} 3,256,256 2,256,256 1,256,384
./kbench_gemm ab 12904 5760 1920 0 ${1:-5} 3,256,192 1,256,384 1,256,192
./kbench_gemm ab 12904 1920 7680 2 ${1:-5} 3,256,192 1,256,384 1,256,192
./kbench_gemm ab 12904 1920 1920 2 ${1:-5} 3,256,192 1,256,192 0,256,192
./kbench_gemm ab 12904 3840 1920 0 ${1:-5} 3,256,256 3,256,192 1,256,384
./kbench_gemm ab 3226 7680 1920 1 ${1:-5} 3,256,256 2,256,256 0,192,128
./kbench_gemm ab 3226 1920 7680 2 ${1:-5} 3,256,192 0,192,128
./kbench_gemm ab 3226 5760 1920 0 ${1:-5} 3,256,192 0,192,128 0,128,192
timeout 300 ./probe_gemm_template 3
} > ../../gpurun_out/t8_ab.txt 2>&1
cat ../../gpurun_out/t8_ab.txt
