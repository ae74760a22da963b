cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -2
bash tools/model_ab.sh ORV_GEMM_GM 4 0
bash tools/train_ab.sh ORV_GEMM_GM 4 0 | tail -4
