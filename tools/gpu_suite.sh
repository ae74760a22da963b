#!/bin/bash
# the GPU test suite with its full output kept (-> gpurun_out/pytest_gpu_full.txt)
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -v > gpurun_out/pytest_gpu_full.txt 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu_full.txt
grep -n "Fatal\|Segmentation\|Abort\|rc=\|passed\|failed" gpurun_out/pytest_gpu_full.txt | head; tail -5 gpurun_out/pytest_gpu_full.txt | cut -c1-300
