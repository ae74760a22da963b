#!/bin/bash
# training step with the TN weight-gradient GEMM (ORV_WGRAD_TN=1) against the transposes + NT path, same box, interleaved; gradient tests first
cd /root/repo; mkdir -p gpurun_out; export PYTHONPATH=/root/repo
{
ORV_WGRAD_TN=1 timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do for v in 0 1; do
  echo -n "ORV_WGRAD_TN=$v : "; ORV_WGRAD_TN=$v timeout 300 python bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
} > gpurun_out/train_tn_ab.txt 2>&1
cat gpurun_out/train_tn_ab.txt
