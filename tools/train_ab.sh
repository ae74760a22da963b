#!/bin/bash
# backward / training tests, then the 2B training leg with one switch at two values (same box, interleaved)
# usage: train_ab.sh ENVVAR [VAL0 VAL1]
cd /root/repo; mkdir -p gpurun_out
V0=${2:-0}; V1=${3:-1}
{
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py -x -q 2>&1 | tail -3
for r in 1 2; do for f in $V0 $V1; do echo -n "train $1=$f : "; env $1=$f python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done
} > gpurun_out/train_ab.txt 2>&1
cat gpurun_out/train_ab.txt
