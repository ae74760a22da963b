#!/bin/bash
# attention backward: dQ and dK/dV passes forked onto two streams vs back to back (interleaved), then the backward tests and a training leg
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2 3; do
  echo -n "back to back: "; ORV_ATTN_BWD_FORK=0 python tools/time_attn_bwd.py
  echo -n "forked      : "; ORV_ATTN_BWD_FORK=1 python tools/time_attn_bwd.py
done
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py -x -q 2>&1 | tail -3
for f in 0 1; do echo "train FORK=$f"; ORV_ATTN_BWD_FORK=$f python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('config'))"; done
} > gpurun_out/bwd_fork.txt 2>&1
cat gpurun_out/bwd_fork.txt
