cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py -x -q 2>&1 | tail -3
for r in 1 2; do
  echo -n "train new : "; python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  echo -n "train old : "; ORV_LIB=/root/repo/tools/bin/av_base/liborv_mi355.so python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
