cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -2
for r in 1 2 3; do
  echo -n "new fork=0: "; ORV_ATTN_BWD_FORK=0 python tools/time_attn_bwd.py 2>/dev/null | tail -1
  echo -n "new fork=1: "; ORV_ATTN_BWD_FORK=1 python tools/time_attn_bwd.py 2>/dev/null | tail -1
  echo -n "old fork=1: "; ORV_LIB=/root/repo/tools/bin/av_base/liborv_mi355.so ORV_ATTN_BWD_FORK=1 python tools/time_attn_bwd.py 2>/dev/null | tail -1
done
