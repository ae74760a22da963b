#!/bin/bash
# attention backward dK / dV pass with the shared fragment buffer (tools/bin/gv_fragbuf: -DORV_BW_FRAGBUF) against the in-tree build in the SFT step
# (configs[2], 12 timed steps) and the checkpointed 5B step, same box, interleaved
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2 3; do for n in base fragbuf; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$n 2B train : "; ORV_LIB=$L python bench.py --mode train --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])"
done; done
for n in base fragbuf base fragbuf; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$n 5B ckpt  : "; ORV_LIB=$L python bench.py --mode train --model 5b --grad-ckpt --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])"
done
echo -n "training tests with fragbuf: "; ORV_LIB=/root/repo/tools/bin/gv_fragbuf/liborv_mi355.so timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py -x -q 2>&1 | tail -1
} > gpurun_out/r6_fragbuf_train_ab.txt 2>&1
cat gpurun_out/r6_fragbuf_train_ab.txt
