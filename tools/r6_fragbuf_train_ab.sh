#!/bin/bash
# attention backward dK / dV pass: the shipped shared fragment buffer against the round-5 form (tools/bin/gv_nofrag: -DORV_BW_NO_FRAGBUF) -
# parity tests on the shipped build, the pair standalone, then the SFT step (configs[2], 12 timed steps), same box, interleaved
cd /root/repo; mkdir -p gpurun_out; export PYTHONPATH=/root/repo
{
echo -n "tests (shipped build): "; timeout 1800 python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py tests/test_gpu_kernels.py -x -q -k "not deterministic" 2>&1 | tail -1
for r in 1 2 3; do for n in nofrag base; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$n pair forked: "; ORV_LIB=$L timeout 120 python tools/time_attn_bwd.py 2>/dev/null
done; done
for r in 1 2 3 4; do for n in nofrag base; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$n 2B train : "; ORV_LIB=$L python bench.py --mode train --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])"
done; done
} > gpurun_out/r6_fragbuf_train_ab.txt 2>&1
cat gpurun_out/r6_fragbuf_train_ab.txt
