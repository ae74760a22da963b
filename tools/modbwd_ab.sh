#!/bin/bash
# mod_bwd_dgrad_kernel before / after (per-kernel average from rocprofv3 --kernel-trace --stats of a 3-step training bench), parity tests first
cd /root/repo; mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_backward.py -m gpu -x -q -k "modulation or tables or mod" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for n in attn_head base; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  rm -rf /tmp/mb_$n; ORV_LIB=$L rocprofv3 --kernel-trace --stats -d /tmp/mb_$n -o mb -- python /root/repo/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > /tmp/mb_$n.log 2>&1
  echo "== $n"; f=$(find /tmp/mb_$n -name "*kernel_stats.csv" | head -1); grep -h "mod_bwd\|mod_tables" $f | cut -c1-200
  grep '^{' /tmp/mb_$n.log | tail -1 | python -c "import json,sys; print('ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"
done
} > gpurun_out/modbwd_ab.txt 2>&1
cat gpurun_out/modbwd_ab.txt
