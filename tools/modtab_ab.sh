#!/bin/bash
# modulation tables: LDS-staged kernel + batched conditioning loads vs the previous build (tools/bin/gv_attn_head), same box
cd /root/repo; mkdir -p gpurun_out
fmt='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
{
python -m pytest tests -m gpu -x -q -k "modulation or mod_tables or tables" 2>&1 | tail -2
for r in 1 2 3; do
  echo -n "prev lib        : "; ORV_LIB=/root/repo/tools/bin/gv_attn_head/liborv_mi355.so python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"
  echo -n "new, direct     : "; ORV_MOD_TABLES_LDS=0 python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"
  echo -n "new, LDS-staged : "; python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  ORV_MOD_TABLES_LDS=$v rocprofv3 --kernel-trace --stats -d /tmp/mt_prof_$v -o mt -- python /root/repo/bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --eager --steps 5 --warmup 2 > /dev/null 2>&1
  echo "LDS=$v:"; grep -h "mod_tables" /tmp/mt_prof_$v/*/*kernel_stats.csv /tmp/mt_prof_$v/*kernel_stats.csv 2>/dev/null | cut -c1-160
done
} > gpurun_out/modtab_ab.txt 2>&1
cat gpurun_out/modtab_ab.txt
