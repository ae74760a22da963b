#!/bin/bash
# LayerNorm-modulate variants (tools/bin/gv_<name>, built by tools/variants.sh FILE=norm.hip) in the headline step: per-launch duration of
# ln_mod_rows_kernel from rocprofv3 kernel-trace stats + the un-profiled step time, interleaved on one box.  usage: r6_ln_ab.sh name...
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
{
for r in 1 2; do for n in "$@"; do
  L=/root/repo/tools/bin/gv_$n/liborv_mi355.so; [ "$n" = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  O=/tmp/prof_ln_$n; rm -rf $O
  ORV_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python /root/repo/bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 10 --warmup 3 > /dev/null 2>&1
  K=$(python3 - "$O/p_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ln_mod_rows_kernel" in r["Name"]:
        print("ln_mod_rows calls %s avg_us %.2f" % (r["Calls"], float(r["AverageNs"]) / 1e3)); break
PY
)
  S=$(ORV_LIB=$L python /root/repo/bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$n : step $S ms | $K"
done; done
} > /root/repo/gpurun_out/r6_ln_ab.txt 2>&1
cat /root/repo/gpurun_out/r6_ln_ab.txt
