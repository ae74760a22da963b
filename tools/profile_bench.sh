#!/bin/bash
# rocprofv3 kernel-trace stats of the headline bench command (copied into profiles/ by the developer).
# usage: bash tools/profile_bench.sh <tag> [bench args...]
TAG=${1:-r1}; shift
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python /root/repo/bench.py --no-cpu-baseline "$@" > $OUT/bench_stdout.log 2>&1
python3 - "$OUT/${TAG}_kernel_stats.csv" > $OUT/${TAG}_kernel_stats_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[:25]:
    name = r["Name"]
    if len(name) > 100: name = name[:97] + "..."
    print(f"{name:100s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
PY
cat $OUT/${TAG}_kernel_stats_summary.txt | head -16; tail -1 $OUT/bench_stdout.log | cut -c1-300
