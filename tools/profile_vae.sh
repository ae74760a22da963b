#!/bin/bash
# rocprofv3 kernel-trace stats of a full-size VAE decode + encode (tools/vae_bench.py).  usage: bash tools/profile_vae.sh <tag> [B] [iters]
TAG=${1:-r2}; shift
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_vae_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python /root/repo/tools/vae_bench.py "$@" > $OUT/stdout.log 2>&1 < /dev/null
python3 - "$OUT/${TAG}_kernel_stats.csv" > $OUT/${TAG}_vae_kernel_stats_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[:25]:
    name = r["Name"]
    if len(name) > 100: name = name[:97] + "..."
    print(f"{name:100s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
PY
find $OUT -name "*.csv" -size +4M -delete
cat $OUT/${TAG}_vae_kernel_stats_summary.txt | head -24; tail -2 $OUT/stdout.log
