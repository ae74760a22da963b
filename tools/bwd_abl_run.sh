#!/bin/bash
# per-kernel durations of the attention-backward ablation builds (rocprofv3 kernel-trace stats of tools/time_attn_bwd.py)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/bwd_abl; rm -rf $O; mkdir -p $O
for v in $(ls /root/repo/tools/bin | grep "^bw_" | sed "s/bw_//"); do
  ORV_LIB=/root/repo/tools/bin/bw_$v/liborv_mi355.so rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $v -- python /root/repo/tools/time_attn_bwd.py > $O/log_$v.txt 2>&1
  python3 - "$O/${v}_kernel_stats.csv" $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_bwd" in r["Name"]: print(f"{sys.argv[2]:10s} {r['Name'][:60]:60s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done | tee $O/summary.txt
find $O -name "*.csv" -delete
