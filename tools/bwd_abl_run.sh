#!/bin/bash
# per-kernel durations of the attention-backward ablation builds (rocprofv3 kernel-trace stats of tools/time_attn_bwd.py, passes back to back)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/bwd_abl; rm -rf $O; mkdir -p $O
for v in full noy notr nob128 nolds nomfma nodma mfmaonly full; do
  ORV_ATTN_BWD_FORK=0 ORV_LIB=/root/repo/tools/bin/bw_$v/liborv_mi355.so rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $v -- python /root/repo/tools/time_attn_bwd.py > $O/log_$v.txt 2>&1
  python3 - "$O/${v}_kernel_stats.csv" $v <<'PY'
import csv, sys
d = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_bwd_dq" in r["Name"]: d["dq"] = float(r["AverageNs"]) / 1e3
    if "attn_bwd_dkv" in r["Name"]: d["dkv"] = float(r["AverageNs"]) / 1e3
print(f"{sys.argv[2]:10s} dq {d.get('dq', 0):8.1f} us   dkv {d.get('dkv', 0):8.1f} us")
PY
done | tee $O/summary.txt
find $O -name "*.csv" -delete
