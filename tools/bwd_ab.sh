#!/bin/bash
# attention backward A/B: VARIANTS="name name ..." (tools/bin/dv_<name>/liborv_mi355.so; "base" = the tree's build), interleaved;
# per-kernel times from rocprofv3 with the passes back to back (ORV_ATTN_BWD_FORK=0) and the forked pair by events
cd /root/repo; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
VARIANTS=${VARIANTS:-"base"}
{
for r in 1 2; do for v in $VARIANTS; do
  L=/root/repo/tools/bin/dv_$v/liborv_mi355.so; [ $v = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$v pair (forked): "; ORV_LIB=$L timeout 300 python /root/repo/tools/time_attn_bwd.py 2>/dev/null
  rm -rf /tmp/pb; ORV_ATTN_BWD_FORK=0 ORV_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o x -- python /root/repo/tools/time_attn_bwd.py > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/pb/**/x_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn_bwd' in r['Name']: print('   %-30s %8.1f us' % (r['Name'].split('::')[-1][:30], float(r['AverageNs'])/1e3))
PY
done; done
} > /root/repo/gpurun_out/bwd_ab.txt 2>&1
cat /root/repo/gpurun_out/bwd_ab.txt
