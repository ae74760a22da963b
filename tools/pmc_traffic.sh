#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one TCC pass), no tracing flags mixed in.
# usage: bash tools/pmc_traffic.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/{fetch,write}_counter_collection.csv + hbm_traffic.json
TAG=${1:-r1}; shift
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
for pass in fetch:FETCH_SIZE write:WRITE_SIZE; do
  name=${pass%%:*}; ctr=${pass##*:}
  rocprofv3 --pmc $ctr --output-format csv -d $OUT -o $name -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/log_$name.txt 2>&1
done
python3 /root/repo/tools/pmc_traffic.py $OUT "$@" | tee $OUT/summary.txt
