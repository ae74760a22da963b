#!/bin/bash
# Per-kernel PMC profile of the SHIPPED kernels inside bench.py (separate rocprofv3 --pmc passes, no tracing flags mixed in).
# usage: bash tools/pmc_bench.sh TAG [bench args...]    -> gpurun_out/pmc_bench_TAG/{sq1,sq2,fetch,write}*.csv + summary.txt + hbm_traffic.json
TAG=${1:-r2}; shift
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_bench_$TAG
rm -rf $OUT; mkdir -p $OUT
# PMC_CMD overrides the profiled command (default: two steps of bench.py with the given arguments)
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT -o $name -- ${PMC_CMD:-python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS} > $OUT/log_$name.txt 2>&1 < /dev/null; }
BENCH_ARGS="$@"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 /root/repo/tools/pmc_bench.py $OUT | tee $OUT/summary.txt
# the raw per-dispatch CSVs are hundreds of MB (gpurun copies back at most 64 MiB): keep the aggregate only
[ -n "$KEEP_CSV" ] || find $OUT -name "*.csv" -delete
