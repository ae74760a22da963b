#!/bin/bash
# PMC sweep of one GEMM shape (separate --pmc passes, no tracing flags mixed in).
cd /tmp && export TMPDIR=/tmp
B=/root/repo/tools/bin/kbench_gemm
OUT=/root/repo/gpurun_out/pmc_g
rm -rf $OUT; mkdir -p $OUT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $OUT -o p$i -- $B bench ${1:-12904} ${2:-7680} ${3:-1920} ${4:-1} 3 > $OUT/log$i.txt 2>&1
done
ls $OUT
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_g/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if "gemm" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
PY
