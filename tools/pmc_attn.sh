#!/bin/bash
cd /tmp && export TMPDIR=/tmp
B=/root/repo/tools/bin/kbench_attn
OUT=/root/repo/gpurun_out/pmc_a
rm -rf $OUT; mkdir -p $OUT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1))
  FUSED=1 rocprofv3 --pmc $pmc --output-format csv -d $OUT -o p$i -- $B 4 > $OUT/log$i.txt 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_a/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if "attn" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
PY
tail -3 $OUT/log3.txt
