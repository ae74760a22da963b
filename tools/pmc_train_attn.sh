#!/bin/bash
# SQ counters of the attention forward / backward kernels inside one SFT step (separate --pmc passes, no tracing flags)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_ta
rm -rf $OUT; mkdir -p $OUT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $OUT -o p$i -- python /root/repo/bench.py --mode train --steps 1 --warmup 1 --layers 4 --no-cpu-baseline > $OUT/log$i.txt 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_ta/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n=r["Kernel_Name"]
        for key in ("attn_bwd_dkv","attn_bwd_dq","attn_fwd_v1"):
            if key in n: agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key,d in agg.items():
        print(key, {k: round(sum(v)/len(v)) for k,v in d.items()})
PY
