#!/bin/bash
# PMC sweep of one GEMM shape (separate --pmc passes, no tracing flags mixed in).
# usage: [ORV_GEMM_TILE=ring,bm,bn] bash tools/pmc_gemm.sh TAG M N K EPI   -> gpurun_out/pmc_gemm_TAG/
TAG=${1:-g}; shift
cd /tmp && export TMPDIR=/tmp
B=/root/repo/tools/bin/kbench_gemm
OUT=/root/repo/gpurun_out/pmc_gemm_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $OUT -o p$i -- $B bench ${1:-12904} ${2:-7680} ${3:-1920} ${4:-1} 3 > $OUT/log$i.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
tot={}
for f in sorted(glob.glob(out+'/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    name=None
    for r in rows:
        if "gemm" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"])); name=r["Kernel_Name"]
    for k,v in agg.items(): tot[k]=sum(v)/len(v)
    if name: tot["kernel"]=name[:80]
with open(out+'/summary.txt','w') as fh:
    for k,v in tot.items():
        line=f"{k} {v}"; print(line); fh.write(line+"\n")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "GRBM_GUI_ACTIVE" in tot:
        line=f"mfma_busy_frac(per SIMD, 1024 SIMDs) {tot['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*tot['GRBM_GUI_ACTIVE']):.3f}"; print(line); fh.write(line+"\n")
PY
grep bench $OUT/log1.txt
