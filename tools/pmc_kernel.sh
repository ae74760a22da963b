#!/bin/bash
# PMC sweep of one standalone kernel benchmark (separate --pmc passes, no tracing flags mixed in).
# usage: bash tools/pmc_kernel.sh TAG KERNEL_SUBSTRING -- <command...>     -> gpurun_out/pmc_TAG/summary.txt
TAG=$1; PAT=$2; shift 3
CMD="$@"
OUT=/root/repo/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /root/repo/tools/bin && rocprofv3 --pmc $pmc --output-format csv -d $OUT -o p$i -- $CMD) > $OUT/log$i.txt 2>&1
done
python3 - "$OUT" "$PAT" <<'PY'
import csv,glob,collections,sys
out,pat=sys.argv[1],sys.argv[2]
tot={}; dur=[]
for f in sorted(glob.glob(out+'/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); tot["kernel"]=r["Kernel_Name"][:100]
            tot["VGPR"]=r["VGPR_Count"]; tot["AGPR"]=r["Accum_VGPR_Count"]; tot["LDS"]=r["LDS_Block_Size"]
            if r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in agg.items():
        v=sorted(v); tot[k]=v[len(v)//2]
with open(out+'/summary.txt','w') as fh:
    def emit(s): print(s); fh.write(s+"\n")
    for k,v in tot.items(): emit(f"{k} {v}")
    if dur and "GRBM_GUI_ACTIVE" in tot:
        d=sorted(dur)[len(dur)//2]; clk=tot["GRBM_GUI_ACTIVE"]/8/d
        emit(f"duration_ns(median, under PMC) {d}")
        emit(f"effective_clock_GHz (GRBM_GUI_ACTIVE/8 XCD/duration) {clk:.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
            emit(f"mfma_busy_frac (MFMA_BUSY / (1024 SIMD x GUI_ACTIVE/8)) {tot['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*tot['GRBM_GUI_ACTIVE']/8):.3f}")
        if "SQ_WAVE_CYCLES" in tot:
            for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VMEM"):
                if k in tot: emit(f"{k}/SQ_WAVE_CYCLES {tot[k]/tot['SQ_WAVE_CYCLES']:.3f}")
PY
find $OUT -name "*.csv" -size +200k -delete
