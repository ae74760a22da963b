#!/bin/bash
# where the graph-replayed step spends time that is not kernel time: rocprofv3 kernel trace of a short headline run, gaps between
# consecutive kernels of the timed region -> gpurun_out/gap_trace.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_gap; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o gap -- python /root/repo/bench.py --no-cpu-baseline --no-legs --no-vae --steps 12 --warmup 3 > $OUT/bench_stdout.log 2>&1
python3 - $OUT/gap_kernel_trace.csv > /root/repo/gpurun_out/gap_trace.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# steps: one sched_step_kernel per denoise step
idx = [i for i, e in enumerate(ev) if "sched_step_kernel" in e[2]]
print("kernels", len(ev), "steps", len(idx))
# graph-replayed steps = warmup 3 .. 3+12 (then the eager timeline loop)
for lo, hi, tag in ((idx[4], idx[13], "graph-replayed steps 5..13"), (idx[-8], idx[-2], "eager timeline steps")):
    seg = ev[lo:hi + 1]
    nsteps = sum(1 for e in seg if "sched_step_kernel" in e[2]) - 1
    span = seg[-1][0] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg[:-1])
    gaps = collections.Counter(); gapn = collections.Counter(); total_gap = 0; ov = 0
    for a, b in zip(seg[:-1], seg[1:]):
        g = b[0] - a[1]
        if g > 0:
            total_gap += g
            key = a[2].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:] + " -> " + b[2].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
            gaps[key] += g; gapn[key] += 1
        else:
            ov += -g
    print(f"== {tag}: {nsteps} steps, span/step {span/nsteps/1e6:.3f} ms, kernel time/step {busy/nsteps/1e6:.3f} ms, gaps/step {total_gap/nsteps/1e6:.3f} ms, overlap/step {ov/nsteps/1e6:.3f} ms")
    for k, v in gaps.most_common(14):
        print(f"   {v/nsteps/1e3:8.1f} us/step in {gapn[k]/nsteps:6.1f} gaps  {k}")
PY
cat /root/repo/gpurun_out/gap_trace.txt; rm -rf $OUT
