#!/bin/bash
# headline step with one environment switch off / on (same box, interleaved): model_ab.sh ENVVAR VAL0 VAL1
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2 3; do for f in $2 $3; do echo -n "$1=$f : "; env $1=$f python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python tools/bench_line_brief.py; done; done
} > gpurun_out/model_ab.txt 2>&1
cat gpurun_out/model_ab.txt
