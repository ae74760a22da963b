#!/bin/bash
# B = 1 and B = 2 steps with the packed path off / on (same box, interleaved)
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2; do for b in 1 2; do for f in 0 1; do echo -n "B=$b ORV_GEMM_PACKED=$f : "; ORV_GEMM_PACKED=$f python bench.py --batch $b --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python tools/bench_line_brief.py; done; done; done
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "full_depth or golden" 2>&1 | tail -3
} > gpurun_out/b1_ab.txt 2>&1
cat gpurun_out/b1_ab.txt
