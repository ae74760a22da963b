#!/bin/bash
# 192-row t8 tiles (gemm_t8r192_kernel) off / on in the single-clip and two-clip steps, same box, interleaved: r192_ab.sh
cd /root/repo; mkdir -p gpurun_out
{
for B in 1 2 4; do for r in 1 2 3; do for f in 0 1; do echo -n "B=$B ORV_GEMM_R192=$f : "; env ORV_GEMM_R192=$f python bench.py --batch $B --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>gpurun_out/r192_err.txt | python tools/bench_line_brief.py; done; done; done
} > gpurun_out/r192_ab.txt 2>&1
cat gpurun_out/r192_ab.txt; tail -3 gpurun_out/r192_err.txt
