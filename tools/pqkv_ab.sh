#!/bin/bash
# packed LayerNorm output -> q | k | v on gemm_d8 (ORV_PACKED_QKV) off / on at one and two clips, same box, interleaved
cd /root/repo; mkdir -p gpurun_out
{
for B in 1 2; do for r in 1 2 3; do for f in 0 1; do echo -n "B=$B ORV_PACKED_QKV=$f : "; env ORV_PACKED_QKV=$f python bench.py --batch $B --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>gpurun_out/pqkv_err.txt | python tools/bench_line_brief.py; done; done; done
} > gpurun_out/pqkv_ab.txt 2>&1
cat gpurun_out/pqkv_ab.txt | cut -c1-260; tail -3 gpurun_out/pqkv_err.txt
