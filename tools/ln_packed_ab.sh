#!/bin/bash
# LayerNorm output in the packed layout (q | k | v and / or FFN1 on the d8 GEMM): step time at B = 4 and B = 1, same box, interleaved; parity first
cd /root/repo; mkdir -p gpurun_out
{
ORV_PACKED_QKV=1 ORV_PACKED_FFN1=1 timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "full_depth or full_width" 2>&1 | tail -3
for r in 1 2; do for b in 4 1; do for c in "0 0" "1 0" "0 1" "1 1"; do set -- $c
  echo -n "B=$b QKV=$1 FFN1=$2 : "; ORV_PACKED_QKV=$1 ORV_PACKED_FFN1=$2 python bench.py --batch $b --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python tools/bench_line_brief.py
done; done; done
} > gpurun_out/ln_packed_ab.txt 2>&1
cat gpurun_out/ln_packed_ab.txt
