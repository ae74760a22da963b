#!/bin/bash
# in-model sweep of the GEMM tile-walk knobs after the round-4 DMA change (same box, two rounds): ORV_GEMM_GM, ORV_GEMM_WALK_BACK
cd /root/repo; mkdir -p gpurun_out
fmt='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], " ".join("%.4f" % k["avg_ms"] for k in d["kernels"][:5]))'
run() { timeout 200 python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"; }
{
for r in 1 2; do
  echo -n "default       : "; run
  for g in 2 3 4 6 8 13; do echo -n "GM=$g          : "; ORV_GEMM_GM=$g run; done
  for w in 0 1; do echo -n "WALK_BACK=$w   : "; ORV_GEMM_WALK_BACK=$w run; done
done
} > gpurun_out/knob_sweep.txt 2>&1
cat gpurun_out/knob_sweep.txt
