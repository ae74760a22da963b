#!/bin/bash
# LayerNorm-modulate rows-per-wave setting in the one-clip step (ORV_LN_ROWS: unset = automatic, 0 = one-row kernel, n = fixed), same box, interleaved
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2; do for f in auto 0 1 3 4; do echo -n "ORV_LN_ROWS=$f : "; if [ $f = auto ]; then E=""; else E="ORV_LN_ROWS=$f"; fi; env $E python bench.py --batch 1 --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done
} > gpurun_out/ln_rows_b1.txt 2>&1
cat gpurun_out/ln_rows_b1.txt
