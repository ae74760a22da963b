#!/bin/bash
# tile-walk knobs on the round-5 tree (d8 for FFN2 / out-projection): ORV_GEMM_GM x ORV_GEMM_WALK_BACK, headline step, same box, two interleaved rounds
cd /root/repo; mkdir -p gpurun_out
{
for r in 1 2; do for gm in d 2 4 8 16; do for wb in d 0 1; do
  E=""; [ $gm != d ] && E="$E ORV_GEMM_GM=$gm"; [ $wb != d ] && E="$E ORV_GEMM_WALK_BACK=$wb"
  echo -n "GM=$gm WALK_BACK=$wb : "; env $E python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], ' '.join('%.4f' % k['avg_ms'] for k in d['kernels'][:5]))"
done; done; done
} > gpurun_out/knob_sweep_r5.txt 2>&1
cat gpurun_out/knob_sweep_r5.txt
