#!/bin/bash
# rows-per-wave LayerNorm (ln_mod_rows_kernel) vs the one-row kernel, in the model on one box: ORV_LN_ROWS=0 (old) / unset (auto R) / fixed R
cd /root/repo; mkdir -p gpurun_out
fmt='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], " ".join("%s %.4f" % (k["kernel"].split("(")[0][-28:], k["avg_ms"]) for k in d["kernels"] if "ln_mod" in k["kernel"] or "N=7680" in k["kernel"] or "N=5760" in k["kernel"]))'
{
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "layernorm or ln_mod or norm" 2>&1 | tail -2
for r in 1 2 3; do for n in 0 auto 4 6 8; do
  echo -n "ORV_LN_ROWS=$n : "
  if [ $n = auto ]; then python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"
  else ORV_LN_ROWS=$n python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"; fi
done; done
} > gpurun_out/ln_rows_ab.txt 2>&1
cat gpurun_out/ln_rows_ab.txt
