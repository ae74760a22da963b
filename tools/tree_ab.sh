#!/bin/bash
# headline step: round-3 tree (tools/bin/r3tree: git archive of 922b317 + its built library) vs the working tree, interleaved on one box
cd /root/repo; mkdir -p gpurun_out
fmt='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], " ".join("%s %.4f" % (k["kernel"].split("(")[0][-24:], k["avg_ms"]) for k in d["kernels"][:5]))'
{
for r in 1 2 3; do
  echo -n "r3   : "; (cd tools/bin/r3tree && python bench.py --no-legs --no-vae --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null) | python -c "$fmt"
  echo -n "tree : "; python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$fmt"
done
} > gpurun_out/tree_ab.txt 2>&1
cat gpurun_out/tree_ab.txt
