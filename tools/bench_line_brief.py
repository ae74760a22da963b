"""stdin: bench.py output; prints ms_per_step and the per-kernel averages of the JSON line (tools/model_ab.sh)."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["ms_per_step"], " | ".join("%s %.4f" % (k["kernel"].replace("gemm_", "").replace("_kernel", "").replace(" ", "")[:44], k["avg_ms"]) for k in d.get("kernels", [])[:7]))
