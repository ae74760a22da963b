#!/bin/bash
# attention forward A/B: the tree's build against tools/bin/dv_prevfwd (the library before the change under test): attention parity tests,
# standalone interleaved timing at B = 4 / B = 1 (shift-free and online kernels), then the B = 4 step with either library
cd /root/repo; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -3
cd tools/bin
for r in 1 2 3; do for v in prevfwd base; do
  L=/root/repo/tools/bin/dv_$v; [ $v = base ] && L=/root/repo/orv_amd
  echo -n "$v B=4: "; LD_LIBRARY_PATH=$L FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 4
  echo -n "$v B=1: "; LD_LIBRARY_PATH=$L FUSED=1 BOUND=12 ITERS=40 ./kbench_attn 1
  echo -n "$v B=4 online: "; LD_LIBRARY_PATH=$L FUSED=1 ITERS=40 ./kbench_attn 4
done; done
cd /root/repo
for r in 1 2; do for v in prevfwd base; do
  L=/root/repo/tools/bin/dv_$v/liborv_mi355.so; [ $v = base ] && L=/root/repo/orv_amd/liborv_mi355.so
  echo -n "$v step: "; ORV_LIB=$L python bench.py --no-vae --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python tools/bench_line_brief.py | cut -c1-130
done; done
} > gpurun_out/attn_fwd_ab.txt 2>&1
cat gpurun_out/attn_fwd_ab.txt
