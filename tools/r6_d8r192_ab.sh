#!/bin/bash
# round 6: 192-row d8 tiles (gemm_d8r192_kernel) - parity first, then the one- / two- / four-clip steps with the tiles off / on (same box,
# interleaved), the B = 2 packed-operand choices, and the configs[4] training step with the transposed operands padded to 64 / 128
cd /root/repo; mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "d8 or packed" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -k "transpose or wgrad" 2>&1 | tail -2
B="python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 20 --warmup 5"
for r in 1 2; do for b in 1 2 4; do for f in 0 1; do echo -n "B=$b ORV_GEMM_D8R192=$f : "; ORV_GEMM_D8R192=$f $B --batch $b 2>/dev/null | python tools/bench_line_brief.py; done; done; done
for r in 1 2; do
  echo -n "B=2 r192 out=0 qkv=auto : "; ORV_PACKED_OUT=0 $B --batch 2 2>/dev/null | python tools/bench_line_brief.py
  echo -n "B=2 r192 out=1 qkv=1    : "; ORV_PACKED_OUT=1 ORV_PACKED_QKV=1 $B --batch 2 2>/dev/null | python tools/bench_line_brief.py
  echo -n "B=1 r192 qkv=0          : "; ORV_PACKED_QKV=0 $B --batch 1 2>/dev/null | python tools/bench_line_brief.py
done
for r in 1 2; do for f in 64 128; do echo -n "5B ckpt ORV_TRANSPOSE_PAD=$f : "; ORV_TRANSPOSE_PAD=$f python bench.py --mode train --model 5b --grad-ckpt --steps 4 --warmup 2 --batch 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['achieved_tflops_attn_ffn'])"; done; done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_training.py -x -q 2>&1 | tail -3
} > gpurun_out/r6_d8r192_ab.txt 2>&1
cat gpurun_out/r6_d8r192_ab.txt
