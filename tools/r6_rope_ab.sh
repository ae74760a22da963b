#!/bin/bash
# round 6: RoPE fused into the qk-LayerNorm projection epilogue (ORV_FUSED_ROPE=1, default) against projection + orv_qkv_prep (=0): parity first,
# then the configs[4] training step (CogVideoX1.5-5B, checkpointed and not), same box, interleaved
cd /root/repo; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "rope or qk_layernorm or d8" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_training.py -x -q -s 2>&1 | grep -E "passed|failed|5B|configs\[4\]" | tail -8
for r in 1 2 3; do for f in 0 1; do
  echo -n "5B ckpt ORV_FUSED_ROPE=$f : "; ORV_FUSED_ROPE=$f python bench.py --mode train --model 5b --grad-ckpt --steps 4 --warmup 2 --batch 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['achieved_tflops_attn_ffn'], d['final_loss'])"
done; done
for r in 1 2; do for f in 0 1; do
  echo -n "5B      ORV_FUSED_ROPE=$f : "; ORV_FUSED_ROPE=$f python bench.py --mode train --model 5b --steps 4 --warmup 2 --batch 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['achieved_tflops_attn_ffn'], d['final_loss'])"
done; done
} > gpurun_out/r6_rope_ab.txt 2>&1
cat gpurun_out/r6_rope_ab.txt
