#!/bin/bash
# round 6, first GPU call: the new full-size parity tests first (fail fast), then the whole GPU suite, then the default bench line
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "full_depth or trained_like or 480x640" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r6_parity_new.txt; cat gpurun_out/r6_parity_new.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/pytest_gpu.txt; cat gpurun_out/pytest_gpu.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python -c "
import json
d=json.load(open('gpurun_out/bench_default.json')); print(d['ms_per_step'], d['achieved_tflops_attn_ffn'], d['frac_mfma_peak_attn_ffn'], 'eager', d['eager_ms_per_step']); [print('   ',k['kernel'],k['avg_ms'],k['tflops']) for k in d['kernels']]
for k in ('roofline','attention_softmax','vae_decode','b1','attn_online','attn_mixed','cond','train','train_5b_ckpt','cpu_baseline','lib'): print(k, json.dumps(d.get(k))[:600])
"
