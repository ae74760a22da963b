#!/bin/bash
# full GPU test suite + the default bench line (-> gpurun_out/)
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/pytest_gpu.txt; cat gpurun_out/pytest_gpu.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python -c "
import json
d=json.load(open('gpurun_out/bench_default.json')); print(d['ms_per_step'], d['achieved_tflops_attn_ffn'], d['frac_mfma_peak_attn_ffn'], 'eager', d['eager_ms_per_step']); [print('   ',k['kernel'],k['avg_ms'],k['tflops']) for k in d['kernels']]
for k in ('roofline','vae_decode','b1','train','cpu_baseline','lib'): print(k, json.dumps(d.get(k))[:600])
"
