#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "attention or golden or full_depth" 2>&1 | tail -5 > gpurun_out/attn_pytest.txt; cat gpurun_out/attn_pytest.txt
for v in 1 0 1 0; do ORV_ATTN_PP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae > gpurun_out/bench_pp$v.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/bench_pp$v.json')); print('PP=$v', d['ms_per_step'], d['achieved_tflops_attn_ffn']); [print('   ',k['kernel'],k['avg_ms'],k['tflops']) for k in d['kernels'][:5]]
"; done
ORV_ATTN_PP=1 ORV_ATTN_STATIC=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae > gpurun_out/bench_pp_online.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/bench_pp_online.json')); print('PP online', d['ms_per_step']); [print('   ',k['kernel'],k['avg_ms'],k['tflops']) for k in d['kernels'][:3]]"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --batch 1 > gpurun_out/bench_b1.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/bench_b1.json')); print('B1', d['ms_per_step']); [print('   ',k['kernel'],k['avg_ms'],k['tflops']) for k in d['kernels'][:3]]"
