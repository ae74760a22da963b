#!/bin/bash
# full GPU test suite + headline bench lines with A/B switches (-> gpurun_out/)
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/pytest_gpu.txt; cat gpurun_out/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_b4.json 2> gpurun_out/bench_b4.err
ORV_GEMM_T8=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae > gpurun_out/bench_b4_not8.json 2>/dev/null
ORV_ATTN_STATIC=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae > gpurun_out/bench_b4_nostatic.json 2>/dev/null
ORV_GEMM_QKV_SPLIT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae > gpurun_out/bench_b4_nosplit.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --batch 1 > gpurun_out/bench_b1.json 2>/dev/null
python -c "
import json
for f in ('bench_b4','bench_b4_not8','bench_b4_nostatic','bench_b4_nosplit','bench_b1'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['ms_per_step'], d['achieved_tflops_attn_ffn']); [print('   ',k['kernel'],k['avg_ms'],k['tflops']) for k in d['kernels']]
    except Exception as e: print(f, 'FAILED', e)
"
