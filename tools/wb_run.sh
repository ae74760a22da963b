cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
bash tools/model_ab.sh ORV_GEMM_WALK_BACK 0 1
for f in 0 1; do echo -n "B=1 WALK_BACK=$f : "; ORV_GEMM_WALK_BACK=$f python bench.py --no-legs --batch 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
