#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "192_row" 2>&1 | tail -40
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "full_width_single_layer" 2>&1 | tail -30
ORV_GEMM_D8R192=0 python bench.py --no-legs --no-vae --no-cpu-baseline --no-pmc --steps 5 --warmup 3 --batch 1 2>&1 | tail -15 | cut -c1-400
python - <<'PY'
import torch, sys
sys.path.insert(0, '/root/repo')
from orv_amd import ops
from orv_amd._lib import lib
dev = torch.device('cuda:0'); BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
def chk(bm, bn, M, K, epi, N=None):
    N = N or bn * 3
    A = torch.randn(M, K, device=dev, generator=g).to(BF); W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(BF)
    bias = torch.randn(N, device=dev, generator=g).to(BF); R = torch.randn(M, N, device=dev, generator=g).to(BF)
    ref = A.float() @ W.float().t() + bias.float()
    if epi == 1: ref = torch.nn.functional.gelu(ref, approximate='tanh')
    if epi == 2: ref = ref + R.float()
    Ap = ops.pack_rows16(A, M, K)
    C = torch.full((M, N), float('nan'), dtype=BF, device=dev)
    lib().orv_gemm_force_tile(5, bm, bn)
    try:
        ops.gemm(Ap, W, bias, C, M, N, K, epilogue=epi, a_packed=True, **(dict(R=R, ldr=N) if epi == 2 else {}))
    except Exception as e:
        print('ERR', bm, bn, M, K, epi, e); return
    finally:
        lib().orv_gemm_force_tile(0, 0, 0)
    torch.cuda.synchronize()
    err = (C.float() - ref).abs()
    bad = (~torch.isfinite(C.float())) | (err > 0.05 + 0.02 * ref.abs())
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print(f'bm={bm} bn={bn} M={M} K={K} epi={epi}: bad={int(bad.sum())} rows[{rows[:6].tolist()}..{rows[-3:].tolist()}] n_rows={rows.numel()} cols[{cols[:6].tolist()}..] n_cols={cols.numel()} maxerr={float(err[torch.isfinite(err)].max()):.3f}')
for bm in (256, 192):
    for bn in (128, 192):
        for (M, K) in ((100, 384), (3226, 1920), (3226, 384)):
            for epi in (0, 2):
                chk(bm, bn, M, K, epi)
PY
} > gpurun_out/r6_dbg.txt 2>&1
cat gpurun_out/r6_dbg.txt
