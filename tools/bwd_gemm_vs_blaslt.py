"""Training-side plain GEMMs (weight and data gradients of the four linears of a block) - the library path (transposes + NT kernel,
training._wgrad / _dgrad) against hipBLASLt through torch.mm on the un-transposed operands, same process, interleaved: python tools/bwd_gemm_vs_blaslt.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orv_amd import training

dev = torch.device("cuda:0")
torch.manual_seed(0)
M = 12904


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {"orv": 0.0, "blaslt": 0.0}
for name, N, K in (("qkv", 5760, 1920), ("out", 1920, 1920), ("ffn1", 7680, 1920), ("ffn2", 1920, 7680)):
    dY = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    X = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = (torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02)
    dW = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    dX = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    bs = torch.zeros(N, device=dev, dtype=torch.float32)
    r = {}
    for rnd in range(2):
        r.setdefault("wgrad orv", []).append(timed(lambda: training._wgrad(dY, X, dW, M, N, K, accumulate=False, bias_sum=bs)))
        r.setdefault("wgrad blaslt", []).append(timed(lambda: (torch.mm(dY.t(), X, out=dW), dY.float().sum(0) if False else None)))
        r.setdefault("dgrad orv", []).append(timed(lambda: training._dgrad(dY, W, dX, M, N, K)))
        r.setdefault("dgrad blaslt", []).append(timed(lambda: torch.mm(dY, W, out=dX)))
    ref = torch.mm(dY.t().float(), X.float())
    training._wgrad(dY, X, dW, M, N, K, accumulate=False)
    e1 = ((dW.float() - ref).norm() / ref.norm()).item()
    e2 = ((torch.mm(dY.t(), X).float() - ref).norm() / ref.norm()).item()
    print(f"{name:5s} N={N} K={K}: " + "  ".join(f"{k} {min(v):.0f}" for k, v in r.items()) + f" us   rel-L2 vs fp32: orv {e1:.1e} blaslt {e2:.1e}", flush=True)
    tot["orv"] += min(r["wgrad orv"]) + min(r["dgrad orv"]); tot["blaslt"] += min(r["wgrad blaslt"]) + min(r["dgrad blaslt"])
print("per layer (us):", tot)
