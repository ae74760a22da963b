"""Control measurement: the library GEMMs against hipBLASLt (torch.nn.functional.linear) on the block's shapes, random bf16 operands, same
process, interleaved.  Plain GEMM + bias only (hipBLASLt has no gated-residual / qk-LayerNorm / packed-layout epilogues); the point is the
K-loop rate a tuned vendor kernel reaches on this box under the same power limit.  usage: python tools/hipblaslt_compare.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from orv_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [("FFN1", 12904, 7680, 1920), ("FFN2", 12904, 1920, 7680), ("q|k|v", 12904, 5760, 1920), ("out", 12904, 1920, 1920),
          ("8192^3", 8192, 8192, 8192), ("FFN1 B=1", 3226, 7680, 1920), ("FFN2 B=1", 3226, 1920, 7680)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        res.setdefault("orv", []).append(timed(lambda: ops.gemm(a, w, b, out, M, N, K)))
        res.setdefault("blaslt", []).append(timed(lambda: F.linear(a, w, b)))
    ref = F.linear(a, w, b).float()
    err = ((out.float() - ref).norm() / ref.norm()).item()
    fl = 2.0 * M * N * K
    fmt = lambda ts: " / ".join("%.0f" % (fl / (t * 1e-3) / 1e12) for t in ts)
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}  orv {fmt(res['orv'])} TF   hipBLASLt {fmt(res['blaslt'])} TF   rel-L2 orv vs hipBLASLt {err:.1e}", flush=True)
