"""Forced-tile timing of the N = 1920 projections at one and two clips (which tile should the planner take?): python tools/r192_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orv_amd import ops
from orv_amd._lib import lib

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (3226, 6452):
    for N, K, epi in ((1920, 1920, 0), (1920, 1920, 2), (1920, 7680, 2), (3840, 1920, 0), (7680, 1920, 1)):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        r = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(R=r, ldr=N) if epi == 2 else {}
        row = []
        for tile in ((0, 0, 0), (3, 192, 192), (3, 256, 192), (3, 192, 256), (3, 256, 256), (0, 192, 128), (0, 128, 192), (1, 256, 128)):
            if N % tile[2 if tile[1] else 0 or 2] if tile[1] else False:
                continue
            lib().orv_gemm_force_tile(*tile)
            try:
                ts = [timed(lambda: ops.gemm(a, w, b, out, M, N, K, epilogue=epi, **kw)) for _ in range(2)]
                row.append("%s %.1f" % ("default" if not tile[1] else "%d:%dx%d" % tile, min(ts)))
            except Exception as e:
                row.append("%d:%dx%d n/a" % tile)
            finally:
                lib().orv_gemm_force_tile(0, 0, 0)
        print(f"M={M} N={N} K={K} epi={epi}: " + " | ".join(row) + "  us", flush=True)
