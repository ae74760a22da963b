"""Forced-tile timing of the out-projection weight gradient's GEMM (dW[1920, 1920] = dY^T X over 12928 contraction rows): python tools/wgrad_out_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orv_amd import ops
from orv_amd._lib import lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((1920, 1920, 12928), (5760, 1920, 12928), (1920, 7680, 12928)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = []
    for tile in ((0, 0, 0), (0, 128, 128), (0, 128, 64), (0, 192, 128), (0, 128, 192), (0, 256, 64), (0, 256, 128), (1, 256, 128), (1, 256, 192), (2, 256, 128), (3, 256, 192), (3, 192, 192), (3, 256, 256), (3, 192, 256)):
        lib().orv_gemm_force_tile(*tile)
        try:
            t = min(timed(lambda: ops.gemm(a, w, None, out, M, N, K)) for _ in range(2))
            row.append("%s %.0f" % ("default" if not tile[1] else "%d:%dx%d" % tile, t))
        except Exception:
            row.append("%d:%dx%d n/a" % tile)
        finally:
            lib().orv_gemm_force_tile(0, 0, 0)
    print(f"M={M} N={N} K={K}: " + " | ".join(row) + " us", flush=True)
