"""Per-wave segment times of the two attention-backward passes (build first:
cd orv_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DORV_SEG_TRACE -c attention_bwd.hip -o /tmp/bwd_seg.o && link into
tools/bin/bv_seg/liborv_mi355.so as tools/attn_variants.sh does; run with ORV_LIB=tools/bin/bv_seg/liborv_mi355.so ORV_ATTN_BWD_FORK=0)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
from orv_amd import ops
from orv_amd._lib import lib
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, S, H = 4, 3226, 30
D = H * 64; s_pad = (S + 63) // 64 * 64
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * S, 3 * D, device=dev, generator=g) * 0.5).to(BF)
out = torch.randn(B * S, D, device=dev, generator=g).to(BF); dout = torch.randn(B * S, D, device=dev, generator=g).to(BF)
lse = torch.randn(B, H, S, device=dev, generator=g) + 8
nl = torch.empty(B, H, s_pad, dtype=torch.float32, device=dev); nd = torch.empty_like(nl)
dqkv = torch.empty_like(qkv)
nwg = ((S + 255) // 256) * H * B
trace = torch.zeros(2 * nwg * 8 * 4, dtype=torch.int64, device=dev)
fn = lib().orv_debug_attn_bwd_trace
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(trace.data_ptr()) == 0
f = lambda: ops.attention_bwd(qkv, None, None, out, dout, None, lse, nl, nd, dqkv, B, S, H, s_pad, 1.0 / 1.4426950408889634)
for _ in range(3): f()
torch.cuda.synchronize()
a = trace.cpu().numpy().astype(np.float64).reshape(2, nwg, 8, 4) * 10.0 / 51.0      # ns per tile
for k, name in ((0, "dQ pass"), (1, "dK/dV pass")):
    x = a[k]
    x = x[x[:, :, 0].min(axis=1) > 0]
    for half, sl in (("first half", slice(0, 4)), ("second half", slice(4, 8))):
        v = x[:, sl, :].mean(axis=(0, 1))
        print(f"{name:11s} {half:11s}: per tile  matrix {v[0]:7.1f} ns | barrier {v[1]:6.1f} | vector {v[2]:7.1f} | barrier {v[3]:6.1f} | sum {v.sum():7.1f}")
