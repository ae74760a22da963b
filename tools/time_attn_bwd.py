"""Times orv_attention_bwd (dq + dkv passes) at the 2B training shape.  ORV_LIB=<path to liborv_mi355.so> picks another build."""
import sys, torch
sys.path.insert(0, "/root/repo")
from orv_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, S, H = 4, 3226, 30
D = H * 64; s_pad = (S + 63) // 64 * 64
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * S, 3 * D, device=dev, generator=g) * 0.5).to(BF)
out = torch.randn(B * S, D, device=dev, generator=g).to(BF); dout = torch.randn(B * S, D, device=dev, generator=g).to(BF)
lse = torch.randn(B, H, S, device=dev, generator=g) + 8
qT = torch.zeros(B, H, 64, s_pad, dtype=BF, device=dev); kT = torch.zeros_like(qT); doT = torch.zeros_like(qT)
ops.head_transpose(qkv, 0, qT, B, S, H, s_pad, ld=3 * D); ops.head_transpose(qkv, D, kT, B, S, H, s_pad, ld=3 * D)
ops.head_transpose(dout, 0, doT, B, S, H, s_pad, ld=D)
nl = torch.empty(B, H, s_pad, dtype=torch.float32, device=dev); nd = torch.empty_like(nl)
dqkv = torch.empty_like(qkv)
import os
if os.environ.get("ORV_ATTN_BWD_PP", "1") != "0":
    qT = kT = doT = None          # ping-pong kernels: no transposed copies
f = lambda: ops.attention_bwd(qkv, qT, kT, out, dout, doT, lse, nl, nd, dqkv, B, S, H, s_pad, 1.0 / 1.4426950408889634)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("attention_bwd (prep + dq + dkv): %.1f us" % (e0.elapsed_time(e1) / 10 * 1e3))
