"""Micro-benchmark of orv_layernorm_modulate_bwd and orv_gated_residual_bwd at the 2B training shape (B 4, S 3226, D 1920): ORV_LIB picks the build."""
import sys, torch
sys.path.insert(0, "/root/repo")
from orv_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, S, nt, P, D = 4, 3226, 226, 600, 1920
G = 1 + (S - nt) // P
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
x, dy, dres = rnd(B * S, D).to(BF), rnd(B * S, D).to(BF), rnd(B * S, D).to(BF)
dx = torch.empty_like(x)
gamma, beta = rnd(D).to(BF), rnd(D).to(BF)
mod = rnd(B, G, 3 * D)
dscale, dshift = torch.zeros(B, G, D, device=dev), torch.zeros(B, G, D, device=dev)
dgamma, dbeta = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
grp = ops.groups(S, nt, P)
def ln():
    ops.layernorm_modulate_bwd(dy, x, dres, dx, gamma, beta, mod[..., D:2 * D], dscale, dshift, dgamma, dbeta, G * 3 * D, 3 * D, grp, B, D, 1e-5)
y = rnd(B * S, D).to(BF); dgate = torch.zeros(B, G, D, device=dev); dyo = torch.empty_like(y)
def gt():
    ops.gated_residual_bwd(dy, y, mod[..., 2 * D:], dgate, dyo, G * 3 * D, 3 * D, grp, B, D)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("layernorm_modulate_bwd %.1f us (checksum %.6e) | gated_residual_bwd %.1f us (checksum %.6e)" % (timeit(ln), float(dx.float().sum()), timeit(gt), float(dyo.float().sum())))
