"""Times the HBM-bound backward kernels at the 2B training shape (B=4, S=3226, D=1920) with their outputs toggled."""
import sys, torch
sys.path.insert(0, "/root/repo")
from orv_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, S, Nt, D, G = 4, 3226, 226, 1920, 6
M = B * S
g = torch.Generator(device=dev).manual_seed(0)
dy = torch.randn(M, D, device=dev, generator=g).to(BF); x = torch.randn(M, D, device=dev, generator=g).to(BF)
dres = torch.randn(M, D, device=dev, generator=g).to(BF); dx = torch.empty_like(x)
gam = torch.randn(D, device=dev).to(BF); bet = torch.randn(D, device=dev).to(BF)
mod = torch.randn(B, G, 3 * D, device=dev); dmod = torch.zeros_like(mod)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
grp = ops.groups(S, Nt, 600)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
full = lambda: ops.layernorm_modulate_bwd(dy, x, dres, dx, gam, bet, mod[..., D:2*D], dmod[..., D:2*D], dmod[..., :D], dg, db, G*3*D, 3*D, grp, B, D, 1e-5)
noscale = lambda: ops.layernorm_modulate_bwd(dy, x, dres, dx, gam, bet, None, None, None, dg, db, 0, 0, grp, B, D, 1e-5)
nogb = lambda: ops.layernorm_modulate_bwd(dy, x, dres, dx, gam, bet, mod[..., D:2*D], dmod[..., D:2*D], dmod[..., :D], None, None, G*3*D, 3*D, grp, B, D, 1e-5)
bare = lambda: ops.layernorm_modulate_bwd(dy, x, None, dx, gam, bet, None, None, None, None, None, 0, 0, grp, B, D, 1e-5)
print("ln_mod_bwd full %.1f us | no scale tables %.1f | no dgamma/dbeta %.1f | dx only %.1f" % (t(full), t(noscale), t(nogb), t(bare)))
dyo = torch.empty_like(x)
print("gated_bwd %.1f us" % t(lambda: ops.gated_residual_bwd(dy, x, mod[..., 2*D:], dmod[..., 2*D:], dyo, G*3*D, 3*D, grp, B, D)))
out = torch.zeros(D, device=dev)
print("colsum D %.1f us" % t(lambda: ops.colsum(dy, out, M, D)))
xn = torch.empty_like(x)
print("ln_mod fwd %.1f us" % t(lambda: ops.layernorm_modulate(x, xn, gam, bet, mod[..., D:2*D], mod[..., :D], G*3*D, 3*D, grp, B, D, 1e-5)))
