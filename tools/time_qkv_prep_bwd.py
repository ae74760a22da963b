"""Times orv_qkv_prep_bwd at the 2B training shape (B=4, S=3226, H=30) and prints checksums.  ORV_LIB=<path to liborv_mi355.so> picks another build."""
import sys, torch
sys.path.insert(0, "/root/repo")
from orv_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, S, H, nt = 4, 3226, 30, 226
D = H * 64
g = torch.Generator(device=dev).manual_seed(0)
raw = torch.randn(B * S, 3 * D, device=dev, generator=g).to(BF)
dqkv0 = torch.randn(B * S, 3 * D, device=dev, generator=g).to(BF)
gq, gk = (torch.randn(64, device=dev, generator=g).to(BF) for _ in range(2))
outs = [torch.zeros(64, device=dev) for _ in range(4)]
dqkv = dqkv0.clone()
f = lambda: ops.qkv_prep_bwd(raw, dqkv, gq, gk, None, *outs, B, S, H, nt, 1e-6)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
dqkv.copy_(dqkv0); [o.zero_() for o in outs]; f(); torch.cuda.synchronize()
print("qkv_prep_bwd: %.1f us   checksum dq|dk %.6e  dgq %.6e dbk %.6e" % (e0.elapsed_time(e1) / 20 * 1e3, dqkv[:, :2 * D].float().abs().sum().item(), outs[0].sum().item(), outs[3].sum().item()))
