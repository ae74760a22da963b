import torch, math, sys
sys.path.insert(0, "/root/repo")
from orv_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, S, H = 4, 3226, 30
g = torch.Generator().manual_seed(11)
dq = (torch.randn(B * S, 3 * H * 64, generator=g) * 0.9).to(dev, BF)
nb = ops.attention_ws_bytes(B, S, H); print("ws bytes", nb)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def run(w):
    out = torch.empty(B * S, H * 64, dtype=BF, device=dev); lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention_fwd(dq, None, out, B, S, H, 0, 1.0 / 1.4426950408889634, lse=lse, score_bound=30.0, ws=w)
    return out.float().view(B, S, H, 64), lse
o0, l0 = run(None); o1, l1 = run(ws)
for b, h in ((3, 29), (3, 28), (3, 27), (0, 0)):
    d = (o1[b, :, h] - o0[b, :, h]); print(b, h, "rel_l2 split vs unsplit", (d.norm() / o0[b, :, h].norm()).item(), "max abs", d.abs().max().item(), "lse max diff", (l1[b, h] - l0[b, h]).abs().max().item())
    # per q-tile
    for qt in range(13):
        sl = slice(qt * 256, min(S, qt * 256 + 256)); dd = d[sl]
        print("   qtile", qt, (dd.norm() / o0[b, sl, h].norm()).item())
import time
for w, name in ((None, "unsplit"), (ws, "split")):
    for _ in range(3): run(w)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run(w)
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 20 * 1e3, "ms")
