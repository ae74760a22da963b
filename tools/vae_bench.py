"""Full-size VAE decode / encode timing (real 2B geometry, random weights): python tools/vae_bench.py [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orv_amd.vae import AutoencoderKLCogVideoX
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
with torch.device(dev):
    vae = AutoencoderKLCogVideoX()
for p in vae.parameters():
    if p.ndim > 1:
        p.data.normal_(0, 1.0 / (p[0].numel() ** 0.5))
vae = vae.to(torch.bfloat16).eval()
z = torch.randn(B, 16, 5, 40, 60, device=dev, dtype=torch.bfloat16)
x = torch.rand(B, 3, 1, 320, 480, device=dev, dtype=torch.bfloat16) * 2 - 1
vae.decode(z); vae.encode(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    vae.decode(z)
torch.cuda.synchronize()
td = (time.perf_counter() - t0) / iters
t0 = time.perf_counter()
for _ in range(iters):
    vae.encode(x)
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / iters
print(f"VAE decode B={B}: {td * 1e3:.1f} ms ({td * 1e3 / B:.1f} ms/clip); encode of one reference frame per clip: {te * 1e3:.1f} ms; "
      f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
