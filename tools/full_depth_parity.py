"""One-off: the whole CogVideoX-2B stack (30 blocks, S=3226, B=1, bench weights/inputs) through the HIP path and through the fp32
CPU oracle - does the bf16 error stay bounded with depth?  (~1 min of host time; not part of the pytest suite.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import dit

dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = {**bench.CFG_2B, "num_layers": L}
model = bench.build_model(cfg, dev)
model.action_embed.forced_mask = torch.zeros(1, dtype=torch.bool)
lat, img, prompt, actions = bench.synthetic_inputs(1, dev, torch.bfloat16)
x = torch.cat([lat, img], dim=2)
ts = torch.tensor([500], device=dev)
with torch.no_grad():
    out = model(x, prompt, {"actions": actions}, ts, return_dict=False)[0].float().cpu()
sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
t0 = time.time()
with torch.no_grad():
    ref = dit.dit_forward(sd, dict(model.config), x.float().cpu(), prompt.float().cpu(), ts.cpu(), actions=actions.float().cpu(),
                          is_mask=torch.zeros(1, dtype=torch.bool))[0]
print(f"layers={L} oracle {time.time() - t0:.1f} s; rel-L2(HIP, oracle) = {((out - ref).norm() / ref.norm()).item():.4e}; "
      f"max|ref| = {ref.abs().max().item():.3f}; max|err| = {(out - ref).abs().max().item():.4f}")
