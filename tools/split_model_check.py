"""B = 4 forward (30 blocks, bench weights) vs four B = 1 forwards, clip by clip; run under ORV_ATTN_SPLIT=0 / 1 (read once per process).
Saves the B = 4 output so that the two settings can be compared with each other."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0"); BF = torch.bfloat16
rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
cfg = dict(bench.CFG_2B); model = bench.build_model(cfg, dev)
lat, img, prompt, actions = bench.synthetic_inputs(4, dev, BF)
x = torch.cat([lat, img], dim=2); ts = torch.tensor([500, 999, 19, 259], device=dev)
with torch.no_grad():
    model.action_embed.forced_mask = torch.zeros(4, dtype=torch.bool)
    out4 = model(x, prompt, {"actions": actions}, ts, return_dict=False)[0].float().cpu()
    out4b = model(x, prompt, {"actions": actions}, ts, return_dict=False)[0].float().cpu()
    model.action_embed.forced_mask = torch.zeros(1, dtype=torch.bool)
    singles = [model(x[b:b + 1], prompt[b:b + 1], {"actions": actions[b:b + 1]}, ts[b:b + 1], return_dict=False)[0].float().cpu() for b in range(4)]
tag = os.environ.get("ORV_ATTN_SPLIT", "1")
print("SPLIT=%s  B4 repeat identical: %s   B4 vs B1 per clip: %s" % (tag, torch.equal(out4, out4b), " ".join("%.3e" % rel(out4[b:b + 1], singles[b]) for b in range(4))))
torch.save(out4, "/tmp/out4_split%s.pt" % tag)
other = "/tmp/out4_split%s.pt" % ("0" if tag == "1" else "1")
if os.path.exists(other):
    o = torch.load(other); print("  B4 split=1 vs split=0 per clip: " + " ".join("%.3e" % rel(out4[b:b + 1], o[b:b + 1]) for b in range(4)))
