import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from orv_amd import sft, schedulers
from orv_amd.optim import FusedAdamW
dev = torch.device("cuda:0")
model = bench.build_model(bench.CFG_2B, dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model.action_embed.forced_mask = torch.zeros(B, dtype=torch.bool)
lat, img, prompt, actions = bench.synthetic_inputs(B, dev, torch.bfloat16)
sched = schedulers.CogVideoXDDIMScheduler(**bench.SCHED)
opt = FusedAdamW(model.parameters(), lr=1e-5, betas=(0.9, 0.95), weight_decay=1e-3, max_grad_norm=1.0)
b = sft.Batch(lat, img, prompt, actions, None, None, torch.ones(lat.shape[1], dtype=torch.bool, device=dev), 1)
for _ in range(2):
    sft.sft_step(model, sched, opt, b)
torch.cuda.synchronize()
print("B=%d peak allocated %.1f GB, reserved %.1f GB" % (B, torch.cuda.max_memory_allocated() / 2**30, torch.cuda.max_memory_reserved() / 2**30))
