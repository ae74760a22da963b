"""Micro-benchmark of orv_modulation_tables_bwd at the 2B training shape (31 tables, width 5760, E 512, B 4, T 5, text): ORV_LIB picks the build."""
import torch, time
from orv_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
n_tab, B, T, E, width = 31, 4, 5, 512, 5760
g = torch.Generator().manual_seed(0)
W = [(torch.randn(2 * width, E, generator=g) * 0.05).to(dev, BF) for _ in range(n_tab)]
ptrs = torch.tensor([w.data_ptr() for w in W], dtype=torch.int64, device=dev)
dtab = torch.randn(n_tab, B, 1 + T, width, generator=g).to(dev)
cond_v = torch.randn(B * T, E, generator=g).to(dev, BF); cond_t = torch.randn(B, E, generator=g).to(dev, BF)
dcv = torch.zeros(B * T, E, device=dev); dct = torch.zeros(B, E, device=dev)
for _ in range(3): ops.modulation_tables_bwd(dtab, cond_v, cond_t, ptrs, dcv, dct, n_tab, B, T, E, width, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.modulation_tables_bwd(dtab, cond_v, cond_t, ptrs, dcv, dct, n_tab, B, T, E, width, True)
e1.record(); torch.cuda.synchronize()
print("orv_modulation_tables_bwd: %.1f us per call (wgrad + dgrad launches)" % (e0.elapsed_time(e1) / 20 * 1000), "checksum %.6e" % float(dcv.double().sum()))
