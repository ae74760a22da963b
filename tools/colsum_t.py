import torch, time, sys
sys.path.insert(0, "/root/repo")
from orv_amd import ops
dev = torch.device("cuda:0")
for C in (1920, 5760, 7680):
    x = torch.randn(12904, C, device=dev).to(torch.bfloat16)
    out = torch.zeros(C, device=dev)
    for _ in range(3): ops.colsum(x, out, 12904, C)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.colsum(x, out, 12904, C)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(C, f"{dt*1e6:.1f} us  {12904*C*2/dt/1e12:.2f} TB/s")
