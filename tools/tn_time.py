"""Weight-gradient GEMM dW = dY^T X at the 2B training shapes: orv_gemm_tn_bf16 against the two transposes + the NT kernel (training._wgrad's path)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from orv_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
Mtok = 12904
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K in (("QKV", 5760, 1920), ("FFN1 as dW^T", 1920, 7680), ("FFN2", 1920, 7680), ("out", 1920, 1920)):
    g = torch.Generator(device=dev).manual_seed(1)
    dY = (torch.randn(Mtok, N, device=dev, generator=g) * 0.5).to(BF); X = (torch.randn(Mtok, K, device=dev, generator=g) * 0.5).to(BF)
    dW = torch.empty(N, K, dtype=BF, device=dev); dW2 = torch.empty(N, K, dtype=BF, device=dev)
    def old():
        dYT = ops.transpose(dY, Mtok, N); XT = ops.transpose(X, Mtok, K)
        ops.gemm(dYT, XT, None, dW, N, K, dYT.shape[1])
    def new():
        ops.gemm_tn(dY, X, dW2, N, K, Mtok)
    t_old, t_new = timeit(old), timeit(new)
    err = (dW.float() - dW2.float()).abs().max().item(); ref = dW.float().abs().max().item()
    print("%-14s [%5d x %5d] K=%d : transposes + NT %7.1f us | TN %7.1f us | max diff %.3g of %.3g" % (name, N, K, Mtok, t_old, t_new, err, ref))
