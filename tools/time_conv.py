"""Times orv_conv_gemm_bf16 at the decoder's stage shapes (B=1): python tools/time_conv.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from orv_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def t(T, H, W, Cin, Cout, n=20):
    x = torch.randn(1, T + 2, H, W, Cin, device=dev).to(BF)
    w = (torch.randn(Cout, 27 * Cin, device=dev) / (27 * Cin) ** 0.5).to(BF)
    b = torch.zeros(Cout, device=dev, dtype=BF)
    out = torch.empty(T * H * W, Cout, device=dev, dtype=BF)
    f = lambda: ops.conv_gemm(x, w, b, out, 1, T + 2, H, W, Cin, T, H, W, 3, 3, 3, 1, 1, 0, 0, 2, Cout)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * T * H * W * 27 * Cin * Cout
    print(f"T={T} {H}x{W} {Cin}->{Cout}: M={T*H*W} tiles256={-(-T*H*W//256)*max(1,Cout//256)}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")
t(3, 40, 60, 512, 512); t(2, 40, 60, 512, 512); t(5, 80, 120, 512, 512); t(5, 80, 120, 512, 256); t(4, 80, 120, 256, 256)
t(9, 160, 240, 256, 256); t(8, 160, 240, 256, 256); t(9, 320, 480, 256, 128); t(9, 320, 480, 128, 128); t(8, 320, 480, 128, 128)
