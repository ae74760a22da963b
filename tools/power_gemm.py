"""Socket power and shader clock (rocm-smi, sampled while the kernel loops) of ONE GEMM kernel at the FFN2 shape: t8 (row-major A) and d8 (packed A),
random against zero operands, ~6 s each.  Evidence for DESIGN.md 4.1 finding 4: on random data the GEMMs sit at the package power limit and the
clock gives; on zeros neither is reached.  usage (GPU box): python tools/power_gemm.py > gpurun_out/power_gemm.txt"""
import re, subprocess, threading, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orv_amd import ops
from orv_amd._lib import lib

dev = torch.device("cuda:0")
M, N, K = 12904, 1920, 7680


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\): ([\d.]+)", out)
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(c.group(1)) if c else None)


print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout.strip().splitlines()[-2:])
for fill in ("random", "zero"):
    g = torch.Generator(device=dev).manual_seed(1)
    A = (torch.randn(M, K, device=dev, generator=g) if fill == "random" else torch.zeros(M, K, device=dev)).to(torch.bfloat16)
    W = ((torch.randn(N, K, device=dev, generator=g) * 0.05) if fill == "random" else torch.zeros(N, K, device=dev)).to(torch.bfloat16)
    Ap = ops.pack_rows16(A, M, K)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    for kern in (("d8",) if os.environ.get("PG_D8_ONLY") else ("t8", "d8")):
        lib().orv_gemm_force_tile(3 if kern == "t8" else 5, 256, 192)
        samples, stop = [], threading.Event()

        def sampler():
            time.sleep(1.5)
            while not stop.is_set():
                samples.append(smi())
                time.sleep(0.25)
        th = threading.Thread(target=sampler)
        th.start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t0 < 6.0:
            for _ in range(200):
                ops.gemm(Ap if kern == "d8" else A, W, None, C, M, N, K, a_packed=kern == "d8")
            n += 200
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        stop.set(); th.join()
        ms = e0.elapsed_time(e1) / n
        pw = [s[0] for s in samples if s[0]]; ck = [s[1] for s in samples if s[1]]
        print(f"{fill:6s} {kern}: {ms:.4f} ms per launch = {2.0 * M * N * K / ms / 1e9:6.0f} TFLOP/s | package power W: median {sorted(pw)[len(pw) // 2]:.0f} max {max(pw):.0f} "
              f"| sclk MHz: median {sorted(ck)[len(ck) // 2]} min {min(ck)} max {max(ck)} | {len(pw)} samples")
    lib().orv_gemm_force_tile(0, 0, 0)
