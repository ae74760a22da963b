"""ISA guard (CPU, cross-compile only): gemm_d8.hip issues its A loads from inline asm and waits for them by hand, so the compiler may legally read
or copy a destination register before the data has landed (cdna_hip_programming.md 5.7).  Round 6 met exactly that in one instantiation of the
refactored kernel (a v_mov of in-flight A registers on a loop edge: garbage rows at BN = 128); the stream cursors now wrap without a branch and
`tools/check_async_load_hazard.py` scans every instantiation of the file for reads of in-flight registers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_d8_kernels_never_read_an_in_flight_load_destination(tmp_path):
    asm = tmp_path / "gemm_d8.s"
    src = os.path.join(ROOT, "orv_amd", "csrc", "gemm_d8.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-w", "-S", "--cuda-device-only",
                        "-o", str(asm), src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_load_hazard.py"), str(asm), "gemm_d8"],
                       capture_output=True, text=True, timeout=300)
    kernels = [l for l in c.stdout.splitlines() if l.startswith("_Z")]
    assert len(kernels) >= 22, c.stdout[-2000:]                     # 14 x gemm_d8_kernel + 8 x gemm_d8r192_kernel
    assert c.returncode == 0 and all(l.rstrip().endswith("CLEAN") for l in kernels), c.stdout[-3000:]
