"""CPU: `python bench.py --gpus N` must start its own N ranks (the driver's command shape)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_self_launches_two_ranks_dry_run():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    for mode in ("denoise", "train"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--mode", mode],
                             capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout                  # rank 0 only
        rec = json.loads(lines[0])
        assert rec == {"dry_run": True, "n_gpus": 2, "ranks_seen": 2, "mode": mode}


def test_bench_single_rank_dry_run_needs_no_launcher():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["ranks_seen"] == 1
