import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    """-> (config, extra, inputs, weights(fp32), outputs) of a tests/golden fixture."""
    from safetensors import safe_open
    path = os.path.join(GOLDEN, name + ".safetensors")
    ins, wts, outs = {}, {}, {}
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
        for k in f.keys():
            t = f.get_tensor(k)
            if k.startswith("in."):
                ins[k[3:]] = t.float() if t.is_floating_point() else t
            elif k.startswith("w."):
                wts[k[2:]] = t.float()
            else:
                outs[k[4:]] = t
    return json.loads(meta["config"]), json.loads(meta["extra"]), ins, wts, outs


@pytest.fixture(scope="session")
def golden():
    return load_golden
