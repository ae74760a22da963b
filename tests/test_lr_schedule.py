"""LR schedules of the train tail (``orv_amd.optim.get_scheduler``; the reference: train_cogvideox_control_to_video_sft.py:740-747, :1107,
base_train.yaml:160-164 ``cosine_with_restarts`` / warm-up 1000 / cycles 1).  Known answers derived by hand from the published
diffusers lambdas (the leaf cannot be imported here), plus an independent cross-check against ``torch.optim.lr_scheduler.LambdaLR`` driven by
lambdas written out a second time in this file."""
import math

import pytest
import torch

from orv_amd.optim import LambdaSchedule, get_scheduler


class _Opt:                      # what FusedAdamW exposes to a scheduler
    def __init__(self, lr):
        self.param_groups = [{"lr": lr, "params": []}]


def _trace(sch, n):
    out = [sch.get_last_lr()[0]]
    for _ in range(n):
        sch.step()
        out.append(sch.get_last_lr()[0])
    return out


def test_cosine_with_restarts_known_answers():
    # warm-up 4, T = 20, two cycles: ramp 0, 1/4, ..., then 0.5 (1 + cos(pi ((2 prog) mod 1))), prog = (s - 4) / 16
    lr = _trace(get_scheduler("cosine_with_restarts", _Opt(2e-4), num_warmup_steps=4, num_training_steps=20, num_cycles=2), 22)
    assert lr[0] == 0.0 and lr[2] == pytest.approx(1e-4) and lr[4] == pytest.approx(2e-4)
    assert lr[8] == pytest.approx(1e-4)                  # prog 1/4 -> cos(pi / 2) = 0
    assert lr[6] == pytest.approx(2e-4 * 0.5 * (1 + math.cos(math.pi / 4)))
    assert lr[12] == pytest.approx(2e-4)                 # prog 1/2: the restart
    assert lr[16] == pytest.approx(1e-4)
    assert lr[20] == 0.0 and lr[22] == 0.0               # prog >= 1


def test_reference_config_schedule():
    # base_train.yaml: cosine_with_restarts, 1 cycle, warm-up 1000; the script multiplies both counts by num_processes (8)
    opt = _Opt(2e-4)
    sch = get_scheduler("cosine_with_restarts", optimizer=opt, num_warmup_steps=1000 * 8, num_training_steps=30000 * 8, num_cycles=1, power=1.0)
    assert opt.param_groups[0]["lr"] == 0.0              # the first optimizer step runs at lr 0 (LambdaLR applies lambda(0) at construction)
    for _ in range(4000):
        sch.step()
    assert sch.get_last_lr()[0] == pytest.approx(1e-4)
    sd = sch.state_dict()
    sch2 = get_scheduler("cosine_with_restarts", optimizer=_Opt(2e-4), num_warmup_steps=8000, num_training_steps=240000, num_cycles=1)
    sch2.load_state_dict(sd)
    sch.step(); sch2.step()
    assert sch2.get_last_lr() == sch.get_last_lr() and sch2.last_epoch == 4001


@pytest.mark.parametrize("name", ["constant", "constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial"])
def test_against_torch_lambdalr(name):
    W, T, nc, power, lr0 = 5, 40, 3, 2.0, 1e-3

    def lam(s):                                          # second, independent statement of the published lambdas
        if name == "constant":
            return 1.0
        if s < W:
            return s / max(1, W)
        if name == "constant_with_warmup":
            return 1.0
        if name == "linear":
            return max(0.0, (T - s) / max(1, T - W))
        p = (s - W) / max(1, T - W)
        if name == "cosine":
            return max(0.0, 0.5 * (1 + math.cos(math.pi * nc * 2 * p)))
        if name == "cosine_with_restarts":
            return 0.0 if p >= 1 else max(0.0, 0.5 * (1 + math.cos(math.pi * ((nc * p) % 1.0))))
        if s > T:
            return 1e-7 / lr0
        return ((lr0 - 1e-7) * (1 - (s - W) / (T - W)) ** power + 1e-7) / lr0
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([p], lr=lr0)
    tsch = torch.optim.lr_scheduler.LambdaLR(topt, lam)
    ours = get_scheduler(name, _Opt(lr0), num_warmup_steps=W, num_training_steps=T, num_cycles=nc, power=power)
    for s in range(T + 5):
        assert ours.get_last_lr()[0] == pytest.approx(tsch.get_last_lr()[0], rel=1e-12, abs=1e-18), (name, s)
        topt.step(); tsch.step(); ours.step()


def test_errors():
    with pytest.raises(ValueError):
        get_scheduler("cosine", _Opt(1e-3), num_warmup_steps=1)                       # needs num_training_steps
    with pytest.raises(ValueError):
        get_scheduler("linear", _Opt(1e-3), num_training_steps=10)                    # needs num_warmup_steps
    with pytest.raises(ValueError):
        get_scheduler("piecewise_constant", _Opt(1e-3), step_rules="1:10,0.1")
    assert isinstance(get_scheduler("constant", _Opt(1e-3)), LambdaSchedule)
