"""CPU: independent pin of the diffusers leaf formulas ("oracle half-pinned").  tests/golden/leaf_kat.json holds
hand-derived known answers (tools/make_leaf_kat.py: plain math / decimal, closed forms in comments); BOTH restatements are held
to them - oracle/leaf.py (the checker) and the product's host-side tables (orv_amd.schedulers / embeddings / utils)."""
import json
import math
import os

import pytest
import torch

from conftest import GOLDEN
from oracle import leaf
from orv_amd import embeddings, schedulers

KAT = json.load(open(os.path.join(GOLDEN, "leaf_kat.json")))
SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
             set_alpha_to_one=True, prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=3.0,
             timestep_spacing="trailing")
IMPLS = [("oracle", leaf), ("product", schedulers)]


@pytest.mark.parametrize("who,mod", IMPLS)
def test_alphas_cumprod_and_timesteps(who, mod):
    s = mod.CogVideoXDDIMScheduler(**SCHED)
    ac = s.alphas_cumprod.double()
    for t, want in KAT["alphas_cumprod"].items():
        assert abs(float(ac[int(t)]) - want) <= 1e-12 * max(1.0, abs(want)), (who, t)
    assert float(ac[999]) == 0.0 and abs(float(ac[0]) - KAT["alphas_cumprod_closed_form_t0"]) < 1e-14
    for n, want in KAT["trailing_timesteps"].items():
        s.set_timesteps(int(n))
        assert s.timesteps.tolist() == want, (who, n)


def test_ddim_step_oracle():
    k = KAT["ddim"]
    s = leaf.CogVideoXDDIMScheduler(**SCHED)
    s.set_timesteps(k["N"])
    x, v = torch.tensor([k["x"]], dtype=torch.float64), torch.tensor([k["v"]], dtype=torch.float64)
    for t in (999, 499, 19):
        prev, x0 = s.step(v, t, x, return_dict=False)
        assert abs(float(prev) - k[f"t{t}"][0]) < 1e-12 and abs(float(x0) - k[f"t{t}"][1]) < 1e-12, t


def test_ddim_and_dpm_coefficients_product():
    k = KAT["ddim"]
    s = schedulers.CogVideoXDDIMScheduler(**SCHED)
    s.set_timesteps(k["N"])
    for t in (999, 499, 19):
        sa, sb, cx, cd = s.step_coefficients(t)
        x0 = sa * k["x"] - sb * k["v"]
        assert abs(cx * k["x"] + cd * x0 - k[f"t{t}"][0]) < 1e-12 and abs(x0 - k[f"t{t}"][1]) < 1e-12, t
    d = schedulers.CogVideoXDPMScheduler(**SCHED)
    d.set_timesteps(KAT["dpm"]["N"])
    for name, (t, tb) in {"t979_back999": (979, None), "t499_back519": (499, 519), "t999_first": (999, None)}.items():
        sa, sb, m1, m2, mn, m3, m4, prev = d.step_coefficients(t, tb)
        want = KAT["dpm"][name]
        assert abs(m1 - want["m1"]) < 1e-12 and abs(m2 - want["m2"]) < 1e-12 and abs(mn - want["mn"]) < 1e-12, name
        if "m3" in want:
            assert abs(m3 - want["m3"]) < 1e-10 and abs(m4 - want["m4"]) < 1e-10


def test_dpm_coefficients_oracle():
    s = leaf.CogVideoXDPMScheduler(**SCHED)
    ac = s.alphas_cumprod
    for name, (t, tb) in {"t979_back999": (979, None), "t499_back519": (499, 519)}.items():
        m1, m2, mn, m3, m4 = s.coefficients(ac[t], ac[t - 20], None if tb is None else ac[tb])
        want = KAT["dpm"][name]
        assert abs(float(m1) - want["m1"]) < 1e-12 and abs(float(m2) - want["m2"]) < 1e-12 and abs(float(mn) - want["mn"]) < 1e-12
        if tb is not None:
            assert abs(float(m3) - want["m3"]) < 1e-10 and abs(float(m4) - want["m4"]) < 1e-10
    # first step: a_t = 0 -> x_prev = sqrt(ap) x0 + sqrt(1 - ap) z, x0 = -v   (limits, no inf/nan)
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x, v = torch.tensor([0.75], dtype=torch.float64), torch.tensor([-1.25], dtype=torch.float64)
    prev, x0 = s.step(v, None, 999, None, x, generator=g)
    z = torch.randn(1, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    w = KAT["dpm"]["t999_first"]
    assert float(x0) == 1.25 and abs(float(prev) - (-w["m2"] * 1.25 + w["mn"] * float(z))) < 1e-12


def test_timestep_embedding_oracle():
    k = KAT["timestep_embedding"]
    for t in (0, 1, 999):
        got = leaf.get_timestep_embedding(torch.tensor([float(t)]), k["dim"], flip_sin_to_cos=True, downscale_freq_shift=0)[0]
        assert torch.allclose(got.double(), torch.tensor(k[f"t{t}"], dtype=torch.float64), atol=2e-4 if t == 999 else 1e-6), t


def test_sincos3d_both():
    k = KAT["sincos3d"]
    o = leaf.get_3d_sincos_pos_embed(k["D"], (k["gw"], k["gh"]), k["T"], k["spatial_scale"], 1.0).reshape(k["T"], k["gh"], k["gw"], -1)
    p = embeddings.sincos_3d(k["D"], k["gw"], k["gh"], k["T"], k["spatial_scale"], 1.0).reshape(k["T"], k["gh"], k["gw"], -1)
    for key, want in k["rows"].items():
        t, h, w = map(int, key.split(","))
        want = torch.tensor(want, dtype=torch.float64)
        assert torch.allclose(o[t, h, w].double(), want, atol=1e-6), ("oracle", key)
        assert torch.allclose(p[t, h, w].double(), want, atol=1e-6), ("product", key)


def test_rope3d_both_and_apply():
    k = KAT["rope3d"]
    crops = ((0, 0), (k["gh"], k["gw"]))
    oc, osn = leaf.get_3d_rotary_pos_embed(k["head_dim"], crops, (k["gh"], k["gw"]), k["T"])
    pc, psn = embeddings.rope_3d(k["head_dim"], crops, (k["gh"], k["gw"]), k["T"])
    for key, (cos, sin) in k["rows"].items():
        t, h, w = map(int, key.split(","))
        row = (t * k["gh"] + h) * k["gw"] + w
        for got_c, got_s, who in ((oc, osn, "oracle"), (pc, psn, "product")):
            assert torch.allclose(got_c[row].double(), torch.tensor(cos, dtype=torch.float64), atol=1e-6), (who, key)
            assert torch.allclose(got_s[row].double(), torch.tensor(sin, dtype=torch.float64), atol=1e-6), (who, key)
    a = KAT["rope_apply"]
    t, h, w = map(int, a["row"].split(","))
    row = (t * k["gh"] + h) * k["gw"] + w
    x = torch.tensor(a["x"], dtype=torch.float32).reshape(1, 1, 1, 64)
    y = leaf.apply_rotary_emb(x, (oc[row:row + 1], osn[row:row + 1]))
    assert torch.allclose(y.reshape(-1).double(), torch.tensor(a["y"], dtype=torch.float64), atol=1e-6)


def test_gelu_and_gaussian_oracle():
    for v, want in KAT["gelu_tanh"].items():
        got = torch.nn.functional.gelu(torch.tensor([float(v)], dtype=torch.float64), approximate="tanh")
        assert abs(float(got) - want) < 1e-12
    g = KAT["diag_gauss"]
    params = torch.tensor([g["mean"]] * 4 + g["logvar"], dtype=torch.float64).reshape(1, 8, 1, 1, 1)
    dist = leaf.DiagonalGaussianDistribution(params)
    got = dist.mean + dist.std * g["eps"]
    assert torch.allclose(got.reshape(-1), torch.tensor(g["sample"], dtype=torch.float64), rtol=1e-12)
    assert math.isclose(float(dist.std.reshape(-1)[3]), math.exp(10.0), rel_tol=1e-12)       # logvar 25 clamps to 20
