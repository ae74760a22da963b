"""CPU: the host-side mirror of the reference surface - config, state-dict keys, checkpoint round trip, conversion of a
vanilla CogVideoX checkpoint, scheduler tables/coefficients vs the oracle, RoPE/sincos tables, C-ABI symbol export."""
import json
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden
from oracle import leaf

from orv_amd import schedulers, utils
from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj, CogVideoXImageToVideoPipelineTraj
from orv_amd.embeddings import sincos_3d

CASES = ["fwd_actions", "fwd_cond", "fwd_nomod", "fwd_rope", "fwd_pt2_ofs", "fwd_multiview", "fwd_train_recon"]


@pytest.mark.parametrize("name", CASES)
def test_state_dict_keys_equal_reference(name):
    """Golden weights are the reference module's own state_dict(): key sets and shapes must match exactly."""
    cfg, _, _, w, _ = load_golden(name)
    m = CogVideoXTransformer3DModelTraj(**cfg)
    sd = m.state_dict()
    assert set(sd) == set(w)
    assert all(sd[k].shape == w[k].shape for k in w)
    for k, v in cfg.items():
        assert getattr(m.config, k) == v


def test_c_abi_exports_every_declared_symbol():
    from orv_amd import _lib
    header = open(os.path.join(ROOT, "include", "orv_mi355.h")).read()
    declared = set(re.findall(r"\b(orv_[a-z0-9_]+)\s*\(", header)) - {"orv_groups_t", "orv_rowmap_t", "orv_gemm_t"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = _lib.lib()                      # loads without a GPU; getattr fails if a symbol is missing
    assert handle.orv_version() == 1
    for name in declared:
        assert getattr(handle, name) is not None


def test_save_load_roundtrip_and_error_convention(tmp_path):
    cfg, _, _, w, _ = load_golden("fwd_actions")
    m = CogVideoXTransformer3DModelTraj(**cfg)
    m.load_state_dict(w)
    m.save_pretrained(str(tmp_path / "transformer"), max_shard_size="1MB")        # forces sharding + index json
    assert json.load(open(tmp_path / "transformer" / "config.json"))["_class_name"] == "CogVideoXTransformer3DModelTraj"
    m2 = CogVideoXTransformer3DModelTraj.from_pretrained(str(tmp_path), subfolder="transformer", torch_dtype=torch.bfloat16)
    assert all(torch.equal(m2.state_dict()[k].float(), w[k].to(torch.bfloat16).float()) for k in w)
    # missing / unexpected keys -> RuntimeError, like cogvideox_control.py:955-967
    from safetensors.torch import load_file, save_file
    d = tmp_path / "broken"
    os.makedirs(d)
    sd = dict(w)
    sd.pop("proj_out.bias")
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    json.dump({**cfg, "_class_name": "CogVideoXTransformer3DModelTraj"}, open(d / "config.json", "w"))
    with pytest.raises(RuntimeError, match="not found in pretrained weights"):
        CogVideoXTransformer3DModelTraj.from_pretrained(str(d))


def test_vanilla_cogvideox_2b_conversion(tmp_path):
    """THUDM*CogVideoX*-2b* folder (16 input channels, class CogVideoXTransformer3DModel) -> 32 channels, new half zero."""
    base = dict(num_attention_heads=2, attention_head_dim=64, in_channels=16, out_channels=16, time_embed_dim=64,
                text_embed_dim=96, num_layers=1, sample_width=12, sample_height=8, sample_frames=9, max_text_seq_length=8)
    src = CogVideoXTransformer3DModelTraj(**base, modulate_encoder_hidden_states=True)
    keep = {k: v for k, v in src.state_dict().items() if not k.startswith("action_embed")}
    d = tmp_path / "THUDM" / "CogVideoX-2b" / "transformer"
    os.makedirs(d)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in keep.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    json.dump({**base, "_class_name": "CogVideoXTransformer3DModel"}, open(d / "config.json", "w"))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        m = CogVideoXTransformer3DModelTraj.from_pretrained("THUDM/CogVideoX-2b", subfolder="transformer", sample_height=8,
                                                            sample_width=12, sample_frames=9,
                                                            modulate_encoder_hidden_states=True,
                                                            loaded_pretrained_model_name_or_path="THUDM/CogVideoX-2b")
    finally:
        os.chdir(cwd)
    assert m.config.in_channels == 32 and m.config.from_t2v
    wgt = m.patch_embed.proj.weight
    assert torch.equal(wgt[:, :16], keep["patch_embed.proj.weight"]) and torch.all(wgt[:, 16:] == 0)
    with pytest.raises(RuntimeError, match="modulate_encoder_hidden_states"):
        CogVideoXTransformer3DModelTraj(**base, loaded_pretrained_model_name_or_path="THUDM/CogVideoX-2b")


def test_same_class_checkpoint_converts_when_kwargs_add_modules(tmp_path):
    """train...sft.py:276-283 passes extra_init_kwargs (multiview / recon_action / visual_guidance) when finetuning FROM an ORV
    checkpoint: the strict load fails, the reference falls through to from_config(base.config, **kwargs) +
    load_state_dict(strict=False) and, for a non-multiview source, copies transformer_blocks[i] into mv_blocks[i]
    (cogvideox_control.py:970-1050)."""
    cfg, _, _, w, _ = load_golden("fwd_actions")
    src = CogVideoXTransformer3DModelTraj(**cfg)
    src.load_state_dict(w)
    src.save_pretrained(str(tmp_path / "transformer"))
    m = CogVideoXTransformer3DModelTraj.from_pretrained(str(tmp_path), subfolder="transformer", torch_dtype=torch.float32,
                                                        multiview=True, max_n_view=3, recon_action=True)
    assert m.config.multiview and m.config.recon_action and m.action_recon is not None
    sd = m.state_dict()
    for k, v in w.items():                                  # everything the checkpoint holds arrives unchanged
        assert torch.equal(sd[k], v), k
    for i, blk in enumerate(m.mv_blocks):                   # stage-3 initialisation: attention + norm1 copied from the 3-D block
        ref = m.transformer_blocks[i].state_dict()
        for k, v in blk.state_dict().items():
            if k in ref:
                assert torch.equal(v, ref[k]), (i, k)
        assert blk.proj_out.weight.abs().max() == 0        # zero-init stays (not in the 3-D block)
    # only the multiview blocks train afterwards (:641-656)
    assert all(p.requires_grad == n.startswith("mv_blocks.") for n, p in m.named_parameters())
    # a multiview checkpoint is loaded directly next time (no copy): change an mv weight, save, reload
    with torch.no_grad():
        m.mv_blocks[0].attn1.to_q.weight.add_(1.0)
    m.save_pretrained(str(tmp_path / "mv" / "transformer"))
    m2 = CogVideoXTransformer3DModelTraj.from_pretrained(str(tmp_path / "mv"), subfolder="transformer")
    assert torch.equal(m2.mv_blocks[0].attn1.to_q.weight, m.mv_blocks[0].attn1.to_q.weight)
    # a shape mismatch is still an error in the conversion path
    with pytest.raises(RuntimeError):
        CogVideoXTransformer3DModelTraj.from_pretrained(str(tmp_path), subfolder="transformer", time_embed_dim=32)


def test_pipeline_rejects_wrong_transformer_type():
    with pytest.raises(ValueError, match="must be of type CogVideoXTransformer3DModelTraj"):
        CogVideoXImageToVideoPipelineTraj(transformer=torch.nn.Linear(2, 2), scheduler=None)


KW = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
          set_alpha_to_one=True, prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=3.0,
          timestep_spacing="trailing")


@pytest.mark.parametrize("n", [50, 3, 7])
def test_scheduler_tables_and_coefficients_match_oracle(n):
    ours, ref = schedulers.CogVideoXDPMScheduler(**KW), leaf.CogVideoXDPMScheduler(**KW)
    assert torch.allclose(ours.alphas_cumprod, ref.alphas_cumprod, rtol=0, atol=1e-15)
    ours.set_timesteps(n), ref.set_timesteps(n)
    ts = ref.timesteps.tolist()
    assert ours.timesteps.tolist() == ts
    for i, t in enumerate(ts):
        tb = ts[i - 1] if i else None
        sa, sb, m1, m2, mn, m3, m4, prev = ours.step_coefficients(t, tb)
        a_t = ref.alphas_cumprod[t]
        a_p = ref.alphas_cumprod[prev] if prev >= 0 else ref.final_alpha_cumprod
        r = ref.coefficients(a_t, a_p, ref.alphas_cumprod[tb] if tb is not None else None)
        want = [float(r[0]), float(r[1]), float(r[2])]
        for g, w_ in zip((m1, m2, mn), want):
            assert abs(g - w_) <= 1e-9 * max(1.0, abs(w_))
        if tb is not None and prev >= 0:
            assert abs(m3 - float(r[3])) <= 1e-9 * max(1, abs(float(r[3]))) and abs(m4 - float(r[4])) <= 1e-9 * max(1, abs(float(r[4])))
    d, dr = schedulers.CogVideoXDDIMScheduler(**KW), leaf.CogVideoXDDIMScheduler(**KW)
    d.set_timesteps(n), dr.set_timesteps(n)
    x, v = torch.randn(4, 3), torch.randn(4, 3)
    for t in ts:
        sa, sb, cx, cd = d.step_coefficients(t)
        want = dr.step(v, t, x, return_dict=False)[0]
        got = cx * x + cd * (sa * x - sb * v)
        torch.testing.assert_close(got, want.float(), atol=1e-5, rtol=1e-5)
    # training-side helpers
    t3 = torch.tensor([0, 500, 999])
    x0, nz = torch.randn(3, 2, 4), torch.randn(3, 2, 4)
    torch.testing.assert_close(d.add_noise(x0, nz, t3), dr.add_noise(x0, nz, t3))
    torch.testing.assert_close(d.get_velocity(x0, nz, t3), dr.get_velocity(x0, nz, t3))


def test_positional_tables_match_oracle():
    a = sincos_3d(1920, 30, 20, 5, 1.875, 1.0)
    b = leaf.get_3d_sincos_pos_embed(1920, (30, 20), 5, 1.875, 1.0).flatten(0, 1).float()
    assert torch.equal(a, b)
    for (h, w, f, pt) in [(320, 480, 5, None), (256, 384, 8, 2), (480, 640, 5, None)]:
        cos, sin = utils.prepare_rotary_positional_embeddings(h, w, f, patch_size_t=pt)
        gh, gw = h // 16, w // 16
        if pt is None:
            crops = utils.get_resize_crop_region_for_grid((gh, gw), 45, 30)
            rc, rs = leaf.get_3d_rotary_pos_embed(64, crops, (gh, gw), f)
        else:
            rc, rs = leaf.get_3d_rotary_pos_embed(64, None, (gh, gw), (f + pt - 1) // pt, grid_type="slice", max_size=(30, 45))
        assert torch.equal(cos, rc) and torch.equal(sin, rs)
    _, extra, _, _, _ = load_golden("misc_actions")      # crop helper pinned by the reference's own function
    for key, want in extra["crops"].items():
        h, w = map(int, key.split("x"))
        got = utils.get_resize_crop_region_for_grid((h, w), 45, 30)
        assert [list(got[0]), list(got[1])] == [list(want[0]), list(want[1])]


def test_c_abi_argument_validation_and_no_gpu_behaviour():
    """Every entry point validates before touching the device: bad arguments come back as a non-zero code with a message
    in orv_last_error() (no exception, no abort); without a GPU orv_device_check fails loudly instead of falling back."""
    import ctypes
    from orv_amd import _lib
    h = _lib.lib()
    err = lambda: h.orv_last_error().decode()
    g = _lib.Gemm()
    assert h.orv_gemm_bf16(ctypes.byref(g), None) != 0 and "null operand" in err()
    buf = (ctypes.c_uint16 * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    g.A, g.W, g.C, g.M, g.N, g.K, g.lda, g.ldw, g.ldc = p, p, p, 8, 64, 100, 104, 104, 64
    assert h.orv_gemm_bf16(ctypes.byref(g), None) != 0 and "K=100 must be a multiple of 64" in err()
    assert h.orv_qkv_prep(p, p, None, None, None, None, None, None, 1, 100, 2, 0, 64, 1e-6, 1.0, None) != 0
    assert "s_pad=64 must be S=100 rounded up to 64" in err()
    assert h.orv_colsum(p, 12, p, 4, 12, None) != 0 and "multiples of 8" in err()
    assert h.orv_adamw_flat(p, p, p, p, 1000, p, p, 1, 1e-3, 0.9, 0.95, 1e-8, 0.0, 1, None, None) != 0
    assert "multiple of 2048" in err()
    assert h.orv_gather_rows(p, 64, p, p, 64, 4, 60, None) != 0 and "bad arguments" in err()
    if not torch.cuda.is_available():
        assert h.orv_device_check(0) != 0 and err()
        from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj
        cfg, _, ins, w, _ = load_golden("fwd_noactions")
        m = CogVideoXTransformer3DModelTraj(**cfg)
        with pytest.raises(RuntimeError, match="MI355X only"):
            m(ins["hidden_states"], ins["encoder_hidden_states"], {}, ins["timestep"])


@pytest.mark.parametrize("name", ["collate_single", "collate_mv2"])
def test_collate_matches_reference_golden(name, tmp_path):
    """orv_amd.data.CollateFunctionControl against the reference's class run on the same samples (fixture generated by
    oracle/gen_golden.py from orv/dataset/dataset.py:2053-2142), and the cached-latent file readers on files written in the
    reference's on-disk layout ([2C, F, h, w] moments)."""
    from orv_amd import data
    _, extra, ins, _, outs = load_golden(name)
    n = extra["n"]
    samples = []
    for i in range(n):
        d = {k.split(".", 1)[1]: v for k, v in ins.items() if k.startswith(f"{i}.")}
        samples.append({**d, "prompt": extra["prompts"][i], "metainfo": extra["metainfos"][i]})
    got = data.CollateFunctionControl(torch.bfloat16, True)(samples)
    for k in ("prompt_embeds", "latents", "images"):
        assert got[k].dtype == torch.bfloat16 and torch.equal(got[k].float(), outs[k]), k
    for k in ("actions", "latents_depth", "latents_label"):
        assert torch.equal(got["controls"][k].float(), outs["controls." + k]), k
    for k in ("num_views", "num_frames", "image_width", "image_height", "prompts"):
        assert got[k] == extra[k], k
    # file readers: one clip + per-view controls round-trip through the .pt layout
    v = extra["num_views"]
    lat = samples[0]["latents"]                                    # [v*F, 2C, h, w]
    F = lat.shape[0] // v
    torch.save(lat[:F].permute(1, 0, 2, 3).contiguous(), tmp_path / "clip.pt")
    torch.save(samples[0]["image"][:1].permute(1, 0, 2, 3).contiguous(), tmp_path / "ref.pt")
    clip = data.load_latent_clip(str(tmp_path), "clip.pt", "ref.pt", frame_ids=list(range(4 * F)))
    assert torch.equal(clip["latents"], lat[:F]) and clip["image"].shape == (1, *lat.shape[1:])
    with pytest.raises(RuntimeError, match="mismatched latent video"):
        data.load_latent_clip(str(tmp_path), "clip.pt", "ref.pt", frame_ids=[4 * F + 3], is_sliced=False)
    paths = []
    for vi in range(v):
        torch.save(samples[0]["latents_depth"][vi * F:(vi + 1) * F].permute(1, 0, 2, 3).contiguous(), tmp_path / f"d{vi}.pt")
        paths.append(f"d{vi}.pt")
    ctl = data.load_latent_controls(str(tmp_path), ["depth"], latent_depth_paths=paths)
    assert torch.equal(ctl["latents_depth"], samples[0]["latents_depth"]) and "latents_label" not in ctl


def test_bucket_sampler_matches_reference_golden():
    """orv_amd.data.BucketSampler yields the reference class's exact (index, ref_num, n_view) stream for a seeded `random`
    (fixture produced by the AST-sliced reference class, oracle/gen_golden.py::bucket_sampler_case), both epochs, including
    the reference's quirks (left-overs dropped when shuffle=False; left-over buckets carried into the next epoch)."""
    import random
    from orv_amd.data import BucketSampler
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "bucket_sampler.json")))

    class DS:
        resolutions = sorted(set(zip(fx["refs"], fx["views"])))

        def __len__(self):
            return len(fx["refs"])

        def get_ref_nums_for_all_samples(self):
            return list(fx["refs"])

        def get_n_views_for_all_samples(self, train=True):
            return list(fx["views"])

    for c in fx["cases"]:
        random.seed(c["seed"])
        s = BucketSampler(DS(), batch_size=c["batch_size"], shuffle=c["shuffle"], drop_last=c["drop_last"])
        assert [list(x) for x in s] == c["order"], c
        assert [list(x) for x in s] == c["second_epoch"], c
        assert len(s) == c["length"]
        # every full batch is shape-homogeneous: what keeps the per-rank batches rectangular
        full = c["order"][: len(c["order"]) // c["batch_size"] * c["batch_size"]] if c["drop_last"] else []
        for i in range(0, len(full), c["batch_size"]):
            assert len({(r, v) for _, r, v in full[i:i + c["batch_size"]]}) == 1


def test_attention_score_bound_holds_on_random_layernorm_outputs():
    """Attention.score_bound (feeds the fixed-shift softmax): |q' . k'| * scale * log2 e <= bound for LayerNorm outputs with random
    affine parameters, RoPE-rotated or not - including inputs built to be as aligned as possible (q = k direction)."""
    import math
    import torch
    from orv_amd.cogvideox_control import Attention, prime_score_bounds
    g = torch.Generator().manual_seed(0)
    for trial in range(6):
        at = Attention(128, 2, 64, True, True)
        with torch.no_grad():
            for p_, sc in ((at.norm_q.weight, 1.0), (at.norm_q.bias, 0.5), (at.norm_k.weight, 1.0), (at.norm_k.bias, 0.5)):
                p_.copy_(torch.randn(64, generator=g) * sc * (0.3 + trial * 0.3))
        bound, bound_plain = at.score_bound(0.125), at.score_bound(0.125, rope=False)
        # round 6: per-channel (no RoPE) <= per-pair (any rotation) <= the round-5 product of per-vector maxima
        gq_, bq_, gk_, bk_ = (t.detach() for t in (at.norm_q.weight, at.norm_q.bias, at.norm_k.weight, at.norm_k.bias))
        old = 1.02 * 0.125 * 1.4426950408889634 * float((8 * gq_.abs().max() + bq_.norm()) * (8 * gk_.abs().max() + bk_.norm()))
        assert bound_plain <= bound * (1 + 1e-6) and bound <= old * (1 + 1e-6), (bound_plain, bound, old)
        x = torch.randn(4000, 64, generator=g) * torch.rand(4000, 1, generator=g) * 10
        x[:64] = torch.eye(64) * 100                           # one-hot rows: xhat puts almost all its norm on one channel
        x[64:128] = torch.sign(at.norm_q.weight.detach() * at.norm_k.weight.detach()) * torch.rand(64, 64, generator=g)
        ln = lambda v, n: torch.nn.functional.layer_norm(v, (64,), n.weight, n.bias, 1e-6)
        with torch.no_grad():
            qq, kk = ln(x, at.norm_q), ln(x, at.norm_k)
            ang = torch.rand(4000, 32, generator=g) * 6.28
            rot = lambda v, a: torch.stack([v[:, 0::2] * a.cos() - v[:, 1::2] * a.sin(), v[:, 0::2] * a.sin() + v[:, 1::2] * a.cos()], -1).flatten(1)
            # adversarial rows for the per-channel term: all of xhat's norm on the channel (pair) where |gamma_q gamma_k| peaks
            cstar = int((gq_ * gk_).abs().argmax())
            x[128:130] = 0
            x[128, cstar], x[129, cstar ^ 1] = 50.0, -50.0
            qq, kk = ln(x, at.norm_q), ln(x, at.norm_k)
            for a_, b_, bd in ((qq, kk, bound_plain), (qq, kk, bound), (rot(qq, ang), rot(kk, ang.flip(0)), bound)):
                s = (a_ @ b_.t()).abs().max().item() * 0.125 * 1.4426950408889634
                assert s <= bd, (trial, s, bd)
    many = [Attention(128, 2, 64, True, True) for _ in range(5)]
    prime_score_bounds(many, 0.125)                            # batched form == per-module form
    for m in many:
        assert m._bound is not None and math.isclose(m.score_bound(0.125), 1.02 * 8 * 8 * 0.125 * 1.4426950408889634, rel_tol=1e-6)


def test_constructor_and_call_signatures_equal_the_reference():
    """Names, order and defaults of ``CogVideoXTransformer3DModelTraj.__init__`` (the 37 registered config arguments, positional use
    included) and of ``CogVideoXImageToVideoPipelineTraj.__call__`` against tests/golden/signatures.json, which
    oracle/gen_golden.py wrote with ``inspect`` from the imported reference classes (cogvideox_control.py:452-494, :1228-1257)."""
    import inspect
    import json
    import os
    from orv_amd.cogvideox_control import CogVideoXImageToVideoPipelineTraj, CogVideoXTransformer3DModelTraj
    with open(os.path.join(os.path.dirname(__file__), "golden", "signatures.json")) as f:
        ref = json.load(f)

    def sig(fn):
        out = []
        for n, p_ in inspect.signature(fn).parameters.items():
            if n == "self" or p_.kind != p_.POSITIONAL_OR_KEYWORD:
                continue
            d = p_.default
            out.append([n, None if d is inspect.Parameter.empty else (d if isinstance(d, (int, float, str, bool, type(None))) else repr(d))])
        return out
    assert sig(CogVideoXTransformer3DModelTraj.__init__) == ref["transformer_init"] and len(ref["transformer_init"]) == 37
    assert sig(CogVideoXImageToVideoPipelineTraj.__call__) == ref["pipeline_call"]
    m = CogVideoXTransformer3DModelTraj(2, 64, 32, num_layers=1, sample_width=12, sample_height=8, sample_frames=9)
    assert (m.config.num_attention_heads, m.config.attention_head_dim, m.config.in_channels, m.config.num_layers) == (2, 64, 32, 1)


def test_graph_weights_key_sees_replaced_parameters():
    """``GraphedTransformer._weights_version`` cached the parameter LIST and re-collected it only when the
    parameter COUNT changed, so ``m.weight = nn.Parameter(...)`` and ``load_state_dict(assign=True)`` (same count, new objects whose
    storage the captured graph has never seen) left the key unchanged and the HIP graph replayed the old weights.  The key now
    follows torch's parameter-registration hook.  CPU-only: the key never touches the GPU."""
    from orv_amd.cogvideox_control import GraphedTransformer
    cfg, _, _, w, _ = load_golden("fwd_actions")
    m = CogVideoXTransformer3DModelTraj(**cfg)
    m.load_state_dict(w)
    gt = GraphedTransformer(m)
    k0 = gt._weights_version()
    assert gt._weights_version() == k0                                  # stable without changes
    lin = m.transformer_blocks[0].ff.net[2]
    lin.weight = torch.nn.Parameter(lin.weight.detach().clone() * 2)    # replaced object, same count
    k1 = gt._weights_version()
    assert k1 != k0
    m.load_state_dict({k: v.clone() for k, v in w.items()}, assign=True)   # every parameter replaced, same count
    k2 = gt._weights_version()
    assert k2 != k1 and k2 != k0
    with torch.no_grad():
        m.proj_out.bias.add_(1.0)                                       # in-place edit: _version
    k3 = gt._weights_version()
    assert k3 != k2
    lin.weight.data = lin.weight.data.clone()                           # storage move: data_ptr
    assert gt._weights_version() != k3


def test_softmax_kernel_census_follows_the_qk_layernorm_gains():
    """``model.softmax_kernel_census()`` (round 6; `bench.py` prints it as `attention_softmax`): which layers' CURRENT norm_q / norm_k parameters
    keep the fixed-shift softmax kernel (bound <= orv_attention_static_limit) and which fall to the online kernel.  Host arithmetic + one C-ABI
    constant: runs without a GPU."""
    import torch
    from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj
    m = CogVideoXTransformer3DModelTraj(num_attention_heads=2, attention_head_dim=64, in_channels=8, out_channels=4, num_layers=4, text_embed_dim=32,
                                        time_embed_dim=32, sample_width=12, sample_height=8, sample_frames=9, max_text_seq_length=8)
    c0 = m.softmax_kernel_census()
    assert c0["layers_static"] == 4 and c0["layers_online"] == 0 and c0["static_limit_log2"] == 90.0
    assert abs(c0["max_bound_log2"] - 1.02 * 64 * 0.125 * 1.4426950408889634) < 0.02          # gamma 1, beta 0: 64 x scale x log2 e
    with torch.no_grad():
        m.transformer_blocks[1].attn1.norm_q.weight[7] = 5.0          # one outlier channel in q AND the same channel in k: 25 x 64 x 0.18 > 90
        m.transformer_blocks[1].attn1.norm_k.weight[7] = 5.0
        m.transformer_blocks[2].attn1.norm_q.weight[7] = 5.0          # outliers in DIFFERENT channels: the per-channel bound keeps the layer static
        m.transformer_blocks[2].attn1.norm_k.weight[40] = 5.0
    c1 = m.softmax_kernel_census()
    assert c1["layers_static"] == 3 and c1["layers_online"] == 1, c1
    c2 = m.softmax_kernel_census(rope=True)                           # per pair (7 and 40 are in different pairs too)
    assert c2["layers_static"] == 3 and c2["layers_online"] == 1, c2
