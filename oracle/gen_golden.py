"""Generate tests/golden/*.safetensors by running the REAL reference code (build container only).

    python -m oracle.gen_golden            # needs /root/reference; writes tests/golden/

Each fixture holds inputs ``in.*``, weights ``w.*`` (the reference module's state_dict) and outputs
``out.*`` of ORV's own ``forward`` / ``__call__`` (see oracle/ref_harness.py for how the reference is
imported).  Config and scalar arguments travel in the safetensors metadata as JSON.  All fp32,
seed 1234.  Fixtures are data only - no reference source text is stored.
"""
from __future__ import annotations

import json
import os
import sys

import torch
from safetensors.torch import save_file

from . import ref_harness

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

TINY = dict(num_attention_heads=2, attention_head_dim=64, in_channels=32, out_channels=16, time_embed_dim=64,
            text_embed_dim=96, num_layers=2, sample_width=12, sample_height=8, sample_frames=9,
            max_text_seq_length=8, modulate_encoder_hidden_states=True, num_control_blocks=2)


def _randomize_zero_init(model):
    """Make zero-initialised layers non-trivial, then snap every weight to a bf16-representable value so the
    fixture can store weights losslessly as bf16 (the product runs bf16 weights)."""
    g = torch.Generator().manual_seed(99)
    for name, p in model.named_parameters():
        if p.detach().abs().max() == 0:
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.05)
        p.data.copy_(p.data.to(torch.bfloat16).to(torch.float32))


def _q(x):
    """bf16-representable fp32 (inputs are stored as bf16 too)."""
    return x.to(torch.bfloat16).to(torch.float32)


def _canonical_header(path):
    """safetensors keeps ``__metadata__`` in a hash map: its entries come out in a per-process random order, which made regenerated
    fixtures differ bytewise although every tensor and every metadata value was identical.  Rewrite the header with
    the metadata entries sorted (same bytes, same length: nothing else moves)."""
    import struct
    with open(path, "r+b") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        raw = f.read(n)
        hdr = json.loads(raw)
        if "__metadata__" in hdr:
            hdr["__metadata__"] = dict(sorted(hdr["__metadata__"].items()))
        out = json.dumps(hdr, separators=(",", ":"), ensure_ascii=False).encode()
        assert len(out) <= n, (len(out), n)
        f.seek(8)
        f.write(out + b" " * (n - len(out)))


def _save(name, cfg, tensors, extra=None):
    meta = {"config": json.dumps(cfg, sort_keys=True), "extra": json.dumps(extra or {}, sort_keys=True)}
    # keys in sorted order, so that a regeneration is byte-identical (tensors and metadata were, 9 of 19 files differed in
    # serialisation order only)
    tensors = {k: v.detach().contiguous().clone() for k, v in sorted(tensors.items()) if v is not None}
    for k, v in tensors.items():
        if (k.startswith("w.") or k.startswith("in.")) and v.dtype == torch.float32 and "rope" not in k:
            assert torch.equal(v, _q(v)), k
            tensors[k] = v.to(torch.bfloat16)
    path = os.path.join(OUT, name + ".safetensors")
    save_file(tensors, path, metadata=meta)
    _canonical_header(path)
    print(f"{name}: {sum(v.numel() * v.element_size() for v in tensors.values()) / 1e6:.2f} MB")


def forward_case(cc, utils, name, cfg_over, b=2, t=3, n_act=8, mask=(False, False), cond=False, rope=False,
                 ofs=None, num_views=1, training=False, actions=True, seed=1234):
    torch.manual_seed(seed)
    cfg = {**TINY, **cfg_over}
    model = cc.CogVideoXTransformer3DModelTraj(**cfg)
    _randomize_zero_init(model)
    model.train(training)
    h, w = cfg["sample_height"], cfg["sample_width"]
    x = _q(torch.randn(b, t * num_views, cfg["in_channels"], h, w))
    e = _q(torch.randn(b, cfg["max_text_seq_length"], cfg["text_embed_dim"]))
    ts = torch.randint(0, 1000, (b,))
    ctrl = {}
    if actions:
        ctrl["actions"] = _q(torch.randn(b, n_act, 7))
    if cond:
        ctrl["depths"] = _q(torch.randn(b, t * num_views, cfg["in_channels"], h, w))
        ctrl["labels"] = _q(torch.randn(b, t * num_views, cfg["in_channels"], h, w))
    rot = None
    if rope:
        rot = utils.prepare_rotary_positional_embeddings(
            height=h * 8, width=w * 8, num_frames=t, vae_scale_factor_spatial=8, patch_size=cfg["patch_size"] if "patch_size" in cfg else 2,
            patch_size_t=cfg.get("patch_size_t"), attention_head_dim=cfg["attention_head_dim"], device=None)
    ofs_t = None if ofs is None else torch.full((1,), float(ofs))
    with torch.no_grad(), ref_harness.forced_action_mask(list(mask)):
        out, is_mask, recon = model(hidden_states=x, encoder_hidden_states=e, controls_or_guidances=dict(ctrl),
                                    timestep=ts, ofs=ofs_t, image_rotary_emb=rot, return_dict=False,
                                    num_views=num_views)
    tensors = {"in.hidden_states": x, "in.encoder_hidden_states": e, "in.timestep": ts, "out.sample": out}
    for k, v in ctrl.items():
        tensors["in." + k] = v
    if rot is not None:
        tensors["in.rope_cos"], tensors["in.rope_sin"] = rot
    if is_mask is not None:
        tensors["out.is_action_mask"] = is_mask
    if recon is not None:
        tensors["out.actions_recon"] = recon
    for k, v in model.state_dict().items():
        tensors["w." + k] = v
    full_cfg = {k: v for k, v in dict(model.config).items() if k != "kwargs"}
    _save(name, full_cfg, tensors, dict(ofs=ofs, num_views=num_views, training=training, mask=list(mask)))


def pipeline_case(cc, name, sched_cls, steps=3, guidance=1.0, seed=1234, with_actions=True, dtype=torch.float32,
                  dynamic_cfg=False, keep_steps=None):
    """ORV's own ``CogVideoXImageToVideoPipelineTraj.__call__`` (prepare_latents + denoise loop) on a tiny model.
    ``dtype=torch.bfloat16`` runs the reference the way its entry points do (inference_control_to_video.py:31,95): every
    ``randn_tensor`` draw (image-latent sample, initial latents, the DPM noise) then consumes the CPU generator in bf16,
    which is what the product does, so those fixtures can be replayed with ``generator=`` instead of pre-drawn tensors.
    ``keep_steps`` (1-based) stores only those steps' latents - the 50-step loops (cogvideox_control.py:1237 default,
    :1402-1473) keep steps 1, 10, 25, 50."""
    from . import leaf
    torch.manual_seed(seed)
    cfg = {**TINY, "sample_frames": 9}
    model = cc.CogVideoXTransformer3DModelTraj(**cfg)
    _randomize_zero_init(model)
    model.eval()
    model.to(dtype)
    sched = sched_cls(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                      clip_sample=False, set_alpha_to_one=True, prediction_type="v_prediction",
                      rescale_betas_zero_snr=True, snr_shift_scale=3.0, timestep_spacing="trailing")
    pipe = ref_harness.make_pipeline(cc, model, sched)
    b, h, w = 2, 8, 12
    image = _q(torch.randn(b, 32, 1, h, w)).to(dtype)   # un-sampled moments of the reference frame [B, 2C, F, H, W]
    e = _q(torch.randn(b, 8, 96)).to(dtype)
    ne = _q(torch.randn(b, 8, 96)).to(dtype)
    actions = _q(torch.randn(b, 8, 7)).to(dtype)
    gen = torch.Generator().manual_seed(4321)
    trace = []

    def cb(pipe_, i, t, kw):
        trace.append(kw["latents"].clone())
        return {}

    # NB: under CFG the reference doubles latents/prompts but NOT the actions (cogvideox_control.py:1409-1431),
    # so guidance_scale > 1 only runs without action controls.
    with ref_harness.forced_action_mask([False] * b):
        out = pipe(image=image, prompt=None, negative_prompt=None, height=h * 8, width=w * 8, num_frames=9,
                   num_inference_steps=steps, guidance_scale=guidance, use_dynamic_cfg=dynamic_cfg, generator=gen,
                   prompt_embeds=e,
                   negative_prompt_embeds=ne if guidance > 1 else None, output_type="latent",
                   controls_or_guidances={"actions": actions} if with_actions else {}, callback_on_step_end=cb)
    tensors = {"in.image": image.float(), "in.prompt_embeds": e.float(), "in.negative_prompt_embeds": ne.float(),
               "in.actions": actions.float(), "out.latents": out.frames.float()}
    assert len(trace) == steps
    for i, tr in enumerate(trace):
        if keep_steps is None or (i + 1) in keep_steps:
            tensors[f"out.step{i}"] = tr.float()
    for k, v in model.state_dict().items():
        tensors["w." + k] = v.float()
    full_cfg = {k: v for k, v in dict(model.config).items() if k != "kwargs"}
    extra = dict(steps=steps, guidance=guidance, gen_seed=4321, scheduler=sched_cls.__name__,
                 with_actions=with_actions, dtype=str(dtype).replace("torch.", ""),
                 dynamic_cfg=dynamic_cfg, final_guidance=float(pipe.guidance_scale))
    if keep_steps is not None:
        extra["keep_steps"] = sorted(keep_steps)
    _save(name, full_cfg, tensors, extra)


def pipeline_bf16_cases(cc):
    """The reference pipeline as its entry points run it (bf16 end to end, CPU generator): DPM (the scheduler both
    entry points install, inference_control_to_video.py:91), DDIM, and the dynamic-CFG branch (:1436-1443)."""
    from . import leaf
    bf = torch.bfloat16
    pipeline_case(cc, "pipe_dpm_bf16", leaf.CogVideoXDPMScheduler, steps=4, dtype=bf)
    pipeline_case(cc, "pipe_ddim_bf16", leaf.CogVideoXDDIMScheduler, dtype=bf)
    pipeline_case(cc, "pipe_dpm_dyncfg_bf16", leaf.CogVideoXDPMScheduler, steps=4, guidance=3.0, with_actions=False, dtype=bf,
                  dynamic_cfg=True)


def pipeline_50step_cases(cc):
    """configs[1] IS the 50-step loop (``num_inference_steps: int = 50``, cogvideox_control.py:1237; the loop :1402-1473): the
    reference run bf16 end to end for 50 steps with DDIM and with DPM (two noise draws per step from the CPU generator), latents
    kept at steps 1, 10, 25, 50 (50-step latents rel-L2 <= 5e-2 with identical noise)."""
    from . import leaf
    bf = torch.bfloat16
    pipeline_case(cc, "pipe_ddim50_bf16", leaf.CogVideoXDDIMScheduler, steps=50, dtype=bf, keep_steps=(1, 10, 25, 50))
    pipeline_case(cc, "pipe_dpm50_bf16", leaf.CogVideoXDPMScheduler, steps=50, dtype=bf, keep_steps=(1, 10, 25, 50))


def transformer_config_case():
    """The field sets of the reference's own transformer configs (config/transformer/*.json, selected by
    config/traj_image_1.4b_*.yaml:14-17 through ``from_config(load_config(path), **extra_init_kwargs)``,
    train_cogvideox_control_to_video_sft.py:286-290): D = 1792 / 28 heads, D = 1536 / 24 heads + RoPE, in_channels = 256.  Data
    only (keys and values), one entry per file."""
    import glob
    out = {}
    for path in sorted(glob.glob(os.path.join(ref_harness.REFERENCE_ROOT, "config", "transformer", "*.json"))):
        with open(path) as f:
            out[os.path.basename(path)] = json.load(f)
    with open(os.path.join(OUT, "reference_transformer_configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("transformer configs:", sorted(out))


def misc_case(cc, comp, utils):
    """Small standalone functions ORV authored: action padding + ActionEmbed/ActionRecon, crop-region helper."""
    torch.manual_seed(1234)
    ae = comp.ActionEmbed(state_dim=7, hidden_size=64, compress_ratio=4, patch_size_t=None, mask=True)
    ar = comp.ActionRecon(state_dim=7, hidden_size=64, compress_ratio=4)
    a = _q(torch.randn(3, 11, 7))
    for p in list(ae.parameters()) + list(ar.parameters()):
        p.data.copy_(_q(p.data))
    with torch.no_grad(), ref_harness.forced_action_mask([True, False, False]):
        emb, m = ae(a)
        rec = ar(emb)
    tensors = {"in.actions": a, "out.emb": emb, "out.mask": m, "out.recon": rec}
    for k, v in ae.state_dict().items():
        tensors["w.action_embed." + k] = v
    for k, v in ar.state_dict().items():
        tensors["w.action_recon." + k] = v
    crops = {f"{h}x{w}": utils.get_resize_crop_region_for_grid((h, w), 45, 30) for h, w in [(20, 30), (30, 45), (16, 24), (30, 40)]}
    _save("misc_actions", {}, tensors, dict(crops=crops))


def collate_case():
    """orv/dataset/dataset.py:2053-2142 ``CollateFunctionControl`` on cached-latent samples (what the SFT loop consumes at
    :864-886), singleview and 2-view."""
    Collate = ref_harness.load_reference_class("orv/dataset/dataset.py", "CollateFunctionControl")
    g = torch.Generator().manual_seed(1234)
    for name, v in (("collate_single", 1), ("collate_mv2", 2)):
        data = []
        for i in range(3):
            data.append({"prompt": f"p{i}", "prompt_embeds": _q(torch.randn(8, 96, generator=g)),
                         "actions": _q(torch.randn(8, 7, generator=g)),
                         "latents": _q(torch.randn(v * 3, 32, 8, 12, generator=g)),
                         "image": _q(torch.randn(v * 1, 32, 8, 12, generator=g)),
                         "latents_depth": _q(torch.randn(v * 3, 32, 8, 12, generator=g)),
                         "latents_label": _q(torch.randn(v * 3, 32, 8, 12, generator=g)),
                         "metainfo": {"num_view": v, "num_frame": 9, "episode": i}})
        out = Collate(weight_dtype=torch.bfloat16, load_tensors=True)(data)
        tensors = {}
        for i, d in enumerate(data):
            for k in ("prompt_embeds", "actions", "latents", "image", "latents_depth", "latents_label"):
                tensors[f"in.{i}.{k}"] = d[k]
        for k in ("prompt_embeds", "latents", "images"):
            tensors["out." + k] = out[k].float()
        for k, val in out["controls"].items():
            tensors["out.controls." + k] = val.float()
        _save(name, {}, tensors, dict(num_views=out["num_views"], num_frames=out["num_frames"], image_width=out["image_width"],
                                      image_height=out["image_height"], prompts=out["prompts"], n=len(data),
                                      metainfos=[d["metainfo"] for d in data]))


def bucket_sampler_case():
    """orv/dataset/dataset.py:1972-2050 ``BucketSampler`` (AST-sliced; it only needs ``random`` and a data source with
    ``resolutions`` / ``get_ref_nums_for_all_samples`` / ``get_n_views_for_all_samples``): the yielded (index, ref_num, n_view)
    order for a seeded ``random``, two epochs each (left-over buckets survive an epoch in the reference)."""
    import random

    class _Console:
        @staticmethod
        def log(*a, **k):
            pass

    Ref = ref_harness.load_reference_class(
        "orv/dataset/dataset.py", "BucketSampler",
        extra_globals={"Sampler": torch.utils.data.Sampler, "CONSOLE": _Console, "RobotDataset": object,
                       "MultiViewRobotDataset": int, "random": random})

    class DS:
        def __init__(self, refs, views):
            self.refs, self.views = refs, views
            self.resolutions = sorted(set(zip(refs, views)))

        def __len__(self):
            return len(self.refs)

        def get_ref_nums_for_all_samples(self):
            return list(self.refs)

        def get_n_views_for_all_samples(self, train=True):
            return list(self.views)

    rng = random.Random(99)
    refs = [rng.choice([1, 2, 3]) for _ in range(53)]
    views = [rng.choice([1, 3]) for _ in range(53)]
    cases = []
    for bs, shuffle, drop in [(4, True, False), (4, True, True), (4, False, False), (8, True, False), (3, False, True)]:
        random.seed(1234)
        smp = Ref(DS(refs, views), batch_size=bs, shuffle=shuffle, drop_last=drop)
        order = [list(x) for x in smp]
        second = [list(x) for x in smp]
        cases.append(dict(batch_size=bs, shuffle=shuffle, drop_last=drop, seed=1234, order=order, second_epoch=second,
                          length=len(smp)))
    with open(os.path.join(OUT, "bucket_sampler.json"), "w") as f:
        json.dump(dict(refs=refs, views=views, cases=cases), f)
    print("bucket_sampler:", [len(c["order"]) for c in cases])


def signature_case():
    """Names, order and defaults of the reference's registered ``CogVideoXTransformer3DModelTraj.__init__`` parameters and of the
    pipeline's ``__call__`` (read with ``inspect`` from the imported reference classes): the drop-in surface as data."""
    import inspect
    cc, _, _ = ref_harness.load_reference()
    out = {}
    for label, fn in (("transformer_init", cc.CogVideoXTransformer3DModelTraj.__init__),
                      ("pipeline_call", cc.CogVideoXImageToVideoPipelineTraj.__call__)):
        fn = inspect.unwrap(fn)
        params = [p_ for n, p_ in inspect.signature(fn).parameters.items() if n != "self" and p_.kind == p_.POSITIONAL_OR_KEYWORD]
        out[label] = [[p_.name, None if p_.default is inspect.Parameter.empty else (p_.default if isinstance(p_.default, (int, float, str, bool, type(None))) else repr(p_.default))] for p_ in params]
    with open(os.path.join(OUT, "signatures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("signatures:", {k: len(v) for k, v in out.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "sampler":
        return bucket_sampler_case()
    if len(sys.argv) > 1 and sys.argv[1] == "collate":      # add the data-format fixtures without touching the others
        return collate_case()
    if len(sys.argv) > 1 and sys.argv[1] == "signatures":
        return signature_case()
    if len(sys.argv) > 1 and sys.argv[1] == "configs":
        return transformer_config_case()
    cc, comp, utils = ref_harness.load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "pipe_bf16":    # round-2 additions only
        return pipeline_bf16_cases(cc)
    if len(sys.argv) > 1 and sys.argv[1] == "pipe50":       # round-5 additions only
        return pipeline_50step_cases(cc)
    forward_case(cc, utils, "fwd_actions", {})
    forward_case(cc, utils, "fwd_actions_masked", {}, mask=(True, False))
    forward_case(cc, utils, "fwd_cond", {"visual_guidance": True}, cond=True)
    forward_case(cc, utils, "fwd_noactions", {}, actions=False)
    forward_case(cc, utils, "fwd_nomod", {"modulate_encoder_hidden_states": False})
    forward_case(cc, utils, "fwd_nomod_noactions", {"modulate_encoder_hidden_states": False}, actions=False)
    forward_case(cc, utils, "fwd_rope", {"use_rotary_positional_embeddings": True}, rope=True)
    forward_case(cc, utils, "fwd_pt2_ofs", {"use_rotary_positional_embeddings": True, "patch_size_t": 2,
                                            "ofs_embed_dim": 64, "sample_frames": 13,
                                            "loaded_pretrained_model_name_or_path": "THUDM/CogVideoX1.5-5b-I2V"},
                 t=4, n_act=15, rope=True, ofs=2.0)
    forward_case(cc, utils, "fwd_multiview", {"multiview": True, "max_n_view": 3}, num_views=2)
    forward_case(cc, utils, "fwd_train_recon", {"recon_action": True}, training=True, n_act=6)
    pipeline_case(cc, "pipe_ddim", __import__("oracle.leaf", fromlist=["x"]).CogVideoXDDIMScheduler)
    pipeline_case(cc, "pipe_dpm", __import__("oracle.leaf", fromlist=["x"]).CogVideoXDPMScheduler)
    pipeline_case(cc, "pipe_ddim_cfg", __import__("oracle.leaf", fromlist=["x"]).CogVideoXDDIMScheduler, guidance=3.0,
                  with_actions=False)
    pipeline_bf16_cases(cc)
    pipeline_50step_cases(cc)
    misc_case(cc, comp, utils)
    collate_case()
    bucket_sampler_case()
    signature_case()
    transformer_config_case()


if __name__ == "__main__":
    sys.exit(main())
