"""The pipeline's call surface as the reference's entry points use it
(/root/reference/orv/pipeline/inference_control_to_video.py:71-146): from_pretrained / save_pretrained / scheduler swap /
.to / vae + T5 delegation / PIL in, PIL out.  VAE and T5 ARITHMETIC is out of scope: the tests attach stand-in
objects with the diffusers / transformers interfaces and check the exact calls the pipeline makes on them.  CPU tests stop
before the transformer forward (GPU only); the `-m gpu` test runs the script body end to end."""
import json
import os

import numpy as np
import PIL.Image
import pytest
import torch

from conftest import load_golden
from orv_amd import schedulers
from orv_amd.cogvideox_control import (CogVideoXImageToVideoPipelineTraj, CogVideoXTransformer3DModelTraj, FrozenConfig)
from orv_amd.components import VideoProcessor

SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
             set_alpha_to_one=True, prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=3.0,
             timestep_spacing="trailing")


class FakeLatentDist:
    def __init__(self, moments):
        self.mean, self.logvar = moments.chunk(2, dim=1)

    def sample(self, generator=None):
        eps = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + torch.exp(0.5 * self.logvar.clamp(-30, 20)) * eps.to(self.mean.device)


class FakeVAE:
    """diffusers AutoencoderKLCogVideoX interface: .config, encode(x[B,3,F,H,W]).latent_dist, decode(z[B,16,f,h,w]).sample,
    enable_slicing/tiling, eval, to."""

    def __init__(self):
        self.config = FrozenConfig(block_out_channels=[128, 256, 256, 512], temporal_compression_ratio=4,
                                   scaling_factor=1.15258426, latent_channels=16, invert_scale_latents=True)
        self.calls = []

    def encode(self, x):
        self.calls.append(("encode", tuple(x.shape), x.dtype))
        b, c, f, h, w = x.shape
        pooled = torch.nn.functional.avg_pool2d(x.float().reshape(b * c * f, 1, h, w), 8).reshape(b, c, f, h // 8, w // 8)
        moments = torch.cat([pooled.repeat(1, 6, 1, 1, 1)[:, :16], torch.full((b, 16, f, h // 8, w // 8), -4.0,
                                                                               device=x.device)], dim=1).to(x.dtype)
        return type("EncOut", (), {"latent_dist": FakeLatentDist(moments)})()

    def decode(self, z):
        self.calls.append(("decode", tuple(z.shape), z.dtype))
        b, c, f, h, w = z.shape
        rgb = z[:, :3].float().repeat_interleave(8, dim=3).repeat_interleave(8, dim=4)
        frames = torch.cat([rgb[:, :, :1], rgb[:, :, 1:].repeat_interleave(4, dim=2)], dim=2)   # 1 + 4 (f - 1) frames
        return type("DecOut", (), {"sample": torch.tanh(frames)})()

    def enable_slicing(self):
        self.calls.append(("enable_slicing",))

    def enable_tiling(self):
        self.calls.append(("enable_tiling",))

    def eval(self):
        return self

    def to(self, device=None, dtype=None):
        self.calls.append(("to", str(device), dtype))
        return self


class FakeTokenizer:
    def __init__(self):
        self.calls = []

    def __call__(self, prompt, **kw):
        self.calls.append((list(prompt), kw))
        ids = torch.zeros(len(prompt), kw["max_length"], dtype=torch.long)
        for i, p in enumerate(prompt):
            ids[i, : min(len(p), kw["max_length"])] = torch.tensor([ord(ch) % 97 + 1 for ch in p[: kw["max_length"]]])
        return type("Enc", (), {"input_ids": ids})()


class FakeT5(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.emb = torch.nn.Embedding(128, dim)

    @property
    def device(self):
        return self.emb.weight.device

    def forward(self, ids):
        return (self.emb(ids),)


def tiny_model():
    cfg, _, _, w, _ = load_golden("pipe_ddim")
    m = CogVideoXTransformer3DModelTraj(**cfg)
    m.load_state_dict(w)
    return cfg, m


def test_from_pretrained_save_pretrained_roundtrip(tmp_path):
    cfg, m = tiny_model()
    pipe = CogVideoXImageToVideoPipelineTraj(transformer=m, scheduler=schedulers.CogVideoXDDIMScheduler(**SCHED), vae=FakeVAE())
    pipe.save_pretrained(str(tmp_path / "ckpt"))
    idx = json.load(open(tmp_path / "ckpt" / "model_index.json"))
    assert idx["_class_name"] == "CogVideoXImageToVideoPipelineTraj" and idx["scheduler"][1] == "CogVideoXDDIMScheduler"
    assert os.path.exists(tmp_path / "ckpt" / "transformer" / "config.json")
    # inference_control_to_video.py:71-72: the whole pipeline from one directory
    p2 = CogVideoXImageToVideoPipelineTraj.from_pretrained(str(tmp_path / "ckpt"), torch_dtype=torch.bfloat16)
    assert isinstance(p2.scheduler, schedulers.CogVideoXDDIMScheduler) and p2.scheduler.config.snr_shift_scale == 3.0
    assert p2.transformer.dtype == torch.bfloat16
    assert all(torch.equal(a, b.to(torch.bfloat16)) for a, b in zip(p2.transformer.state_dict().values(), m.state_dict().values()))
    # :74-84: transformer loaded separately and handed in, components passed as objects
    tr = CogVideoXTransformer3DModelTraj.from_pretrained(str(tmp_path / "ckpt"), subfolder="transformer", torch_dtype=torch.bfloat16)
    vae = FakeVAE()
    p3 = CogVideoXImageToVideoPipelineTraj.from_pretrained(str(tmp_path / "ckpt"), transformer=tr, vae=vae, torch_dtype=torch.bfloat16)
    assert p3.transformer is tr and p3.vae is vae
    # :91 scheduler swap, :95-108 configuration calls
    p3.scheduler = schedulers.CogVideoXDPMScheduler.from_config(p3.scheduler.config, timestep_spacing="trailing")
    assert isinstance(p3.scheduler, schedulers.CogVideoXDPMScheduler) and p3.scheduler.config.timestep_spacing == "trailing"
    p3.to("cpu", dtype=torch.bfloat16)
    p3.vae.enable_slicing(), p3.vae.enable_tiling(), p3.transformer.eval(), p3.vae.eval()
    p3.transformer.gradient_checkpointing = False
    assert ("to", "cpu", torch.bfloat16) in vae.calls
    assert p3.invert_scale_latents is True
    p3.vae.config = FrozenConfig({**p3.vae.config, "invert_scale_latents": False})      # the script's override, read at call time
    assert p3.invert_scale_latents is False
    with pytest.raises(ValueError, match="must be of type CogVideoXTransformer3DModelTraj"):
        CogVideoXImageToVideoPipelineTraj(transformer=torch.nn.Linear(1, 1), scheduler=p3.scheduler)


def test_encode_prompt_delegates_to_attached_t5():
    cfg, m = tiny_model()
    tok, t5 = FakeTokenizer(), FakeT5(cfg["text_embed_dim"])
    pipe = CogVideoXImageToVideoPipelineTraj(tokenizer=tok, text_encoder=t5, transformer=m,
                                             scheduler=schedulers.CogVideoXDDIMScheduler(**SCHED))
    pe, ne = pipe.encode_prompt("pick up the cup", "blurry", do_classifier_free_guidance=True, num_videos_per_prompt=2,
                                max_sequence_length=8, device=torch.device("cpu"), dtype=torch.float32)
    assert pe.shape == (2, 8, cfg["text_embed_dim"]) and ne.shape == pe.shape
    (p_call, p_kw), (n_call, n_kw) = tok.calls
    assert p_call == ["pick up the cup"] and n_call == ["blurry"]
    assert p_kw == dict(padding="max_length", max_length=8, truncation=True, add_special_tokens=True, return_tensors="pt")
    ids = tok(["pick up the cup"], max_length=8).input_ids
    assert torch.equal(pe[0], t5(ids)[0][0]) and torch.equal(pe[0], pe[1])
    # no guidance: the negative prompt is never encoded
    tok.calls.clear()
    pe, ne = pipe.encode_prompt("a", "b", do_classifier_free_guidance=False, max_sequence_length=8, device=torch.device("cpu"),
                                dtype=torch.float32)
    assert ne is None and len(tok.calls) == 1
    # without the objects the call explains itself
    bare = CogVideoXImageToVideoPipelineTraj(transformer=m, scheduler=schedulers.CogVideoXDDIMScheduler(**SCHED))
    with pytest.raises(NotImplementedError, match="tokenizer"):
        bare.encode_prompt("a", None, do_classifier_free_guidance=False)
    with pytest.raises(ValueError, match="Provide either `prompt` or `prompt_embeds`"):
        bare.check_inputs(torch.zeros(1, 32, 1, 8, 12), None, 64, 96, None, ["latents"])
    with pytest.raises(ValueError, match="divisible by 8"):
        bare.check_inputs(torch.zeros(1, 32, 1, 8, 12), "x", 60, 96, None, ["latents"])


def test_video_processor_pil_roundtrip_and_latent_passthrough():
    vp = VideoProcessor(vae_latent_channels=16, vae_scale_factor=8)
    rng = np.random.default_rng(0)
    img = PIL.Image.fromarray(rng.integers(0, 255, (64, 96, 3), dtype=np.uint8))
    x = vp.preprocess(img, height=64, width=96)
    assert x.shape == (1, 3, 64, 96) and x.min() >= -1 and x.max() <= 1
    assert torch.allclose(x[0], torch.from_numpy(np.array(img)).permute(2, 0, 1).float() / 127.5 - 1, atol=1e-6)
    assert vp.preprocess(img, height=32, width=48).shape == (1, 3, 32, 48)                 # Lanczos resize
    lat = torch.randn(2, 32, 1, 8, 12)
    assert vp.preprocess(lat, height=64, width=96) is lat                                   # components.py:355-365
    vid = torch.stack([x[0], -x[0]], dim=1)[None]                                           # [1, 3, 2, 64, 96]
    frames = vp.postprocess_video(vid, output_type="pil")
    assert len(frames) == 1 and len(frames[0]) == 2 and frames[0][0].size == (96, 64)
    assert np.abs(np.array(frames[0][0]).astype(int) - np.array(img).astype(int)).max() <= 1
    assert vp.postprocess_video(vid, output_type="pt").shape == (1, 2, 3, 64, 96)
    assert vp.postprocess_video(vid, output_type="np").shape == (1, 2, 64, 96, 3)


def test_prepare_latents_encodes_rgb_through_attached_vae():
    """:1150-1167 - 4-D RGB reference frames go through vae.encode one clip at a time, latent_dist.sample(generator), then the
    scaling factor (inverted when the VAE config says so, :1186-1191), zero frame padding and randn_tensor latents."""
    cfg, m = tiny_model()
    vae = FakeVAE()
    pipe = CogVideoXImageToVideoPipelineTraj(transformer=m, scheduler=schedulers.CogVideoXDDIMScheduler(**SCHED), vae=vae)
    img = torch.rand(2, 3, 64, 96) * 2 - 1
    gen = torch.Generator().manual_seed(7)
    lat, img_lat = pipe.prepare_latents(img, batch_size=2, num_channels_latents=16, num_frames=9, num_views=1, height=64, width=96,
                                        dtype=torch.float32, device=torch.device("cpu"), generator=gen)
    assert [c[:2] for c in vae.calls] == [("encode", (1, 3, 1, 64, 96))] * 2
    assert lat.shape == (2, 3, 16, 8, 12) and img_lat.shape == (2, 3, 16, 8, 12)
    assert torch.all(img_lat[:, 1:] == 0)
    gen2 = torch.Generator().manual_seed(7)
    want = torch.cat([vae.encode(img[i][None, :, None]).latent_dist.sample(gen2) for i in range(2)])   # [2,16,1,8,12]
    assert torch.allclose(img_lat[:, 0], want[:, :, 0] / 1.15258426)                                    # invert_scale_latents=True
    assert torch.equal(lat, torch.randn(2, 3, 16, 8, 12, generator=gen2))                               # same generator stream
    bare = CogVideoXImageToVideoPipelineTraj(transformer=m, scheduler=schedulers.CogVideoXDDIMScheduler(**SCHED))
    with pytest.raises(NotImplementedError, match="vae"):
        bare.prepare_latents(img, 2, 16, 9, 1, 64, 96, torch.float32, torch.device("cpu"), gen)


@pytest.mark.gpu
def test_inference_script_body_runs_against_orv_amd(tmp_path):
    """The body of generate_video() (inference_control_to_video.py:71-146) with only the imports changed: pipeline from a
    directory, DPM scheduler swap, .to(device, dtype), vae slicing/tiling, invert_scale_latents override, then
    pipe(image=<PIL>, prompt=<str>, negative_prompt=<str>, controls_or_guidances={'actions', 'depths': None}, ...,
    generator=<CPU>, output_type='pil').frames[0] -> list of PIL frames."""
    dev = "cuda"
    dtype = torch.bfloat16
    cfg, m = tiny_model()
    CogVideoXImageToVideoPipelineTraj(transformer=m, scheduler=schedulers.CogVideoXDDIMScheduler(**SCHED)).save_pretrained(
        str(tmp_path / "model"))
    transformer = CogVideoXTransformer3DModelTraj.from_pretrained(str(tmp_path / "model"), subfolder='transformer', torch_dtype=dtype)
    vae, tok, t5 = FakeVAE(), FakeTokenizer(), FakeT5(cfg["text_embed_dim"])
    pipe = CogVideoXImageToVideoPipelineTraj.from_pretrained(str(tmp_path / "model"), transformer=transformer, vae=vae,
                                                             text_encoder=t5, tokenizer=tok, torch_dtype=dtype)
    for param in pipe.transformer.parameters():
        param.requires_grad = False
    pipe.scheduler = schedulers.CogVideoXDPMScheduler.from_config(pipe.scheduler.config, timestep_spacing='trailing')
    pipe.to(dev, dtype=dtype)
    pipe.vae.enable_slicing()
    pipe.vae.enable_tiling()
    pipe.transformer.eval()
    pipe.text_encoder.eval()
    pipe.vae.eval()
    pipe.transformer.gradient_checkpointing = False
    pipe.vae.config = FrozenConfig({**pipe.vae.config, "invert_scale_latents": False})
    rng = np.random.default_rng(1)
    image = PIL.Image.fromarray(rng.integers(0, 255, (64, 96, 3), dtype=np.uint8))
    width, height = image.size
    actions = torch.randn(8, 7)
    pipeline_args = {
        'image': image, 'prompt': 'move the arm left',
        'negative_prompt': 'The video is not of a high quality, it has a low resolution.',
        'controls_or_guidances': {'actions': actions.unsqueeze(0).to(dev, dtype), 'depths': None},
        'num_frames': actions.size(0) + 1, 'height': height, 'width': width, 'guidance_scale': 1.0, 'use_dynamic_cfg': False,
        'num_inference_steps': 3, 'max_sequence_length': cfg["max_text_seq_length"],
    }
    pipe.transformer.action_embed.forced_mask = torch.zeros(1, dtype=torch.bool)
    generator = torch.Generator().manual_seed(42)
    with torch.no_grad():
        video = pipe(**pipeline_args, generator=generator, output_type='pil').frames[0]
    assert len(video) == 9 and all(isinstance(f, PIL.Image.Image) and f.size == (96, 64) for f in video)
    kinds = [c[0] for c in vae.calls]
    assert kinds.count("encode") == 1 and kinds.count("decode") == 1
    dec = [c for c in vae.calls if c[0] == "decode"][0]
    assert dec[1] == (1, 16, 3, 8, 12)                                   # [B, C, f, h, w] handed to vae.decode
    assert [c[0] for c in tok.calls] == [['move the arm left']]          # guidance 1.0: negative prompt not encoded
    # same seed -> same frames (CPU generator drives the image sample, the initial latents and the DPM noise)
    generator = torch.Generator().manual_seed(42)
    with torch.no_grad():
        again = pipe(**pipeline_args, generator=generator, output_type='pil').frames[0]
    assert all(np.array_equal(np.array(a), np.array(b)) for a, b in zip(video, again))


def test_install_serves_the_reference_import_blocks_unchanged():
    """``orv_amd.install()`` (orv_amd/dropin.py): the import statements of the reference's entry points
    (inference_control_to_video.py:7-17, evaluation_control_to_video.py:17-23 - committed as a list of statements in
    tests/golden/reference_import_blocks.json) are exec'd AS WRITTEN in a fresh interpreter after ``install()`` and must resolve the hot-path names to
    this package: the north star's "drops into scripts/inference_control_to_video.sh unchanged" made literal."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, sys
import orv_amd
names = orv_amd.install()
assert orv_amd.install() == names                      # idempotent
blocks = json.load(open("tests/golden/reference_import_blocks.json"))
for key, lines in blocks.items():
    if key.startswith("_"):
        continue
    ns = {}
    for line in lines:
        exec(line, ns)
    import orv_amd.cogvideox_control as cc, orv_amd.schedulers as sch, orv_amd.utils as ut, orv_amd.data as data
    assert ns["CogVideoXTransformer3DModelTraj"] is cc.CogVideoXTransformer3DModelTraj, key
    assert ns["CogVideoXImageToVideoPipelineTraj"] is cc.CogVideoXImageToVideoPipelineTraj, key
    assert ns["CogVideoXDPMScheduler"] is sch.CogVideoXDPMScheduler, key
    assert callable(ns["export_to_video"]) and hasattr(ns["CONSOLE"], "log")
    cfg = ns["FrozenDict"](a=1, invert_scale_latents=False)
    assert cfg["a"] == 1 and cfg.invert_scale_latents is False
from orv.utils import prepare_rotary_positional_embeddings
from orv.models.components import ActionEmbed
from diffusers.schedulers.scheduling_ddim_cogvideox import CogVideoXDDIMScheduler
import orv_amd.components
assert prepare_rotary_positional_embeddings is ut.prepare_rotary_positional_embeddings
assert ActionEmbed is orv_amd.components.ActionEmbed and CogVideoXDDIMScheduler is sch.CogVideoXDDIMScheduler
import orv.dataset.dataset as od
if getattr(od, "__orv_amd_standin__", False):          # no reference on the path: this package's collate / sampler, raw-video classes refuse
    assert od.CollateFunctionControl is data.CollateFunctionControl and od.BucketSampler is data.BucketSampler
    try:
        od.DemoRobotDataset(data_root="x")
        raise SystemExit("DemoRobotDataset stand-in must refuse construction")
    except NotImplementedError:
        pass
print("INSTALL-OK", len(names))
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "INSTALL-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_uninstall_restores_what_install_displaced():
    """with a real ``diffusers.schedulers.scheduling_*_cogvideox`` / ``orv.utils`` in the environment, ``uninstall()`` must put
    the originals back (modules, parent attributes and the RoPE helper patched in place) - fake "real" modules stand in for them here."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, types
def mod(name, pkg=False, **attrs):
    m = types.ModuleType(name)
    if pkg: m.__path__ = []
    for k, v in attrs.items(): setattr(m, k, v)
    sys.modules[name] = m
    if "." in name:
        parent, _, leaf = name.rpartition(".")
        setattr(sys.modules[parent], leaf, m)
    return m
mod("diffusers", pkg=True); mod("diffusers.schedulers", pkg=True)
real_dpm = mod("diffusers.schedulers.scheduling_dpm_cogvideox", CogVideoXDPMScheduler="REAL-DPM")
real_ddim = mod("diffusers.schedulers.scheduling_ddim_cogvideox", CogVideoXDDIMScheduler="REAL-DDIM")
mod("orv", pkg=True)
real_rope = lambda *a, **k: "REAL-ROPE"
real_utils = mod("orv.utils", CONSOLE="REAL-CONSOLE", prepare_rotary_positional_embeddings=real_rope)
import orv_amd, orv_amd.dropin as d, orv_amd.utils as ut
orv_amd.install(); orv_amd.install()
assert sys.modules["diffusers.schedulers.scheduling_dpm_cogvideox"] is not real_dpm
assert sys.modules["orv.utils"] is real_utils and real_utils.prepare_rotary_positional_embeddings is ut.prepare_rotary_positional_embeddings
assert real_utils.CONSOLE == "REAL-CONSOLE"
d.uninstall()
assert sys.modules["diffusers.schedulers.scheduling_dpm_cogvideox"] is real_dpm
assert sys.modules["diffusers.schedulers"].scheduling_dpm_cogvideox is real_dpm
assert sys.modules["diffusers.schedulers.scheduling_ddim_cogvideox"] is real_ddim
assert sys.modules["orv.utils"] is real_utils and real_utils.prepare_rotary_positional_embeddings is real_rope
assert "orv.models.cogvideox_control" not in sys.modules and not hasattr(sys.modules["orv"], "models")
print("UNINSTALL-OK")
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "UNINSTALL-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_graph_cache_key_sees_structural_edits():
    """``GraphedTransformer._weights_version`` must move for edits that bypass ``register_parameter`` - a pre-built submodule
    attached (``blk.ff = other``), blocks truncated, ``del m.weight``, a write into ``_parameters`` - or a stale graph is replayed."""
    import torch
    from torch import nn
    from orv_amd.cogvideox_control import CogVideoXTransformer3DModelTraj, GraphedTransformer, FeedForward
    cfg = dict(num_attention_heads=2, attention_head_dim=64, in_channels=8, out_channels=4, num_layers=3, text_embed_dim=32,
               time_embed_dim=32, sample_width=12, sample_height=8, sample_frames=9, max_text_seq_length=8)
    m = CogVideoXTransformer3DModelTraj(**cfg)
    g = GraphedTransformer(m)
    v0 = g._weights_version()
    assert g._weights_version() == v0
    m.transformer_blocks[0].ff = FeedForward(m.inner_dim)                        # pre-built submodule: Module.__setattr__ -> register_module hook
    v1 = g._weights_version()
    assert v1 != v0
    del m.transformer_blocks[2:]                                                 # ModuleList.__delitem__ rewrites _modules directly
    v2 = g._weights_version()
    assert v2 != v1 and v2[0] < v1[0]
    del m.transformer_blocks[0].ff.net[2].bias                                   # Module.__delattr__
    v3 = g._weights_version()
    assert v3 != v2 and v3[0] == v2[0] - 1
    m.transformer_blocks[0].ff.net[2]._parameters["bias"] = nn.Parameter(torch.zeros(m.inner_dim))     # behind every hook
    v4 = g._weights_version()
    assert v4 != v3 and v4[0] == v2[0]
    assert g._weights_version() == v4
