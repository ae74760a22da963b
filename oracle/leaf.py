"""CPU restatement of the *diffusers* leaf classes the ORV hot path is built on.

TEST INFRASTRUCTURE ONLY.  Nothing under ``orv_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do (as the checker, never as
the thing measured or shipped).

PARITY UNPINNED for everything in this file: the arithmetic restated here lives in the
third-party package ``diffusers`` (reference requirement ``diffusers>=0.31.2``,
/root/reference/requirements.txt:22; recorded version ``0.32.0.dev0``,
/root/reference/config/transformer/base_1.4b_480_320.json:3) which is neither vendored in
/root/reference nor installable here (no network).  The formulas follow the published
diffusers 0.32 sources as summarised in SURVEY.md Appendix C and are anchored on the reference's
call sites (cited per class).  The classes keep diffusers' attribute names so that
(a) ``state_dict()`` keys equal real checkpoints' keys and (b) ``oracle/ref_harness.py`` can
run ORV's own ``forward`` methods (which subclass these) to pin the ORV-authored logic.

All math is eager PyTorch on CPU in whatever dtype the module is in (fp32 for parity tests).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------------------
# embeddings  (call sites: cogvideox_control.py:531-547, 666-674, 763-769, 788)
# --------------------------------------------------------------------------------------
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0,
                           scale=1.0, max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos = num_channels, flip_sin_to_cos
        self.downscale_freq_shift, self.scale = downscale_freq_shift, scale

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos,
                                      self.downscale_freq_shift, self.scale)


def _act(name):
    return {"silu": nn.SiLU, "swish": nn.SiLU, "gelu": nn.GELU, "relu": nn.ReLU, "mish": nn.Mish}[name]()


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = _act(act_fn)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


def _sincos_1d(embed_dim, pos):
    omega = torch.arange(embed_dim // 2, dtype=torch.float64, device=pos.device) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = torch.outer(pos.reshape(-1).to(torch.float64), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def get_3d_sincos_pos_embed(embed_dim, spatial_size, temporal_size, spatial_interpolation_scale=1.0,
                            temporal_interpolation_scale=1.0, device=None, output_type="pt"):
    """[T, gh*gw, D]; temporal quarter first, then (w-coord | h-coord) spatial halves."""
    if embed_dim % 4 != 0:
        raise ValueError("`embed_dim` must be divisible by 4")
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)
    gw, gh = spatial_size
    d_sp, d_t = 3 * embed_dim // 4, embed_dim // 4
    grid_h = torch.arange(gh, dtype=torch.float32, device=device) / spatial_interpolation_scale
    grid_w = torch.arange(gw, dtype=torch.float32, device=device) / spatial_interpolation_scale
    gx, gy = torch.meshgrid(grid_w, grid_h, indexing="xy")            # each [gh, gw]
    sp = torch.cat([_sincos_1d(d_sp // 2, gx), _sincos_1d(d_sp // 2, gy)], dim=1)   # [gh*gw, d_sp]
    grid_t = torch.arange(temporal_size, dtype=torch.float32, device=device) / temporal_interpolation_scale
    tp = _sincos_1d(d_t, grid_t)                                         # [T, d_t]
    sp = sp[None].repeat_interleave(temporal_size, dim=0)
    tp = tp[:, None].repeat_interleave(gh * gw, dim=1)
    return torch.cat([tp, sp], dim=-1)


class CogVideoXPatchEmbed(nn.Module):
    def __init__(self, patch_size=2, patch_size_t=None, in_channels=16, embed_dim=1920, text_embed_dim=4096,
                 bias=True, sample_width=90, sample_height=60, sample_frames=49, temporal_compression_ratio=4,
                 max_text_seq_length=226, spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0,
                 use_positional_embeddings=True, use_learned_positional_embeddings=True):
        super().__init__()
        self.patch_size, self.patch_size_t, self.embed_dim = patch_size, patch_size_t, embed_dim
        self.sample_height, self.sample_width, self.sample_frames = sample_height, sample_width, sample_frames
        self.temporal_compression_ratio, self.max_text_seq_length = temporal_compression_ratio, max_text_seq_length
        self.spatial_interpolation_scale = spatial_interpolation_scale
        self.temporal_interpolation_scale = temporal_interpolation_scale
        self.use_positional_embeddings = use_positional_embeddings
        self.use_learned_positional_embeddings = use_learned_positional_embeddings
        if patch_size_t is None:
            self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(patch_size, patch_size), stride=patch_size,
                                  bias=bias)
        else:
            self.proj = nn.Linear(in_channels * patch_size * patch_size * patch_size_t, embed_dim)
        self.text_proj = nn.Linear(text_embed_dim, embed_dim)
        if use_positional_embeddings or use_learned_positional_embeddings:
            pos = self._get_positional_embeddings(sample_height, sample_width, sample_frames)
            self.register_buffer("pos_embedding", pos, persistent=use_learned_positional_embeddings)

    def _get_positional_embeddings(self, sample_height, sample_width, sample_frames, device=None):
        ph, pw = sample_height // self.patch_size, sample_width // self.patch_size
        pt = (sample_frames - 1) // self.temporal_compression_ratio + 1
        pos = get_3d_sincos_pos_embed(self.embed_dim, (pw, ph), pt, self.spatial_interpolation_scale,
                                      self.temporal_interpolation_scale, device=device).flatten(0, 1)
        joint = torch.zeros(1, self.max_text_seq_length + ph * pw * pt, self.embed_dim, dtype=torch.float32,
                            device=device)
        joint[:, self.max_text_seq_length:] = pos.to(torch.float32)
        return joint

    def forward(self, text_embeds, image_embeds):
        text_embeds = self.text_proj(text_embeds)
        b, t, c, h, w = image_embeds.shape
        if self.patch_size_t is None:
            x = self.proj(image_embeds.reshape(-1, c, h, w))
            x = x.view(b, t, *x.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)
        else:
            p, pt = self.patch_size, self.patch_size_t
            x = image_embeds.permute(0, 1, 3, 4, 2).reshape(b, t // pt, pt, h // p, p, w // p, p, c)
            x = x.permute(0, 1, 3, 5, 7, 2, 4, 6).flatten(4, 7).flatten(1, 3)
            x = self.proj(x)
        embeds = torch.cat([text_embeds, x], dim=1).contiguous()
        if self.use_positional_embeddings or self.use_learned_positional_embeddings:
            if self.use_learned_positional_embeddings and (self.sample_width != w or self.sample_height != h):
                raise ValueError("learned positional embeddings need the trained sample size")
            pre = (t - 1) * self.temporal_compression_ratio + 1
            if self.sample_height != h or self.sample_width != w or self.sample_frames != pre:
                pos = self._get_positional_embeddings(h, w, pre, device=embeds.device)
            else:
                pos = self.pos_embedding
            embeds = embeds + pos.to(dtype=embeds.dtype)
        return embeds


def apply_rotary_emb(x, freqs_cis, use_real=True, use_real_unbind_dim=-1):
    """x [B,H,S,D], freqs (cos[S,D], sin[S,D]); pairs are (2i, 2i+1)."""
    cos, sin = freqs_cis
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + rot.float() * sin).to(x.dtype)


def _rope_1d(dim, pos, theta=10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32, device=pos.device)[: dim // 2] / dim))
    ang = torch.outer(pos, freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size, theta=10000, use_real=True,
                            grid_type="linspace", max_size=None, device=None):
    gh, gw = grid_size
    if grid_type == "linspace":
        start, stop = crops_coords
        pos_h = torch.linspace(start[0], stop[0] * (gh - 1) / gh, gh, device=device, dtype=torch.float32)
        pos_w = torch.linspace(start[1], stop[1] * (gw - 1) / gw, gw, device=device, dtype=torch.float32)
        pos_t = torch.linspace(0, temporal_size * (temporal_size - 1) / temporal_size, temporal_size,
                               device=device, dtype=torch.float32)
    elif grid_type == "slice":
        mh, mw = max_size
        pos_h = torch.arange(mh, device=device, dtype=torch.float32)
        pos_w = torch.arange(mw, device=device, dtype=torch.float32)
        pos_t = torch.arange(temporal_size, device=device, dtype=torch.float32)
    else:
        raise ValueError("Invalid value passed for `grid_type`.")
    dt, dh, dw = embed_dim // 4, embed_dim // 8 * 3, embed_dim // 8 * 3
    (tc, ts), (hc, hs), (wc, ws) = _rope_1d(dt, pos_t, theta), _rope_1d(dh, pos_h, theta), _rope_1d(dw, pos_w, theta)
    if grid_type == "slice":
        tc, ts, hc, hs, wc, ws = tc[:temporal_size], ts[:temporal_size], hc[:gh], hs[:gh], wc[:gw], ws[:gw]

    def combine(ft, fh, fw):
        ft = ft[:, None, None, :].expand(-1, gh, gw, -1)
        fh = fh[None, :, None, :].expand(temporal_size, -1, gw, -1)
        fw = fw[None, None, :, :].expand(temporal_size, gh, -1, -1)
        return torch.cat([ft, fh, fw], dim=-1).reshape(temporal_size * gh * gw, -1)

    return combine(tc, hc, wc), combine(ts, hs, ws)


# --------------------------------------------------------------------------------------
# attention / FFN / norms   (call sites: cogvideox_control.py:41-58, 153, 200, 290-311, 351-391)
# --------------------------------------------------------------------------------------
class Attention(nn.Module):
    """Subset of diffusers ``Attention`` as configured by ORV (self-attention, qk LayerNorm)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 qk_norm=None, eps=1e-5, out_bias=True, processor=None, **_):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.heads, self.is_cross_attention = heads, cross_attention_dim is not None
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, self.inner_dim, bias=bias)
        if qk_norm is None:
            self.norm_q = self.norm_k = None
        elif qk_norm == "layer_norm":
            self.norm_q = nn.LayerNorm(dim_head, eps=eps, elementwise_affine=True)
            self.norm_k = nn.LayerNorm(dim_head, eps=eps, elementwise_affine=True)
        else:
            raise ValueError(qk_norm)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor

    def prepare_attention_mask(self, *a, **k):
        raise NotImplementedError("attention masks are never used on the ORV path")

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class CogVideoXAttnProcessor2_0:
    """Base processor; ORV overrides ``__call__`` (cogvideox_control.py:200-270)."""

    def __init__(self):
        pass


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False,
                 inner_dim=None, bias=True):
        super().__init__()
        inner_dim = inner_dim or int(dim * mult)
        dim_out = dim_out or dim
        if activation_fn == "gelu-approximate":
            act = GELU(dim, inner_dim, approximate="tanh", bias=bias)
        elif activation_fn == "gelu":
            act = GELU(dim, inner_dim, bias=bias)
        else:
            raise ValueError(f"activation {activation_fn} is not used by any ORV config")
        net = [act, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out, bias=bias)]
        if final_dropout:
            net.append(nn.Dropout(dropout))
        self.net = nn.ModuleList(net)

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class CogVideoXLayerNormZero(nn.Module):
    def __init__(self, conditioning_dim, embedding_dim, elementwise_affine=True, eps=1e-5, bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 6 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps=eps, elementwise_affine=elementwise_affine)


class AdaLayerNorm(nn.Module):
    def __init__(self, embedding_dim, num_embeddings=None, output_dim=None, norm_elementwise_affine=False,
                 norm_eps=1e-5, chunk_dim=0):
        super().__init__()
        self.chunk_dim = chunk_dim
        output_dim = output_dim or embedding_dim * 2
        self.emb = nn.Embedding(num_embeddings, embedding_dim) if num_embeddings is not None else None
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, output_dim)
        self.norm = nn.LayerNorm(output_dim // 2, norm_eps, norm_elementwise_affine)


class CogVideoXBlock(nn.Module):
    """Base block: only ``ff`` survives ORV's override (norm1/norm2/attn1 are rebuilt, :378-391)."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, time_embed_dim, dropout=0.0,
                 activation_fn="gelu-approximate", attention_bias=False, qk_norm=True, norm_elementwise_affine=True,
                 norm_eps=1e-5, final_dropout=True, ff_inner_dim=None, ff_bias=True, attention_out_bias=True):
        super().__init__()
        self.norm1 = CogVideoXLayerNormZero(time_embed_dim, dim, norm_elementwise_affine, norm_eps, bias=True)
        self.attn1 = Attention(query_dim=dim, dim_head=attention_head_dim, heads=num_attention_heads,
                               qk_norm="layer_norm" if qk_norm else None, eps=1e-6, bias=attention_bias,
                               out_bias=attention_out_bias, processor=CogVideoXAttnProcessor2_0())
        self.norm2 = CogVideoXLayerNormZero(time_embed_dim, dim, norm_elementwise_affine, norm_eps, bias=True)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout,
                              inner_dim=ff_inner_dim, bias=ff_bias)


class CogVideoXTransformer3DModelBase(nn.Module):
    """Constructor of diffusers ``CogVideoXTransformer3DModel`` (attributes ORV's forward reads)."""

    def __init__(self, num_attention_heads=30, attention_head_dim=64, in_channels=16, out_channels=16,
                 flip_sin_to_cos=True, freq_shift=0, time_embed_dim=512, ofs_embed_dim=None, text_embed_dim=4096,
                 num_layers=30, dropout=0.0, attention_bias=True, sample_width=90, sample_height=60,
                 sample_frames=49, patch_size=2, patch_size_t=None, temporal_compression_ratio=4,
                 max_text_seq_length=226, activation_fn="gelu-approximate", timestep_activation_fn="silu",
                 norm_elementwise_affine=True, norm_eps=1e-5, spatial_interpolation_scale=1.875,
                 temporal_interpolation_scale=1.0, use_rotary_positional_embeddings=False,
                 use_learned_positional_embeddings=False, patch_bias=True, **_):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.patch_embed = CogVideoXPatchEmbed(
            patch_size=patch_size, patch_size_t=patch_size_t, in_channels=in_channels, embed_dim=inner,
            text_embed_dim=text_embed_dim, bias=patch_bias, sample_width=sample_width, sample_height=sample_height,
            sample_frames=sample_frames, temporal_compression_ratio=temporal_compression_ratio,
            max_text_seq_length=max_text_seq_length, spatial_interpolation_scale=spatial_interpolation_scale,
            temporal_interpolation_scale=temporal_interpolation_scale,
            use_positional_embeddings=not use_rotary_positional_embeddings,
            use_learned_positional_embeddings=use_learned_positional_embeddings)
        self.embedding_dropout = nn.Dropout(dropout)
        self.time_proj = Timesteps(inner, flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(inner, time_embed_dim, timestep_activation_fn)
        self.ofs_proj = self.ofs_embedding = None
        if ofs_embed_dim:
            self.ofs_proj = Timesteps(ofs_embed_dim, flip_sin_to_cos, freq_shift)
            self.ofs_embedding = TimestepEmbedding(ofs_embed_dim, ofs_embed_dim, timestep_activation_fn)
        self.transformer_blocks = nn.ModuleList([
            CogVideoXBlock(dim=inner, num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                           time_embed_dim=time_embed_dim, dropout=dropout, activation_fn=activation_fn,
                           attention_bias=attention_bias, norm_elementwise_affine=norm_elementwise_affine,
                           norm_eps=norm_eps) for _ in range(num_layers)])
        self.norm_final = nn.LayerNorm(inner, norm_eps, norm_elementwise_affine)
        self.norm_out = AdaLayerNorm(embedding_dim=time_embed_dim, output_dim=2 * inner,
                                     norm_elementwise_affine=norm_elementwise_affine, norm_eps=norm_eps, chunk_dim=1)
        out_dim = patch_size * patch_size * out_channels * (patch_size_t or 1)
        self.proj_out = nn.Linear(inner, out_dim)
        self.gradient_checkpointing = False


# --------------------------------------------------------------------------------------
# schedulers / sampling helpers (call sites: cogvideox_control.py:1152,1175,1219,1304,1446-1457;
# train_cogvideox_control_to_video_sft.py:341,842,1042,1067)
# --------------------------------------------------------------------------------------
def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    device = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(generator, list):
        shape = (1,) + tuple(shape[1:])
        return torch.cat([randn_tensor(shape, g, device, dtype) for g in generator], dim=0)
    rand_device = device
    if generator is not None and generator.device.type != device.type and generator.device.type == "cpu":
        rand_device = torch.device("cpu")
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)


class DiagonalGaussianDistribution:
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std, self.var = torch.exp(0.5 * self.logvar), torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator=None):
        eps = randn_tensor(self.mean.shape, generator=generator, device=self.parameters.device,
                           dtype=self.parameters.dtype)
        return self.mean + self.std * eps

    def mode(self):
        return self.mean


def cogvideox_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                             beta_schedule="scaled_linear", snr_shift_scale=3.0, rescale_betas_zero_snr=True):
    if beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
    elif beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float64)
    else:
        raise NotImplementedError(beta_schedule)
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    if rescale_betas_zero_snr:
        s = ac.sqrt()
        s0, sT = s[0].clone(), s[-1].clone()
        s = (s - sT) * (s0 / (s0 - sT))
        ac = s ** 2
    return ac


class _SchedulerConfig(dict):
    __getattr__ = dict.__getitem__


class _CogVideoXSchedulerBase:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, snr_shift_scale=3.0, **_):
        self.config = _SchedulerConfig(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
            steps_offset=steps_offset, prediction_type=prediction_type, timestep_spacing=timestep_spacing,
            rescale_betas_zero_snr=rescale_betas_zero_snr, snr_shift_scale=snr_shift_scale)
        self.alphas_cumprod = cogvideox_alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule,
                                                       snr_shift_scale, rescale_betas_zero_snr)
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    @classmethod
    def from_config(cls, config, **kw):
        cfg = dict(config)
        cfg.update(kw)
        return cls(**cfg)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            ts = (np.arange(0, num_inference_steps) * (n // num_inference_steps)).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif sp == "trailing":
            ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts).to(device)

    def _x0(self, model_output, sample, a_t):
        pt = self.config.prediction_type
        if pt == "v_prediction":
            return (a_t ** 0.5) * sample - ((1 - a_t) ** 0.5) * model_output
        if pt == "epsilon":
            return (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        if pt == "sample":
            return model_output
        raise ValueError(pt)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        a = ac[timesteps.to(original_samples.device)]
        sa, sb = (a ** 0.5).flatten(), ((1 - a) ** 0.5).flatten()
        while sa.ndim < original_samples.ndim:
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise

    def get_velocity(self, sample, noise, timesteps):
        ac = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)
        a = ac[timesteps.to(sample.device)]
        sa, sb = (a ** 0.5).flatten(), ((1 - a) ** 0.5).flatten()
        while sa.ndim < sample.ndim:
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * noise - sb * sample


class CogVideoXDDIMScheduler(_CogVideoXSchedulerBase):
    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = self._x0(model_output, sample, a_t)
        A = ((1 - a_p) / (1 - a_t)) ** 0.5
        Bc = a_p ** 0.5 - a_t ** 0.5 * A
        prev_sample = A * sample + Bc * x0
        return (prev_sample, x0) if not return_dict else _SchedulerConfig(prev_sample=prev_sample,
                                                                          pred_original_sample=x0)


class CogVideoXDPMScheduler(_CogVideoXSchedulerBase):
    @staticmethod
    def coefficients(a_t, a_p, a_b):
        """(m1, m2, m_noise, m3, m4) of DPM-Solver++(2M) SDE; float64 0-dim tensors (m3/m4 None on first step)."""
        lamb = ((a_t / (1 - a_t)) ** 0.5).log()
        lamb_next = ((a_p / (1 - a_p)) ** 0.5).log()
        h = lamb_next - lamb
        m1 = ((1 - a_p) / (1 - a_t)) ** 0.5 * (-h).exp()
        m2 = (-2 * h).expm1() * a_p ** 0.5
        mn = (1 - a_p) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
        if a_b is None:
            return m1, m2, mn, None, None
        r = (lamb - ((a_b / (1 - a_b)) ** 0.5).log()) / h
        return m1, m2, mn, 1 + 1 / (2 * r), 1 / (2 * r)

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, eta=0.0,
             use_clipped_model_output=False, generator=None, variance_noise=None, return_dict=False):
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        a_b = self.alphas_cumprod[int(timestep_back)] if timestep_back is not None else None
        x0 = self._x0(model_output, sample, a_t)
        m1, m2, mn, m3, m4 = self.coefficients(a_t, a_p, a_b)
        noise = randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
        prev_sample = m1 * sample - m2 * x0 + mn * noise
        if old_pred_original_sample is None or prev < 0:
            return prev_sample, x0
        d = m3 * x0 - m4 * old_pred_original_sample
        noise = randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
        return m1 * sample - m2 * d + mn * noise, x0


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kw):
    if timesteps is not None:
        raise ValueError("custom timesteps are not supported by the CogVideoX schedulers")
    scheduler.set_timesteps(num_inference_steps, device=device, **kw)
    return scheduler.timesteps, num_inference_steps
