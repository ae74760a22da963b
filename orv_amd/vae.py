"""MI355X-native ``AutoencoderKLCogVideoX``: the VAE the reference decodes with at
/root/reference/orv/models/cogvideox_control.py:1476-1479 (``decode_latents`` -> ``vae.decode(latents).sample``) and encodes the
reference frame with at :1161-1166 (``vae.encode(image).latent_dist``).  Interface, module tree and ``state_dict`` keys are
diffusers' (``encoder.down_blocks.{i}.resnets.{j}.conv1.conv.weight``, ``decoder.up_blocks.{i}.resnets.{j}.norm1.conv_y.conv.*``,
``decoder.up_blocks.{i}.upsamplers.0.conv.*`` ...), so a ``THUDM/CogVideoX-2b`` ``vae/`` folder loads unchanged.

**Parity unpinned** (DESIGN.md): the arithmetic lives in diffusers, which is absent here; the CPU oracle (oracle/vae.py) is a
restatement of the published algorithm and the tests compare the HIP path with it on random weights, plus structural
properties (causality in time, frame counts, first-frame rules).

MI355X design (vae.hip): activations stay CHANNELS-LAST bf16 ``[B, T, H, W, C]`` from ``conv_in`` to ``conv_out``; every
convolution is ``orv_vae_im2col`` (gather with the causal / zero padding, stride and nearest-neighbour upsampling folded in)
+ ``orv_gemm_bf16`` (bias, and the resnet's residual add as GEMM epilogue 2); GroupNorm / SpatialNorm / SiLU are one
statistics pass + one fused apply pass; ``conv_y(zq)`` / ``conv_b(zq)`` are evaluated ONCE at latent resolution (a 1x1x1
convolution commutes with nearest-neighbour resizing) and looked up by index.  The patch matrix is built in slabs of at most
``_PATCH_BYTES`` so the decode of one 17 x 320 x 480 clip peaks at a few GB of the 288.  No tiling / slicing is needed at this
HBM size: ``enable_tiling`` / ``enable_slicing`` are accepted and ignored.  There is no CPU fallback.
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import torch
from torch import nn

from . import ops
from .cogvideox_control import FrozenConfig, _NoForward

BF16 = torch.bfloat16
_PATCH_BYTES = 2 << 30
_IMPLICIT_GEMM = os.environ.get("ORV_VAE_IMPLICIT_GEMM", "1") != "0"     # A/B switch: 0 = patch matrix + plain GEMM everywhere


class CogVideoXCausalConv3d(_NoForward):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, pad_mode="first"):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        self.conv = nn.Conv3d(in_channels, out_channels, k, stride=(stride, 1, 1), padding=(0, (k[1] - 1) // 2, (k[2] - 1) // 2))


class CogVideoXSpatialNorm3D(_NoForward):
    def __init__(self, f_channels, zq_channels, groups=32):
        super().__init__()
        self.norm_layer = nn.GroupNorm(num_channels=f_channels, num_groups=groups, eps=1e-6, affine=True)
        self.conv_y = CogVideoXCausalConv3d(zq_channels, f_channels, kernel_size=1)
        self.conv_b = CogVideoXCausalConv3d(zq_channels, f_channels, kernel_size=1)


class CogVideoXResnetBlock3D(_NoForward):
    def __init__(self, in_channels, out_channels=None, groups=32, eps=1e-6, spatial_norm_dim=None):
        super().__init__()
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels, self.eps = in_channels, out_channels, eps
        if spatial_norm_dim is None:
            self.norm1 = nn.GroupNorm(num_channels=in_channels, num_groups=groups, eps=eps)
            self.norm2 = nn.GroupNorm(num_channels=out_channels, num_groups=groups, eps=eps)
        else:
            self.norm1 = CogVideoXSpatialNorm3D(in_channels, spatial_norm_dim, groups)
            self.norm2 = CogVideoXSpatialNorm3D(out_channels, spatial_norm_dim, groups)
        self.conv1 = CogVideoXCausalConv3d(in_channels, out_channels, 3)
        self.conv2 = CogVideoXCausalConv3d(out_channels, out_channels, 3)
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv3d(in_channels, out_channels, kernel_size=1)


class CogVideoXUpsample3D(_NoForward):
    def __init__(self, in_channels, out_channels, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.compress_time = compress_time


class CogVideoXDownsample3D(_NoForward):
    def __init__(self, in_channels, out_channels, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=2, padding=0)
        self.compress_time = compress_time


class CogVideoXMidBlock3D(_NoForward):
    def __init__(self, in_channels, num_layers=2, groups=32, eps=1e-6, spatial_norm_dim=None):
        super().__init__()
        self.resnets = nn.ModuleList([CogVideoXResnetBlock3D(in_channels, in_channels, groups, eps, spatial_norm_dim)
                                      for _ in range(num_layers)])


class CogVideoXUpBlock3D(_NoForward):
    def __init__(self, in_channels, out_channels, num_layers, groups, eps, spatial_norm_dim, add_upsample, compress_time):
        super().__init__()
        self.resnets = nn.ModuleList([CogVideoXResnetBlock3D(in_channels if i == 0 else out_channels, out_channels, groups, eps,
                                                             spatial_norm_dim) for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([CogVideoXUpsample3D(out_channels, out_channels, compress_time)]) if add_upsample else None


class CogVideoXDownBlock3D(_NoForward):
    def __init__(self, in_channels, out_channels, num_layers, groups, eps, add_downsample, compress_time):
        super().__init__()
        self.resnets = nn.ModuleList([CogVideoXResnetBlock3D(in_channels if i == 0 else out_channels, out_channels, groups, eps)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([CogVideoXDownsample3D(out_channels, out_channels, compress_time)]) if add_downsample else None


class CogVideoXDecoder3D(_NoForward):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, norm_eps, norm_num_groups,
                 temporal_compression_ratio):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = CogVideoXCausalConv3d(in_channels, rev[0], 3)
        self.mid_block = CogVideoXMidBlock3D(rev[0], 2, norm_num_groups, norm_eps, in_channels)
        self.up_blocks = nn.ModuleList()
        out_c, levels = rev[0], int(math.log2(temporal_compression_ratio))
        for i in range(len(rev)):
            prev, out_c = out_c, rev[i]
            self.up_blocks.append(CogVideoXUpBlock3D(prev, out_c, layers_per_block + 1, norm_num_groups, norm_eps, in_channels,
                                                     add_upsample=i != len(rev) - 1, compress_time=i < levels))
        self.norm_out = CogVideoXSpatialNorm3D(rev[-1], in_channels, norm_num_groups)
        self.conv_act = nn.SiLU()
        self.conv_out = CogVideoXCausalConv3d(rev[-1], out_channels, 3)


class CogVideoXEncoder3D(_NoForward):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, norm_eps, norm_num_groups,
                 temporal_compression_ratio):
        super().__init__()
        levels = int(math.log2(temporal_compression_ratio))
        self.conv_in = CogVideoXCausalConv3d(in_channels, block_out_channels[0], 3)
        self.down_blocks = nn.ModuleList()
        out_c = block_out_channels[0]
        for i in range(len(block_out_channels)):
            prev, out_c = out_c, block_out_channels[i]
            self.down_blocks.append(CogVideoXDownBlock3D(prev, out_c, layers_per_block, norm_num_groups, norm_eps,
                                                         add_downsample=i != len(block_out_channels) - 1, compress_time=i < levels))
        self.mid_block = CogVideoXMidBlock3D(block_out_channels[-1], 2, norm_num_groups, norm_eps, None)
        self.norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = CogVideoXCausalConv3d(block_out_channels[-1], 2 * out_channels, 3)


class DiagonalGaussianDistribution:
    """diffusers' class as the pipeline uses it (``.sample(generator)``, ``.mode()``) over moments [B, 2C, F, h, w]."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        from .cogvideox_control import _randn
        eps = _randn(self.mean.shape, generator, self.parameters.device, self.parameters.dtype)
        return self.mean + self.std * eps

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


_VAE_DEFAULTS = dict(in_channels=3, out_channels=3, down_block_types=("CogVideoXDownBlock3D",) * 4,
                     up_block_types=("CogVideoXUpBlock3D",) * 4, block_out_channels=(128, 256, 256, 512), latent_channels=16,
                     layers_per_block=3, act_fn="silu", norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4,
                     sample_height=480, sample_width=720, scaling_factor=1.15258426, shift_factor=None, latents_mean=None,
                     latents_std=None, force_upcast=True, use_quant_conv=False, use_post_quant_conv=False,
                     invert_scale_latents=False)


class AutoencoderKLCogVideoX(nn.Module):
    config_name = "config.json"

    def __init__(self, **kwargs):
        super().__init__()
        cfg = {**_VAE_DEFAULTS, **{k: v for k, v in kwargs.items() if not k.startswith("_")}}
        if cfg["use_quant_conv"] or cfg["use_post_quant_conv"]:
            raise NotImplementedError("quant_conv / post_quant_conv are off in every CogVideoX VAE config")
        if cfg["act_fn"] != "silu":
            raise NotImplementedError("the fused norm kernel implements SiLU")
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = FrozenConfig(cfg)
        c = self.config
        a = (c.block_out_channels, c.layers_per_block, c.norm_eps, c.norm_num_groups, c.temporal_compression_ratio)
        self.encoder = CogVideoXEncoder3D(c.in_channels, c.latent_channels, *a)
        self.decoder = CogVideoXDecoder3D(c.latent_channels, c.out_channels, *a)
        self._packed = {}
        self.use_slicing = self.use_tiling = False
        self.tile_sample_min_height, self.tile_sample_min_width = c.sample_height // 2, c.sample_width // 2
        down = 2 ** (len(c.block_out_channels) - 1)
        self.tile_latent_min_height = int(self.tile_sample_min_height / down)
        self.tile_latent_min_width = int(self.tile_sample_min_width / down)
        self.tile_overlap_factor_height, self.tile_overlap_factor_width = 1 / 6, 1 / 5

    # ---- diffusers surface the reference's entry points touch ----
    # Both entry points call ``pipe.vae.enable_slicing(); pipe.vae.enable_tiling()`` (/root/reference/orv/pipeline/
    # inference_control_to_video.py:98-99, evaluation_control_to_video.py:274-275).  Slicing (one batch element at a time) changes
    # no arithmetic - every GroupNorm statistic is per batch element - and is accepted as a flag.  Tiling DOES: a latent larger
    # than one latent tile (30 x 45 for the 480 x 720 VAE config) is decoded tile by tile, each tile with its own GroupNorm
    # statistics and conv caches, and the seams are blended (``_tiled``).  Off by default, as in diffusers.
    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def enable_tiling(self, tile_sample_min_height: Optional[int] = None, tile_sample_min_width: Optional[int] = None,
                      tile_overlap_factor_height: Optional[float] = None, tile_overlap_factor_width: Optional[float] = None):
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        down = 2 ** (len(self.config.block_out_channels) - 1)
        self.tile_latent_min_height = int(self.tile_sample_min_height / down)
        self.tile_latent_min_width = int(self.tile_sample_min_width / down)
        self.tile_overlap_factor_height = tile_overlap_factor_height or self.tile_overlap_factor_height
        self.tile_overlap_factor_width = tile_overlap_factor_width or self.tile_overlap_factor_width

    def disable_tiling(self):
        self.use_tiling = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_config(cls, config, **kwargs):
        return cls(**{**{k: v for k, v in dict(config).items() if not k.startswith("_")}, **kwargs})

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, torch_dtype=None, **kwargs):
        from .checkpoint import load_state_dict_dir
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name), "r", encoding="utf-8") as f:
            cfg = json.load(f)
        model = cls.from_config(cfg, **kwargs)
        missing, unexpected = model.load_state_dict(load_state_dict_dir(d), strict=False)
        if missing or unexpected:
            raise RuntimeError(f"AutoencoderKLCogVideoX checkpoint mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        return model.to(torch_dtype) if torch_dtype is not None else model

    def save_pretrained(self, save_directory, **kw):
        from .checkpoint import save_state_dict_dir
        os.makedirs(save_directory, exist_ok=True)
        save_state_dict_dir(self.state_dict(), save_directory, max_shard_size=kw.get("max_shard_size", "5GB"))
        with open(os.path.join(save_directory, self.config_name), "w", encoding="utf-8") as f:
            json.dump({**{k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()},
                       "_class_name": "AutoencoderKLCogVideoX", "_diffusers_version": "0.32.0.dev0"}, f, indent=2)

    def _prepack(self, fn):
        """Touch the packed-weight cache of every convolution ``fn`` (decode / encode of one frame batch) uses, on the current
        stream: tiles that later run on side streams must find the packed tensors complete."""
        net = self.decoder if fn == self._decode_batch else self.encoder
        for m in net.modules():
            if isinstance(m, (nn.Conv3d, nn.Conv2d)) and m.weight.ndim >= 4:
                self._w(m)
        if net is self.encoder:
            self._w(self._padded_conv_in())

    # ---- weights in GEMM form (cached per weight version) ----
    def _w(self, conv: nn.Module):
        """[Npad, Kpad] bf16 with column = tap * Cin + ci (the patch matrix's column order) and the padded bias."""
        w = conv.weight
        key = id(conv)
        ver = (w.data_ptr(), w._version)
        hit = self._packed.get(key)
        if hit is None or hit[0] != ver:
            co, ci = w.shape[:2]
            taps = w[0, 0].numel()
            wp = w.detach().reshape(co, ci, taps).permute(0, 2, 1).reshape(co, taps * ci)
            kpad = (taps * ci + 63) // 64 * 64
            npad = (co + 63) // 64 * 64
            full = torch.zeros(npad, kpad, dtype=BF16, device=w.device)
            full[:co, :taps * ci] = wp
            b = None
            if conv.bias is not None:
                b = torch.zeros(npad, dtype=BF16, device=w.device)
                b[:co] = conv.bias.detach()
            hit = (ver, full, b, kpad, npad)
            self._packed[key] = hit
        return hit[1:]

    # ---- building blocks on channels-last tensors [B, T, H, W, C] ----
    def _conv(self, x, conv, k3, out_thw, stride=1, pad_lo=1, ups_s=0, ups_t=0, residual=None, cc=None, lead=0):
        """Causal / per-frame convolution as im2col slabs + GEMM.  ``k3`` = (kt, kh, kw); ``residual`` [.., Cout] is added in the
        GEMM epilogue.  ``cc`` = (prev, new) conv_cache dicts of the frame batching: with an entry for this convolution the last
        kt-1 input frames of the previous batch stand in front of ``x`` instead of copies of its first frame.  ``lead`` = kt-1
        says ``x`` already has that many free frames in front (``_norm(..., lead=)``): the context is written there, the
        activation itself is not copied."""
        t_shift = 0
        if lead:
            prev, new = cc
            x[:, :lead] = prev[id(conv)] if id(conv) in prev else x[:, lead:lead + 1]
            t_shift = lead
            new[id(conv)] = x[:, -lead:].clone()
        elif k3[0] > 1 and cc is not None:
            prev, new = cc
            if id(conv) in prev:
                x = torch.cat([prev[id(conv)], x], dim=1)
                t_shift = k3[0] - 1
                new[id(conv)] = x[:, -(k3[0] - 1):].clone()
            else:
                first = x[:, :1].expand(-1, k3[0] - 1, -1, -1, -1)
                new[id(conv)] = torch.cat([first, x], dim=1)[:, -(k3[0] - 1):].clone()
        B, Ts, Hs, Ws, C = x.shape
        T, H, W = out_thw
        wp, bias, kpad, npad = self._w(conv)
        co = conv.weight.shape[0]
        M = B * T * H * W
        direct = co == npad
        out = torch.empty(M, npad, dtype=BF16, device=x.device)
        if C % 64 == 0 and _IMPLICIT_GEMM:
            # implicit GEMM: the LDS-DMA gathers the tap-shifted channel lines itself, no patch matrix (kpad == taps * C here)
            res2d = residual.reshape(M, co) if (residual is not None and direct) else None
            ops.conv_gemm(x, wp, bias, out, B, Ts, Hs, Ws, C, T, H, W, k3[0], k3[1], k3[2], stride, pad_lo, ups_s, ups_t, t_shift,
                          npad, R=res2d, ldr=co)
            if not direct:
                out = out[:, :co].contiguous()
                if residual is not None:
                    out = out + residual.reshape(M, co)
            return out.view(B, T, H, W, co)
        rows = max(256, min(M, (_PATCH_BYTES // (2 * kpad)) // 256 * 256))
        patch = torch.empty(min(rows, M), kpad, dtype=BF16, device=x.device)
        res2d = residual.reshape(M, co) if (residual is not None and direct) else None
        for m0 in range(0, M, rows):
            mc = min(rows, M - m0)
            ops.vae_im2col(x, patch, B, Ts, Hs, Ws, C, T, H, W, k3[0], k3[1], k3[2], stride, pad_lo, ups_s, ups_t, t_shift, kpad,
                           m0, mc)
            if res2d is not None:
                ops.gemm(patch, wp, bias, out[m0:m0 + mc], mc, npad, kpad, epilogue=2, R=res2d[m0:m0 + mc], ldr=co)
            else:
                ops.gemm(patch, wp, bias, out[m0:m0 + mc], mc, npad, kpad)
        if not direct:
            out = out[:, :co].contiguous()
            if residual is not None:
                out = out + residual.reshape(M, co)
        return out.view(B, T, H, W, co)

    def _pointwise(self, x2d, conv):
        """1x1x1 convolution of [M, Cin] rows (conv_shortcut, conv_y / conv_b on the latent)."""
        wp, bias, kpad, npad = self._w(conv)
        M, ci = x2d.shape
        co = conv.weight.shape[0]
        if ci != kpad:
            x2d = torch.nn.functional.pad(x2d, (0, kpad - ci))
        out = torch.empty(M, npad, dtype=BF16, device=x2d.device)
        ops.gemm(x2d.contiguous(), wp, bias, out, M, npad, kpad)
        return out if co == npad else out[:, :co].contiguous()

    def _norm(self, x, norm, zq, silu=True, lead=0):
        """GroupNorm (+ SpatialNorm modulation by the latent ``zq`` [B, Tz, hz, wz, Cz]) (+ SiLU).  The result has ``lead`` extra
        frames in front, not written here (room for the next convolution's causal context)."""
        B, T, H, W, C = x.shape
        spatial = isinstance(norm, CogVideoXSpatialNorm3D)
        gn = norm.norm_layer if spatial else norm
        sums = ops.vae_groupnorm_stats(x, B, T * H * W, C, gn.num_groups)
        zy = zb = None
        Tz = hz = wz = 0
        if spatial:
            _, Tz, hz, wz, cz = zq.shape
            z2 = zq.reshape(-1, cz)
            zy, zb = self._pointwise(z2, norm.conv_y.conv), self._pointwise(z2, norm.conv_b.conv)
        out = torch.empty(B, lead + T, H, W, C, dtype=x.dtype, device=x.device)
        ops.vae_norm_apply(x, out, sums, gn.weight, gn.bias, zy, zb, B, T, H, W, C, gn.num_groups, Tz, hz, wz, gn.eps, silu, lead)
        return out

    def _resnet(self, x, blk, zq, cc):
        B, T, H, W, C = x.shape
        lead = 2 if cc is not None else 0
        h = self._norm(x, blk.norm1, zq, lead=lead)
        h = self._conv(h, blk.conv1.conv, (3, 3, 3), (T, H, W), cc=cc, lead=lead)
        h = self._norm(h, blk.norm2, zq, lead=lead)
        sc = x
        if blk.in_channels != blk.out_channels:
            sc = self._pointwise(x.reshape(-1, C), blk.conv_shortcut).view(B, T, H, W, blk.out_channels)
        return self._conv(h, blk.conv2.conv, (3, 3, 3), (T, H, W), residual=sc, cc=cc, lead=lead)

    def _upsample(self, x, up):
        B, T, H, W, C = x.shape
        ups_t, To = 0, T
        if up.compress_time and T > 1:
            ups_t, To = (2, 1 + 2 * (T - 1)) if T % 2 == 1 else (1, 2 * T)
        return self._conv(x, up.conv, (1, 3, 3), (To, 2 * H, 2 * W), ups_s=1, ups_t=ups_t)

    def _downsample(self, x, down):
        B, T, H, W, C = x.shape
        if down.compress_time and T > 1:
            # avg_pool1d(k=2, s=2) over time with the first frame of an odd clip kept apart: a handful of frames, host glue
            if T % 2 == 1:
                rest = x[:, 1:].reshape(B, (T - 1) // 2, 2, H, W, C).float().mean(2).to(BF16)
                x = torch.cat([x[:, :1], rest], dim=1).contiguous()
            else:
                x = x.reshape(B, T // 2, 2, H, W, C).float().mean(2).to(BF16).contiguous()
            T = x.shape[1]
        Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1                       # pad (0, 1, 0, 1), kernel 3, stride 2
        return self._conv(x, down.conv, (1, 3, 3), (T, Ho, Wo), stride=2, pad_lo=0)

    @staticmethod
    def _channels_last(x, cpad=0):
        x = x.permute(0, 2, 3, 4, 1)
        if cpad:
            x = torch.nn.functional.pad(x, (0, cpad))
        return x.to(BF16).contiguous()

    def _check(self, x):
        if not x.is_cuda:
            raise RuntimeError("orv_amd.vae runs on MI355X only: move the VAE and its input to the GPU (no CPU fallback)")
        if self.dtype != BF16:
            raise RuntimeError(f"orv_amd.vae kernels are bf16: call vae.to(torch.bfloat16) (got {self.dtype})")

    # ---- public: decode / encode ----
    num_latent_frames_batch_size = 2          # diffusers' frame batching: every GroupNorm normalises over ONE batch of frames,
    num_sample_frames_batch_size = 8          # so the batching is part of the arithmetic, not a memory knob

    @staticmethod
    def frame_batches(num_frames, size):
        """``max(n // size, 1)`` batches, the FIRST one takes the remainder (5 latent frames -> [0,3) [3,5))."""
        nb, rem = max(num_frames // size, 1), num_frames % size
        return [(size * i + (0 if i == 0 else rem), min(size * (i + 1) + rem, num_frames)) for i in range(nb)]

    def _decode_batch(self, zq, cc):
        d = self.decoder
        B, T, H, W, _ = zq.shape
        x = self._conv(zq, d.conv_in.conv, (3, 3, 3), (T, H, W), cc=cc)
        for r in d.mid_block.resnets:
            x = self._resnet(x, r, zq, cc)
        for up in d.up_blocks:
            for r in up.resnets:
                x = self._resnet(x, r, zq, cc)
            if up.upsamplers is not None:
                x = self._upsample(x, up.upsamplers[0])
        _, T, H, W, _ = x.shape
        lead = 2 if cc is not None else 0
        x = self._norm(x, d.norm_out, zq, lead=lead)
        return self._conv(x, d.conv_out.conv, (3, 3, 3), (T, H, W), cc=cc, lead=lead)

    def _encode_batch(self, h, cc):
        e = self.encoder
        B, T, H, W, _ = h.shape
        h = self._conv(h, self._padded_conv_in(), (3, 3, 3), (T, H, W), cc=cc)
        for blk in e.down_blocks:
            for r in blk.resnets:
                h = self._resnet(h, r, None, cc)
            if blk.downsamplers is not None:
                h = self._downsample(h, blk.downsamplers[0])
        for r in e.mid_block.resnets:
            h = self._resnet(h, r, None, cc)
        _, T, H, W, _ = h.shape
        lead = 2 if cc is not None else 0
        h = self._norm(h, e.norm_out, None, lead=lead)
        return self._conv(h, e.conv_out.conv, (3, 3, 3), (T, H, W), cc=cc, lead=lead)

    def _batched(self, fn, x, size):
        outs, prev = [], {}
        for a, b in self.frame_batches(x.shape[1], size):
            new = {}
            outs.append(fn(x[:, a:b].contiguous(), (prev, new)))
            prev = new
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)

    MAX_TILE_STREAMS = 3

    def _tiled(self, fn, x, frame_batch, tile_h, tile_w, blend_h_ext, blend_w_ext, limit_h, limit_w):
        """Tiled decode / encode on channels-last ``x`` [B, T, H, W, C] (diffusers ``tiled_decode`` / ``tiled_encode``): ``fn`` on
        overlapping tiles (stride = tile x (1 - overlap factor); each tile frame-batched with its own conv caches and GroupNorm
        statistics), then every tile is blended IN PLACE, in raster order, with the (already blended) tile above and the one
        to its left (``orv_vae_blend``), cropped to ``limit`` and concatenated."""
        H, W = x.shape[2], x.shape[3]
        step_h = int(tile_h * (1 - self.tile_overlap_factor_height))
        step_w = int(tile_w * (1 - self.tile_overlap_factor_width))
        # The tiles are independent until the blend (own frame batching, own GroupNorm statistics, own conv caches) and small: at the
        # reference geometry (40 x 60 latent: 30x45, 30x24, 15x45, 15x24) most of their launches fill a fraction of the 256 CUs.  Every
        # tile therefore runs on its own HIP stream (forked behind the input, joined before the blends); same kernels on the same
        # data, so the result is bit-identical to the serial loop (``ORV_VAE_TILE_STREAMS=0``).
        coords = [(i, j) for i in range(0, H, step_h) for j in range(0, W, step_w)]
        n_cols = len(range(0, W, step_w))
        use_streams = x.is_cuda and len(coords) > 1 and os.environ.get("ORV_VAE_TILE_STREAMS", "1") != "0"
        tiles = []
        if use_streams:
            self._prepack(fn)                    # weight repacking happens once, on THIS stream, before anybody forks
            cur = torch.cuda.current_stream(x.device)
            # at most MAX_TILE_STREAMS side streams, tiles dealt round-robin (the caching allocator keeps one pool PER stream and
            # every in-flight tile's activations are live at once, so an uncapped stream count multiplies peak memory with the tile count;
            # the reference geometry has 4 tiles = 3 side streams, larger frames reuse them)
            n_side = min(len(coords) - 1, self.MAX_TILE_STREAMS)
            side = getattr(self, "_tile_streams", None)
            if side is None or len(side) < n_side or side[0].device != x.device:
                side = self._tile_streams = [torch.cuda.Stream(device=x.device) for _ in range(n_side)]
            fork = torch.cuda.Event()
            fork.record(cur)
            for k, (i, j) in enumerate(coords):
                st = cur if k == 0 else side[(k - 1) % n_side]
                if k and k <= n_side:
                    st.wait_event(fork)
                with torch.cuda.stream(st):
                    t = self._batched(fn, x[:, :, i:i + tile_h, j:j + tile_w].contiguous(), frame_batch).contiguous()
                if k:
                    t.record_stream(cur)         # produced on a side stream, consumed (blend / cat) and freed on this one
                tiles.append(t)
            for st in side[:n_side]:
                cur.wait_stream(st)
        else:
            tiles = [self._batched(fn, x[:, :, i:i + tile_h, j:j + tile_w].contiguous(), frame_batch).contiguous() for i, j in coords]
        rows = [tiles[r * n_cols:(r + 1) * n_cols] for r in range(len(tiles) // n_cols)]
        result_rows = []
        for i, row in enumerate(rows):
            result_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    above = rows[i - 1][j]
                    ops.vae_blend(above, tile, min(above.shape[2], tile.shape[2], blend_h_ext), horizontal=False)
                if j > 0:
                    left = row[j - 1]
                    ops.vae_blend(left, tile, min(left.shape[3], tile.shape[3], blend_w_ext), horizontal=True)
                result_row.append(tile[:, :, :limit_h, :limit_w])
            result_rows.append(torch.cat(result_row, dim=3))
        return torch.cat(result_rows, dim=2)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """z [B, 16, f, h, w] -> sample [B, 3, 1 + 4 (f - 1), 8 h, 8 w]."""
        self._check(z)
        zc = self._channels_last(z)
        if self.use_tiling and (z.shape[-1] > self.tile_latent_min_width or z.shape[-2] > self.tile_latent_min_height):
            bh = int(self.tile_sample_min_height * self.tile_overlap_factor_height)
            bw = int(self.tile_sample_min_width * self.tile_overlap_factor_width)
            x = self._tiled(self._decode_batch, zc, self.num_latent_frames_batch_size, self.tile_latent_min_height,
                            self.tile_latent_min_width, bh, bw, self.tile_sample_min_height - bh, self.tile_sample_min_width - bw)
        else:
            x = self._batched(self._decode_batch, zc, self.num_latent_frames_batch_size)
        sample = x.permute(0, 4, 1, 2, 3).contiguous().to(z.dtype)
        return DecoderOutput(sample) if return_dict else (sample,)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x [B, 3, F, H, W] in [-1, 1] -> latent_dist over moments [B, 32, 1 + (F - 1) / 4, H / 8, W / 8]."""
        self._check(x)
        h = self._channels_last(x, cpad=(-x.shape[1]) % 8)            # RGB padded to 8 channels (zero weights beyond 3)
        if self.use_tiling and (x.shape[-1] > self.tile_sample_min_width or x.shape[-2] > self.tile_sample_min_height):
            bh = int(self.tile_latent_min_height * self.tile_overlap_factor_height)
            bw = int(self.tile_latent_min_width * self.tile_overlap_factor_width)
            h = self._tiled(self._encode_batch, h, self.num_sample_frames_batch_size, self.tile_sample_min_height,
                            self.tile_sample_min_width, bh, bw, self.tile_latent_min_height - bh, self.tile_latent_min_width - bw)
        else:
            h = self._batched(self._encode_batch, h, self.num_sample_frames_batch_size)
        moments = h.permute(0, 4, 1, 2, 3).contiguous().to(x.dtype)
        dist = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(dist) if return_dict else (dist,)

    def _padded_conv_in(self):
        """encoder.conv_in sees 3 input channels; the gather works on 16-byte channel groups, so a view of the convolution with
        the input channels zero-padded to 8 is kept beside the parameter (rebuilt when the weight changes)."""
        conv = self.encoder.conv_in.conv
        ver = (conv.weight.data_ptr(), conv.weight._version)
        hit = self._packed.get("conv_in_pad")
        if hit is None or hit[0] != ver:
            ci = conv.weight.shape[1]
            pad = (-ci) % 8
            shim = nn.Conv3d(ci + pad, conv.weight.shape[0], 3, bias=conv.bias is not None).to(conv.weight.device, conv.weight.dtype)
            with torch.no_grad():
                shim.weight.zero_()
                shim.weight[:, :ci] = conv.weight
                if conv.bias is not None:
                    shim.bias.copy_(conv.bias)
            hit = (ver, shim)
            self._packed["conv_in_pad"] = hit
        return hit[1]
