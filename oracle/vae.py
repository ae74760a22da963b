"""CPU oracle: restatement of diffusers' ``AutoencoderKLCogVideoX`` (decoder, and the encoder path used for ONE reference
frame), the VAE the reference decodes / encodes with at /root/reference/orv/models/cogvideox_control.py:1161-1166 (encode of
the reference image) and :1476-1479 (``decode_latents`` -> ``vae.decode(latents).sample``).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **PARITY UNPINNED**: the VAE lives entirely in diffusers (>= 0.32,
``models/autoencoders/autoencoder_kl_cogvideox.py``), which is neither in /root/reference nor installable here, no checkpoint
is reachable, and nothing in the reference's tree holds a VAE input/output pair.  The formulas below are the published
diffusers algorithm restated from its structure (module names == diffusers attribute names == checkpoint keys):

* ``CogVideoXCausalConv3d``  temporal padding = (k_t - 1) copies of the FIRST frame in front (pad_mode "first" of the
  CogVideoX configs) - or, from the second FRAME BATCH on, the last (k_t - 1) input frames of the previous batch
  (``conv_cache``); zero spatial padding.
* Frame batching            ``decode`` runs the decoder on batches of 2 latent frames (the first batch takes the remainder:
  5 latent frames = 3 + 2), ``encode`` on batches of 8 frames (17 = 9 + 8), carrying every causal convolution's
  ``conv_cache`` from one batch to the next.  Convolutions therefore see the whole clip, but every GroupNorm normalises
  over ONE batch of frames - the result is NOT that of running the whole clip at once.
* ``CogVideoXSpatialNorm3D`` GroupNorm(32, eps 1e-6)(f) * conv_y(zq') + conv_b(zq'), zq' = nearest-neighbour resize of the
  latent to f's size - with the FIRST frame resized separately when f has an odd number (> 1) of frames.
* ``CogVideoXResnetBlock3D`` norm1 -> SiLU -> conv1 -> norm2 -> SiLU -> conv2, + (1x1x1 ``conv_shortcut``)(input).
* ``CogVideoXUpsample3D``    nearest x2 in H, W (and in T for ``compress_time`` blocks: the first frame of an odd-length clip is
  only resized spatially, the others are doubled), then a per-frame Conv2d 3x3.
* ``CogVideoXDownsample3D``  (``compress_time``: avg_pool1d(k=2, s=2) over time, first frame kept apart for odd lengths), pad
  (0,1,0,1) zeros, per-frame Conv2d 3x3 stride 2.
* Decoder: conv_in -> mid (2 resnets) -> 4 up blocks (4 resnets each, upsample on all but the last; time doubled in the
  first two) -> norm_out -> SiLU -> conv_out.  Encoder mirrors it with plain GroupNorm resnets and no latent conditioning.

* Tiling (``enable_tiling()``, which both reference entry points call: /root/reference/orv/pipeline/
  inference_control_to_video.py:98-99, evaluation_control_to_video.py:274-275) is ARITHMETIC, not a memory knob - restated
  here FROM MEMORY of diffusers >= 0.32, unpinned like the rest: tile size in sample space = (sample_height / 2, sample_width / 2)
  of the VAE config (480 x 720 -> 240 x 360), in latent space that / 2^(len(block_out_channels) - 1) (30 x 45); overlap factors
  1/6 (height) and 1/5 (width).  ``decode`` of a latent larger than one latent tile in either direction runs the frame-batched
  decoder on every tile z[.., i : i + 30, j : j + 45] with i, j stepping by int(30 * 5/6) = 25, int(45 * 4/5) = 36 - every tile
  with its OWN GroupNorm statistics and conv caches - then blends each tile, in raster order and IN PLACE, with the
  (already blended) tile above (``blend_v``, extent int(240 / 6) = 40 rows) and to the left (``blend_h``, int(360 / 5) = 72
  columns): b[y] = a[-extent + y] (1 - y / extent) + b[y] (y / extent); crops it to (240 - 40) x (360 - 72) and concatenates.
  ``encode`` mirrors it with sample tiles 240 x 360 stepping 200 / 288, latent blend extents 5 / 9 and crop 25 x 36.  A 40 x 60
  latent (320 x 480 video) is 4 tiles - 30x45, 30x24, 15x45, 15x24 - i.e. 1.29 x the voxels of the untiled decode.
  ``enable_slicing()`` (one batch element at a time) changes nothing: every statistic is per batch element already.

THUDM/CogVideoX-2b VAE config: block_out_channels (128, 256, 256, 512), layers_per_block 3, latent_channels 16,
norm_num_groups 32, temporal_compression_ratio 4, scaling_factor 1.15258426, no quant / post-quant conv.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .leaf import DiagonalGaussianDistribution


class ConvCache:
    """conv_cache of diffusers' frame-batched encode / decode: per causal convolution, the last (k_t - 1) input frames of the
    previous batch (``prev``), and what this batch leaves for the next (``new``)."""

    def __init__(self, prev=None):
        self.prev, self.new = prev or {}, {}


class CogVideoXCausalConv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, pad_mode="first"):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        self.time_kernel_size = k[0]
        self.pad_mode = pad_mode
        stride = stride if isinstance(stride, tuple) else (stride, 1, 1)
        self.conv = nn.Conv3d(in_channels, out_channels, k, stride=stride, padding=(0, (k[1] - 1) // 2, (k[2] - 1) // 2))

    def forward(self, x, cc: Optional[ConvCache] = None):
        if self.time_kernel_size > 1:
            if cc is not None and id(self) in cc.prev:
                x = torch.cat([cc.prev[id(self)], x], dim=2)
            else:
                x = torch.cat([x[:, :, :1]] * (self.time_kernel_size - 1) + [x], dim=2)
            if cc is not None:
                cc.new[id(self)] = x[:, :, -(self.time_kernel_size - 1):].clone()
        return self.conv(x)


class CogVideoXSpatialNorm3D(nn.Module):
    def __init__(self, f_channels, zq_channels, groups=32):
        super().__init__()
        self.norm_layer = nn.GroupNorm(num_channels=f_channels, num_groups=groups, eps=1e-6, affine=True)
        self.conv_y = CogVideoXCausalConv3d(zq_channels, f_channels, kernel_size=1)
        self.conv_b = CogVideoXCausalConv3d(zq_channels, f_channels, kernel_size=1)

    def forward(self, f, zq):
        if f.shape[2] > 1 and f.shape[2] % 2 == 1:
            z_first = F.interpolate(zq[:, :, :1], size=(1,) + tuple(f.shape[-2:]))
            z_rest = F.interpolate(zq[:, :, 1:], size=(f.shape[2] - 1,) + tuple(f.shape[-2:]))
            zq = torch.cat([z_first, z_rest], dim=2)
        else:
            zq = F.interpolate(zq, size=tuple(f.shape[-3:]))
        return self.norm_layer(f) * self.conv_y(zq) + self.conv_b(zq)


class CogVideoXResnetBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels=None, groups=32, eps=1e-6, spatial_norm_dim=None, pad_mode="first"):
        super().__init__()
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        if spatial_norm_dim is None:
            self.norm1 = nn.GroupNorm(num_channels=in_channels, num_groups=groups, eps=eps)
            self.norm2 = nn.GroupNorm(num_channels=out_channels, num_groups=groups, eps=eps)
        else:
            self.norm1 = CogVideoXSpatialNorm3D(in_channels, spatial_norm_dim, groups)
            self.norm2 = CogVideoXSpatialNorm3D(out_channels, spatial_norm_dim, groups)
        self.conv1 = CogVideoXCausalConv3d(in_channels, out_channels, 3, pad_mode=pad_mode)
        self.conv2 = CogVideoXCausalConv3d(out_channels, out_channels, 3, pad_mode=pad_mode)
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv3d(in_channels, out_channels, kernel_size=1)

    def forward(self, x, zq=None, cc=None):
        h = self.norm1(x, zq) if zq is not None else self.norm1(x)
        h = self.conv1(F.silu(h), cc)
        h = self.norm2(h, zq) if zq is not None else self.norm2(h)
        h = self.conv2(F.silu(h), cc)
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut(x)
        return h + x


class CogVideoXUpsample3D(nn.Module):
    def __init__(self, in_channels, out_channels, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.compress_time = compress_time

    def forward(self, x):
        if self.compress_time:
            if x.shape[2] > 1 and x.shape[2] % 2 == 1:
                first = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
                rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
                x = torch.cat([first, rest], dim=2)
            elif x.shape[2] > 1:
                x = F.interpolate(x, scale_factor=2.0)
            else:
                x = F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
        else:
            b, c, t, h, w = x.shape
            x = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), scale_factor=2.0)
            x = x.reshape(b, t, c, *x.shape[2:]).permute(0, 2, 1, 3, 4)
        b, c, t, h, w = x.shape
        x = self.conv(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
        return x.reshape(b, t, *x.shape[1:]).permute(0, 2, 1, 3, 4)


class CogVideoXDownsample3D(nn.Module):
    def __init__(self, in_channels, out_channels, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=2, padding=0)
        self.compress_time = compress_time

    def forward(self, x):
        if self.compress_time:
            b, c, t, h, w = x.shape
            x = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, t)
            if x.shape[-1] % 2 == 1:
                first, rest = x[..., 0], x[..., 1:]
                if rest.shape[-1] > 0:
                    rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
                x = torch.cat([first[..., None], rest], dim=-1)
            else:
                x = F.avg_pool1d(x, kernel_size=2, stride=2)
            x = x.reshape(b, h, w, c, x.shape[-1]).permute(0, 3, 4, 1, 2)
        x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        b, c, t, h, w = x.shape
        x = self.conv(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
        return x.reshape(b, t, x.shape[1], x.shape[2], x.shape[3]).permute(0, 2, 1, 3, 4)


class CogVideoXMidBlock3D(nn.Module):
    def __init__(self, in_channels, num_layers=2, groups=32, eps=1e-6, spatial_norm_dim=None):
        super().__init__()
        self.resnets = nn.ModuleList([CogVideoXResnetBlock3D(in_channels, in_channels, groups, eps, spatial_norm_dim)
                                      for _ in range(num_layers)])

    def forward(self, x, zq=None, cc=None):
        for r in self.resnets:
            x = r(x, zq, cc)
        return x


class CogVideoXUpBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, groups, eps, spatial_norm_dim, add_upsample, compress_time):
        super().__init__()
        self.resnets = nn.ModuleList([CogVideoXResnetBlock3D(in_channels if i == 0 else out_channels, out_channels, groups, eps,
                                                             spatial_norm_dim) for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([CogVideoXUpsample3D(out_channels, out_channels, compress_time)]) if add_upsample else None

    def forward(self, x, zq, cc=None):
        for r in self.resnets:
            x = r(x, zq, cc)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x)
        return x


class CogVideoXDownBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, groups, eps, add_downsample, compress_time):
        super().__init__()
        self.resnets = nn.ModuleList([CogVideoXResnetBlock3D(in_channels if i == 0 else out_channels, out_channels, groups, eps)
                                      for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([CogVideoXDownsample3D(out_channels, out_channels, compress_time)]) if add_downsample else None

    def forward(self, x, cc=None):
        for r in self.resnets:
            x = r(x, None, cc)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x)
        return x


class CogVideoXDecoder3D(nn.Module):
    def __init__(self, in_channels=16, out_channels=3, block_out_channels=(128, 256, 256, 512), layers_per_block=3,
                 norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = CogVideoXCausalConv3d(in_channels, rev[0], 3)
        self.mid_block = CogVideoXMidBlock3D(rev[0], 2, norm_num_groups, norm_eps, in_channels)
        self.up_blocks = nn.ModuleList()
        out_c = rev[0]
        levels = int(math.log2(temporal_compression_ratio))
        for i in range(len(rev)):
            prev, out_c = out_c, rev[i]
            self.up_blocks.append(CogVideoXUpBlock3D(prev, out_c, layers_per_block + 1, norm_num_groups, norm_eps, in_channels,
                                                     add_upsample=i != len(rev) - 1, compress_time=i < levels))
        self.norm_out = CogVideoXSpatialNorm3D(rev[-1], in_channels, norm_num_groups)
        self.conv_act = nn.SiLU()
        self.conv_out = CogVideoXCausalConv3d(rev[-1], out_channels, 3)

    def forward(self, z, cc=None):
        h = self.conv_in(z, cc)
        h = self.mid_block(h, z, cc)
        for up in self.up_blocks:
            h = up(h, z, cc)
        return self.conv_out(self.conv_act(self.norm_out(h, z)), cc)


class CogVideoXEncoder3D(nn.Module):
    def __init__(self, in_channels=3, out_channels=16, block_out_channels=(128, 256, 256, 512), layers_per_block=3,
                 norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4):
        super().__init__()
        levels = int(math.log2(temporal_compression_ratio))
        self.conv_in = CogVideoXCausalConv3d(in_channels, block_out_channels[0], 3)
        self.down_blocks = nn.ModuleList()
        out_c = block_out_channels[0]
        for i in range(len(block_out_channels)):
            prev, out_c = out_c, block_out_channels[i]
            self.down_blocks.append(CogVideoXDownBlock3D(prev, out_c, layers_per_block, norm_num_groups, norm_eps,
                                                         add_downsample=i != len(block_out_channels) - 1,
                                                         compress_time=i < levels))
        self.mid_block = CogVideoXMidBlock3D(block_out_channels[-1], 2, norm_num_groups, norm_eps, None)
        self.norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = CogVideoXCausalConv3d(block_out_channels[-1], 2 * out_channels, 3)

    def forward(self, x, cc=None):
        h = self.conv_in(x, cc)
        for d in self.down_blocks:
            h = d(h, cc)
        h = self.mid_block(h, None, cc)
        return self.conv_out(self.conv_act(self.norm_out(h)), cc)


class AutoencoderKLCogVideoX(nn.Module):
    """encode(x[B,3,F,H,W]).latent_dist / decode(z[B,16,f,h,w]).sample, as the pipeline calls them."""

    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), latent_channels=16,
                 layers_per_block=3, norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4,
                 scaling_factor=1.15258426, invert_scale_latents=False, sample_height=480, sample_width=720):
        super().__init__()
        self.use_tiling = self.use_slicing = False
        self.tile_sample_min_height, self.tile_sample_min_width = sample_height // 2, sample_width // 2
        self.tile_latent_min_height = int(self.tile_sample_min_height / (2 ** (len(block_out_channels) - 1)))
        self.tile_latent_min_width = int(self.tile_sample_min_width / (2 ** (len(block_out_channels) - 1)))
        self.tile_overlap_factor_height, self.tile_overlap_factor_width = 1 / 6, 1 / 5
        self.encoder = CogVideoXEncoder3D(in_channels, latent_channels, block_out_channels, layers_per_block, norm_eps,
                                          norm_num_groups, temporal_compression_ratio)
        self.decoder = CogVideoXDecoder3D(latent_channels, out_channels, block_out_channels, layers_per_block, norm_eps,
                                          norm_num_groups, temporal_compression_ratio)
        self.config = dict(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                           latent_channels=latent_channels, layers_per_block=layers_per_block, norm_eps=norm_eps,
                           norm_num_groups=norm_num_groups, temporal_compression_ratio=temporal_compression_ratio,
                           scaling_factor=scaling_factor, invert_scale_latents=invert_scale_latents, sample_height=sample_height,
                           sample_width=sample_width)

    num_latent_frames_batch_size = 2
    num_sample_frames_batch_size = 8

    @staticmethod
    def frame_batches(num_frames, size):
        """diffusers' batching: ``max(n // size, 1)`` batches, the FIRST one takes the remainder."""
        nb, rem = max(num_frames // size, 1), num_frames % size
        return [(size * i + (0 if i == 0 else rem), min(size * (i + 1) + rem, num_frames)) for i in range(nb)]

    def _batched(self, net, x, size):
        outs, prev = [], None
        for a, b in self.frame_batches(x.shape[2], size):
            cc = ConvCache(prev)
            outs.append(net(x[:, :, a:b], cc))
            prev = cc.new
        return torch.cat(outs, dim=2)

    # ---- tiling (from memory of diffusers >= 0.32; see the module header) ----
    def enable_tiling(self, tile_sample_min_height=None, tile_sample_min_width=None, tile_overlap_factor_height=None,
                      tile_overlap_factor_width=None):
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        down = 2 ** (len(self.config["block_out_channels"]) - 1)
        self.tile_latent_min_height = int(self.tile_sample_min_height / down)
        self.tile_latent_min_width = int(self.tile_sample_min_width / down)
        self.tile_overlap_factor_height = tile_overlap_factor_height or self.tile_overlap_factor_height
        self.tile_overlap_factor_width = tile_overlap_factor_width or self.tile_overlap_factor_width

    def disable_tiling(self):
        self.use_tiling = False

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    @staticmethod
    def blend_v(a, b, blend_extent):
        blend_extent = min(a.shape[3], b.shape[3], blend_extent)
        for y in range(blend_extent):
            b[:, :, :, y, :] = a[:, :, :, -blend_extent + y, :] * (1 - y / blend_extent) + b[:, :, :, y, :] * (y / blend_extent)
        return b

    @staticmethod
    def blend_h(a, b, blend_extent):
        blend_extent = min(a.shape[4], b.shape[4], blend_extent)
        for x in range(blend_extent):
            b[:, :, :, :, x] = a[:, :, :, :, -blend_extent + x] * (1 - x / blend_extent) + b[:, :, :, :, x] * (x / blend_extent)
        return b

    def _tiled(self, net, x, frame_batch, tile_h, tile_w, blend_h_ext, blend_w_ext, limit_h, limit_w):
        """Shared body of tiled_encode / tiled_decode: ``net`` on overlapping tiles (own frame batching, own conv caches), then
        the raster-order in-place seam blend, crop and concatenation."""
        height, width = x.shape[3], x.shape[4]
        step_h = int(tile_h * (1 - self.tile_overlap_factor_height))
        step_w = int(tile_w * (1 - self.tile_overlap_factor_width))
        rows = []
        for i in range(0, height, step_h):
            rows.append([self._batched(net, x[:, :, :, i:i + tile_h, j:j + tile_w], frame_batch) for j in range(0, width, step_w)])
        result_rows = []
        for i, row in enumerate(rows):
            result_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self.blend_v(rows[i - 1][j], tile, blend_h_ext)
                if j > 0:
                    tile = self.blend_h(row[j - 1], tile, blend_w_ext)
                result_row.append(tile[:, :, :, :limit_h, :limit_w])
            result_rows.append(torch.cat(result_row, dim=4))
        return torch.cat(result_rows, dim=3)

    def tiled_encode(self, x):
        bh = int(self.tile_latent_min_height * self.tile_overlap_factor_height)
        bw = int(self.tile_latent_min_width * self.tile_overlap_factor_width)
        return self._tiled(self.encoder, x, self.num_sample_frames_batch_size, self.tile_sample_min_height, self.tile_sample_min_width,
                           bh, bw, self.tile_latent_min_height - bh, self.tile_latent_min_width - bw)

    def tiled_decode(self, z):
        bh = int(self.tile_sample_min_height * self.tile_overlap_factor_height)
        bw = int(self.tile_sample_min_width * self.tile_overlap_factor_width)
        return self._tiled(self.decoder, z, self.num_latent_frames_batch_size, self.tile_latent_min_height, self.tile_latent_min_width,
                           bh, bw, self.tile_sample_min_height - bh, self.tile_sample_min_width - bw)

    def encode(self, x):
        if self.use_tiling and (x.shape[-1] > self.tile_sample_min_width or x.shape[-2] > self.tile_sample_min_height):
            return DiagonalGaussianDistribution(self.tiled_encode(x))
        return DiagonalGaussianDistribution(self._batched(self.encoder, x, self.num_sample_frames_batch_size))

    def decode(self, z):
        if self.use_tiling and (z.shape[-1] > self.tile_latent_min_width or z.shape[-2] > self.tile_latent_min_height):
            return self.tiled_decode(z)
        return self._batched(self.decoder, z, self.num_latent_frames_batch_size)
