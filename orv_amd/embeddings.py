"""Host-side (one-off) positional tables for the DiT: 3-D sin-cos (CogVideoX-2B) and 3-D RoPE (5B / *_rope configs).

These are computed once per shape on the host in float64/float32 and uploaded; they are not on the per-step path.
Semantics follow diffusers ``get_3d_sincos_pos_embed`` / ``get_3d_rotary_pos_embed`` as called by the reference at
/root/reference/orv/models/cogvideox_control.py:531-547,666-674 and /root/reference/orv/utils.py:196-239.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def _sincos_1d(dim: int, pos: torch.Tensor) -> torch.Tensor:
    omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
    ang = pos.reshape(-1).to(torch.float64)[:, None] * omega[None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=1)


def sincos_3d(embed_dim: int, grid_w: int, grid_h: int, frames: int, spatial_scale: float = 1.0,
              temporal_scale: float = 1.0) -> torch.Tensor:
    """[frames * grid_h * grid_w, embed_dim] float32; temporal quarter first, then (x | y) spatial halves."""
    if embed_dim % 4:
        raise ValueError("`embed_dim` must be divisible by 4")
    d_sp, d_t = 3 * embed_dim // 4, embed_dim // 4
    ys = (torch.arange(grid_h, dtype=torch.float32) / spatial_scale)[:, None].expand(grid_h, grid_w)
    xs = (torch.arange(grid_w, dtype=torch.float32) / spatial_scale)[None, :].expand(grid_h, grid_w)
    spatial = torch.cat([_sincos_1d(d_sp // 2, xs), _sincos_1d(d_sp // 2, ys)], dim=1)          # [gh*gw, d_sp]
    temporal = _sincos_1d(d_t, torch.arange(frames, dtype=torch.float32) / temporal_scale)      # [T, d_t]
    n = grid_h * grid_w
    out = torch.cat([temporal[:, None, :].expand(frames, n, d_t), spatial[None].expand(frames, n, d_sp)], dim=-1)
    return out.reshape(frames * n, embed_dim).to(torch.float32)


def _rope_axis(dim: int, pos: torch.Tensor, theta: float = 10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = pos.to(torch.float32)[:, None] * freqs[None, :]
    return ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)


def rope_3d(head_dim: int, crops: Optional[Tuple[Tuple[int, int], Tuple[int, int]]], grid_hw: Tuple[int, int],
            frames: int, grid_type: str = "linspace", max_hw: Optional[Tuple[int, int]] = None):
    """(cos, sin), each [frames*gh*gw, head_dim] float32; feature split t:h:w = d/4 : 3d/8 : 3d/8."""
    gh, gw = grid_hw
    if grid_type == "linspace":
        (t0, l0), (t1, l1) = crops
        ph = torch.linspace(t0, t1 * (gh - 1) / gh, gh, dtype=torch.float32)
        pw = torch.linspace(l0, l1 * (gw - 1) / gw, gw, dtype=torch.float32)
        ptm = torch.linspace(0, frames * (frames - 1) / frames, frames, dtype=torch.float32)
    elif grid_type == "slice":
        ph = torch.arange(max_hw[0], dtype=torch.float32)
        pw = torch.arange(max_hw[1], dtype=torch.float32)
        ptm = torch.arange(frames, dtype=torch.float32)
    else:
        raise ValueError("Invalid value passed for `grid_type`.")
    dt, dh, dw = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3
    out = []
    for k in (0, 1):
        ft, fh, fw = _rope_axis(dt, ptm)[k][:frames], _rope_axis(dh, ph)[k][:gh], _rope_axis(dw, pw)[k][:gw]
        full = torch.cat([ft[:, None, None, :].expand(frames, gh, gw, dt), fh[None, :, None, :].expand(frames, gh, gw, dh),
                          fw[None, None, :, :].expand(frames, gh, gw, dw)], dim=-1)
        out.append(full.reshape(frames * gh * gw, head_dim).contiguous())
    return out[0], out[1]
