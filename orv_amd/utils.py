"""Mirror of the hot-path helpers of ``orv.utils`` (/root/reference/orv/utils.py:178-239)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .embeddings import rope_3d


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """orv/utils.py:178-193."""
    tw, th = tgt_width, tgt_height
    h, w = src
    r = h / w
    if r > (th / tw):
        resize_height, resize_width = th, int(round(th / h * w))
    else:
        resize_width, resize_height = tw, int(round(tw / w * h))
    top, left = int(round((th - resize_height) / 2.0)), int(round((tw - resize_width) / 2.0))
    return (top, left), (top + resize_height, left + resize_width)


def prepare_rotary_positional_embeddings(height: int, width: int, num_frames: int, vae_scale_factor_spatial: int = 8,
                                         patch_size: int = 2, patch_size_t: Optional[int] = None,
                                         attention_head_dim: int = 64, device: Optional[torch.device] = None,
                                         base_height: int = 480, base_width: int = 720) -> Tuple[torch.Tensor, torch.Tensor]:
    """orv/utils.py:196-239 (same signature): RoPE cos/sin tables [T*h*w, head_dim] fp32."""
    gh = height // (vae_scale_factor_spatial * patch_size)
    gw = width // (vae_scale_factor_spatial * patch_size)
    bw = base_width // (vae_scale_factor_spatial * patch_size)
    bh = base_height // (vae_scale_factor_spatial * patch_size)
    if patch_size_t is None:
        crops = get_resize_crop_region_for_grid((gh, gw), bw, bh)
        cos, sin = rope_3d(attention_head_dim, crops, (gh, gw), num_frames)
    else:
        frames = (num_frames + patch_size_t - 1) // patch_size_t
        cos, sin = rope_3d(attention_head_dim, None, (gh, gw), frames, grid_type="slice", max_hw=(bh, bw))
    return cos.to(device=device), sin.to(device=device)
