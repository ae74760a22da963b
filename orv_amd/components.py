"""Mirror of ``orv.models.components`` for the hot path (/root/reference/orv/models/components.py:13-104).

``ActionEmbed`` / ``ActionRecon`` keep the reference's constructor kwargs, attribute names and ``state_dict`` keys
(``mlp.0``, ``mlp.3``, ``mask_embed`` / ``mlp.0``, ``mlp.2``); their arithmetic runs in ``orv_skinny_linear``.
The ``nn.Linear`` / ``nn.Embedding`` members are parameter containers only - their torch ``forward`` is never called.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from . import ops


@dataclass
class Transformer3DModelTrajOutput:
    """components.py:13-17 (``sample`` comes from diffusers' Transformer2DModelOutput)."""
    sample: torch.Tensor
    is_action_mask: Optional[torch.Tensor] = None
    actions_recon: Optional[torch.Tensor] = None

    def __getitem__(self, i):
        return (self.sample, self.is_action_mask, self.actions_recon)[i]


class ActionEmbed(nn.Module):
    """7-DoF x ``compress_ratio`` action chunks -> one ``hidden_size`` embedding per latent frame (components.py:20-71)."""

    def __init__(self, state_dim: int, hidden_size: int, dropout: float = 0., compress_ratio: int = 1,
                 patch_size_t: Optional[int] = None, mask: Optional[bool] = False) -> None:
        super().__init__()
        self.state_dim, self.compress_ratio = state_dim, compress_ratio
        self.patch_size_t = patch_size_t or 1
        self.mask = mask
        self.mlp = nn.Sequential(
            nn.Linear(state_dim * compress_ratio * self.patch_size_t, hidden_size * 4, bias=True),
            nn.GELU(approximate='tanh'), nn.Dropout(dropout),
            nn.Linear(hidden_size * 4, hidden_size, bias=True), nn.Dropout(dropout))
        self.mask_embed = nn.Embedding(num_embeddings=1, embedding_dim=hidden_size)
        # parity hook: when set (bool[B]) it replaces the reference's `torch.rand(B) < 0.1` draw (components.py:67)
        self.forced_mask: Optional[torch.Tensor] = None

    def forward(self, x: torch.Tensor):
        B, F, state_dim = x.shape
        if state_dim != self.state_dim:
            raise ValueError(f'Got mismatched {x.shape=} and {self.state_dim=}.')
        x = torch.cat([torch.zeros_like(x[:, :1]), x], dim=1)           # pad the first frame
        if self.compress_ratio > 1:
            x = x.reshape(B, (F + 1) // self.compress_ratio, -1)
        if self.patch_size_t > 1:
            x = x.reshape(B, x.shape[1] // self.patch_size_t, -1)
        frames = x.shape[1]
        w0, w3 = self.mlp[0], self.mlp[3]
        h = ops.skinny_linear(x.reshape(B * frames, -1).to(torch.bfloat16), w0.weight, w0.bias, act_out="gelu_tanh")
        emb = ops.skinny_linear(h, w3.weight, w3.bias).view(B, frames, -1)
        if self.forced_mask is not None:
            if self.forced_mask.device != x.device or self.forced_mask.dtype != torch.bool:
                self.forced_mask = self.forced_mask.to(device=x.device, dtype=torch.bool)   # once: no H2D copy per call
            is_mask = self.forced_mask
        else:
            is_mask = torch.rand(B, device=x.device) < 0.1
        if self.mask:
            # same result as the reference's masked assignment, without its host sync on `is_mask.sum() > 0`
            emb = torch.where(is_mask[:, None, None], self.mask_embed.weight[None].to(emb.dtype), emb)
        return emb, is_mask


class ActionRecon(nn.Module):
    """Auxiliary head reconstructing actions from the embedding (components.py:74-104)."""

    def __init__(self, state_dim: int, hidden_size: int, compress_ratio: int = 1) -> None:
        super().__init__()
        self.state_dim, self.compress_ratio = state_dim, compress_ratio
        self.mlp = nn.Sequential(nn.Linear(hidden_size, hidden_size * 4, bias=True), nn.GELU(approximate='tanh'),
                                 nn.Linear(hidden_size * 4, state_dim * compress_ratio, bias=True))

    def forward(self, x: torch.Tensor):
        B, F, _ = x.shape
        h = ops.skinny_linear(x.reshape(B * F, -1).contiguous(), self.mlp[0].weight, self.mlp[0].bias,
                              act_out="gelu_tanh")
        y = ops.skinny_linear(h, self.mlp[2].weight, self.mlp[2].bias).view(B, F, -1)
        if self.compress_ratio > 1:
            y = y.reshape(B, int(F * self.compress_ratio), y.shape[-1] // self.compress_ratio).contiguous()
        return y[:, 1:]
