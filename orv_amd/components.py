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


class VideoProcessor:
    """Host-side image / video pre- and post-processing of the pipeline: the reference's ``VideoProcessor`` override
    (/root/reference/orv/models/components.py:236-386) over diffusers' ``VaeImageProcessor`` defaults (``do_resize``,
    ``do_normalize``, Lanczos resampling, [0,1] -> [-1,1]).  CPU work on one reference frame per clip; nothing here is on the
    denoise path.  What ORV added is kept: 5-D tensors are accepted and pre-encoded latents (``C == vae_latent_channels`` or
    twice that: un-sampled moments) pass through untouched (:347-365)."""

    def __init__(self, vae_latent_channels: int = 16, vae_scale_factor: int = 8, do_resize: bool = True,
                 do_normalize: bool = True, resample: str = "lanczos"):
        from .cogvideox_control import FrozenConfig
        self.config = FrozenConfig(vae_latent_channels=vae_latent_channels, vae_scale_factor=vae_scale_factor,
                                   do_resize=do_resize, do_normalize=do_normalize, resample=resample, do_binarize=False,
                                   do_convert_rgb=False, do_convert_grayscale=False)

    # -- helpers with diffusers' semantics --
    def get_default_height_width(self, image, height=None, width=None):
        import PIL.Image
        if height is None:
            height = image.height if isinstance(image, PIL.Image.Image) else image.shape[-2]
        if width is None:
            width = image.width if isinstance(image, PIL.Image.Image) else image.shape[-1]
        f = self.config.vae_scale_factor
        return height - height % f, width - width % f

    def _resize_pil(self, img, height, width):
        import PIL.Image
        modes = {"lanczos": PIL.Image.LANCZOS, "bilinear": PIL.Image.BILINEAR, "bicubic": PIL.Image.BICUBIC,
                 "nearest": PIL.Image.NEAREST}
        return img.resize((width, height), resample=modes[self.config.resample])

    @staticmethod
    def pil_to_numpy(images):
        import numpy as np
        return np.stack([np.array(i).astype(np.float32) / 255.0 for i in images], axis=0)

    @staticmethod
    def numpy_to_pt(images):
        if images.ndim == 3:
            images = images[..., None]
        return torch.from_numpy(images.transpose(0, 3, 1, 2))

    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None, **_unused) -> torch.Tensor:
        import numpy as np
        import PIL.Image
        if torch.is_tensor(image) and image.ndim in (4, 5):
            ch = image.shape[1]
            if ch in (self.config.vae_latent_channels, self.config.vae_latent_channels * 2):
                return image                                      # latents / moments: no preprocessing (:355-365)
        if not isinstance(image, list):
            image = [image]
        if isinstance(image[0], PIL.Image.Image):
            if self.config.do_resize:
                height, width = self.get_default_height_width(image[0], height, width)
                image = [self._resize_pil(i, height, width) for i in image]
            image = self.numpy_to_pt(self.pil_to_numpy(image))
        elif isinstance(image[0], np.ndarray):
            image = np.concatenate(image, axis=0) if image[0].ndim == 4 else np.stack(image, axis=0)
            image = self.numpy_to_pt(image)
            height, width = self.get_default_height_width(image, height, width)
            if self.config.do_resize:
                image = torch.nn.functional.interpolate(image, size=(height, width))
        elif torch.is_tensor(image[0]):
            image = torch.cat(image, dim=0) if image[0].ndim in (4, 5) else torch.stack(image, dim=0)
            ch = image.shape[1]
            if ch in (self.config.vae_latent_channels, self.config.vae_latent_channels * 2):
                return image
            height, width = self.get_default_height_width(image, height, width)
            if self.config.do_resize and image.ndim == 4:
                image = torch.nn.functional.interpolate(image, size=(height, width))
        else:
            raise ValueError("Input is in incorrect format. Currently, we only support PIL.Image.Image, np.ndarray, torch.Tensor")
        if self.config.do_normalize and image.min() >= 0:
            image = 2.0 * image - 1.0
        return image

    def postprocess_video(self, video: torch.Tensor, output_type: str = "pil"):
        """video [B, C, F, H, W] in [-1, 1] -> per clip: list of PIL frames ('pil'), [F,H,W,C] arrays ('np') or [F,C,H,W]
        tensors ('pt'), as diffusers' ``VideoProcessor.postprocess_video``."""
        import numpy as np
        import PIL.Image
        if output_type not in ("pil", "np", "pt"):
            raise ValueError(f"output_type={output_type} is not supported. Make sure to choose one of ['np', 'pt', 'pil']")
        outs = []
        for b in range(video.shape[0]):
            frames = (video[b].permute(1, 0, 2, 3).float() / 2 + 0.5).clamp(0, 1)          # [F, C, H, W] in [0, 1]
            if output_type == "pt":
                outs.append(frames)
                continue
            arr = frames.cpu().permute(0, 2, 3, 1).numpy()
            if output_type == "np":
                outs.append(arr)
                continue
            u8 = (arr * 255).round().astype("uint8")
            outs.append([PIL.Image.fromarray(f.squeeze(-1), mode="L") if f.shape[-1] == 1 else PIL.Image.fromarray(f) for f in u8])
        if output_type == "np":
            return np.stack(outs)
        if output_type == "pt":
            return torch.stack(outs)
        return outs
