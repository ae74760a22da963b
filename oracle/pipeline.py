"""CPU oracle: restatement of the denoise loop, latent preparation and the training-step tail.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows

* ``CogVideoXImageToVideoPipelineTraj.prepare_latents``  /root/reference/orv/models/cogvideox_control.py:1115-1225
* ``CogVideoXImageToVideoPipelineTraj.__call__`` loop    :1402-1473  (latent-in / latent-out part only)
* train step tail  /root/reference/orv/pipeline/train_cogvideox_control_to_video_sft.py:1005-1079

Scheduler arithmetic comes from oracle/leaf.py (diffusers semantics, PARITY UNPINNED).
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch

from . import leaf
from .dit import dit_forward


def prepare_latents(image, batch_size, num_channels_latents, num_frames, height, width, dtype, generator=None,
                    latents=None, scaling_factor=1.15258426, invert_scale_latents=False, init_noise_sigma=1.0,
                    vae_scale_factor_spatial=8, vae_scale_factor_temporal=4, patch_size_t=None, num_views=1):
    """cogvideox_control.py:1115-1225 for pre-encoded inputs (image.ndim == 5, [B, C|2C, F, H, W])."""
    t_lat = (num_frames - 1) // vae_scale_factor_temporal + 1
    shape = (batch_size, num_views * t_lat, num_channels_latents, height // vae_scale_factor_spatial,
             width // vae_scale_factor_spatial)
    if patch_size_t is not None:
        shape = shape[:1] + (shape[1] + shape[1] % patch_size_t,) + shape[2:]
    if image.ndim != 5:
        raise RuntimeError(f"Invalid dimensions of image input: {image.shape=}")
    c = image.size(1)
    if c == 2 * num_channels_latents:
        image_latents = leaf.DiagonalGaussianDistribution(image).sample(generator).permute(0, 2, 1, 3, 4)
    elif c == num_channels_latents:
        image_latents = image.permute(0, 2, 1, 3, 4)
    else:
        raise RuntimeError(f"Invalid input channels {image.shape=} while {num_channels_latents=}!")
    image_latents = (1 / scaling_factor if invert_scale_latents else scaling_factor) * image_latents
    b, vf = image_latents.shape[:2]
    image_latents = image_latents.reshape(b, num_views, vf // num_views, *image_latents.shape[2:])
    f_img = image_latents.size(2)
    if f_img > t_lat:
        raise RuntimeError(f"Invalid input image_frames={f_img} while num_frames={t_lat}!")
    pad = torch.zeros(batch_size, num_views, t_lat - f_img, num_channels_latents, *shape[3:], dtype=dtype)
    image_latents = torch.cat([image_latents, pad], dim=2)
    if patch_size_t is not None:
        first = image_latents[:, :, : image_latents.size(1) % patch_size_t]
        image_latents = torch.cat([first, image_latents], dim=2)
    image_latents = image_latents.flatten(1, 2)
    if latents is None:
        latents = leaf.randn_tensor(shape, generator=generator, device="cpu", dtype=dtype)
    return latents * init_noise_sigma, image_latents


def denoise(sd, cfg, scheduler, latents, image_latents, prompt_embeds, controls, num_inference_steps=50,
            guidance_scale=1.0, use_dynamic_cfg=False, negative_prompt_embeds=None, generator=None, is_mask=None,
            image_rotary_emb=None, ofs=None, step_callback: Optional[Callable] = None):
    """cogvideox_control.py:1402-1473. ``is_mask`` (bool[B] or per-step list) replaces ActionEmbed's RNG draw."""
    cfg_on = guidance_scale > 1.0
    if cfg_on:
        prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
    scheduler.set_timesteps(num_inference_steps)
    timesteps = scheduler.timesteps
    old_x0 = None
    dtype = prompt_embeds.dtype
    gs = guidance_scale
    for i, t in enumerate(timesteps):
        x = torch.cat([latents] * 2) if cfg_on else latents
        img = torch.cat([image_latents] * 2) if cfg_on else image_latents
        x = torch.cat([scheduler.scale_model_input(x, t), img], dim=2)
        m = is_mask[i] if isinstance(is_mask, (list, tuple)) else is_mask
        noise_pred = dit_forward(sd, cfg, x, prompt_embeds, t.expand(x.shape[0]), actions=controls.get("actions"),
                                 depths=controls.get("depths"), labels=controls.get("labels"), is_mask=m,
                                 image_rotary_emb=image_rotary_emb, ofs=ofs)[0].float()
        if use_dynamic_cfg:
            gs = 1 + guidance_scale * ((1 - math.cos(math.pi * ((num_inference_steps - t.item()) /
                                                                 num_inference_steps) ** 5.0)) / 2)
        if cfg_on:
            u, c = noise_pred.chunk(2)
            noise_pred = u + gs * (c - u)
        if isinstance(scheduler, leaf.CogVideoXDPMScheduler):
            latents, old_x0 = scheduler.step(noise_pred, old_x0, t, timesteps[i - 1] if i > 0 else None, latents,
                                             generator=generator)
        else:
            latents = scheduler.step(noise_pred, t, latents, return_dict=False)[0]
        latents = latents.to(dtype)
        if step_callback is not None:
            step_callback(i, t, latents)
    return latents


def sft_loss(sd, cfg, scheduler, video_latents, image_latents, prompt_embeds, actions, noise, timesteps,
             is_mask=None, frame_mask=None, depths=None, labels=None, image_rotary_emb=None, ofs=None):
    """train_cogvideox_control_to_video_sft.py:1039-1079: add_noise -> forward -> get_velocity (= x0-hat) ->
    mean(1/(1-abar_t) * (x0_hat - x0)^2)."""
    noisy = scheduler.add_noise(video_latents, noise, timesteps)
    x = torch.cat([noisy, image_latents], dim=2)
    out, action_emb, recon = dit_forward(sd, cfg, x, prompt_embeds, timesteps, actions=actions, depths=depths,
                                         labels=labels, is_mask=is_mask, image_rotary_emb=image_rotary_emb,
                                         ofs=ofs, training=True)
    pred = scheduler.get_velocity(out, noisy, timesteps)
    ac = scheduler.alphas_cumprod.to(torch.float32)[timesteps]
    wgt = 1 / (1 - ac)
    while wgt.ndim < pred.ndim:
        wgt = wgt.unsqueeze(-1)
    if frame_mask is None:
        frame_mask = torch.ones(video_latents.size(1), dtype=torch.bool)
    err = wgt * (pred[:, frame_mask] - video_latents[:, frame_mask]) ** 2
    loss = torch.mean(err.reshape(video_latents.shape[0], -1), dim=1).mean()
    return loss, out, recon
