"""Train-step tail of the reference's SFT script as two functions
(/root/reference/orv/pipeline/train_cogvideox_control_to_video_sft.py):

* ``prepare_batch``  :862-964   cached VAE moments -> latents (Gaussian sample x scaling factor, '[B,C,F,H,W] ->
  [B,F,C,H,W]' in ONE kernel, ``orv_gaussian_sample``), image-latent zero padding, p_t frame padding + frame mask,
  noised-image dropout.
* ``sft_step``       :1005-1104 noise / timestep draw, ``add_noise``, transformer forward (HIP kernels, activations
  kept), ``get_velocity`` -> x0-hat, ``mean(1/(1-abar_t) (x0-hat - x0)^2)`` over the unpadded frames, optional action
  reconstruction loss (:1081-1090), backward (hand-written adjoint), gradient all-reduce for data parallel, global-norm
  clip + fused AdamW.

Everything heavy runs in the HIP library; this file is the host-side order of operations only, written so a maintainer
can swap the body of the reference's loop for two calls.
"""
from __future__ import annotations

import random
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import ops
from .utils import prepare_rotary_positional_embeddings

VAE_SCALING_FACTOR = 1.15258426          # train script :75 (CogVideoX-2B VAE); 0.7 for the 5B VAE is passed explicitly
VAE_SCALE_FACTOR_SPATIAL = 8
ACTION_LOSS_WEIGHT = {'rot_loss': 0.4, 'pos_loss': 5, 'grip_loss': 1}      # :1083-1085


@dataclass
class Batch:
    video_latents: torch.Tensor              # [B, F, C, H, W] bf16
    image_latents: torch.Tensor              # [B, F, C, H, W] bf16 (zero padded past the reference frames)
    prompt_embeds: torch.Tensor              # [B, Nt, text_dim]
    actions: Optional[torch.Tensor]          # [B, 4*(F-1), 7] or None
    depth_latents: Optional[torch.Tensor]
    label_latents: Optional[torch.Tensor]
    frame_mask: torch.Tensor                 # bool[F]
    num_views: int = 1


def sample_latents(moments: torch.Tensor, scaling_factor: float = VAE_SCALING_FACTOR,
                   generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """``DiagonalGaussianDistribution(moments).sample() * sf`` then ``permute(0,2,1,3,4)`` (:887-895), fused.
    moments: [B, 2C, F, H, W] (mean ‖ logvar, logvar clamped to [-30, 20] as diffusers does)."""
    eps = torch.randn(moments.shape[0], moments.shape[1] // 2, *moments.shape[2:], device=moments.device,
                      dtype=torch.float32, generator=generator)
    return ops.gaussian_sample(moments.to(torch.bfloat16), eps, scaling_factor)


def prepare_batch(batch: Dict, device, patch_size_t: Optional[int] = None, use_traj: bool = True, use_cond: bool = False,
                  noised_image_dropout: float = 0.0, scaling_factor: float = VAE_SCALING_FACTOR, num_views: int = 1,
                  generator: Optional[torch.Generator] = None) -> Batch:
    """``batch`` is what the reference's collate hands the loop when ``load_tensors`` is on (:864-886):
    ``latents`` / ``images`` moments, ``prompt_embeds``, ``controls`` = {actions, latents_depth, latents_label}."""
    videos = batch['latents'].to(device, non_blocking=True)
    images = batch['images'].to(device, non_blocking=True)
    prompts = batch['prompt_embeds'].to(device, non_blocking=True)
    controls = batch.get('controls', {})
    actions = depths = labels = None
    if use_traj and controls.get('actions', None) is not None:
        actions = controls['actions'].to(device, non_blocking=True)
    if use_cond:
        if 'latents_depth' in controls:
            depths = controls['latents_depth'].to(device, non_blocking=True)
        if 'latents_label' in controls:
            labels = controls['latents_label'].to(device, non_blocking=True)
    video_latents = sample_latents(videos, scaling_factor, generator)
    image_latents = sample_latents(images, scaling_factor, generator)

    num_frames = video_latents.size(1)                                                 # :943-964
    pad_frames = 0
    if patch_size_t and num_frames % patch_size_t != 0:
        if num_views > 1:
            raise RuntimeError("frame padding is not defined for multiview batches (:946-947)")
        pad_frames = patch_size_t - num_frames % patch_size_t
        video_latents = torch.cat([video_latents, video_latents.new_zeros(
            (video_latents.size(0), pad_frames, *video_latents.shape[2:]))], dim=1)
        if actions is not None:
            actions = torch.cat([actions, actions.new_zeros((actions.size(0), pad_frames * 4, actions.size(2)))], dim=1)
    frame_mask = torch.ones(video_latents.size(1), device=device, dtype=torch.bool)
    if pad_frames > 0:
        frame_mask[-pad_frames:] = False

    pad = image_latents.new_zeros((video_latents.size(0), video_latents.size(1) - image_latents.size(1),
                                   *video_latents.shape[2:]))                          # :966-970
    image_latents = torch.cat([image_latents, pad], dim=1)
    depth_latents = sample_latents(depths, scaling_factor, generator) if depths is not None else None
    label_latents = sample_latents(labels, scaling_factor, generator) if labels is not None else None
    if random.random() < noised_image_dropout:                                         # :988-989
        image_latents = torch.zeros_like(image_latents)
    return Batch(video_latents, image_latents, prompts.to(torch.bfloat16), actions, depth_latents, label_latents,
                 frame_mask, num_views)


def sft_loss(transformer, scheduler, b: Batch, noise: torch.Tensor, timesteps: torch.Tensor, use_rope: bool = False,
             is_ofs_embed: bool = False):
    """:1039-1090.  Returns (loss, dict of parts); ``loss.backward()`` runs the hand-written adjoint."""
    cfg = transformer.config
    video_latents = b.video_latents
    batch_size, num_frames, _, height, width = video_latents.shape
    if b.num_views > 1:
        num_frames //= b.num_views
    image_rotary_emb = None
    if use_rope:                                                                       # :1021-1034
        image_rotary_emb = prepare_rotary_positional_embeddings(
            height=height * VAE_SCALE_FACTOR_SPATIAL, width=width * VAE_SCALE_FACTOR_SPATIAL, num_frames=num_frames,
            vae_scale_factor_spatial=VAE_SCALE_FACTOR_SPATIAL, patch_size=cfg.patch_size, patch_size_t=cfg.patch_size_t,
            attention_head_dim=cfg.attention_head_dim, device=video_latents.device)
    ofs_emb = b.image_latents.new_full((1,), fill_value=2.0) if is_ofs_embed else None  # :1037
    noisy = scheduler.add_noise(video_latents, noise, timesteps)                       # :1042
    model_in = torch.cat([noisy, b.image_latents], dim=2)
    depths = torch.cat([b.depth_latents, b.depth_latents], dim=2) if b.depth_latents is not None else None
    labels = torch.cat([b.label_latents, b.label_latents], dim=2) if b.label_latents is not None else None
    video_output, is_action_mask, actions_recon = transformer(
        hidden_states=model_in, encoder_hidden_states=b.prompt_embeds,
        controls_or_guidances=dict(actions=b.actions, depths=depths, labels=labels), timestep=timesteps, ofs=ofs_emb,
        image_rotary_emb=image_rotary_emb, return_dict=False, num_views=b.num_views)
    video_pred = scheduler.get_velocity(video_output, noisy, timesteps)                # :1068
    ac = scheduler.alphas_cumprod.to(device=video_latents.device, dtype=torch.float32)
    weights = 1 / (1 - ac[timesteps])
    while weights.ndim < video_pred.ndim:
        weights = weights.unsqueeze(-1)
    fm = b.frame_mask
    loss = torch.mean((weights * (video_pred[:, fm].float() - video_latents[:, fm].float()) ** 2).reshape(batch_size, -1),
                      dim=1).mean()
    parts = {"denoise": loss.detach()}
    if cfg.recon_action and actions_recon is not None and is_action_mask is not None:  # :1081-1090
        keep = ~is_action_mask
        # the reference guards with a host sync on keep.sum() > 0; an all-masked batch gives NaN means there, so the
        # same condition is evaluated here (one tiny device->host read per step, as in the reference)
        if bool(keep.any()):
            rot, pos, grip = type(transformer).compute_action_loss(b.actions, actions_recon, loss_weight=ACTION_LOSS_WEIGHT,
                                                                   mask=keep)
            loss = loss + (rot + pos + grip)
            parts.update(rot=rot.detach(), pos=pos.detach(), grip=grip.detach())
    return loss, parts


def sft_step(transformer, scheduler, optimizer, b: Batch, generator: Optional[torch.Generator] = None,
             use_rope: bool = False, is_ofs_embed: bool = False, data_parallel: bool = False,
             gradient_accumulation_steps: int = 1, micro_step: int = 0, lr_scheduler=None):
    """One micro-batch of the loop body (:863-1107).  ``optimizer`` is ``orv_amd.optim.FusedAdamW`` (global-norm clip inside,
    on device).  With ``data_parallel`` the flat bf16 gradient buffer is averaged over the RCCL group.

    Gradient accumulation follows ``accelerator.accumulate`` (:863; reference default 4 micro-batches,
    base_train.yaml ``gradient_accumulation_steps``): the loss is divided by ``gradient_accumulation_steps``
    (``accelerator.backward``), gradients add up in ``p.grad`` and NO collective runs on the first N-1 micro-batches (DDP
    ``no_sync``); the optimizer step, the ONE gradient exchange and ``zero_grad`` happen on the micro-batch that closes the
    window (``(micro_step + 1) % N == 0``).  The backward-overlapped exchange needs the gradients of one backward to be
    final when a block closes, so it is used only without accumulation; with accumulation the accumulated buffer is
    exchanged once after the last backward (same bytes, one step later start).  The learning-rate scaling is the caller's,
    as in the reference (:496-501 scales lr by accumulation x batch x processes when ``scale_lr``); ``lr_scheduler``
    (``orv_amd.optim.get_scheduler``: the reference's ``diffusers.optimization.get_scheduler`` call at :740-747, ``cosine_with_restarts``
    in every shipped config) is stepped once per call, behind the optimizer, as ``lr_scheduler.step()`` at :1107 is - accelerate's wrapper
    then skips it on non-synchronising micro-batches and steps it ``num_processes`` times on synchronising ones, which is why the script
    multiplies the warm-up / total counts by ``num_processes``: here it advances by ``world`` on synchronising micro-batches.  Returns
    ``(loss, parts)``; ``parts["grad_norm"]`` is present on synchronising micro-batches only."""
    dev = b.video_latents.device
    noise = torch.randn(b.video_latents.shape, device=dev, dtype=torch.float32, generator=generator).to(b.video_latents.dtype)
    timesteps = torch.randint(0, scheduler.config.num_train_timesteps, (b.video_latents.shape[0],), dtype=torch.int64,
                              device=dev, generator=generator)                          # :1013-1019
    loss, parts = sft_loss(transformer, scheduler, b, noise, timesteps, use_rope, is_ofs_embed)
    n_acc = max(1, int(gradient_accumulation_steps))
    sync = (micro_step + 1) % n_acc == 0
    world = 1
    if data_parallel:
        import torch.distributed as dist
        world = dist.get_world_size()
    if world > 1 and n_acc == 1:   # the exchange starts inside the backward, block by block (sharding.FlatGradReducer)
        transformer._dp_grad_hook = optimizer.begin_overlapped_allreduce()
    try:
        (loss / n_acc if n_acc > 1 else loss).backward()
    finally:
        transformer._dp_grad_hook = None
    if sync:
        parts["grad_norm"] = optimizer.step(average_over=world)       # (finishes) the exchange, averages, clips, updates
        optimizer.zero_grad()
        if lr_scheduler is not None:                                  # :1107 through accelerate's AcceleratedScheduler (split_batches False)
            for _ in range(world):
                lr_scheduler.step()
            parts["lr"] = lr_scheduler.get_last_lr()[0]
    return loss.detach(), parts
