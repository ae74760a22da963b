"""CogVideoX DDIM / DPM-Solver++(2M, SDE) schedulers for the MI355X denoise loop.

Same constructor kwargs, ``set_timesteps`` / ``step`` / ``add_noise`` / ``get_velocity`` / ``scale_model_input`` /
``from_config`` surface as diffusers' ``CogVideoXDDIMScheduler`` / ``CogVideoXDPMScheduler``, which the reference uses at
/root/reference/orv/models/cogvideox_control.py:1304,1446-1457, /root/reference/orv/pipeline/inference_control_to_video.py:91
and /root/reference/orv/pipeline/train_cogvideox_control_to_video_sft.py:341,842,1042,1067.

Split of work: the noise schedule and the per-step scalar coefficients are host float64 arithmetic (as in the
reference); the per-element update runs in ONE fused HIP kernel (``orv_sched_step``: CFG combine + x0 prediction +
multistep combine + noise add + bf16 cast) instead of ~8 elementwise torch launches and an fp32 round trip.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import ops


class SchedulerConfig(dict):
    __getattr__ = dict.__getitem__


def alphas_cumprod_table(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                         snr_shift_scale=3.0, rescale_betas_zero_snr=False) -> np.ndarray:
    if beta_schedule == "scaled_linear":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float64)
    else:
        raise NotImplementedError(f"{beta_schedule} is not implemented")
    ac = np.cumprod(1.0 - betas)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)            # SNR shift
    if rescale_betas_zero_snr:                                          # zero terminal SNR
        s = np.sqrt(ac)
        s0, sT = s[0], s[-1]
        ac = ((s - sT) * (s0 / (s0 - sT))) ** 2
    return ac


class _CogVideoXScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.0120,
                 beta_schedule: str = "scaled_linear", trained_betas=None, clip_sample: bool = True,
                 set_alpha_to_one: bool = True, steps_offset: int = 0, prediction_type: str = "epsilon",
                 clip_sample_range: float = 1.0, sample_max_value: float = 1.0, timestep_spacing: str = "leading",
                 rescale_betas_zero_snr: bool = False, snr_shift_scale: float = 3.0, **unused):
        if prediction_type != "v_prediction":
            raise NotImplementedError("the fused step kernel implements v_prediction (every CogVideoX config)")
        self.config = SchedulerConfig(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
            set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type,
            clip_sample_range=clip_sample_range, sample_max_value=sample_max_value,
            timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr,
            snr_shift_scale=snr_shift_scale)
        self._ac = alphas_cumprod_table(num_train_timesteps, beta_start, beta_end, beta_schedule, snr_shift_scale,
                                        rescale_betas_zero_snr)
        self.alphas_cumprod = torch.from_numpy(self._ac.copy())
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self._ac[0])
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {n}")
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
            raise ValueError(f"{sp} is not supported. Choose one of 'leading', 'trailing' or 'linspace'.")
        self.timesteps = torch.from_numpy(ts).to(device)

    def _alphas(self, timestep, timestep_back=None):
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self._ac[t])
        a_p = float(self._ac[prev]) if prev >= 0 else self.final_alpha_cumprod
        a_b = None if timestep_back is None else float(self._ac[int(timestep_back)])
        return prev, a_t, a_p, a_b

    # --- training-side helpers (elementwise on [B,...]; the reference calls them once per micro-batch) ---
    def _coef(self, like, timesteps):
        a = self.alphas_cumprod.to(device=like.device, dtype=like.dtype)[timesteps.to(like.device)]
        sa, sb = (a ** 0.5).flatten(), ((1 - a) ** 0.5).flatten()
        while sa.ndim < like.ndim:
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa, sb

    def add_noise(self, original_samples, noise, timesteps):
        sa, sb = self._coef(original_samples, timesteps)
        return sa * original_samples + sb * noise

    def get_velocity(self, sample, noise, timesteps):
        sa, sb = self._coef(sample, timesteps)
        return sa * noise - sb * sample


class CogVideoXDDIMScheduler(_CogVideoXScheduler):
    def step_coefficients(self, timestep):
        """(sa, sb, cx, cd): x0 = sa*x - sb*v ; x_prev = cx*x + cd*x0."""
        _, a_t, a_p, _ = self._alphas(timestep)
        A = math.sqrt((1 - a_p) / (1 - a_t))
        return math.sqrt(a_t), math.sqrt(1 - a_t), A, math.sqrt(a_p) - math.sqrt(a_t) * A

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True, model_output_uncond=None,
             guidance_scale: float = 1.0):
        if eta != 0.0:
            raise NotImplementedError("eta > 0 is never used on the ORV path")
        sa, sb, cx, cd = self.step_coefficients(timestep)
        prev, x0 = ops.sched_step(sample, model_output, model_output_uncond, guidance_scale, None, None, sa, sb, 1.0,
                                  0.0, cx, cd, 0.0)
        return (prev, x0) if not return_dict else SchedulerConfig(prev_sample=prev, pred_original_sample=x0)


def _log(x):
    return -math.inf if x == 0 else math.log(x)


class CogVideoXDPMScheduler(_CogVideoXScheduler):
    def step_coefficients(self, timestep, timestep_back):
        """(sa, sb, m1, m2, m_noise, m3, m4, second_order) in float64, inf-safe at both ends of the schedule."""
        prev, a_t, a_p, a_b = self._alphas(timestep, timestep_back)
        lamb = 0.5 * (_log(a_t) - _log(1 - a_t))
        lamb_next = math.inf if a_p >= 1.0 else 0.5 * (_log(a_p) - _log(1 - a_p))
        h = lamb_next - lamb
        e_h = math.exp(-h) if h != math.inf else 0.0
        e_2h = math.exp(-2 * h) if h != math.inf else 0.0
        m1 = math.sqrt((1 - a_p) / (1 - a_t)) * e_h
        m2 = (e_2h - 1.0 if h == math.inf else math.expm1(-2 * h)) * math.sqrt(a_p)
        mn = math.sqrt(1 - a_p) * math.sqrt(1 - e_2h)
        m3, m4 = 1.0, 0.0
        if a_b is not None:
            lamb_prev = 0.5 * (_log(a_b) - _log(1 - a_b))
            h_last = lamb - lamb_prev
            if h == math.inf:      # last step: prev_timestep < 0, the multistep branch is skipped anyway
                r = 0.0 if h_last != math.inf else 1.0
            else:
                r = h_last / h
            if r == math.inf:
                m3, m4 = 1.0, 0.0
            elif r != 0.0:
                m3, m4 = 1 + 1 / (2 * r), 1 / (2 * r)
        return math.sqrt(a_t), math.sqrt(1 - a_t), m1, m2, mn, m3, m4, prev

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise=None, return_dict: bool = False,
             model_output_uncond=None, guidance_scale: float = 1.0):
        sa, sb, m1, m2, mn, m3, m4, prev = self.step_coefficients(timestep, timestep_back)
        # noise is drawn exactly like the reference (randn_tensor: CPU generator -> CPU draw in the sample dtype -> copy)
        noise = _randn_like(sample, generator)
        second = not (old_pred_original_sample is None or prev < 0)
        if second:
            noise = _randn_like(sample, generator)   # the reference draws twice and uses the second draw
        x_prev, x0 = ops.sched_step(sample, model_output, model_output_uncond, guidance_scale,
                                    old_pred_original_sample if second else None, noise, sa, sb, m3, m4, m1, -m2, mn)
        return x_prev, x0


def _randn_like(sample, generator):
    dev = sample.device
    if generator is not None and generator.device.type == "cpu" and dev.type != "cpu":
        n = torch.randn(sample.shape, generator=generator, device="cpu", dtype=sample.dtype).to(dev)
    else:
        n = torch.randn(sample.shape, generator=generator, device=dev, dtype=sample.dtype)
    return n.to(torch.float32)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, **kwargs):
    if timesteps is not None:
        raise ValueError("custom timesteps are not supported by the CogVideoX schedulers")
    scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, num_inference_steps
