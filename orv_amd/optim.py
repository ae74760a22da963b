"""Fused AdamW + global-norm clipping for the bf16-parameter training of the reference
(/root/reference/config/base_train.yaml:143-153: AdamW, betas 0.9/0.95, wd 1e-3, eps 1e-8, max_grad_norm 1.0;
/root/reference/orv/pipeline/train_cogvideox_control_to_video_sft.py:1095-1104: clip then step).  One ``orv_sumsq`` launch
per gradient accumulates the squared norm on the device, the clip coefficient stays on the device (no host sync in the
step), and ``orv_adamw`` applies clip + moment update + decoupled weight decay + bf16 write in one pass per parameter.
Moments are fp32 (the reference keeps them in the parameter dtype; 13.5 GB for 1.69 B parameters is affordable here)."""
from __future__ import annotations

from typing import Iterable

import torch

from . import ops


class FusedAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-3,
                 max_grad_norm: float = 1.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.state = {}
        self.step_count = 0
        self.param_groups = [{"lr": lr, "params": self.params}]      # lr schedulers poke param_groups[0]["lr"]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self) -> float:
        """Returns the pre-clip global gradient norm (one host read, after all kernels are queued)."""
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return 0.0
        dev = live[0].device
        ss = torch.zeros(1, dtype=torch.float32, device=dev)
        for p in live:
            ops.sumsq(p.grad.contiguous(), ss)
        norm = ss.sqrt()
        clip = torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0) if self.max_grad_norm else torch.ones_like(norm)
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        for p in live:
            st = self.state.get(id(p))
            if st is None:
                st = (torch.zeros(p.shape, dtype=torch.float32, device=dev), torch.zeros(p.shape, dtype=torch.float32, device=dev))
                self.state[id(p)] = st
            ops.adamw(p.data, p.grad.contiguous(), st[0], st[1], lr, self.betas[0], self.betas[1], self.eps,
                      self.weight_decay, self.step_count, clip)
        return float(norm.item())
