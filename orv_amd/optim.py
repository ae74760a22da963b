"""Fused AdamW + global-norm clipping for the bf16-parameter training of the reference
(/root/reference/config/base_train.yaml:143-153: AdamW, betas 0.9/0.95, wd 1e-3, eps 1e-8, max_grad_norm 1.0;
/root/reference/orv/pipeline/train_cogvideox_control_to_video_sft.py:1095-1104: clip then step).

MI355X layout: every trainable parameter becomes a view into ONE flat bf16 buffer (segments padded to 2048 elements), with
flat bf16 gradient and fp32 moment buffers beside it (1.69 B parameters: 3.4 + 3.4 + 6.8 + 6.8 GB of the 288 GB).  A step
is then: gather the autograd gradients into the flat buffer (multi-tensor copy) -> optional RCCL all-reduce (SUM; the 1 / world of the
average is folded into the clip coefficient) of the flat buffer in 256 MB pieces -> ONE ``orv_sumsq`` -> clip coefficient on the device (no host sync before the update) -> ONE
``orv_adamw_flat`` (clip + moments + decoupled decay + bf16 write, 16-byte accesses).  Parameters that received no gradient
in a step are skipped exactly as ``torch.optim.AdamW`` skips them (segment activity mask) and every parameter carries its
OWN step count for the bias correction (``state[p]["step"]`` in torch).  Under data parallel a parameter is active iff it has a gradient on ANY rank (the usage mask rides in the last segment of the flat gradient buffer, so it arrives with the gradient
exchange itself: DDP's ``find_unused_parameters`` bookkeeping, which the reference enables, base_train.yaml:181); a rank without one contributes zeros,
so the collective schedule and the update are identical on all ranks and equal to the single-GPU trajectory.  Moments
are fp32 (the reference keeps them in the parameter dtype)."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import _state, ops

_SEG = 2048          # elements per workgroup of orv_adamw_flat; every segment is padded to a multiple of it
_AR_CHUNK = 128 * 1024 * 1024   # bf16 elements per all-reduce call (256 MB: large enough that xGMI link bandwidth, not
#                                 launch latency, bounds the ring)


class FusedAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-3,
                 max_grad_norm: float = 1.0):
        ps = [p for p in params if p.requires_grad]
        # flat-buffer order: parameters the model tagged as "gradient final when its block's backward ends" first (model order: a block's
        # six weights are contiguous), everything else behind them - see CogVideoXTransformer3DModelTraj._set_trainable_parameters and
        # sharding.FlatGradReducer.  Untagged parameter lists (any other model) keep their order.
        self.params: List[torch.nn.Parameter] = ([p for p in ps if getattr(p, "_orv_grad_early", False)]
                                                 + [p for p in ps if not getattr(p, "_orv_grad_early", False)])
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.param_groups = [{"lr": lr, "params": self.params}]      # lr schedulers poke param_groups[0]["lr"]
        self._flat = None
        self._handed = set()      # segments handed to the backward for in-place writing since the last step() / zero_grad()
        self._dirty = set()       # segments that hold data (written in place or copied into) and were not zeroed since

    def _grad_segment(self, i: int) -> torch.Tensor:
        return self._flat["views_g"][i]

    # ---- flat storage ----
    def _build(self):
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.bfloat16 or p.device != dev:
                raise RuntimeError("FusedAdamW: parameters must be bf16 tensors on one GPU (call model.to(device, bfloat16) first)")
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += (p.numel() + _SEG - 1) // _SEG * _SEG
        total = o
        # the gradient buffer carries one more segment: the usage mask of this rank (1.0 per parameter with a gradient), so that
        # under data parallel "used on ANY rank" arrives with the gradient exchange itself (SUM > 0) instead of a second collective
        tail = (len(self.params) + _SEG - 1) // _SEG * _SEG
        flat_p = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        views_p, views_g = [], []
        flat_g = torch.zeros(total + tail, dtype=torch.bfloat16, device=dev)
        for p, off in zip(self.params, offs):
            v = flat_p[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v                                   # the module keeps its Parameter objects; only their storage moves
            views_p.append(v)
            views_g.append(flat_g[off:off + p.numel()].view(p.shape))
        self._flat = dict(
            p=flat_p, g=flat_g, g_params=flat_g[:total], g_mask=flat_g[total:total + len(self.params)],
            reduce_starts=offs + [total, total + tail], m=torch.zeros(total, dtype=torch.float32, device=dev),
            v=torch.zeros(total, dtype=torch.float32, device=dev), views_g=views_g,
            seg_start=torch.tensor(offs + [total], dtype=torch.int64, device=dev),
            active=torch.zeros(len(self.params), dtype=torch.uint8, device=dev), active_host=[False] * len(self.params),
            seg_step=torch.zeros(len(self.params), dtype=torch.int32, device=dev))
        _state.register_grad_views(self.params, self)
        _state.bump_weights_epoch()      # parameter storage moved (p.data = view): captured graphs / derived-weight caches are stale

    # ---- data parallel: exchange overlapped with the backward ----
    def begin_overlapped_allreduce(self):
        """Returns ``hook(params, grads)`` for ``training.backward``: called whenever the gradients of a group of parameters
        are final, it copies them into the flat buffer and lets a ``FlatGradReducer`` start summing every contiguous final
        run over the process group while the backward continues.  ``step(average_over=world)`` then only sends the rest."""
        from .sharding import FlatGradReducer
        if self._flat is None:
            self._build()
        f = self._flat
        index = {id(p): i for i, p in enumerate(self.params)}
        red = FlatGradReducer(f["g"], f["reduce_starts"])      # parameter segments + the usage-mask segment (sent by finish())
        filled = set()

        def hook(params, grads):
            # Which segments one call makes final is decided by the CODE PATH of the backward, not by the data: the big weights
            # of a block have their gradient here, its small fp32-accumulated ones (biases, LayerNorm affine) are converted
            # at the very end on every rank alike, and the parameters whose gradient may exist on one rank only (action
            # reconstruction head, mask embedding, control fuse) are in no block list - they travel in ``finish`` as zeros
            # where a rank has none.  So every rank issues the same collectives in the same order.
            idx, srcs, dsts = [], [], []
            for p in params:
                i, g = index.get(id(p)), grads.get(id(p))
                if i is None or g is None or i in filled:
                    continue
                idx.append(i)
                if g.data_ptr() != f["views_g"][i].data_ptr():      # else: the backward wrote it in place (_state.grad_view)
                    srcs.append(g); dsts.append(f["views_g"][i])
            if not idx:
                return
            if srcs:
                torch._foreach_copy_(dsts, srcs)
            filled.update(idx)
            red.ready(idx)

        self._overlap = (red, filled)
        return hook

    def averaged_grads(self):
        """{parameter: averaged gradient copy} after a data-parallel ``step()`` (the flat buffer holds the SUM over ranks, see ``step``)."""
        import torch.distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if self._flat is None:
            return {}
        return {p: (v.float() / world).to(v.dtype) for p, v in zip(self.params, self._flat["views_g"])}

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None
        # a segment written in place whose gradient is dropped here (skipped / overflow step) keeps its data: remember it so
        # the next step() zeroes it unless a new gradient overwrites it
        self._dirty |= self._handed
        self._handed.clear()

    @torch.no_grad()
    def step(self, average_over: Optional[int] = None) -> float:
        """One update.  ``average_over=world_size`` first averages the flat gradient buffer over the default process group
        (data parallel, one exchange per step).  NOTE: under data parallel the exchange is a SUM and 1 / world is folded
        into the clip coefficient and the returned norm, so after ``step()`` the ``p.grad`` views that alias the flat segments hold
        ``world x`` the mean gradient (also for segments this rank never produced); code that inspects ``p.grad`` afterwards
        (per-parameter norm logging) must divide by the world size - ``averaged_grads()`` returns such copies.  Returns the pre-clip global gradient norm (the only host read, issued after
        every kernel of the step is queued)."""
        if not self.params:
            return 0.0
        if self._flat is None:
            self._build()
        f = self._flat
        overlap = getattr(self, "_overlap", None)
        self._overlap = None
        distributed = overlap is not None or bool(average_over and average_over > 1)
        has_grad = [p.grad is not None for p in self.params]
        dev = f["p"].device
        if not distributed and not any(has_grad):
            return 0.0
        done = overlap[1] if overlap else ()
        self._dirty |= self._handed          # segments the backward wrote in place since the last step
        self._handed.clear()
        self._dirty.update(done)
        # A segment without a gradient this step must read as zeros for the global norm (and, under data parallel, for the
        # exchange): zero every one that holds data - a parameter that lost its gradient (zero_grad on a skipped step,
        # torch.autograd.grad, a branch not taken this step) - not only the active -> inactive transitions.
        stale = [i for i in self._dirty if not has_grad[i] and i not in done]
        if stale:
            torch._foreach_zero_([f["views_g"][i] for i in stale])
            self._dirty.difference_update(stale)
        pairs = [(p.grad, v) for i, (p, v, a) in enumerate(zip(self.params, f["views_g"], has_grad))
                 if a and i not in done and p.grad.data_ptr() != v.data_ptr()]     # in-place gradients need no copy
        srcs, dsts = [s_ for s_, _ in pairs], [d_ for _, d_ in pairs]
        if srcs:
            torch._foreach_copy_(dsts, srcs)
        self._dirty.update(i for i, h in enumerate(has_grad) if h)
        world = 1
        if distributed:
            # Data parallel: a parameter is updated iff it has a gradient on ANY rank (DDP's local_used_map with
            # find_unused_parameters, /root/reference/config/base_train.yaml:181): a rank without one contributes zeros, and
            # a parameter unused on every rank is skipped as torch.optim.AdamW skips grad-None parameters (no decay, no
            # moment decay, no step-count increment) - the same trajectory as the single-GPU run.  The mask rides in the last
            # segment of the gradient buffer (1.0 where this rank has a gradient; SUM over ranks > 0 <=> used somewhere), so
            # the step issues gradient collectives only, at the same points on every rank; it stays on the device.
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                world = dist.get_world_size()
            f["g_mask"].copy_(torch.tensor(has_grad, dtype=torch.bfloat16))
            if overlap is not None:
                overlap[0].finish(average=False)         # the rest of the buffer (mask segment included), wait
                self.last_exchange = overlap[0]          # bookkeeping for bench.py (collectives, bytes, exposed time): .stats()
            else:
                from .sharding import allreduce_flat_
                allreduce_flat_(f["g"], _AR_CHUNK, average=False)
            f["active"].copy_((f["g_mask"] > 0).to(torch.uint8))
            f["active_host"] = None                      # device-side mask: the host copy is unknown
            if world > 1:
                # the exchange wrote the summed gradient into EVERY segment, also those this rank has no gradient for: all of
                # them hold data now and must be zeroed before the next exchange / norm unless a new gradient overwrites them
                self._dirty.update(range(len(self.params)))
        elif has_grad != f["active_host"]:
            f["active"].copy_(torch.tensor(has_grad, dtype=torch.uint8))
            f["active_host"] = has_grad
        # The buffer holds the SUM over ranks; the 1 / world of the average is folded into the coefficient handed to the update
        # kernel (no separate pass over 3.4 GB): norm of the mean gradient = norm of the sum / world.
        ss = torch.zeros(1, dtype=torch.float32, device=dev)
        ops.sumsq(f["g_params"], ss)
        norm = ss.sqrt() / world
        clip = torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0) if self.max_grad_norm else torch.ones_like(norm)
        if world > 1:
            clip = clip / world
        self.step_count += 1
        f["seg_step"].add_(f["active"].to(torch.int32))          # per-parameter step counts (torch.optim.AdamW state["step"])
        ops.adamw_flat(f["p"], f["g"], f["m"], f["v"], f["seg_start"], f["active"], self.param_groups[0]["lr"], self.betas[0],
                       self.betas[1], self.eps, self.weight_decay, self.step_count, clip, seg_step=f["seg_step"])
        _state.bump_weights_epoch()      # parameters changed without a tensor._version bump: drop derived-weight caches
        return float(norm.item())

    # ---- checkpointing (torch.optim-like) ----
    def state_dict(self):
        """Flat moments plus the layout they are stored in (parameter element counts and segment offsets), so that a resume
        with a different trainable set fails loudly instead of mis-assigning moments."""
        layout = [int(p.numel()) for p in self.params]
        if self._flat is None:
            return {"step": self.step_count, "exp_avg": None, "exp_avg_sq": None, "numels": layout}
        return {"step": self.step_count, "exp_avg": self._flat["m"], "exp_avg_sq": self._flat["v"], "numels": layout,
                "seg_start": self._flat["seg_start"].tolist(), "seg_step": self._flat["seg_step"]}

    def load_state_dict(self, sd):
        layout = [int(p.numel()) for p in self.params]
        if sd.get("numels") is not None and list(sd["numels"]) != layout:
            if sorted(sd["numels"]) == sorted(layout):
                raise ValueError("FusedAdamW.load_state_dict: same parameters, another flat-buffer ORDER - the checkpoint predates round 6, which "
                                 "lays the block weights out first (one all-reduce per block); its moments cannot be mapped by position.  Resume "
                                 "the parameters from the model checkpoint and restart the moments, or load with the build that wrote it")
            raise ValueError("FusedAdamW.load_state_dict: the checkpoint was written for a different set of trainable parameters "
                             f"({len(sd['numels'])} segments vs {len(layout)} here, or different sizes)")
        self.step_count = int(sd["step"])
        if sd.get("exp_avg") is not None:
            if self._flat is None:
                self._build()
            if sd["exp_avg"].numel() != self._flat["m"].numel():
                raise ValueError("FusedAdamW.load_state_dict: moment buffers do not match the flat layout")
            self._flat["m"].copy_(sd["exp_avg"])
            self._flat["v"].copy_(sd["exp_avg_sq"])
            if sd.get("seg_step") is not None:
                self._flat["seg_step"].copy_(sd["seg_step"])
            else:                           # checkpoints written before per-parameter counts existed
                self._flat["seg_step"].fill_(self.step_count)


# ---- learning-rate schedules of the train tail (train_cogvideox_control_to_video_sft.py:740-747, :1107; base_train.yaml:160-164) ----
# The reference calls ``diffusers.optimization.get_scheduler(name, optimizer=, num_warmup_steps=, num_training_steps=, num_cycles=, power=)``
# and ``lr_scheduler.step()`` after every optimizer step.  diffusers is not installable here: the multipliers below restate its published
# ``get_*_schedule*`` lambdas (LEAF, unpinned like oracle/leaf.py; hand-derived known answers in tests/test_lr_schedule.py).  Each returns
# lr(step) / lr_base for the number of ``step()`` calls made so far.
def _lr_lambda(name: str, num_warmup_steps: int, num_training_steps: Optional[int], num_cycles, power: float, lr_init: float):
    import math
    W = int(num_warmup_steps or 0)
    T = num_training_steps
    name = getattr(name, "value", name)
    if name not in ("constant", "constant_with_warmup") and T is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")       # diffusers' message
    warm = lambda s: float(s) / float(max(1, W))
    if name == "constant":
        return lambda s: 1.0
    if name == "constant_with_warmup":
        return lambda s: warm(s) if s < W else 1.0
    if name == "linear":
        return lambda s: warm(s) if s < W else max(0.0, float(T - s) / float(max(1, T - W)))
    if name == "cosine":
        nc = 0.5 if num_cycles is None else float(num_cycles)      # NB the reference passes lr_num_cycles (1) for every schedule name

        def cosine(s):
            if s < W:
                return warm(s)
            prog = float(s - W) / float(max(1, T - W))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * nc * 2.0 * prog)))
        return cosine
    if name == "cosine_with_restarts":                             # base_train.yaml:161 - the schedule every shipped config trains with
        nc = 1 if num_cycles is None else num_cycles

        def restarts(s):
            if s < W:
                return warm(s)
            prog = float(s - W) / float(max(1, T - W))
            if prog >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(nc) * prog) % 1.0))))
        return restarts
    if name == "polynomial":
        lr_end = 1e-7
        if not lr_init > lr_end:
            raise ValueError(f"lr_end ({lr_end}) must be smaller than initial lr ({lr_init})")

        def poly(s):
            if s < W:
                return warm(s)
            if s > T:
                return lr_end / lr_init
            pct = 1 - (s - W) / (T - W)
            return ((lr_init - lr_end) * pct ** power + lr_end) / lr_init
        return poly
    raise ValueError(f"unknown lr scheduler {name!r} (constant, constant_with_warmup, linear, cosine, cosine_with_restarts, polynomial)")


class LambdaSchedule:
    """``torch.optim.lr_scheduler.LambdaLR`` semantics on anything with ``param_groups`` (``FusedAdamW`` is not a ``torch.optim.Optimizer``):
    lr = base_lr x lambda(number of ``step()`` calls), applied at construction for step 0 - so with a warm-up the first optimizer
    step runs at lr 0, exactly as the reference's loop does (train...sft.py:1100-1107: ``optimizer.step()`` then ``lr_scheduler.step()``)."""

    def __init__(self, optimizer, lr_lambda, last_epoch: int = -1):
        self.optimizer, self.lr_lambda = optimizer, lr_lambda
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def step(self):
        self.last_epoch += 1
        self._last_lr = [b * self.lr_lambda(self.last_epoch) for b in self.base_lrs]
        for g, lr in zip(self.optimizer.param_groups, self._last_lr):
            g["lr"] = lr

    def get_last_lr(self):
        return list(self._last_lr)

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs), "_last_lr": list(self._last_lr)}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lrs = int(sd["last_epoch"]), list(sd["base_lrs"])
        self._last_lr = [b * self.lr_lambda(self.last_epoch) for b in self.base_lrs]
        for g, lr in zip(self.optimizer.param_groups, self._last_lr):
            g["lr"] = lr


def get_scheduler(name, optimizer, step_rules=None, num_warmup_steps: Optional[int] = None, num_training_steps: Optional[int] = None,
                  num_cycles: int = 1, power: float = 1.0, last_epoch: int = -1) -> LambdaSchedule:
    """Drop-in for ``diffusers.optimization.get_scheduler`` as the train script calls it (:740-747) - same argument names and defaults;
    ``piecewise_constant`` (``step_rules``) is not used by any shipped config and is refused."""
    if step_rules is not None or getattr(name, "value", name) == "piecewise_constant":
        raise ValueError("piecewise_constant / step_rules is not provided (no ORV config uses it)")
    if getattr(name, "value", name) != "constant" and num_warmup_steps is None:
        raise ValueError(f"{name} requires `num_warmup_steps`, please provide that argument.")
    lr0 = optimizer.param_groups[0].get("initial_lr", optimizer.param_groups[0]["lr"])
    return LambdaSchedule(optimizer, _lr_lambda(name, num_warmup_steps or 0, num_training_steps, num_cycles, power, lr0), last_epoch)
