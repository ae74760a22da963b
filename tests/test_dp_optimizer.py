"""CPU, world_size 2 over gloo: FusedAdamW's data-parallel bookkeeping.

The two HIP kernels of the step (orv_sumsq, orv_adamw_flat_steps) cannot run here, so THIS TEST replaces them with torch
stand-ins (test infrastructure, never shipped); everything else - flat buffer, stale-segment zeroing, usage mask in the last
segment of the gradient buffer, SUM exchange with 1 / world folded into the clip coefficient, overlapped reducer - is the
product code.  Scenario: a parameter whose gradient exists on one rank only, on neither rank, and again on one rank, over
consecutive steps (the action-reconstruction head / mask embedding of
/root/reference/orv/pipeline/train_cogvideox_control_to_video_sft.py:1093 under find_unused_parameters).  The two-rank run
must follow the single-process trajectory on the rank-averaged gradients, and both ranks must issue identical collectives."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _install_standins():
    from orv_amd import ops

    def sumsq(g, out):
        out.add_(g.float().pow(2).sum())

    def adamw_flat(p, g, m, v, seg_start, seg_active, lr, beta1, beta2, eps, weight_decay, step, clip_coef=None, seg_step=None):
        starts = seg_start.tolist()
        c = float(clip_coef) if clip_coef is not None else 1.0
        for i in range(seg_active.numel()):
            if not int(seg_active[i]):
                continue
            a, b = starts[i], starts[i + 1]
            t = int(seg_step[i]) if seg_step is not None else step
            gi = g[a:b].float() * c
            m[a:b].mul_(beta1).add_(gi, alpha=1 - beta1)
            v[a:b].mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
            mh, vh = m[a:b] / (1 - beta1 ** t), v[a:b] / (1 - beta2 ** t)
            p[a:b].copy_((p[a:b].float() * (1 - lr * weight_decay) - lr * mh / (vh.sqrt() + eps)).to(p.dtype))

    ops.sumsq, ops.adamw_flat = sumsq, adamw_flat


# which rank has a gradient for which parameter at which step: A always both, B rank 0 only / nobody / rank 0 only / nobody,
# C rank 0 only at every step (rank 1 NEVER has one: the stale-segment case of ADVICE r3), D nobody ever
_USAGE = {
    "A": [(1, 1)] * 4,
    "B": [(1, 0), (0, 0), (1, 0), (0, 0)],
    "C": [(1, 0)] * 4,
    "D": [(0, 0)] * 4,
}
_SHAPES = {"A": (64, 40), "B": (3000,), "C": (17, 9), "D": (5,)}


def _grad(name, step, rank):
    g = torch.Generator().manual_seed(1000 * step + 10 * rank + sum(map(ord, name)))
    return (torch.randn(_SHAPES[name], generator=g) * 0.3).to(torch.bfloat16)


def _make_params():
    g = torch.Generator().manual_seed(7)
    return {n: torch.nn.Parameter(torch.randn(s, generator=g).to(torch.bfloat16)) for n, s in _SHAPES.items()}


def _run_steps(rank, world, overlapped, log):
    from orv_amd.optim import FusedAdamW
    params = _make_params()
    opt = FusedAdamW(params.values(), lr=1e-2, max_grad_norm=0.5)
    norms = []
    for step in range(4):
        hook = opt.begin_overlapped_allreduce() if overlapped else None
        grads = {}
        for n, p in params.items():
            if _USAGE[n][step][rank]:
                p.grad = _grad(n, step, rank)
                grads[id(p)] = p.grad
        if hook is not None:
            hook([params["A"]], grads)        # the block parameters become final first; the rest travels in finish()
        log.append(("step", step))
        norms.append(opt.step(average_over=world))
        opt.zero_grad()
    return {n: p.detach().float().clone() for n, p in params.items()}, norms


def _worker(rank, world, port, overlapped, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_standins()
    log = []
    real = dist.all_reduce

    def logged(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        log.append((int(t.numel()), str(op), str(t.dtype)))
        return real(t, op=op, group=group, async_op=async_op)

    dist.all_reduce = logged
    final, norms = _run_steps(rank, world, overlapped, log)
    out.put((rank, {n: v.numpy().copy() for n, v in final.items()}, norms, log))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(overlapped):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlapped, q)) for r in range(2)]
    [p.start() for p in procs]
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return got


def _reference_trajectory():
    """Single process, hand-rolled: gradient = (g_rank0 + g_rank1) / 2 with zeros for a rank without one, parameter skipped
    (no decay, no moment update, no step count) when no rank has a gradient; global-norm clip over the averaged gradients."""
    params = {n: p.detach().float().clone() for n, p in _make_params().items()}
    m = {n: torch.zeros(s) for n, s in _SHAPES.items()}
    v = {n: torch.zeros(s) for n, s in _SHAPES.items()}
    t = {n: 0 for n in _SHAPES}
    lr, b1, b2, eps, wd, max_norm = 1e-2, 0.9, 0.95, 1e-8, 1e-3, 0.5
    norms = []
    for step in range(4):
        gsum = {}
        for n in _SHAPES:
            u = _USAGE[n][step]
            if any(u):
                # the bf16 buffer after the SUM exchange
                gsum[n] = sum((_grad(n, step, r).float() if u[r] else torch.zeros(_SHAPES[n])) for r in range(2)).to(torch.bfloat16).float()
        norm = torch.sqrt(sum((g ** 2).sum() for g in gsum.values())) / 2
        clip = min(1.0, max_norm / (float(norm) + 1e-6))
        norms.append(float(norm))
        for n, g in gsum.items():
            gi = g * (clip / 2)
            t[n] += 1
            m[n] = b1 * m[n] + (1 - b1) * gi
            v[n] = b2 * v[n] + (1 - b2) * gi * gi
            mh, vh = m[n] / (1 - b1 ** t[n]), v[n] / (1 - b2 ** t[n])
            params[n] = (params[n] * (1 - lr * wd) - lr * mh / (vh.sqrt() + eps)).to(torch.bfloat16).float()
    return params, norms


def _check(overlapped):
    got = _run_two_ranks(overlapped)
    ref, ref_norms = _reference_trajectory()
    (_, p0, n0, log0), (_, p1, n1, log1) = got
    # identical collectives, in the same order, on both ranks - and gradient SUMs only (no separate mask collective)
    assert log0 == log1
    colls = [e for e in log0 if e[0] != "step"]
    assert colls and all("SUM" in op and "bfloat16" in dt for _, op, dt in colls)
    per_step = [sum(1 for e in log0[i:] if e[0] != "step") for i, e in enumerate(log0) if e[0] == "step"]
    assert len(set(a - b for a, b in zip(per_step, per_step[1:] + [0]))) == 1      # same number of collectives every step
    for n in _SHAPES:
        a, b = torch.from_numpy(p0[n]), torch.from_numpy(p1[n])
        assert torch.equal(a, b), f"ranks diverged on {n}"
        assert torch.allclose(a, ref[n], rtol=0, atol=2e-2 * float(ref[n].abs().max())), f"{n} left the single-process trajectory"
        # bf16-exact except where fp32 summation order differs by an ulp
        assert (a - ref[n]).abs().max() <= 2 ** -7 * float(ref[n].abs().max())
    assert n0 == n1
    assert all(abs(x - y) <= 1e-3 * max(1.0, y) for x, y in zip(n0, ref_norms))
    # D never had a gradient anywhere: untouched
    assert torch.equal(torch.from_numpy(p0["D"]), _make_params()["D"].detach().float())


def test_two_rank_rank_divergent_usage_follows_single_process_trajectory():
    _check(overlapped=False)


def test_two_rank_rank_divergent_usage_overlapped_reducer():
    _check(overlapped=True)


# ---- gradient accumulation: `accelerator.accumulate` windows of 2 micro-batches (train_cogvideox_control_to_video_sft.py:863,
# base_train.yaml gradient_accumulation_steps) with a parameter that is UNUSED in part of the window / on one rank / in the whole window
# (find_unused_parameters, base_train.yaml:181).  Gradients add up in p.grad (bf16, as autograd accumulates), NO collective runs on the
# micro-batch that does not close the window (DDP no_sync), the window's closing micro-batch exchanges the accumulated buffer once.
# usage[name][window][micro] = (rank 0 has a gradient, rank 1 has a gradient)
_ACC_USAGE = {
    "A": [[(1, 1), (1, 1)]] * 3,
    "B": [[(1, 0), (0, 0)], [(0, 0), (0, 1)], [(0, 0), (0, 0)]],      # first micro-batch only / second only on the other rank / whole window unused
    "C": [[(0, 0), (1, 0)], [(1, 0), (1, 0)], [(1, 0), (0, 0)]],      # rank 1 never
    "D": [[(0, 0), (0, 0)]] * 3,
}


def _acc_grad(name, window, micro, rank):
    return _grad(name, 10 * window + micro + 1, rank)


def _run_accum(rank, world, log):
    from orv_amd.optim import FusedAdamW
    params = _make_params()
    opt = FusedAdamW(params.values(), lr=1e-2, max_grad_norm=0.5)
    norms, n_acc = [], 2
    for window in range(3):
        for micro in range(n_acc):
            for n, p in params.items():
                if _ACC_USAGE[n][window][micro][rank]:
                    g = (_acc_grad(n, window, micro, rank).float() / n_acc).to(torch.bfloat16)      # accelerator.backward(loss / n_acc)
                    p.grad = g if p.grad is None else (p.grad + g)                                   # autograd accumulation, bf16
            log.append(("micro", window, micro))
            if micro == n_acc - 1:                                                                   # sft_step: sync micro-batch
                norms.append(opt.step(average_over=world))
                opt.zero_grad()
    return {n: p.detach().float().clone() for n, p in params.items()}, norms


def _accum_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_standins()
    log = []
    real = dist.all_reduce

    def logged(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        log.append((int(t.numel()), str(op), str(t.dtype)))
        return real(t, op=op, group=group, async_op=async_op)

    dist.all_reduce = logged
    final, norms = _run_accum(rank, world, log)
    out.put((rank, {n: v.numpy().copy() for n, v in final.items()}, norms, log))
    dist.barrier()
    dist.destroy_process_group()


def _accum_reference():
    params = {n: p.detach().float().clone() for n, p in _make_params().items()}
    m = {n: torch.zeros(s) for n, s in _SHAPES.items()}
    v = {n: torch.zeros(s) for n, s in _SHAPES.items()}
    t = {n: 0 for n in _SHAPES}
    lr, b1, b2, eps, wd, max_norm = 1e-2, 0.9, 0.95, 1e-8, 1e-3, 0.5
    norms = []
    for window in range(3):
        gsum = {}
        for n in _SHAPES:
            per_rank = []
            for r in range(2):
                acc = None
                for micro in range(2):
                    if _ACC_USAGE[n][window][micro][r]:
                        g = (_acc_grad(n, window, micro, r).float() / 2).to(torch.bfloat16)
                        acc = g if acc is None else (acc + g)
                per_rank.append(acc)
            if any(a is not None for a in per_rank):
                gsum[n] = sum((a.float() if a is not None else torch.zeros(_SHAPES[n])) for a in per_rank).to(torch.bfloat16).float()
        norm = torch.sqrt(sum((g ** 2).sum() for g in gsum.values())) / 2
        clip = min(1.0, max_norm / (float(norm) + 1e-6))
        norms.append(float(norm))
        for n, g in gsum.items():
            gi = g * (clip / 2)
            t[n] += 1
            m[n] = b1 * m[n] + (1 - b1) * gi
            v[n] = b2 * v[n] + (1 - b2) * gi * gi
            mh, vh = m[n] / (1 - b1 ** t[n]), v[n] / (1 - b2 ** t[n])
            params[n] = (params[n] * (1 - lr * wd) - lr * mh / (vh.sqrt() + eps)).to(torch.bfloat16).float()
    return params, norms, t


def test_two_rank_gradient_accumulation_with_unused_parameters_across_the_window():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, p0, n0, log0), (_, p1, n1, log1) = got
    assert log0 == log1                                    # identical collective sequence on both ranks
    # no collective between a window's first micro-batch marker and its second (no_sync); every closing micro-batch issues the same number
    counts, cur = {}, None
    for e in log0:
        if e[0] == "micro":
            cur = (e[1], e[2]); counts[cur] = 0
        else:
            assert "SUM" in e[1] and "bfloat16" in e[2]
            counts[cur] += 1
    assert all(counts[(w, 0)] == 0 for w in range(3)), counts
    assert len({counts[(w, 1)] for w in range(3)}) == 1 and counts[(0, 1)] >= 1, counts
    ref, ref_norms, steps = _accum_reference()
    assert steps == {"A": 3, "B": 2, "C": 3, "D": 0}       # B skipped in the window nobody used it (no decay, no moment decay, no step count)
    for n in _SHAPES:
        a, b = torch.from_numpy(p0[n]), torch.from_numpy(p1[n])
        assert torch.equal(a, b), f"ranks diverged on {n}"
        assert (a - ref[n]).abs().max() <= 2 ** -7 * float(ref[n].abs().max()), f"{n} left the single-process trajectory"
    assert n0 == n1 and all(abs(x - y) <= 1e-3 * max(1.0, y) for x, y in zip(n0, ref_norms))
    assert torch.equal(torch.from_numpy(p0["D"]), _make_params()["D"].detach().float())
