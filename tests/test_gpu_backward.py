"""GPU parity of the backward (training) kernels against torch autograd on an fp32 CPU restatement of the same forward op.
Tolerance: bf16 gradients vs fp32 reference |err| <= 2e-2 |ref| + 1.5e-2 max|ref| (gradients are sums of bf16-rounded terms)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
LOG2E = 1.4426950408889634


def _dev():
    return torch.device("cuda:0")


def q(x):
    return x.to(BF).float()


def close(got, ref, rtol=2e-2, afrac=1.5e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    atol = afrac * ref.abs().max().item() + 1e-6
    bad = (got - ref).abs() > (rtol * ref.abs() + atol)
    assert not bad.any(), f"max err {(got - ref).abs().max().item():.4g} vs atol {atol:.4g} ({int(bad.sum())} bad of {bad.numel()})"


def test_transpose_and_colsum():
    from orv_amd import ops
    dev = _dev()
    for R, C in [(100, 64), (3226, 1920), (777, 192), (200, 72)]:
        x = q(torch.randn(R, C))
        t = ops.transpose(x.to(dev, BF), R, C)
        ld = (R + 127) // 128 * 128                     # round 6: the contraction index is padded to the t8 kernels' K % 128 == 0
        assert t.shape == (C, ld)
        t64 = ops.transpose(x.to(dev, BF), R, C, ld_dst=(R + 63) // 64 * 64)      # an explicit stride is honoured
        assert t64.shape == (C, (R + 63) // 64 * 64) and torch.equal(t64[:, :R], t[:, :R])
        if R % 8 == 0:                                   # out= given: its row stride is the destination stride (the source needs columns % 8)
            back = torch.full((R, (C + 63) // 64 * 64 + 64), 7.0, dtype=BF, device=dev)
            ops.transpose(t[:, :R].contiguous(), C, R, out=back)
            assert torch.equal(back[:, :C].float().cpu(), x) and torch.all(back[:, C:] == 0)
        assert torch.equal(t[:, :R].float().cpu(), x.t()) and torch.all(t[:, R:] == 0)
        out = torch.zeros(C, dtype=torch.float32, device=dev)
        ops.colsum(x.to(dev, BF), out, R, C)
        close(out, x.sum(0), rtol=1e-3, afrac=1e-3)
        # the fused form (orv_transpose_colsum_bf16): the same transpose bit for bit, the sums accumulated INTO the buffer
        acc = torch.full((C,), 2.0, dtype=torch.float32, device=dev)
        t2 = ops.transpose(x.to(dev, BF), R, C, colsum=acc)
        assert torch.equal(t2, t)
        close(acc - 2.0, x.sum(0), rtol=1e-3, afrac=1e-3)
    # a strided source (a column block of a wider matrix), as the packed q | k | v gradient is read
    x = q(torch.randn(500, 256))
    xd = x.to(dev, BF)
    acc = torch.zeros(128, dtype=torch.float32, device=dev)
    t3 = ops.transpose(xd[:, 64:192], 500, 128, ld_src=256, colsum=acc)
    assert torch.equal(t3[:, :500].float().cpu(), x[:, 64:192].t()) and torch.all(t3[:, 500:] == 0)
    close(acc, x[:, 64:192].sum(0), rtol=1e-3, afrac=1e-3)


def test_scatter_f32_to_bf16_segments():
    """orv_scatter_f32_to_bf16: fp32 arena segments -> bf16 destinations of different lengths, one launch; untouched neighbours."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    arena = torch.randn(5000, generator=g).to(dev)
    flat = torch.full((4096,), 7.0, dtype=BF, device=dev)
    segs = [(64, flat[10:74]), (1000, flat[128:128 + 1920].view(30, 64)), (3000, flat[3000:3001]), (3500, flat[3200:3200 + 300])]
    for _ in range(2):                                   # the second call takes the cached tables
        ops.scatter_f32_to_bf16(arena, segs)
    ref = torch.full((4096,), 7.0)
    for off, d in segs:
        start = (d.data_ptr() - flat.data_ptr()) // 2
        ref[start:start + d.numel()] = arena[off:off + d.numel()].cpu()
    assert torch.equal(flat.float().cpu(), ref.to(BF).float())
    with pytest.raises(ValueError):
        ops.scatter_f32_to_bf16(arena, [(4990, flat[:64])])       # source range past the arena


def test_gemm_dgrad_wgrad_via_transposes():
    """dX = dY W and dW = dY^T X through the NT GEMM + orv_transpose_bf16; GELU adjoint fused as epilogue 3."""
    from orv_amd import ops
    dev = _dev()
    M, N, K = 1000, 384, 256
    g = torch.Generator().manual_seed(1)
    X, W, dY = q(torch.randn(M, K, generator=g)), q(torch.randn(N, K, generator=g) * 0.1), q(torch.randn(M, N, generator=g))
    U = q(torch.randn(M, K, generator=g))
    dX_ref = (dY @ W) * torch.autograd.functional.jacobian(lambda u: torch.nn.functional.gelu(u, approximate="tanh").sum(), U)
    dW_ref = dY.t() @ X
    dYd, Xd, Wd = dY.to(dev, BF), X.to(dev, BF), W.to(dev, BF)
    WT = ops.transpose(Wd, N, K)                          # [K, N_pad]
    dX = torch.empty(M, K, dtype=BF, device=dev)
    npad = WT.shape[1]
    dYp = torch.nn.functional.pad(dYd, (0, npad - N)) if npad != N else dYd
    ops.gemm(dYp.contiguous(), WT, None, dX, M, K, npad, epilogue=3, R=U.to(dev, BF), ldr=K)
    close(dX, dX_ref)
    dYT, XT = ops.transpose(dYd, M, N), ops.transpose(Xd, M, K)      # [N, M_pad], [K, M_pad]
    dW = q(torch.randn(N, K, generator=g)).to(dev, BF)               # pre-existing gradient: accumulate
    dW0 = dW.float().cpu().clone()
    ops.gemm(dYT, XT, None, dW, N, K, dYT.shape[1], epilogue=2, R=dW, ldr=K)
    close(dW, dW0 + dW_ref)


@pytest.mark.parametrize("D", [128, 1920])
def test_layernorm_modulate_and_gated_residual_backward(D):
    from orv_amd import ops
    dev = _dev()
    B, S, nt, P = 2, 75, 7, 17
    G = 1 + (S - nt) // P
    g = torch.Generator().manual_seed(D)
    x = q(torch.randn(B * S, D, generator=g) * 2 + 0.3).requires_grad_()
    gamma, beta = q(torch.randn(D, generator=g)).requires_grad_(), q(torch.randn(D, generator=g)).requires_grad_()
    mod = torch.randn(B, G, 2 * D, generator=g).requires_grad_()
    rows = torch.arange(B * S)
    s = rows % S
    grp = torch.where(s < nt, torch.zeros_like(s), 1 + (s - nt) // P)
    y = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5) * (1 + mod[rows // S, grp, D:]) + mod[rows // S, grp, :D]
    dy, dres = q(torch.randn(B * S, D, generator=g)), q(torch.randn(B * S, D, generator=g))
    y.backward(dy)
    md = mod.detach().to(dev)
    dmod = torch.zeros(B, G, 2 * D, dtype=torch.float32, device=dev)
    dgam, dbet = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx = torch.empty(B * S, D, dtype=BF, device=dev)
    ops.layernorm_modulate_bwd(dy.to(dev, BF), x.detach().to(dev, BF), dres.to(dev, BF), dx, gamma.detach().to(dev, BF),
                               beta.detach().to(dev, BF), md[..., D:], dmod[..., D:], dmod[..., :D], dgam, dbet, G * 2 * D,
                               2 * D, ops.groups(S, nt, P), B, D, 1e-5)
    close(dx, x.grad + dres)
    close(dmod, mod.grad)
    close(dgam, gamma.grad)
    close(dbet, beta.grad)
    # gated residual: out = x + gate * y
    yb = q(torch.randn(B * S, D, generator=g))
    gate = torch.randn(B, G, D, generator=g).requires_grad_()
    out = (gate[rows // S, grp] * yb)
    dout = q(torch.randn(B * S, D, generator=g))
    out.backward(dout)
    dgate = torch.zeros(B, G, D, device=dev)
    dyb = torch.empty(B * S, D, dtype=BF, device=dev)
    ops.gated_residual_bwd(dout.to(dev, BF), yb.to(dev, BF), gate.detach().to(dev), dgate, dyb, G * D, D, ops.groups(S, nt, P), B, D)
    close(dyb, gate.detach()[rows // S, grp] * dout)
    close(dgate, gate.grad)


@pytest.mark.parametrize("B,S,H,nt,use_rope", [(1, 200, 2, 8, False), (2, 333, 2, 13, True), (1, 3226, 2, 226, False),
                                                (1, 64, 1, 0, False)])
def test_attention_backward(B, S, H, nt, use_rope):
    """qkv_prep -> attention forward (fused) -> backward chain vs autograd through LayerNorm(64) + RoPE + softmax attention."""
    from orv_amd import ops
    from oracle import leaf
    dev = _dev()
    g = torch.Generator().manual_seed(S + 7)
    D = H * 64
    raw = q(torch.randn(B * S, 3 * D, generator=g) * 1.5).requires_grad_()
    gq, bq, gk, bk = (q(torch.randn(64, generator=g) * 0.5 + (1 if i % 2 == 0 else 0)).requires_grad_() for i in range(4))
    rope = None
    if use_rope:
        ang = torch.rand(S - nt, 32, generator=g) * 6.28
        rope = (ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous())
    x = raw.view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    qq = torch.nn.functional.layer_norm(x[0], (64,), gq, bq, 1e-6)
    kk = torch.nn.functional.layer_norm(x[1], (64,), gk, bk, 1e-6)
    if rope is not None:
        qq = torch.cat([qq[:, :, :nt], leaf.apply_rotary_emb(qq[:, :, nt:], rope)], dim=2)
        kk = torch.cat([kk[:, :, :nt], leaf.apply_rotary_emb(kk[:, :, nt:], rope)], dim=2)
    o = (torch.softmax(qq @ kk.transpose(-1, -2) / 8.0, dim=-1) @ x[2]).transpose(1, 2).reshape(B * S, D)
    do = q(torch.randn(B * S, D, generator=g))
    o.backward(do)
    # HIP path
    s_pad = (S + 63) // 64 * 64
    work = raw.detach().to(dev, BF).clone()
    rawd = raw.detach().to(dev, BF)
    vT, qT, kT, doT = (torch.zeros(B, H, 64, s_pad, dtype=BF, device=dev) for _ in range(4))
    rd = None if rope is None else tuple(r.to(dev) for r in rope)
    ops.qkv_prep(work, vT, gq.detach().to(dev, BF), bq.detach().to(dev, BF), gk.detach().to(dev, BF), bk.detach().to(dev, BF), rd,
                 B, S, H, nt, s_pad, 1e-6, q_premul=0.125 * LOG2E)
    out = torch.empty(B * S, D, dtype=BF, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention_fwd(work, vT, out, B, S, H, s_pad, 1.0 / LOG2E, lse=lse)
    dod = do.to(dev, BF)
    ops.head_transpose(work, 0, qT, B, S, H, s_pad, ld=3 * D)
    ops.head_transpose(work, D, kT, B, S, H, s_pad, ld=3 * D)
    ops.head_transpose(dod, 0, doT, B, S, H, s_pad, ld=D)
    nl, nd = (torch.empty(B, H, s_pad, dtype=torch.float32, device=dev) for _ in range(2))
    dqkv = torch.zeros(B * S, 3 * D, dtype=BF, device=dev)
    ops.attention_bwd(work, qT, kT, out, dod, doT, lse, nl, nd, dqkv, B, S, H, s_pad, 0.125)
    dgq, dbq, dgk, dbk = (torch.zeros(64, device=dev) for _ in range(4))
    ops.qkv_prep_bwd(rawd, dqkv, gq.detach().to(dev, BF), gk.detach().to(dev, BF), rd, dgq, dbq, dgk, dbk, B, S, H, nt, 1e-6)
    ref = raw.grad.view(B * S, 3, D)
    got = dqkv.float().cpu().view(B * S, 3, D)
    close(got[:, 2], ref[:, 2])                      # dv
    close(got[:, 1], ref[:, 1], afrac=2e-2)          # dk (through LayerNorm backward)
    close(got[:, 0], ref[:, 0], afrac=2e-2)          # dq
    close(dgq, gq.grad, afrac=2e-2), close(dbq, bq.grad, afrac=2e-2)
    close(dgk, gk.grad, afrac=2e-2)
    # d/d(beta_k) is analytically zero (softmax is invariant to a common shift of all keys): only bf16 rounding noise of the
    # summed dk rows remains, so judge it on the scale of the other norm_k gradient
    assert (dbk.cpu() - bk.grad).abs().max().item() <= 2e-2 * gk.grad.abs().max().item()


def test_small_linear_bwd_adamw_sumsq():
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    R, N, K = 24, 640, 512
    x, W = q(torch.randn(R, K, generator=g)), q(torch.randn(N, K, generator=g) * 0.1)
    dy = torch.randn(R, N, generator=g)
    dW = torch.zeros(N, K, dtype=BF, device=dev)
    db = torch.zeros(N, device=dev)
    dx = torch.zeros(R, K, device=dev)
    ops.small_linear_bwd(dy.to(dev), x.to(dev, BF), W.to(dev, BF), dW, db, dx, R, N, K, accumulate=False)
    close(dW, dy.t() @ x, rtol=1e-2, afrac=5e-3)
    close(db, dy.sum(0), rtol=1e-3, afrac=1e-4)
    close(dx, dy @ W, rtol=1e-3, afrac=1e-3)
    # AdamW vs torch.optim.AdamW (fp32 reference on bf16-representable values)
    n = 10000
    p0, gr = q(torch.randn(n, generator=g)), q(torch.randn(n, generator=g) * 0.1)
    pr = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-3)
    pd, m, v = p0.to(dev, BF), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    clip = torch.tensor([0.5], device=dev)
    for step in (1, 2, 3):
        pr.grad = gr * 0.5
        opt.step()
        ops.adamw(pd, gr.to(dev, BF), m, v, 1e-2, 0.9, 0.95, 1e-8, 1e-3, step, clip)
    close(pd, pr.detach(), rtol=1e-2, afrac=6e-3)     # three bf16 roundings of the parameter
    ss = torch.zeros(1, device=dev)
    ops.sumsq(gr.to(dev, BF), ss)
    assert abs(ss.item() - float((gr ** 2).sum())) <= 1e-3 * float((gr ** 2).sum())


def test_fused_adamw_flat_matches_torch_adamw():
    """FusedAdamW (flat buffers, one sumsq + one adamw launch, device-side clip) against torch.optim.AdamW +
    clip_grad_norm_ in fp32 on the same bf16-rounded gradients; a parameter without gradient is left untouched; derived
    weight caches (packed QKV) see the update."""
    from orv_amd.cogvideox_control import Attention
    from orv_amd.optim import FusedAdamW
    dev = _dev()
    torch.manual_seed(0)
    at = Attention(128, 2, 64, bias=True, out_bias=True).to(dev, BF)
    extra = torch.nn.Parameter(torch.randn(777, device=dev).to(BF))          # odd size: exercises the segment padding
    unused = torch.nn.Parameter(torch.randn(33, device=dev).to(BF))
    params = list(at.parameters()) + [extra, unused]
    ref = [p.detach().float().clone().requires_grad_(True) for p in params]
    topt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-3)
    opt = FusedAdamW(params, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-3, max_grad_norm=1.0)
    unused0 = unused.detach().clone()
    for it in range(3):
        for p, r in zip(params, ref):
            if p is unused:
                continue
            g = (torch.randn(p.shape, device=dev) * (3.0 if it == 0 else 0.05)).to(BF)    # step 0 clips, later ones do not
            p.grad = g.clone()
            r.grad = g.float()
        tn = torch.nn.utils.clip_grad_norm_([r for r in ref if r.grad is not None], 1.0)
        topt.step()
        n = opt.step()
        opt.zero_grad()
        assert abs(n - tn.item()) <= 2e-3 * tn.item()
    for p, r in zip(params, ref):
        if p is unused:
            assert torch.equal(p.detach(), unused0)
            continue
        close(p.detach().float().cpu(), r.detach().cpu(), rtol=2e-2, afrac=1e-2)       # bf16 parameter storage
    w, b = at.packed_qkv()
    assert torch.equal(w, torch.cat([at.to_q.weight, at.to_k.weight, at.to_v.weight]).detach())


@pytest.mark.parametrize("B,T,text", [(2, 3, True), (9, 5, True), (4, 1, False)])
def test_modulation_tables_bwd_matches_torch(B, T, text):
    """orv_modulation_tables_bwd (all AdaLN linears of a backward in two launches, incl. more than 32 conditioning rows)
    against torch autograd of the same linears."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 10 + T)
    n_tab, E, width = 3, 64, 96
    ntot = width * (2 if text else 1)
    Ws = [q(torch.randn(ntot, E, generator=g) * 0.2) for _ in range(n_tab)]
    cond_v, cond_t = q(torch.randn(B * T, E, generator=g)), q(torch.randn(B, E, generator=g))
    dtab = torch.randn(n_tab, B, 1 + T, width, generator=g)
    Wd = [w.to(dev, BF) for w in Ws]
    ptrs = torch.tensor([w.data_ptr() for w in Wd], dtype=torch.int64, device=dev)
    dcv = torch.zeros(B * T, E, device=dev)
    dct = torch.zeros(B, E, device=dev)
    gW, gb = ops.modulation_tables_bwd(dtab.to(dev), cond_v.to(dev, BF), cond_t.to(dev, BF), ptrs, dcv, dct, n_tab, B, T, E, width,
                                       text)
    ref_dcv, ref_dct = torch.zeros(B * T, E), torch.zeros(B, E)
    for t in range(n_tab):
        dv = dtab[t][:, 1:].reshape(B * T, width)
        close(gW[t][:width].float().cpu(), dv.t() @ cond_v, rtol=2e-2, afrac=1e-2)
        close(gb[t][:width].cpu(), dv.sum(0), rtol=1e-4, afrac=1e-5)
        ref_dcv += dv @ Ws[t][:width]
        if text:
            dt = dtab[t][:, 0]
            close(gW[t][width:].float().cpu(), dt.t() @ cond_t, rtol=2e-2, afrac=1e-2)
            close(gb[t][width:].cpu(), dt.sum(0), rtol=1e-4, afrac=1e-5)
            ref_dct += dt @ Ws[t][width:]
    close(dcv.cpu(), ref_dcv, rtol=1e-3, afrac=1e-4)
    if text:
        close(dct.cpu(), ref_dct, rtol=1e-3, afrac=1e-4)


def test_fused_adamw_per_parameter_step_counts_and_state_layout():
    """A parameter that gets no gradient in some steps keeps its OWN step count (torch.optim.AdamW ``state[p]['step']``): its
    bias correction must follow that count, not the optimizer's global one.  Also: state_dict carries the segment layout and
    load_state_dict refuses a different trainable set."""
    from orv_amd.optim import FusedAdamW
    dev = _dev()
    torch.manual_seed(1)
    always = torch.nn.Parameter(torch.randn(3000, device=dev).to(BF))
    sometimes = torch.nn.Parameter((0.01 * torch.randn(500, device=dev)).to(BF))   # small values: bf16 ulp << one update
    params = [always, sometimes]
    ref = [p.detach().float().clone().requires_grad_(True) for p in params]
    topt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-3)
    opt = FusedAdamW(params, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-3, max_grad_norm=0.0)
    for it in range(6):
        for p, r in zip(params, ref):
            if p is sometimes and it in (0, 1, 2, 4):       # first gradient arrives at global step 4 (its own step 1)
                p.grad, r.grad = None, None
                continue
            g = (torch.randn(p.shape, device=dev) * 0.1).to(BF)
            p.grad, r.grad = g.clone(), g.float()
        topt.step()
        opt.step()
        opt.zero_grad()
    assert opt._flat["seg_step"].tolist() == [6, 2]
    for p, r in zip(params, ref):
        close(p.detach().float().cpu(), r.detach().cpu(), rtol=2e-2, afrac=1e-2)
    # a parameter's first update is exactly lr in magnitude; with the GLOBAL count (4) its bias corrections would make it
    # 0.56 lr: a 4.4e-3 error per element, far above bf16 rounding at these magnitudes
    assert (ref[1].detach() - sometimes.detach().float()).abs().max().item() < 1.5e-3
    sd = opt.state_dict()
    assert sd["numels"] == [3000, 500] and sd["seg_start"][-1] % 2048 == 0
    opt2 = FusedAdamW([torch.nn.Parameter(p.detach().clone()) for p in params], lr=1e-2)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2._flat["m"], opt._flat["m"]) and opt2._flat["seg_step"].tolist() == [6, 2]
    with pytest.raises(ValueError, match="different set of trainable parameters"):
        FusedAdamW([torch.nn.Parameter(always.detach().clone())], lr=1e-2).load_state_dict(sd)
