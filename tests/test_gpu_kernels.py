"""GPU parity of each HIP kernel (through the C ABI via orv_amd.ops) against the CPU oracle / a plain fp32 restatement.

Tolerances: bit-exact for index work (patchify/unpatchify); bf16 kernels vs fp32 reference:
|err| <= 1.6e-2*|ref| + 1e-2*max|ref|.
"""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit, leaf  # noqa: E402  (checker only)

BF = torch.bfloat16


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def q(x):
    return x.to(BF).float()


def close(got, ref, rtol=1.6e-2, afrac=1e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    atol = afrac * ref.abs().max().item() + 1e-6
    bad = (got - ref).abs() > (rtol * ref.abs() + atol)
    assert not bad.any(), f"max err {(got - ref).abs().max().item():.4g} vs atol {atol:.4g} ({int(bad.sum())} bad)"


def test_library_targets_gfx950():
    from orv_amd._lib import lib, check
    check(lib().orv_device_check(0), "orv_device_check")


@pytest.mark.parametrize("M,N,K,epi", [(64, 128, 128, 0), (300, 192, 256, 1), (777, 384, 512, 2), (3226, 1920, 1920, 2),
                                       (700, 64, 1920, 0), (1000, 640, 7680, 1)])
def test_gemm_epilogues(M, N, K, epi):
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    A, W = q(torch.randn(M, K, generator=g)), q(torch.randn(N, K, generator=g) * 0.05)
    bias, R = q(torch.randn(N, generator=g)), q(torch.randn(M, N, generator=g))
    seq, nt, P = (M + 1) // 2, 17, 50
    B = math.ceil(M / seq)
    G = 1 + math.ceil((seq - nt) / P)
    gate = torch.randn(B, G, N, generator=g)
    ref = A @ W.t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if epi == 2:
        rows = torch.arange(M)
        s = rows % seq
        grp = torch.where(s < nt, torch.zeros_like(s), 1 + (s - nt) // P)
        ref = R + gate[rows // seq, grp] * ref
    C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    ops.gemm(A.to(dev, BF), W.to(dev, BF), bias.to(dev, BF), C, M, N, K, epilogue=epi, R=R.to(dev, BF), ldr=N,
             gate=gate.to(dev), gate_b=G * N, gate_g=N, grp=ops.groups(seq, nt, P))
    close(C, ref)


def test_gemm_row_remap_and_table_residual():
    """patch-embed form: rows scattered into a joint [B,S,D] buffer + [n,D] table added (r_mod)."""
    from orv_amd import ops
    dev = _dev()
    B, Nv, Nt, D, K = 3, 100, 9, 128, 128
    S = Nt + Nv
    g = torch.Generator().manual_seed(3)
    A, W, bias = q(torch.randn(B * Nv, K, generator=g)), q(torch.randn(D, K, generator=g) * 0.1), q(torch.randn(D, generator=g))
    pos = q(torch.randn(Nv, D, generator=g))
    x = torch.zeros(B * S, D, dtype=BF, device=dev)
    ops.gemm(A.to(dev, BF), W.to(dev, BF), bias.to(dev, BF), x, B * Nv, D, K, epilogue=2, R=pos.to(dev, BF), r_mod=Nv, ldr=D,
             cmap=ops.rowmap(Nv, S, Nt))
    ref = (A @ W.t() + bias).view(B, Nv, D) + pos
    got = x.view(B, S, D)
    close(got[:, Nt:], ref)
    assert torch.all(got[:, :Nt] == 0)          # text rows untouched


@pytest.mark.parametrize("D", [128, 1920, 3072])
def test_layernorm_modulate(D):
    from orv_amd import ops
    dev = _dev()
    B, S, nt, P = 2, 75, 7, 17
    G = 1 + (S - nt) // P
    g = torch.Generator().manual_seed(D)
    x = q(torch.randn(B * S, D, generator=g) * 2 + 0.5)
    gamma, beta = q(torch.randn(D, generator=g)), q(torch.randn(D, generator=g))
    mod = torch.randn(B, G, 2 * D, generator=g)
    rows = torch.arange(B * S)
    s = rows % S
    grp = torch.where(s < nt, torch.zeros_like(s), 1 + (s - nt) // P)
    ln = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5)
    ref = ln * (1 + mod[rows // S, grp, D:]) + mod[rows // S, grp, :D]
    y = torch.empty(B * S, D, dtype=BF, device=dev)
    md = mod.to(dev)
    ops.layernorm_modulate(x.to(dev, BF), y, gamma.to(dev, BF), beta.to(dev, BF), md[..., D:], md[..., :D], G * 2 * D,
                           2 * D, ops.groups(S, nt, P), B, D, 1e-5)
    close(y, ref)
    # plain LayerNorm reading only the video rows (row map)
    nv = S - nt
    y2 = torch.empty(B * nv, D, dtype=BF, device=dev)
    ops.layernorm_modulate(x.to(dev, BF), y2, gamma.to(dev, BF), beta.to(dev, BF), None, None, 0, 0, ops.groups(nv, 0, 0),
                           B, D, 1e-5, xmap=ops.rowmap(nv, S, nt))
    close(y2, ln.view(B, S, D)[:, nt:].reshape(B * nv, D))


@pytest.mark.parametrize("D,B,S,nt,P", [(1920, 3, 1033, 26, 201), (1920, 1, 3226, 226, 600), (128, 2, 1500, 7, 149), (512, 5, 421, 0, 0)])
def test_layernorm_modulate_rows_per_wave_kernel(D, B, S, nt, P):
    """Launches of >= 2048 rows without a row map take ln_mod_rows_kernel (a wave owns R consecutive rows, keeps the folded factor
    vectors of the current token group in registers, prefetches the next row): group changes inside a wave's row range, a batch
    boundary inside it, a ragged last wave and D < 512 (one chunk, masked lanes) against the fp32 reference."""
    from orv_amd import ops
    dev = _dev()
    G = 1 + ((S - nt) // P if P else 1)
    g = torch.Generator().manual_seed(D + S)
    x = q(torch.randn(B * S, D, generator=g) * 2 + 0.5)
    gamma, beta = q(torch.randn(D, generator=g)), q(torch.randn(D, generator=g))
    mod = torch.randn(B, G + 1, 2 * D, generator=g)
    rows = torch.arange(B * S)
    s = rows % S
    grp = torch.where(s < nt, torch.zeros_like(s), 1 + ((s - nt) // P if P else torch.zeros_like(s)))
    ln = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5)
    ref = ln * (1 + mod[rows // S, grp, D:]) + mod[rows // S, grp, :D]
    y = torch.full((B * S, D), float("nan"), dtype=BF, device=dev)
    md = mod.to(dev)
    ops.layernorm_modulate(x.to(dev, BF), y, gamma.to(dev, BF), beta.to(dev, BF), md[..., D:], md[..., :D], (G + 1) * 2 * D,
                           2 * D, ops.groups(S, nt, P), B, D, 1e-5)
    assert B * S >= 2048 or D == 512           # (the last case stays on the one-row kernel: same bound either way)
    close(y, ref)


def test_gemm_operand_of_4gib_and_more_leaves_the_t8_kernel():
    """gemm_t8_kernel addresses A and W with 32-bit byte offsets; a call whose A spans >= 4 GiB (here: a wide leading dimension)
    must be planned onto another kernel and still be right."""
    from orv_amd import ops
    dev = _dev()
    M, N, K = 2304, 768, 256
    lda = 1 << 20                                  # 2304 rows x 2 MiB = 4.5 GiB span
    g = torch.Generator().manual_seed(11)
    A, W = q(torch.randn(M, K, generator=g)), q(torch.randn(N, K, generator=g) * 0.1)
    big = torch.zeros(M * lda, dtype=BF, device=dev)
    Av = big.view(M, lda)
    Av[:, :K] = A.to(dev, BF)
    C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    ops.gemm(Av, W.to(dev, BF), None, C, M, N, K, lda=lda)
    close(C, A @ W.t())
    del big


@pytest.mark.parametrize("M,N,K,acc", [(512, 256, 640, False), (1920, 1920, 3226, False), (1920, 7680, 1000, True), (5760, 1920, 777, False),
                                        (264, 192, 64, True), (1920, 768, 12904, False)])
def test_gemm_tn_weight_gradient_form(M, N, K, acc):
    """orv_gemm_tn_bf16: C[M, N] (+)= A[K, M]^T . W[K, N] from row-major operands (transposing LDS reads, zero page behind the last contraction
    row): both tile widths, contraction lengths that are not multiples of 64 / 128, a ragged last row tile (M % 256 != 0), accumulation."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    A, W = q(torch.randn(K, M, generator=g) * 0.5), q(torch.randn(K, N, generator=g) * 0.5)
    C0 = q(torch.randn(M, N, generator=g))
    C = C0.to(dev, BF).clone() if acc else torch.full((M, N), float("nan"), dtype=BF, device=dev)
    ops.gemm_tn(A.to(dev, BF), W.to(dev, BF), C, M, N, K, accumulate=acc)
    ref = A.t() @ W / 1.0
    if acc:
        ref = ref + C0
    scale = (K ** 0.5) * 0.25
    assert torch.isfinite(C.float()).all()
    err = (C.float().cpu() - ref).abs().max().item()
    assert err <= 1.6e-2 * ref.abs().max().item() + 1e-2 * scale, (err, ref.abs().max().item())


@pytest.mark.parametrize("Mtok", [3300, 3226])
def test_gemm_tn_is_bit_identical_to_the_transposes_plus_nt_path(Mtok):
    """The opt-in weight-gradient path (ORV_WGRAD_TN=1: orv_gemm_tn_bf16 on the row-major dY, X) against what training._wgrad does by default
    (orv_transpose_bf16 twice + the NT kernel) at a training shape of one clip: same products, same fp32 accumulation order per K-tile -> equal bits."""
    from orv_amd import ops
    dev = _dev()
    N, K = 5760, 1920
    g = torch.Generator(device=dev).manual_seed(5)
    dY = (torch.randn(Mtok, N, device=dev, generator=g) * 0.5).to(BF)
    X = (torch.randn(Mtok, K, device=dev, generator=g) * 0.5).to(BF)
    dW_tn = torch.empty(N, K, dtype=BF, device=dev)
    ops.gemm_tn(dY, X, dW_tn, N, K, Mtok)
    dYT, XT = ops.transpose(dY, Mtok, N), ops.transpose(X, Mtok, K)
    dW_nt = torch.empty(N, K, dtype=BF, device=dev)
    ops.gemm(dYT, XT, None, dW_nt, N, K, dYT.shape[1])
    ref = dY.float().t() @ X.float()
    close(dW_nt, ref.cpu())
    if dYT.shape[1] % 128 == 0:            # the NT call is on the t8 kernel (same K-tile order): bit-identical
        assert torch.equal(dW_tn, dW_nt)
    else:
        close(dW_tn, ref.cpu())


def _attention_reference(qkv, B, S, H, gq, bq, gk, bk, rope, nt):
    D = H * 64
    x = qkv.view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)      # [3,B,H,S,64]
    qq = torch.nn.functional.layer_norm(x[0], (64,), gq, bq, 1e-6)
    kk = torch.nn.functional.layer_norm(x[1], (64,), gk, bk, 1e-6)
    if rope is not None:
        qq = torch.cat([qq[:, :, :nt], leaf.apply_rotary_emb(q(qq[:, :, nt:]), rope)], dim=2)
        kk = torch.cat([kk[:, :, :nt], leaf.apply_rotary_emb(q(kk[:, :, nt:]), rope)], dim=2)
    qq, kk = q(qq), q(kk)                                     # the kernel stores normalised q/k as bf16
    p = torch.softmax(qq @ kk.transpose(-1, -2) / 8.0, dim=-1)
    o = p @ x[2]
    lse = torch.logsumexp(qq @ kk.transpose(-1, -2) / 8.0, dim=-1)
    return o.transpose(1, 2).reshape(B * S, D), lse, qq, kk


LOG2E = 1.4426950408889634


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("B,S,H,nt,use_rope", [(2, 200, 2, 8, False), (1, 3226, 3, 226, False), (2, 333, 2, 13, True),
                                                (1, 64, 1, 0, False), (1, 257, 2, 1, True), (3, 100, 5, 4, False),
                                                (1, 17, 1, 0, False), (2, 700, 7, 30, True)])
def test_qkv_prep_and_attention(B, S, H, nt, use_rope, fused):
    """fused: softmax scale * log2(e) folded into q by orv_qkv_prep (q_premul) -> the attention kernel's exp2 fast path."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(S)
    D = H * 64
    qkv = q(torch.randn(B * S, 3 * D, generator=g) * 1.5)
    gq, bq, gk, bk = (q(torch.randn(64, generator=g) * 0.5 + (1 if i % 2 == 0 else 0)) for i in range(4))
    rope = None
    if use_rope:
        ang = torch.rand(S - nt, 32, generator=g) * 6.28
        rope = (ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous())
    ref, lse_ref, q_ref, k_ref = _attention_reference(qkv, B, S, H, gq, bq, gk, bk, rope, nt)
    s_pad = (S + 63) // 64 * 64
    dq = qkv.to(dev, BF).clone()
    vT = torch.full((B, H, 64, s_pad), float("nan"), dtype=BF, device=dev)
    ops.qkv_prep(dq, vT, gq.to(dev, BF), bq.to(dev, BF), gk.to(dev, BF), bk.to(dev, BF),
                 None if rope is None else tuple(r.to(dev) for r in rope), B, S, H, nt, s_pad, 1e-6,
                 q_premul=0.125 * LOG2E if fused else 1.0)
    got = dq.float().cpu().view(B, S, 3, H, 64)
    close(got[:, :, 0].transpose(1, 2), q_ref * (0.125 * LOG2E if fused else 1.0))
    close(got[:, :, 1].transpose(1, 2), k_ref)
    assert torch.equal(got[:, :, 2], qkv.view(B, S, 3, H, 64)[:, :, 2])          # v third untouched
    # vT layout: pos = key with bits 2 and 3 swapped inside each 16-key group; zero padding (bit-exact copy)
    key = torch.arange(s_pad)
    pos = (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1)
    want = torch.zeros(B, H, 64, s_pad)
    want[..., pos[:S]] = qkv.view(B, S, 3, H, 64)[:, :, 2].permute(0, 2, 3, 1)
    assert torch.equal(vT.float().cpu(), want)
    out = torch.full((B * S, D), float("nan"), dtype=BF, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention_fwd(dq, vT, out, B, S, H, s_pad, 1.0 / LOG2E if fused else 0.125, lse=lse)
    close(out, ref)
    close(lse, lse_ref, rtol=2e-2, afrac=5e-3)
    # shipped form: V read in place from the packed projection through ds_read_b64_tr_b16 (no V^T): same contraction order, so
    # the result is bit-identical to the pre-transposed form
    out2 = torch.full((B * S, D), float("nan"), dtype=BF, device=dev)
    lse2 = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention_fwd(dq, None, out2, B, S, H, s_pad, 1.0 / LOG2E if fused else 0.125, lse=lse2)
    assert torch.equal(out2, out) and torch.equal(lse2, lse)


@pytest.mark.parametrize("fused", [False, True])
def test_attention_online_softmax_rescale_branch(fused):
    """A key far above the rest, late in the sequence, forces a large running-max jump (the rescale path)."""
    from orv_amd import ops
    dev = _dev()
    B, S, H = 1, 320, 1
    g = torch.Generator().manual_seed(11)
    qkv = q(torch.randn(B * S, 192, generator=g))
    qkv[300, 64:128] = qkv[5, 0:64] * 8           # k[300] aligned with q[5]
    ident = (torch.ones(64), torch.zeros(64))
    s_pad = 320
    dq = qkv.to(dev, BF).clone()
    vT = torch.zeros(B, H, 64, s_pad, dtype=BF, device=dev)
    # no LayerNorm here: build vT by hand through qkv_prep with unit gamma on pre-normalised rows is not the point;
    # feed q,k through the same LN as the reference
    ref, _, _, _ = _attention_reference(qkv, B, S, H, ident[0], ident[1], ident[0], ident[1], None, 0)
    ops.qkv_prep(dq, vT, ident[0].to(dev, BF), ident[1].to(dev, BF), ident[0].to(dev, BF), ident[1].to(dev, BF), None, B, S,
                 H, 0, s_pad, 1e-6, q_premul=0.125 * LOG2E if fused else 1.0)
    out = torch.empty(B * S, 64, dtype=BF, device=dev)
    ops.attention_fwd(dq, vT, out, B, S, H, s_pad, 1.0 / LOG2E if fused else 0.125)
    close(out, ref)
    out2 = torch.empty(B * S, 64, dtype=BF, device=dev)
    ops.attention_fwd(dq, None, out2, B, S, H, s_pad, 1.0 / LOG2E if fused else 0.125)      # V in place (shipped form)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("pt", [None, 2])
def test_patchify_unpatchify_bit_exact(pt):
    from orv_amd import ops
    dev = _dev()
    B, T, C, H, W = 2, 4, 32, 8, 12
    x = q(torch.randn(B, T, C, H, W))
    tok = ops.patchify(x.to(dev, BF), None, 2, pt)
    assert torch.equal(tok.float().cpu(), dit.patchify(x, 2, pt))
    a, b = x[:, :, :16].contiguous(), x[:, :, 16:].contiguous()          # channel-concat read in place
    tok2 = ops.patchify(a.to(dev, BF), b.to(dev, BF), 2, pt)
    assert torch.equal(tok2, tok)
    Fo = 2 * 2 * (pt or 1) * 16
    y = q(torch.randn(B, tok.shape[1], Fo))
    out = ops.unpatchify(y.to(dev, BF), B, T, 16, H, W, 2, pt)
    assert torch.equal(out.float().cpu(), dit.unpatchify(y, B, T, H, W, 2, pt))


def test_skinny_linear_and_timestep_embedding():
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    for M, N, K in [(2, 512, 1920), (10, 5760, 512), (20, 2048, 28), (37, 28, 2048)]:
        x, W, b = q(torch.randn(M, K, generator=g)), q(torch.randn(N, K, generator=g) * 0.05), q(torch.randn(N, generator=g))
        ref = torch.nn.functional.gelu(x @ W.t() + b, approximate="tanh")
        got = ops.skinny_linear(x.to(dev, BF), W.to(dev, BF), b.to(dev, BF), act_out="gelu_tanh")
        close(got, ref)
    # broadcast add + SiLU on the input, fp32 output with a row map (the AdaLN table form)
    B, T, E, N = 2, 5, 512, 384
    a, t = q(torch.randn(B * T, E, generator=g)), q(torch.randn(B, E, generator=g))
    W, b = q(torch.randn(N, E, generator=g) * 0.05), q(torch.randn(N, generator=g))
    ref = torch.nn.functional.silu(q(a.view(B, T, E) + t[:, None])) @ W.t() + b
    table = torch.zeros(B, T + 1, N, dtype=torch.float32, device=dev)
    ops.skinny_linear(a.to(dev, BF), W.to(dev, BF), b.to(dev, BF), xb=t.to(dev, BF), xb_rep=T, act_in="silu", out=table,
                      ldo=N, omap=ops.rowmap(T, T + 1, 1))
    close(table[:, 1:], ref, rtol=1e-2, afrac=5e-3)
    assert torch.all(table[:, 0] == 0)
    ts = torch.tensor([999.0, 19.0, 500.0])
    emb = ops.timestep_embedding(ts.to(dev), 1920, True, 0.0)
    close(emb, leaf.get_timestep_embedding(ts, 1920, True, 0), rtol=1e-2, afrac=1e-2)


def test_sched_step_matches_scheduler_math():
    from orv_amd import schedulers
    dev = _dev()
    kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
              clip_sample=False, set_alpha_to_one=True, prediction_type="v_prediction", rescale_betas_zero_snr=True,
              snr_shift_scale=3.0, timestep_spacing="trailing")
    g = torch.Generator().manual_seed(9)
    x, v = q(torch.randn(2, 5, 16, 8, 12, generator=g)), q(torch.randn(2, 5, 16, 8, 12, generator=g))
    for n in (50, 3):
        ours, ref = schedulers.CogVideoXDDIMScheduler(**kw), leaf.CogVideoXDDIMScheduler(**kw)
        ours.set_timesteps(n), ref.set_timesteps(n)
        for t in ref.timesteps.tolist()[:3] + ref.timesteps.tolist()[-2:]:
            want = ref.step(v, t, x, return_dict=False)[0]
            got = ours.step(v.to(dev, BF), t, x.to(dev, BF), return_dict=False)[0]
            close(got, want, rtol=8e-3, afrac=1e-3)
        ours, ref = schedulers.CogVideoXDPMScheduler(**kw), leaf.CogVideoXDPMScheduler(**kw)
        ours.set_timesteps(n), ref.set_timesteps(n)
        ts = ref.timesteps.tolist()
        old_o = old_r = None
        xo, xr = x.to(dev, BF), x.clone()
        for i, t in enumerate(ts):
            go, gr = torch.Generator().manual_seed(100 + i), torch.Generator().manual_seed(100 + i)
            # fp32 restatement of leaf.CogVideoXDPMScheduler.step with the noise drawn like the reference does for a
            # bf16 pipeline (randn_tensor in the sample dtype; two draws on second-order steps, the second one used)
            prev = t - 1000 // n
            a_t, a_p = ref.alphas_cumprod[t], (ref.alphas_cumprod[prev] if prev >= 0 else ref.final_alpha_cumprod)
            a_b = ref.alphas_cumprod[ts[i - 1]] if i else None
            m1, m2, mn, m3, m4 = ref.coefficients(a_t, a_p, a_b)
            x0 = (a_t ** 0.5).float() * xr - ((1 - a_t) ** 0.5).float() * v
            noise = torch.randn(x.shape, generator=gr, dtype=BF).float()
            d = x0
            if not (old_r is None or prev < 0):
                d = m3.float() * x0 - m4.float() * old_r
                noise = torch.randn(x.shape, generator=gr, dtype=BF).float()
            nr = m1.float() * xr - m2.float() * d + mn.float() * noise
            no, old_o = ours.step(v.to(dev, BF), old_o, t, ts[i - 1] if i else None, xo, generator=go)
            close(no, nr, rtol=8e-3, afrac=1e-3)
            close(old_o, x0, rtol=1e-4, afrac=1e-5)
            xr, old_r, xo = no.float().cpu(), x0, no
            if i == 3 and n == 50:
                break


@pytest.mark.parametrize("B,T,E,width,text,with_act", [(2, 5, 512, 5760, True, True), (4, 5, 512, 3840, False, True),
                                                        (3, 1, 64, 384, True, False), (9, 5, 512, 384, True, True)])
def test_modulation_tables(B, T, E, width, text, with_act):
    """All AdaLN linears in one launch vs F.linear(silu(temb + a)) / F.linear(silu(temb)) (cogvideox_control.py:117-130)."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 100 + T)
    n_tab = 3
    temb = q(torch.randn(B, E, generator=g))
    act = q(torch.randn(B, T, E, generator=g)) if with_act else None
    Ws = [q(torch.randn(width * (2 if text else 1), E, generator=g) * 0.05) for _ in range(n_tab)]
    bs = [q(torch.randn(width * (2 if text else 1), generator=g)) for _ in range(n_tab)]
    dW, db = [w.to(dev, BF) for w in Ws], [b.to(dev, BF) for b in bs]
    wp = torch.tensor([w.data_ptr() for w in dW], dtype=torch.int64, device=dev)
    bp = torch.tensor([b.data_ptr() for b in db], dtype=torch.int64, device=dev)
    out = ops.modulation_tables(temb.to(dev, BF), None if act is None else act.to(dev, BF), wp, bp, n_tab, B, T, E, width, text)
    cond_v = torch.nn.functional.silu(q(temb[:, None] + act)) if with_act else torch.nn.functional.silu(temb)[:, None]
    for i in range(n_tab):
        ref_v = q(cond_v) @ Ws[i][:width].t() + bs[i][:width]
        close(out[i, :, 1:], ref_v, rtol=1e-2, afrac=5e-3)
        if text:
            ref_t = q(torch.nn.functional.silu(temb)) @ Ws[i][width:].t() + bs[i][width:]
            close(out[i, :, 0], ref_t, rtol=1e-2, afrac=5e-3)
        else:
            assert torch.all(out[i, :, 0] == 0)


def test_gather_and_gated_scatter_rows():
    """MVBlock token regrouping (cogvideox_control.py:328-331,:346-347): exact permutation, one-rounding gated add."""
    from orv_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    Bv, S, Nt, D = 4, 20, 4, 128
    x = torch.randn(Bv * S, D, generator=g).to(dev, torch.bfloat16)
    R = 50
    idx = torch.randperm(Bv * S, generator=g)[:R].to(torch.int32).to(dev)
    dst = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
    ops.gather_rows(x, idx, dst, R, D)
    assert torch.equal(dst, x[idx.long()])
    y = torch.randn(R, D, generator=g).to(dev, torch.bfloat16)
    gate = torch.randn(Bv, 3 * D, generator=g).to(dev)
    x2 = x.clone()
    ops.scatter_gated_rows(y, idx, gate[:, 2 * D:], 3 * D, x2, R, D, S, Nt)
    ref = x.clone().float()
    il = idx.long()
    vid = (il % S) >= Nt
    ref[il[vid]] += gate[il[vid] // S, 2 * D:] * y[vid].float()
    # fp32 fma vs mul+add may flip a rare bf16 rounding tie: allow one bf16 ulp on <0.1% of the elements
    d = (x2.float() - ref.to(torch.bfloat16).float()).abs()
    assert (d > 0).float().mean().item() < 1e-3 and d.max().item() <= 2 ** -5
    assert torch.equal(x2[il[~vid]], x[il[~vid]])


@pytest.mark.parametrize("M,H", [(300, 2), (3226, 30)])
def test_qkv_gemm_with_fused_qk_layernorm(M, H):
    """GEMM epilogue 4 (+ orv_head_transpose for V^T) against the unfused pair it replaces: epilogue 0 + orv_qkv_prep."""
    from orv_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M)
    D = H * 64
    B, S = 1, M
    s_pad = (S + 63) // 64 * 64
    x = torch.randn(M, D, generator=g).to(dev, torch.bfloat16)
    W = (torch.randn(3 * D, D, generator=g) * 0.03).to(dev, torch.bfloat16)
    bias = (torch.randn(3 * D, generator=g) * 0.1).to(dev, torch.bfloat16)
    gq, bq, gk, bk = ((torch.randn(64, generator=g) * 0.2 + (1.0 if i % 2 == 0 else 0.0)).to(dev, torch.bfloat16) for i in range(4))
    premul = 0.125 * 1.4426950408889634
    ref = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
    ops.gemm(x, W, bias, ref, M, 3 * D, D)
    raw = ref.clone()
    vT_ref = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=dev)
    ops.qkv_prep(ref, vT_ref, gq, bq, gk, bk, None, B, S, H, 0, s_pad, 1e-6, q_premul=premul)
    out = torch.empty_like(ref)
    y = torch.empty_like(ref)
    ops.gemm(x, W, bias, out, M, 3 * D, D, epilogue=4, Y=y, qknorm=(gq, bq, gk, bk, 1e-6, premul, H))
    vT = torch.zeros_like(vT_ref)
    ops.head_transpose(out, 2 * D, vT, B, S, H, s_pad, ld=3 * D)
    assert torch.equal(y, raw)                                   # Y = the raw projection, bit for bit
    assert torch.equal(out[:, 2 * D:], ref[:, 2 * D:]) and torch.equal(vT, vT_ref)
    # q / k: same arithmetic from the fp32 accumulator instead of the bf16-rounded projection -> bf16-level differences
    d = (out[:, :2 * D].float() - ref[:, :2 * D].float()).abs()
    scale = ref[:, :2 * D].float().abs().max().item()
    assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-3 * scale


@pytest.mark.parametrize("N,K,epi", [(5760, 1920, 0), (7680, 1920, 1), (1920, 7680, 2), (5760, 1920, 4)])
def test_ring_gemm_is_deterministic_under_repetition(N, K, epi):
    """Race screen for the persistent ring GEMM (counted vmcnt waits, DMA stream running across tile boundaries, epilogue
    stores draining under the next tile): 25 back-to-back launches of the B=4 shapes must agree bit for bit, and with a
    launch whose stores are forced to drain (ORV_GEMM_DBG=7 path is exercised separately by the partial last M tile)."""
    from orv_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N + K)
    M = 12904
    x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.03).to(dev, torch.bfloat16)
    bias = (torch.randn(N, generator=g) * 0.1).to(dev, torch.bfloat16)
    R = torch.randn(M, N, generator=g).to(dev, torch.bfloat16) if epi == 2 else None
    gate = torch.randn(4, 6, N, generator=g).to(dev) if epi == 2 else None
    kw = {}
    if epi == 2:
        kw = dict(R=R, ldr=N, gate=gate, gate_b=6 * N, gate_g=N, grp=ops.groups(3226, 226, 600))
    if epi == 4:
        one = torch.ones(64, dtype=torch.bfloat16, device=dev)
        kw = dict(qknorm=(one, None, one, None, 1e-6, 0.18, N // 192))
    outs = []
    for _ in range(25):
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(x, W, bias, out, M, N, K, epilogue=epi, **kw)
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert torch.isfinite(outs[0].float()).all()


def test_gemm_shape_fuzz_against_torch():
    """Random (M, N, K, epilogue) over everything the tile chooser can pick (ring 384/256/192/128, simple kernel, one tile to many
    rounds, K = 64 ... 4096, ragged M) against torch fp32 on the GPU; tolerance = bf16 output rounding + K-length accumulation."""
    from orv_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    g = torch.Generator(device=dev).manual_seed(7)
    cases = [(1, 64, 64, 0), (255, 192, 64, 1), (257, 384, 128, 2), (4097, 768, 64, 0), (12904, 384, 64, 0), (70000, 64, 192, 1)]
    for _ in range(34):
        M = int(rng.choice([1, 7, 63, 64, 65, 127, 129, 300, 511, 1000, 2049, 3226, 5000, 9999, 12904, 20000]))
        N = int(rng.choice([64, 128, 192, 256, 320, 384, 640, 768, 1152, 1536, 1920, 3072, 5760]))
        K = int(rng.choice([64, 128, 192, 320, 512, 1024, 1920, 4096]))
        cases.append((M, N, K, int(rng.integers(0, 3))))
    for M, N, K, epi in cases:
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev, generator=g) * (1.0 / K ** 0.5)).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        ref = A.float() @ W.float().t() + bias.float()
        kw = {}
        if epi == 1:
            ref = torch.nn.functional.gelu(ref, approximate="tanh")
        if epi == 2:
            R = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
            ref = R.float() + ref
            kw = dict(R=R, ldr=N)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm(A, W, bias, out, M, N, K, epilogue=epi, **kw)
        err = (out.float() - ref).abs()
        tol = 1.6e-2 * ref.abs() + 2e-2
        assert bool((err <= tol).all()), (M, N, K, epi, err.max().item())


_TILE_CHECK = r"""
import sys, torch
from orv_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)
bad = []
for M, N, K in [(3226, 1920 if int(sys.argv[1]) % 192 == 0 or int(sys.argv[1]) in (64, 128) else 1536, 1920), (700, 768, 256), (257, 768, 128)]:
    if N % int(sys.argv[1]):
        continue
    for epi in (0, 1, 2, 3):
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        R = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        ref = A.float() @ W.float().t() + bias.float()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref, approximate="tanh")
        if epi == 2:
            ref = R.float() + ref
        if epi == 3:      # acc * GELU_tanh'(R), no bias (dgrad of FeedForward net.2 fused with the GELU adjoint)
            x = R.float().requires_grad_(True)
            (dg,) = torch.autograd.grad(torch.nn.functional.gelu(x, approximate="tanh").sum(), x)
            ref = (A.float() @ W.float().t()) * dg
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm(A, W, None if epi == 3 else bias, out, M, N, K, epilogue=epi, **(dict(R=R, ldr=N) if epi >= 2 else {}))
        err = (out.float() - ref).abs()
        if not bool((err <= 1.6e-2 * ref.abs() + 2e-2).all()):
            bad.append((M, N, K, epi, float(err.max())))
print("BAD", bad) if bad else print("TILE-OK")
"""


@pytest.mark.parametrize("tile", ["3,256,256", "3,256,192", "2,256,256", "2,256,128", "1,256,384", "1,256,256", "1,256,192", "1,256,128", "0,256,192",
                                  "0,256,128", "0,256,64", "0,128,192", "0,128,128", "0,128,64", "0,192,128"])
def test_every_gemm_tile_instantiation_forced(tile):
    """The tile chooser only ever picks what its cost model prefers; here every candidate (phased / ring / simple, every tile
    shape) is pinned with ORV_GEMM_TILE in a fresh process and checked on ragged M, single- and multi-round grids, epilogues
    0-3 against torch fp32."""
    import os, subprocess, sys
    env = dict(os.environ, ORV_GEMM_TILE=tile)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _TILE_CHECK, tile.split(",")[2]], cwd=root, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "TILE-OK" in r.stdout, (tile, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("bn", [256, 192])
def test_t8_gemm_matches_simple_kernel_on_every_epilogue_feature(bn):
    """gemm_t8_kernel (16x16x32 MFMA, permuted W rows, 16-byte epilogue pieces) against the simple kernel pinned in the same
    process (orv_gemm_force_tile), on everything the epilogue can do: bias, GELU, gated residual with per-token-group gates that
    straddle 16-row blocks, row scatter (cmap), r_mod residual, the Y side output, GELU adjoint, ragged M, one-tile and
    multi-round grids; BN = 256 also the fused qk-LayerNorm epilogue.  Both kernels accumulate in fp32 over the same bf16
    inputs: outputs agree to bf16 rounding of slightly different summation orders."""
    from orv_amd import ops
    from orv_amd._lib import lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5 + bn)
    other = (0, 256, 128) if bn == 256 else (0, 256, 192)

    def run(tile, fn):
        lib().orv_gemm_force_tile(*tile)
        try:
            return fn()
        finally:
            lib().orv_gemm_force_tile(0, 0, 0)

    def close(a, b, what):
        a, b = a.float(), b.float()
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), what
        err = (a - b).abs()
        assert bool((err <= 1.0e-2 * b.abs() + 1.5e-2).all()), (what, err.max().item())

    N = bn * 3
    for M, K in [(100, 128), (3226 * 2, 256), (70000, 128)]:
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        R = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        for epi in (0, 1, 2, 3):
            def fn():
                C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
                Y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
                ops.gemm(A, W, None if epi == 3 else bias, C, M, N, K, epilogue=epi, Y=Y, **(dict(R=R, ldr=N) if epi >= 2 else {}))
                return C, Y
            (c1, y1), (c0, y0) = run((3, 256, bn), fn), run(other, fn)
            close(c1, c0, (bn, M, K, epi, "C")), close(y1, y0, (bn, M, K, epi, "Y"))
    # gated residual with token groups (text rows + frames of 37 rows: gate rows change inside 16-row blocks), row scatter into
    # a joint buffer and a residual addressed modulo r_mod
    seq, n_text, per_group, Bn = 500, 19, 37, 3
    Mv = (seq - n_text) * Bn
    K = 128
    A = torch.randn(Mv, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    n_groups = 1 + (seq - n_text + per_group - 1) // per_group
    gate = torch.randn(Bn, n_groups, N, device=dev, generator=g)
    base = torch.randn(Bn * seq, N, device=dev, generator=g).to(torch.bfloat16)

    def fn_gate():
        C = base.clone()
        ops.gemm(A, W, bias, C, Mv, N, K, epilogue=2, R=C, ldr=N, gate=gate, gate_b=n_groups * N, gate_g=N,
                 grp=ops.Groups(seq, n_text, per_group), cmap=ops.RowMap(seq - n_text, seq, n_text))
        return C
    close(run((3, 256, bn), fn_gate), run(other, fn_gate), (bn, "gate + cmap"))
    Rm = torch.randn(97, N, device=dev, generator=g).to(torch.bfloat16)

    def fn_rmod():
        C = torch.full((Mv, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm(A, W, bias, C, Mv, N, K, epilogue=2, R=Rm, ldr=N, r_mod=97)
        return C
    close(run((3, 256, bn), fn_rmod), run(other, fn_rmod), (bn, "r_mod"))
    if bn == 256:
        heads = 4
        Nq = 3 * heads * 64
        M, K = 3226, 256
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = (torch.randn(Nq, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(Nq, device=dev, generator=g).to(torch.bfloat16)
        aff = [torch.randn(64, device=dev, generator=g).to(torch.bfloat16) for _ in range(4)]

        def fn_qk():
            C = torch.full((M, Nq), float("nan"), dtype=torch.bfloat16, device=dev)
            Y = torch.full((M, Nq), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(A, W, bias, C, M, Nq, K, epilogue=4, Y=Y, qknorm=(aff[0], aff[1], aff[2], aff[3], 1e-6, 0.18, heads))
            return C, Y
        (c1, y1), (c0, y0) = run((3, 256, 256), fn_qk), run((0, 256, 128), fn_qk)
        close(y1, y0, "qknorm Y")
        # LayerNorm output of near-identical inputs: compare in units of the affine scale
        assert ((c1.float() - c0.float()).abs() <= 2e-2 * c0.float().abs() + 6e-2).all()


@pytest.mark.parametrize("bn", [256, 192])
def test_t8_192_row_tile_is_bit_identical_to_the_256_row_tile(bn):
    """gemm_t8r192_kernel (the t8 kernel on 192-row tiles: a wave owns 96 x BN / 4, waves 6 / 7 stage no A pieces) against gemm_t8_kernel
    pinned in the same process: the same K order per output element, so every epilogue it has - bias, GELU (+ the Y side output, + the
    packed P16 store at BN = 256), gated residual with per-token-group gates that straddle 16-row blocks, row scatter, r_mod residual,
    the fused qk LayerNorm (BN = 256) - must agree BIT FOR BIT, on ragged M, single-tile, one-clip (17 row tiles) and multi-round grids."""
    from orv_amd import ops
    from orv_amd._lib import lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(50 + bn)

    def run(tile, fn):
        lib().orv_gemm_force_tile(*tile)
        try:
            return fn()
        finally:
            lib().orv_gemm_force_tile(0, 0, 0)

    def same(a, b, what):
        assert torch.isfinite(a.float()).all(), what
        assert torch.equal(a, b), (what, (a.float() - b.float()).abs().max().item())

    N = bn * 3
    for M, K in [(100, 128), (191, 256), (193, 128), (3226, 384), (3226 * 2, 256), (40000, 128)]:
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        R = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        for epi in (0, 1, 2):
            def fn():
                C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
                Y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
                ops.gemm(A, W, bias, C, M, N, K, epilogue=epi, Y=Y, **(dict(R=R, ldr=N) if epi == 2 else {}))
                return C, Y
            (c1, y1), (c0, y0) = run((3, 192, bn), fn), run((3, 256, bn), fn)
            same(c1, c0, (bn, M, K, epi, "C")), same(y1, y0, (bn, M, K, epi, "Y"))
        if bn == 256 and (M + 191) // 192 * 192 <= ops.packed_rows(M):
            def fn_packed():                                  # GELU epilogue into the packed P16 layout (FFN1 -> FFN2), unpacked for the comparison
                Cp = torch.zeros(ops.packed_rows(M), N, dtype=torch.bfloat16, device=dev)
                ops.gemm(A, W, bias, Cp, M, N, K, epilogue=1, c_packed=True)
                return ops.unpack_rows16(Cp, M, N)
            same(run((3, 192, bn), fn_packed), run((3, 256, bn), fn_packed), (bn, M, K, "packed C"))
    seq, n_text, per_group, Bn = 500, 19, 37, 3
    Mv = (seq - n_text) * Bn
    K = 128
    A = torch.randn(Mv, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    n_groups = 1 + (seq - n_text + per_group - 1) // per_group
    gate = torch.randn(Bn, n_groups, N, device=dev, generator=g)
    base = torch.randn(Bn * seq, N, device=dev, generator=g).to(torch.bfloat16)

    def fn_gate():
        C = base.clone()
        ops.gemm(A, W, bias, C, Mv, N, K, epilogue=2, R=C, ldr=N, gate=gate, gate_b=n_groups * N, gate_g=N,
                 grp=ops.Groups(seq, n_text, per_group), cmap=ops.RowMap(seq - n_text, seq, n_text))
        return C
    same(run((3, 192, bn), fn_gate), run((3, 256, bn), fn_gate), (bn, "gate + cmap"))
    Rm = torch.randn(97, N, device=dev, generator=g).to(torch.bfloat16)

    def fn_rmod():
        C = torch.full((Mv, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm(A, W, bias, C, Mv, N, K, epilogue=2, R=Rm, ldr=N, r_mod=97)
        return C
    same(run((3, 192, bn), fn_rmod), run((3, 256, bn), fn_rmod), (bn, "r_mod"))
    if bn == 256:
        heads = 4
        Nq = 2 * heads * 64                                   # q | k: what the split projection hands to this kernel
        for M in (3226, 777):
            K = 256
            A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
            W = (torch.randn(Nq, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(Nq, device=dev, generator=g).to(torch.bfloat16)
            aff = [torch.randn(64, device=dev, generator=g).to(torch.bfloat16) for _ in range(4)]

            def fn_qk():
                C = torch.full((M, Nq), float("nan"), dtype=torch.bfloat16, device=dev)
                Y = torch.full((M, Nq), float("nan"), dtype=torch.bfloat16, device=dev)
                ops.gemm(A, W, bias, C, M, Nq, K, epilogue=4, Y=Y, qknorm=(aff[0], aff[1], aff[2], aff[3], 1e-6, 0.18, heads))
                return C, Y
            (c1, y1), (c0, y0) = run((3, 192, 256), fn_qk), run((3, 256, 256), fn_qk)
            same(c1, c0, ("qknorm C", M)), same(y1, y0, ("qknorm Y", M))


@pytest.mark.parametrize("B,S,H,nt,use_rope", [(2, 200, 2, 8, False), (1, 3226, 3, 226, False), (2, 333, 2, 13, True), (1, 17, 1, 0, False)])
def test_attention_fixed_shift_softmax_with_a_score_bound(B, S, H, nt, use_rope):
    """orv_attention_fwd_bounded: ORV's qk LayerNorm bounds every score by (8 max|gamma_q| + ||beta_q||)(8 max|gamma_k| + ||beta_k||)
    * scale * log2 e (Attention.score_bound), and the kernel uses that bound as a FIXED softmax shift (no running max / rescale).
    Against the fp32 reference (same tolerance as the online kernel), against the online kernel itself, lse included, ragged
    last key tile, RoPE (norm preserving).  The bound really holds on the data: checked against the reference q', k'."""
    from orv_amd import ops
    from orv_amd.cogvideox_control import Attention
    dev = _dev()
    g = torch.Generator().manual_seed(S + 1)
    D = H * 64
    qkv = q(torch.randn(B * S, 3 * D, generator=g) * 1.5)
    gq, bq, gk, bk = (q(torch.randn(64, generator=g) * 0.1 + (1 if i % 2 == 0 else 0)) for i in range(4))
    rope = None
    if use_rope:
        ang = torch.rand(S - nt, 32, generator=g) * 6.28
        rope = (ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous())
    ref, lse_ref, q_ref, k_ref = _attention_reference(qkv, B, S, H, gq, bq, gk, bk, rope, nt)
    at = Attention(D, H, 64, True, True)
    with torch.no_grad():
        at.norm_q.weight.copy_(gq); at.norm_q.bias.copy_(bq); at.norm_k.weight.copy_(gk); at.norm_k.bias.copy_(bk)
    bound = at.score_bound(0.125)
    smax = (torch.einsum("bhqd,bhkd->bhqk", q_ref, k_ref).abs().max() * 0.125 * LOG2E).item()
    assert smax <= bound <= 40.0, (smax, bound)
    s_pad = (S + 63) // 64 * 64
    dq = qkv.to(dev, BF).clone()
    ops.qkv_prep(dq, None, gq.to(dev, BF), bq.to(dev, BF), gk.to(dev, BF), bk.to(dev, BF),
                 None if rope is None else tuple(r.to(dev) for r in rope), B, S, H, nt, s_pad, 1e-6, q_premul=0.125 * LOG2E)
    out = torch.full((B * S, D), float("nan"), dtype=BF, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention_fwd(dq, None, out, B, S, H, s_pad, 1.0 / LOG2E, lse=lse, score_bound=bound)
    close(out, ref)
    close(lse, lse_ref, rtol=2e-2, afrac=5e-3)
    out_online = torch.empty_like(out)
    lse_online = torch.empty_like(lse)
    ops.attention_fwd(dq, None, out_online, B, S, H, s_pad, 1.0 / LOG2E, lse=lse_online)
    close(out, out_online.float().cpu(), rtol=1.6e-2, afrac=4e-3)
    assert (lse - lse_online).abs().max().item() <= 2e-3


def test_attention_bound_too_large_takes_the_online_kernel():
    """Adversarially large qk-LayerNorm gains: the bound exceeds what the fixed shift may absorb (90 log2 units), and the bounded
    entry point must run the online kernel - bit-identical to orv_attention_fwd, correct against the reference."""
    from orv_amd import ops
    from orv_amd.cogvideox_control import Attention
    dev = _dev()
    B, S, H = 1, 300, 2
    g = torch.Generator().manual_seed(3)
    qkv = q(torch.randn(B * S, 3 * H * 64, generator=g))
    gq, bq, gk, bk = q(torch.full((64,), 3.0)), q(torch.zeros(64)), q(torch.full((64,), 2.5)), q(torch.randn(64, generator=g))
    at = Attention(H * 64, H, 64, True, True)
    with torch.no_grad():
        at.norm_q.weight.copy_(gq); at.norm_q.bias.copy_(bq); at.norm_k.weight.copy_(gk); at.norm_k.bias.copy_(bk)
    bound = at.score_bound(0.125)
    from orv_amd._lib import lib
    assert bound > lib().orv_attention_static_limit(1) == 90.0
    dq = qkv.to(dev, BF).clone()
    ops.qkv_prep(dq, None, gq.to(dev, BF), bq.to(dev, BF), gk.to(dev, BF), bk.to(dev, BF), None, B, S, H, 0, 320, 1e-6,
                 q_premul=0.125 * LOG2E)
    out, out_online = torch.empty(B * S, H * 64, dtype=BF, device=dev), torch.empty(B * S, H * 64, dtype=BF, device=dev)
    ops.attention_fwd(dq, None, out, B, S, H, 320, 1.0 / LOG2E, score_bound=bound)
    ops.attention_fwd(dq, None, out_online, B, S, H, 320, 1.0 / LOG2E)
    assert torch.equal(out, out_online)
    # scores reach +-50 log2 units here, so the bf16 rounding of q' / k' alone moves P by several per cent: the reference is the
    # fp32 softmax over the SAME bf16 q', k', v the kernel read (exp2: q' carries scale * log2 e)
    t = dq.float().cpu().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    pr = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", t[0], t[1]) * math.log(2.0), dim=-1)
    close(out, torch.einsum("bhqk,bhkd->bhqd", pr, t[2]).transpose(1, 2).reshape(B * S, H * 64))


def _gamma_case(gq_val, gk_val, seed=5, B=2, S=520, H=3):
    from orv_amd import ops
    from orv_amd.cogvideox_control import Attention
    dev = _dev()
    g = torch.Generator().manual_seed(seed)
    qkv = q(torch.randn(B * S, 3 * H * 64, generator=g))
    gq, bq = q(torch.full((64,), gq_val) * (1 + 0.05 * torch.randn(64, generator=g)).clamp(0.8, 1.0)), q(torch.randn(64, generator=g) * 0.2)
    gk, bk = q(torch.full((64,), gk_val) * (1 + 0.05 * torch.randn(64, generator=g)).clamp(0.8, 1.0)), q(torch.randn(64, generator=g) * 0.2)
    at = Attention(H * 64, H, 64, True, True)
    with torch.no_grad():
        at.norm_q.weight.copy_(gq); at.norm_q.bias.copy_(bq); at.norm_k.weight.copy_(gk); at.norm_k.bias.copy_(bk)
    at = at.to(dev)                          # the device-side bound lives where the norm parameters live
    s_pad = (S + 63) // 64 * 64
    dq = qkv.to(dev, BF).clone()
    ops.qkv_prep(dq, None, gq.to(dev, BF), bq.to(dev, BF), gk.to(dev, BF), bk.to(dev, BF), None, B, S, H, 0, s_pad, 1e-6,
                 q_premul=0.125 * LOG2E)
    # fp32 softmax over the SAME bf16 q', k', v the kernel reads (exp2: q' carries scale * log2 e)
    t = dq.float().cpu().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = torch.einsum("bhqd,bhkd->bhqk", t[0], t[1])
    pr = torch.softmax(sc * math.log(2.0), dim=-1)
    ref = torch.einsum("bhqk,bhkd->bhqd", pr, t[2]).transpose(1, 2).reshape(B * S, H * 64)
    lse_ref = torch.logsumexp(sc * math.log(2.0), dim=-1)
    return at, dq, ref, lse_ref, sc.abs().max().item(), (B, S, H, s_pad)


def test_attention_fixed_shift_fast_path_at_gamma_product_5():
    """a trained qk LayerNorm with max|gamma_q| max|gamma_k| = 5 (bound ~ 60 log2 units: ordinary for a trained
    model, far above the 11.8 of random init and above round 3's limit of 40) still runs the shift-free softmax: P = exp2(s)
    spans [2^-60, 2^60] and stays a normal fp32 / bf16 number.  Against the fp32 softmax over the same bf16 q', k', v, lse
    included, and against the online kernel."""
    from orv_amd import ops
    from orv_amd._lib import lib
    at, dq, ref, lse_ref, smax, (B, S, H, s_pad) = _gamma_case(2.5, 2.0)
    bound = at.score_bound(0.125)
    assert 40.0 < bound <= lib().orv_attention_static_limit(1), bound       # the dispatch takes the fixed-shift kernel
    assert smax <= bound and smax > 12.0, smax                              # the bound holds, and the scores really are large
    dev = dq.device
    out = torch.full((B * S, H * 64), float("nan"), dtype=BF, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention_fwd(dq, None, out, B, S, H, s_pad, 1.0 / LOG2E, lse=lse, score_bound=bound)
    close(out, ref)
    close(lse, lse_ref, rtol=2e-3, afrac=2e-3)
    out_online, lse_online = torch.empty_like(out), torch.empty_like(lse)
    ops.attention_fwd(dq, None, out_online, B, S, H, s_pad, 1.0 / LOG2E, lse=lse_online)
    close(out, out_online.float().cpu(), rtol=1.6e-2, afrac=4e-3)
    assert (lse - lse_online).abs().max().item() <= 2e-3 * max(1.0, lse_online.abs().max().item())


@pytest.mark.parametrize("gq_val,gk_val", [(1.0, 1.0), (2.5, 2.0), (4.0, 3.5)])
def test_attention_device_side_bound_selects_the_same_kernel(gq_val, gk_val):
    """orv_attention_fwd_bounded_dev (training: the bound changes with every optimizer step and is never read back to the host):
    both softmax forms are launched and the device scalar picks one - bit-identical to the host-side dispatch with the same bound,
    below the limit (fixed-shift kernel) and above it (online kernel; 4 x 3.5 gives ~ 165 log2 units)."""
    from orv_amd import ops
    at, dq, ref, lse_ref, smax, (B, S, H, s_pad) = _gamma_case(gq_val, gk_val, seed=9)
    dev = dq.device
    bound = at.score_bound(0.125)
    bdev = at.score_bound_dev(0.125)
    assert bdev.is_cuda and bdev.numel() == 1 and abs(bdev.item() - bound) <= 1e-4 * bound
    out_h, out_d = (torch.full((B * S, H * 64), float("nan"), dtype=BF, device=dev) for _ in range(2))
    lse_h, lse_d = (torch.full((B, H, S), float("nan"), dtype=torch.float32, device=dev) for _ in range(2))
    ops.attention_fwd(dq, None, out_h, B, S, H, s_pad, 1.0 / LOG2E, lse=lse_h, score_bound=bound)
    ops.attention_fwd(dq, None, out_d, B, S, H, s_pad, 1.0 / LOG2E, lse=lse_d, score_bound_dev=bdev)
    import os
    if os.environ.get("ORV_ATTN_M16") == "1":      # the opt-in 16x16x32 kernel serves the host-side dispatch only: same values, other rounding
        close(out_d, out_h.float().cpu(), rtol=1.6e-2, afrac=4e-3)
    else:
        assert torch.equal(out_h, out_d) and torch.equal(lse_h, lse_d)
    close(out_d, ref)


def _split_case(B, S, H, seed, expect_split=True):
    """bounded attention with and without the key-split workspace on random q' | k' | v (post-LayerNorm scale)."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(seed)
    D = H * 64
    dq = (torch.randn(B * S, 3 * D, generator=g) * 0.9).to(dev, BF)
    nb = ops.attention_ws_bytes(B, S, H)
    assert (nb > 0) == expect_split, nb
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    outs = []
    for use_ws in (None, ws, ws):
        out = torch.full((B * S, D), float("nan"), dtype=BF, device=dev)
        lse = torch.full((B, H, S), float("nan"), dtype=torch.float32, device=dev)
        ops.attention_fwd(dq, None, out, B, S, H, 0, 1.0 / LOG2E, lse=lse, score_bound=30.0, ws=use_ws)
        outs.append((out, lse))
    (o0, l0), (o1, l1), (o2, l2) = outs
    assert torch.isfinite(o1.float()).all() and torch.isfinite(l1).all()
    assert torch.equal(o1, o2) and torch.equal(l1, l2)                 # fixed part order: deterministic
    close(o1, o0.float().cpu(), rtol=1.6e-2, afrac=4e-3)               # split items differ in fp32 summation order only
    assert (l1 - l0).abs().max().item() <= 2e-3
    return dq, o1, l1, nb


def test_attention_key_split_last_round_at_the_headline_shape():
    """VERDICT r3 #2(a), built and measured, opt-in (ORV_ATTN_SPLIT=1, read once per process: child process here): 1560 workgroups on
    512 slots - the 24 items beyond three full rounds are cut into 8 key ranges each whose unnormalised partials add under the
    shift-free softmax (orv_attention_fwd_bounded_ws).  Same results as the unsplit kernel to rounding, bit-reproducible, and the split
    heads (the last two of the last clip) against the fp32 softmax over the same bf16 operands."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORV_ATTN_SPLIT") != "1":
        from orv_amd import ops
        assert ops.attention_ws_bytes(4, 3226, 30) == 0                      # off by default: B = 4 stays bit-identical to 4 x B = 1
        env = dict(os.environ, ORV_ATTN_SPLIT="1")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "key_split_last_round", "-p", "no:cacheprovider"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        return
    B, S, H = 4, 3226, 30
    dq, out, lse, nb = _split_case(B, S, H, seed=11)
    assert nb >= 24 * 8 * (256 * 64 + 256) * 4
    t = dq.view(B, S, 3, H, 64)
    for b, h in ((3, 29), (3, 28), (0, 0)):
        qh, kh, vh = (t[b, :, i, h].float().cpu() for i in range(3))
        sc = (qh @ kh.T) * math.log(2.0)
        ref = torch.softmax(sc, dim=-1) @ vh
        close(out.view(B, S, H, 64)[b, :, h], ref)
        close(lse[b, h], torch.logsumexp(sc, dim=-1), rtol=2e-3, afrac=2e-3)


def test_attention_key_split_small_shapes_in_a_subprocess():
    """The split path at shapes a test can sweep: ORV_ATTN_SLOTS (read once per process) shrinks the slot count the plan assumes, so
    ragged sequences, 2 ... 8 key ranges and partial last query tiles all take it."""
    import os
    import subprocess
    import sys
    shapes = {"16": [(1, 4200, 1), (1, 2200, 2), (1, 2400, 2), (2, 1111, 3)],      # 8 / 4 / 2 key ranges, not split
              "8": [(1, 700, 3), (3, 700, 2)],                                     # 2 / 2
              "40": [(2, 1600, 3)]}                                                # 6
    slots = os.environ.get("ORV_ATTN_SLOTS")
    if slots:
        from orv_amd import ops
        n_split = 0
        for seed, (B, S, H) in enumerate(shapes[slots]):
            split = ops.attention_ws_bytes(B, S, H) > 0
            n_split += split
            dq, out, lse, nb = _split_case(B, S, H, seed, expect_split=split)
            t = dq.view(B, S, 3, H, 64)
            qh, kh, vh = (t[B - 1, :, i, H - 1].float().cpu() for i in range(3))       # the last item is always a split one
            sc = (qh @ kh.T) * math.log(2.0)
            close(out.view(B, S, H, 64)[B - 1, :, H - 1], torch.softmax(sc, dim=-1) @ vh)
        assert n_split >= len(shapes[slots]) - 1
        return
    for slots in shapes:
        env = dict(os.environ, ORV_ATTN_SLOTS=slots, ORV_ATTN_SPLIT="1")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "key_split_small", "-p", "no:cacheprovider"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (slots, r.stdout[-2000:] + r.stderr[-2000:])


def test_attention_16x16x32_variant_matches_in_a_subprocess():
    """The opt-in ping-pong kernel on v_mfma_f32_16x16x32_bf16 (ORV_ATTN_M16=1, read once per process): the bounded-attention
    tests of this file, run again in a child process with the switch on, must pass against the same references."""
    import os
    import subprocess
    import sys
    if os.environ.get("ORV_ATTN_M16") == "1":
        pytest.skip("already the child")
    env = dict(os.environ, ORV_ATTN_M16="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "bound", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


# ---- round 5: packed P16 operands (include/orv_mi355.h orv_gemm_t), the d8 GEMM that reads them, their producers ----
@pytest.mark.parametrize("M,K", [(1, 32), (300, 192), (3226, 1920), (4097, 64)])
def test_pack_rows16_is_index_exact_and_invertible(M, K):
    """orv_pack_rows16 / orv_unpack_rows16: element (r, k) sits at block ((r / 16) * K / 32 + k / 32) KiB, slot ((k % 32) / 8) * 16 + r % 16,
    lane-linear - checked against the index formula itself (bit copies), padding rows zero, round trip exact."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    p = ops.pack_rows16(a, M, K)
    Mp = ops.packed_rows(M)
    assert p.shape == (Mp, K) and Mp % 256 == 0 and Mp - M < 256
    ref = torch.zeros(Mp, K, dtype=BF, device=dev)
    ref[:M] = a
    # [R, 16, C, 4, 8] (row block, row, k block, chunk, element) -> [R, C, 4, 16, 8]
    want = ref.view(Mp // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(Mp, K)
    assert torch.equal(p.view(torch.int16), want.view(torch.int16))
    assert torch.equal(ops.unpack_rows16(p, M, K).view(torch.int16), a.view(torch.int16))


@pytest.mark.parametrize("bn", [192, 128])
def test_d8_192_row_tiles_are_bit_identical_to_the_256_row_kernel(bn):
    """gemm_d8r192_kernel (round 6): the d8 kernel on 192-row tiles - the first wave of every SIMD owns two 16-row blocks, the second one -
    changes WHICH wave computes a row, not how an output element is accumulated: every epilogue (bias, GELU, gated residual with token
    groups + row scatter, fused qk LayerNorm with the Y side output, packed C) must agree with gemm_d8_kernel bit for bit, on ragged M,
    single- and multi-round grids.  Shapes whose 192-row tiling would leave the packed row slots (ceil(M / 256) * 256) are refused."""
    from orv_amd import ops
    from orv_amd._lib import lib
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(90 + bn)
    N = bn * 3

    def same(a, b, what):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (what, (a.float() - b.float()).abs().max().item())

    def pinned(bm, fn):
        lib().orv_gemm_force_tile(5, bm, bn)
        try:
            return fn()
        finally:
            lib().orv_gemm_force_tile(0, 0, 0)

    for M, K in [(100, 384), (3226, 1920), (3226 * 2, 384), (12904, 576), (70001, 192)]:
        assert -(-M // 192) * 192 <= -(-M // 256) * 256
        A = torch.randn(M, K, device=dev, generator=g).to(BF)
        W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(BF)
        bias = torch.randn(N, device=dev, generator=g).to(BF)
        R = torch.randn(M, N, device=dev, generator=g).to(BF)
        Ap = ops.pack_rows16(A, M, K)
        assert pinned(192, lambda: ops.gemm_kernel_name(M, N, K, 0, a_packed=True)) == f"gemm_d8r192_kernel<{bn}, 0>"
        for epi in (0, 1, 2):
            def run():
                C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
                Y = torch.full((M, N), float("nan"), dtype=BF, device=dev)
                ops.gemm(Ap, W, bias, C, M, N, K, epilogue=epi, Y=Y, a_packed=True, **(dict(R=R, ldr=N) if epi == 2 else {}))
                return C, Y
            (c0, y0), (c1, y1) = pinned(256, run), pinned(192, run)
            assert torch.isfinite(c1.float()).all()
            same(c0, c1, (bn, M, K, epi, "C")), same(y0, y1, (bn, M, K, epi, "Y"))
        # packed C (the FFN1 -> FFN2 hand-off): compare the unpacked rows [0, M)
        def run_pc():
            Cp = torch.zeros(ops.packed_rows(M), N, dtype=BF, device=dev)
            ops.gemm(Ap, W, bias, Cp, M, N, K, epilogue=1, a_packed=True, c_packed=True)
            return ops.unpack_rows16(Cp, M, N)
        same(pinned(256, run_pc), pinned(192, run_pc), (bn, M, K, "packed C"))
    # 192-row tiles that would run past the packed row slots are refused (M = 500: 3 x 192 = 576 > 512)
    with pytest.raises(RuntimeError):
        pinned(192, lambda: ops.gemm_kernel_name(500, N, 384, 0, a_packed=True) or (_ for _ in ()).throw(RuntimeError("no kernel")))
    # gated residual with token groups (gate rows change inside 16-row blocks) + row scatter into a joint buffer
    seq, n_text, per_group, Bn = 500, 19, 37, 3
    Mv, K = (seq - n_text) * Bn, 384
    assert -(-Mv // 192) * 192 <= -(-Mv // 256) * 256
    A = torch.randn(Mv, K, device=dev, generator=g).to(BF)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(BF)
    bias = torch.randn(N, device=dev, generator=g).to(BF)
    n_groups = 1 + (seq - n_text + per_group - 1) // per_group
    gate = torch.randn(Bn, n_groups, N, device=dev, generator=g)
    base = torch.randn(Bn * seq, N, device=dev, generator=g).to(BF)
    Ap = ops.pack_rows16(A, Mv, K)

    def fn_gate():
        C = base.clone()
        ops.gemm(Ap, W, bias, C, Mv, N, K, epilogue=2, R=C, ldr=N, gate=gate, gate_b=n_groups * N, gate_g=N,
                 grp=ops.Groups(seq, n_text, per_group), cmap=ops.RowMap(seq - n_text, seq, n_text), a_packed=True)
        return C
    same(pinned(256, fn_gate), pinned(192, fn_gate), (bn, "gate + cmap"))
    # fused qk LayerNorm (a wave owns whole rows, 64 columns = one head, in both row layouts)
    heads = 2 if bn == 128 else 3
    Nq, M, K = 3 * heads * 64, 3226, 384
    A = torch.randn(M, K, device=dev, generator=g).to(BF)
    W = (torch.randn(Nq, K, device=dev, generator=g) / K ** 0.5).to(BF)
    bias = torch.randn(Nq, device=dev, generator=g).to(BF)
    aff = [torch.randn(64, device=dev, generator=g).to(BF) for _ in range(4)]
    Ap = ops.pack_rows16(A, M, K)

    def fn_qk():
        C = torch.full((M, Nq), float("nan"), dtype=BF, device=dev)
        Y = torch.full((M, Nq), float("nan"), dtype=BF, device=dev)
        ops.gemm(Ap, W, bias, C, M, Nq, K, epilogue=4, Y=Y, qknorm=(aff[0], aff[1], aff[2], aff[3], 1e-6, 0.18, heads), a_packed=True)
        return C, Y
    (c0, y0), (c1, y1) = pinned(256, fn_qk), pinned(192, fn_qk)
    same(c0, c1, (bn, "qk LayerNorm C")), same(y0, y1, (bn, "qk LayerNorm Y"))


@pytest.mark.parametrize("N,K,epi", [(1920, 7680, 2), (1920, 1920, 2), (5760, 1920, 4), (7680, 1920, 1)])
def test_d8_kernels_are_deterministic_under_repetition(N, K, epi):
    """Race screen for the d8 kernels at the shapes the one-clip step takes them (gemm_d8r192_kernel: M = 3226) and at the headline shape
    (gemm_d8_kernel: M = 12904): hand-counted vmcnt waits, A registers in flight across the tile boundary, branch-free cursor wraps, one
    barrier per K-tile - 20 back-to-back launches must agree bit for bit (a read of an in-flight register or a buffer reused one barrier early
    shows up as run-to-run differences long before it shows up as a wrong mean)."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(N + K + epi)
    for M in (3226, 12904):
        name = ops.gemm_kernel_name(M, N, K, epi, a_packed=True, c_packed=(epi == 1))
        assert name is not None and name.startswith("gemm_d8"), name
        x = torch.randn(M, K, generator=g).to(dev, BF)
        W = (torch.randn(N, K, generator=g) * 0.03).to(dev, BF)
        bias = (torch.randn(N, generator=g) * 0.1).to(dev, BF)
        xp = ops.pack_rows16(x, M, K)
        kw = {}
        if epi == 2:
            kw = dict(R=torch.randn(M, N, generator=g).to(dev, BF), ldr=N, gate=torch.randn(M // 3226, 6, N, generator=g).to(dev),
                      gate_b=6 * N, gate_g=N, grp=ops.groups(3226, 226, 600))
        if epi == 4:
            one = torch.ones(64, dtype=BF, device=dev)
            kw = dict(qknorm=(one, None, one, None, 1e-6, 0.18, N // 192))
        outs = []
        for _ in range(20):
            out = torch.empty(ops.packed_rows(M) if epi == 1 else M, N, dtype=BF, device=dev)
            ops.gemm(xp, W, bias, out, M, N, K, epilogue=epi, a_packed=True, c_packed=(epi == 1), **kw)
            outs.append(out)
        torch.cuda.synchronize()
        first = ops.unpack_rows16(outs[0], M, N) if epi == 1 else outs[0]
        assert torch.isfinite(first.float()).all(), (name, M)
        for o in outs[1:]:
            oo = ops.unpack_rows16(o, M, N) if epi == 1 else o
            assert torch.equal(oo, first), (name, M)


@pytest.mark.parametrize("bn", [256, 192])
def test_d8_gemm_on_packed_a_is_bit_identical_to_t8(bn):
    """gemm_d8_kernel (A straight to registers from the packed layout, W through four LDS buffers, one barrier per K-tile) accumulates every
    output element in gemm_t8_kernel's order (K-tiles ascending, k halves 0 / 1, the same k chunk per lane group) and runs the same epilogue
    arithmetic: bias, GELU, gated residual with per-token-group gates straddling 16-row blocks, row scatter, r_mod, GELU adjoint, the Y side
    output - bit-identical; the fused qk LayerNorm reduces in another lane order (8-lane DPP sums vs half-swaps): tolerance.  Ragged M,
    one K-tile triple (K = 192) up to K = 1920, single- and multi-round grids."""
    from orv_amd import ops
    from orv_amd._lib import lib
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(50 + bn)
    N = bn * 3

    def same(a, b, what):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (what, (a.float() - b.float()).abs().max().item())

    for M, K in [(100, 384), (3226 * 2, 384), (12904, 1920), (70000, 384)]:      # K % 128 == 0 (t8) and K % 192 == 0 (d8)
        A = torch.randn(M, K, device=dev, generator=g).to(BF)
        W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(BF)
        bias = torch.randn(N, device=dev, generator=g).to(BF)
        R = torch.randn(M, N, device=dev, generator=g).to(BF)
        Ap = ops.pack_rows16(A, M, K)
        assert ops.gemm_kernel_name(M, N, K, 0, a_packed=True).startswith("gemm_d8")        # gemm_d8_kernel or gemm_d8r192_kernel
        for epi in (0, 1, 2, 3):
            outs = []
            for packed in (False, True):
                C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
                Y = torch.full((M, N), float("nan"), dtype=BF, device=dev)
                lib().orv_gemm_force_tile(5 if packed else 3, 256, bn)       # N = 3 bn divides by both widths: pin the one under test
                try:
                    ops.gemm(Ap if packed else A, W, None if epi == 3 else bias, C, M, N, K, epilogue=epi, Y=Y, a_packed=packed,
                             **(dict(R=R, ldr=N) if epi >= 2 else {}))
                finally:
                    lib().orv_gemm_force_tile(0, 0, 0)
                outs.append((C, Y))
            same(outs[0][0], outs[1][0], (bn, M, K, epi, "C")), same(outs[0][1], outs[1][1], (bn, M, K, epi, "Y"))
    # gated residual with token groups (gate rows change inside 16-row blocks), row scatter into a joint buffer, r_mod residual
    # K = 192 (one trip of the three-K-tile loop: the t8 kernel cannot run it) against whatever the row-major cost model picks: tolerance
    M, K = 777, 192
    A = torch.randn(M, K, device=dev, generator=g).to(BF)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(BF)
    bias = torch.randn(N, device=dev, generator=g).to(BF)
    c0 = torch.empty(M, N, dtype=BF, device=dev)
    c1 = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    ops.gemm(A, W, bias, c0, M, N, K, epilogue=1)
    lib().orv_gemm_force_tile(5, 256, bn)
    try:
        ops.gemm(ops.pack_rows16(A, M, K), W, bias, c1, M, N, K, epilogue=1, a_packed=True)
    finally:
        lib().orv_gemm_force_tile(0, 0, 0)
    assert ((c1.float() - c0.float()).abs() <= 1.0e-2 * c0.float().abs() + 1.5e-2).all()
    seq, n_text, per_group, Bn = 500, 19, 37, 3
    Mv = (seq - n_text) * Bn
    K = 384
    A = torch.randn(Mv, K, device=dev, generator=g).to(BF)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(BF)
    bias = torch.randn(N, device=dev, generator=g).to(BF)
    n_groups = 1 + (seq - n_text + per_group - 1) // per_group
    gate = torch.randn(Bn, n_groups, N, device=dev, generator=g)
    base = torch.randn(Bn * seq, N, device=dev, generator=g).to(BF)
    Rm = torch.randn(97, N, device=dev, generator=g).to(BF)
    Ap = ops.pack_rows16(A, Mv, K)

    def both(fn):
        res = []
        for packed in (False, True):
            lib().orv_gemm_force_tile(5 if packed else 3, 256, bn)
            try:
                res.append(fn(Ap if packed else A, packed))
            finally:
                lib().orv_gemm_force_tile(0, 0, 0)
        return res

    def fn_gate(a, packed):
        C = base.clone()
        ops.gemm(a, W, bias, C, Mv, N, K, epilogue=2, R=C, ldr=N, gate=gate, gate_b=n_groups * N, gate_g=N,
                 grp=ops.Groups(seq, n_text, per_group), cmap=ops.RowMap(seq - n_text, seq, n_text), a_packed=packed)
        return C

    def fn_rmod(a, packed):
        C = torch.full((Mv, N), float("nan"), dtype=BF, device=dev)
        ops.gemm(a, W, bias, C, Mv, N, K, epilogue=2, R=Rm, ldr=N, r_mod=97, a_packed=packed)
        return C
    same(*both(fn_gate), (bn, "gate + cmap")), same(*both(fn_rmod), (bn, "r_mod"))
    # fused qk LayerNorm (both widths: the d8 kernel normalises 64-column groups at any BN; t8 only at BN = 256)
    heads = 4 if bn == 256 else 6
    Nq, M, K = 3 * heads * 64, 3226, 384
    A = torch.randn(M, K, device=dev, generator=g).to(BF)
    W = (torch.randn(Nq, K, device=dev, generator=g) / K ** 0.5).to(BF)
    bias = torch.randn(Nq, device=dev, generator=g).to(BF)
    aff = [torch.randn(64, device=dev, generator=g).to(BF) for _ in range(4)]
    Ap = ops.pack_rows16(A, M, K)

    def fn_qk(a, packed):
        C = torch.full((M, Nq), float("nan"), dtype=BF, device=dev)
        Y = torch.full((M, Nq), float("nan"), dtype=BF, device=dev)
        ops.gemm(a, W, bias, C, M, Nq, K, epilogue=4, Y=Y, qknorm=(aff[0], aff[1], aff[2], aff[3], 1e-6, 0.18, heads), a_packed=packed)
        return C, Y
    (c0, y0) = fn_qk(A, False)
    lib().orv_gemm_force_tile(5, 256, bn)
    try:
        assert ops.gemm_kernel_name(M, Nq, K, 4, a_packed=True) == f"gemm_d8_kernel<{bn}, 4>"
        (c1, y1) = fn_qk(Ap, True)
    finally:
        lib().orv_gemm_force_tile(0, 0, 0)
    assert torch.isfinite(c1.float()).all() and torch.isfinite(y1.float()).all()
    assert ((y1.float() - y0.float()).abs() <= 1.0e-2 * y0.float().abs() + 1.5e-2).all()
    assert ((c1.float() - c0.float()).abs() <= 2e-2 * c0.float().abs() + 6e-2).all()
    # q | k against gemm_t8_kernel<256, 4> pinned (on 256 and on 192 rows): the d8 epilogue sums the LayerNorm statistics in t8's order -
    # BIT-identical, so that a one-clip call (q | k | v on d8) and a four-clip call (the t8 pair) give the same clip
    qk = 2 * heads * 64
    Cr = torch.full((M, qk), float("nan"), dtype=BF, device=dev)
    Yr = torch.full((M, qk), float("nan"), dtype=BF, device=dev)
    for bm in (256, 192):
        lib().orv_gemm_force_tile(3, bm, 256)
        try:
            ops.gemm(A, W[:qk].contiguous(), bias[:qk].contiguous(), Cr, M, qk, K, epilogue=4, Y=Yr,
                     qknorm=(aff[0], aff[1], aff[2], aff[3], 1e-6, 0.18, heads))
        finally:
            lib().orv_gemm_force_tile(0, 0, 0)
        assert torch.equal(y1[:, :qk], Yr) and torch.equal(c1[:, :qk], Cr), bm


def test_packed_c_from_the_gelu_epilogues_and_the_attention_kernel():
    """The producers of the packed layout: gemm_t8_kernel<256, 1> (FFN1 of the product path) and gemm_d8_kernel<.., 0 / 1> store a block
    pair as one lane-linear KiB, attn_fwd_pp_kernel its 16-byte output pieces into slot order - each unpacks to exactly what the row-major
    form of the same call wrote."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(77)
    for M, N, K in [(300, 768, 384), (12904, 7680, 1920), (3226, 1024, 256)]:
        A = torch.randn(M, K, device=dev, generator=g).to(BF)
        W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(BF)
        bias = torch.randn(N, device=dev, generator=g).to(BF)
        C = torch.empty(M, N, dtype=BF, device=dev)
        ops.gemm(A, W, bias, C, M, N, K, epilogue=1)
        name = ops.gemm_kernel_name(M, N, K, 1, c_packed=True)
        assert name in ("gemm_t8_kernel<256, 1>", "gemm_t8r192_kernel<256, 1>"), name      # 192-row tiles where their count fits the CUs better
        Cp = torch.full((ops.packed_rows(M), N), float("nan"), dtype=BF, device=dev)
        ops.gemm(A, W, bias, Cp, M, N, K, epilogue=1, c_packed=True)
        # the row-major call may have taken another kernel (cost model); compare against the 256-row t8 kernel pinned
        from orv_amd._lib import lib
        lib().orv_gemm_force_tile(3, 256, 256)
        try:
            ops.gemm(A, W, bias, C, M, N, K, epilogue=1)
        finally:
            lib().orv_gemm_force_tile(0, 0, 0)
        assert torch.equal(ops.unpack_rows16(Cp, M, N).view(torch.int16), C.view(torch.int16)), (M, N, K)
        if K % 192 == 0:
            Ap = ops.pack_rows16(A, M, K)
            for epi in (0, 1):
                Cp2 = torch.full((ops.packed_rows(M), N), float("nan"), dtype=BF, device=dev)
                C2 = torch.empty(M, N, dtype=BF, device=dev)
                ops.gemm(Ap, W, bias, Cp2, M, N, K, epilogue=epi, a_packed=True, c_packed=True)
                ops.gemm(Ap, W, bias, C2, M, N, K, epilogue=epi, a_packed=True)
                assert torch.equal(ops.unpack_rows16(Cp2, M, N).view(torch.int16), C2.view(torch.int16)), (M, N, K, epi)
    # attention
    for B, S, H in [(2, 333, 2), (1, 3226, 3), (3, 100, 5)]:
        D = H * 64
        qkv = (torch.randn(B * S, 3 * D, device=dev, generator=g) * 0.7).to(BF)
        bound = float((qkv[:, :D].float().abs().max() * qkv[:, D:2 * D].float().abs().max() * 64).item())       # crude, holds
        bound = min(bound, 89.0)
        qs = qkv.clone()
        qs[:, :D] = (qkv[:, :D].float() * (bound / 64 / max(1e-6, (qkv[:, :D].float().abs().max() * qkv[:, D:2 * D].float().abs().max()).item()))).to(BF)
        out = torch.full((B * S, D), float("nan"), dtype=BF, device=dev)
        ops.attention_fwd(qs, None, out, B, S, H, 0, 1.0 / LOG2E, score_bound=bound)
        assert ops.attention_packed_ok(bound, 1.0 / LOG2E)
        outp = torch.full((ops.packed_rows(B * S), D), float("nan"), dtype=BF, device=dev)
        ops.attention_fwd(qs, None, outp, B, S, H, 0, 1.0 / LOG2E, score_bound=bound, out_packed=True)
        assert torch.isfinite(out.float()).all()
        assert torch.equal(ops.unpack_rows16(outp, B * S, D).view(torch.int16), out.view(torch.int16)), (B, S, H)


def test_attention_w64_variant_passes_the_attention_tests():
    """attn_fwd_w64_kernel (64 query rows per wave: every K / V^T fragment feeds two MFMAs; two 4-wave groups with their own 256-row item per
    workgroup, the ping-pong barrier choreography between the groups; opt-in ORV_ATTN_W64=1 - measured slower than the 8-wave kernel,
    profiles/r5_attention_w64.txt): the shift-free attention tests and the packed-output test of this file, run in a fresh interpreter with
    the switch on."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_kernels.py", "-x", "-q", "-k",
                        "fixed_shift or packed_c_from or gamma_product"], cwd=root, env=dict(os.environ, ORV_ATTN_W64="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_shift_free_attention_near_its_limit_with_strongly_negative_rows():
    """the shift-free kernel takes P = exp2(s) for |s| <= 90 with no shift.  Rows whose scores ALL sit near -bound give P ~ 2^-85
    (a normal bf16 / fp32 number), P . v products ~ 2^-92 and a row sum ~ S 2^-85: nothing may flush before the 1 / l rescale.  A bound of 88,
    query rows built so that every score is -85 + small next to ordinary rows, against the online kernel (which subtracts the running max and
    never sees such magnitudes) and the fp32 reference."""
    from orv_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    B, S, H = 1, 700, 2
    D = H * 64
    qkv = torch.randn(B * S, 3 * D, generator=g) * 0.1
    qkv[:, 2 * D:] = torch.randn(B * S, D, generator=g)                    # v: ordinary magnitudes
    for h in range(H):
        qkv[:, D + h * 64] = 8.0                                           # every key: component 0 = 8
        qkv[::3, h * 64] = -10.625                                         # every third query: component 0 = -85 / 8  ->  score = -85 + O(0.1)
    qkv = q(qkv).to(dev, BF)
    qf, kf = qkv[:, :D].float().view(S, H, 64), qkv[:, D:2 * D].float().view(S, H, 64)
    s_all = torch.einsum("qhd,khd->hqk", qf, kf)
    bound = 88.0
    assert s_all.abs().max().item() <= bound and s_all[:, ::3].max().item() < -80.0
    out = torch.full((B * S, D), float("nan"), dtype=BF, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention_fwd(qkv, None, out, B, S, H, 0, 1.0 / LOG2E, lse=lse, score_bound=bound)
    out_online = torch.empty_like(out)
    lse_online = torch.empty_like(lse)
    ops.attention_fwd(qkv, None, out_online, B, S, H, 0, 1.0 / LOG2E, lse=lse_online)
    # the kernel's scores are in log2 units (q pre-multiplied): P = 2^s = exp(s ln 2)
    ref = torch.einsum("hqk,khd->qhd", torch.softmax(s_all * 0.6931471805599453, dim=-1), qkv[:, 2 * D:].float().view(S, H, 64)).reshape(S, D)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    close(out, ref)
    close(out, out_online.float().cpu(), rtol=1.6e-2, afrac=4e-3)
    assert (lse - lse_online).abs().max().item() <= 2e-3
