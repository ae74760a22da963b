"""Training path of the DiT: forward that saves activations + hand-written backward through every op, exposed to
``torch.autograd`` as ONE custom Function so that the reference's train step
(/root/reference/orv/pipeline/train_cogvideox_control_to_video_sft.py:1051-1104: ``transformer(...)`` -> loss ->
``accelerator.backward(loss)`` -> clip -> ``optimizer.step()``) and DDP's gradient hooks work unchanged.

Design: no recomputation (288 GB of HBM holds every saved activation: ~0.94 GB per block at B=4, 28 GB for 30 blocks);
dgrad/wgrad reuse the NT MFMA GEMM on K-contiguous operands produced by ``orv_transpose_bf16``
(dX = dY . W via W^T; dW = dY^T . X via dY^T, X^T), with the GELU adjoint and gradient accumulation fused as epilogues;
the attention adjoint is two MFMA passes (dQ; dK,dV); LayerNorm / gate / modulation-table adjoints are HBM-bound kernels
with fp32 atomics for the per-frame sums.  Parameters and gradients are bf16 like the reference's (`:384`, no fp32 master).

The tiny conditioning branch (time embedding, ActionEmbed, SiLU of [B, T, 512] tensors: < 1e-4 of the FLOPs) differentiates
its activations with plain torch ops on those few-kB tensors; all matmuls, including theirs, run in HIP kernels.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import ops

BF16 = torch.bfloat16
LOG2E = 1.4426950408889634


def _zeros_like_grad(p):
    return torch.zeros_like(p, dtype=BF16)


class _Saved:
    """Activations of one forward, consumed by exactly one backward."""
    pass


def _acc_grad(store: Dict, p: torch.nn.Parameter) -> torch.Tensor:
    """bf16 gradient accumulator for parameter p (zero-initialised once per backward)."""
    g = store.get(id(p))
    if g is None:
        g = torch.zeros_like(p, dtype=BF16)
        store[id(p)] = g
    return g


def _wgrad(dY2d, X2d, dW, M, N, K):
    """dW[N,K] += dY[M,N]^T . X[M,K]  through two transposes and the NT GEMM (contraction over the M rows)."""
    dYT = ops.transpose(dY2d, M, N)           # [N, M_pad]
    XT = ops.transpose(X2d, M, K)             # [K, M_pad]
    if K % 64 == 0:
        ops.gemm(dYT, XT, None, dW, N, K, dYT.shape[1], epilogue=2, R=dW, ldr=K)
    else:                                     # odd widths (tiny test configs only): pad the output columns
        kp = (K + 63) // 64 * 64
        XTp = torch.zeros(kp, XT.shape[1], dtype=BF16, device=XT.device)
        XTp[:K] = XT
        tmp = torch.empty(N, kp, dtype=BF16, device=XT.device)
        ops.gemm(dYT, XTp, None, tmp, N, kp, dYT.shape[1])
        dW.add_(tmp[:, :K])


def _pad_k(a, w):
    K = a.shape[1]
    if K % 64:
        pad = 64 - K % 64
        a, w = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(w, (0, pad))
    return a.contiguous(), w.contiguous()


def _dgrad(dY2d, W, dX, M, N, K, epilogue=0, R=None):
    """dX[M,K] = dY[M,N] . W[N,K]  (contraction over N) via W^T [K, N_pad]."""
    WT = ops.transpose(W.detach().contiguous(), N, K)          # [K, N_pad]
    npad = WT.shape[1]
    a = dY2d if npad == N else torch.nn.functional.pad(dY2d, (0, npad - N)).contiguous()
    ops.gemm(a, WT, None, dX, M, K, npad, epilogue=epilogue, R=R, ldr=K)
    return dX


def forward_train(model, hidden_states, encoder_hidden_states, controls, timestep, ofs=None, image_rotary_emb=None):
    """Same arithmetic as ``CogVideoXTransformer3DModelTraj.forward`` (inference kernels), keeping what backward needs."""
    c = model.config
    dev = hidden_states.device
    if c.multiview or c.visual_guidance:
        raise NotImplementedError("training path covers the single-view trajectory model (BASELINE config 3) for now")
    B, T, C, Hh, Ww = hidden_states.shape
    p, pt = c.patch_size, c.patch_size_t
    D, heads, E = model.inner_dim, c.num_attention_heads, c.time_embed_dim
    mod_text = bool(c.modulate_encoder_hidden_states)
    Nt = encoder_hidden_states.shape[1] if mod_text else 0
    Tq = T // (pt or 1)
    P = (Hh // p) * (Ww // p)
    Nv = Tq * P
    S = Nt + Nv
    M = B * S
    s_pad = (S + 63) // 64 * 64
    sv = _Saved()
    sv.dims = dict(B=B, T=T, Hh=Hh, Ww=Ww, D=D, heads=heads, E=E, Nt=Nt, Nv=Nv, S=S, M=M, s_pad=s_pad, P=P, mod_text=mod_text)
    e = lambda *shape, dt=BF16: torch.empty(*shape, dtype=dt, device=dev)

    # conditioning branch (tiny): matmuls in HIP, activations' adjoints later via torch on [B, T, E] tensors
    tvec = torch.as_tensor(timestep, device=dev).reshape(-1).to(torch.float32)
    if tvec.numel() == 1 and B > 1:
        tvec = tvec.expand(B)
    te = model.time_embedding
    sv.t_emb = ops.timestep_embedding(tvec.contiguous(), D, c.flip_sin_to_cos, c.freq_shift)
    sv.te_u1 = ops.skinny_linear(sv.t_emb, te.linear_1.weight, te.linear_1.bias)                 # pre-SiLU
    sv.te_h1 = torch.nn.functional.silu(sv.te_u1.float()).to(BF16)
    temb = ops.skinny_linear(sv.te_h1, te.linear_2.weight, te.linear_2.bias)
    if model.ofs_embedding is not None:
        raise NotImplementedError("ofs embedding (CogVideoX1.5) training adjoint not built yet")
    sv.temb = temb

    pe = model.patch_embed
    sv.tokens = ops.patchify(hidden_states.to(BF16), None, p, pt)
    x = e(M, D)
    a2, w2 = _pad_k(sv.tokens.view(B * Nv, -1), pe.proj.weight.reshape(D, -1))
    pos = pe.video_pos_table(T, Hh, Ww, dev)
    ops.gemm(a2, w2, pe.proj.bias, x, B * Nv, D, a2.shape[1], epilogue=2 if pos is not None else 0, R=pos, r_mod=Nv, ldr=D,
             cmap=ops.rowmap(Nv, S, Nt))
    if mod_text:
        sv.text2d = encoder_hidden_states.to(BF16).reshape(B * Nt, -1).contiguous()
        a2, w2 = _pad_k(sv.text2d, pe.text_proj.weight)
        ops.gemm(a2, w2, pe.text_proj.bias, x, B * Nt, D, a2.shape[1], cmap=ops.rowmap(Nt, S, 0))

    action_emb = is_mask = None
    actions = controls.get('actions', None)
    sv.has_actions = actions is not None
    if actions is not None:
        actions = actions.to(device=dev)
        res = (actions.size(1) + 1) % 4
        padf = 4 - res if res > 0 else 0
        if padf:
            actions = torch.cat([actions.new_zeros((actions.shape[0], padf, actions.shape[2])), actions], dim=1)
        ae = model.action_embed
        xa = torch.cat([torch.zeros_like(actions[:, :1]), actions], dim=1)
        xa = xa.reshape(B, (actions.shape[1] + 1) // ae.compress_ratio, -1)
        if ae.patch_size_t > 1:
            xa = xa.reshape(B, xa.shape[1] // ae.patch_size_t, -1)
        Ta = xa.shape[1]
        sv.ae_in = xa.reshape(B * Ta, -1).to(BF16).contiguous()
        sv.ae_u = ops.skinny_linear(sv.ae_in, ae.mlp[0].weight, ae.mlp[0].bias)
        sv.ae_h = torch.nn.functional.gelu(sv.ae_u.float(), approximate="tanh").to(BF16)
        emb = ops.skinny_linear(sv.ae_h, ae.mlp[3].weight, ae.mlp[3].bias).view(B, Ta, E)
        is_mask = ae.forced_mask.to(dev, torch.bool) if ae.forced_mask is not None else torch.rand(B, device=dev) < 0.1
        if ae.mask:
            emb = torch.where(is_mask[:, None, None], ae.mask_embed.weight[None].to(emb.dtype), emb)
        sv.is_mask = is_mask
        action_emb = emb.contiguous()
    sv.action_emb = action_emb
    Ta = action_emb.shape[1] if action_emb is not None else 1
    per_group = Nv // Ta if action_emb is not None else 0
    G = 1 + Ta
    sv.dims.update(Ta=Ta, G=G, per_group=per_group)
    grp = ops.groups(S, Nt, per_group)
    L = c.num_layers
    ptr = model._pointer_tables(dev)
    mod = ops.modulation_tables(temb, action_emb, ptr["w_blk"], ptr["b_blk"], 2 * L, B, Ta, E, 3 * D, mod_text)
    modf = ops.modulation_tables(temb, action_emb, ptr["w_out"], ptr["b_out"], 1, B, Ta, E, 2 * D, False)[0]
    sv.mod, sv.modf = mod, modf
    rope = None
    if image_rotary_emb is not None:
        rope = tuple(r.to(device=dev, dtype=torch.float32).contiguous() for r in image_rotary_emb)
    sv.rope = rope

    mb, mg = G * 3 * D, 3 * D
    scale = 1.0 / math.sqrt(c.attention_head_dim)
    work = e(M, 3 * D)
    vT = torch.zeros(B, heads, 64, s_pad, dtype=BF16, device=dev)
    sv.layers = []
    for i, blk in enumerate(model.transformer_blocks):
        m1, m2 = mod[2 * i], mod[2 * i + 1]
        at = blk.attn1
        ly = _Saved()
        ly.x0 = x
        ly.xn1 = e(M, D)
        ops.layernorm_modulate(x, ly.xn1, blk.norm1.norm.weight, blk.norm1.norm.bias, m1[..., D:2 * D], m1[..., :D], mb, mg,
                               grp, B, D, c.norm_eps)
        wqkv, bqkv = at.packed_qkv()
        ly.qkv_raw = e(M, 3 * D)
        ops.gemm(ly.xn1, wqkv, bqkv, ly.qkv_raw, M, 3 * D, D)
        work.copy_(ly.qkv_raw)
        ops.qkv_prep(work, vT, at.norm_q.weight, at.norm_q.bias, at.norm_k.weight, at.norm_k.bias, rope, B, S, heads, Nt,
                     s_pad, at.eps, q_premul=scale * LOG2E)
        ly.att = e(M, D)
        ly.lse = e(B, heads, S, dt=torch.float32)
        ops.attention_fwd(work, vT, ly.att, B, S, heads, s_pad, 1.0 / LOG2E, lse=ly.lse)
        ly.x1, ly.y1 = e(M, D), e(M, D)
        ops.gemm(ly.att, at.to_out[0].weight, at.to_out[0].bias, ly.x1, M, D, D, epilogue=2, R=x, ldr=D, gate=m1[..., 2 * D:],
                 gate_b=mb, gate_g=mg, grp=grp, Y=ly.y1, ldy=D)
        ly.xn2 = e(M, D)
        ops.layernorm_modulate(ly.x1, ly.xn2, blk.norm2.norm.weight, blk.norm2.norm.bias, m2[..., D:2 * D], m2[..., :D], mb,
                               mg, grp, B, D, c.norm_eps)
        f0, f2 = blk.ff.net[0].proj, blk.ff.net[2]
        FF = f0.weight.shape[0]
        ly.u, ly.h = e(M, FF), e(M, FF)
        ops.gemm(ly.xn2, f0.weight, f0.bias, ly.h, M, FF, D, epilogue=1, Y=ly.u, ldy=FF)
        x2, ly.y2 = e(M, D), e(M, D)
        ops.gemm(ly.h, f2.weight, f2.bias, x2, M, D, FF, epilogue=2, R=ly.x1, ldr=D, gate=m2[..., 2 * D:], gate_b=mb,
                 gate_g=mg, grp=grp, Y=ly.y2, ldy=D)
        x = x2
        sv.layers.append(ly)
    sv.x_last = x
    gv = ops.groups(Nv, 0, per_group)
    sv.vis, sv.vis2 = e(B * Nv, D), e(B * Nv, D)
    ops.layernorm_modulate(x, sv.vis, model.norm_final.weight, model.norm_final.bias, None, None, 0, 0, gv, B, D, c.norm_eps,
                           xmap=ops.rowmap(Nv, S, Nt))
    no = model.norm_out
    ops.layernorm_modulate(sv.vis, sv.vis2, no.norm.weight, no.norm.bias, modf[..., D:], modf[..., :D], G * 2 * D, 2 * D, gv, B,
                           D, c.norm_eps)
    fo = model.proj_out.weight.shape[0]
    if fo % 64:
        raise NotImplementedError("proj_out width must be a multiple of 64 on the training path")
    out_tok = e(B * Nv, fo)
    ops.gemm(sv.vis2, model.proj_out.weight, model.proj_out.bias, out_tok, B * Nv, fo, D)
    c_out = fo // (p * p * (pt or 1))
    output = ops.unpatchify(out_tok, B, T, c_out, Hh, Ww, p, pt)
    actions_recon = None
    return output, is_mask, actions_recon, sv


def backward(model, sv, dout) -> Dict[int, torch.Tensor]:
    """Gradients of every trainable parameter (bf16, keyed by id(param)) given dL/d(sample)."""
    c = model.config
    d = sv.dims
    B, T, Hh, Ww, D, heads, E = d["B"], d["T"], d["Hh"], d["Ww"], d["D"], d["heads"], d["E"]
    Nt, Nv, S, M, s_pad, G, Ta, per_group = d["Nt"], d["Nv"], d["S"], d["M"], d["s_pad"], d["G"], d["Ta"], d["per_group"]
    mod_text = d["mod_text"]
    dev = dout.device
    p, pt = c.patch_size, c.patch_size_t
    grads: Dict[int, torch.Tensor] = {}
    e = lambda *shape, dt=BF16: torch.empty(*shape, dtype=dt, device=dev)
    z32 = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
    L = c.num_layers
    dmod = z32(2 * L, B, G, 3 * D)
    dmodf = z32(B, G, 2 * D)
    grp = ops.groups(S, Nt, per_group)
    gv = ops.groups(Nv, 0, per_group)
    mb, mg = G * 3 * D, 3 * D
    scale = 1.0 / math.sqrt(c.attention_head_dim)

    def f32_to_param_grad(param, g32):
        acc = _acc_grad(grads, param)
        acc.add_(g32.to(BF16).view_as(acc))

    # ---- head: unpatchify^T = patchify ; proj_out ; norm_out ; norm_final ----
    fo = model.proj_out.weight.shape[0]
    d_tok = ops.patchify(dout.to(BF16).contiguous(), None, p, pt).view(B * Nv, fo)
    _wgrad(d_tok, sv.vis2, _acc_grad(grads, model.proj_out.weight), B * Nv, fo, D)
    bsum = z32(fo)
    ops.colsum(d_tok, bsum, B * Nv, fo)
    f32_to_param_grad(model.proj_out.bias, bsum)
    dvis2 = e(B * Nv, D)
    _dgrad(d_tok, model.proj_out.weight, dvis2, B * Nv, fo, D)
    no = model.norm_out
    dvis = e(B * Nv, D)
    dg, db_ = z32(D), z32(D)
    ops.layernorm_modulate_bwd(dvis2, sv.vis, None, dvis, no.norm.weight, no.norm.bias, sv.modf[..., D:], dmodf[..., D:],
                               dmodf[..., :D], dg, db_, G * 2 * D, 2 * D, gv, B, D, c.norm_eps)
    f32_to_param_grad(no.norm.weight, dg), f32_to_param_grad(no.norm.bias, db_)
    dx = torch.zeros(M, D, dtype=BF16, device=dev)          # text rows get no gradient from the head
    dg, db_ = z32(D), z32(D)
    ops.layernorm_modulate_bwd(dvis, sv.x_last, None, dx, model.norm_final.weight, model.norm_final.bias, None, None, None, dg,
                               db_, 0, 0, gv, B, D, c.norm_eps, xmap=ops.rowmap(Nv, S, Nt))
    f32_to_param_grad(model.norm_final.weight, dg), f32_to_param_grad(model.norm_final.bias, db_)

    # ---- blocks, last to first ----
    work = e(M, 3 * D)
    vT = torch.zeros(B, heads, 64, s_pad, dtype=BF16, device=dev)
    qT, kT, doT = (torch.zeros(B, heads, 64, s_pad, dtype=BF16, device=dev) for _ in range(3))
    nl, nd = e(B, heads, s_pad, dt=torch.float32), e(B, heads, s_pad, dt=torch.float32)
    for i in reversed(range(L)):
        blk, ly = model.transformer_blocks[i], sv.layers[i]
        at = blk.attn1
        m1, m2 = sv.mod[2 * i], sv.mod[2 * i + 1]
        dm1, dm2 = dmod[2 * i], dmod[2 * i + 1]
        f0, f2 = blk.ff.net[0].proj, blk.ff.net[2]
        FF = f0.weight.shape[0]
        # FFN branch: x2 = x1 + g2 * y2
        dy2 = e(M, D)
        ops.gated_residual_bwd(dx, ly.y2, m2[..., 2 * D:], dm2[..., 2 * D:], dy2, mb, mg, grp, B, D)
        _wgrad(dy2, ly.h, _acc_grad(grads, f2.weight), M, D, FF)
        bs = z32(D); ops.colsum(dy2, bs, M, D); f32_to_param_grad(f2.bias, bs)
        du = e(M, FF)
        _dgrad(dy2, f2.weight, du, M, D, FF, epilogue=3, R=ly.u)          # GELU adjoint fused
        _wgrad(du, ly.xn2, _acc_grad(grads, f0.weight), M, FF, D)
        bs = z32(FF); ops.colsum(du, bs, M, FF); f32_to_param_grad(f0.bias, bs)
        dxn2 = e(M, D)
        _dgrad(du, f0.weight, dxn2, M, FF, D)
        dx1 = e(M, D)
        dg, db_ = z32(D), z32(D)
        ops.layernorm_modulate_bwd(dxn2, ly.x1, dx, dx1, blk.norm2.norm.weight, blk.norm2.norm.bias, m2[..., D:2 * D],
                                   dm2[..., D:2 * D], dm2[..., :D], dg, db_, mb, mg, grp, B, D, c.norm_eps)
        f32_to_param_grad(blk.norm2.norm.weight, dg), f32_to_param_grad(blk.norm2.norm.bias, db_)
        # attention branch: x1 = x0 + g1 * y1
        dy1 = e(M, D)
        ops.gated_residual_bwd(dx1, ly.y1, m1[..., 2 * D:], dm1[..., 2 * D:], dy1, mb, mg, grp, B, D)
        wo = at.to_out[0]
        _wgrad(dy1, ly.att, _acc_grad(grads, wo.weight), M, D, D)
        bs = z32(D); ops.colsum(dy1, bs, M, D); f32_to_param_grad(wo.bias, bs)
        datt = e(M, D)
        _dgrad(dy1, wo.weight, datt, M, D, D)
        work.copy_(ly.qkv_raw)
        ops.qkv_prep(work, vT, at.norm_q.weight, at.norm_q.bias, at.norm_k.weight, at.norm_k.bias, sv.rope, B, S, heads, Nt,
                     s_pad, at.eps, q_premul=scale * LOG2E)
        ops.head_transpose(work, 0, qT, B, S, heads, s_pad, ld=3 * D)
        ops.head_transpose(work, D, kT, B, S, heads, s_pad, ld=3 * D)
        ops.head_transpose(datt, 0, doT, B, S, heads, s_pad, ld=D)
        dqkv = e(M, 3 * D)
        ops.attention_bwd(work, qT, kT, ly.att, datt, doT, ly.lse, nl, nd, dqkv, B, S, heads, s_pad, scale)
        dgq, dbq, dgk, dbk = z32(64), z32(64), z32(64), z32(64)
        ops.qkv_prep_bwd(ly.qkv_raw, dqkv, at.norm_q.weight, at.norm_k.weight, sv.rope, dgq, dbq, dgk, dbk, B, S, heads, Nt,
                         at.eps)
        f32_to_param_grad(at.norm_q.weight, dgq), f32_to_param_grad(at.norm_q.bias, dbq)
        f32_to_param_grad(at.norm_k.weight, dgk), f32_to_param_grad(at.norm_k.bias, dbk)
        wqkv, _ = at.packed_qkv()
        dwqkv = torch.zeros(3 * D, D, dtype=BF16, device=dev)
        _wgrad(dqkv, ly.xn1, dwqkv, M, 3 * D, D)
        for j, lin in enumerate((at.to_q, at.to_k, at.to_v)):
            _acc_grad(grads, lin.weight).add_(dwqkv[j * D:(j + 1) * D])
        bs = z32(3 * D); ops.colsum(dqkv, bs, M, 3 * D)
        for j, lin in enumerate((at.to_q, at.to_k, at.to_v)):
            if lin.bias is not None:
                f32_to_param_grad(lin.bias, bs[j * D:(j + 1) * D])
        dxn1 = e(M, D)
        _dgrad(dqkv, wqkv, dxn1, M, 3 * D, D)
        dx0 = e(M, D)
        dg, db_ = z32(D), z32(D)
        ops.layernorm_modulate_bwd(dxn1, ly.x0, dx1, dx0, blk.norm1.norm.weight, blk.norm1.norm.bias, m1[..., D:2 * D],
                                   dm1[..., D:2 * D], dm1[..., :D], dg, db_, mb, mg, grp, B, D, c.norm_eps)
        f32_to_param_grad(blk.norm1.norm.weight, dg), f32_to_param_grad(blk.norm1.norm.bias, db_)
        dx = dx0

    # ---- patch embed (weights only; the latents need no gradient) ----
    pe = model.patch_embed
    dxj = dx.view(B, S, D)
    dvis_rows = dxj[:, Nt:].reshape(B * Nv, D).contiguous()
    wp = pe.proj.weight.reshape(D, -1)
    gp = torch.zeros(D, sv.tokens.shape[-1], dtype=BF16, device=dev)
    _wgrad(dvis_rows, sv.tokens.view(B * Nv, -1), gp, B * Nv, D, sv.tokens.shape[-1])
    _acc_grad(grads, pe.proj.weight).add_(gp.view_as(pe.proj.weight))
    bs = z32(D); ops.colsum(dvis_rows, bs, B * Nv, D); f32_to_param_grad(pe.proj.bias, bs)
    if mod_text:
        dtxt = dxj[:, :Nt].reshape(B * Nt, D).contiguous()
        Kt = sv.text2d.shape[1]
        _wgrad(dtxt, sv.text2d, _acc_grad(grads, pe.text_proj.weight), B * Nt, D, Kt)
        bs = z32(D); ops.colsum(dtxt, bs, B * Nt, D); f32_to_param_grad(pe.text_proj.bias, bs)

    # ---- modulation tables -> AdaLN linears -> conditioning (temb, action embedding) ----
    temb32 = sv.temb.float()
    if sv.action_emb is not None:
        pre_v = (sv.temb[:, None] + sv.action_emb).float()                  # bf16 add like the reference, then fp32
    else:
        pre_v = temb32[:, None]
    cond_v = torch.nn.functional.silu(pre_v).to(BF16).reshape(B * Ta, E).contiguous()
    cond_t = torch.nn.functional.silu(temb32).to(BF16).contiguous()
    d_cond_v, d_cond_t = z32(B * Ta, E), z32(B, E)

    def table_bwd(lin, dtab, width, text):
        dv = dtab[:, 1:].reshape(B * Ta, width).contiguous()
        gw = _acc_grad(grads, lin.weight)
        gb32 = z32(lin.weight.shape[0])
        ops.small_linear_bwd(dv, cond_v, lin.weight[:width], gw[:width], gb32[:width], d_cond_v, B * Ta, width, E)
        if text:
            dt = dtab[:, 0].contiguous()
            ops.small_linear_bwd(dt, cond_t, lin.weight[width:], gw[width:], gb32[width:], d_cond_t, B, width, E)
        if lin.bias is not None:
            f32_to_param_grad(lin.bias, gb32)

    for i, blk in enumerate(model.transformer_blocks):
        table_bwd(blk.norm1.linear, dmod[2 * i], 3 * D, mod_text)
        table_bwd(blk.norm2.linear, dmod[2 * i + 1], 3 * D, mod_text)
    table_bwd(model.norm_out.linear, dmodf, 2 * D, False)

    def dsilu(x):
        s = torch.sigmoid(x)
        return s * (1 + x * (1 - s))

    d_pre_v = d_cond_v.view(B, Ta, E) * dsilu(pre_v)
    d_temb = d_pre_v.sum(1) + d_cond_t * dsilu(temb32)
    if sv.action_emb is not None:
        ae = model.action_embed
        d_emb = d_pre_v
        if ae.mask:
            # masked samples took the embedding from mask_embed.weight
            f32_to_param_grad(ae.mask_embed.weight, (d_emb * sv.is_mask[:, None, None]).sum((0, 1))[None])
            d_emb = d_emb * (~sv.is_mask)[:, None, None]
        d_emb2 = d_emb.reshape(B * Ta, E).contiguous()
        w3, w0 = ae.mlp[3], ae.mlp[0]
        d_h = z32(B * Ta, w3.weight.shape[1])
        gb = z32(E)
        ops.small_linear_bwd(d_emb2, sv.ae_h, w3.weight, _acc_grad(grads, w3.weight), gb, d_h, B * Ta, E, w3.weight.shape[1])
        f32_to_param_grad(w3.bias, gb)
        u = sv.ae_u.float()
        a_, b_ = 0.7978845608028654, 0.044715
        th = torch.tanh(a_ * (u + b_ * u ** 3))
        d_u = (d_h * (0.5 * (1 + th) + 0.5 * u * (1 - th * th) * a_ * (1 + 3 * b_ * u * u))).contiguous()
        gb = z32(w0.weight.shape[0])
        ops.small_linear_bwd(d_u, sv.ae_in, None, _acc_grad(grads, w0.weight), gb, None, B * Ta, w0.weight.shape[0],
                             w0.weight.shape[1])
        f32_to_param_grad(w0.bias, gb)
    te = model.time_embedding
    d_h1 = z32(B, E)
    gb = z32(E)
    ops.small_linear_bwd(d_temb.contiguous(), sv.te_h1, te.linear_2.weight, _acc_grad(grads, te.linear_2.weight), gb, d_h1, B, E, E)
    f32_to_param_grad(te.linear_2.bias, gb)
    d_u1 = (d_h1 * dsilu(sv.te_u1.float())).contiguous()
    gb = z32(E)
    ops.small_linear_bwd(d_u1, sv.t_emb, None, _acc_grad(grads, te.linear_1.weight), gb, None, B, E, D)
    f32_to_param_grad(te.linear_1.bias, gb)
    return grads


class DiTFunction(torch.autograd.Function):
    """Whole-transformer autograd node: forward = HIP forward saving activations, backward = HIP backward.  The
    parameters are passed as inputs so autograd (and DDP's hooks) see their gradients."""

    @staticmethod
    def forward(ctx, model, hidden_states, encoder_hidden_states, controls, timestep, ofs, image_rotary_emb, *params):
        with torch.no_grad():
            out, is_mask, recon, sv = forward_train(model, hidden_states, encoder_hidden_states, controls, timestep, ofs,
                                                    image_rotary_emb)
        ctx.model, ctx.sv, ctx.params = model, sv, params
        ctx.mark_non_differentiable(is_mask) if is_mask is not None else None
        return (out, is_mask) if is_mask is not None else (out, torch.zeros(0, dtype=torch.bool, device=out.device))

    @staticmethod
    def backward(ctx, dout, _dmask):
        with torch.no_grad():
            grads = backward(ctx.model, ctx.sv, dout.contiguous())
        ctx.sv = None
        outs = []
        for p_ in ctx.params:
            g = grads.get(id(p_))
            outs.append(g.to(p_.dtype) if (g is not None and p_.requires_grad) else None)
        return (None, None, None, None, None, None, None, *outs)


def forward_with_grad(model, hidden_states, encoder_hidden_states, controls, timestep, ofs=None, image_rotary_emb=None):
    params = tuple(p_ for p_ in model.parameters())
    out, is_mask = DiTFunction.apply(model, hidden_states, encoder_hidden_states, controls, timestep, ofs, image_rotary_emb,
                                     *params)
    return out, (is_mask if is_mask.numel() else None), None
