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
import os
from typing import Dict, List, Optional

import torch

from . import _state, ops

_SCATTER_GRADS = os.environ.get("ORV_SCATTER_GRADS", "1") != "0"      # A/B switch (small gradients: arena -> flat buffer directly)

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


_WGRAD_TN = os.environ.get("ORV_WGRAD_TN", "0") == "1"


def _wgrad(dY2d, X2d, dW, M, N, K, accumulate=True, bias_sum=None):
    """dW[N,K] (+)= dY[M,N]^T . X[M,K]  through two transposes and the NT GEMM (contraction over the M rows).
    ``bias_sum`` (fp32 [N]): += column sums of dY, taken by the transpose that reads dY anyway."""
    if (_WGRAD_TN and N % 8 == 0 and K % 8 == 0 and dY2d.is_contiguous() and X2d.is_contiguous() and dW.is_contiguous()
            and dW.data_ptr() % 16 == 0):
        # opt-in (ORV_WGRAD_TN=1): orv_gemm_tn_bf16 reads dY and X row-major (transposing LDS reads) - no transposed copies; the bias
        # gradient then costs its own pass over dY.  Only where the tile count fills the chip (the [1920, 1920] out-projection gradient is
        # 80 tiles of 256 x 192: slower); a tall gradient ([7680, 1920]) is computed as dW^T = X^T dY on 256-wide tiles and transposed.
        # Measured against the transposes + NT kernel: profiles/r4_gemm_tn.txt.
        def tiles(rows, cols):
            """256-row x (256 | 192)-column tiles of a [rows, cols] gradient; 0 when neither width divides cols"""
            if cols % 256 == 0:
                return -(-rows // 256) * (cols // 256)
            if cols % 192 == 0:
                return -(-rows // 256) * (cols // 192)
            return 0
        direct, flipped = tiles(N, K), (tiles(K, N) if not accumulate else 0)
        if max(direct, flipped) >= 200:
            if bias_sum is not None:
                ops.colsum(dY2d, bias_sum, M, N)
            if flipped >= 200 and (direct < 200 or (N % 256 == 0 and K % 256 != 0)):
                dWT = torch.empty(K, N, dtype=BF16, device=dW.device)
                ops.gemm_tn(X2d, dY2d, dWT, K, N, M)
                ops.transpose(dWT, K, N, out=dW.view(N, K))
            else:
                ops.gemm_tn(dY2d, X2d, dW.view(N, K), N, K, M, accumulate=accumulate)
            return
    dYT = ops.transpose(dY2d, M, N, colsum=bias_sum)           # [N, M_pad]
    XT = ops.transpose(X2d, M, K)             # [K, M_pad]
    if not accumulate and N % 256 == 0 and K % 256 != 0 and K % 64 == 0 and N >= 2 * K:
        # tall gradient whose long side only is a multiple of 256 (FeedForward net.0: [7680, 1920]): compute dW^T = X^T . dY with
        # the long side as the GEMM's N, where the 256 x 256 phased kernel applies, and transpose the 30 MB result
        # (0.47 -> 0.39 + 0.03 ms at M = 12904)
        dWT = torch.empty(K, N, dtype=BF16, device=dW.device)
        ops.gemm(XT, dYT, None, dWT, K, N, dYT.shape[1])
        ops.transpose(dWT, K, N, out=dW.view(N, K))
    elif K % 64 == 0 and not accumulate:
        ops.gemm(dYT, XT, None, dW, N, K, dYT.shape[1])
    elif K % 64 == 0:
        ops.gemm(dYT, XT, None, dW, N, K, dYT.shape[1], epilogue=2, R=dW, ldr=K)
    else:                                     # odd widths (tiny test configs only): pad the output columns
        kp = (K + 63) // 64 * 64
        XTp = torch.zeros(kp, XT.shape[1], dtype=BF16, device=XT.device)
        XTp[:K] = XT
        tmp = torch.empty(N, kp, dtype=BF16, device=XT.device)
        ops.gemm(dYT, XTp, None, tmp, N, kp, dYT.shape[1])
        if accumulate:
            dW.add_(tmp[:, :K])
        else:
            dW.copy_(tmp[:, :K])


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


class _AttnBufs:
    """Per-(batch, sequence) scratch of the attention forward/backward (not saved activations)."""

    def __init__(self, B_, S_, heads, dev):
        self.s_pad = (S_ + 63) // 64 * 64
        # the ping-pong backward kernels read K^T / Q'^T / dO^T with transposing LDS reads: no per-head transposed copies
        self.qT = self.kT = self.doT = None
        if os.environ.get("ORV_ATTN_BWD_PP", "1") == "0":
            z = lambda: torch.zeros(B_, heads, 64, self.s_pad, dtype=BF16, device=dev)
            self.qT, self.kT, self.doT = z(), z(), z()
        self.nl = torch.empty(B_, heads, self.s_pad, dtype=torch.float32, device=dev)
        self.nd = torch.empty(B_, heads, self.s_pad, dtype=torch.float32, device=dev)


def _attn_forward(at, xn, ly, bufs, B_, S_, n_text, heads, rope, scale):
    """QKV GEMM -> qk LayerNorm / RoPE / V^T -> flash attention; keeps qkv_raw, att, lse on ``ly``."""
    D = heads * 64
    M_ = B_ * S_
    dev = xn.device
    from .cogvideox_control import CogVideoXTransformer3DModelTraj as _M
    ly.qkv_raw = torch.empty(M_, 3 * D, dtype=BF16, device=dev)       # raw projection: input of the qk-LayerNorm adjoint
    ly.qkvn = torch.empty(M_, 3 * D, dtype=BF16, device=dev)          # q' | k' | v as the attention kernels read them (kept:
    _M._qkv_projection(at, xn, ly.qkvn, rope, B_, S_, heads, n_text, bufs.s_pad, scale, raw=ly.qkv_raw)   # 148 MB/layer)
    ly.att = torch.empty(M_, D, dtype=BF16, device=dev)
    ly.lse = torch.empty(B_, heads, S_, dtype=torch.float32, device=dev)
    ops.attention_fwd(ly.qkvn, None, ly.att, B_, S_, heads, bufs.s_pad, 1.0 / LOG2E, lse=ly.lse, score_bound_dev=at.score_bound_dev(scale))


def _attn_backward(at, ly, xn, datt, bufs, B_, S_, n_text, heads, rope, scale, grads, f32_to_param_grad, z32):
    """Adjoint of ``_attn_forward``: datt [M, D] -> gradient w.r.t. xn [M, D]; parameter gradients into ``grads``."""
    D = heads * 64
    M_ = B_ * S_
    dev = xn.device
    dqkv = torch.empty(M_, 3 * D, dtype=BF16, device=dev)
    if bufs.qT is not None:              # ORV_ATTN_BWD_PP=0: the two-pass kernels of rounds 1-2 on per-head transposed copies (A/B)
        ops.head_transpose(ly.qkvn, 0, bufs.qT, B_, S_, heads, bufs.s_pad, ld=3 * D)
        ops.head_transpose(ly.qkvn, D, bufs.kT, B_, S_, heads, bufs.s_pad, ld=3 * D)
        ops.head_transpose(datt, 0, bufs.doT, B_, S_, heads, bufs.s_pad, ld=D)
    ops.attention_bwd(ly.qkvn, bufs.qT, bufs.kT, ly.att, datt, bufs.doT, ly.lse, bufs.nl, bufs.nd, dqkv, B_, S_, heads,
                      bufs.s_pad, scale)
    dgq, dbq, dgk, dbk = z32(64), z32(64), z32(64), z32(64)
    ops.qkv_prep_bwd(ly.qkv_raw, dqkv, at.norm_q.weight, at.norm_k.weight, rope, dgq, dbq, dgk, dbk, B_, S_, heads, n_text,
                     at.eps)
    f32_to_param_grad(at.norm_q.weight, dgq), f32_to_param_grad(at.norm_q.bias, dbq)
    f32_to_param_grad(at.norm_k.weight, dgk), f32_to_param_grad(at.norm_k.bias, dbk)
    wqkv, _ = at.packed_qkv()
    lins = (at.to_q, at.to_k, at.to_v)
    want_bias = lins[0].bias is not None and any(l.bias.requires_grad for l in lins)
    bs = z32(3 * D) if want_bias else None
    bias_done = False
    if any(l.weight.requires_grad for l in lins):
        dwqkv = torch.empty(3 * D, D, dtype=BF16, device=dev)
        _wgrad(dqkv, xn, dwqkv, M_, 3 * D, D, accumulate=False, bias_sum=bs)
        bias_done = True
        for j, lin in enumerate(lins):
            if not lin.weight.requires_grad:
                continue
            if id(lin.weight) in grads:
                grads[id(lin.weight)].add_(dwqkv[j * D:(j + 1) * D])
            else:                               # first (normally only) contribution: one copy into the optimizer's segment
                g = _state.grad_view(lin.weight)    # (was: zero-fill + add + the optimizer's copy: 7 passes over 7.4 MB, now 2)
                if g is None:
                    g = torch.empty_like(lin.weight, dtype=BF16)
                g.copy_(dwqkv[j * D:(j + 1) * D])
                grads[id(lin.weight)] = g
    if want_bias:
        if not bias_done:
            ops.colsum(dqkv, bs, M_, 3 * D)
        for j, lin in enumerate(lins):
            f32_to_param_grad(lin.bias, bs[j * D:(j + 1) * D])
    dxn = torch.empty(M_, D, dtype=BF16, device=dev)
    _dgrad(dqkv, wqkv, dxn, M_, 3 * D, D)
    return dxn


def _mv_index(b, v, f, Nt, P, S, dev):
    """Row index of the '(b v) (f s) -> (b f) (v s)' regrouping (text '(b v) n -> (b f) (v n)'), as in
    ``CogVideoXTransformer3DModelTraj._mv_state``."""
    bi = torch.arange(b, device=dev).view(b, 1, 1, 1)
    fi = torch.arange(f, device=dev).view(1, f, 1, 1)
    vi = torch.arange(v, device=dev).view(1, 1, v, 1)
    txt = ((bi * v + vi) * S + torch.arange(Nt, device=dev).view(1, 1, 1, Nt)).expand(b, f, v, Nt)
    vid = (bi * v + vi) * S + Nt + fi * P + torch.arange(P, device=dev).view(1, 1, 1, P)
    return torch.cat([txt.reshape(b, f, v * Nt), vid.reshape(b, f, v * P)], dim=2).to(torch.int32).contiguous().view(-1)


def forward_train(model, hidden_states, encoder_hidden_states, controls, timestep, ofs=None, image_rotary_emb=None,
                  num_views=1, image_rotary_emb_view=None):
    """Same arithmetic as ``CogVideoXTransformer3DModelTraj.forward`` (inference kernels), keeping what backward needs."""
    c = model.config
    dev = hidden_states.device
    nv_ = num_views
    if nv_ > 1:                                                                      # :756-758
        bb, vf = hidden_states.shape[:2]
        hidden_states = hidden_states.reshape(bb * nv_, vf // nv_, *hidden_states.shape[2:])
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(nv_, dim=0)
    B, T, C, Hh, Ww = hidden_states.shape
    b0 = B // nv_
    p, pt = c.patch_size, c.patch_size_t
    D, heads, E = model.inner_dim, c.num_attention_heads, c.time_embed_dim
    mod_text = bool(c.modulate_encoder_hidden_states)
    Nt = encoder_hidden_states.shape[1] if mod_text else 0
    Tq = T // (pt or 1)
    P = (Hh // p) * (Ww // p)
    Nv = Tq * P
    S = Nt + Nv
    M = B * S
    sv = _Saved()
    sv.dims = dict(B=B, T=T, Hh=Hh, Ww=Ww, D=D, heads=heads, E=E, Nt=Nt, Nv=Nv, S=S, M=M, P=P, mod_text=mod_text, nv=nv_,
                   b0=b0)
    e = lambda *shape, dt=BF16: torch.empty(*shape, dtype=dt, device=dev)

    # conditioning branch (tiny): matmuls in HIP, activations' adjoints later via torch on [B, T, E] tensors
    tvec = torch.as_tensor(timestep, device=dev).reshape(-1).to(torch.float32)
    if tvec.numel() == 1 and b0 > 1:
        tvec = tvec.expand(b0)
    te = model.time_embedding
    sv.t_emb = ops.timestep_embedding(tvec.contiguous(), D, c.flip_sin_to_cos, c.freq_shift)
    sv.te_u1 = ops.skinny_linear(sv.t_emb, te.linear_1.weight, te.linear_1.bias)                 # pre-SiLU
    sv.te_h1 = torch.nn.functional.silu(sv.te_u1.float()).to(BF16)
    temb = ops.skinny_linear(sv.te_h1, te.linear_2.weight, te.linear_2.bias)
    sv.has_ofs = model.ofs_embedding is not None
    if sv.has_ofs:                                                                   # :771-775
        oe = model.ofs_embedding
        ovec = torch.as_tensor(ofs, device=dev).reshape(-1).to(torch.float32)
        sv.o_emb = ops.timestep_embedding(ovec.contiguous(), c.ofs_embed_dim, c.flip_sin_to_cos, c.freq_shift)
        sv.oe_u1 = ops.skinny_linear(sv.o_emb, oe.linear_1.weight, oe.linear_1.bias)
        sv.oe_h1 = torch.nn.functional.silu(sv.oe_u1.float()).to(BF16)
        temb = temb + ops.skinny_linear(sv.oe_h1, oe.linear_2.weight, oe.linear_2.bias)
    if nv_ > 1:
        temb = temb.repeat_interleave(nv_, dim=0).contiguous()
    sv.temb = temb

    pe = model.patch_embed
    sv.tokens = ops.patchify(hidden_states.to(BF16), None, p, pt)
    x = e(M, D)
    a2, w2 = _pad_k(sv.tokens.view(B * Nv, -1), pe.proj.weight.reshape(D, -1))
    pos = pe.video_pos_table(T, Hh, Ww, dev)
    cpos = pos
    pos_mod = Nv
    if nv_ > 1:
        pos, pos_mod = model._view_pos_table(pos, nv_, T, P, dev), nv_ * Nv
    vmap = ops.rowmap(Nv, S, Nt)
    ops.gemm(a2, w2, pe.proj.bias, x, B * Nv, D, a2.shape[1], epilogue=2 if pos is not None else 0, R=pos, r_mod=pos_mod, ldr=D,
             cmap=vmap)
    if mod_text:
        sv.text2d = encoder_hidden_states.to(BF16).reshape(B * Nt, -1).contiguous()
        a2, w2 = _pad_k(sv.text2d, pe.text_proj.weight)
        ops.gemm(a2, w2, pe.text_proj.bias, x, B * Nt, D, a2.shape[1], cmap=ops.rowmap(Nt, S, 0))

    action_emb = is_mask = actions_recon = None
    actions = controls.get('actions', None)
    sv.has_actions = actions is not None
    sv.has_recon = False
    if actions is not None:
        actions = actions.to(device=dev)
        res = (actions.size(1) + 1) % 4
        padf = 4 - res if res > 0 else 0
        if padf:
            actions = torch.cat([actions.new_zeros((actions.shape[0], padf, actions.shape[2])), actions], dim=1)
        ae = model.action_embed
        xa = torch.cat([torch.zeros_like(actions[:, :1]), actions], dim=1)
        xa = xa.reshape(b0, (actions.shape[1] + 1) // ae.compress_ratio, -1)
        if ae.patch_size_t > 1:
            xa = xa.reshape(b0, xa.shape[1] // ae.patch_size_t, -1)
        Ta = xa.shape[1]
        sv.ae_in = xa.reshape(b0 * Ta, -1).to(BF16).contiguous()
        sv.ae_u = ops.skinny_linear(sv.ae_in, ae.mlp[0].weight, ae.mlp[0].bias)
        sv.ae_h = torch.nn.functional.gelu(sv.ae_u.float(), approximate="tanh").to(BF16)
        emb = ops.skinny_linear(sv.ae_h, ae.mlp[3].weight, ae.mlp[3].bias).view(b0, Ta, E)
        is_mask = ae.forced_mask.to(dev, torch.bool) if ae.forced_mask is not None else torch.rand(b0, device=dev) < 0.1
        if ae.mask:
            emb = torch.where(is_mask[:, None, None], ae.mask_embed.weight[None].to(emb.dtype), emb)
        sv.is_mask = is_mask
        if nv_ > 1:                                                                  # :815-816
            emb = emb.repeat_interleave(nv_, dim=0)
        action_emb = emb.contiguous()
        if model.training and c.recon_action and model.action_recon is not None:    # :822-825, components.py:92-104
            ar = model.action_recon
            sv.has_recon, sv.ar_pad = True, padf
            sv.ar_in = action_emb.reshape(B * Ta, E)
            sv.ar_u = ops.skinny_linear(sv.ar_in, ar.mlp[0].weight, ar.mlp[0].bias)
            sv.ar_h = torch.nn.functional.gelu(sv.ar_u.float(), approximate="tanh").to(BF16)
            y = ops.skinny_linear(sv.ar_h, ar.mlp[2].weight, ar.mlp[2].bias).view(B, Ta, -1)
            if ar.compress_ratio > 1:
                y = y.reshape(B, int(Ta * ar.compress_ratio), y.shape[-1] // ar.compress_ratio)
            actions_recon = y[:, 1 + padf:].contiguous()
    sv.action_emb = action_emb

    # occupancy-derived visual guidance (:828-858)
    sv.ctrl_tokens = []
    if c.visual_guidance:
        ctoks = []
        for key in ('depths', 'labels'):
            cm = controls.get(key, None)
            if cm is None:
                continue
            if nv_ > 1:
                cm = cm.reshape(cm.shape[0] * nv_, cm.shape[1] // nv_, *cm.shape[2:])
            tk = ops.patchify(cm.to(device=dev, dtype=BF16), None, p, pt)
            a2c, w2c = _pad_k(tk.view(B * Nv, -1), pe.proj.weight.reshape(D, -1))
            ctok = e(B * Nv, D)
            ops.gemm(a2c, w2c, pe.proj.bias, ctok, B * Nv, D, a2c.shape[1], epilogue=2 if cpos is not None else 0, R=cpos,
                     r_mod=Nv, ldr=D)
            ctoks.append(ctok)
            sv.ctrl_tokens.append(tk.view(B * Nv, -1))
        if ctoks:
            assert len(ctoks) == model.num_control_keys, \
                f'Mismatched number of controls: {len(ctoks)=} but {model.num_control_keys=}.'
            K2 = model.num_control_keys * D
            sv.comb = e(B * Nv, K2)
            for jx, ctok in enumerate(ctoks):
                ops.add_rows(x, vmap, ctok, sv.comb, jx * D, B * Nv, D, ldo=K2)
            icl = model.initial_combine_linear
            ops.gemm(sv.comb, icl.weight, icl.bias, x, B * Nv, D, K2, epilogue=2, R=x, ldr=D, cmap=vmap)

    Ta = action_emb.shape[1] if action_emb is not None else 1
    if Nv % Ta:
        raise ValueError(f"{Nv} video tokens cannot be split over {Ta} action frames")
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
    sv.rope_view = None
    if image_rotary_emb_view is not None:
        sv.rope_view = tuple(r.to(device=dev, dtype=torch.float32).contiguous() for r in image_rotary_emb_view)

    mb, mg = G * 3 * D, 3 * D
    scale = 1.0 / math.sqrt(c.attention_head_dim)
    from .cogvideox_control import prime_score_bounds
    prime_score_bounds([b.attn1 for b in model.transformer_blocks] + ([b.attn1 for b in model.mv_blocks] if c.multiview else []), scale,
                       on_device=True)      # no device -> host read in the training step
    bufs = _AttnBufs(B, S, heads, dev)
    sv.mv = None
    if c.multiview:                                                                  # :273-348
        if Nv % T:
            raise ValueError(f"{Nv} video tokens cannot be split over {T} frames for the multiview blocks")
        Pm = Nv // T
        mv = _Saved()
        mv.idx = _mv_index(b0, nv_, T, Nt, Pm, S, dev)
        mv.Sm, mv.Bm = nv_ * (Nt + Pm), b0 * T
        mv.R = mv.Bm * mv.Sm
        mv.n_text = nv_ * Nt
        wmv, bmv = model._mv_pointer_tables(dev)
        mv.mod = ops.modulation_tables(temb, None, wmv, bmv, L, B, 1, E, 3 * D, mod_text)        # [L, B, 2, 3D]
        mv.grp0 = ops.groups(S, Nt, 0)
        mv.bufs = _AttnBufs(mv.Bm, mv.Sm, heads, dev)
        # gate of (b f)-batch row groups {text, view 0, view 1, ...}: text rows of the attention output are dropped (zero gate)
        mv.grp = ops.groups(mv.Sm, mv.n_text, Pm)
        sv.mv = mv
    # One transformer block (+ its multiview block) as a function of its input: the forward proper, and - under gradient
    # checkpointing - the recompute the backward asks for (the reference wraps every block in torch.utils.checkpoint,
    # cogvideox_control.py:867-899; enabled by config/traj_image_2b_multiview.yaml:33 and BASELINE configs[4]).  Everything else it
    # needs (modulation tables, row maps, RoPE tables, scratch) lives in `sv` / this closure for the lifetime of the step.
    mv = sv.mv
    rope_view = sv.rope_view
    def run_block(i, x):
        blk = model.transformer_blocks[i]
        mly = None
        if mv is not None:
            mblk, mly, m = model.mv_blocks[i], _Saved(), mv.mod[i]
            mly.x_in = x
            xn = e(M, D)
            ops.layernorm_modulate(x, xn, mblk.norm1.norm.weight, mblk.norm1.norm.bias, m[..., D:2 * D], m[..., :D],
                                   2 * 3 * D, 3 * D, mv.grp0, B, D, c.norm_eps)
            mly.xm = e(mv.R, D)
            ops.gather_rows(xn, mv.idx, mly.xm, mv.R, D)
            _attn_forward(mblk.attn1, mly.xm, mly, mv.bufs, mv.Bm, mv.Sm, mv.n_text, heads, rope_view, scale)
            mly.ao, mly.y = e(mv.R, D), e(mv.R, D)
            wo = mblk.attn1.to_out[0]
            ops.gemm(mly.att, wo.weight, wo.bias, mly.ao, mv.R, D, D)
            ops.gemm(mly.ao, mblk.proj_out.weight, mblk.proj_out.bias, mly.y, mv.R, D, D)
            x = x.clone()
            ops.scatter_gated_rows(mly.y, mv.idx, m[:, 1, 2 * D:], 2 * 3 * D, x, mv.R, D, S, Nt)
        m1, m2 = mod[2 * i], mod[2 * i + 1]
        at = blk.attn1
        ly = _Saved()
        ly.x0 = x
        ly.xn1 = e(M, D)
        ops.layernorm_modulate(x, ly.xn1, blk.norm1.norm.weight, blk.norm1.norm.bias, m1[..., D:2 * D], m1[..., :D], mb, mg,
                               grp, B, D, c.norm_eps)
        _attn_forward(at, ly.xn1, ly, bufs, B, S, Nt, heads, rope, scale)
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
        return x2, ly, mly

    # gradient checkpointing: keep only every block's INPUT (49.5 MB per block at B = 4 instead of ~0.94 GB of activations);
    # `backward` rebuilds the block's activations with `sv.recompute(i)` right before it needs them - same kernels, same inputs,
    # deterministic: gradients are bit-identical to the run that keeps everything
    sv.checkpointed = bool(getattr(model, "gradient_checkpointing", False)) and model.training
    checkpointed = sv.checkpointed
    layers = sv.layers = []
    if mv is not None:
        mv.layers = []
    for i in range(len(model.transformer_blocks)):
        x_in = x
        x, ly, mly = run_block(i, x)
        if checkpointed:
            stub = _Saved()
            stub.block_input = x_in
            ly, mly = stub, (None if mly is None else stub)
        layers.append(ly)
        if mv is not None:
            mv.layers.append(mly)

    def recompute(i):
        """Rebuild block i's saved activations from its input (checkpointed step); frees them again when the caller drops them."""
        _, ly, mly = run_block(i, layers[i].block_input)
        layers[i] = ly
        if mv is not None:
            mv.layers[i] = mly
    # (the closures reference `layers`, `mv` and locals, never `sv` itself: no reference cycle through sv.recompute, so a step's
    #  activations are released by reference counting the moment the autograd node dies - a cycle here kept the previous step's
    #  ~100 GB alive until the next garbage collection: 5B went from 329 to 889 ms per step and 105 to 200 GiB)
    sv.recompute = recompute
    sv.x_last = x
    gv = ops.groups(Nv, 0, per_group)
    sv.vis, sv.vis2 = e(B * Nv, D), e(B * Nv, D)
    ops.layernorm_modulate(x, sv.vis, model.norm_final.weight, model.norm_final.bias, None, None, 0, 0, gv, B, D, c.norm_eps,
                           xmap=ops.rowmap(Nv, S, Nt))
    no = model.norm_out
    ops.layernorm_modulate(sv.vis, sv.vis2, no.norm.weight, no.norm.bias, modf[..., D:], modf[..., :D], G * 2 * D, 2 * D, gv, B,
                           D, c.norm_eps)
    fo = model.proj_out.weight.shape[0]
    sv.fo_pad = (fo + 63) // 64 * 64
    wo_, bo_ = model.proj_out.weight, model.proj_out.bias
    if sv.fo_pad != fo:                       # CogVideoX1.5 (p_t = 2): 128 -> fits; tiny test configs: zero-padded columns
        wo_ = torch.nn.functional.pad(wo_, (0, 0, 0, sv.fo_pad - fo)).contiguous()
        bo_ = torch.nn.functional.pad(bo_, (0, sv.fo_pad - fo)).contiguous() if bo_ is not None else None
    out_tok = e(B * Nv, sv.fo_pad)
    ops.gemm(sv.vis2, wo_, bo_, out_tok, B * Nv, sv.fo_pad, D)
    if sv.fo_pad != fo:
        out_tok = out_tok[:, :fo].contiguous()
    c_out = fo // (p * p * (pt or 1))
    output = ops.unpatchify(out_tok, B, T, c_out, Hh, Ww, p, pt)
    if nv_ > 1:                                                                      # :941
        output = output.reshape(b0, nv_ * T, *output.shape[2:])
    return output, is_mask, actions_recon, sv


def _gelu_tanh_grad(u):
    a_, b_ = 0.7978845608028654, 0.044715
    th = torch.tanh(a_ * (u + b_ * u ** 3))
    return 0.5 * (1 + th) + 0.5 * u * (1 - th * th) * a_ * (1 + 3 * b_ * u * u)


def backward(model, sv, dout, drecon=None, grad_hook=None) -> Dict[int, torch.Tensor]:
    """Gradients of every trainable parameter (bf16, keyed by id(param)) given dL/d(sample) (and dL/d(actions_recon)).
    ``grad_hook(params, grads)`` is called as soon as the gradients of a group of parameters are final (data-parallel
    exchange overlapped with the rest of the backward, ``FusedAdamW.begin_overlapped_allreduce``)."""
    c = model.config
    d = sv.dims
    B, T, Hh, Ww, D, heads, E = d["B"], d["T"], d["Hh"], d["Ww"], d["D"], d["heads"], d["E"]
    Nt, Nv, S, M, G, Ta, per_group = d["Nt"], d["Nv"], d["S"], d["M"], d["G"], d["Ta"], d["per_group"]
    mod_text, nv_, b0 = d["mod_text"], d["nv"], d["b0"]
    dev = dout.device
    p, pt = c.patch_size, c.patch_size_t
    grads: Dict[int, torch.Tensor] = {}
    e = lambda *shape, dt=BF16: torch.empty(*shape, dtype=dt, device=dev)
    L = c.num_layers
    # fp32 accumulators (norm / bias gradients, tables, ...) come out of ONE zero-filled arena: one memset instead of ~400
    # tiny fills per step, and the gradients that live in it are converted to bf16 with ONE cast at the end
    arena = torch.zeros(2 * L * B * G * 3 * D + B * G * 2 * D + (L + 4) * 64 * 1024, dtype=torch.float32, device=dev)
    arena_ptr = [0]
    pending = []          # (param, arena offset, numel)

    def z32(*shape):
        n = 1
        for k in shape:
            n *= int(k)
        a = (arena_ptr[0] + 63) // 64 * 64
        if a + n > arena.numel():
            return torch.zeros(*shape, dtype=torch.float32, device=dev)
        arena_ptr[0] = a + n
        return arena[a:a + n].view(*shape)

    dmod = z32(2 * L, B, G, 3 * D)
    dmodf = z32(B, G, 2 * D)
    grp = ops.groups(S, Nt, per_group)
    gv = ops.groups(Nv, 0, per_group)
    mb, mg = G * 3 * D, 3 * D
    scale = 1.0 / math.sqrt(c.attention_head_dim)

    def f32_to_param_grad(param, g32):
        if param is None or not param.requires_grad:
            return
        if g32.is_contiguous() and g32.untyped_storage().data_ptr() == arena.untyped_storage().data_ptr():
            pending.append((param, g32.storage_offset(), g32.numel()))      # converted with the rest of the arena at the end
            return
        g = g32.to(BF16).view(param.shape)
        if id(param) in grads:
            grads[id(param)].add_(g)
        else:                                   # first (normally only) contribution: no zero-fill + add
            grads[id(param)] = g

    def wgrad_p(param, dY2d, X2d, M_, N_, K_, bias=None):
        """Weight gradient of one nn.Linear and, with ``bias``, its bias gradient: the column sums of dY come out of the transpose
        that makes dY^T (one read of dY; a frozen weight leaves the bias to orv_colsum)."""
        want_bias = bias is not None and bias.requires_grad
        bs = z32(N_) if want_bias else None
        if not param.requires_grad:             # frozen weights (e.g. everything but mv_blocks, :641-656) cost no GEMM
            if want_bias:
                ops.colsum(dY2d, bs, M_, N_)
                f32_to_param_grad(bias, bs)
            return
        if id(param) in grads:
            _wgrad(dY2d, X2d, grads[id(param)], M_, N_, K_, bias_sum=bs)
        else:                                   # first (normally only) contribution: plain store, no zero-fill + re-read
            g = _state.grad_view(param)         # straight into the fused optimizer's flat gradient buffer when there is one
            if g is None:
                g = torch.empty_like(param, dtype=BF16)
            grads[id(param)] = g
            _wgrad(dY2d, X2d, g, M_, N_, K_, accumulate=False, bias_sum=bs)
        if want_bias:
            f32_to_param_grad(bias, bs)

    # ---- head: unpatchify^T = patchify ; proj_out ; norm_out ; norm_final ----
    if nv_ > 1:
        dout = dout.reshape(B, T, *dout.shape[2:])
    fo = model.proj_out.weight.shape[0]
    d_tok = ops.patchify(dout.to(BF16).contiguous(), None, p, pt).view(B * Nv, fo)
    wo_ = model.proj_out.weight
    if sv.fo_pad != fo:
        d_tok = torch.nn.functional.pad(d_tok, (0, sv.fo_pad - fo)).contiguous()
        wo_ = torch.nn.functional.pad(wo_.detach(), (0, 0, 0, sv.fo_pad - fo)).contiguous()
    if model.proj_out.weight.requires_grad:
        gwo = torch.zeros(sv.fo_pad, D, dtype=BF16, device=dev)
        _wgrad(d_tok, sv.vis2, gwo, B * Nv, sv.fo_pad, D)
        _acc_grad(grads, model.proj_out.weight).add_(gwo[:fo])
    if model.proj_out.bias is not None and model.proj_out.bias.requires_grad:
        bsum = z32(sv.fo_pad)
        ops.colsum(d_tok, bsum, B * Nv, sv.fo_pad)
        f32_to_param_grad(model.proj_out.bias, bsum[:fo])
    dvis2 = e(B * Nv, D)
    _dgrad(d_tok, wo_, dvis2, B * Nv, sv.fo_pad, D)
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
    bufs = _AttnBufs(B, S, heads, dev)
    mv = sv.mv
    if mv is not None:
        dmv = z32(L, B, 2, 3 * D)
        ones_gate = torch.ones(B, D, dtype=torch.float32, device=dev)
    for i in reversed(range(L)):
        if sv.checkpointed:
            sv.recompute(i)              # gradient checkpointing: this block's activations are rebuilt from its input now
        blk, ly = model.transformer_blocks[i], sv.layers[i]
        at = blk.attn1
        m1, m2 = sv.mod[2 * i], sv.mod[2 * i + 1]
        dm1, dm2 = dmod[2 * i], dmod[2 * i + 1]
        f0, f2 = blk.ff.net[0].proj, blk.ff.net[2]
        FF = f0.weight.shape[0]
        # FFN branch: x2 = x1 + g2 * y2
        dy2 = e(M, D)
        ops.gated_residual_bwd(dx, ly.y2, m2[..., 2 * D:], dm2[..., 2 * D:], dy2, mb, mg, grp, B, D)
        wgrad_p(f2.weight, dy2, ly.h, M, D, FF, bias=f2.bias)
        du = e(M, FF)
        _dgrad(dy2, f2.weight, du, M, D, FF, epilogue=3, R=ly.u)          # GELU adjoint fused
        wgrad_p(f0.weight, du, ly.xn2, M, FF, D, bias=f0.bias)
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
        wgrad_p(wo.weight, dy1, ly.att, M, D, D, bias=wo.bias)
        datt = e(M, D)
        _dgrad(dy1, wo.weight, datt, M, D, D)
        dxn1 = _attn_backward(at, ly, ly.xn1, datt, bufs, B, S, Nt, heads, sv.rope, scale, grads, f32_to_param_grad, z32)
        dx0 = e(M, D)
        dg, db_ = z32(D), z32(D)
        ops.layernorm_modulate_bwd(dxn1, ly.x0, dx1, dx0, blk.norm1.norm.weight, blk.norm1.norm.bias, m1[..., D:2 * D],
                                   dm1[..., D:2 * D], dm1[..., :D], dg, db_, mb, mg, grp, B, D, c.norm_eps)
        f32_to_param_grad(blk.norm1.norm.weight, dg), f32_to_param_grad(blk.norm1.norm.bias, db_)
        dx = dx0
        if grad_hook is not None:       # everything of this block but its two AdaLN linears (their tables close at the end)
            grad_hook(list(at.parameters()) + list(blk.ff.parameters()) + list(blk.norm1.norm.parameters())
                      + list(blk.norm2.norm.parameters()), grads)
        if mv is not None:
            # MVBlock adjoint: x_out[idx[r]] = x_in[idx[r]] + gate[view] * y[r] on the video rows (text output dropped)
            mblk, mly, m = model.mv_blocks[i], mv.layers[i], mv.mod[i]
            Pm = Nv // T
            dxm = e(mv.R, D)
            ops.gather_rows(dx, mv.idx, dxm, mv.R, D)
            # gate / its gradient laid out per (b f)-batch: group 0 = text (zero), 1 + v = view v
            gate_mv = z32(b0, T, 1 + nv_, D)
            gate_mv[:, :, 1:] = m[:, 1, 2 * D:].reshape(b0, 1, nv_, D)
            dgate_mv = z32(b0 * T, 1 + nv_, D)
            dy = e(mv.R, D)
            ops.gated_residual_bwd(dxm, mly.y, gate_mv.view(b0 * T, 1 + nv_, D), dgate_mv, dy, (1 + nv_) * D, D, mv.grp, mv.Bm,
                                   D)
            dmv[i][:, 1, 2 * D:] += dgate_mv.view(b0, T, 1 + nv_, D)[:, :, 1:].sum(1).reshape(B, D)
            wgrad_p(mblk.proj_out.weight, dy, mly.ao, mv.R, D, D, bias=mblk.proj_out.bias)
            dao = e(mv.R, D)
            _dgrad(dy, mblk.proj_out.weight, dao, mv.R, D, D)
            wo = mblk.attn1.to_out[0]
            wgrad_p(wo.weight, dao, mly.att, mv.R, D, D, bias=wo.bias)
            datt = e(mv.R, D)
            _dgrad(dao, wo.weight, datt, mv.R, D, D)
            dxm_n = _attn_backward(mblk.attn1, mly, mly.xm, datt, mv.bufs, mv.Bm, mv.Sm, mv.n_text, heads, sv.rope_view, scale,
                                   grads, f32_to_param_grad, z32)
            # gather^T: video rows are a bijection (scatter), text rows were replicated over the f frames (sum)
            dxn = torch.zeros(M, D, dtype=BF16, device=dev)
            ops.scatter_gated_rows(dxm_n, mv.idx, ones_gate, D, dxn, mv.R, D, S, Nt)
            if Nt:
                dtxt = dxm_n.view(b0, T, mv.Sm, D)[:, :, :mv.n_text].float().sum(1)            # [b, v*Nt, D]
                dxn.view(B, S, D)[:, :Nt] = dtxt.view(B, Nt, D).to(BF16)
            dx_in = e(M, D)
            dg, db_ = z32(D), z32(D)
            ops.layernorm_modulate_bwd(dxn, mly.x_in, dx, dx_in, mblk.norm1.norm.weight, mblk.norm1.norm.bias, m[..., D:2 * D],
                                       dmv[i][..., D:2 * D], dmv[i][..., :D], dg, db_, 2 * 3 * D, 3 * D, mv.grp0, B, D,
                                       c.norm_eps)
            f32_to_param_grad(mblk.norm1.norm.weight, dg), f32_to_param_grad(mblk.norm1.norm.bias, db_)
            dx = dx_in
        if sv.checkpointed:             # the recomputed activations of this block are dead: give the memory back
            del ly
            sv.layers[i] = None
            if mv is not None:
                mly = None
                mv.layers[i] = None

    # ---- visual-guidance fuse (:846-858): x_vis += icl([x_vis + c0 | x_vis + c1]) ----
    pe = model.patch_embed
    dxj = dx.view(B, S, D)
    dvis_rows = dxj[:, Nt:].reshape(B * Nv, D).contiguous()
    wp = pe.proj.weight.reshape(D, -1)
    Kp = sv.tokens.shape[-1]
    gp = None
    if pe.proj.weight.requires_grad:
        gp = torch.zeros(D, Kp, dtype=BF16, device=dev)
    dpb = z32(D)
    if sv.ctrl_tokens:
        icl = model.initial_combine_linear
        K2 = icl.weight.shape[1]
        wgrad_p(icl.weight, dvis_rows, sv.comb, B * Nv, D, K2, bias=icl.bias)
        dcomb = e(B * Nv, K2)
        _dgrad(dvis_rows, icl.weight, dcomb, B * Nv, D, K2)
        acc32 = dvis_rows.float()
        for jx, tk in enumerate(sv.ctrl_tokens):
            dct = dcomb[:, jx * D:(jx + 1) * D].contiguous()
            acc32 += dct.float()
            if gp is not None:
                _wgrad(dct, tk, gp, B * Nv, D, Kp)
            ops.colsum(dct, dpb, B * Nv, D)
        dvis_rows = acc32.to(BF16)

    # ---- patch embed (weights only; the latents need no gradient) ----
    if gp is not None:
        _wgrad(dvis_rows, sv.tokens.view(B * Nv, -1), gp, B * Nv, D, Kp)
        _acc_grad(grads, pe.proj.weight).add_(gp.view_as(pe.proj.weight))
    ops.colsum(dvis_rows, dpb, B * Nv, D)
    f32_to_param_grad(pe.proj.bias, dpb)
    if mod_text:
        dtxt = dxj[:, :Nt].reshape(B * Nt, D).contiguous()
        Kt = sv.text2d.shape[1]
        wgrad_p(pe.text_proj.weight, dtxt, sv.text2d, B * Nt, D, Kt, bias=pe.text_proj.bias)

    # ---- modulation tables -> AdaLN linears -> conditioning (temb, action embedding) ----
    temb32 = sv.temb.float()
    if sv.action_emb is not None:
        pre_v = (sv.temb[:, None] + sv.action_emb).float()                  # bf16 add like the reference, then fp32
    else:
        pre_v = temb32[:, None]
    cond_v = torch.nn.functional.silu(pre_v).to(BF16).reshape(B * Ta, E).contiguous()
    cond_t = torch.nn.functional.silu(temb32).to(BF16).contiguous()
    d_cond_v, d_cond_t = z32(B * Ta, E), z32(B, E)

    def tables_bwd(lins, dtab, w_ptrs, width, text, cv, dcv, T_):
        """Adjoint of one ``ops.modulation_tables`` call: all its AdaLN linears in two launches."""
        gW, gb = ops.modulation_tables_bwd(dtab, cv, cond_t, w_ptrs, dcv, d_cond_t, len(lins), B, T_, E, width, text)
        gb16 = gb.to(BF16)
        for i, lin in enumerate(lins):
            if lin.weight.requires_grad:
                grads[id(lin.weight)] = gW[i]
            if lin.bias is not None and lin.bias.requires_grad:
                grads[id(lin.bias)] = gb16[i]

    ptr = model._pointer_tables(dev)
    tables_bwd([n.linear for blk in model.transformer_blocks for n in (blk.norm1, blk.norm2)], dmod, ptr["w_blk"], 3 * D,
               mod_text, cond_v, d_cond_v, Ta)
    tables_bwd([model.norm_out.linear], dmodf[None], ptr["w_out"], 2 * D, False, cond_v, d_cond_v, Ta)
    if mv is not None:                         # MVBlock.norm1 has no action term: both row groups see silu(temb)
        wmv, _ = model._mv_pointer_tables(dev)
        tables_bwd([mblk.norm1.linear for mblk in model.mv_blocks], dmv, wmv, 3 * D, mod_text, cond_t, d_cond_t, 1)

    def dsilu(x):
        s = torch.sigmoid(x)
        return s * (1 + x * (1 - s))

    d_pre_v = d_cond_v.view(B, Ta, E) * dsilu(pre_v)
    d_temb = d_pre_v.sum(1) + d_cond_t * dsilu(temb32)
    if sv.action_emb is not None:
        ae = model.action_embed
        d_emb = d_pre_v
        if sv.has_recon and drecon is not None:
            # ActionRecon adjoint (components.py:92-104): un-slice (first row + pad rows dropped), Linear-GELU-Linear
            ar = model.action_recon
            n_out = ar.mlp[2].weight.shape[0]
            dyr = z32(B, Ta * ar.compress_ratio, n_out // ar.compress_ratio)
            dyr[:, 1 + sv.ar_pad:] = drecon.float()
            dyr = dyr.reshape(B * Ta, n_out).contiguous()
            w2r, w0r = ar.mlp[2], ar.mlp[0]
            d_hr = z32(B * Ta, w2r.weight.shape[1])
            gb = z32(n_out)
            gw2 = _acc_grad(grads, w2r.weight) if w2r.weight.requires_grad else torch.zeros_like(w2r.weight, dtype=BF16)
            ops.small_linear_bwd(dyr, sv.ar_h, w2r.weight, gw2, gb, d_hr, B * Ta, n_out, w2r.weight.shape[1])
            f32_to_param_grad(w2r.bias, gb)
            d_ur = (d_hr * _gelu_tanh_grad(sv.ar_u.float())).contiguous()
            gb = z32(w0r.weight.shape[0])
            d_in = z32(B * Ta, E)
            gw0 = _acc_grad(grads, w0r.weight) if w0r.weight.requires_grad else torch.zeros_like(w0r.weight, dtype=BF16)
            ops.small_linear_bwd(d_ur, sv.ar_in, w0r.weight, gw0, gb, d_in, B * Ta, w0r.weight.shape[0], E)
            f32_to_param_grad(w0r.bias, gb)
            d_emb = d_emb + d_in.view(B, Ta, E)
        if nv_ > 1:
            d_emb = d_emb.view(b0, nv_, Ta, E).sum(1)
        if ae.mask:
            # masked samples took the embedding from mask_embed.weight
            f32_to_param_grad(ae.mask_embed.weight, (d_emb * sv.is_mask[:, None, None]).sum((0, 1))[None])
            d_emb = d_emb * (~sv.is_mask)[:, None, None]
        d_emb2 = d_emb.reshape(b0 * Ta, E).contiguous()
        w3, w0 = ae.mlp[3], ae.mlp[0]
        d_h = z32(b0 * Ta, w3.weight.shape[1])
        gb = z32(E)
        gw3 = _acc_grad(grads, w3.weight) if w3.weight.requires_grad else torch.zeros_like(w3.weight, dtype=BF16)
        ops.small_linear_bwd(d_emb2, sv.ae_h, w3.weight, gw3, gb, d_h, b0 * Ta, E, w3.weight.shape[1])
        f32_to_param_grad(w3.bias, gb)
        d_u = (d_h * _gelu_tanh_grad(sv.ae_u.float())).contiguous()
        gb = z32(w0.weight.shape[0])
        gw0 = _acc_grad(grads, w0.weight) if w0.weight.requires_grad else torch.zeros_like(w0.weight, dtype=BF16)
        ops.small_linear_bwd(d_u, sv.ae_in, None, gw0, gb, None, b0 * Ta, w0.weight.shape[0], w0.weight.shape[1])
        f32_to_param_grad(w0.bias, gb)
    if nv_ > 1:
        d_temb = d_temb.view(b0, nv_, E).sum(1)

    def mlp2_bwd(emb_mod, d_out, h1, u1, x_in):
        """TimestepEmbedding (Linear-SiLU-Linear) adjoint for d_out [rows, E]."""
        rows = d_out.shape[0]
        l1, l2 = emb_mod.linear_1, emb_mod.linear_2
        d_h1 = z32(rows, l2.weight.shape[1])
        gb = z32(l2.weight.shape[0])
        g2 = _acc_grad(grads, l2.weight) if l2.weight.requires_grad else torch.zeros_like(l2.weight, dtype=BF16)
        ops.small_linear_bwd(d_out.contiguous(), h1, l2.weight, g2, gb, d_h1, rows, l2.weight.shape[0], l2.weight.shape[1])
        f32_to_param_grad(l2.bias, gb)
        d_u1 = (d_h1 * dsilu(u1.float())).contiguous()
        gb = z32(l1.weight.shape[0])
        g1 = _acc_grad(grads, l1.weight) if l1.weight.requires_grad else torch.zeros_like(l1.weight, dtype=BF16)
        ops.small_linear_bwd(d_u1, x_in, None, g1, gb, None, rows, l1.weight.shape[0], l1.weight.shape[1])
        f32_to_param_grad(l1.bias, gb)

    mlp2_bwd(model.time_embedding, d_temb, sv.te_h1, sv.te_u1, sv.t_emb)
    if sv.has_ofs:
        rows_o = sv.o_emb.shape[0]
        d_o = d_temb if rows_o == d_temb.shape[0] else d_temb.sum(0, keepdim=True)
        mlp2_bwd(model.ofs_embedding, d_o, sv.oe_h1, sv.oe_u1, sv.o_emb)
    if pending:
        # gradients accumulated in the arena: a parameter's first (normally only) contribution goes straight into its segment of
        # the fused optimizer's flat gradient buffer when there is one - one scatter-cast launch for all of them instead of a
        # bf16 copy of the arena and ~45 copy kernels in optimizer.step(); the rest as before
        direct, rest = [], []
        for param, off, n in pending:
            g = _state.grad_view(param) if id(param) not in grads and _SCATTER_GRADS else None
            if g is not None and g.is_contiguous():
                grads[id(param)] = g
                direct.append((off, g))
            else:
                rest.append((param, off, n))
        ops.scatter_f32_to_bf16(arena, direct)
        if rest:
            a16 = arena[:arena_ptr[0]].to(BF16)
            for param, off, n in rest:
                g = a16[off:off + n].view(param.shape)
                if id(param) in grads:
                    grads[id(param)].add_(g)
                else:
                    grads[id(param)] = g
    return grads


class DiTFunction(torch.autograd.Function):
    """Whole-transformer autograd node: forward = HIP forward saving activations, backward = HIP backward.  The
    parameters are passed as inputs so autograd (and DDP's hooks) see their gradients."""

    @staticmethod
    def forward(ctx, model, hidden_states, encoder_hidden_states, controls, timestep, ofs, image_rotary_emb, num_views,
                image_rotary_emb_view, *params):
        with torch.no_grad():
            out, is_mask, recon, sv = forward_train(model, hidden_states, encoder_hidden_states, controls, timestep, ofs,
                                                    image_rotary_emb, num_views, image_rotary_emb_view)
        ctx.model, ctx.sv, ctx.params = model, sv, params
        dev = out.device
        if is_mask is None:
            is_mask = torch.zeros(0, dtype=torch.bool, device=dev)
        ctx.mark_non_differentiable(is_mask)
        if recon is None:
            recon = torch.zeros(0, dtype=out.dtype, device=dev)
            ctx.mark_non_differentiable(recon)
        return out, is_mask, recon

    @staticmethod
    def backward(ctx, dout, _dmask, drecon):
        with torch.no_grad():
            grads = backward(ctx.model, ctx.sv, dout.contiguous(), drecon if ctx.sv.has_recon else None,
                             grad_hook=getattr(ctx.model, "_dp_grad_hook", None))
        ctx.sv = None
        outs = []
        for p_ in ctx.params:
            g = grads.get(id(p_))
            outs.append(g.to(p_.dtype) if (g is not None and p_.requires_grad) else None)
        return (None, None, None, None, None, None, None, None, None, *outs)


def forward_with_grad(model, hidden_states, encoder_hidden_states, controls, timestep, ofs=None, image_rotary_emb=None,
                      num_views=1, image_rotary_emb_view=None):
    params = tuple(p_ for p_ in model.parameters())
    out, is_mask, recon = DiTFunction.apply(model, hidden_states, encoder_hidden_states, controls, timestep, ofs,
                                            image_rotary_emb, num_views, image_rotary_emb_view, *params)
    return out, (is_mask if is_mask.numel() else None), (recon if recon.numel() else None)
