"""Thin tensor-level wrappers over the C ABI (one Python function per ``orv_*`` entry point).

Every function takes CUDA(HIP) tensors, checks dtype/contiguity, passes raw device pointers and the current stream to
liborv_mi355.so and returns torch tensors that own the output memory.  torch is plumbing here (device memory + streams);
all arithmetic happens in the HIP kernels.  Nothing in this module can run on CPU tensors.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os
import torch

from ._lib import Gemm, Groups, RowMap, check, lib

BF16 = torch.bfloat16
ACT = {None: 0, "none": 0, "silu": 1, "gelu_tanh": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional live per-launch timing (bench.py's roofline leg): when a dict is installed here, gemm/attention launches are
# bracketed by HIP events recorded on the launch stream (torch's current stream), keyed by kernel + shape.
_TIMELINE = None


def start_timeline():
    global _TIMELINE
    _TIMELINE = {}


def stop_timeline():
    """-> {key: [ms, ...]} of every bracketed launch since start_timeline()."""
    global _TIMELINE
    tl, _TIMELINE = _TIMELINE, None
    torch.cuda.synchronize()
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in (tl or {}).items()}


class _timed:
    def __init__(self, key):
        self.key = key

    def __enter__(self):
        if _TIMELINE is not None:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _TIMELINE is not None:
            self.b.record()
            _TIMELINE.setdefault(self.key, []).append((self.a, self.b))


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"orv_amd.ops: `{name}` must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"orv_amd.ops: `{name}` must be {dtype} (got {t.dtype})")
    return t


def groups(seq: int, n_text: int, per_group: int) -> Groups:
    return Groups(seq, n_text, per_group)


def rowmap(rows: int = 0, bstride: int = 0, off: int = 0) -> RowMap:
    return RowMap(rows, bstride, off)


def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos: bool = True, freq_shift: float = 0.0):
    t = _need(t.contiguous(), torch.float32, "t")
    out = torch.empty(t.numel(), dim, dtype=BF16, device=t.device)
    check(lib().orv_timestep_embedding(_p(t), _p(out), t.numel(), dim, int(flip_sin_to_cos), float(freq_shift),
                                       _stream()), "orv_timestep_embedding")
    return out


def skinny_linear(x, W, bias=None, xb=None, xb_rep=1, act_in=None, act_out=None, out=None, out_f32=False, ldo=None,
                  omap: Optional[RowMap] = None):
    """out[m] = act_out(W . act_in(x[m] + xb[m // xb_rep]) + bias);  x [M,K] bf16, W [N,K] bf16."""
    x = _need(x.contiguous(), BF16, "x")
    W = _need(W.contiguous(), BF16, "W")
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if out_f32 else BF16, device=x.device)
        ldo = N
    check(lib().orv_skinny_linear(_p(x), _p(xb), xb_rep, _p(W), _p(bias), _p(out), M, N, K, ACT[act_in], ACT[act_out],
                                  int(out.dtype == torch.float32), ldo, omap or RowMap(0, 0, 0), _stream()),
          "orv_skinny_linear")
    return out


def patchify(src0, src1=None, p: int = 2, pt: Optional[int] = None):
    src0 = _need(src0.contiguous(), BF16, "src0")
    B, T, c0, H, W = src0.shape
    c1 = 0
    if src1 is not None:
        src1 = _need(src1.contiguous(), BF16, "src1")
        c1 = src1.shape[2]
    ptt = pt or 1
    tok = torch.empty(B, (T // ptt) * (H // p) * (W // p), (c0 + c1) * ptt * p * p, dtype=BF16, device=src0.device)
    check(lib().orv_patchify(_p(src0), c0, _p(src1), c1, _p(tok), B, T, H, W, p, pt or 0, _stream()), "orv_patchify")
    return tok


def unpatchify(x, B, T, C, H, W, p: int = 2, pt: Optional[int] = None):
    x = _need(x.contiguous(), BF16, "x")
    out = torch.empty(B, T, C, H, W, dtype=BF16, device=x.device)
    check(lib().orv_unpatchify(_p(x), _p(out), B, T, C, H, W, p, pt or 0, _stream()), "orv_unpatchify")
    return out


def gather_rows(src, idx, dst, R, D, ld_src=None, ld_dst=None):
    check(lib().orv_gather_rows(_p(src), ld_src or D, _p(idx), _p(dst), ld_dst or D, R, D, _stream()), "orv_gather_rows")
    return dst


def scatter_gated_rows(y, idx, gate, gate_b, x, R, D, seq, n_text, ldy=None, ldx=None):
    check(lib().orv_scatter_gated_rows(_p(y), ldy or D, _p(idx), _p(gate), gate_b, _p(x), ldx or D, R, D, seq, n_text,
                                       _stream()), "orv_scatter_gated_rows")
    return x


def add_rows(a, amap: Optional[RowMap], b, out, col_off, M, D, lda=None, ldb=None, ldo=None):
    _need(a, BF16, "a"), _need(b, BF16, "b"), _need(out, BF16, "out")
    check(lib().orv_add_rows(_p(a), lda or D, amap or RowMap(0, 0, 0), _p(b), ldb or D, _p(out), ldo or D, col_off, M, D,
                             _stream()), "orv_add_rows")
    return out


def layernorm_modulate(x, y, gamma, beta, scale, shift, mod_b, mod_g, grp: Groups, batch, D, eps, ldx=None, ldy=None,
                       xmap: Optional[RowMap] = None, out_packed=False):
    """``out_packed``: ``y`` is a packed P16 buffer [packed_rows(batch * seq), D] (``orv_layernorm_modulate_packed``)."""
    _need(x, BF16, "x"), _need(y, BF16, "y")
    if out_packed:
        check(lib().orv_layernorm_modulate_packed(_p(x), ldx or D, _p(y), _p(gamma), _p(beta), _p(scale), _p(shift), mod_b, mod_g, grp,
                                                  batch, D, float(eps), _stream()), "orv_layernorm_modulate_packed")
        return y
    check(lib().orv_layernorm_modulate(_p(x), ldx or D, xmap or RowMap(0, 0, 0), _p(y), ldy or D, _p(gamma), _p(beta),
                                       _p(scale), _p(shift), mod_b, mod_g, grp, batch, D, float(eps), _stream()),
          "orv_layernorm_modulate")
    return y


def gemm_tn(A, W, C, M, N, K, accumulate=False, lda=None, ldw=None, ldc=None):
    """C[M, N] (+)= A[K, M]^T . W[K, N]  (both operands row-major over the K contraction rows: dW = dY^T X without transposes)."""
    _need(A, BF16, "A"), _need(W, BF16, "W"), _need(C, BF16, "C")
    check(lib().orv_gemm_tn_bf16(_p(A), lda or M, _p(W), ldw or N, _p(C), ldc or N, M, N, K, int(bool(accumulate)), _stream()),
          "orv_gemm_tn_bf16")
    return C


def modulation_tables(temb, action_emb, w_ptrs, b_ptrs, n_tab, B, T, E, width, text, out=None):
    """All AdaLN tables of a forward: out fp32 [n_tab, B, 1+T, width]; w_ptrs/b_ptrs int64 device tensors of pointers."""
    _need(temb, BF16, "temb")
    if out is None:
        out = torch.zeros(n_tab, B, 1 + T, width, dtype=torch.float32, device=temb.device)
    check(lib().orv_modulation_tables(_p(temb), _p(action_emb), _p(w_ptrs), _p(b_ptrs), _p(out), n_tab, B, T, E, width,
                                      int(text), _stream()), "orv_modulation_tables")
    return out


def qkv_prep(qkv, vT, gq, bq, gk, bk, rope: Optional[Tuple[torch.Tensor, torch.Tensor]], B, S, H, n_text, s_pad, eps,
             q_premul: float = 1.0, src=None):
    """In place on ``qkv``, or (``src`` given) out of place: reads ``src``, writes q' / k' / v to ``qkv``."""
    _need(qkv, BF16, "qkv")
    if vT is not None:
        _need(vT, BF16, "vT")
    cos = sin = None
    if rope is not None:
        cos, sin = (_need(r.contiguous(), torch.float32, "rope") for r in rope)
    check(lib().orv_qkv_prep_from(_p(qkv if src is None else src), _p(qkv), _p(vT), _p(gq), _p(bq), _p(gk), _p(bk), _p(cos),
                                  _p(sin), B, S, H, n_text, s_pad, float(eps), float(q_premul), _stream()), "orv_qkv_prep")


def packed_rows(rows: int) -> int:
    """Row slots of a packed P16 buffer holding ``rows`` rows (include/orv_mi355.h ``orv_gemm_t``: rounded up to the 256-row GEMM tile)."""
    return int(lib().orv_packed_rows(int(rows)))


def pack_rows16(src, M, K, dst=None, ld_src=None):
    """Row-major ``src`` [M, K] -> packed P16 ``dst`` [packed_rows(M), K] (bit copies, padding rows zero)."""
    _need(src, BF16, "src")
    if dst is None:
        dst = torch.empty(packed_rows(M), K, dtype=BF16, device=src.device)
    check(lib().orv_pack_rows16(_p(src), ld_src or K, _p(_need(dst, BF16, "dst")), M, K, _stream()), "orv_pack_rows16")
    return dst


def unpack_rows16(src, M, K, dst=None, ld_dst=None):
    """Packed P16 ``src`` -> row-major ``dst`` [M, K]."""
    _need(src, BF16, "src")
    if dst is None:
        dst = torch.empty(M, K, dtype=BF16, device=src.device)
    check(lib().orv_unpack_rows16(_p(src), _p(_need(dst, BF16, "dst")), ld_dst or K, M, K, _stream()), "orv_unpack_rows16")
    return dst


def gemm_kernel_name(M, N, K, epilogue=0, a_packed=False, c_packed=False) -> Optional[str]:
    """Kernel symbol ``gemm`` launches for this call, or None when no kernel takes the packed-operand combination."""
    import ctypes
    buf = ctypes.create_string_buffer(96)
    if a_packed or c_packed:
        rc = lib().orv_gemm_kernel_name_packed(M, N, K, epilogue, int(bool(a_packed)), int(bool(c_packed)), buf, 96)
        return buf.value.decode() if rc == 0 else None
    check(lib().orv_gemm_kernel_name(M, N, K, epilogue, buf, 96), "orv_gemm_kernel_name")
    return buf.value.decode()


def gemm(A, W, bias, C, M, N, K, epilogue=0, R=None, r_mod=0, gate=None, gate_b=0, gate_g=0, grp: Optional[Groups] = None,
         cmap: Optional[RowMap] = None, lda=None, ldw=None, ldc=None, ldr=None, Y=None, ldy=None, qknorm=None,
         a_packed=False, c_packed=False):
    """``qknorm`` = (gamma_q, beta_q, gamma_k, beta_k, eps, q_premul, heads) with ``epilogue=4``: the QKV projection with the
    per-head qk LayerNorm fused (no RoPE).  ``a_packed`` / ``c_packed``: A is read / C is written in the packed P16 layout
    (include/orv_mi355.h ``orv_gemm_t``; buffers of ``packed_rows(M)`` row slots)."""
    _need(A, BF16, "A"), _need(W, BF16, "W"), _need(C, BF16, "C")
    g = Gemm()
    if qknorm is not None:
        gq, bq, gk, bk, eps, premul, heads = qknorm
        g.qn_gamma_q, g.qn_beta_q, g.qn_gamma_k, g.qn_beta_k = _p(gq), _p(bq), _p(gk), _p(bk)
        g.qn_eps, g.qn_premul, g.qn_heads = float(eps), float(premul), int(heads)
    g.A, g.lda, g.W, g.ldw, g.bias = _p(A), lda or K, _p(W), ldw or K, _p(bias)
    g.C, g.ldc, g.M, g.N, g.K, g.epilogue = _p(C), ldc or N, M, N, K, epilogue
    g.R, g.ldr, g.r_mod = _p(R), ldr or N, r_mod
    g.gate, g.gate_b, g.gate_g = _p(gate), gate_b, gate_g
    g.grp = grp or Groups(0, 0, 0)
    g.cmap = cmap or RowMap(0, 0, 0)
    g.Y, g.ldy = _p(Y), ldy or N
    g.a_packed, g.c_packed = int(bool(a_packed)), int(bool(c_packed))
    with _timed(("gemm", M, N, K, epilogue) + ((int(bool(a_packed)), int(bool(c_packed))) if (a_packed or c_packed) else ())):
        check(lib().orv_gemm_bf16(g, _stream()), "orv_gemm_bf16")
    return C


def linear(x2d, W, bias=None, epilogue=0):
    """Convenience: dense [M,K] x [N,K]^T (+bias, optional GELU); pads K to a multiple of 64 for odd test shapes."""
    M, K = x2d.shape
    N = W.shape[0]
    if K % 64:
        pad = 64 - K % 64
        x2d = torch.nn.functional.pad(x2d, (0, pad))
        W = torch.nn.functional.pad(W, (0, pad))
        K += pad
    x2d, W = x2d.contiguous(), W.contiguous()
    C = torch.empty(M, N, dtype=BF16, device=x2d.device)
    return gemm(x2d, W, bias, C, M, N, K, epilogue)


def attention_ws_bytes(B, S, H):
    """Workspace bytes ``attention_fwd(..., ws=)`` wants for this shape (0: the shape's grid is not key-split)."""
    return int(lib().orv_attention_ws_bytes(int(B), int(S), int(H)))


def attention_packed_ok(score_bound, scale) -> bool:
    """Can ``attention_fwd(..., out_packed=True)`` run this call?  (Only the shift-free ping-pong kernel writes the packed layout.)"""
    return (score_bound is not None and 0.0 < float(score_bound) <= float(lib().orv_attention_static_limit(1))
            and abs(float(scale) * 1.4426950408889634 - 1.0) < 1e-6
            and os.environ.get("ORV_ATTN_PP", "1") != "0" and os.environ.get("ORV_ATTN_STATIC", "1") != "0")


def attention_fwd(qkv, vT, out, B, S, H, s_pad, scale, lse=None, ld_qkv=None, ld_out=None, score_bound=None, score_bound_dev=None, ws=None,
                  out_packed=False):
    """``score_bound`` (V in place only): a guaranteed upper bound of |q . k| * scale * log2(e) over the whole call - the kernel
    then runs its fixed-shift softmax when the bound is small enough (``orv_attention_fwd_bounded``).  ``score_bound_dev``: the
    same bound as a one-element fp32 DEVICE tensor (``orv_attention_fwd_bounded_dev``: no host read; training).
    ``out_packed``: ``out`` is a packed P16 buffer [packed_rows(B S), H 64] (``orv_attention_fwd_packed``; check ``attention_packed_ok``)."""
    _need(qkv, BF16, "qkv"), _need(out, BF16, "out")
    if out_packed:
        with _timed(("attention", B, S, H)):
            check(lib().orv_attention_fwd_packed(_p(qkv), ld_qkv or 3 * H * 64, _p(out), _p(lse), B, S, H, float(scale), float(score_bound),
                                                 _stream()), "orv_attention_fwd_packed")
        return out
    if score_bound_dev is not None and vT is None:
        _need(score_bound_dev, torch.float32, "score_bound_dev")
        if score_bound_dev.numel() != 1:
            raise ValueError("attention_fwd: score_bound_dev must hold exactly one fp32 value")
        with _timed(("attention", B, S, H)):
            check(lib().orv_attention_fwd_bounded_dev(_p(qkv), ld_qkv or 3 * H * 64, _p(out), ld_out or H * 64, _p(lse), B, S, H,
                                                      float(scale), _p(score_bound_dev), _stream()), "orv_attention_fwd_bounded_dev")
        return out
    if score_bound is not None and vT is None and ws is not None:
        # key-split last round (orv_attention_fwd_bounded_ws): ws from attention_ws_bytes(B, S, H), reusable on one stream
        if ws.dtype != torch.uint8 or not ws.is_cuda:
            raise ValueError("attention_fwd: ws must be a uint8 CUDA tensor")
        with _timed(("attention", B, S, H)):
            check(lib().orv_attention_fwd_bounded_ws(_p(qkv), ld_qkv or 3 * H * 64, _p(out), ld_out or H * 64, _p(lse), B, S, H,
                                                     float(scale), float(score_bound), _p(ws), ws.numel(), _stream()),
                  "orv_attention_fwd_bounded_ws")
        return out
    if score_bound is not None and vT is None:
        with _timed(("attention", B, S, H)):
            check(lib().orv_attention_fwd_bounded(_p(qkv), ld_qkv or 3 * H * 64, _p(out), ld_out or H * 64, _p(lse), B, S, H,
                                                  float(scale), float(score_bound), _stream()), "orv_attention_fwd_bounded")
        return out
    if vT is not None:               # legacy form: pre-transposed V (attn_fwd_v1); None = V read in place (attn_fwd_v2)
        _need(vT, BF16, "vT")
    with _timed(("attention", B, S, H)):
        check(lib().orv_attention_fwd(_p(qkv), ld_qkv or 3 * H * 64, _p(vT), _p(out), ld_out or H * 64, _p(lse), B, S,
                                      H, s_pad, float(scale), _stream()), "orv_attention_fwd")
    return out


def sched_step(x, v_c, v_u, guidance_scale, old_x0, noise, sa, sb, m3, m4, cx, cd, cn, want_x0=True):
    x = _need(x.contiguous(), BF16, "x")
    v_c = _need(v_c.contiguous(), BF16, "v_c")
    if v_u is not None:
        v_u = _need(v_u.contiguous(), BF16, "v_u")
    if old_x0 is not None:
        old_x0 = _need(old_x0.contiguous(), torch.float32, "old_x0")
    if noise is not None:
        noise = _need(noise.contiguous(), torch.float32, "noise")
    for name, o in (("v_c", v_c), ("v_u", v_u), ("old_x0", old_x0), ("noise", noise)):
        if o is not None and o.numel() != x.numel():
            raise ValueError(f"sched_step: {name} has {o.numel()} elements, x has {x.numel()}")
    x_out = torch.empty_like(x)
    x0 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_x0 else None
    check(lib().orv_sched_step(_p(x), _p(v_c), _p(v_u), float(guidance_scale), _p(old_x0), _p(noise), _p(x_out), _p(x0),
                               float(sa), float(sb), float(m3), float(m4), float(cx), float(cd), float(cn), x.numel(),
                               _stream()), "orv_sched_step")
    return x_out, x0


def gaussian_sample(moments, eps, scale):
    moments = _need(moments.contiguous(), BF16, "moments")
    eps = _need(eps.contiguous(), torch.float32, "eps")
    B, C2, F, H, W = moments.shape
    out = torch.empty(B, F, C2 // 2, H, W, dtype=BF16, device=moments.device)
    check(lib().orv_gaussian_sample(_p(moments), _p(eps), _p(out), B, C2 // 2, F, H * W, float(scale), _stream()),
          "orv_gaussian_sample")
    return out


# ---- backward (training) ----------------------------------------------------------------------------------------------
_TPAD = 64 if os.environ.get("ORV_TRANSPOSE_PAD", "128") == "64" else 128      # A/B switch: 64 = the round-1..5 padding


def transpose(src, R, C, ld_dst=None, out=None, ld_src=None, colsum=None):
    """[R, C] bf16 -> [C, ld_dst] (zero padded): K-contiguous operand for dgrad/wgrad.  ld_dst defaults to the row stride of ``out`` when
    one is given, else to R rounded up to 128 - the t8 GEMM kernels need K % 128 == 0, and a contraction padded to 64 only
    (CogVideoX1.5-5B at B = 4: 7048 tokens -> 7104) sent every weight-gradient GEMM of configs[4] to the older ring kernel (round 6:
    profiles/r6_train_5b_ckpt_kernel_stats_summary.txt, 17 % of the step in gemm_pp_kernel).
    ``colsum`` (fp32 [C], accumulated into): the column sums of ``src`` from the same pass (bias gradient beside dY^T)."""
    _need(src, BF16, "src")
    if ld_dst is None:
        ld_dst = int(out.stride(0)) if out is not None else (R + _TPAD - 1) // _TPAD * _TPAD
    if out is None:
        out = torch.empty(C, ld_dst, dtype=BF16, device=src.device)
    if colsum is not None:
        _need(colsum, torch.float32, "colsum")
        if colsum.numel() < C or not colsum.is_contiguous():
            raise ValueError(f"transpose: colsum needs {C} contiguous fp32 elements")
        if os.environ.get("ORV_TRANSPOSE_COLSUM", "1") == "0":       # A/B switch: two passes over src
            check(lib().orv_transpose_bf16(_p(src), ld_src or C, _p(out), ld_dst, R, C, _stream()), "orv_transpose_bf16")
            check(lib().orv_colsum(_p(src), ld_src or C, _p(colsum), R, C, _stream()), "orv_colsum")
            return out
        check(lib().orv_transpose_colsum_bf16(_p(src), ld_src or C, _p(out), ld_dst, R, C, _p(colsum), _stream()),
              "orv_transpose_colsum_bf16")
        return out
    check(lib().orv_transpose_bf16(_p(src), ld_src or C, _p(out), ld_dst, R, C, _stream()), "orv_transpose_bf16")
    return out


_scatter_tables = {}


def scatter_f32_to_bf16(src, segments):
    """``segments``: [(offset into ``src`` (fp32, flat), destination bf16 tensor (contiguous))] - every destination receives
    bf16(src[offset : offset + numel]) in ONE launch.  The device-side tables are cached per (offsets, destinations) list: the
    backward of a given model produces the same list every step."""
    _need(src, torch.float32, "src")
    if not segments:
        return
    for o, d in segments:                    # every call: a recycled address must not smuggle another dtype past the cached table
        _need(d, BF16, "destination")
        if not d.is_contiguous() or o < 0 or o + d.numel() > src.numel():
            raise ValueError("scatter_f32_to_bf16: destinations must be contiguous and sources inside `src`")
    key = (src.device.index, tuple((int(o), d.data_ptr(), d.numel()) for o, d in segments))
    tab = _scatter_tables.get(key)
    if tab is None:
        if len(_scatter_tables) > 16:
            _scatter_tables.clear()
        dev = src.device
        tab = (torch.tensor([k[0] for k in key[1]], dtype=torch.int64, device=dev),
               torch.tensor([k[1] for k in key[1]], dtype=torch.int64, device=dev),
               torch.tensor([k[2] for k in key[1]], dtype=torch.int32, device=dev), max(k[2] for k in key[1]))
        _scatter_tables[key] = tab
    check(lib().orv_scatter_f32_to_bf16(_p(src), _p(tab[0]), _p(tab[1]), _p(tab[2]), len(segments), tab[3], _stream()),
          "orv_scatter_f32_to_bf16")


def colsum(src, out, R, C, ld=None):
    _need(src, BF16, "src"), _need(out, torch.float32, "out")
    check(lib().orv_colsum(_p(src), ld or C, _p(out), R, C, _stream()), "orv_colsum")
    return out


def gated_residual_bwd(dout, y, gate, dgate, dy, mod_b, mod_g, grp, batch, D):
    scratch = torch.empty(lib().orv_gated_residual_bwd_scratch(grp, batch, D), dtype=torch.float32, device=dout.device)
    check(lib().orv_gated_residual_bwd(_p(dout), _p(y), _p(gate), _p(dgate), _p(dy), _p(scratch), mod_b, mod_g, grp, batch, D,
                                       _stream()), "orv_gated_residual_bwd")
    return dy


def layernorm_modulate_bwd(dy, x, dres, dx, gamma, beta, scale, dscale, dshift, dgamma, dbeta, mod_b, mod_g, grp, batch, D,
                           eps, xmap: Optional[RowMap] = None):
    scratch = None
    if dgamma is not None or dbeta is not None or scale is not None:
        scratch = torch.empty(lib().orv_layernorm_modulate_bwd_scratch(grp, batch, D), dtype=torch.float32, device=dy.device)
    check(lib().orv_layernorm_modulate_bwd(_p(dy), _p(x), xmap or RowMap(0, 0, 0), _p(dres), _p(dx), _p(gamma), _p(beta),
                                           _p(scale), _p(dscale), _p(dshift), _p(dgamma), _p(dbeta), _p(scratch), mod_b, mod_g,
                                           grp, batch, D, float(eps), _stream()), "orv_layernorm_modulate_bwd")
    return dx


def modulation_tables_bwd(dtab, cond_v, cond_t, w_ptrs, d_cond_v, d_cond_t, n_tab, B, T, E, width, text):
    """Adjoint of ``modulation_tables``: returns (gW bf16 [n_tab, width*(1+text), E], gb fp32 [n_tab, width*(1+text)])."""
    _need(dtab, torch.float32, "dtab"), _need(cond_v, BF16, "cond_v")
    ntot = width * (2 if text else 1)
    gW = torch.empty(n_tab, ntot, E, dtype=BF16, device=dtab.device)
    gb = torch.empty(n_tab, ntot, dtype=torch.float32, device=dtab.device)
    check(lib().orv_modulation_tables_bwd(_p(dtab), _p(cond_v), _p(cond_t), _p(w_ptrs), _p(gW), _p(gb), _p(d_cond_v),
                                          _p(d_cond_t), n_tab, B, T, E, width, int(text), _stream()),
          "orv_modulation_tables_bwd")
    return gW, gb


def small_linear_bwd(dy, x, W, dW, db, dx, R, N, K, accumulate=True, ldy=None, ldx=None, lddx=None):
    _need(dy, torch.float32, "dy")
    check(lib().orv_small_linear_bwd(_p(dy), ldy or N, _p(x), ldx or K, _p(W), _p(dW), _p(db), _p(dx), lddx or K, R, N, K,
                                     int(accumulate), _stream()), "orv_small_linear_bwd")


def adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, clip_coef=None):
    check(lib().orv_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                          float(weight_decay), int(step), _p(clip_coef), _stream()), "orv_adamw")


def adamw_flat(p, g, m, v, seg_start, seg_active, lr, beta1, beta2, eps, weight_decay, step, clip_coef=None, seg_step=None):
    check(lib().orv_adamw_flat_steps(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(seg_start), _p(seg_active), _p(seg_step),
                                     seg_active.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                     float(weight_decay), int(step), _p(clip_coef), _stream()), "orv_adamw_flat_steps")


def sumsq(g, out):
    check(lib().orv_sumsq(_p(g), g.numel(), _p(out), _stream()), "orv_sumsq")


def head_transpose(src, col0, dst, B, S, H, s_pad, ld=None):
    check(lib().orv_head_transpose(_p(src), ld or src.shape[-1], col0, _p(dst), B, S, H, s_pad, _stream()),
          "orv_head_transpose")
    return dst


def attention_bwd(qkv, qT, kT, out, dout, doT, lse, neg_lse2, neg_delta, dqkv, B, S, H, s_pad, scale):
    check(lib().orv_attention_bwd(_p(qkv), 3 * H * 64, _p(qT), _p(kT), _p(out), _p(dout), H * 64, _p(doT), _p(lse),
                                  _p(neg_lse2), _p(neg_delta), _p(dqkv), 3 * H * 64, B, S, H, s_pad, float(scale), _stream()),
          "orv_attention_bwd")
    return dqkv


def qkv_prep_bwd(qkv_raw, dqkv, gq, gk, rope, dgq, dbq, dgk, dbk, B, S, H, n_text, eps):
    cos = sin = None
    if rope is not None:
        cos, sin = rope
    nblk = ((S + 63) // 64) * H * B
    scratch = torch.empty(nblk * 256, dtype=torch.float32, device=dqkv.device)
    check(lib().orv_qkv_prep_bwd(_p(qkv_raw), _p(dqkv), _p(gq), _p(gk), _p(cos), _p(sin), _p(dgq), _p(dbq), _p(dgk), _p(dbk),
                                 _p(scratch), B, S, H, n_text, float(eps), _stream()), "orv_qkv_prep_bwd")


# ---- VAE building blocks (vae.hip): channels-last activations [B, T, H, W, C] ----
def vae_im2col(src, dst, B, Ts, Hs, Ws, C, T, H, W, kt, kh, kw, stride, pad_lo, ups_s, ups_t, t_shift, Kpad, m0, mc):
    _need(src, BF16, "src"), _need(dst, BF16, "dst")
    check(lib().orv_vae_im2col(_p(src), _p(dst), B, Ts, Hs, Ws, C, T, H, W, kt, kh, kw, stride, pad_lo, int(ups_s), int(ups_t),
                               int(t_shift), Kpad, int(m0), int(mc), _stream()), "orv_vae_im2col")
    return dst


def conv_gemm(src, Wp, bias, out, B, Ts, Hs, Ws, C, T, H, W, kt, kh, kw, stride, pad_lo, ups_s, ups_t, t_shift, N, R=None, ldr=None):
    """Implicit-GEMM convolution (no patch matrix): out[B*T*H*W, N] = conv(src) + bias (+ R)."""
    from ._lib import Conv
    _need(src, BF16, "src"), _need(Wp, BF16, "W"), _need(out, BF16, "out")
    M, K = B * T * H * W, kt * kh * kw * C
    g = Gemm()
    g.A, g.lda, g.W, g.ldw, g.bias = None, 0, _p(Wp), K, _p(bias)
    g.C, g.ldc, g.M, g.N, g.K, g.epilogue = _p(out), N, M, N, K, 2 if R is not None else 0
    g.R, g.ldr, g.r_mod, g.gate = _p(R), ldr or N, 0, None
    g.grp, g.cmap = Groups(0, 0, 0), RowMap(0, 0, 0)
    c = Conv(_p(src), B, Ts, Hs, Ws, C, T, H, W, kt, kh, kw, stride, pad_lo, int(ups_s), int(ups_t), int(t_shift))
    check(lib().orv_conv_gemm_bf16(g, c, _stream()), "orv_conv_gemm_bf16")
    return out


def vae_groupnorm_stats(x, B, N, C, G):
    """-> sums fp32 [B, G, 2] (sum, sum of squares per group), deterministic."""
    _need(x, BF16, "x")
    sums = torch.empty(B, G, 2, dtype=torch.float32, device=x.device)
    scratch = torch.empty(lib().orv_vae_groupnorm_scratch(B, int(N), C, G), dtype=torch.float32, device=x.device)
    check(lib().orv_vae_groupnorm_stats(_p(x), _p(sums), _p(scratch), B, int(N), C, G, _stream()), "orv_vae_groupnorm_stats")
    return sums


def vae_blend(a, b, extent, horizontal):
    """In-place seam blend of channels-last tiles a, b [B, T, H, W, C] (diffusers blend_v / blend_h): see orv_vae_blend."""
    _need(a, BF16, "a"), _need(b, BF16, "b")
    if not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("vae_blend: tiles must be contiguous")
    B, T, Ha, Wa, C = a.shape
    _, _, Hb, Wb, _ = b.shape
    check(lib().orv_vae_blend(_p(a), _p(b), B * T, Ha, Wa, Hb, Wb, C, int(extent), int(bool(horizontal)), _stream()), "orv_vae_blend")
    return b


def vae_norm_apply(x, out, sums, gamma, beta, zy, zb, B, T, H, W, C, G, Tz, hz, wz, eps, silu, out_lead=0):
    """``out`` is [B, out_lead + T, H, W, C]; frames [0, out_lead) of each clip are left untouched."""
    _need(x, BF16, "x"), _need(out, BF16, "out"), _need(gamma, BF16, "gamma"), _need(beta, BF16, "beta")
    check(lib().orv_vae_norm_apply(_p(x), _p(out), _p(sums), _p(gamma), _p(beta), _p(zy), _p(zb), B, T, H, W, C, G, Tz, hz, wz,
                                   float(eps), int(bool(silu)), int(out_lead), _stream()), "orv_vae_norm_apply")
    return out
