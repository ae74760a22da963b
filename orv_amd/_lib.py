"""ctypes binding of liborv_mi355.so (the C ABI declared in include/orv_mi355.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORV_LIB: another build of the same library (developer A/B runs: tools/*.sh); the default is the in-tree build
LIB_PATH = os.environ.get("ORV_LIB") or os.path.join(_HERE, "liborv_mi355.so")


class Groups(Structure):
    _fields_ = [("seq", c_int), ("n_text", c_int), ("per_group", c_int)]


class RowMap(Structure):
    _fields_ = [("rows", c_int), ("bstride", c_int), ("off", c_int)]


class Gemm(Structure):
    _fields_ = [("A", c_void_p), ("lda", c_int), ("W", c_void_p), ("ldw", c_int), ("bias", c_void_p),
                ("C", c_void_p), ("ldc", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("epilogue", c_int),
                ("R", c_void_p), ("ldr", c_int), ("r_mod", c_int), ("gate", c_void_p), ("gate_b", c_long),
                ("gate_g", c_long), ("grp", Groups), ("cmap", RowMap), ("Y", c_void_p), ("ldy", c_int),
                ("qn_gamma_q", c_void_p), ("qn_beta_q", c_void_p), ("qn_gamma_k", c_void_p), ("qn_beta_k", c_void_p),
                ("qn_eps", c_float), ("qn_premul", c_float), ("qn_heads", c_int), ("a_packed", c_int), ("c_packed", c_int)]


class Conv(Structure):
    _fields_ = [("src", c_void_p)] + [(n, c_int) for n in ("B", "Ts", "Hs", "Ws", "C", "T", "H", "W", "kt", "kh", "kw", "stride",
                                                             "pad_lo", "ups_s", "ups_t", "t_shift")]


# name -> (restype, argtypes); every symbol include/orv_mi355.h declares
SIGNATURES = {
    "orv_version": (c_int, []),
    "orv_last_error": (c_char_p, []),
    "orv_device_check": (c_int, [c_int]),
    "orv_timestep_embedding": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "orv_skinny_linear": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, RowMap, c_void_p]),
    "orv_patchify": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_void_p]),
    "orv_unpatchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "orv_gather_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "orv_scatter_gated_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    "orv_add_rows": (c_int, [c_void_p, c_int, RowMap, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "orv_layernorm_modulate": (c_int, [c_void_p, c_int, RowMap, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_long, c_long, Groups, c_int, c_int, c_float, c_void_p]),
    "orv_layernorm_modulate_packed": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, Groups,
                                              c_int, c_int, c_float, c_void_p]),
    "orv_modulation_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_void_p]),
    "orv_qkv_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                             c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "orv_qkv_prep_from": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "orv_gemm_kernel_name": (c_int, [c_int, c_int, c_int, c_int, ctypes.c_char_p, c_int]),
    "orv_gemm_kernel_name_packed": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_char_p, c_int]),
    "orv_gemm_force_tile": (c_int, [c_int, c_int, c_int]),
    "orv_gemm_force_epoch": (c_int, []),
    "orv_gemm_tn_bf16": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "orv_gemm_bf16": (c_int, [POINTER(Gemm), c_void_p]),
    "orv_packed_rows": (c_long, [c_long]),
    "orv_pack_rows16": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p]),
    "orv_unpack_rows16": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "orv_attention_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                  c_float, c_void_p]),
    "orv_attention_fwd_bounded": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "orv_attention_fwd_packed": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "orv_attention_static_limit": (c_float, [c_int]),
    "orv_attention_ws_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "orv_attention_fwd_bounded_ws": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p,
                                             ctypes.c_size_t, c_void_p]),
    "orv_attention_fwd_bounded_dev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "orv_transpose_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "orv_transpose_colsum_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "orv_colsum": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "orv_gated_residual_bwd_scratch": (c_long, [Groups, c_int, c_int]),
    "orv_gated_residual_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, Groups, c_int,
                                       c_int, c_void_p]),
    "orv_layernorm_modulate_bwd_scratch": (c_long, [Groups, c_int, c_int]),
    "orv_layernorm_modulate_bwd": (c_int, [c_void_p, c_void_p, RowMap, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, Groups, c_int,
                                           c_int, c_float, c_void_p]),
    "orv_modulation_tables_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                          c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "orv_small_linear_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_int, c_int, c_int, c_void_p]),
    "orv_adamw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float, c_int,
                          c_void_p, c_void_p]),
    "orv_adamw_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_int, c_float, c_float,
                               c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "orv_adamw_flat_steps": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int,
                                     c_float, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "orv_scatter_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "orv_sumsq": (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    "orv_head_transpose": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "orv_attention_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "orv_qkv_prep_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "orv_sched_step": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                               c_float, c_float, c_float, c_float, c_float, c_float, c_long, c_void_p]),
    "orv_gaussian_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "orv_vae_im2col": (c_int, [c_void_p, c_void_p] + [c_int] * 17 + [c_long, c_long, c_void_p]),
    "orv_conv_gemm_bf16": (c_int, [POINTER(Gemm), POINTER(Conv), c_void_p]),
    "orv_vae_groupnorm_scratch": (c_long, [c_int, c_long, c_int, c_int]),
    "orv_vae_groupnorm_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_void_p]),
    "orv_vae_norm_apply": (c_int, [c_void_p] * 7 + [c_int] * 9 + [c_float, c_int, c_int, c_void_p]),
    "orv_vae_blend": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load liborv_mi355.so once; raise loudly if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C orv_amd/csrc` "
                               "(hipcc --offload-arch=gfx950). orv_amd has no CPU/PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().orv_last_error()
        raise RuntimeError(f"liborv_mi355 {what} failed (rc={rc}): {msg.decode() if msg else ''}")
