// Shared device/host helpers for liborv_mi355 (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/orv_mi355.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

void orv_set_error(const char* fmt, ...);
int orv_check_launch(const char* what);

#define ORV_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            orv_set_error(__VA_ARGS__);        \
            return ORV_EINVAL;                 \
        }                                      \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2, RNE: clang lowers the __bf16 vector convert to one v_cvt_pk_bf16_f32 on gfx950
typedef float orv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 orv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    orv_f32x2 v;
    v.x = lo;
    v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, orv_bf16x2));
}

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), is x sigmoid(2u) = x / (1 + exp(-2u)); with -2u log2(e) = x (c0 + c1 x^2):
    // five plain VALU instructions + v_exp_f32 + v_rcp_f32 per element (round 3's form x - x / (1 + exp(2u)) took eight + two; the GELU
    // epilogue is bound by exactly these).  Large |x|: exp2 -> inf gives x * 0 = -0, exp2 -> 0 gives x: no NaN.
    constexpr float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
    const float z = x * __builtin_fmaf(x * x, c1, c0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}
// d/dx of gelu_tanh
__device__ __forceinline__ float gelu_tanh_grad(float x) {
    const float a = 0.7978845608028654f, b = 0.044715f;
    const float x2 = x * x;
    const float e = __builtin_amdgcn_exp2f((2.0f * a * 1.4426950408889634f) * (x + b * x * x2));
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);        // tanh(a (x + b x^3))
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * a * (1.0f + 3.0f * b * x2);
}
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// XCD-aware work-item order for 1-D grids: workgroup L runs on XCD L % 8 (round-robin dispatch); XCD x takes the x-th
// contiguous chunk of the `total` work items, so items that share operands (the query tiles of one attention head, the
// tiles of one GEMM panel group) meet in ONE 4 MB L2 instead of being fetched by all eight.  Bijective for any total.
__device__ __forceinline__ int orv_xcd_item(int L, int total) {
    const int q = total >> 3, r = total & 7, x = L & 7, j = L >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the same sum on the VALU alone (four DPP steps inside each row of 16 lanes, then the gfx950 half-swaps across rows): no LDS crossbar
// round trips (wave_sum above is six dependent ds_bpermute, ~100 cycles each)
__device__ __forceinline__ float wave_sum_valu(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xF, 0xF, true));    // row_mirror
    { const unsigned u = __float_as_uint(v); const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const unsigned u = __float_as_uint(v); const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return v;
}
// sum over the 8 lanes of an aligned group (lane ^ 1, ^ 2, ^ 4) on the VALU alone (DPP): the __shfl_xor form is three dependent ds_bpermute
// round trips through the LDS crossbar, ~100 cycles each
__device__ __forceinline__ float orv_sum8(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xF, 0xF, true));    // row_half_mirror (quads are uniform)
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// group of row s within one batch element (see orv_groups_t)
__device__ __forceinline__ int orv_group_of(int s, int n_text, int per_group) {
    if (s < n_text) return 0;
    return per_group > 0 ? 1 + (s - n_text) / per_group : 1;
}

// async global -> LDS copy of 16 B per lane; LDS destination is wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// 4 bytes per lane: 64 consecutive floats of a vector land at lds_wave_base[0..63]
__device__ __forceinline__ void glds4(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
