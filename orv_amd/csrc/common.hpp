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
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))),  tanh(u) = 1 - 2 / (1 + exp(2u))
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
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
