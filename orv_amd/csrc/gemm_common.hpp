// Shared by the GEMM translation units (gemm.hip: simple / ring / phased / convolution kernels; gemm_t8.hip: the 16x16x32
// 8-phase kernel): argument block, tile -> workgroup map, device query.
#pragma once
#include "common.hpp"

namespace orv_gemm {

inline int orv_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

struct GemmArgs {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    const bf16_t* bias;
    bf16_t* C; long ldc;
    int M, N, K;
    const bf16_t* R; long ldr; int r_mod;
    const float* gate; long gate_b, gate_g;
    int seq, n_text, per_group;
    int c_rows, c_bstride, c_off;
    bf16_t* Y; long ldy;   // optional second output: the pre-epilogue value acc + bias (saved for backward)
    // epilogue 4 (fused qk LayerNorm of the QKV projection): norm_q / norm_k affine [64], eps, q pre-multiplier, heads
    const bf16_t *qn_gq, *qn_bq, *qn_gk, *qn_bk; float qn_eps, qn_premul; int qn_heads;
    int a_packed, c_packed;   // A / C in the packed P16 layout (include/orv_mi355.h orv_gemm_t; gemm_d8.hip)
    int tiles_m, tiles_n;
    int dbg;  // ORV_GEMM_DBG: 1 = skip main-loop loads, 2 = skip MFMAs (ablation only)
    // Walk the tile list from its end.  A GEMM's A operand was written by the kernel just before it, lowest rows first; the big ones
    // (GELU output 198 MB, q | k | v gradient 149 MB) do not fit the 256 MB Infinity Cache beside the rest of the traffic, and a
    // walk that starts at the OLDEST rows streams through an LRU cache without ever hitting.  Measured in the model, same box,
    // interleaved: FFN2 0.3405 -> 0.3319 ms, out-projection 0.1116 -> 0.1083 ms, LayerNorm-fed GEMMs unchanged (step 41.57 ->
    // 41.12 ms); the 2B training step another 1 % when the backward GEMMs do it too (profiles/r3_gemm_walk_back.txt).
    int walk_back;
    // Start stagger of the persistent t8 workgroups (gemm_t8.hip): workgroup group g = (blockIdx / 8) % stagger_groups begins its first
    // tile g * stagger_ticks (10 ns each) late, so the epilogue store bursts of the groups do not coincide.  0 / 1 groups: off.
    int stagger_groups, stagger_ticks;
    int grid_cap;   // > 0: at most this many persistent workgroups (experiment knob, ORV_T8_GRID)
    int gm;   // super-tile height in tiles (0: GM).  8 for the long-K, narrow GEMM (FFN2: 10 column tiles, 120 K-tiles): in the model
              // 0.3252 -> 0.3178 ms; every other per-block shape is best at or indifferent to 4 (sweep 1 / 2 / 3 / 4 / 6 / 8 / 13 in
              // profiles/r3_gemm_walk_back.txt).  ORV_GEMM_GM overrides.
};

constexpr int BK = 64;
constexpr int GM = 4;  // super-tile height in tiles

// XCD-aware tile mapping (bijective for any grid size): block b runs on XCD b % 8; each XCD gets a contiguous chunk of the
// tile list, ordered in GM x (tiles_n) groups walked m-fastest so concurrent CUs of an XCD share A and W panels in L2.
__device__ __forceinline__ void tile_of_index(const GemmArgs& p, int b, int nb, int& tm, int& tn) {
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, j = b >> 3;
    int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    if (p.walk_back) L = nb - 1 - L;
    const int gm = p.gm > 0 ? p.gm : GM;
    const int per = gm * p.tiles_n;
    const int gid = L / per, rem = L % per;
    const int first_m = gid * gm;
    const int gsize = min(p.tiles_m - first_m, gm);
    tm = first_m + rem % gsize;
    tn = rem / gsize;
}
__device__ __forceinline__ void tile_of_block(const GemmArgs& p, int& tm, int& tn) { tile_of_index(p, blockIdx.x, gridDim.x, tm, tn); }


// Fused qk LayerNorm (epilogue 4) of gemm_t8 / gemm_d8: the two roundings that the vectoriser would otherwise place differently per kernel (it
// fused every second square of one kernel's chain and none of the other's) are written out, so that the kernels agree BIT FOR BIT - a one-clip
// call (q | k | v on gemm_d8) and a four-clip call (the gemm_t8 pair) must give the same clip (tests: d8 == t8, full-depth batch consistency).
__device__ __forceinline__ float qkln_sq(float x, float sq) { return __builtin_fmaf(x, x, sq); }
__device__ __forceinline__ float qkln_affine(float x, float rstd, float ga, float be, float post) {
    return __builtin_fmaf(x * rstd, ga, be) * post;
}

// gemm_t8.hip: launches gemm_t8_kernel<BN, EPI> (bm = 256) or gemm_t8r192_kernel<BN, EPI> (bm = 192), BN = 256 or 192; a.tiles_m / a.tiles_n
// must be set for bm, BN
int launch_t8(const GemmArgs& a, int bn, int epi, hipStream_t st, int bm = 256);
// gemm_t8.hip: TN form C[M, N] (+)= A[K, M]^T . W[K, N] (both operands row-major over the contraction index), BN = 256 or 192
int launch_t8_tn(const GemmArgs& a, int bn, int accumulate, hipStream_t st);
// gemm_t8.hip: the four-wave 256 x 256 experiment (ORV_GEMM_TILE=4,256,256)
int launch_t4(const GemmArgs& a, int epi, hipStream_t st);
// gemm_d8.hip: gemm_d8_kernel<BN, EPI> (A straight to registers, W through four LDS buffers; BN = 256, 192 or 128, K % 192 == 0) or, bm = 192,
// gemm_d8r192_kernel<BN, EPI> (BN = 192 or 128); a.tiles_m must be set for bm
int launch_d8(const GemmArgs& a, int bn, int epi, hipStream_t st, int bm = 256);

}  // namespace orv_gemm
