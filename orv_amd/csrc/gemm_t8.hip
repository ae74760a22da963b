// bf16 GEMM  C = epilogue(A[M,K] . W[N,K]^T + bias), the 256-row "t8" kernel: the 8-phase schedule of
// /opt/skills/guides/cdna_hip_programming.md 5 ("The 256^2 8-phase template") in the form the control probe
// (tools/probe_gemm_template.cpp, profiles/r3_gemm_template_control_ab.txt) measured 11-18 % ahead of gemm_ph_kernel on the
// same box: v_mfma_f32_16x16x32_bf16, 8 waves as 2 (M) x 4 (N), a wave owns 128 rows x BN/4 columns, BK = 64, operands staged by
// global_load_lds into st_16x32 subtiles (16 rows x 32 k = 1 KiB = one DMA instruction of a wave; byte ^= ((byte >> 9) & 1) << 5
// applied to the DMA SOURCE lane map and to the ds_read address), two LDS buffers, the second half of the workgroup (waves 4-7: the
// second wave of every SIMD) one barrier behind the first, ONE counted vmcnt per K-tile.  On top of the template:
//   * persistent: one workgroup per CU walks the XCD-aware tile list; every half-tile DMA stream keeps its own (tile, K-tile)
//     cursor and simply continues into the next output tile, so a tile's first two K-tiles land under the previous epilogue
//     (K = 1920 is only 30 K-tiles: the exposed prologue was ~10 % of such a tile).
//   * C^T accumulators (W fragment as the MFMA A operand): a lane holds 4 consecutive output columns of one row per 16x16 block.
//     Which W row sits in which LDS row is free (the DMA source address is per lane), so the W rows of a wave are PERMUTED on the
//     way in such that lane group g = lane >> 4 ends up with 8 CONTIGUOUS columns per block pair: 16-byte stores / residual
//     loads, 64 contiguous bytes per row and instruction, no cross-lane exchange in the epilogue.
//   * BN = 256: 4 phases per K-tile exactly as the template (quadrants (m0,n0) (m1,n0) (m1,n1) (m0,n1), loads 12 / 8 / 4 / 0
//     ds_read_b128 + one half-tile of DMA per phase, vmcnt(6)).
//     BN = 192 (N = 1920 / 5760 of the 2B model: 510 / 1530 tiles): a wave owns 128 x 48 = 8 x 3 blocks; 3 phases per K-tile of
//     16 MFMAs each - (m0, n01) | (m0, n2) + (m1, n2) | (m1, n01) - with loads 12 / 10 / 0 and DMA 3 / 2 / 2 instructions, vmcnt(4).
//   * epilogues 0-3 as gemm.hip (bias, GELU, gated residual, GELU adjoint; optional Y = acc + bias; row scatter); epilogue 4
//     (fused qk LayerNorm) for BN = 256, where a wave's 64 columns are exactly one head.
// Requires K % 128 == 0 and N % BN == 0 (the tile chooser checks).
#include "gemm_common.hpp"

namespace {
using namespace orv_gemm;
#ifdef ORV_T8_SCHED3
constexpr bool T8_SCHED3_ON = true;
#else
constexpr bool T8_SCHED3_ON = false;
#endif

__device__ __forceinline__ float sum_xor16(float v) {      // v(lane) + v(lane ^ 16)
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float v) {      // v(lane) + v(lane ^ 32)
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
    f[0] = bf2f(u.x & 0xffff); f[1] = bf2f(u.x >> 16); f[2] = bf2f(u.y & 0xffff); f[3] = bf2f(u.y >> 16);
    f[4] = bf2f(u.z & 0xffff); f[5] = bf2f(u.z >> 16); f[6] = bf2f(u.w & 0xffff); f[7] = bf2f(u.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

// Column layout of a wave's BN/4 columns (relative to nbase = tn * BN + wc * BN/4), lane group g = lane >> 4, register e:
//   BN = 256: block blk = 2 P + t (pair P = 0, 1):  column 32 P + 8 g + 4 t + e      -> 8-column pieces at 32 P + 8 g
//   BN = 192: blocks 0, 1 (pair 0):                 column 8 g + 4 blk + e           -> one 8-column piece at 8 g
//             block 2:                              column 32 + 4 g + e              -> one 4-column piece
template <int BN, int EPI, int MB = 4>      // MB: 16-row blocks per m half of a wave (4: 256-row tiles, 3: 192-row tiles)
__device__ __forceinline__ void t8_epilogue(const GemmArgs& p, f32x4 (&acc)[2][MB][BN / 64], const int mbase, const int nbase, const int lane) {
    constexpr int NP = BN == 256 ? 2 : 1;            // 8-column pieces per row
    constexpr bool TAIL = BN == 192;                 // plus one 4-column piece
    const int g = lane >> 4, r16 = lane & 15;
#ifdef ORV_T8_ABL_NOSTORE    // ablation build (tools/t8_epi_abl.sh, wrong results): the epilogue computes everything, no store is executed
    const bool st_ok = p.M < 0;
#else
    constexpr bool st_ok = true;
#endif
    int col8[NP];
#pragma unroll
    for (int P = 0; P < NP; ++P) col8[P] = nbase + 32 * P + 8 * g;
    const int col4 = nbase + 32 + 4 * g;

    // bias of this lane's columns (vector loads: the column depends on the lane group)
    float b8[NP][8], b4[4];
#pragma unroll
    for (int P = 0; P < NP; ++P)
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[P][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) b4[e] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int P = 0; P < NP; ++P) unpack8(*(const uint4*)(p.bias + col8[P]), b8[P]);
        if (TAIL) {
            const uint2 u = *(const uint2*)(p.bias + col4);
            b4[0] = bf2f(u.x & 0xffff); b4[1] = bf2f(u.x >> 16); b4[2] = bf2f(u.y & 0xffff); b4[3] = bf2f(u.y >> 16);
        }
    }

    if constexpr (EPI == 4) {
        // fused qk LayerNorm(64) (diffusers Attention.norm_q / norm_k as called at cogvideox_control.py:243-247) + the softmax
        // pre-multiplier of q: BN = 256, the wave's 64 columns are ONE head of q | k | v; a row's 64 values sit in the four lanes
        // r16 + 16 g (16 each): statistics = lane-local sums + two half-swaps.
        static_assert(BN == 256 || EPI != 4, "epilogue 4 needs whole heads per wave");
        const int region = __builtin_amdgcn_readfirstlane(nbase / (p.qn_heads * 64));     // 0 = q, 1 = k, 2 = v
        const bf16_t* gam = region == 0 ? p.qn_gq : p.qn_gk;
        const bf16_t* bet = region == 0 ? p.qn_bq : p.qn_bk;
        const float post = region == 0 ? p.qn_premul : 1.f;
        float ga[NP][8], be[NP][8];
#pragma unroll
        for (int P = 0; P < NP; ++P) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { ga[P][e] = 1.f; be[P][e] = 0.f; }
            if (region < 2 && gam) unpack8(*(const uint4*)(gam + 32 * P + 8 * g), ga[P]);
            if (region < 2 && bet) unpack8(*(const uint4*)(bet + 32 * P + 8 * g), be[P]);
        }
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mbase + mh * (16 * MB) + mb * 16 + r16;
                const bool valid = m < p.M;              // the four lanes of a row agree
                const long orow = min(m, p.M - 1);
                float v[NP][8];
                float s = 0.f;
#pragma unroll
                for (int P = 0; P < NP; ++P)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[P][e] = acc[mh][mb][2 * P + (e >> 2)][e & 3] + b8[P][e]; s += v[P][e]; }
                if (p.Y && valid && st_ok) {
#pragma unroll
                    for (int P = 0; P < NP; ++P) *(uint4*)(p.Y + orow * p.ldy + col8[P]) = pack8(v[P]);
                }
                if (region < 2) {
                    const float mean = sum_xor32(sum_xor16(s)) * (1.f / 64.f);
                    float sq = 0.f;
#pragma unroll
                    for (int P = 0; P < NP; ++P)
#pragma unroll
                        for (int e = 0; e < 8; ++e) { v[P][e] -= mean; sq = qkln_sq(v[P][e], sq); }      // explicit fma chain: gemm_common.hpp
                    const float rstd = rsqrtf(sum_xor32(sum_xor16(sq)) * (1.f / 64.f) + p.qn_eps);
#pragma unroll
                    for (int P = 0; P < NP; ++P)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[P][e] = qkln_affine(v[P][e], rstd, ga[P][e], be[P][e], post);
                }
                if (valid && st_ok) {
#pragma unroll
                    for (int P = 0; P < NP; ++P) *(uint4*)(p.C + orow * p.ldc + col8[P]) = pack8(v[P]);
                }
            }
        return;
    } else {
        // gate row cache (EPI 2): 16-row blocks almost always lie inside one (batch element, token group) and consecutive blocks
        // share it - the fp32 gate values of this lane's columns are reloaded only when the block's gate row changes
        float g8[NP][8], g4[4];
        const float* g_cached = nullptr;
#pragma unroll
        for (int P = 0; P < NP; ++P)
#pragma unroll
            for (int e = 0; e < 8; ++e) g8[P][e] = 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] = 1.f;

#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            // row operands of the four 16-row blocks of this half are requested together
            long orow[MB], rr[MB];
            bool valid[MB];
            uint4 r8[MB][NP];
            uint2 r4[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mbase + mh * (16 * MB) + mb * 16 + r16;
                valid[mb] = m < p.M;
                const int mc = min(m, p.M - 1);
                orow[mb] = mc;
                if (p.c_rows > 0) orow[mb] = (long)(mc / p.c_rows) * p.c_bstride + p.c_off + mc % p.c_rows;
                rr[mb] = p.r_mod > 0 ? mc % p.r_mod : orow[mb];
                if (EPI == 2 || EPI == 3) {
#pragma unroll
                    for (int P = 0; P < NP; ++P) r8[mb][P] = *(const uint4*)(p.R + rr[mb] * p.ldr + col8[P]);
                    if (TAIL) r4[mb] = *(const uint2*)(p.R + rr[mb] * p.ldr + col4);
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                bool g_lane = false;             // block straddles a frame / text boundary: per-lane gate rows
                const float* grow = nullptr;
                if (EPI == 2 && p.gate) {
                    const int mf = __builtin_amdgcn_readfirstlane(mbase + mh * (16 * MB) + mb * 16), ml = min(mf + 15, p.M - 1);
                    long of = min(mf, p.M - 1), ol = ml;
                    if (p.c_rows > 0) {
                        of = (long)(of / p.c_rows) * p.c_bstride + p.c_off + of % p.c_rows;
                        ol = (long)(ml / p.c_rows) * p.c_bstride + p.c_off + ml % p.c_rows;
                    }
                    const int bf_ = (int)(of / p.seq), bl_ = (int)(ol / p.seq);
                    const int gf_ = orv_group_of((int)(of % p.seq), p.n_text, p.per_group);
                    const int gl_ = orv_group_of((int)(ol % p.seq), p.n_text, p.per_group);
                    if (bf_ == bl_ && gf_ == gl_) {
                        const float* gr = p.gate + bf_ * p.gate_b + gf_ * p.gate_g;
                        if (gr != g_cached) {
                            g_cached = gr;
#pragma unroll
                            for (int P = 0; P < NP; ++P) {
                                const float4 a = *(const float4*)(gr + col8[P]), b = *(const float4*)(gr + col8[P] + 4);
                                g8[P][0] = a.x; g8[P][1] = a.y; g8[P][2] = a.z; g8[P][3] = a.w;
                                g8[P][4] = b.x; g8[P][5] = b.y; g8[P][6] = b.z; g8[P][7] = b.w;
                            }
                            if (TAIL) { const float4 a = *(const float4*)(gr + col4); g4[0] = a.x; g4[1] = a.y; g4[2] = a.z; g4[3] = a.w; }
                        }
                    } else {
                        g_lane = true;
                        const int bidx = (int)(orow[mb] / p.seq), s = (int)(orow[mb] % p.seq);
                        grow = p.gate + bidx * p.gate_b + orv_group_of(s, p.n_text, p.per_group) * p.gate_g;
                    }
                }
                bf16_t* crow = p.C + orow[mb] * p.ldc;
                bf16_t* yrow = p.Y ? p.Y + orow[mb] * p.ldy : nullptr;
#pragma unroll
                for (int P = 0; P < NP; ++P) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[mh][mb][2 * P + (e >> 2)][e & 3] + b8[P][e];
                    if (yrow && valid[mb] && st_ok) *(uint4*)(yrow + col8[P]) = pack8(v);
                    if (EPI == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
                    }
                    if (EPI == 2) {
                        float rv[8], gg[8];
                        unpack8(r8[mb][P], rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) gg[e] = g8[P][e];
                        if (g_lane) {
                            const float4 a = *(const float4*)(grow + col8[P]), b = *(const float4*)(grow + col8[P] + 4);
                            gg[0] = a.x; gg[1] = a.y; gg[2] = a.z; gg[3] = a.w; gg[4] = b.x; gg[5] = b.y; gg[6] = b.z; gg[7] = b.w;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = rv[e] + gg[e] * v[e];
                    }
                    if (EPI == 3) {
                        float rv[8];
                        unpack8(r8[mb][P], rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= gelu_tanh_grad(rv[e]);
                    }
                    if (BN == 256 && EPI == 1 && p.c_packed) {
                        // packed P16 output (orv_gemm_t.c_packed: the hidden state of FeedForward, cogvideox_control.py:439 -> :440, as the A
                        // operand of gemm_d8): lane (r16, g) holds columns 32 P + 8 g + (0..7) of row r16 = slot 16 g + r16 of packed block
                        // (row block, column block) - one lane-linear, contiguous 1-KiB store; the buffer has tiles_m * 256 row slots (no mask)
                        if (st_ok) *(uint4*)((char*)p.C + ((((long)((mbase + mh * (16 * MB) + mb * 16) >> 4)) * (p.ldc >> 5) + ((nbase + 32 * P) >> 5)) << 10) + lane * 16) = pack8(v);
                    } else
                    if (valid[mb] && st_ok) *(uint4*)(crow + col8[P]) = pack8(v);
                }
                if (TAIL) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mh][mb][BN / 64 - 1][e] + b4[e];
                    if (yrow && valid[mb] && st_ok) *(uint2*)(yrow + col4) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    if (EPI == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
                    }
                    if (EPI == 2 || EPI == 3) {
                        const float rv[4] = {bf2f(r4[mb].x & 0xffff), bf2f(r4[mb].x >> 16), bf2f(r4[mb].y & 0xffff), bf2f(r4[mb].y >> 16)};
                        if (EPI == 2) {
                            float gg[4] = {g4[0], g4[1], g4[2], g4[3]};
                            if (g_lane) { const float4 a = *(const float4*)(grow + col4); gg[0] = a.x; gg[1] = a.y; gg[2] = a.z; gg[3] = a.w; }
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = rv[e] + gg[e] * v[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= gelu_tanh_grad(rv[e]);
                        }
                    }
                    if (valid[mb] && st_ok) *(uint2*)(crow + col4) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: epilogue through a per-wave LDS transpose (the shipped one; -DORV_T8_EPI_DIRECT keeps the register-layout epilogue above).
// Why: in the accumulator layout lane (r = lane & 15, g = lane >> 4) owns 16-byte pieces of row r, so ONE store / residual-load
// instruction touches 16 different rows and the four lanes that share a row's 64-byte segment are 16 lanes apart - the address
// coalescer sees 64 separate 16-byte requests.  Measured (profiles/r4_gemm_epilogue_ablation.txt): the stores alone cost 10-12 % of
// every K = 1920 GEMM (~6 us per 128-KB tile and CU = 11 B/clk/CU, the same with half the CUs active: a per-CU request rate, not a
// memory burst), the residual loads of the gated epilogue about as much again.  Here every 16-row block of the wave goes through a
// 4-KiB fp32 scratch image [16 rows][64 columns] in LDS (wave-private: no barrier) and comes back with lane l = (row l >> 3,
// 8-column chunk l & 7): 8 adjacent lanes hold one row's 128 contiguous bytes (BN = 256; 96 of them at BN = 192, lanes with chunk >= 6
// idle), a store / load instruction is 8 rows x one full line.  Image: 16-byte chunk L of row r sits at chunk L ^ (r & 7)
// (conflict-free for the 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups).
// The arithmetic is the register epilogue's, element for element (same fp32 expressions), so results are bit-identical to it.
__device__ __forceinline__ float sum8(float v) {             // sum over the 8 lanes lane & ~7 .. lane | 7
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    return v;
}

template <int BN, int EPI, int MB = 4>
__device__ __forceinline__ void t8_epilogue_lds(const GemmArgs& p, f32x4 (&acc)[2][MB][BN / 64], const int mbase, const int nbase, const int lane,
                                                char* const scr) {
    constexpr int NBW = BN / 64;                     // 16-column accumulator blocks per wave
    constexpr int NC = BN / 32;                      // 8-column chunks per row of the wave (8 or 6)
    static_assert(BN == 256 || EPI != 4, "epilogue 4 needs whole heads per wave");
    const int g = lane >> 4, r16 = lane & 15;        // accumulator side
    const int c = lane & 7, rr8 = lane >> 3;         // transposed side: row 8 j + rr8 of the block, columns 8 c .. 8 c + 7 of the wave
    const bool lane_on = c < NC;
    const int col8 = nbase + 8 * min(c, NC - 1);
#ifdef ORV_T8_ABL_NOSTORE
    const bool st_ok = p.M < 0;
#else
    constexpr bool st_ok = true;
#endif
    // scratch offsets
    int woff[NBW];
#pragma unroll
    for (int blk = 0; blk < NBW; ++blk) {
        const int L = BN == 256 ? 8 * (blk >> 1) + 2 * g + (blk & 1) : (blk < 2 ? 2 * g + blk : 8 + g);
        woff[blk] = r16 * 256 + ((L ^ (r16 & 7)) << 4);
    }
    const int roff0 = rr8 * 256 + (((2 * c) ^ rr8) << 4), roff1 = rr8 * 256 + (((2 * c + 1) ^ rr8) << 4);      // + j * 2048

    float b8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[e] = 0.f;
    if (p.bias) unpack8(*(const uint4*)(p.bias + col8), b8);

    // epilogue 4: qk LayerNorm parameters of this lane's 8 channels of the head
    float ga[8], be[8];
    int region = 2;
    float post = 1.f;
    if constexpr (EPI == 4) {
        region = __builtin_amdgcn_readfirstlane(nbase / (p.qn_heads * 64));     // 0 = q, 1 = k, 2 = v
        const bf16_t* gam = region == 0 ? p.qn_gq : p.qn_gk;
        const bf16_t* bet = region == 0 ? p.qn_bq : p.qn_bk;
        post = region == 0 ? p.qn_premul : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ga[e] = 1.f; be[e] = 0.f; }
        if (region < 2 && gam) unpack8(*(const uint4*)(gam + 8 * c), ga);
        if (region < 2 && bet) unpack8(*(const uint4*)(bet + 8 * c), be);
    }
    // gate row cache (EPI 2): the fp32 gate values of this lane's 8 columns, reloaded only when the block's gate row changes
    float g8[8];
    const float* g_cached = nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) g8[e] = 1.f;

    // Row operands (residual / GELU-adjoint input): a rolling prefetch four 16-row blocks deep - blocks 0-3 are requested here, block
    // b + 4 right before block b is processed, into the registers block b - 4... (b & 3) held - so the second half's loads fly while
    // the first half is worked on: one exposed load latency per tile instead of one per half, no extra registers.
    uint4 r8[MB][2];
    auto load_rows = [&](int blk8, uint4 (&dst)[2]) {       // blk8 = mh * MB + mb
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mc = min(mbase + blk8 * 16 + 8 * j + rr8, p.M - 1);
            long rr = mc;
            if (p.r_mod > 0) rr = mc % p.r_mod;
            else if (p.c_rows > 0) rr = (long)(mc / p.c_rows) * p.c_bstride + p.c_off + mc % p.c_rows;
            dst[j] = *(const uint4*)(p.R + rr * p.ldr + col8);
        }
    };
    if (EPI == 2 || EPI == 3) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) load_rows(mb, r8[mb]);
    }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        long orow[MB][2];
        bool valid[MB][2];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = mbase + mh * (16 * MB) + mb * 16 + 8 * j + rr8;
                valid[mb][j] = m < p.M && lane_on;
                const int mc = min(m, p.M - 1);
                orow[mb][j] = mc;
                if (p.c_rows > 0) orow[mb][j] = (long)(mc / p.c_rows) * p.c_bstride + p.c_off + mc % p.c_rows;
            }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            // accumulators -> scratch (fp32, natural column order), back as rows
#pragma unroll
            for (int blk = 0; blk < NBW; ++blk) *(f32x4*)(scr + woff[blk]) = acc[mh][mb][blk];
            float v[2][8];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 lo = *(const f32x4*)(scr + j * 2048 + roff0), hi = *(const f32x4*)(scr + j * 2048 + roff1);
                v[j][0] = lo[0]; v[j][1] = lo[1]; v[j][2] = lo[2]; v[j][3] = lo[3];
                v[j][4] = hi[0]; v[j][5] = hi[1]; v[j][6] = hi[2]; v[j][7] = hi[3];
            }
            // gate row of this 16-row block (EPI 2): wave-uniform when the block lies inside one (batch element, token group)
            bool g_lane = false;
            if (EPI == 2 && p.gate) {
                const int mf = __builtin_amdgcn_readfirstlane(mbase + mh * (16 * MB) + mb * 16), ml = min(mf + 15, p.M - 1);
                long of = min(mf, p.M - 1), ol = ml;
                if (p.c_rows > 0) {
                    of = (long)(of / p.c_rows) * p.c_bstride + p.c_off + of % p.c_rows;
                    ol = (long)(ml / p.c_rows) * p.c_bstride + p.c_off + ml % p.c_rows;
                }
                const int bf_ = (int)(of / p.seq), bl_ = (int)(ol / p.seq);
                const int gf_ = orv_group_of((int)(of % p.seq), p.n_text, p.per_group);
                const int gl_ = orv_group_of((int)(ol % p.seq), p.n_text, p.per_group);
                if (bf_ == bl_ && gf_ == gl_) {
                    const float* gr = p.gate + bf_ * p.gate_b + gf_ * p.gate_g;
                    if (gr != g_cached) {
                        g_cached = gr;
                        const float4 a = *(const float4*)(gr + col8), b = *(const float4*)(gr + col8 + 4);
                        g8[0] = a.x; g8[1] = a.y; g8[2] = a.z; g8[3] = a.w; g8[4] = b.x; g8[5] = b.y; g8[6] = b.z; g8[7] = b.w;
                    }
                } else g_lane = true;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float (&w)[8] = v[j];
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] += b8[e];
                bf16_t* crow = p.C + orow[mb][j] * p.ldc + col8;
                if (p.Y && valid[mb][j] && st_ok) *(uint4*)(p.Y + orow[mb][j] * p.ldy + col8) = pack8(w);
                if constexpr (EPI == 4) {
                    if (region < 2) {
                        float s = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) s += w[e];
                        const float mean = sum8(s) * (1.f / 64.f);
                        float sq = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { w[e] -= mean; sq = qkln_sq(w[e], sq); }
                        const float rstd = rsqrtf(sum8(sq) * (1.f / 64.f) + p.qn_eps);
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] = qkln_affine(w[e], rstd, ga[e], be[e], post);
                    }
                }
                if (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = gelu_tanh(w[e]);
                }
                if (EPI == 2) {
                    float rv[8], gg[8];
                    unpack8(r8[mb][j], rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gg[e] = g8[e];
                    if (g_lane) {            // block straddles a frame / text boundary: this lane's row picks its own gate row
                        const int bidx = (int)(orow[mb][j] / p.seq), sq_ = (int)(orow[mb][j] % p.seq);
                        const float* grow = p.gate + bidx * p.gate_b + orv_group_of(sq_, p.n_text, p.per_group) * p.gate_g;
                        const float4 a = *(const float4*)(grow + col8), b = *(const float4*)(grow + col8 + 4);
                        gg[0] = a.x; gg[1] = a.y; gg[2] = a.z; gg[3] = a.w; gg[4] = b.x; gg[5] = b.y; gg[6] = b.z; gg[7] = b.w;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = rv[e] + gg[e] * w[e];
                }
                if (EPI == 3) {
                    float rv[8];
                    unpack8(r8[mb][j], rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] *= gelu_tanh_grad(rv[e]);
                }
                if (valid[mb][j] && st_ok) *(uint4*)crow = pack8(w);
            }
            if ((EPI == 2 || EPI == 3) && mh == 0) load_rows(MB + mb, r8[mb]);       // this block's registers take block + 4
        }
    }
}

#ifdef ORV_T8_TRPROBE
// timing probe (WRONG results; tools/t8_tr_probe.sh, profiles/r4_gemm_tn_probe.txt): what would a TN main loop (wgrad without the operand
// transposes) cost?  Every ds_read_b128 of a fragment becomes two ds_read_b64_tr_b16 at the conflict-free addresses a row-major [8 m][64 n]
// piece image would use, and the LDS-DMA is issued from inline asm (hipcc puts s_waitcnt vmcnt(0) in front of a transposing read that
// follows a DMA builtin).
typedef short t8_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char t8_lds_char;
__device__ __forceinline__ void t8_glds_asm(const char* sbase, unsigned voff, const void* lds_dst) {
    unsigned keep;
    const unsigned long long u = (unsigned long long)(uintptr_t)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const unsigned long long su = ((unsigned long long)hi_ << 32) | lo;
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(su), "s"(d) : "memory");
}
__device__ __forceinline__ bf16x8 t8_tr2(t8_lds_char* a) {
    const t8_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) t8_v4s*)(a));
    const t8_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) t8_v4s*)(a + 512));
    union { bf16x8 v; t8_v4s h[2]; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}
#endif
// MB = 16-row blocks per m half of a wave: 4 = the 256-row tile (gemm_t8_kernel), 3 = a 192-row tile (gemm_t8r192_kernel, round 5: M = 3226, one
// clip, is 17 row tiles of 192 - 510 / 255 tiles for N = 7680 / 3840 instead of 390 / 195 tiles of 256 rows on 256 CUs - and M = 6452 is 34).
// Same streams, phases, LDS regions (an A half keeps its 16-KiB region and uses 12) and epilogues; a wave owns 96 x BN / 4, the A
// half-tiles are 96 rows = 12 pieces, so waves 6 and 7 issue no A pieces (a_on) and wait for their W pieces only (T8_VMCNT).
template <int BN, int EPI>
__global__ __launch_bounds__(512) void gemm_t8_kernel(const GemmArgs p) {
    constexpr int MB = 4;
#include "gemm_t8_body.inc"
}
template <int BN, int EPI>
__global__ __launch_bounds__(512) void gemm_t8r192_kernel(const GemmArgs p) {
    constexpr int MB = 3;
#include "gemm_t8_body.inc"
}

// ---------------------------------------------------------------------------------------------------------------
// EXPERIMENT (round 4, ORV_GEMM_TILE=4,256,256 only - never chosen by the cost model): the 256 x 256 tile on FOUR waves, one per SIMD,
// 512 registers each: a wave owns 128 x 128 (64 accumulator quads = 256 registers), so a K step of 32 needs 16 fragment reads per
// wave instead of the 24 of two 128 x 64 waves (a third less LDS traffic), and ONE barrier per K step replaces the eight per K-tile of the
// 8-wave ping-pong.  Nothing overlaps across waves here: a wave's own stream interleaves the MFMAs of step s with the fragment reads of step
// s + 1 (second register set) and the LDS-DMA of step s + 4 into the region step s just left (four 32-KiB step regions: three steps = 1.5 K-tiles
// of prefetch distance).
// The 128 columns of a wave are two of gemm_t8_kernel's 64-column wave tiles side by side (same W-row permutation per 64 columns), so the
// epilogues are t8's, called once per half.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void t4_glds_asm(const char* sbase, unsigned voff, const void* lds_dst) {
    unsigned keep;
    const unsigned long long u = (unsigned long long)(uintptr_t)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const unsigned long long su = ((unsigned long long)hi_ << 32) | lo;
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(su), "s"(d) : "memory");
}
// the accumulators are pinned to AGPRs, in place ("+a"): left to the register allocator the 64 quads were spread over both files with
// v_accvgpr_write copies in front of most MFMAs and the DMA offsets spilled (382 TF)
__device__ __forceinline__ void t4_mfma(f32x4& acc, const bf16x8& b, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}
template <int EPI>
__global__ __launch_bounds__(256) void gemm_t4_kernel(const GemmArgs p) {
    constexpr int BN = 256, REG = 32768, NREG = 4, SCR = NREG * REG;      // step region: A 16 blocks | B 16 blocks of 1 KiB (st_16x32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int ns = p.K / 32;                                              // K steps per tile (K % 128 == 0)

    // ---- DMA: 32 pieces per step; waves 0, 1 move the A blocks 8 w .. 8 w + 7, waves 2, 3 the B blocks ----
    const int lsw = lane ^ ((lane >> 5) << 1);
    const int srow = lsw >> 2, schunk = lsw & 3;
    const bool dmaA = wave < 2;                                           // wave-uniform
    unsigned long long gb_;
    {
        const unsigned long long u = (unsigned long long)(uintptr_t)(dmaA ? (const void*)p.A : (const void*)p.W);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        gb_ = ((unsigned long long)hi_ << 32) | lo;
    }
    const long ldg = dmaA ? p.lda : p.ldw;
    unsigned off0, off1, off2, off3, off4, off5, off6, off7;              // per-lane byte offsets of this wave's eight pieces (named: an array went to scratch)
    int s_dma = 0, t_dma = blockIdx.x;                                    // cursor of the DMA stream: (tile, step)
#define T4_ROW(J, TM, TN)                                                                                             \
    (dmaA ? min((TM) * 256 + (wave * 8 + (J)) * 16 + srow, p.M - 1)                                                   \
          : (TN) * BN + (wave - 2) * 128 + (((J) >> 2) & 1) * 64 +                                                    \
                (((J) >> 1) & 1) * 32 + 8 * (srow >> 2) + 4 * ((J) & 1) + (srow & 3))
#define T4_OFF(J, TM, TN) (unsigned)(((long)T4_ROW(J, TM, TN) * ldg + schunk * 8) * 2)
#define T4_SETUP(TILE)                                                                                                \
    {                                                                                                                 \
        int tm_, tn_;                                                                                                 \
        tile_of_index(p, min((TILE), ntiles - 1), ntiles, tm_, tn_);                                                  \
        off0 = T4_OFF(0, tm_, tn_); off1 = T4_OFF(1, tm_, tn_); off2 = T4_OFF(2, tm_, tn_); off3 = T4_OFF(3, tm_, tn_); \
        off4 = T4_OFF(4, tm_, tn_); off5 = T4_OFF(5, tm_, tn_); off6 = T4_OFF(6, tm_, tn_); off7 = T4_OFF(7, tm_, tn_); \
    }
#define T4_ADVANCE() if (__builtin_expect(++s_dma == ns, 0)) { s_dma = 0; t_dma += gridDim.x; T4_SETUP(t_dma) }
    T4_SETUP(t_dma)
    const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + wave * 8192;     // LDS byte address; + region * REG + j * 1024
#ifdef ORV_T4_NODMA     // ablation (wrong results)
#define T4_DMA1(OFF, LDS) { asm volatile("" :: "v"(OFF), "s"(sb_), "s"(LDS)); }
#else
#define T4_DMA1(OFF, LDS)                                                                                             \
    {                                                                                                                 \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                 \
                     :: "v"(OFF), "s"(sb_), "s"(LDS) : "memory", "m0");                                                \
    }
#endif
#define T4_ISSUE_ALL(REGION)                                                                                          \
    {                                                                                                                 \
        const unsigned long long sb_ = gb_ + (unsigned long long)s_dma * 64;                                          \
        T4_DMA1(off0, dst + (REGION) * REG) T4_DMA1(off1, dst + (REGION) * REG + 1024) T4_DMA1(off2, dst + (REGION) * REG + 2048) \
        T4_DMA1(off3, dst + (REGION) * REG + 3072) T4_DMA1(off4, dst + (REGION) * REG + 4096) T4_DMA1(off5, dst + (REGION) * REG + 5120) \
        T4_DMA1(off6, dst + (REGION) * REG + 6144) T4_DMA1(off7, dst + (REGION) * REG + 7168)                          \
        T4_ADVANCE()                                                                                                  \
    }

    // ---- fragments ----
    const int fro = ((lane & 15) * 64 + (lane >> 4) * 16) ^ (((lane >> 3) & 1) << 5);
    const char* const rdA = smem + (wr * 8) * 1024 + fro;                 // + region * REG + (mh * 4 + mb) * 1024
    const char* const rdB = smem + 16384 + (wc * 8) * 1024 + fro;         // + region * REG + (nh * 4 + blk) * 1024
    f32x4 acc[2][2][4][4];                                                // [nh][mh][mb][blk]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][8], fb[2][8];                                            // [set][block]
    auto read_set = [&](int set, int region) {
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[set][i] = *(const bf16x8*)(rdA + region * REG + i * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i) fb[set][i] = *(const bf16x8*)(rdB + region * REG + i * 1024);
    };
#define T4_FENCE() __builtin_amdgcn_sched_barrier(0);
    // one K step: [own DMA of step s + 1 landed (steps s + 2, s + 3 stay in flight) | barrier] then eight groups of { one DMA piece of
    // step s + 4 -> REGION = s % 4 ; two fragment reads of step s + 1 ; eight MFMAs of step s }
#ifdef ORV_T4_NOREAD
#define T4_RD(DST, SRC) asm volatile("" : "+v"(DST));
#else
#define T4_RD(DST, SRC) DST = *(const bf16x8*)(SRC);
#endif
#define T4_MF(SET, G, I) t4_mfma(acc[(G) >> 2][(I) >> 2][(I) & 3][(G) & 3], fb[SET][G], fa[SET][I]);
#ifdef ORV_T4_V1
#define T4_GROUP(SET, REGION, G, OFF)                                                                                 \
    {                                                                                                                 \
        T4_DMA1(OFF, dst + (REGION) * REG + (G) * 1024)                                                               \
        fa[(SET) ^ 1][G] = *(const bf16x8*)(rdA + (((REGION) + 1) & 3) * REG + (G) * 1024);                            \
        fb[(SET) ^ 1][G] = *(const bf16x8*)(rdB + (((REGION) + 1) & 3) * REG + (G) * 1024);                            \
        T4_MF(SET, G, 0) T4_MF(SET, G, 1) T4_MF(SET, G, 2) T4_MF(SET, G, 3) T4_MF(SET, G, 4) T4_MF(SET, G, 5) T4_MF(SET, G, 6) T4_MF(SET, G, 7) \
        T4_FENCE()                                                                                                    \
    }
#else
    // the memory instructions sit BETWEEN MFMAs (an MFMA occupies the pipe for 16 cycles: the issue slots behind it are free)
#define T4_GROUP(SET, REGION, G, OFF)                                                                                 \
    {                                                                                                                 \
        T4_MF(SET, G, 0)                                                                                              \
        T4_DMA1(OFF, dst + (REGION) * REG + (G) * 1024)                                                               \
        T4_MF(SET, G, 1)                                                                                              \
        T4_FENCE()                                                                                                    \
        T4_RD(fa[(SET) ^ 1][G], rdA + (((REGION) + 1) & 3) * REG + (G) * 1024)                                        \
        T4_FENCE()                                                                                                    \
        T4_MF(SET, G, 2) T4_MF(SET, G, 3)                                                                             \
        T4_FENCE()                                                                                                    \
        T4_RD(fb[(SET) ^ 1][G], rdB + (((REGION) + 1) & 3) * REG + (G) * 1024)                                        \
        T4_FENCE()                                                                                                    \
        T4_MF(SET, G, 4) T4_MF(SET, G, 5) T4_MF(SET, G, 6) T4_MF(SET, G, 7)                                           \
        T4_FENCE()                                                                                                    \
    }
#endif
#define T4_STEP(SET, REGION)                                                                                          \
    {                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                                             \
        T4_FENCE() __builtin_amdgcn_s_barrier(); T4_FENCE()                                                           \
        const unsigned long long sb_ = gb_ + (unsigned long long)s_dma * 64;                                          \
        T4_GROUP(SET, REGION, 0, off0) T4_GROUP(SET, REGION, 1, off1) T4_GROUP(SET, REGION, 2, off2) T4_GROUP(SET, REGION, 3, off3) \
        T4_GROUP(SET, REGION, 4, off4) T4_GROUP(SET, REGION, 5, off5) T4_GROUP(SET, REGION, 6, off6) T4_GROUP(SET, REGION, 7, off7) \
        T4_ADVANCE()                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        T4_FENCE()                                                                                                    \
    }

    // prologue: steps 0 .. 3 of the first tile in flight (all four regions), fragments of step 0 in set 0
    T4_ISSUE_ALL(0) T4_ISSUE_ALL(1) T4_ISSUE_ALL(2) T4_ISSUE_ALL(3)
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    T4_FENCE() __builtin_amdgcn_s_barrier(); T4_FENCE()
    read_set(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T4_FENCE()
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // step s: MFMAs on the fragments of region s % 4 (read during step s - 1, in set s & 1); that region is refilled with step s + 4
        for (int s = 0; s < ns; s += 4) {
            T4_STEP(0, 0) T4_STEP(1, 1) T4_STEP(0, 2) T4_STEP(1, 3)
        }
        int tm, tn;
        tile_of_index(p, tile, ntiles, tm, tn);
        constexpr bool lds_epi = EPI != 1 && EPI != 4;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            if constexpr (lds_epi) t8_epilogue_lds<BN, EPI>(p, acc[nh], tm * 256 + wr * 128, tn * BN + wc * 128 + nh * 64, lane, smem + SCR + wave * 4096);
            else t8_epilogue<BN, EPI>(p, acc[nh], tm * 256 + wr * 128, tn * BN + wc * 128 + nh * 64, lane);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 4; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef T4_STEP
#undef T4_GROUP
#undef T4_MF
#undef T4_FENCE
#undef T4_ISSUE_ALL
#undef T4_DMA1
#undef T4_ADVANCE
#undef T4_SETUP
#undef T4_OFF
#undef T4_ROW
}

template <int EPI>
int launch_t4_one(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = 4 * 32768 + 4 * 4096;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_t4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    const int grid = min(a.tiles_m * a.tiles_n, orv_num_cus());
    hipLaunchKernelGGL((gemm_t4_kernel<EPI>), dim3(grid), dim3(256), smem, st, a);
    return orv_check_launch("orv_gemm_bf16");
}

// ---------------------------------------------------------------------------------------------------------------
// TN form (round 4): C[M, N] (+)= A[K, M]^T . W[K, N] with BOTH operands row-major over the CONTRACTION index (the weight gradient
// dW = dY^T X of training.py::_wgrad: A = dY [tokens, out], W = X [tokens, in]) - no orv_transpose_bf16 / transpose_colsum launches.
// Same 8-phase schedule, tile sizes, LDS budget and instruction counts as gemm_t8_kernel (its K-tile macros are reused); what differs:
//   * LDS image: a 1-KiB piece is 8 contraction rows x 64 columns (one 128-byte line per row and 8 lanes: full lines), region = 8 row groups
//     x 2 column halves; 32-byte segment sg of row m sits at sg ^ f(m), f(m) = bit 1 of m | (bit 3 of m) << 1, so that the eight rows one
//     half-wave of a transposing read touches fall into eight different 32-byte bank groups.
//   * fragments: two ds_read_b64_tr_b16 per (block, k half) - lane i of a 16-lane group supplies 8 bytes of row (i >> 2) and receives column i
//     of 4 consecutive rows; k half 1 = + 4 row groups, second read = + 4 rows.
//   * the DMA is issued from inline asm (behind the builtin hipcc puts s_waitcnt vmcnt(0) in front of transposing reads); K position =
//     64 rows of the wave-uniform base; K-tiles that reach past the last contraction row take a per-lane path whose rows >= K come from a
//     zero page (both operands: the products vanish, nothing is masked afterwards).
//   * columns are in natural order (a DMA chunk is 8 consecutive columns: no W-row permutation), so the epilogue stores 8-byte pieces; it is
//     plain or accumulating (C += ...), nothing else: K = 12904 rows make the epilogue irrelevant.
// Measured against the transposes + NT kernel: profiles/r4_gemm_tn.txt.
// ---------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) uint4 t8_zero_page[16];
typedef short tn_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char tn_lds_char;
__device__ __forceinline__ void tn_glds_sv(const char* sbase, unsigned voff, const void* lds_dst) {
    const unsigned long long u = (unsigned long long)(uintptr_t)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const unsigned long long su = ((unsigned long long)hi_ << 32) | lo;
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(su), "s"(d) : "memory", "m0");
}
__device__ __forceinline__ void tn_glds_v(const void* gsrc, const void* lds_dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(d) : "memory", "m0");
}
__device__ __forceinline__ bf16x8 tn_tr2(tn_lds_char* a) {
    const tn_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_v4s*)(a));
    const tn_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_v4s*)(a + 512));
    union { bf16x8 v; tn_v4s h[2]; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}
#undef T8_SETUP_A
#undef T8_SETUP_B
#undef T8_NEXT_A
#undef T8_NEXT_B
#undef T8_GLDS
#undef T8_ISSUE_A0
#undef T8_ISSUE_A1
#undef T8_ISSUE_B0
#undef T8_ISSUE_B1
#undef T8_READ_A
#undef T8_READ_B01
#undef T8_READ_B23
template <int BN, int ACC>
__global__ __launch_bounds__(512) void gemm_t8_tn_kernel(const GemmArgs p) {
    constexpr int NBW = BN / 64;
    constexpr int MB = 4;                  // the reused K-tile macros of t8_body: 256-row tiles, every wave feeds both operands
    constexpr bool a_on = true;
    constexpr int HALF = 16384;
    constexpr int BUF = BN == 256 ? 65536 : 57344;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = ((p.K + 127) / 128) * 2;              // K-tiles of 64 contraction rows, padded to an even count
    const int nkfull = p.K / 64;                         // K-tiles with all 64 rows inside the operands

    // ---- DMA side: this lane's row of a piece and its 16-byte chunk ----
    const int r8 = lane >> 3, p8 = lane & 7;
    const int mloc = wave * 8 + r8;                      // contraction row inside the K-tile
    const int fdma = ((mloc >> 1) & 1) | (((mloc >> 3) & 1) << 1);
    const int cch = (((p8 >> 1) ^ fdma) << 1) | (p8 & 1);   // logical 16-byte chunk (8 columns) of the 64-column piece held by physical chunk p8
    char* const dst2 = smem + wave * 2048;
    char* const dst1 = smem + 3 * HALF + wave * 1024;
    unsigned oA0[2], oA1[2], oB0[2], oB1[2];
    int kA0 = 0, kA1 = 0, kB0 = 0, kB1 = 0;
    int tA0 = blockIdx.x, tA1 = blockIdx.x, tB0 = blockIdx.x, tB1 = blockIdx.x;
#define T8_SETUP_A(OFF, H, TILE)                                                                                     \
    {                                                                                                                \
        int tm_, tn_;                                                                                                \
        tile_of_index(p, min((TILE), ntiles - 1), ntiles, tm_, tn_);                                                 \
        const int c0_ = tm_ * 256 + 64 * (H) + cch * 8;                                                              \
        OFF[0] = (unsigned)(((long)mloc * p.lda + min(c0_, p.M - 8)) * 2);                                           \
        OFF[1] = (unsigned)(((long)mloc * p.lda + min(c0_ + 128, p.M - 8)) * 2);                                     \
    }
#define T8_SETUP_B(OFF, H, TILE)                                                                                     \
    {                                                                                                                \
        int tm_, tn_;                                                                                                \
        tile_of_index(p, min((TILE), ntiles - 1), ntiles, tm_, tn_);                                                 \
        const int c0_ = tn_ * BN + 128 * (H) + cch * 8;                                                              \
        OFF[0] = (unsigned)(((long)mloc * p.ldw + c0_) * 2);                                                         \
        OFF[1] = (unsigned)(((long)mloc * p.ldw + c0_ + 64) * 2);                                                    \
    }
#define T8_NEXT_A(OFF, KC, TC, H)                                                                                    \
    if (__builtin_expect(++KC == nk, 0)) { KC = 0; TC += gridDim.x; T8_SETUP_A(OFF, H, TC) }
#define T8_NEXT_B(OFF, KC, TC, H)                                                                                    \
    if (__builtin_expect(++KC == nk, 0)) { KC = 0; TC += gridDim.x; T8_SETUP_B(OFF, H, TC) }
#define T8_GLDS(BASE, LD, KC, OFF, DST)                                                                              \
    {                                                                                                                \
        const char* const sb_ = (const char*)(BASE) + (long)(KC) * 128 * (LD);                                       \
        if (__builtin_expect((KC) < nkfull, 1)) tn_glds_sv(sb_, OFF, DST);                                           \
        else tn_glds_v((KC) * 64 + mloc < p.K ? (const void*)(sb_ + (OFF)) : (const void*)((const char*)t8_zero_page + p8 * 16), DST); \
    }
#define T8_ISSUE_A0(S) { T8_GLDS(p.A, p.lda, kA0, oA0[0], dst2 + (S) * BUF) T8_GLDS(p.A, p.lda, kA0, oA0[1], dst2 + (S) * BUF + 1024) T8_NEXT_A(oA0, kA0, tA0, 0) }
#define T8_ISSUE_A1(S) { T8_GLDS(p.A, p.lda, kA1, oA1[0], dst2 + (S) * BUF + HALF) T8_GLDS(p.A, p.lda, kA1, oA1[1], dst2 + (S) * BUF + HALF + 1024) T8_NEXT_A(oA1, kA1, tA1, 1) }
#define T8_ISSUE_B0(S) { T8_GLDS(p.W, p.ldw, kB0, oB0[0], dst2 + (S) * BUF + 2 * HALF) T8_GLDS(p.W, p.ldw, kB0, oB0[1], dst2 + (S) * BUF + 2 * HALF + 1024) T8_NEXT_B(oB0, kB0, tB0, 0) }
#define T8_ISSUE_B1(S)                                                                                               \
    {                                                                                                                \
        if constexpr (BN == 256) { T8_GLDS(p.W, p.ldw, kB1, oB1[0], dst2 + (S) * BUF + 3 * HALF) T8_GLDS(p.W, p.ldw, kB1, oB1[1], dst2 + (S) * BUF + 3 * HALF + 1024) } \
        else { T8_GLDS(p.W, p.ldw, kB1, oB1[0], dst1 + (S) * BUF) }                                                  \
        T8_NEXT_B(oB1, kB1, tB1, 1)                                                                                  \
    }
    T8_SETUP_A(oA0, 0, tA0)
    T8_SETUP_A(oA1, 1, tA1)
    T8_SETUP_B(oB0, 0, tB0)
    T8_SETUP_B(oB1, 1, tB1)

    // ---- fragment reads ----
    tn_lds_char* const smem3 = (tn_lds_char*)smem;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int fr = ((i16 >> 3) & 1) | ((g4 & 1) << 1);
    const int lb2 = g4 * 2048 + (i16 >> 2) * 128 + (i16 & 3) * 8, lb1 = g4 * 1024 + (i16 >> 2) * 128 + (i16 & 3) * 8;
    tn_lds_char* trA[4];
    tn_lds_char* trB[2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) trA[mb] = smem3 + lb2 + wr * 1024 + ((mb ^ fr) << 5);
#pragma unroll
    for (int t = 0; t < 2; ++t) trB[t] = smem3 + 2 * HALF + lb2 + (wc >> 1) * 1024 + (((2 * (wc & 1) + t) ^ fr) << 5);
    tn_lds_char* const trB1 = smem3 + 3 * HALF + lb1 + ((wc ^ fr) << 5);           // BN = 192: region 1 = one piece per row group
#define T8_READ_A(MH, S)                                                                                             \
    _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)                                                                 \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) fa[MH][mb][kh] = tn_tr2(trA[mb] + (S) * BUF + (MH) * HALF + kh * 8192);
#define T8_READ_B01(S)                                                                                               \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                    \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) fb[t][kh] = tn_tr2(trB[t] + (S) * BUF + kh * 8192);
#define T8_READ_B23(S)                                                                                               \
    _Pragma("unroll") for (int t = 0; t < NBW - 2; ++t)                                                              \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                             \
            fb[2 + t][kh] = BN == 256 ? tn_tr2(trB[t] + (S) * BUF + HALF + kh * 8192) : tn_tr2(trB1 + (S) * BUF + kh * 4096);

    f32x4 acc[2][4][NBW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < NBW; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4][2], fb[NBW][2];

    if constexpr (BN == 256) {
        T8_ISSUE_A0(0) T8_ISSUE_B0(0) T8_ISSUE_A1(0) T8_ISSUE_B1(0)
        T8_ISSUE_B0(1) T8_ISSUE_A0(1) T8_ISSUE_A1(1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        T8_ISSUE_A0(0) T8_ISSUE_B0(0) T8_ISSUE_A1(0) T8_ISSUE_B1(0)
        T8_ISSUE_B0(1) T8_ISSUE_A0(1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    T8_BAR()
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (wr == 1) { T8_BAR() }
        for (int kt = 0; kt < nk; kt += 2) {
            if constexpr (BN == 256) { T8_KTILE_256(0) T8_KTILE_256(1) }
            else { T8_KTILE_192(0) T8_KTILE_192(1) }
        }
        if (wr == 0) { T8_BAR() }
        int tm, tn;
        tile_of_index(p, tile, ntiles, tm, tn);
        // epilogue: lane holds C[row = .. + i16][col = .. + 4 g4 + (0..3)] of every block: 8-byte pieces
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int row = tm * 256 + wr * 128 + mh * 64 + mb * 16 + i16;
                if (row < p.M) {
#pragma unroll
                    for (int t = 0; t < NBW; ++t) {
                        const int col = tn * BN + (t < 2 ? 32 * wc + 16 * t : (BN == 256 ? 128 + 32 * wc + 16 * (t - 2) : 128 + 16 * wc)) + 4 * g4;
                        f32x4 v = acc[mh][mb][t];
                        bf16_t* cp = p.C + (long)row * p.ldc + col;
                        if constexpr (ACC) {
                            const uint2 u = *(const uint2*)cp;
                            v[0] += bf2f(u.x & 0xffff); v[1] += bf2f(u.x >> 16); v[2] += bf2f(u.y & 0xffff); v[3] += bf2f(u.y >> 16);
                        }
                        *(uint2*)cp = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    }
                }
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < NBW; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int BN, int ACC>
int launch_tn_one(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (BN == 256 ? 65536 : 57344);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_t8_tn_kernel<BN, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    const int grid = min(a.tiles_m * a.tiles_n, orv_num_cus());
    hipLaunchKernelGGL((gemm_t8_tn_kernel<BN, ACC>), dim3(grid), dim3(512), smem, st, a);
    return orv_check_launch("orv_gemm_tn_bf16");
}

template <int BN, int EPI, int BM = 256>
int launch_one(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (BN == 256 ? 65536 : 57344) + 8 * 4096;      // two K-tile buffers + the epilogue scratch (160 KiB at BN = 256)
    static bool attr_done = false;
    if (!attr_done) {
        if constexpr (BM == 256) (void)hipFuncSetAttribute((const void*)gemm_t8_kernel<BN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        else (void)hipFuncSetAttribute((const void*)gemm_t8r192_kernel<BN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    int grid = min(a.tiles_m * a.tiles_n, orv_num_cus());
    if (a.grid_cap > 0) grid = min(grid, a.grid_cap);
    if constexpr (BM == 256) hipLaunchKernelGGL((gemm_t8_kernel<BN, EPI>), dim3(grid), dim3(512), smem, st, a);
    else hipLaunchKernelGGL((gemm_t8r192_kernel<BN, EPI>), dim3(grid), dim3(512), smem, st, a);
    return orv_check_launch("orv_gemm_bf16");
}

}  // namespace

namespace orv_gemm {
int launch_t8_tn(const GemmArgs& a, int bn, int accumulate, hipStream_t st) {
    if (bn == 256) return accumulate ? launch_tn_one<256, 1>(a, st) : launch_tn_one<256, 0>(a, st);
    if (bn == 192) return accumulate ? launch_tn_one<192, 1>(a, st) : launch_tn_one<192, 0>(a, st);
    orv_set_error("orv_gemm_tn_bf16: no kernel for BN=%d", bn);
    return ORV_EINVAL;
}
int launch_t4(const GemmArgs& a, int epi, hipStream_t st) {
    switch (epi) {
        case 0: return launch_t4_one<0>(a, st);
        case 1: return launch_t4_one<1>(a, st);
        case 2: return launch_t4_one<2>(a, st);
    }
    orv_set_error("orv_gemm_bf16: no t4 kernel for epilogue %d", epi);
    return ORV_EINVAL;
}
int launch_t8(const GemmArgs& a, int bn, int epi, hipStream_t st, int bm) {
    if ((long)a.M * a.lda * 2 >= (1L << 32) || (long)a.N * a.ldw * 2 >= (1L << 32)) {
        orv_set_error("orv_gemm_bf16: the t8 kernel addresses A / W with 32-bit byte offsets (M=%d lda=%ld N=%d ldw=%ld)", a.M, a.lda, a.N, a.ldw);
        return ORV_EINVAL;
    }
    if (a.a_packed || (a.c_packed && (bn != 256 || epi != 1 || a.ldc != a.N || a.c_rows > 0 || a.Y))) {
        orv_set_error("orv_gemm_bf16: the t8 kernel reads row-major A and writes packed C only as BN = 256, epilogue 1, ldc == N, no row map / Y");
        return ORV_EINVAL;
    }
    if (bm == 192) {      // the 192-row tile: the epilogues of the inference forward (single-clip and two-clip shapes)
        if (bn == 256) {
            switch (epi) {
                case 0: return launch_one<256, 0, 192>(a, st);
                case 1: return launch_one<256, 1, 192>(a, st);
                case 2: return launch_one<256, 2, 192>(a, st);
                case 4: return launch_one<256, 4, 192>(a, st);
            }
        } else if (bn == 192) {
            switch (epi) {
                case 0: return launch_one<192, 0, 192>(a, st);
                case 1: return launch_one<192, 1, 192>(a, st);
                case 2: return launch_one<192, 2, 192>(a, st);
            }
        }
        orv_set_error("orv_gemm_bf16: no 192-row t8 kernel for BN=%d epilogue %d", bn, epi);
        return ORV_EINVAL;
    }
    if (bn == 256) {
        switch (epi) {
            case 0: return launch_one<256, 0>(a, st);
            case 1: return launch_one<256, 1>(a, st);
            case 2: return launch_one<256, 2>(a, st);
            case 3: return launch_one<256, 3>(a, st);
            case 4: return launch_one<256, 4>(a, st);
        }
    } else if (bn == 192) {
        switch (epi) {
            case 0: return launch_one<192, 0>(a, st);
            case 1: return launch_one<192, 1>(a, st);
            case 2: return launch_one<192, 2>(a, st);
            case 3: return launch_one<192, 3>(a, st);
        }
    }
    orv_set_error("orv_gemm_bf16: no t8 kernel for BN=%d epilogue %d", bn, epi);
    return ORV_EINVAL;
}
}  // namespace orv_gemm
