// bf16 GEMM  C = epilogue(A[M,K] . W[N,K]^T + bias)  for gfx950 (MI355X), hand-written MFMA kernel.
//
// Replaces every nn.Linear / Conv2d(k=s=p) on the ORV denoise path (to_q/k/v, to_out, FeedForward,
// text_proj, patch proj, proj_out, initial_combine_linear; see include/orv_mi355.h for reference lines).
//
// Structure (one workgroup = 8 waves = BM x BN output tile, BK = 64):
//   * both operands are K-contiguous, so a tile row is one 128-B line; tiles are staged HBM/L2 -> LDS with
//     global_load_lds (16 B per lane, no VGPR round trip), double buffered, ONE barrier per K-tile.
//   * LDS image is lane-linear (a glds requirement), so the bank-conflict swizzle is applied on the per-lane
//     SOURCE address and again on the ds_read address: 16-B chunk c of row r lives at slot c ^ ((r >> 1) & 7).
//     A ds_read_b128 lane group (16 rows, same chunk) then hits 16 distinct 16-B slots of the 256-B bank row.
//   * waves are arranged 4 (M) x 2 (N); each wave owns a (BM/4) x (BN/2) sub-tile as 32x32 MFMA blocks,
//     v_mfma_f32_32x32x16_bf16 with the W fragment as the A operand, so the accumulator holds C^T:
//     lane&31 = output row m, register quad = 4 consecutive output columns n  -> 8-byte epilogue stores.
//   * workgroup -> tile map is XCD-aware (block b runs on XCD b % 8): each XCD gets a contiguous chunk of the
//     tile list, ordered in GM x 8 super-tiles so the 32 CUs of an XCD share A row-panels and W column-panels in L2.
#include "gemm_common.hpp"
#include <type_traits>
#include <stdlib.h>
#include <string.h>

namespace {

using namespace orv_gemm;

// Epilogue shared by both kernels.  acc[i][j][4q+e] = C[m][n] with m = mbase + j*32 + (lane&31),
// n = nbase + i*32 + 8q + 4*(lane>>5) + e  (C^T accumulator layout: 4 consecutive columns per register quad).
// Memory side: lanes l and l+32 own the two 8-byte halves of one 16-byte column group of the SAME row, so two column
// groups (register quads 2u, 2u+1) are exchanged with v_permlane32_swap and every lane moves ONE aligned 16-byte piece
// (lower half-wave: group 2u, upper: group 2u+1): half the store/load instructions of the 8-byte form and 32
// contiguous bytes per row and instruction (measured: the 8-byte stores cost 15 % of the FFN1 GEMM).
__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b) {
    // a of the upper half-wave <-> b of the lower half-wave
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

__device__ __forceinline__ float sum_with_partner_half(float v) {   // v(lane) + v(lane ^ 32)
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// two register quads (this lane's 4 columns of column groups 2u and 2u+1) -> one aligned 16-byte piece per lane
__device__ __forceinline__ void store_quads16(bf16_t* row, int n16, const float (&a)[4], const float (&b)[4]) {
    uint32_t a0 = pack2bf(a[0], a[1]), a1 = pack2bf(a[2], a[3]), b0 = pack2bf(b[0], b[1]), b1 = pack2bf(b[2], b[3]);
    swap_halves(a0, b0);
    swap_halves(a1, b1);
    *(uint4*)(row + n16) = make_uint4(a0, a1, b0, b1);
}

// Epilogue 4: the QKV projection with the per-head LayerNorm(64) of q and k (diffusers Attention.norm_q / norm_k as called
// at cogvideox_control.py:243-247) and the softmax pre-multiplier of q applied in registers; the v third is stored as is.
// A wave's BN/2 columns are whole heads of ONE of q | k | v (host-checked), a head is two adjacent 32-column blocks, and
// a row's 64 values sit in this lane (32) and lane ^ 32 (32): the statistics are lane-local sums plus one half-wave swap.
// Optional Y keeps acc + bias (the raw projection) for the LayerNorm adjoint.
template <int NB, int MB>
__device__ __forceinline__ void gemm_epilogue_qknorm(const GemmArgs& p, f32x16 (&acc)[NB][MB], int mbase, int nbase, int lane) {
    static_assert(NB % 2 == 0, "a wave must cover whole 64-wide heads");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int l31 = lane & 31, hi = lane >> 5;
    const int region = __builtin_amdgcn_readfirstlane(nbase / (p.qn_heads * 64));   // 0 = q, 1 = k, 2 = v
    const bf16_t* gam = region == 0 ? p.qn_gq : p.qn_gk;
    const bf16_t* bet = region == 0 ? p.qn_bq : p.qn_bk;
    const float post = region == 0 ? p.qn_premul : 1.f;
    // One head (two 32-column blocks of one row block) at a time, fenced with sched_barrier: the 192 accumulator registers
    // leave no room for the scheduler to overlap heads (it did, and spilled ~700 registers).
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int m = mbase + j * 32 + l31;
        const bool valid = m < p.M;               // lane and lane ^ 32 hold the same row
        bf16_t* crow = p.C + (long)min(m, p.M - 1) * p.ldc;
        bf16_t* yrow = p.Y ? p.Y + (long)min(m, p.M - 1) * p.ldy : nullptr;
#pragma unroll
        for (int hh = 0; hh < NB / 2; ++hh) {
            __builtin_amdgcn_sched_barrier(0);
            float v[2][16];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float bb[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) {
                        const int nq = __builtin_amdgcn_readfirstlane(nbase + (2 * hh + ii) * 32 + q * 8);
                        const u32x4 b8 = *(const __attribute__((address_space(4))) u32x4*)(uintptr_t)(p.bias + nq);
                        const uint32_t bx = hi ? b8[2] : b8[0], by = hi ? b8[3] : b8[1];
                        bb[0] = bf2f(bx & 0xffff); bb[1] = bf2f(bx >> 16); bb[2] = bf2f(by & 0xffff); bb[3] = bf2f(by >> 16);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[ii][q * 4 + e] = acc[2 * hh + ii][j][q * 4 + e] + bb[e];
                }
            if (yrow) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float a[4], b[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] = v[ii][(2 * u) * 4 + e]; b[e] = v[ii][(2 * u + 1) * 4 + e]; }
                        if (valid) store_quads16(yrow, nbase + (2 * hh + ii) * 32 + (2 * u + hi) * 8, a, b);
                    }
            }
            if (region < 2) {
                float s = 0.f;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += v[ii][r];
                const float mean = sum_with_partner_half(s) * (1.f / 64.f);
                float sq = 0.f;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { v[ii][r] -= mean; sq += v[ii][r] * v[ii][r]; }
                const float rstd = rsqrtf(sum_with_partner_half(sq) * (1.f / 64.f) + p.qn_eps);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float g[4] = {1.f, 1.f, 1.f, 1.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
                        if (gam) {
                            const u32x4 g8 = *(const __attribute__((address_space(4))) u32x4*)(uintptr_t)(gam + ii * 32 + q * 8);
                            const uint32_t gx = hi ? g8[2] : g8[0], gy = hi ? g8[3] : g8[1];
                            g[0] = bf2f(gx & 0xffff); g[1] = bf2f(gx >> 16); g[2] = bf2f(gy & 0xffff); g[3] = bf2f(gy >> 16);
                        }
                        if (bet) {
                            const u32x4 b8 = *(const __attribute__((address_space(4))) u32x4*)(uintptr_t)(bet + ii * 32 + q * 8);
                            const uint32_t bx = hi ? b8[2] : b8[0], by = hi ? b8[3] : b8[1];
                            bb[0] = bf2f(bx & 0xffff); bb[1] = bf2f(bx >> 16); bb[2] = bf2f(by & 0xffff); bb[3] = bf2f(by >> 16);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[ii][q * 4 + e] = (v[ii][q * 4 + e] * rstd * g[e] + bb[e]) * post;
                    }
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float a[4], b[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = v[ii][(2 * u) * 4 + e]; b[e] = v[ii][(2 * u + 1) * 4 + e]; }
                    if (valid) store_quads16(crow, nbase + (2 * hh + ii) * 32 + (2 * u + hi) * 8, a, b);
                }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NB, int MB, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[NB][MB], int mbase, int nbase, int lane) {
    if constexpr (EPI == 4) {
        gemm_epilogue_qknorm<NB, MB>(p, acc, mbase, nbase, lane);
        return;
    }
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int m = mbase + j * 32 + l31;
        if (m >= p.M) continue;
        long orow = m;
        if (p.c_rows > 0) orow = (long)(m / p.c_rows) * p.c_bstride + p.c_off + m % p.c_rows;
        bf16_t* crow = p.C + orow * p.ldc;
        bf16_t* yrow = p.Y ? p.Y + orow * p.ldy : nullptr;
        const bf16_t* rrow = nullptr;
        const float* grow = nullptr;
        if (EPI == 2 || EPI == 3) {
            const long rr = p.r_mod > 0 ? m % p.r_mod : orow;
            rrow = p.R + rr * p.ldr;
            if (p.gate) {
                const int bidx = (int)(orow / p.seq), s = (int)(orow % p.seq);
                grow = p.gate + bidx * p.gate_b + orv_group_of(s, p.n_text, p.per_group) * p.gate_g;
            }
        }
        // Every vector load issued here queues BEHIND the DMA pieces already in flight for the next tile (vmcnt retires in
        // order), i.e. costs a full loaded-memory-pipeline latency: so the bias comes through the scalar cache (uniform
        // address, s_load), and the per-row operands (residual, gate) of a row block are all requested up front.
        constexpr int IB = NB <= 4 ? NB : 2;   // blocks whose row operands are requested together (register budget)
        bool g_uniform = false;     // all rows of this 32-row block share one gate row -> gate through the scalar cache too
        const float* g_srow = nullptr;
        if (EPI == 2 && p.gate) {
            const int mf = __builtin_amdgcn_readfirstlane(mbase + j * 32), ml = min(mf + 31, p.M - 1);
            long of = mf, ol = ml;
            if (p.c_rows > 0) {
                of = (long)(mf / p.c_rows) * p.c_bstride + p.c_off + mf % p.c_rows;
                ol = (long)(ml / p.c_rows) * p.c_bstride + p.c_off + ml % p.c_rows;
            }
            const int bf_ = (int)(of / p.seq), bl_ = (int)(ol / p.seq);
            const int gf_ = orv_group_of((int)(of % p.seq), p.n_text, p.per_group);
            const int gl_ = orv_group_of((int)(ol % p.seq), p.n_text, p.per_group);
            g_uniform = (bf_ == bl_) && (gf_ == gl_);
            g_srow = p.gate + bf_ * p.gate_b + gf_ * p.gate_g;
        }
#pragma unroll
        for (int i0 = 0; i0 < NB; i0 += IB) {
        uint32_t rq[IB][2][2][2];   // [block][u][quad t][dword]: residual / pre-activation operand, this lane's 4 columns
        if (EPI == 2 || EPI == 3) {
#pragma unroll
            for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint4 rr = *(const uint4*)(rrow + nbase + (i0 + ii) * 32 + (2 * u + hi) * 8);
                    rq[ii][u][0][0] = rr.x; rq[ii][u][0][1] = rr.y; rq[ii][u][1][0] = rr.z; rq[ii][u][1][1] = rr.w;
                }
#pragma unroll
            for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    // loaded: lower = group 2u cols 0-7, upper = group 2u+1 cols 0-7  ->  (quad 2u | quad 2u+1) own 4 columns
                    swap_halves(rq[ii][u][0][0], rq[ii][u][1][0]);
                    swap_halves(rq[ii][u][0][1], rq[ii][u][1][1]);
                }
        }
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
            const int i = i0 + ii;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // this lane's 16-byte piece: column group 2u + hi of the 32-column block
                const int n16 = nbase + i * 32 + (2 * u + hi) * 8;
                uint32_t oc[2][2], oy[2][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q = 2 * u + t;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
                    if (p.bias) {
                        const int nq = __builtin_amdgcn_readfirstlane(nbase + i * 32 + q * 8);   // wave-uniform + constant address space: s_load
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 b8 = *(const __attribute__((address_space(4))) u32x4*)(uintptr_t)(p.bias + nq);
                        const uint32_t bx = hi ? b8[2] : b8[0], by = hi ? b8[3] : b8[1];
                        v[0] += bf2f(bx & 0xffff); v[1] += bf2f(bx >> 16);
                        v[2] += bf2f(by & 0xffff); v[3] += bf2f(by >> 16);
                    }
                    if (yrow) {   // training: keep acc + bias (GELU pre-activation / un-gated branch output)
                        oy[t][0] = pack2bf(v[0], v[1]); oy[t][1] = pack2bf(v[2], v[3]);
                    }
                    if (EPI == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
                    }
                    if (EPI == 2) {
                        float g[4] = {1.f, 1.f, 1.f, 1.f};
                        if (grow) {
                            if (g_uniform) {
                                typedef float f32x4s __attribute__((ext_vector_type(4)));
                                const int nq = __builtin_amdgcn_readfirstlane(nbase + i * 32 + q * 8);
                                const f32x4s ga = *(const __attribute__((address_space(4))) f32x4s*)(uintptr_t)(g_srow + nq);
                                const f32x4s gb = *(const __attribute__((address_space(4))) f32x4s*)(uintptr_t)(g_srow + nq + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) g[e] = hi ? gb[e] : ga[e];
                            } else {   // block straddles a frame / text boundary (about 1 in 19): per-row gate rows
                                const float4 gg = *(const float4*)(grow + nbase + i * 32 + q * 8 + hi * 4);
                                g[0] = gg.x; g[1] = gg.y; g[2] = gg.z; g[3] = gg.w;
                            }
                        }
                        v[0] = bf2f(rq[ii][u][t][0] & 0xffff) + g[0] * v[0]; v[1] = bf2f(rq[ii][u][t][0] >> 16) + g[1] * v[1];
                        v[2] = bf2f(rq[ii][u][t][1] & 0xffff) + g[2] * v[2]; v[3] = bf2f(rq[ii][u][t][1] >> 16) + g[3] * v[3];
                    }
                    if (EPI == 3) {   // backward through GELU(tanh): C = acc * gelu'(U), U = saved pre-activation
                        v[0] *= gelu_tanh_grad(bf2f(rq[ii][u][t][0] & 0xffff)); v[1] *= gelu_tanh_grad(bf2f(rq[ii][u][t][0] >> 16));
                        v[2] *= gelu_tanh_grad(bf2f(rq[ii][u][t][1] & 0xffff)); v[3] *= gelu_tanh_grad(bf2f(rq[ii][u][t][1] >> 16));
                    }
                    oc[t][0] = pack2bf(v[0], v[1]); oc[t][1] = pack2bf(v[2], v[3]);
                }
                if (yrow) {
                    swap_halves(oy[0][0], oy[1][0]);
                    swap_halves(oy[0][1], oy[1][1]);
                    *(uint4*)(yrow + n16) = make_uint4(oy[0][0], oy[0][1], oy[1][0], oy[1][1]);
                }
                swap_halves(oc[0][0], oc[1][0]);   // lower: (A cols 0-3 | A cols 4-7) ; upper: (B cols 0-3 | B cols 4-7)
                swap_halves(oc[0][1], oc[1][1]);
#ifdef ORV_GEMM_ABLATE_NOSTORE
                if (p.dbg == 12345)
#endif
                *(uint4*)(crow + n16) = make_uint4(oc[0][0], oc[0][1], oc[1][0], oc[1][1]);
            }
        }
        }
    }
}

// WM = waves along M (4 or 2), 8 / WM along N.  WM = 2 exists for the 192-row tile: 3226 rows (one clip) are 16.8 tiles of
// 192, so every N = 1920 GEMM of a B = 1 step is 255 tiles - one full round of the 256 CUs.
// NS = LDS stages (K-tiles NS-1 ahead in flight).  Only NS = 2 is instantiated: three and four stages were measured on the
// 192x128 tile (profiles/r2_gemm_b1_tiles.log) - no gain on single-round launches (they are not latency bound) and a 12 % loss
// on multi-round ones, where two 80 KB workgroups per CU overlap each other's barriers and a 120+ KB one runs alone.
template <int BM, int BN, int EPI, int WM = 4, int NS = 2>
__global__ __launch_bounds__(512) void gemm_kernel(const GemmArgs p) {
    constexpr int WN = 8 / WM;
    constexpr int D = NS - 1;            // prefetch distance in K-tiles
    constexpr int MB = BM / (32 * WM);   // 32-row blocks per wave along M
    constexpr int NB = BN / (32 * WN);   // 32-col blocks per wave along N
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BM % 64 == 0 && BN % 64 == 0, "tile / wave grid mismatch");
    constexpr int A_BYTES = BM * 128;  // one stage of A: BM rows x 64 bf16
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_LD = BM / 64;      // glds instructions per wave per stage (each moves 8 rows)
    constexpr int B_LD = BN / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware tile mapping (bijective for any grid size) ----
    int tm, tn;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = b & 7, j = b >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int per = GM * p.tiles_n;
        const int gid = L / per, rem = L % per;
        const int first_m = gid * GM;
        const int gsize = min(p.tiles_m - first_m, GM);
        tm = first_m + rem % gsize;
        tn = rem / gsize;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging source pointers (swizzle on the SOURCE: lane -> row, slot; chunk = slot ^ f(row)) ----
    const bf16_t* a_src[A_LD];
    const bf16_t* b_src[B_LD];
    const int srow = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int row = (wave * A_LD + j) * 8 + srow;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int grow = min(m0 + row, p.M - 1);
        a_src[j] = p.A + (long)grow * p.lda + chunk * 8;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int row = (wave * B_LD + j) * 8 + srow;
        const int chunk = slot ^ ((row >> 1) & 7);
        b_src[j] = p.W + (long)(n0 + row) * p.ldw + chunk * 8;
    }
    // One glds "piece" = 8 tile rows (1 KiB).  Pieces of the NEXT K-tile are issued between the MFMAs of the current one
    // (an LDS-DMA issue costs ~100 cycles of the wave's issue slot; back-to-back after the barrier they idle the matrix pipe).
    constexpr int NP = A_LD + B_LD;
    auto issue_piece = [&](int pc, int s, long koff) {
        if (pc < A_LD) glds16(a_src[pc] + koff, smem + s * STAGE + (wave * A_LD + pc) * 1024);
        else glds16(b_src[pc - A_LD] + koff, smem + s * STAGE + A_BYTES + (wave * B_LD + (pc - A_LD)) * 1024);
    };

    // ---- fragment read offsets ----
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int a_row_off = (wm * (BM / WM) + l31) * 128;
    const int b_row_off = A_BYTES + (wn * (BN / WN) + l31) * 128;

    f32x16 acc[NB][MB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto read_frags = [&](const char* sbase, int ks, bf16x8 (&af)[MB], bf16x8 (&bf)[NB]) {
        const int coff = ((ks * 2 + hi) ^ sw) * 16;
#pragma unroll
        for (int j = 0; j < MB; ++j) af[j] = *(const bf16x8*)(sbase + a_row_off + j * 32 * 128 + coff);
#pragma unroll
        for (int i = 0; i < NB; ++i) bf[i] = *(const bf16x8*)(sbase + b_row_off + i * 32 * 128 + coff);
    };

    const int nk = p.K / BK;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) issue_piece(pc, d, (long)(d < nk ? d : 0) * BK);
    int cs = 0, ns = D;                  // stage read this iteration / stage filled this iteration (tile t + D)
    for (int t = 0; t < nk; ++t) {
        // tile t has landed for this wave (D - 1 younger tiles may still be in flight); after the barrier it has landed for
        // all waves and nobody still reads stage ns (those reads were consumed by the MFMAs of iteration t-1).
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NP) : "memory");
        __syncthreads();
        const char* sbase = smem + cs * STAGE;
        const long koff = (long)((t + D < nk) ? (t + D) : 0) * BK;   // the last iterations re-fetch tile 0 (never read)
        bf16x8 af[2][MB], bf[2][NB];
        read_frags(sbase, 0, af[0], bf[0]);
        auto kstep = [&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            constexpr int NM = NB * MB;
            constexpr int NPC = ((ks + 1) * NP + 2) / 3 - (ks * NP + 2) / 3;   // pieces pc with pc*3/NP == ks
            if constexpr (ks < 3) read_frags(sbase, ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
                if (pc * 3 / NP == ks) issue_piece(pc, ns, koff);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < MB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][i], af[ks & 1][j], acc[i][j], 0, 0, 0);
            // schedule: next k-step's LDS reads first, then MFMAs with the DMA issues spread between them
            if constexpr (ks < 3) __builtin_amdgcn_sched_group_barrier(0x100, MB + NB, 0);
            constexpr int PER = NPC > 0 ? (NM / (NPC + 1) > 0 ? NM / (NPC + 1) : 1) : NM;
            if constexpr (NPC >= 1) { __builtin_amdgcn_sched_group_barrier(0x8, PER, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
            if constexpr (NPC >= 2) { __builtin_amdgcn_sched_group_barrier(0x8, PER, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
            if constexpr (NPC >= 3) { __builtin_amdgcn_sched_group_barrier(0x8, PER, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
            if constexpr (NPC >= 4) { __builtin_amdgcn_sched_group_barrier(0x8, PER, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
            constexpr int LEFT = NM - (NPC > 4 ? 4 : NPC) * PER;
            if constexpr (LEFT > 0) __builtin_amdgcn_sched_group_barrier(0x8, LEFT, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{});
        cs = cs + 1 == NS ? 0 : cs + 1;
        ns = ns + 1 == NS ? 0 : ns + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    gemm_epilogue<NB, MB, EPI>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (VAE): gemm_kernel with the A operand GATHERED by the LDS-DMA itself from a
// channels-last activation [B, Ts, Hs, Ws, C] - no patch matrix in HBM.  Output row m = voxel (b, t, y, x); K index = tap * C +
// channel, so one 64-deep K-tile is 64 channels of ONE tap: a 128-byte line per voxel, and the DMA piece of a lane is the line
// of the tap-shifted source voxel (or of a 256-byte zero page when the tap falls into the zero padding).  Folded into the source
// index exactly like orv_vae_im2col: replicate-first-frame / conv_cache temporal context, zero spatial padding, stride 2,
// nearest x2 upsampling of the source in H, W and T.  Requires C % 64 == 0 (the 16- and 8-channel stems keep the patch matrix).
// ---------------------------------------------------------------------------------------------------------------
struct ConvArgs {
    const bf16_t* src;
    int B, Ts, Hs, Ws, C, T, H, W, kt, kh, kw, stride, pad_lo, ups_s, ups_t, t_shift;
};
__device__ __attribute__((aligned(256))) uint4 orv_zero_page[16];       // 256 zero bytes: source of padded taps

// SIMPLE = stride 1, no upsampling (every resnet convolution): the conv-input grid IS the source grid, so a tap is a 32-bit
// voxel-index offset and the per-piece address arithmetic is ~10 VALU instructions instead of ~30 (at N = 128 the loop has 16
// MFMAs per K-tile and wave: the general form's 64-bit index math was as long as the MFMAs)
template <int BM, int BN, int EPI, bool SIMPLE>
__global__ __launch_bounds__(512) void conv_gemm_kernel(const GemmArgs p, const ConvArgs cv) {
    constexpr int MB = BM / 128, NB = BN / 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_LD = BM / 64, B_LD = BN / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_of_block(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // per DMA piece (8 rows): this lane's output voxel and its 16-byte slot of the 128-byte channel line
    const int srow = lane >> 3, slot = lane & 7;
    int vb[A_LD], vt[A_LD], vy[A_LD], vx[A_LD], vchunk[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int row = (wave * A_LD + j) * 8 + srow;
        vchunk[j] = (slot ^ ((row >> 1) & 7)) * 8;
        long m = min((long)m0 + row, (long)p.M - 1);
        vx[j] = (int)(m % cv.W); m /= cv.W;
        vy[j] = (int)(m % cv.H); m /= cv.H;
        vt[j] = (int)(m % cv.T);
        vb[j] = (int)(m / cv.T);
        if (SIMPLE) vb[j] *= cv.Ts;                                  // first source frame of this batch element
    }
    const bf16_t* b_src[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int row = (wave * B_LD + j) * 8 + srow;
        b_src[j] = p.W + (long)(n0 + row) * p.ldw + (slot ^ ((row >> 1) & 7)) * 8;
    }
    const int cblocks = cv.C >> 6;                                    // K-tiles per tap
    const int Hi = cv.ups_s ? cv.Hs * 2 : cv.Hs, Wi = cv.ups_s ? cv.Ws * 2 : cv.Ws;
    constexpr int NP = A_LD + B_LD;
    auto issue_piece = [&](int pc, int s, int ktile) {
        if (pc < A_LD && SIMPLE) {
            const int tap = ktile / cblocks, c0 = (ktile - tap * cblocks) << 6;          // wave-uniform (SALU)
            const int dx = tap % cv.kw - cv.pad_lo, dy = (tap / cv.kw) % cv.kh - cv.pad_lo;
            const int dt = tap / (cv.kw * cv.kh) + cv.t_shift - (cv.kt - 1);
            const int ti = max(vt[pc] + dt, 0), yi = vy[pc] + dy, xi = vx[pc] + dx;
            const bool ok = (unsigned)yi < (unsigned)cv.Hs && (unsigned)xi < (unsigned)cv.Ws;
            const unsigned idx = (unsigned)((vb[pc] + ti) * cv.Hs + yi) * (unsigned)cv.Ws + (unsigned)xi;   // voxel index < 2^31 (host-checked)
            const bf16_t* g = cv.src + (long)idx * cv.C + (c0 + vchunk[pc]);
            if (!ok) g = (const bf16_t*)orv_zero_page + (lane & 7) * 8;
            glds16(g, smem + s * STAGE + (wave * A_LD + pc) * 1024);
        } else if (pc < A_LD) {
            const int tap = ktile / cblocks, c0 = (ktile - tap * cblocks) << 6;          // wave-uniform
            const int dx = tap % cv.kw, dy = (tap / cv.kw) % cv.kh, dt = tap / (cv.kw * cv.kh);
            int ti = vt[pc] + cv.t_shift - (cv.kt - 1) + dt;
            ti = max(ti, 0);
            const int yi = vy[pc] * cv.stride - cv.pad_lo + dy, xi = vx[pc] * cv.stride - cv.pad_lo + dx;
            const bool ok = yi >= 0 && yi < Hi && xi >= 0 && xi < Wi;
            const int ys = cv.ups_s ? yi >> 1 : yi, xs = cv.ups_s ? xi >> 1 : xi;
            int ts = ti;
            if (cv.ups_t == 1) ts = ti >> 1;
            else if (cv.ups_t == 2) ts = ti == 0 ? 0 : 1 + ((ti - 1) >> 1);
            const bf16_t* g = cv.src + ((((long)vb[pc] * cv.Ts + ts) * cv.Hs + ys) * cv.Ws + xs) * cv.C + c0 + vchunk[pc];
            if (!ok) g = (const bf16_t*)orv_zero_page + (lane & 7) * 8;
            glds16(g, smem + s * STAGE + (wave * A_LD + pc) * 1024);
        } else {
            glds16(b_src[pc - A_LD] + (long)ktile * BK, smem + s * STAGE + A_BYTES + (wave * B_LD + (pc - A_LD)) * 1024);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int a_row_off = (wm * (BM / 4) + l31) * 128;
    const int b_row_off = A_BYTES + (wn * (BN / 2) + l31) * 128;
    f32x16 acc[NB][MB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto read_frags = [&](const char* sbase, int ks, bf16x8 (&af)[MB], bf16x8 (&bf)[NB]) {
        const int coff = ((ks * 2 + hi) ^ sw) * 16;
#pragma unroll
        for (int j = 0; j < MB; ++j) af[j] = *(const bf16x8*)(sbase + a_row_off + j * 32 * 128 + coff);
#pragma unroll
        for (int i = 0; i < NB; ++i) bf[i] = *(const bf16x8*)(sbase + b_row_off + i * 32 * 128 + coff);
    };

    const int nk = p.K / BK;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) issue_piece(pc, 0, 0);
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* sbase = smem + (t & 1) * STAGE;
        const int ns = (t + 1) & 1;
        const int knext = (t + 1 < nk) ? (t + 1) : 0;               // last iteration re-fetches tile 0 (never read)
        // k-step ks: fragments of ks + 1 are read first, this k-step's share of the next tile's DMA pieces (address arithmetic +
        // issue) sits between its MFMAs: the LDS-DMA issue cost (~100 cycles each) hides under the matrix pipe instead of
        // stacking up behind the barrier
        bf16x8 af[2][MB], bf[2][NB];
        read_frags(sbase, 0, af[0], bf[0]);
        auto kstep = [&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            if constexpr (ks < 3) read_frags(sbase, ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
                if (pc * 4 / NP == ks) issue_piece(pc, ns, knext);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < MB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][i], af[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gemm_epilogue<NB, MB, EPI>(p, acc, m0 + wm * (BM / 4), n0 + wn * (BN / 2), lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Strip form of the stride-1 3x3(x3) convolution: the three taps dx = -1, 0, +1 of one (dt, dy, 64-channel block) read the SAME
// 256 + 2 voxel lines, so they are staged once as a strip (row r = flattened voxel m0 - 1 + r) and the three K-tiles read it at
// row offsets 0, 1, 2.  The LDS-DMA moves 33 KB of activations + 3 W tiles per three K-tiles instead of 3 x 32 KB + 3 W tiles: at
// N = 128 (the full-resolution stage of the decoder, DMA-rate bound: 48 KB per K-tile against 1024 MFMA cycles) that is 27 KB per
// K-tile.  A flattened strip crosses image rows: where x + dx leaves the row the neighbouring line belongs to another row, and
// the A fragment of that output voxel is zeroed instead (the tap falls into the zero padding there).  Lines whose own (y + dy)
// is outside the image come from the zero page, t + dt < 0 replicates frame 0 (or reads the conv_cache frames), as in
// conv_gemm_kernel.  K-tile order: (dt, dy) major, channel block, dx minor; the weight column is ((dt*kh + dy)*3 + dx)*C + c.
// ---------------------------------------------------------------------------------------------------------------
// BM = 384 (N = 128 only: 96 accumulator registers) moves a third less W per FLOP than BM = 256.
template <int BM, int BN, int EPI>
__global__ __launch_bounds__(512) void conv_strip_kernel(const GemmArgs p, const ConvArgs cv) {
    constexpr int MB = BM / 128, NB = BN / 64;
    constexpr int SP = BM / 8 + 1;                 // DMA pieces (8 lines each) of one strip: BM + 8 >= BM + 2 lines
    constexpr int SQ = BM / 64;                    // pieces every wave moves; wave 0 moves one more
    static_assert(SQ % 2 == 0, "the strip pieces of a wave are split over the dx = 0 and dx = 1 K-tiles");
    constexpr int STRIP = SP * 1024, WT = BN * 128;
    constexpr int B_LD = BN / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // strip 0 | strip 1 | W 0 | W 1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_of_block(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int srow = lane >> 3, slot = lane & 7;

    // strip pieces of this wave: q = wave + 8 j (j < SQ), wave 0 also q = 8 SQ; lane -> strip line r = 8 q + srow = voxel m0 - 1 + r
    constexpr int SJ = SQ + 1;
    int vbs[SJ], vt[SJ], vy[SJ], vx[SJ], vchunk[SJ];   // first source frame of the batch element, t, y, x of the line's voxel
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
        const int q = j < SQ ? wave + 8 * j : 8 * SQ;
        const int r = q * 8 + srow;
        vchunk[j] = (slot ^ ((r >> 1) & 7)) * 8;
        long m = min(max((long)m0 - 1 + r, 0L), (long)p.M - 1);
        vx[j] = (int)(m % cv.W); m /= cv.W;
        vy[j] = (int)(m % cv.H); m /= cv.H;
        vt[j] = (int)(m % cv.T);
        vbs[j] = (int)(m / cv.T) * cv.Ts;
    }
    const bf16_t* b_src[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int row = (wave * B_LD + j) * 8 + srow;
        b_src[j] = p.W + (long)(n0 + row) * p.ldw + (slot ^ ((row >> 1) & 7)) * 8;
    }
    const int cblocks = cv.C >> 6;
    const int ngrp = cv.kt * cv.kh * cblocks;      // (dt, dy, channel block) groups of three K-tiles
    auto issue_strip = [&](int j, int sb, int grp) {           // piece j of this wave for group grp -> strip buffer sb
        const int tdy = grp / cblocks, c0 = (grp - tdy * cblocks) << 6;            // wave-uniform
        const int dy = tdy % cv.kh - cv.pad_lo, dt = tdy / cv.kh + cv.t_shift - (cv.kt - 1);
        const int ti = max(vt[j] + dt, 0), yi = vy[j] + dy;
        const bool ok = (unsigned)yi < (unsigned)cv.Hs;
        const unsigned idx = (unsigned)((vbs[j] + ti) * cv.Hs + yi) * (unsigned)cv.Ws + (unsigned)vx[j];
        const bf16_t* g = cv.src + (long)idx * cv.C + (c0 + vchunk[j]);
        if (!ok) g = (const bf16_t*)orv_zero_page + (lane & 7) * 8;
        const int q = j < SQ ? wave + 8 * j : 8 * SQ;
        glds16(g, smem + sb * STRIP + q * 1024);
    };
    auto issue_w = [&](int j, int wb, int grp, int dx) {
        const int tdy = grp / cblocks, cb = grp - tdy * cblocks;
        const long kcol = ((long)(tdy * 3 + dx) * cblocks + cb) * BK;
        glds16(b_src[j] + kcol, smem + 2 * STRIP + wb * WT + (wave * B_LD + j) * 1024);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int b_row_off = (wn * (BN / 2) + l31) * 128;
    // output rows of this lane's two A fragments and their x coordinate (for the dx = -1 / +1 edge masks)
    int arow[MB];
    bool edge_lo[MB], edge_hi[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        arow[j] = wm * (BM / 4) + j * 32 + l31;
        const int x = (int)(((long)m0 + arow[j]) % cv.W);
        edge_lo[j] = x == 0;
        edge_hi[j] = x == cv.W - 1;
    }
    f32x16 acc[NB][MB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: strip of group 0 and W of K-tile 0
#pragma unroll
    for (int j = 0; j < SJ; ++j)
        if (j < SQ || wave == 0) issue_strip(j, 0, 0);
#pragma unroll
    for (int j = 0; j < B_LD; ++j) issue_w(j, 0, 0, 0);

    int wb = 0;
    for (int grp = 0; grp < ngrp; ++grp) {
        const int sb = grp & 1;
        const bool more = grp + 1 < ngrp;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // DMA of the next K-tile's W (the last K-tile re-fetches W of K-tile 0, never read: uniform issue count) and this
            // K-tile's share of the next group's strip.  The k-step loop is left to the compiler's scheduler: the hand-placed
            // form of conv_gemm_kernel (fragments one k-step ahead, DMA issues pinned between the MFMAs) measured 3.5 % slower
            // here (71.2 vs 68.7 ms per decode, same box, three interleaved rounds)
            const int ng = dx == 2 ? (more ? grp + 1 : 0) : grp, ndx = dx == 2 ? 0 : dx + 1;
            const char* sA = smem + sb * STRIP;
            const char* sW = smem + 2 * STRIP + wb * WT;
            auto read_frags = [&](int ks, bf16x8 (&af)[MB], bf16x8 (&bf)[NB]) {
#pragma unroll
                for (int j = 0; j < MB; ++j) {
                    const int r = arow[j] + dx;                                   // strip line of tap dx for this output voxel
                    af[j] = *(const bf16x8*)(sA + r * 128 + (((ks * 2 + hi) ^ ((r >> 1) & 7)) * 16));
                    if ((dx == 0 && edge_lo[j]) || (dx == 2 && edge_hi[j])) af[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) bf[i] = *(const bf16x8*)(sW + b_row_off + i * 32 * 128 + (((ks * 2 + hi) ^ sw) * 16));
            };
#pragma unroll
            for (int j = 0; j < B_LD; ++j) issue_w(j, wb ^ 1, ng, ndx);
            if (more) {
                if (dx < 2) {
#pragma unroll
                    for (int j = 0; j < SQ / 2; ++j) issue_strip(dx * (SQ / 2) + j, sb ^ 1, grp + 1);
                } else if (wave == 0) issue_strip(SQ, sb ^ 1, grp + 1);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af1[MB], bf1[NB];
                read_frags(ks, af1, bf1);
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < MB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf1[i], af1[j], acc[i][j], 0, 0, 0);
            }
            wb ^= 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gemm_epilogue<NB, MB, EPI>(p, acc, m0 + wm * (BM / 4), n0 + wn * (BN / 2), lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Ring kernel for the large shapes (BM = 256, persistent: one workgroup per CU walks the tile list).
//   * Operands stream through a ring of NSLOT sub-stages of (256 + BN) rows x 32 K (64-byte rows; 16-byte chunk c of row
//     r lives at slot c ^ ((r >> 2) & 3): conflict-free ds_read_b128), filled by global_load_lds NSLOT-1 sub-stages
//     ahead of the reader.  The DMA stream is continuous across tile boundaries (the next tile's first sub-stages are
//     in flight while the current tile's epilogue runs) and is throttled only by counted s_waitcnt vmcnt(N): it never
//     drains at a barrier.
//   * ONE s_barrier per sub-stage.  Each wave holds the fragments of sub-stage g in registers (read during sub-stage
//     g-1), so after the barrier both waves of a SIMD have MFMAs ready at once: the matrix pipe never waits for LDS.
//     The ds_reads of sub-stage g+1 and the wave's DMA pieces of sub-stage g+NSLOT-1 are issued between the 12
//     v_mfma_f32_32x32x16_bf16 of sub-stage g.
//   * Measured alternatives on this chip (tools/abl.sh, tools/trace_gemm.cpp): a 2-stage BK=64 double buffer with one
//     vmcnt(0)+barrier per tile, and a strict ping-pong of two 4-wave groups (one computing, one reading LDS, a barrier
//     per phase) both stall the matrix pipe ~45 % of the time - the first on lock-step LDS latency + DMA issue bursts,
//     the second on the barrier itself (only one wave per SIMD ever has MFMAs to issue).
// ---------------------------------------------------------------------------------------------------------------
template <int BN, int NSLOT, int EPI, int NPC, bool ONESET = false>
__device__ __forceinline__ void gemm_ring_body(const GemmArgs& p, char* smem, const int wave, const int lane) {
    constexpr int BM = 256;
    constexpr int MB = 2, NB = BN / 64;            // wave sub-tile: 64 rows x BN/2 cols as 32x32 blocks
    constexpr int ROWS = BM + BN;
    constexpr int SLOT = ROWS * 64;                // bytes per sub-stage (32 bf16 per row)
    constexpr int NM = 2 * NB * MB;                // MFMAs per sub-stage per wave
    constexpr int PER = NM / (NPC + 1);
    const int grp = wave >> 2, wq = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nsub = p.K / 32;

    // ---- DMA side: a continuous stream of sub-stages over ALL tiles of this (persistent) workgroup ----
    // piece q = wave + 8i covers rows [16q, 16q+16) of the stacked [A tile; W tile]; lane -> (row, 16-B slot)
    const bf16_t* src[NPC];
    int it_tile = blockIdx.x;      // tile the DMA stream is currently fetching
    int it_sub = 0;                // next sub-stage of that tile
    int it_slot = 0;               // ring slot of the next sub-stage
#define ORV_SETUP_SRC(TILE)                                                                                          \
    {                                                                                                                \
        int tm_, tn_;                                                                                                \
        tile_of_index(p, min((TILE), ntiles - 1), ntiles, tm_, tn_);                                                 \
        _Pragma("unroll") for (int i = 0; i < NPC; ++i) {                                                            \
            const int row = (wave + 8 * i) * 16 + (lane >> 2);                                                       \
            const int chunk = (lane & 3) ^ ((lane >> 4) & 3); /* slot = lane & 3 = chunk ^ ((row >> 2) & 3) */       \
            if (row < BM) src[i] = p.A + (long)min(tm_ * BM + row, p.M - 1) * p.lda + chunk * 8;                     \
            else src[i] = p.W + (long)(tn_ * BN + min(row - BM, BN - 1)) * p.ldw + chunk * 8;                        \
        }                                                                                                            \
    }
    // advance the stream.  Past the last tile it re-fetches the final sub-stage into free ring slots (never read): the DMA
    // count per sub-stage stays uniform, so the counted vmcnt waits need no tail case and the MFMA stream no branches.
#define ORV_ADVANCE()                                                                                                \
    {                                                                                                                \
        it_slot = (it_slot + 1 == NSLOT) ? 0 : it_slot + 1;                                                          \
        if (++it_sub == nsub) {                                                                                      \
            it_tile += gridDim.x;                                                                                    \
            if (it_tile < ntiles) { it_sub = 0; ORV_SETUP_SRC(it_tile) }                                             \
            else it_sub = nsub - 1;                                                                                  \
        }                                                                                                            \
    }
#define ORV_ISSUE_PIECES()                                                                                           \
    _Pragma("unroll") for (int g = 0; g < NPC; ++g)                                                                  \
        glds16(src[g] + (long)it_sub * 32, smem + it_slot * SLOT + (wave + 8 * g) * 1024);
    ORV_SETUP_SRC(it_tile)

    // ---- fragment addressing ----
    const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 2) & 3;
    const int a_off = (wq * 64 + l31) * 64;
    const int b_off = (BM + grp * (BN / 2) + l31) * 64;
    const int coff0 = ((0 * 2 + hi) ^ sw) * 16, coff1 = ((1 * 2 + hi) ^ sw) * 16;
    // two fragment sets [k-step][block]; ONESET (BN = 384: 192 accumulator registers) keeps ONE k-step's fragments only
    bf16x8 fa0[ONESET ? 1 : 2][MB], fb0[ONESET ? 1 : 2][NB], fa1[ONESET ? 1 : 2][ONESET ? 1 : MB], fb1[ONESET ? 1 : 2][ONESET ? 1 : NB];
    int rd_slot = 0;               // ring slot of the next sub-stage to read
#define ORV_READ_FRAGS(FA, FB)                                                                                       \
    if constexpr (!ONESET) {                                                                                         \
        const char* sb_ = smem + rd_slot * SLOT;                                                                     \
        _Pragma("unroll") for (int j = 0; j < MB; ++j) {                                                             \
            FA[0][j] = *(const bf16x8*)(sb_ + a_off + j * 32 * 64 + coff0);                                          \
            FA[1][j] = *(const bf16x8*)(sb_ + a_off + j * 32 * 64 + coff1);                                          \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                             \
            FB[0][i] = *(const bf16x8*)(sb_ + b_off + i * 32 * 64 + coff0);                                          \
            FB[1][i] = *(const bf16x8*)(sb_ + b_off + i * 32 * 64 + coff1);                                          \
        }                                                                                                            \
        rd_slot = (rd_slot + 1 == NSLOT) ? 0 : rd_slot + 1;                                                          \
    }

    f32x16 acc[NB][MB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    constexpr int NSTORE = NB * MB * 2;   // 16-byte C stores one wave issues per full tile (no Y output)
    constexpr int AHEAD = ONESET ? NSLOT - 2 : NSLOT - 3;   // sub-stages that may still be in flight at the top of a sub-stage
    static_assert(AHEAD * NPC + NSTORE <= 63, "vmcnt immediate is 6 bits");
    int st_pending = 0;                   // sub-stages whose DMA pieces were issued before the last epilogue's stores

#ifdef ORV_GEMM_ABLATE_NOMFMA
#define ORV_MFMA(FA, FB, kk, i, j) asm volatile("" ::"v"(FB[kk][i]), "v"(FA[kk][j]));
#else
#define ORV_MFMA(FA, FB, kk, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[kk][i], FA[kk][j], acc[i][j], 0, 0, 0);
#endif
#ifdef ORV_GEMM_ABLATE_NOLOAD
#define ORV_ISSUE_MAIN()
#else
#define ORV_ISSUE_MAIN() ORV_ISSUE_PIECES()
#endif
    // one sub-stage: counted DMA wait + barrier (sub-stage g+1 is complete in LDS, slot g-1 is free), then
    //   ds_reads of g+1 -> fragment set NXT | MFMAs of g on fragment set CUR | DMA pieces of g+NSLOT-1 -> slot g-1
#define ORV_SUBSTAGE(FA_CUR, FB_CUR, FA_NXT, FB_NXT)                                                                 \
    {                                                                                                                \
        if (st_pending > 0) { /* epilogue stores of the previous tile still in flight: they are YOUNGER than the DMA */  \
            --st_pending;     /* pieces this wait is about (vmcnt retires in issue order), so allow them on top      */  \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 3) * NPC + NSTORE) : "memory");                        \
        } else {                                                                                                     \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 3) * NPC) : "memory");                                 \
        }                                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_READ_FRAGS(FA_NXT, FB_NXT)                                                                               \
        ORV_ISSUE_MAIN()                                                                                             \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
            _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                           \
                _Pragma("unroll") for (int j = 0; j < MB; ++j) { ORV_MFMA(FA_CUR, FB_CUR, kk, i, j) }                \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MB + NB), 0);                                               \
        if constexpr (NPC >= 1) { __builtin_amdgcn_sched_group_barrier(0x8, PER, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        if constexpr (NPC >= 2) { __builtin_amdgcn_sched_group_barrier(0x8, PER, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        if constexpr (NPC >= 3) { __builtin_amdgcn_sched_group_barrier(0x8, PER, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        if constexpr (NPC >= 4) { __builtin_amdgcn_sched_group_barrier(0x8, PER, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        __builtin_amdgcn_sched_group_barrier(0x8, NM - 1 - NPC * PER, 0);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_ADVANCE()                                                                                                \
    }

    // single-fragment-set form (ONESET): the wave reads the fragments of ITS OWN current sub-stage, one k-step at a time,
    // and leaves the LDS latency to the other wave of the SIMD; sub-stage g must have landed at the top of sub-stage g,
    // so one more sub-stage may stay in flight for the same ring depth.
#define ORV_READ_K(KK)                                                                                               \
    {                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < MB; ++j)                                                               \
            fa0[0][j] = *(const bf16x8*)(sb1_ + a_off + j * 32 * 64 + ((KK) ? coff1 : coff0));                       \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                               \
            fb0[0][i] = *(const bf16x8*)(sb1_ + b_off + i * 32 * 64 + ((KK) ? coff1 : coff0));                       \
    }
#define ORV_SUBSTAGE1()                                                                                              \
    {                                                                                                                \
        if (st_pending > 0) {                                                                                        \
            --st_pending;                                                                                            \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * NPC + NSTORE) : "memory");                              \
        } else {                                                                                                     \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * NPC) : "memory");                                       \
        }                                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        const char* sb1_ = smem + rd_slot * SLOT;                                                                    \
        rd_slot = (rd_slot + 1 == NSLOT) ? 0 : rd_slot + 1;                                                          \
        ORV_READ_K(0)                                                                                                \
        ORV_ISSUE_MAIN()                                                                                             \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                               \
            _Pragma("unroll") for (int j = 0; j < MB; ++j) { ORV_MFMA(fa0, fb0, 0, i, j) }                           \
        __builtin_amdgcn_sched_group_barrier(0x100, MB + NB, 0);                                                     \
        if constexpr (NPC >= 1) { __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        if constexpr (NPC >= 2) { __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        if constexpr (NPC >= 3) { __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        if constexpr (NPC >= 4) { __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        if constexpr (NPC >= 5) { __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); } \
        __builtin_amdgcn_sched_group_barrier(0x8, NB * MB - 2 * NPC, 0);                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_READ_K(1)                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                               \
            _Pragma("unroll") for (int j = 0; j < MB; ++j) { ORV_MFMA(fa0, fb0, 0, i, j) }                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_ADVANCE()                                                                                                \
    }

    // prologue: stream sub-stages 0 .. NSLOT-2, then fragments of sub-stage 0
#pragma unroll
    for (int s2 = 0; s2 < NSLOT - 1; ++s2) {
        ORV_ISSUE_PIECES()
        ORV_ADVANCE()
    }
    if constexpr (!ONESET) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * NPC) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    ORV_READ_FRAGS(fa0, fb0)
#ifdef ORV_GEMM_TRACE   // tools/trace_gemm.cpp: s_memtime stamps of workgroup 0 / wave 0 (R doubles as the trace buffer, EPI 0/1 only)
    int trace_i = 0;
#define ORV_TRACE(SLOT) if (blockIdx.x == 0 && threadIdx.x == 0 && trace_i < 16) ((unsigned long long*)p.R)[trace_i * 4 + SLOT] = __builtin_readcyclecounter();
#else
#define ORV_TRACE(SLOT)
#endif
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        ORV_TRACE(0)
        for (int j = 0; j < nsub; j += 2) {            // K % 64 == 0, so nsub is even: static fragment-set names
            if constexpr (ONESET) {
                ORV_SUBSTAGE1()
                ORV_SUBSTAGE1()
            } else {
                ORV_SUBSTAGE(fa0, fb0, fa1, fb1)
                ORV_SUBSTAGE(fa1, fb1, fa0, fb0)
            }
        }
        // finished tile -> memory.  vmcnt counts the epilogue's stores too and retires in issue order (loads and stores
        // share the counter on gfx9-class hardware), so the counted DMA waits of the next NSLOT-2 sub-stages - whose pieces
        // were issued BEFORE these stores - simply allow NSTORE more outstanding operations: the stores drain under the
        // next tile's MFMAs instead of stalling the wave here.  Exact store count is needed for that, so tiles with rows
        // past M (predicated stores) and launches with a Y output keep the plain drain.
        int tm, tn;
        ORV_TRACE(1)
        tile_of_index(p, tile, ntiles, tm, tn);
        gemm_epilogue<NB, MB, EPI>(p, acc, tm * BM + wq * 64, tn * BN + grp * (BN / 2), lane);
        ORV_TRACE(2)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j2 = 0; j2 < MB; ++j2)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j2][e] = 0.f;
        if (tm * BM + BM <= p.M && !p.Y && p.dbg != 7) {
            st_pending = ONESET ? NSLOT - 1 : NSLOT - 2;
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ORV_TRACE(3)
#ifdef ORV_GEMM_TRACE
        ++trace_i;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ORV_SETUP_SRC
#undef ORV_ADVANCE
#undef ORV_ISSUE_PIECES
#undef ORV_READ_FRAGS
#undef ORV_MFMA
#undef ORV_ISSUE_MAIN
#undef ORV_SUBSTAGE
#undef ORV_SUBSTAGE1
#undef ORV_READ_K
#undef ORV_TRACE
}

template <int BN, int NSLOT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmArgs p) {
    constexpr int PIECES = (256 + BN) / 16;        // 1-KiB DMA pieces (16 rows) per sub-stage
    constexpr int P0 = (PIECES + 7) / 8;
    // pieces per wave are uniform within each half of the workgroup: waves 0-3 own pieces q = w + 8i, waves 4-7 likewise
    constexpr int NP_G0 = (3 + 8 * (P0 - 1) < PIECES) ? P0 : P0 - 1;
    constexpr int NP_G1 = (7 + 8 * (P0 - 1) < PIECES) ? P0 : P0 - 1;
    static_assert((0 + 8 * (P0 - 1) < PIECES) == (3 + 8 * (P0 - 1) < PIECES) &&
                      (4 + 8 * (P0 - 1) < PIECES) == (7 + 8 * (P0 - 1) < PIECES),
                  "piece split must be uniform per wave group");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) gemm_ring_body<BN, NSLOT, EPI, NP_G0, (BN > 256)>(p, smem, wave, lane);
    else gemm_ring_body<BN, NSLOT, EPI, NP_G1, (BN > 256)>(p, smem, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Phased kernel (BM = 256, BK = 64, persistent): the 8-phase schedule (4 phases per K-tile, two K-tiles per loop trip).
//   * 8 waves as 4 (M) x 2 (N); a wave owns 64 rows x BN/2 columns = 2 x NB blocks of 32x32 (C^T accumulators, as above).
//     The two halves of the workgroup (waves 0-3 | 4-7: the two waves of every SIMD) run ONE barrier apart, so on each SIMD
//     one wave is in a load segment (ds_read + LDS-DMA issue) while its partner is in an MFMA segment.
//   * LDS: 2 stages x [A 256 rows | W BN rows] x 128 B (BK = 64: full 128-byte lines through the DMA path), XOR-swizzled
//     like gemm_kernel.  Every operand tile is stored as two HALF-tiles that are also the unit of consumption: A half h
//     holds m-block h of all four M-waves, W half h holds n-half h (NB/2 blocks) of both N-waves, so a half-tile is dead
//     for the current K-tile as soon as its one phase has read it.  Phase order (m0,n0) (m1,n0) (m1,n1) (m0,n1):
//         P1  read A0 (4 ds_read_b128), W0 (2 NB) | DMA W1 of K-tile g+1 | MFMA (m0,n0)
//         P2  read A1                             | DMA A0 of K-tile g+2 | MFMA (m1,n0)
//         P3  read W1                             | DMA W0 of K-tile g+2 | MFMA (m1,n1)
//         P4  -                                   | DMA A1 of K-tile g+2 | MFMA (m0,n1)   + the ONE counted vmcnt per K-tile
//     (A0 stays in registers for P4).  A half-tile is re-staged two phases after its last read (A0: one phase, its reads are
//     retired by a counted lgkmcnt before P1's first barrier); the vmcnt in P4 leaves the three youngest half-tiles in
//     flight and proves K-tile g+1 complete one barrier before its first read.
//   * The DMA stream is continuous across the output tiles of the persistent workgroup (each half-tile keeps its own
//     (output tile, K-tile) cursor), so the next tile's first two K-tiles land under the epilogue; past the last tile the
//     stream re-fetches it into dead half-tiles (uniform counts, no tail case).
// ---------------------------------------------------------------------------------------------------------------
template <int BN, int EPI>
__global__ __launch_bounds__(512) void gemm_ph_kernel(const GemmArgs p) {
    constexpr int BM = 256, MB = 2, NB = BN / 64, NBH = NB / 2;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    constexpr int APH = 2, BPH = BN / 128;     // DMA pieces (8 rows x 128 B) per wave per half-tile
    constexpr int HB = BN / 2, QB = BN / 4;    // W rows per half-tile / per N-wave inside a half-tile
    static_assert(NB % 2 == 0 && QB % 32 == 0 && NBH * 4 <= 15, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K / BK;                   // even (host-checked)

    // ---- DMA side: four half-tile streams, each with its own cursor ----
    const bf16_t *sA0[APH], *sA1[APH], *sB0[BPH], *sB1[BPH];
    int kA0 = 0, kA1 = 0, kB0 = 0, kB1 = 0;
    int tA0 = blockIdx.x, tA1 = blockIdx.x, tB0 = blockIdx.x, tB1 = blockIdx.x;
#ifdef ORV_PH_ABLATE_NODMA      /* ablation builds (tools/ph_abl.sh): separate compilations, never a run-time branch */
#define ORV_PH_GLDS(G, L) asm volatile("" ::"v"(G));
#else
#define ORV_PH_GLDS(G, L) glds16(G, L);
#endif
#ifdef ORV_PH_ABLATE_NOMFMA
#define ORV_PH_ONE_MFMA(B_, A_, C_) asm volatile("" ::"v"(B_), "v"(A_));
#else
#define ORV_PH_ONE_MFMA(B_, A_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B_, A_, C_, 0, 0, 0);
#endif
#define ORV_PH_SETUP_A(H, ARR, TILE)                                                                                 \
    {                                                                                                                \
        int tm_, tn_;                                                                                                \
        tile_of_index(p, min((TILE), ntiles - 1), ntiles, tm_, tn_);                                                 \
        _Pragma("unroll") for (int i = 0; i < APH; ++i) {                                                            \
            const int r = (H) * 128 + (wave + 8 * i) * 8 + (lane >> 3);       /* LDS row of the A region */          \
            const int trow = ((r >> 5) & 3) * 64 + (H) * 32 + (r & 31);       /* tile row it holds */                \
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);                                                           \
            ARR[i] = p.A + (long)min(tm_ * BM + trow, p.M - 1) * p.lda + chunk * 8;                                  \
        }                                                                                                            \
    }
#define ORV_PH_SETUP_B(H, ARR, TILE)                                                                                 \
    {                                                                                                                \
        int tm_, tn_;                                                                                                \
        tile_of_index(p, min((TILE), ntiles - 1), ntiles, tm_, tn_);                                                 \
        _Pragma("unroll") for (int i = 0; i < BPH; ++i) {                                                            \
            const int q = (wave + 8 * i) * 8 + (lane >> 3);                   /* row inside the half-tile */         \
            const int r = (H) * HB + q;                                       /* LDS row of the W region */          \
            const int tcol = (q / QB) * (BN / 2) + (H) * QB + q % QB;         /* tile column it holds */             \
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);                                                           \
            ARR[i] = p.W + (long)(tn_ * BN + tcol) * p.ldw + chunk * 8;                                              \
        }                                                                                                            \
    }
#define ORV_PH_ISSUE_A(H, ARR, KC, TC, S)                                                                            \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < APH; ++i)                                                              \
            ORV_PH_GLDS(ARR[i] + (long)KC * BK, smem + (S) * STAGE + ((H) * 128 + (wave + 8 * i) * 8) * 128);        \
        if (__builtin_expect(++KC == nk, 0)) { KC = 0; TC += gridDim.x; ORV_PH_SETUP_A(H, ARR, TC) }                                      \
    }
#define ORV_PH_ISSUE_B(H, ARR, KC, TC, S)                                                                            \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < BPH; ++i)                                                              \
            ORV_PH_GLDS(ARR[i] + (long)KC * BK, smem + (S) * STAGE + A_BYTES + ((H) * HB + (wave + 8 * i) * 8) * 128); \
        if (__builtin_expect(++KC == nk, 0)) { KC = 0; TC += gridDim.x; ORV_PH_SETUP_B(H, ARR, TC) }                                      \
    }
    ORV_PH_SETUP_A(0, sA0, tA0)
    ORV_PH_SETUP_A(1, sA1, tA1)
    ORV_PH_SETUP_B(0, sB0, tB0)
    ORV_PH_SETUP_B(1, sB1, tB1)

    // ---- fragment addressing: byte = row * 128 + ((2 ks + hi) ^ sw) * 16 ----
    const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = ((ks * 2 + hi) ^ sw) * 16;
    const int a_off = (wq * 32 + l31) * 128;                      // + h * 128 rows
    const int b_off = A_BYTES + (grp * QB + l31) * 128;           // + nh * HB rows + ii * 32 rows

    f32x16 acc[NB][MB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa0[4], fa1[4], fb0[NBH][4], fb1[NBH][4];

#define ORV_PH_READ_A(H, FA, S)                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                 \
        FA[ks] = *(const bf16x8*)(smem + (S) * STAGE + (H) * 128 * 128 + a_off + coff[ks]);
#define ORV_PH_READ_B(H, FB, S)                                                                                      \
    _Pragma("unroll") for (int ii = 0; ii < NBH; ++ii)                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                             \
            FB[ii][ks] = *(const bf16x8*)(smem + (S) * STAGE + ((H) * HB + ii * 32) * 128 + b_off + coff[ks]);
#define ORV_PH_MFMA(J, NH, FA, FB)                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                 \
        _Pragma("unroll") for (int ii = 0; ii < NBH; ++ii) { ORV_PH_ONE_MFMA(FB[ii][ks], FA[ks], acc[(NH) * NBH + ii][J]) } \
    __builtin_amdgcn_s_setprio(0);
#define ORV_PH_BAR()                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    __builtin_amdgcn_s_barrier();                                                                                    \
    __builtin_amdgcn_sched_barrier(0);
    // one K-tile out of stage S (compile-time); the DMA issues target stage S ^ 1 (W1 of the next K-tile) and stage S
#define ORV_PH_KTILE(S)                                                                                              \
    {                                                                                                                \
        /* P1 */                                                                                                     \
        ORV_PH_READ_A(0, fa0, S)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_PH_READ_B(0, fb0, S)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_PH_ISSUE_B(1, sB1, kB1, tB1, (S) ^ 1)                                                                    \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NBH * 4) : "memory");                                             \
        ORV_PH_BAR()                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_PH_MFMA(0, 0, fa0, fb0)                                                                                  \
        ORV_PH_BAR()                                                                                                 \
        /* P2 */                                                                                                     \
        ORV_PH_READ_A(1, fa1, S)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_PH_ISSUE_A(0, sA0, kA0, tA0, S)                                                                          \
        ORV_PH_BAR()                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_PH_MFMA(1, 0, fa1, fb0)                                                                                  \
        ORV_PH_BAR()                                                                                                 \
        /* P3 */                                                                                                     \
        ORV_PH_READ_B(1, fb1, S)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_PH_ISSUE_B(0, sB0, kB0, tB0, S)                                                                          \
        ORV_PH_BAR()                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ORV_PH_MFMA(1, 1, fa1, fb1)                                                                                  \
        ORV_PH_BAR()                                                                                                 \
        /* P4 */                                                                                                     \
        ORV_PH_ISSUE_A(1, sA1, kA1, tA1, S)                                                                          \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * APH + BPH) : "memory");                                         \
        ORV_PH_BAR()                                                                                                 \
        ORV_PH_MFMA(0, 1, fa0, fb1)                                                                                  \
        ORV_PH_BAR()                                                                                                 \
    }

    // prologue: K-tile 0 complete into stage 0, K-tile 1's A0, W0, A1 into stage 1
    ORV_PH_ISSUE_A(0, sA0, kA0, tA0, 0)
    ORV_PH_ISSUE_B(0, sB0, kB0, tB0, 0)
    ORV_PH_ISSUE_A(1, sA1, kA1, tA1, 0)
    ORV_PH_ISSUE_B(1, sB1, kB1, tB1, 0)
    ORV_PH_ISSUE_A(0, sA0, kA0, tA0, 1)
    ORV_PH_ISSUE_B(0, sB0, kB0, tB0, 1)
    ORV_PH_ISSUE_A(1, sA1, kA1, tA1, 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * APH + BPH) : "memory");
    ORV_PH_BAR()

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (grp == 1) { ORV_PH_BAR() }        // the second half of the workgroup runs one barrier behind ...
        for (int kt = 0; kt < nk; kt += 2) {
            ORV_PH_KTILE(0)
            ORV_PH_KTILE(1)
        }
        if (grp == 0) { ORV_PH_BAR() }        // ... and both halves run their epilogues side by side
        int tm, tn;
        tile_of_index(p, tile, ntiles, tm, tn);
        gemm_epilogue<NB, MB, EPI>(p, acc, tm * BM + wq * 64, tn * BN + grp * (BN / 2), lane);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ORV_PH_GLDS
#undef ORV_PH_ONE_MFMA
#undef ORV_PH_SETUP_A
#undef ORV_PH_SETUP_B
#undef ORV_PH_ISSUE_A
#undef ORV_PH_ISSUE_B
#undef ORV_PH_READ_A
#undef ORV_PH_READ_B
#undef ORV_PH_MFMA
#undef ORV_PH_BAR
#undef ORV_PH_KTILE
}

template <int BN>
int launch_ph(const GemmArgs& a, int epi, hipStream_t st) {
    const int smem = 2 * (256 + BN) * 128;
    const int grid = min(a.tiles_m * a.tiles_n, orv_num_cus());
#define ORV_GEMM_CASE(E)                                                                                    \
    case E: {                                                                                               \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            (void)hipFuncSetAttribute((const void*)gemm_ph_kernel<BN, E>,                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem);                    \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((gemm_ph_kernel<BN, E>), dim3(grid), dim3(512), smem, st, a);                    \
        break;                                                                                              \
    }
    switch (epi) {
        ORV_GEMM_CASE(0)
        ORV_GEMM_CASE(1)
        ORV_GEMM_CASE(2)
        ORV_GEMM_CASE(3)
        ORV_GEMM_CASE(4)
        default: orv_set_error("orv_gemm_bf16: bad epilogue %d", epi); return ORV_EINVAL;
    }
#undef ORV_GEMM_CASE
    return orv_check_launch("orv_gemm_bf16");
}

template <int BN, int NSLOT>
int launch_pp(const GemmArgs& a, int epi, hipStream_t st) {
    const int smem = NSLOT * (256 + BN) * 64;
    const int grid = min(a.tiles_m * a.tiles_n, orv_num_cus());   // persistent: one workgroup per CU walks the tile list
#define ORV_GEMM_CASE(E)                                                                                    \
    case E: {                                                                                               \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<BN, NSLOT, E>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem);                    \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((gemm_pp_kernel<BN, NSLOT, E>), dim3(grid), dim3(512), smem, st, a);             \
        break;                                                                                              \
    }
    switch (epi) {
        ORV_GEMM_CASE(0)
        ORV_GEMM_CASE(1)
        ORV_GEMM_CASE(2)
        ORV_GEMM_CASE(3)
        case 4:
            if constexpr ((BN / 2) % 64 == 0) {
                static bool attr_done4 = false;
                if (!attr_done4) {
                    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<BN, NSLOT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
                    attr_done4 = true;
                }
                hipLaunchKernelGGL((gemm_pp_kernel<BN, NSLOT, 4>), dim3(grid), dim3(512), smem, st, a);
                break;
            }
            [[fallthrough]];
        default: orv_set_error("orv_gemm_bf16: bad epilogue %d", epi); return ORV_EINVAL;
    }
#undef ORV_GEMM_CASE
    return orv_check_launch("orv_gemm_bf16");
}

template <int BM, int BN, int WM = 4, int NS = 2>
int launch(const GemmArgs& a, int epi, hipStream_t st) {
    const int smem = NS * (BM + BN) * 128;
    const int grid = a.tiles_m * a.tiles_n;
#define ORV_GEMM_CASE(E)                                                                                    \
    case E: {                                                                                               \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, E, WM, NS>,                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem);                    \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((gemm_kernel<BM, BN, E, WM, NS>), dim3(grid), dim3(512), smem, st, a);               \
        break;                                                                                              \
    }
    switch (epi) {
        ORV_GEMM_CASE(0)
        ORV_GEMM_CASE(1)
        ORV_GEMM_CASE(2)
        ORV_GEMM_CASE(3)
        case 4:
            if constexpr ((BN / (8 / WM)) % 64 == 0) {
                static bool attr_done4 = false;
                if (!attr_done4) {
                    (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, 4, WM, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
                    attr_done4 = true;
                }
                hipLaunchKernelGGL((gemm_kernel<BM, BN, 4, WM, NS>), dim3(grid), dim3(512), smem, st, a);
                break;
            }
            [[fallthrough]];
        default: orv_set_error("orv_gemm_bf16: bad epilogue %d", epi); return ORV_EINVAL;
    }
#undef ORV_GEMM_CASE
    return orv_check_launch("orv_gemm_bf16");
}

}  // namespace

struct GemmCand { int ring, bm, bn; float rate; int min_rounds; };
// relative rates of the d8 candidates (gemm_d8.hip); -DORV_D8_RATE_256=... / _192 at build time for A/B libraries
#ifndef ORV_D8_RATE_256
#define ORV_D8_RATE_256 1.29f
#endif
#ifndef ORV_D8_RATE_192
#define ORV_D8_RATE_192 1.21f
#endif
#define D8_RATE_256 ORV_D8_RATE_256
#define D8_RATE_192 ORV_D8_RATE_192
#ifndef ORV_D8_RATE_128
#define ORV_D8_RATE_128 1.20f
#endif
#define D8_RATE_128 ORV_D8_RATE_128
// 192-row d8 tiles (gemm_d8r192_kernel, round 6): 3/4 of the MFMAs of the 256-row sibling in ~0.81-0.85 of its time
#ifndef ORV_D8R192_RATE_192
#define ORV_D8R192_RATE_192 1.12f
#endif
#ifndef ORV_D8R192_RATE_128
#define ORV_D8R192_RATE_128 1.11f
#endif
// relative rates of the 192-row t8 tiles (gemm_t8r192_kernel): a tile does 3 / 4 of the MFMAs of its 256-row sibling on 7 / 8 of the LDS-DMA bytes
#ifndef ORV_T8R192_RATE_256
#define ORV_T8R192_RATE_256 1.20f
#endif
#ifndef ORV_T8R192_RATE_192
#define ORV_T8R192_RATE_192 1.12f
#endif
#define T8R192_RATE_256 ORV_T8R192_RATE_256
#define T8R192_RATE_192 ORV_T8R192_RATE_192
// C[M, N] (+)= A[K, M]^T . W[K, N]: the TN form of gemm_t8.hip (weight gradients: A = dY [tokens, out], W = X [tokens, in]; reference:
// torch autograd of nn.Linear inside accelerator.backward, train_cogvideox_control_to_video_sft.py:1093).  bf16 in, fp32 accumulate, bf16 out;
// accumulate != 0: C += the product (gradient accumulation).  N % 192 == 0 or N % 256 == 0, M % 8 == 0; any K (rows past K are zeros).
extern "C" int orv_gemm_tn_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, int accumulate,
                                void* stream) {
    ORV_REQUIRE(A && W && C, "orv_gemm_tn_bf16: null operand");
    ORV_REQUIRE(M > 0 && N > 0 && K > 0, "orv_gemm_tn_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    ORV_REQUIRE(M % 8 == 0 && (N % 192 == 0 || N % 256 == 0), "orv_gemm_tn_bf16: M=%d must be a multiple of 8 and N=%d of 192 or 256", M, N);
    ORV_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && lda >= M && ldw >= N && ldc >= N, "orv_gemm_tn_bf16: bad leading dimensions");
    ORV_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 7) == 0, "orv_gemm_tn_bf16: misaligned operand");
    ORV_REQUIRE(64L * lda * 2 + 2L * M < (1L << 32) && 64L * ldw * 2 + 2L * N < (1L << 32), "orv_gemm_tn_bf16: row stride too large");
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.C = (bf16_t*)C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K;
    // tile width: the one whose tile count wastes less of its last round of CUs (ties: 256)
    const int ncu = orv_num_cus();
    auto waste = [&](int bn) {
        if (N % bn) return 1e9;
        const long tiles = (long)((M + 255) / 256) * (N / bn);
        const long rounds = (tiles + ncu - 1) / ncu;
        return (double)(rounds * ncu) * bn;            // CU-rounds x tile width ~ time
    };
    const int bn = waste(256) <= waste(192) ? 256 : 192;
    a.tiles_n = N / bn;
    a.tiles_m = (M + 255) / 256;
    { static int wb = -1; if (wb < 0) { const char* e = getenv("ORV_GEMM_WALK_BACK"); wb = e ? atoi(e) : 2; } a.walk_back = wb == 2 ? 1 : 0; }
    return launch_t8_tn(a, bn, accumulate ? 1 : 0, (hipStream_t)stream);
}

// ORV_GEMM_TILE="ring,bm,bn" / orv_gemm_force_tile() pin one candidate (sweeps, same-process A/B, the per-instantiation tests)
static int g_force_ring = -1, g_force_bm = 0, g_force_bn = 0, g_force_epoch = 0;
extern "C" int orv_gemm_force_epoch(void) { return g_force_epoch; }
extern "C" int orv_gemm_force_tile(int ring, int bm, int bn) {
    ++g_force_epoch;
    if (g_force_ring < 0) g_force_ring = 2;
    if (bm > 0) { g_force_ring = ring; g_force_bm = bm; g_force_bn = bn; }
    else { g_force_ring = 2; g_force_bm = 0; g_force_bn = 0; }
    return ORV_OK;
}

// Tile / kernel choice: every candidate whose BN divides N is priced as
//     rounds(tiles over the CUs) x BM x BN / relative_rate(candidate)
// and the cheapest wins.  The rates are measured on MI355X at M = 12904 (tools/tile_sweep.sh rates); the rounds term is
// what matters at small batch, where a "better" tile that needs one more, nearly empty, round loses to a smaller one that
// fills the chip (B = 1: N = 1920 GEMMs take 195 tiles of 256x128 instead of 260 of 128x192).
// wide: an operand spans 4 GiB or more - the t8 kernel addresses A and W with 32-bit byte offsets and is skipped
static const GemmCand* choose_tile(int M, int N, int K, int epilogue = 0, int heads = 0, bool wide = false, bool packed = false, bool cpacked = false) {
    static const GemmCand cands[] = {
        // ring = 3: the 16x16x32 8-phase kernel of gemm_t8.hip (persistent, BK = 64; needs an even number of K-tiles)
        {3, 256, 256, 1.29f, 0}, {3, 256, 192, 1.21f, 0},
        // the same kernel on 192-row tiles (gemm_t8r192_kernel, round 5): M = 3226 (one clip) is 17 row tiles, M = 6452 is 34 - 510 / 255 / 170 tiles
        // for N = 7680 / 3840 / 1920 where 256 rows give 390 / 195 / 130; at M = 12904 the 256-row tiles stay cheaper (rates: T8R192_RATE_*)
        {3, 192, 256, T8R192_RATE_256, 0}, {3, 192, 192, T8R192_RATE_192, 0},
        // ring = 4: four-wave 256 x 256 experiment of gemm_t8.hip - forced only (rate 0.01 never wins)
        {4, 256, 256, 0.01f, 0},
        // ring = 5: gemm_d8.hip (round 5) - A straight to registers two K-tiles ahead, W through four LDS buffers; needs K % 192 == 0
        {5, 256, 256, D8_RATE_256, 0}, {5, 256, 192, D8_RATE_192, 0}, {5, 256, 128, D8_RATE_128, 0},
        // the d8 kernel on 192-row tiles (gemm_d8r192_kernel, round 6): the first wave of every SIMD owns two 16-row blocks, the second one.
        // M = 3226 (one clip): N = 1920 is 255 tiles of 192 x 128 (one FULL round; 256 x 128 leaves 61 CUs idle), q | k | v 510 of 192 x 192
        {5, 192, 192, ORV_D8R192_RATE_192, 0}, {5, 192, 128, ORV_D8R192_RATE_128, 0},
        // ring = 2: the phased (8-phase, BK = 64) persistent kernel; needs an even number of K-tiles
        // (256x384 does not fit: 192 accumulator + 64 fragment registers of the 256 a wave gets at two waves per SIMD)
        {2, 256, 256, 1.10f, 0}, {2, 256, 128, 0.80f, 0},
        // 256x384 keeps ONE fragment set (192 accumulator registers) and leans on its neighbour tiles to hide the exposed
        // prologue / epilogue: measured +8 % on QKV (765 tiles), a loss when every CU gets a single tile (N = 1920: 255)
        {1, 256, 384, 1.075f, 2}, {1, 256, 256, 1.060f, 0}, {1, 256, 192, 1.000f, 0}, {1, 256, 128, 0.885f, 0},
        {0, 256, 192, 0.975f, 0}, {0, 256, 128, 0.935f, 0}, {0, 256, 64, 0.855f, 0},
        {0, 128, 192, 0.965f, 0}, {0, 128, 128, 0.855f, 0}, {0, 128, 64, 0.760f, 0},
        // 192 rows (2 x 4 wave grid): M = 3226 (one clip) is 17 tiles, 17 x 15 = 255 tiles for N = 1920
        {0, 192, 128, 0.920f, 0},
    };
    if (g_force_ring < 0) {
        g_force_ring = 2;
        if (const char* e = getenv("ORV_GEMM_TILE")) sscanf(e, "%d,%d,%d", &g_force_ring, &g_force_bm, &g_force_bn);
        if (const char* e = getenv("ORV_GEMM_ALGO")) { if (atoi(e) == 0) g_force_ring = 0; }   // legacy switch: simple kernel only
    }
    const int force_ring = g_force_ring, force_bm = g_force_bm, force_bn = g_force_bn;
    static int no_phased = -1;   // ORV_GEMM_PHASED=0: A/B switch for the phased kernel
    if (no_phased < 0) { const char* e = getenv("ORV_GEMM_PHASED"); no_phased = (e && atoi(e) == 0) ? 1 : 0; }
    static int no_t8 = -1;       // ORV_GEMM_T8=0: A/B switch for the t8 kernel
    if (no_t8 < 0) { const char* e = getenv("ORV_GEMM_T8"); no_t8 = (e && atoi(e) == 0) ? 1 : 0; }
    static int no_r192 = -1;     // ORV_GEMM_R192=0: A/B switch for the 192-row t8 tiles
    if (no_r192 < 0) { const char* e = getenv("ORV_GEMM_R192"); no_r192 = (e && atoi(e) == 0) ? 1 : 0; }
    static int no_d8 = -1;       // ORV_GEMM_D8=0: A/B switch for the d8 kernel
    if (no_d8 < 0) { const char* e = getenv("ORV_GEMM_D8"); no_d8 = (e && atoi(e) == 0) ? 1 : 0; }
    static int no_d8r192 = -1;   // ORV_GEMM_D8R192=0: A/B switch for the 192-row d8 tiles
    if (no_d8r192 < 0) { const char* e = getenv("ORV_GEMM_D8R192"); no_d8r192 = (e && atoi(e) == 0) ? 1 : 0; }
    static int no384 = -1;   // ORV_GEMM_BN384=0: A/B switch for the 384-wide variant
    if (no384 < 0) { const char* e = getenv("ORV_GEMM_BN384"); no384 = (e && atoi(e) == 0) ? 1 : 0; }
    const int ncu = orv_num_cus();
    const GemmCand* best = nullptr;
    double best_cost = 0;
    for (const GemmCand& c : cands) {
        if (N % c.bn) continue;
        if (c.ring == 2 && (K % 128 != 0 || no_phased)) continue;
        if (c.ring == 3 && (K % 128 != 0 || no_t8 || wide || (epilogue == 4 && c.bn != 256))) continue;
        if (c.ring == 3 && c.bm == 192 && (no_r192 || epilogue == 3)) continue;
        // packed C has orv_packed_rows(M) = ceil(M / 256) * 256 row slots and the packed store is not masked: the 192-row tiling must fit them
        if (c.ring == 3 && c.bm == 192 && cpacked && (long)((M + 191) / 192) * 192 > (long)((M + 255) / 256) * 256) continue;
        if (c.ring == 4 && (K % 128 != 0 || wide || epilogue > 2 || force_ring != 4)) continue;
        // ring 5 reads A in the packed P16 layout and nothing else does: the caller's a_packed decides the family
        if ((c.ring == 5) != packed) continue;
        // packed C from row-major A: only the t8 kernel's 256-wide GELU epilogue writes it
        if (cpacked && !packed && !(c.ring == 3 && c.bn == 256 && epilogue == 1)) continue;
        if (c.ring == 5 && (K % 192 != 0 || no_d8)) continue;
        // 192-row d8 tiles: the packed A (and C) buffers hold ceil(M / 256) * 256 row slots and neither A loads nor the packed store are masked
        if (c.ring == 5 && c.bm == 192 && (no_d8r192 || epilogue == 3 || (long)((M + 191) / 192) * 192 > (long)((M + 255) / 256) * 256)) continue;
        // epilogue 4 normalises whole 64-wide heads inside a wave (BN / 2 columns) that must not straddle q | k | v
        if (epilogue == 4 && c.ring != 5 && ((c.bm == 192 && c.ring != 3) || (c.bn / 2) % 64 != 0 || (heads * 64) % (c.bn / 2) != 0)) continue;
        if (force_bm && (c.ring != force_ring || c.bm != force_bm || c.bn != force_bn)) continue;
        if (!force_bm && force_ring == 0 && c.ring) continue;
        const long tiles = (long)((M + c.bm - 1) / c.bm) * (N / c.bn);
        // (the 384-wide ring needs neighbour tiles to hide its prologue / epilogue - except when ONE tile per CU is a long K
        // sweep: FFN2, N = 1920 = 5 x 384, K = 7680 is 255 tiles of 120 K-tiles: 0.391 vs 0.403 ms on 256 x 192)
        const bool one_long_round = K >= 4096 && tiles <= ncu && tiles * 20 >= (long)ncu * 19;
        if (!force_bm && tiles < (long)c.min_rounds * ncu && !one_long_round) continue;
        if (!force_bm && c.ring == 1 && c.bn == 384 && no384) continue;
        // full rounds cost 1 each; the last, partial round runs faster than a full one because the chip is power-capped
        // (fewer active CUs clock higher): 0.62 (= 1.5 GHz / 2.4 GHz) + 0.38 x the fraction of CUs it occupies.
        // Rows of the last M tile that do not exist still cost their MFMAs (tiles are counted whole).
        const long full = tiles / ncu, rem = tiles % ncu;
        const double rounds = (double)full + (rem ? 0.62 + 0.38 * (double)rem / ncu : 0.0);
        const double cost = rounds * c.bm * c.bn / c.rate;
        if (!best || cost < best_cost) { best = &c; best_cost = cost; }
    }
    return best;
}

// Launch plan of one orv_gemm_bf16 call: one candidate, or - fused-qk-LayerNorm projection whose width is not a multiple of
// 256 (2B: N = 5760) - two: the t8 kernel normalises whole heads per wave only at BN = 256, so the call is split into q | k
// (N = 2 H 64 = 3840 = 15 x 256, epilogue 4) and v (N = H 64 = 1920, plain bias epilogue into the same packed buffer) whenever the
// cost model puts the q | k part on the t8 kernel: 0.247 vs 0.286 ms per layer at B = 4 (profiles/r3_gemm_t8_ab.txt).  The A
// operand is read by both launches (L2 / Infinity Cache).  ORV_GEMM_QKV_SPLIT=0: A/B switch.
static bool plan_gemm(int M, int N, int K, int epilogue, int heads, const GemmCand*& first, const GemmCand*& second, bool wide = false,
                      bool packed = false, bool cpacked = false) {
    static int qkv_split = -1;
    if (qkv_split < 0) { const char* e = getenv("ORV_GEMM_QKV_SPLIT"); qkv_split = (e && atoi(e) == 0) ? 0 : 1; }
    second = nullptr;
    first = choose_tile(M, N, K, epilogue, heads, wide, packed, cpacked);
    if (packed || cpacked) return first != nullptr;      // the d8 kernel normalises 64-column groups at any BN: no q | k + v split
    if (epilogue == 4 && qkv_split && heads > 0 && N == 3 * heads * 64 && !(first && first->ring == 3)) {
        const GemmCand* qk = choose_tile(M, 2 * heads * 64, K, 4, heads, wide);
        const GemmCand* vv = choose_tile(M, heads * 64, K, 0, 0, wide);
        if (qk && qk->ring == 3 && vv) { first = qk; second = vv; }
    }
    return first != nullptr;
}
static void cand_name(const GemmCand* c, int epilogue, char* buf, int len) {
    if (c->ring == 3 && c->bm == 192) snprintf(buf, len, "gemm_t8r192_kernel<%d, %d>", c->bn, epilogue);
    else if (c->ring == 3) snprintf(buf, len, "gemm_t8_kernel<%d, %d>", c->bn, epilogue);
    else if (c->ring == 5 && c->bm == 192) snprintf(buf, len, "gemm_d8r192_kernel<%d, %d>", c->bn, epilogue);
    else if (c->ring == 5) snprintf(buf, len, "gemm_d8_kernel<%d, %d>", c->bn, epilogue);
    else if (c->ring == 2) snprintf(buf, len, "gemm_ph_kernel<%d, %d>", c->bn, epilogue);
    else if (c->ring) snprintf(buf, len, "gemm_pp_kernel<%d, %d, %d>", c->bn, c->bn == 384 ? 4 : 5, epilogue);
    else if (c->bm == 192) snprintf(buf, len, "gemm_kernel<%d, %d, %d, 2, 2>", c->bm, c->bn, epilogue);
    else snprintf(buf, len, "gemm_kernel<%d, %d, %d, 4, 2>", c->bm, c->bn, epilogue);
}

// the kernel symbol orv_gemm_bf16 launches for a shape, as rocprofv3 prints it (bench.py labels its timings with it)
extern "C" int orv_gemm_kernel_name(int M, int N, int K, int epilogue, char* buf, int len) {
    ORV_REQUIRE(buf && len > 0 && M > 0 && N > 0 && N % 64 == 0, "orv_gemm_kernel_name: bad arguments");
    const GemmCand *c = nullptr, *c2 = nullptr;
    ORV_REQUIRE(plan_gemm(M, N, K, epilogue, epilogue == 4 ? N / 192 : 0, c, c2), "orv_gemm_kernel_name: no tile configuration for N=%d", N);
    cand_name(c, epilogue, buf, len);
    if (c2) {                     // split q | k + v launch: both symbols
        const int n = (int)strlen(buf);
        if (n + 4 < len) { snprintf(buf + n, len - n, " + "); cand_name(c2, 0, buf + n + 3, len - n - 3); }
    }
    return ORV_OK;
}

// orv_gemm_kernel_name for a call with packed operands (orv_gemm_t.a_packed / c_packed): fails when no kernel takes that combination
extern "C" int orv_gemm_kernel_name_packed(int M, int N, int K, int epilogue, int a_packed, int c_packed, char* buf, int len) {
    ORV_REQUIRE(buf && len > 0 && M > 0 && N > 0 && N % 64 == 0, "orv_gemm_kernel_name_packed: bad arguments");
    const GemmCand *c = nullptr, *c2 = nullptr;
    ORV_REQUIRE(plan_gemm(M, N, K, epilogue, epilogue == 4 ? N / 192 : 0, c, c2, false, a_packed != 0, c_packed != 0),
                "orv_gemm_kernel_name_packed: no kernel for N=%d K=%d epilogue %d a_packed=%d c_packed=%d", N, K, epilogue, a_packed, c_packed);
    cand_name(c, epilogue, buf, len);
    return ORV_OK;
}

// launch the chosen candidate
static int gemm_dispatch(GemmArgs a, const GemmCand* best, int epilogue, hipStream_t st) {
    a.tiles_n = a.N / best->bn;
    a.tiles_m = (a.M + best->bm - 1) / best->bm;
    {
        static int gm_env = -1;
        if (gm_env < 0) { const char* e = getenv("ORV_GEMM_GM"); gm_env = e ? atoi(e) : 0; }
        a.gm = gm_env > 0 ? gm_env : (a.K >= 4096 && a.tiles_n <= 16 ? 8 : 0);
    }
    if (best->ring == 3) return launch_t8(a, best->bn, epilogue, st, best->bm);
    if (best->ring == 4) return launch_t4(a, epilogue, st);
    if (best->ring == 5) return launch_d8(a, best->bn, epilogue, st, best->bm);
    if (best->ring == 2) {
        if (best->bn == 256) return launch_ph<256>(a, epilogue, st);
        return launch_ph<128>(a, epilogue, st);
    }
    if (best->ring) {
        if (best->bn == 384) return launch_pp<384, 4>(a, epilogue, st);
        if (best->bn == 256) return launch_pp<256, 5>(a, epilogue, st);
        if (best->bn == 192) return launch_pp<192, 5>(a, epilogue, st);
        return launch_pp<128, 5>(a, epilogue, st);
    }
    if (best->bm == 192) return launch<192, 128, 2>(a, epilogue, st);
    if (best->bm == 256) {
        if (best->bn == 192) return launch<256, 192>(a, epilogue, st);
        if (best->bn == 128) return launch<256, 128>(a, epilogue, st);
        return launch<256, 64>(a, epilogue, st);
    }
    if (best->bn == 192) return launch<128, 192>(a, epilogue, st);
    if (best->bn == 128) return launch<128, 128>(a, epilogue, st);
    return launch<128, 64>(a, epilogue, st);
}

// BN must divide N: 192 divides 1920/5760/7680 (the 2B model) exactly, 128/256 cover 3072-wide (5B) and the tiny test
// widths, 64 the 64-wide proj_out.
extern "C" int orv_gemm_bf16(const orv_gemm_t* g, void* stream) {
    ORV_REQUIRE(g && g->A && g->W && g->C, "orv_gemm_bf16: null operand");
    ORV_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, "orv_gemm_bf16: empty problem M=%d N=%d K=%d", g->M, g->N, g->K);
    ORV_REQUIRE(g->K % 64 == 0, "orv_gemm_bf16: K=%d must be a multiple of 64", g->K);
    ORV_REQUIRE(g->N % 64 == 0, "orv_gemm_bf16: N=%d must be a multiple of 64", g->N);
    ORV_REQUIRE(g->lda % 8 == 0 && g->ldw % 8 == 0 && g->ldc % 8 == 0, "orv_gemm_bf16: misaligned leading dimension");
    ORV_REQUIRE(((uintptr_t)g->C & 15) == 0 && (!g->R || (((uintptr_t)g->R & 15) == 0 && g->ldr % 8 == 0)) &&
                    (!g->Y || (((uintptr_t)g->Y & 15) == 0 && g->ldy % 8 == 0)),
                "orv_gemm_bf16: C / R / Y must be 16-byte aligned with leading dimensions that are multiples of 8");
    ORV_REQUIRE((g->epilogue != 2 && g->epilogue != 3) || g->R, "orv_gemm_bf16: epilogue 2/3 needs R");
    ORV_REQUIRE(g->epilogue != 2 || !g->gate || g->grp.seq > 0, "orv_gemm_bf16: gate needs grp.seq");
    GemmArgs a;
    a.gm = 0;
    a.A = (const bf16_t*)g->A; a.lda = g->lda; a.W = (const bf16_t*)g->W; a.ldw = g->ldw;
    a.bias = (const bf16_t*)g->bias; a.C = (bf16_t*)g->C; a.ldc = g->ldc;
    a.M = g->M; a.N = g->N; a.K = g->K;
    a.R = (const bf16_t*)g->R; a.ldr = g->ldr; a.r_mod = g->r_mod;
    a.gate = g->gate; a.gate_b = g->gate_b; a.gate_g = g->gate_g;
    a.seq = g->grp.seq; a.n_text = g->grp.n_text; a.per_group = g->grp.per_group;
    a.c_rows = g->cmap.rows; a.c_bstride = g->cmap.bstride; a.c_off = g->cmap.off;
    a.Y = (bf16_t*)g->Y; a.ldy = g->ldy;
    a.a_packed = g->a_packed; a.c_packed = g->c_packed;
    a.qn_gq = (const bf16_t*)g->qn_gamma_q; a.qn_bq = (const bf16_t*)g->qn_beta_q; a.qn_gk = (const bf16_t*)g->qn_gamma_k;
    a.qn_bk = (const bf16_t*)g->qn_beta_k; a.qn_eps = g->qn_eps; a.qn_premul = g->qn_premul; a.qn_heads = g->qn_heads;
    if (g->epilogue == 4) {
        ORV_REQUIRE(g->qn_heads > 0 && (g->N == 3 * g->qn_heads * 64 || g->N == 2 * g->qn_heads * 64),
                    "orv_gemm_bf16: epilogue 4 needs N = 3 * heads * 64 (q | k | v) or 2 * heads * 64 (q | k) (N=%d heads=%d)", g->N, g->qn_heads);
        ORV_REQUIRE(g->cmap.rows == 0, "orv_gemm_bf16: epilogue 4 writes rows in place (no cmap)");
    }
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("ORV_GEMM_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
    {   // ORV_T8_STAGGER="groups,ns": start stagger of the t8 workgroups; ORV_T8_GRID: cap of persistent workgroups (experiments)
        static int sg = -1, st_ticks = 0, gcap = 0;
        if (sg < 0) {
            sg = 0;
            if (const char* e = getenv("ORV_T8_STAGGER")) { int ns = 0; if (sscanf(e, "%d,%d", &sg, &ns) == 2) st_ticks = ns / 10; else sg = 0; }
            if (const char* e = getenv("ORV_T8_GRID")) gcap = atoi(e);
        }
        a.stagger_groups = sg; a.stagger_ticks = st_ticks; a.grid_cap = gcap;
    }
    {   // tiles handed out from the END of the list (GemmArgs::walk_back).  ORV_GEMM_WALK_BACK: 0 never, 1 the gated-residual GEMMs
        // only (out-projection, FFN2), 2 (default) every GEMM
        static int wb = -1;
        if (wb < 0) { const char* e = getenv("ORV_GEMM_WALK_BACK"); wb = e ? atoi(e) : 2; }
        a.walk_back = (wb == 2 || (wb == 1 && g->epilogue == 2 && g->gate)) ? 1 : 0;
    }
    hipStream_t st = (hipStream_t)stream;
    const GemmCand *first = nullptr, *second = nullptr;
    const bool wide = (long)g->M * g->lda * 2 >= (1L << 32) || (long)g->N * g->ldw * 2 >= (1L << 32);
    ORV_REQUIRE(!g->a_packed || g->lda == g->K, "orv_gemm_bf16: a packed A has lda == K (lda=%d K=%d)", g->lda, g->K);
    ORV_REQUIRE(!g->c_packed || g->a_packed || g->epilogue == 1, "orv_gemm_bf16: packed C from a row-major A needs epilogue 1 (the t8 GELU epilogue)");
    ORV_REQUIRE(plan_gemm(g->M, g->N, g->K, g->epilogue, g->qn_heads, first, second, wide, g->a_packed != 0, g->c_packed != 0),
                "orv_gemm_bf16: no tile configuration for N=%d (ORV_GEMM_TILE override?)", g->N);
    if (!second) return gemm_dispatch(a, first, g->epilogue, st);
    const int nqk = 2 * g->qn_heads * 64;
    GemmArgs q = a;
    q.N = nqk;
    const int rc = gemm_dispatch(q, first, 4, st);
    if (rc != ORV_OK) return rc;
    GemmArgs v = a;
    v.N = g->N - nqk; v.W = a.W + (long)nqk * a.ldw; v.C = a.C + nqk;
    if (a.bias) v.bias = a.bias + nqk;
    if (a.Y) v.Y = a.Y + nqk;
    return gemm_dispatch(v, second, 0, st);
}

// Implicit-GEMM convolution entry point: g->A is ignored (the A operand is gathered from c->src), g->M = B*T*H*W, g->K = taps * C.
extern "C" int orv_conv_gemm_bf16(const orv_gemm_t* g, const orv_conv_t* c, void* stream) {
    ORV_REQUIRE(g && c && c->src && g->W && g->C, "orv_conv_gemm_bf16: null operand");
    ORV_REQUIRE(c->C % 64 == 0, "orv_conv_gemm_bf16: C=%d must be a multiple of 64 (use orv_vae_im2col + orv_gemm_bf16 otherwise)", c->C);
    ORV_REQUIRE(c->kt >= 1 && c->kh >= 1 && c->kw >= 1 && (c->stride == 1 || c->stride == 2) && c->pad_lo >= 0 && c->ups_t >= 0 &&
                    c->ups_t <= 2 && c->t_shift >= 0 && c->t_shift <= c->kt - 1 && (c->t_shift == 0 || c->ups_t == 0),
                "orv_conv_gemm_bf16: bad kernel geometry");
    ORV_REQUIRE(g->K == c->kt * c->kh * c->kw * c->C && (long)g->M == (long)c->B * c->T * c->H * c->W,
                "orv_conv_gemm_bf16: M / K do not match the convolution geometry");
    ORV_REQUIRE(g->N % 64 == 0 && g->ldw % 8 == 0 && g->ldc % 8 == 0 && (g->epilogue == 0 || g->epilogue == 2),
                "orv_conv_gemm_bf16: N=%d must be a multiple of 64; epilogues 0 (bias) and 2 (residual) only", g->N);
    ORV_REQUIRE(g->epilogue != 2 || g->R, "orv_conv_gemm_bf16: epilogue 2 needs R");
    ORV_REQUIRE((long)c->B * c->Ts * c->Hs * c->Ws < (1L << 31), "orv_conv_gemm_bf16: more than 2^31 source voxels");
    GemmArgs a{};
    a.A = nullptr; a.lda = 0; a.W = (const bf16_t*)g->W; a.ldw = g->ldw; a.bias = (const bf16_t*)g->bias;
    a.C = (bf16_t*)g->C; a.ldc = g->ldc; a.M = g->M; a.N = g->N; a.K = g->K;
    a.R = (const bf16_t*)g->R; a.ldr = g->ldr; a.r_mod = g->r_mod; a.gate = nullptr;
    ConvArgs cv{(const bf16_t*)c->src, c->B, c->Ts, c->Hs, c->Ws, c->C, c->T, c->H, c->W, c->kt, c->kh, c->kw, c->stride, c->pad_lo,
                c->ups_s, c->ups_t, c->t_shift};
    hipStream_t st0 = (hipStream_t)stream;
    static int use_strip = -1;       // ORV_CONV_STRIP=0: A/B switch
    if (use_strip < 0) { const char* e = getenv("ORV_CONV_STRIP"); use_strip = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_strip && c->stride == 1 && !c->ups_s && !c->ups_t && c->Hs == c->H && c->Ws == c->W && c->kw == 3 && c->pad_lo == 1 &&
        c->W >= 2 && g->N % 128 == 0 && g->r_mod == 0) {
        static int bm384 = -1;       // ORV_CONV_BM384=0: A/B switch for the 384-row tile of the 128-wide convolutions
        if (bm384 < 0) { const char* e = getenv("ORV_CONV_BM384"); bm384 = (e && atoi(e) == 0) ? 0 : 1; }
        static int small = -1;       // ORV_CONV_SMALL=0: A/B switch for the small-grid tiles
        if (small < 0) { const char* e = getenv("ORV_CONV_SMALL"); small = (e && atoi(e) == 0) ? 0 : 1; }
        // tile: 256 x 256 where it fills the chip; the low-resolution stages of the decoder (7200 voxels x 512 channels: 58 tiles
        // of 256 x 256 on 256 CUs) take 256 x 128, then 128 x 128 tiles while the grid covers less than half of the CUs
        const int ncu = orv_num_cus();
        int bm = 256, bn = g->N % 256 == 0 ? 256 : 128;
        if (small && bn == 256 && (long)((g->M + 255) / 256) * (g->N / 256) < ncu / 2) bn = 128;      // (at 150-190 of 256 tiles
        if (small && bn == 128 && (long)((g->M + 255) / 256) * (g->N / 128) < ncu / 2) bm = 128;      //  the big tile still wins)
        if (bm == 256 && bn == 128 && bm384 && g->M >= 384 * 256) bm = 384;
        a.tiles_n = g->N / bn;
        a.tiles_m = (g->M + bm - 1) / bm;
        const int smem = 2 * (bm / 8 + 1) * 1024 + 2 * bn * 128;
#define ORV_STRIP_LAUNCH(BM_, BN_, E)                                                                                  \
    {                                                                                                                  \
        static bool done = false;                                                                                      \
        if (!done) { (void)hipFuncSetAttribute((const void*)conv_strip_kernel<BM_, BN_, E>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); done = true; } \
        hipLaunchKernelGGL((conv_strip_kernel<BM_, BN_, E>), dim3(a.tiles_m * a.tiles_n), dim3(512), smem, st0, a, cv); \
    }
        if (bn == 256) { if (g->epilogue == 2) ORV_STRIP_LAUNCH(256, 256, 2) else ORV_STRIP_LAUNCH(256, 256, 0) }
        else if (bm == 384) { if (g->epilogue == 2) ORV_STRIP_LAUNCH(384, 128, 2) else ORV_STRIP_LAUNCH(384, 128, 0) }
        else if (bm == 128) { if (g->epilogue == 2) ORV_STRIP_LAUNCH(128, 128, 2) else ORV_STRIP_LAUNCH(128, 128, 0) }
        else { if (g->epilogue == 2) ORV_STRIP_LAUNCH(256, 128, 2) else ORV_STRIP_LAUNCH(256, 128, 0) }
#undef ORV_STRIP_LAUNCH
        return orv_check_launch("orv_conv_gemm_bf16");
    }
    const int bn = g->N % 256 == 0 ? 256 : (g->N % 128 == 0 ? 128 : 64);     // wider tiles gather the A lines fewer times
    a.tiles_n = g->N / bn;
    a.tiles_m = (g->M + 255) / 256;
    const int smem = 2 * (256 + bn) * 128;
    hipStream_t st = (hipStream_t)stream;
    const bool simple = c->stride == 1 && !c->ups_s && !c->ups_t && c->Hs == c->H && c->Ws == c->W;
#define ORV_CONV_LAUNCH(BN_, E, S_)                                                                                    \
    {                                                                                                                  \
        static bool done = false;                                                                                      \
        if (!done) { (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<256, BN_, E, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); done = true; } \
        hipLaunchKernelGGL((conv_gemm_kernel<256, BN_, E, S_>), dim3(a.tiles_m * a.tiles_n), dim3(512), smem, st, a, cv); \
    }
#define ORV_CONV_CASE(BN_, E) { if (simple) ORV_CONV_LAUNCH(BN_, E, true) else ORV_CONV_LAUNCH(BN_, E, false) }
    if (bn == 256) { if (g->epilogue == 2) ORV_CONV_CASE(256, 2) else ORV_CONV_CASE(256, 0) }
    else if (bn == 128) { if (g->epilogue == 2) ORV_CONV_CASE(128, 2) else ORV_CONV_CASE(128, 0) }
    else { if (g->epilogue == 2) ORV_CONV_CASE(64, 2) else ORV_CONV_CASE(64, 0) }
#undef ORV_CONV_LAUNCH
#undef ORV_CONV_CASE
    return orv_check_launch("orv_conv_gemm_bf16");
}
