// Flash-style attention forward for gfx950: softmax(q k^T * scale) v, non-causal, unmasked, head_dim 64,
// over the joint text+video sequence (F.scaled_dot_product_attention at orv/models/cogvideox_control.py:256-258).
//
// One workgroup = 8 waves = 256 query rows of one (batch, head); each wave owns 32 query rows.
//   * q and k are read IN PLACE from the packed qkv activation [B*S, 3*H*64] (no head-split copy, :239-241);
//     v comes pre-transposed per head from orv_qkv_prep as vT[b,h,d,pos] so that both MFMA A-operands (K rows and
//     V^T rows) are 128-byte K-contiguous lines, staged with global_load_lds into a swizzled, double-buffered LDS ring
//     (same conflict-free chunk ^ ((row>>1)&7) image as the GEMM).
//   * QK^T is computed swapped (S^T = K . Q^T, v_mfma_f32_32x32x16_bf16): the accumulator column is the query row,
//     so each lane holds 32 scores of ONE query row -> the softmax row max/sum is lane-local plus one exchange with
//     lane^32, and the probabilities re-enter the PV MFMA as its B operand straight from registers.
//   * The MFMA contraction order over keys is free; the accumulator hands lane-half `hi` the keys
//     {0-3,8-11}+4*hi of each 16-key group, so orv_qkv_prep stores V^T with the two middle quads of every 16-key group
//     swapped (pos = key with bits 2 and 3 exchanged).  A plain ds_read_b128 of V^T then lines up with P - no permute.
//   * O^T = V^T . P^T accumulates with the query row on the lane axis as well, so the online-softmax rescale is a
//     per-lane scalar multiply.
#include "common.hpp"
#include <stdlib.h>
#include <math.h>

namespace {

constexpr int KV = 64;            // keys per tile
constexpr int TILE_BYTES = 8192;  // 64 rows x 128 B (K tile, and V^T tile)
constexpr int SLOT_BYTES = 2 * TILE_BYTES;
constexpr float RESCALE_THR = 8.f;  // defer the running-max update while exp2 arguments stay <= 8 (P <= 256)

struct AttnArgs {
    const bf16_t* qkv; long ld;
    const bf16_t* vT;
    bf16_t* out; long ld_out;
    float* lse;
    int B, S, H, s_pad;
    float scale_log2;  // scale * log2(e)
    float scale;
    float shift;       // STATIC kernels: fixed softmax shift in log2 units (>= every score the caller can produce)
    // orv_attention_fwd_bounded_dev: the score bound lives in device memory (training: it changes with every optimizer step and must
    // not cost a device -> host read).  Both softmax forms are launched; each workgroup reads the scalar and the form it does not
    // select returns at once (guard_dev == nullptr: no guard).
    const float* guard_dev;
    float guard_limit;
    // key-split tail of attn_fwd_pp_kernel<true, true> (orv_attention_fwd_bounded_ws): workgroups [0, n_full) are whole items; the
    // items from n_full on are cut into ks key ranges each (workgroup n_full + r * ks + part), whose unnormalised partial
    // outputs meet in ws_o / ws_l; ws_cnt[r] counts the arrivals of item r (zeroed by the launch function).
    float* ws_o; float* ws_l; unsigned* ws_cnt;
    int n_full, ks;
    // orv_attention_fwd_packed: `out` is the P16 layout (include/orv_mi355.h orv_gemm_t) of the [B S, H 64] output, ld_out = H 64 - the A
    // operand of the out-projection GEMM (gemm_d8.hip) without a row-major copy in between.  attn_fwd_pp_kernel<true, false> only.
    int out_packed = 0;
};
// true = this kernel form is the one the device-side bound selects
__device__ __forceinline__ bool attn_guard_selects(const AttnArgs& p, bool static_form) {
    if (!p.guard_dev) return true;
    const float b = *p.guard_dev;
    const bool fits = b > 0.f && b <= p.guard_limit;      // NaN / non-positive: the online form
    return fits == static_form;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// -DORV_SEG_TRACE (tools/attn_seg_trace.sh): per-wave time in the matrix segment, at the barrier behind it, in the vector segment and
// at the barrier behind that, summed over the tiles (s_memrealtime ticks of 10 ns), written through the lse pointer
#ifdef ORV_SEG_TRACE
#define SEG_T(I) { const unsigned long long now_ = wall_clock64(); if ((I) > 0) seg_acc[(I) - 1] += now_ - seg_last; seg_last = now_; }
#define SEG_DECL unsigned long long seg_acc[4] = {0, 0, 0, 0}, seg_last = 0;
#define SEG_DUMP if (p.lse && lane == 0) { unsigned long long* tr_ = (unsigned long long*)p.lse + ((long)blockIdx.x * 8 + wave) * 4; for (int i_ = 0; i_ < 4; ++i_) tr_[i_] = seg_acc[i_]; }
#else
#define SEG_T(I)
#define SEG_DECL
#define SEG_DUMP
#endif
// max over this lane and lane ^ 32 without an LDS round trip
__device__ __forceinline__ float max_with_partner_half(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_with_partner_half(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// 2-stage K/V buffer (32 KiB), fragments read just in time, <= 128 VGPRs so that TWO workgroups (4 waves per SIMD) share a
// CU: with head_dim 64 the loop is VALU-issue bound (one v_exp/add/max/cvt per score against 16 MFMAs per 2048 scores), and
// four waves per SIMD hide that better than deeper per-wave software pipelining (a 4-slot ring with QK one tile ahead and
// all fragments up front measured 5-10 % slower and was removed).
template <bool LAZY, bool FUSED>
__global__ __launch_bounds__(512, 4) void attn_fwd_v1_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * SLOT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    // 1-D grid, head-major items (all query tiles of a (b, h) consecutive) handed out per XCD: K / V^T of a head are
    // streamed into one L2, not eight (PMC: 863 MB fetched per launch against 149 MB of operands before)
    const int nqt = (p.S + 255) / 256;
    const int item = orv_xcd_item(blockIdx.x, gridDim.x);
    const int bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q0 = (item % nqt) * 256 + wave * 32;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds q row (q0+l31), d = ks*16 + hi*8 .. +8 ----
    bf16x8 qf[4];
    {
        const int qr = min(q0 + l31, p.S - 1);
        const bf16_t* qp = p.qkv + (row0 + qr) * p.ld + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    // ---- staging: wave w moves K rows 8w..8w+7 and V^T rows (d) 8w..8w+7 of each tile ----
    const int srow = wave * 8 + (lane >> 3), slot = lane & 7;
    const int schunk = slot ^ ((srow >> 1) & 7);
    const bf16_t* kbase = p.qkv + D + h * 64 + schunk * 8;                                   // + key row * ld
    const bf16_t* vsrc = p.vT + ((long)(b * p.H + h) * 64 + srow) * p.s_pad + schunk * 8;  // + kv0
    auto stage_load = [&](int s, int kv0) {
        const int krow = min(kv0 + srow, p.S - 1);
        glds16(kbase + (row0 + krow) * p.ld, smem + s * SLOT_BYTES + wave * 1024);
        glds16(vsrc + kv0, smem + s * SLOT_BYTES + TILE_BYTES + wave * 1024);
    };

    f32x16 oT[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oT[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;  // running max (raw units; log2 units when FUSED) and this lane's partial row sum
    const float c = p.scale_log2;
    const int row_off = l31 * 128;

    const int nt = (p.S + KV - 1) / KV;
    stage_load(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage_load((t + 1) & 1, (t + 1) * KV);
        const char* sK = smem + (t & 1) * SLOT_BYTES;
        const char* sV = sK + TILE_BYTES;

        // S^T[kb] : rows = keys kb*32.., cols = q
        f32x16 sT[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sT[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(sK + kb * 4096 + row_off + (((ks * 2 + hi) ^ sw) * 16));
                sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sT[kb], 0, 0, 0);
            }
        }
        // tail tile: keys >= S contribute nothing
        if (t == nt - 1 && (p.S & (KV - 1)) != 0) {
            const int kv0 = t * KV;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.S) sT[kb][r] = -INFINITY;
                }
        }
        // online softmax: row max over this lane's 32 scores, combined with the partner half (lane ^ 32)
        float tmax = sT[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sT[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sT[1][r]);
        tmax = max_with_partner_half(tmax);
        const float cc = FUSED ? 1.0f : c;
        if (!LAZY || !__all((tmax - m_run) * cc <= RESCALE_THR)) {   // wave-uniform: everything at the old max rescaled once
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = fast_exp2((m_run - m_new) * cc);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) oT[i][e] *= alpha;
        }
        const float mc = m_run * cc;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = FUSED ? fast_exp2(sT[kb][r] - mc) : fast_exp2(fmaf(sT[kb][r], c, -mc));
                sT[kb][r] = pv;
                psum += pv;
            }
        l_run += psum;

        // O^T[db] += V^T[db rows] . P^T ; k-step kk covers 16 keys, P fragment = 8 consecutive accumulator regs
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { bf16x8 v; uint32_t u[4]; } pf;
            const int kb = kk >> 1, r0 = (kk & 1) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) pf.u[i] = pack2bf(sT[kb][r0 + 2 * i], sT[kb][r0 + 2 * i + 1]);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const bf16x8 vf = *(const bf16x8*)(sV + db * 4096 + row_off + (((kk * 2 + hi) ^ sw) * 16));
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, oT[db], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: normalise, store O[q, h*64 + d] with d = db*32 + (r&3) + 8*(r>>2) + 4*hi ----
    const float l_tot = sum_with_partner_half(l_run);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (q < p.S) {
        bf16_t* op = p.out + (row0 + q) * p.ld_out + h * 64 + hi * 4;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                uint2 o;
                o.x = pack2bf(oT[db][qd * 4 + 0] * inv, oT[db][qd * 4 + 1] * inv);
                o.y = pack2bf(oT[db][qd * 4 + 2] * inv, oT[db][qd * 4 + 3] * inv);
                *(uint2*)(op + db * 32 + qd * 8) = o;
            }
        if (p.lse && hi == 0)
            p.lse[((long)b * p.H + h) * p.S + q] = FUSED ? (m_run + __log2f(l_tot)) * 0.6931471805599453f : m_run * p.scale + __logf(l_tot);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// v2: V is read IN PLACE from the packed projection as well (no V^T copy, no orv_head_transpose launch, 50 MB less traffic per
// layer).  The V tile is staged like K ([64 keys][64 d], 128-byte rows, LDS-DMA) and the PV MFMA's A operand (V^T rows) is
// produced by ds_read_b64_tr_b16, gfx950's transposing LDS read: within a 16-lane group lane r points at 4 consecutive
// d of key k0 + r/4 (d block 4 (r % 4)), and lane i receives V[k0 .. k0+3][d0 + i] - four consecutive keys of its own d
// (semantics measured with tools/probe_tr16.cpp).  Two reads (keys +0..3 and +8..11 from 4 hi) line up with the 8 accumulator
// registers of S^T that feed the B operand, so P still goes from the softmax straight into the MFMA.  Bank conflicts: the 4
// key rows of one read are 128 B apart, i.e. two rows per 64-byte bank quarter; the two 64-byte halves of every key row with
// (key >> 1) & 1 are swapped (on the DMA source address, lane-linear LDS image) so the 4 rows land in 4 different quarters.
// ---------------------------------------------------------------------------------------------------------------
typedef short v4i16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_read_pair(const char* a, const char* b) {
    const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)a);
    const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)b);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// STATIC (fused scale only): the caller GUARANTEES score <= p.shift for every (query, key) - ORV always runs the qk LayerNorm
// (cogvideox_control.py:243-247), so |q . k| <= (8 max|gamma_q| + ||beta_q||)(8 max|gamma_k| + ||beta_k||) is known on the host from
// four 64-vectors - and the softmax uses that bound as its FIXED shift: P = exp2(s - shift) <= 1, no running max, no rescale
// branch, no m_run dependency between tiles; -shift rides in the accumulator init of the QK^T MFMAs, so the 32 v_sub and the
// 21-instruction max tree per 64-key tile are gone (10.9 -> ~7 VALU instructions per MFMA).  fp32 l / O accumulate 2^-2 shift ... 1
// without loss for shift <= 60 (P >= 2^-120 stays a normal number; the entry point falls back to the online kernel above that).
template <bool LAZY, bool FUSED, bool STATIC = false>
__global__ __launch_bounds__(512, 4) void attn_fwd_v2_kernel(const AttnArgs p) {
    static_assert(!STATIC || FUSED, "the fixed-shift softmax is built for the pre-multiplied q only");
    __shared__ __attribute__((aligned(16))) char smem[2 * SLOT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int nqt = (p.S + 255) / 256;
    const int item = orv_xcd_item(blockIdx.x, gridDim.x);
    const int bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q0 = (item % nqt) * 256 + wave * 32;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;

    bf16x8 qf[4];
    {
        const int qr = min(q0 + l31, p.S - 1);
        const bf16_t* qp = p.qkv + (row0 + qr) * p.ld + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    // ---- staging: wave w moves K rows 8w..8w+7 and V rows 8w..8w+7 of each 64-key tile ----
    const int srow = wave * 8 + (lane >> 3), slot = lane & 7;
    const int kchunk = slot ^ ((srow >> 1) & 7);              // K: conflict-free image for ds_read_b128 (as the GEMM)
    const int vchunk = slot ^ (((srow >> 1) & 1) << 2);       // V: 64-byte halves swapped on rows with (key >> 1) & 1
    const bf16_t* kbase = p.qkv + D + h * 64 + kchunk * 8;        // + key row * ld
    const bf16_t* vbase = p.qkv + 2 * D + h * 64 + vchunk * 8;
    auto stage_load = [&](int s, int kv0) {
        const long krow = (row0 + min(kv0 + srow, p.S - 1)) * p.ld;   // keys past S: a valid (finite) row; their P is 0
        glds16(kbase + krow, smem + s * SLOT_BYTES + wave * 1024);
        glds16(vbase + krow, smem + s * SLOT_BYTES + TILE_BYTES + wave * 1024);
    };

    f32x16 oT[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oT[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = p.scale_log2;
    const int row_off = l31 * 128;
    // tr-read lane offsets inside the V tile: key row 4 hi + r / 4, d block g16 * 16 + 4 (r % 4), halves swapped when r >> 3
    const int r16 = lane & 15, g16 = (lane >> 4) & 1;
    const int v_row = (4 * hi + (r16 >> 2)) * 128 + g16 * 32 + (r16 & 3) * 8;
    const int v_off0 = v_row + ((r16 >> 3) << 6);             // d block 0 (d 0..31)
    const int v_off1 = v_row + ((1 - (r16 >> 3)) << 6);       // d block 1 (d 32..63)

    const int nt = (p.S + KV - 1) / KV;
    stage_load(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage_load((t + 1) & 1, (t + 1) * KV);
        // a wave whose 32 query rows all lie past S (the last 256-row block of a head: 3226 = 12 x 256 + 154) only stages K / V
        // and meets the barriers: its MFMA / softmax work would be thrown away, and its issue slots go to the other waves
        if (q0 >= p.S) continue;
        const char* sK = smem + (t & 1) * SLOT_BYTES;
        const char* sV = sK + TILE_BYTES;

        f32x16 sT[2];
        const float s_init = STATIC ? -p.shift : 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sT[kb][e] = s_init;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(sK + kb * 4096 + row_off + (((ks * 2 + hi) ^ sw) * 16));
                sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sT[kb], 0, 0, 0);
            }
        }
        if (t == nt - 1 && (p.S & (KV - 1)) != 0) {
            const int kv0 = t * KV;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.S) sT[kb][r] = -INFINITY;
                }
        }
        float psum = 0.f;
        if constexpr (STATIC) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(sT[kb][r]);          // s - shift <= 0: the accumulator started at -shift
                    sT[kb][r] = pv;
                    psum += pv;
                }
        } else {
            float tmax = sT[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sT[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sT[1][r]);
            tmax = max_with_partner_half(tmax);
            const float cc = FUSED ? 1.0f : c;
            if (!LAZY || !__all((tmax - m_run) * cc <= RESCALE_THR)) {
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = fast_exp2((m_run - m_new) * cc);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oT[i][e] *= alpha;
            }
            const float mc = m_run * cc;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = FUSED ? fast_exp2(sT[kb][r] - mc) : fast_exp2(fmaf(sT[kb][r], c, -mc));
                    sT[kb][r] = pv;
                    psum += pv;
                }
        }
        l_run += psum;

        // O^T[db] += V^T[db rows] . P^T ; k-step kk = keys kk*16 + {0-3, 8-11} + 4 hi (the accumulator order of S^T)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { bf16x8 v; uint32_t u[4]; } pf;
            const int kb = kk >> 1, r0 = (kk & 1) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) pf.u[i] = pack2bf(sT[kb][r0 + 2 * i], sT[kb][r0 + 2 * i + 1]);
            const char* vk = sV + kk * 16 * 128;
            const bf16x8 vf0 = tr_read_pair(vk + v_off0, vk + 8 * 128 + v_off0);
            const bf16x8 vf1 = tr_read_pair(vk + v_off1, vk + 8 * 128 + v_off1);
            oT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf.v, oT[0], 0, 0, 0);
            oT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pf.v, oT[1], 0, 0, 0);
        }
    }

    const float l_tot = sum_with_partner_half(l_run);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (q < p.S) {
        bf16_t* op = p.out + (row0 + q) * p.ld_out + h * 64 + hi * 4;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                uint2 o;
                o.x = pack2bf(oT[db][qd * 4 + 0] * inv, oT[db][qd * 4 + 1] * inv);
                o.y = pack2bf(oT[db][qd * 4 + 2] * inv, oT[db][qd * 4 + 3] * inv);
                *(uint2*)(op + db * 32 + qd * 8) = o;
            }
        if (p.lse && hi == 0)
            p.lse[((long)b * p.H + h) * p.S + q] = STATIC ? (p.shift + __log2f(l_tot)) * 0.6931471805599453f
                                                  : FUSED ? (m_run + __log2f(l_tot)) * 0.6931471805599453f : m_run * p.scale + __logf(l_tot);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// v3 "ping-pong" (fused scale only; V in place): the two waves of every SIMD are kept HALF A TILE out of phase by the barrier
// sequence - while one is in its matrix segment X_t = { PV of tile t-1 ; QK^T of tile t } (16 MFMAs, 8 ds_read_b128, 16 transposing
// reads, no VALU) its partner is in its vector segment Y_t = { exp2, row sum, bf16 pack of tile t } (no MFMA), and vice versa.
// With one s_barrier per segment the workgroup's second half (waves 4-7, the second wave of every SIMD) runs one barrier behind
// the first, so on each SIMD the matrix pipe (512 cycles per tile and wave) and the VALU issue port (~90 instructions per tile and
// wave at head_dim 64) are used by different waves at the same time instead of both waves hitting the same pipe right after a
// common barrier (PMC of v2: MFMA busy 0.40-0.45 although neither pipe is saturated).  Registers stay within 128: S_t lives
// X_t -> Y_t, the packed P_t lives Y_t -> X_{t+1}, no second score set - two workgroups per CU as before.
//   * DMA roles: waves 0-3 stage the K tiles (K_{t+1} issued in their X_t behind the transposing reads, waited at the end of
//     Y_t), waves 4-7 the V tiles (V_{t+1} issued in their Y_t, waited at the end of their X_{t+1}).  Two 8-KiB slots per operand
//     (32 KiB: unchanged).
//   * STATIC (the caller bounds |score| by <= 90 log2 units, see orv_attention_fwd_bounded): P = exp2(s) with NO shift at all - P in
//     [2^-90, 2^90], l <= S 2^90, O <= l max|v| stay normal, finite fp32 / bf16 numbers and a common factor cancels in O / l; otherwise the online
//     softmax with the lazy rescale, taken in Y_t between PV_{t-1} and PV_t, i.e. with every pending product already in O (cdna
//     guide T13 ordering).
//   * epilogue: O rows leave as 16-byte pieces (two 8-byte column groups exchanged between lane and lane ^ 32, guide T21).
// ---------------------------------------------------------------------------------------------------------------
#ifndef ORV_ATTN_M16_DEFAULT
#define ORV_ATTN_M16_DEFAULT 0
#endif
#ifndef ORV_ATTN_W64_DEFAULT
#define ORV_ATTN_W64_DEFAULT 0
#endif
constexpr int PP_SLOTS = 2;        // K / V tiles resident per operand (tile t in slot t % 2, staged one tile ahead).  Measured and
// dropped: three slots with the tiles staged TWO ahead from inline-asm LDS-DMA (so that hipcc's vmcnt(0) in front of every
// ds_read_b64_tr_b16 that follows a DMA builtin cannot drain the stream) and counted vmcnt(2) waits - 0.365-0.373 ms against
// 0.344-0.349 ms standalone, 0.3395 against 0.3275 ms in the model (profiles/r3_attention_pingpong.txt): the loop does not wait
// for the DMA, and the third slot pair costs more than it hides.
template <bool STATIC, bool SPLIT = false>
__global__ __launch_bounds__(512, 4) void attn_fwd_pp_kernel(const AttnArgs p) {
    static_assert(STATIC || !SPLIT, "partial results add only under the shift-free softmax");
    __shared__ __attribute__((aligned(16))) char smem[2 * PP_SLOTS * TILE_BYTES];   // K slots | V slots
    if (!attn_guard_selects(p, STATIC)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int nqt = (p.S + 255) / 256;
    // SPLIT: the grid is n_full whole items (the dispatcher's full rounds of 2 workgroups per CU) + the remaining items cut into ks key
    // ranges each, dispatched LAST: what used to be a nearly empty extra round of full-length workgroups becomes ks-times shorter
    // pieces that fill the slots the earlier rounds free.  part < 0: a whole item.
    int item, part = -1, rest = 0;
    if (SPLIT && (int)blockIdx.x >= p.n_full) {
        const int r_ = blockIdx.x - p.n_full;
        rest = r_ / p.ks; part = r_ % p.ks; item = p.n_full + rest;
    } else item = orv_xcd_item(blockIdx.x, SPLIT ? p.n_full : gridDim.x);
    const int bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q0 = (item % nqt) * 256 + wave * 32;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;
    const bool active = q0 < p.S;          // a wave whose 32 query rows all lie past S only stages tiles and meets the barriers
#ifdef ORV_PP_TRACE
    const unsigned long long trace_t0 = wall_clock64();     // tools/attn_trace.sh: workgroup start / end on the 100 MHz clock
#endif

    bf16x8 qf[4];
    {
        const int qr = min(q0 + l31, p.S - 1);
        const bf16_t* qp = p.qkv + (row0 + qr) * p.ld + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    // ---- staging: the four waves of a half move one operand's 64-row tile, wave wq rows 16 wq .. 16 wq + 15 (two pieces) ----
    // this lane's source line of piece j in tile t: key row t * KV + 16 wq + 8 j + lane / 8 (clamped to S - 1: keys past S read a
    // valid row, their P is forced to 0), 16-byte chunk = slot ^ swizzle(row)
    auto src_of = [&](int j, int t) {
        const int sr = wq * 16 + j * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = grp == 0 ? (slot ^ ((sr >> 1) & 7))                      // K: conflict-free image for ds_read_b128
                                   : (slot ^ (((sr >> 1) & 1) << 2));              // V: 64-byte halves swapped when (key >> 1) & 1
        return p.qkv + (row0 + min(t * KV + sr, p.S - 1)) * p.ld + (grp == 0 ? D : 2 * D) + h * 64 + chunk * 8;
    };
    const bf16_t* const sbase0 = src_of(0, 0);
    const bf16_t* const sbase1 = src_of(1, 0);
    char* const sdst = smem + grp * PP_SLOTS * TILE_BYTES + wq * 2048;
    const int nt = (p.S + KV - 1) / KV;
    const bool ragged = (p.S & (KV - 1)) != 0;
    // key tiles [t_lo, t_hi) of this workgroup (a whole item: all of them)
    const int t_lo = (SPLIT && part >= 0) ? (int)((long)part * nt / p.ks) : 0;
    const int t_hi = (SPLIT && part >= 0) ? (int)((long)(part + 1) * nt / p.ks) : nt;
    // this half's operand tile t -> slot t % 2.  The tile offset is wave-uniform (scalar unit): per tile and piece one 64-bit
    // vector add instead of a 64-bit multiply + clamp (that address arithmetic was ~20 % of the kernel's VALU instructions); only
    // the ragged last tile recomputes its clamped addresses (nothing but the two tile-0 pointers stays live across the loop).
    auto stage = [&](int t) {
        char* const d = sdst + (t % PP_SLOTS) * TILE_BYTES;
        if (__builtin_expect(ragged && t == nt - 1, 0)) {
            glds16(src_of(0, t), d);
            glds16(src_of(1, t), d + 1024);
        } else {
            const long toff = (long)t * KV * p.ld;
            glds16(sbase0 + toff, d);
            glds16(sbase1 + toff, d + 1024);
        }
    };

    f32x16 oT[2], sT[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { oT[i][e] = 0.f; sT[i][e] = 0.f; }
    union { bf16x8 v; uint32_t u[4]; } pf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[i].u[e] = 0u;
    float m_run = -INFINITY, l_run = 0.f;
    const int row_off = l31 * 128;
    const int r16 = lane & 15, g16 = (lane >> 4) & 1;
    const int v_row = (4 * hi + (r16 >> 2)) * 128 + g16 * 32 + (r16 & 3) * 8;
    const int v_off0 = v_row + ((r16 >> 3) << 6);
    const int v_off1 = v_row + ((1 - (r16 >> 3)) << 6);
#define PP_BAR()                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    __builtin_amdgcn_sched_barrier(0);
    // matrix segment: PV of tile t - 1 (t > 0), then QK^T of tile t (t < nt).  Fragment reads run two MFMA pairs ahead of their
    // use and the order is pinned (left alone, hipcc puts every ds_read right in front of its MFMA and waits lgkmcnt(0) each time:
    // ~100 exposed cycles per MFMA in a segment whose SIMD partner is busy on the VALU and cannot fill the pipe).  While PV runs the
    // score registers are dead (32 free for V^T fragments), while QK^T runs P is (16 free for K fragments): peak stays under 128.
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0);
    auto seg_x = [&](int t, auto stage_k) {
#ifndef ORV_PP_NOPRIO
        __builtin_amdgcn_s_setprio(1);      // matrix segment first on the SIMD (+1-3 %, same-process A/B)
#endif
        const char* sK = smem + (t % PP_SLOTS) * TILE_BYTES + row_off;
        auto kread = [&](int kb, int ks) { return *(const bf16x8*)(sK + kb * 4096 + (((ks * 2 + hi) ^ sw) * 16)); };
        bf16x8 k0, k1, k2, k3;
        if (t > t_lo) {
            const char* sV = smem + PP_SLOTS * TILE_BYTES + ((t - 1) % PP_SLOTS) * TILE_BYTES;
            bf16x8 va0 = tr_read_pair(sV + v_off0, sV + 8 * 128 + v_off0), va1 = tr_read_pair(sV + v_off1, sV + 8 * 128 + v_off1);
            bf16x8 vb0 = tr_read_pair(sV + 2048 + v_off0, sV + 2048 + 8 * 128 + v_off0), vb1 = tr_read_pair(sV + 2048 + v_off1, sV + 2048 + 8 * 128 + v_off1);
            PP_FENCE()
            oT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0, pf[0].v, oT[0], 0, 0, 0);
            oT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1, pf[0].v, oT[1], 0, 0, 0);
            PP_FENCE()
            va0 = tr_read_pair(sV + 4096 + v_off0, sV + 4096 + 8 * 128 + v_off0); va1 = tr_read_pair(sV + 4096 + v_off1, sV + 4096 + 8 * 128 + v_off1);
            PP_FENCE()
            oT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb0, pf[1].v, oT[0], 0, 0, 0);
            oT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb1, pf[1].v, oT[1], 0, 0, 0);
            PP_FENCE()
            vb0 = tr_read_pair(sV + 6144 + v_off0, sV + 6144 + 8 * 128 + v_off0); vb1 = tr_read_pair(sV + 6144 + v_off1, sV + 6144 + 8 * 128 + v_off1);
            PP_FENCE()
            // the DMA issue sits BEHIND the last transposing read: hipcc puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16
            // that follows an LDS-DMA in program order (the builtin carries no memory operand it could disambiguate), plain
            // ds_read_b128 are left alone - so issued here the only such wait is next segment's, long after the data landed
            stage_k();
            PP_FENCE()
            if (t < t_hi) { k0 = kread(0, 0); k1 = kread(0, 1); }
            PP_FENCE()
            oT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0, pf[2].v, oT[0], 0, 0, 0);
            oT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1, pf[2].v, oT[1], 0, 0, 0);
            PP_FENCE()
            if (t < t_hi) { k2 = kread(0, 2); k3 = kread(0, 3); }
            PP_FENCE()
            oT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb0, pf[3].v, oT[0], 0, 0, 0);
            oT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb1, pf[3].v, oT[1], 0, 0, 0);
            PP_FENCE()
        } else {
            stage_k();
            PP_FENCE()
            k0 = kread(0, 0); k1 = kread(0, 1); k2 = kread(0, 2); k3 = kread(0, 3);
            PP_FENCE()
        }
        if (t < t_hi) {
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            sT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0], z, 0, 0, 0);
            sT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[1], sT[0], 0, 0, 0);
            PP_FENCE()
            k0 = kread(1, 0); k1 = kread(1, 1);
            PP_FENCE()
            sT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2, qf[2], sT[0], 0, 0, 0);
            sT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k3, qf[3], sT[0], 0, 0, 0);
            PP_FENCE()
            k2 = kread(1, 2); k3 = kread(1, 3);
            PP_FENCE()
            sT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0], z, 0, 0, 0);
            sT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[1], sT[1], 0, 0, 0);
            sT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2, qf[2], sT[1], 0, 0, 0);
            sT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k3, qf[3], sT[1], 0, 0, 0);
            PP_FENCE()
        }
#ifndef ORV_PP_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // vector segment: softmax of tile t -> packed P_t
    auto seg_y = [&](int t) {
        if (t == nt - 1 && (p.S & (KV - 1)) != 0) {
            const int kv0 = t * KV;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.S) sT[kb][r] = -INFINITY;
                }
        }
        float psum = 0.f;
        if constexpr (STATIC) {
#ifdef ORV_PP_PSUM4
            float ps4[4] = {0.f, 0.f, 0.f, 0.f};      // four independent chains instead of one 32-deep dependent add chain
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(sT[kb][r]);
                    sT[kb][r] = pv;
                    ps4[r & 3] += pv;
                }
            psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
#else
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(sT[kb][r]);
                    sT[kb][r] = pv;
                    psum += pv;
                }
#endif
        } else {
            float tmax = sT[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sT[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sT[1][r]);
            tmax = max_with_partner_half(tmax);
            if (!__all(tmax - m_run <= RESCALE_THR)) {
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = fast_exp2(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oT[i][e] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(sT[kb][r] - m_run);
                    sT[kb][r] = pv;
                    psum += pv;
                }
        }
        l_run += psum;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kb = kk >> 1, r0 = (kk & 1) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[kk].u[i] = pack2bf(sT[kb][r0 + 2 * i], sT[kb][r0 + 2 * i + 1]);
        }
    };

    // prologue: K_0 (first half) and V_0 (second half) land before anybody reads.
    // Schedule (I_n = the n-th barrier interval; first half: X_t in I_2t, Y_t in I_2t+1; second half one interval later):
    //   K_{t+1}: first half, in its X_t (I_2t, behind the transposing reads), into the slot of K_{t-1} (last read in I_2t-1);
    //            waited at the end of Y_t, read from I_2t+2
    //   V_{t+1}: second half, in its Y_t (I_2t+2), into the slot of V_{t-1} (last read by PV_{t-1} in I_2t+1); waited at the end of
    //            its X_{t+1} (I_2t+3), read from I_2t+4
    stage(t_lo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BAR()
    const bool act = __builtin_amdgcn_readfirstlane((int)active) != 0;       // provably wave-uniform: real branches, no exec masking
    SEG_DECL
    if (grp == 0) {
        if (act) {
            for (int t = t_lo; t < t_hi; ++t) {
                SEG_T(0)
                seg_x(t, [&]() { if (t + 1 < t_hi) stage(t + 1); });
                SEG_T(1)
                PP_BAR()
                SEG_T(2)
                seg_y(t);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SEG_T(3)   // K_{t+1} landed (this wave's pieces); visible after the barrier
                PP_BAR()
                SEG_T(4)
            }
            seg_x(t_hi, [&]() {});
        } else {
            for (int t = t_lo; t < t_hi; ++t) {
                if (t + 1 < t_hi) stage(t + 1);
                PP_BAR()
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PP_BAR()
            }
        }
        PP_BAR()                                            // the second half runs one barrier behind
    } else {
        PP_BAR()
        if (act) {
            for (int t = t_lo; t < t_hi; ++t) {
                SEG_T(0)
                seg_x(t, [&]() {});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SEG_T(1)   // V_t landed (issued in Y_{t-1} / the prologue)
                PP_BAR()
                SEG_T(2)
                if (t + 1 < t_hi) stage(t + 1);
                __builtin_amdgcn_sched_barrier(0);
                seg_y(t);
                SEG_T(3)
                PP_BAR()
                SEG_T(4)
            }
            seg_x(t_hi, [&]() {});
        } else {
            for (int t = t_lo; t < t_hi; ++t) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PP_BAR()
                if (t + 1 < t_hi) stage(t + 1);
                PP_BAR()
            }
        }
    }
#undef PP_BAR
#undef PP_FENCE

    float l_tot = sum_with_partner_half(l_run);
    const int q = q0 + l31;
    if constexpr (SPLIT) {
        if (part >= 0) {
            // Key-range partial: under the shift-free softmax the unnormalised O^T and l of the ranges simply ADD.  Every part writes
            // its slab with write-through (sc1) stores, drains them, and takes a ticket; the workgroup that draws the last ticket sums
            // the ks slabs in PART ORDER (its own included, read back like the others: the result does not depend on who arrives
            // last) and runs the ordinary epilogue.  Hand-off = cdna_hip_programming.md Guideline 16 (R1, counter form): sc1 payload,
            // vmcnt(0) in every storing wave, barrier, ONE relaxed agent-scope ticket; the reducer reads with sc1 loads (L2, never a
            // stale L1 line) - no fence, placement-independent; the counters are zeroed by a memset node ahead of every launch.
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.ws_o, 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(p.ws_l, 0, 0x7fffffff, 0x00020000);
            const int slab0 = rest * p.ks;                                    // slabs of this item: slab0 .. slab0 + ks - 1
            const int row_b = (wave * 32 + l31) * 256 + hi * 16;              // byte offset of this lane's first d quad inside a slab
            if (act) {
                const int so = (slab0 + part) * (256 * 64 * 4) + row_b;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        u32x4_t v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = __float_as_uint(oT[db][qd * 4 + j]);
                        __builtin_amdgcn_raw_buffer_store_b128(v, ro, so + db * 128 + qd * 32, 0, 16);
                    }
                if (hi == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l_tot), rl, ((slab0 + part) * 256 + wave * 32 + l31) * 4, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* const flag = (unsigned*)smem;                           // the K / V tiles are dead
            if (tid == 0) *flag = __hip_atomic_fetch_add(p.ws_cnt + rest, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*flag != (unsigned)(p.ks - 1)) return;
            if (!act) return;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) oT[i][e] = 0.f;
            l_tot = 0.f;
            for (int pp = 0; pp < p.ks; ++pp) {
                const int so = (slab0 + pp) * (256 * 64 * 4) + row_b;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(ro, so + db * 128 + qd * 32, 0, 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) oT[db][qd * 4 + j] += __uint_as_float(v[j]);
                    }
                l_tot += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, ((slab0 + pp) * 256 + wave * 32 + l31) * 4, 0, 16));
            }
        }
    }
    const float inv = 1.0f / l_tot;
    SEG_DUMP
#ifdef ORV_PP_TRACE
    // timeline build only (never the product library): (start, end, HW_ID | XCC_ID << 32, item) per workgroup through the lse pointer
    if (p.lse && tid == 0) {
        unsigned long long* tr = (unsigned long long*)p.lse + (long)blockIdx.x * 4;
        tr[0] = trace_t0; tr[1] = wall_clock64();
        tr[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
        tr[3] = (unsigned long long)item;
    }
    const bool write_lse = false;
#elif defined(ORV_SEG_TRACE)
    const bool write_lse = false;
#else
    const bool write_lse = true;
#endif
    if (q < p.S) {
        // lane holds d = 32 db + 8 qd + 4 hi + (0..3) of row q: groups qd = 2 u (lower half-wave) and 2 u + 1 (upper) are exchanged
        // so that every lane stores ONE aligned 16-byte piece: lower -> d 32 db + 16 u + 0..7, upper -> + 8..15
        bf16_t* op = p.out + (row0 + q) * p.ld_out + h * 64 + hi * 8;
        // packed output: the piece (row m, columns h 64 + 32 db + 16 u + 8 hi + 0..7) is 16-byte slot (2 u + hi) * 16 + m % 16 of block
        // (m / 16, 2 h + db): consecutive rows = consecutive slots, so the 32 lanes of a half-wave write 2 x 256 contiguous bytes
        const long m_ = row0 + q;
        char* const pblk = (char*)p.out + (((m_ >> 4) * (p.ld_out >> 5) + 2 * h) << 10) + ((hi * 16 + (m_ & 15)) << 4);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint32_t a0 = pack2bf(oT[db][(2 * u) * 4 + 0] * inv, oT[db][(2 * u) * 4 + 1] * inv);
                uint32_t a1 = pack2bf(oT[db][(2 * u) * 4 + 2] * inv, oT[db][(2 * u) * 4 + 3] * inv);
                uint32_t b0 = pack2bf(oT[db][(2 * u + 1) * 4 + 0] * inv, oT[db][(2 * u + 1) * 4 + 1] * inv);
                uint32_t b1 = pack2bf(oT[db][(2 * u + 1) * 4 + 2] * inv, oT[db][(2 * u + 1) * 4 + 3] * inv);
                // a of the upper half-wave <-> b of the lower half-wave
                { const auto r = __builtin_amdgcn_permlane32_swap(a0, b0, false, false); a0 = r[0]; b0 = r[1]; }
                { const auto r = __builtin_amdgcn_permlane32_swap(a1, b1, false, false); a1 = r[0]; b1 = r[1]; }
                if (!SPLIT && p.out_packed) *(uint4*)(pblk + db * 1024 + u * 512) = make_uint4(a0, a1, b0, b1);
                else *(uint4*)(op + db * 32 + u * 16) = make_uint4(a0, a1, b0, b1);
            }
        if (write_lse && p.lse && hi == 0)
            p.lse[((long)b * p.H + h) * p.S + q] = (STATIC ? __log2f(l_tot) : m_run + __log2f(l_tot)) * 0.6931471805599453f;
    }
}



// ---------------------------------------------------------------------------------------------------------------
// "w64" (round 5): 64 query rows per wave.  In attn_fwd_pp_kernel every wave re-reads the whole 16-KB K + V tile for its 32 rows:
// 128 KB of fragment reads per 256-row tile step, a fragment feeds one MFMA, and a matrix segment lasts 2.5-3 x its MFMA time
// (profiles/r3_attention_pingpong.txt).  Here a wave owns TWO 32-row query blocks, so every K fragment (QK^T) and every V^T fragment (PV) feeds
// two MFMAs: half the LDS traffic and half the DMA per score, twice the MFMA time behind every read.  256 registers per wave (O^T 64, S^T 64,
// P 32, Q 32, fragments 32): ONE 8-wave workgroup per CU.  A 512-row item would quantise badly (S = 3226: 7 items per head, the last 30 % full;
// 840 items on 256 slots), so the workgroup is TWO independent 4-wave groups, each with its own 256-row item (own (b, h), own K / V slots:
// 64 KiB of LDS) - the items and the grid quantisation of the 8-wave kernel (1560 items = 780 workgroups on 256 CUs) - and the barrier
// choreography of attn_fwd_pp_kernel between them: group 0 (the first wave of every SIMD) is in its matrix segment X_t = { PV_{t-1} ; QK^T_t }
// while group 1 (the second wave of every SIMD) is in its vector segment Y_t, one barrier apart.  A first build with 4-wave workgroups (the
// SIMD partner in another, unsynchronised workgroup) ran the two waves of a SIMD in phase: 0.400 ms against 0.348 (profiles/r5_attention_w64.txt).
// Fixed-shift softmax only (STATIC of the pp kernel), V in place.  Per group and tile: K_{t+1} is issued at the start of X_t, V_t at the start
// of Y_t (inline-asm LDS-DMA: behind the builtin hipcc drains the stream with vmcnt(0) in front of every transposing read), both waited for
// at the end of Y_t.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long w64_uniform64(unsigned long long u) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void w64_glds(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(d) : "memory", "m0");
}
__global__ __launch_bounds__(512, 2) void attn_fwd_w64_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem_all[2 * 2 * PP_SLOTS * TILE_BYTES];   // group 0: K slots | V slots, group 1: K slots | V slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    char* const smem = smem_all + grp * (2 * PP_SLOTS * TILE_BYTES);
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int nqt = (p.S + 255) / 256;
    const int nitems = nqt * p.H * p.B;
    const int item_ = 2 * orv_xcd_item(blockIdx.x, gridDim.x) + grp;       // the two groups take neighbouring query tiles (mostly one head: K / V shared in L2)
    const bool item_ok = item_ < nitems;
    const int item = min(item_, nitems - 1);
    const int bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q0 = (item % nqt) * 256 + wave * 64;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;
    const bool act = __builtin_amdgcn_readfirstlane((int)(item_ok && q0 < p.S)) != 0;

    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qr = min(q0 + 32 * qb + l31, p.S - 1);
        const bf16_t* qp = p.qkv + (row0 + qr) * p.ld + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const bf16x8*)(qp + ks * 16);
    }
    // ---- staging: wave w moves rows 16 w .. 16 w + 15 (two 1-KiB pieces) of the K tile AND of the V tile.  32-bit per-lane byte offsets
    // against a wave-uniform base that carries the tile position (saddr form: no 64-bit vector address arithmetic in the loop) ----
    auto off_of = [&](int op, int j, int t) {
        const int sr = wave * 16 + j * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = op == 0 ? (slot ^ ((sr >> 1) & 7)) : (slot ^ (((sr >> 1) & 1) << 2));
        const int rr = min(t * KV + sr, p.S - 1) - t * KV;                       // keys past S read the last valid row, their P is forced to 0
        return (unsigned)(((long)rr * p.ld + (op == 0 ? D : 2 * D) + h * 64 + chunk * 8) * 2);
    };
    const unsigned ok0 = off_of(0, 0, 0), ok1 = off_of(0, 1, 0), ov0 = off_of(1, 0, 0), ov1 = off_of(1, 1, 0);
    const int nt = (p.S + KV - 1) / KV;
    const bool ragged = (p.S & (KV - 1)) != 0;
    auto stage = [&](int op, int t) {
        const unsigned d = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + op * PP_SLOTS * TILE_BYTES + (t % PP_SLOTS) * TILE_BYTES + wave * 2048;
        const unsigned long long sb = w64_uniform64((unsigned long long)(uintptr_t)(p.qkv + (row0 + (long)t * KV) * p.ld));
        unsigned o0 = op == 0 ? ok0 : ov0, o1 = op == 0 ? ok1 : ov1;
        if (__builtin_expect(ragged && t == nt - 1, 0)) { o0 = off_of(op, 0, t); o1 = off_of(op, 1, t); }
        w64_glds(o0, sb, d);
        w64_glds(o1, sb, d + 1024);
    };

    f32x16 oT[2][2], sT[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { oT[a][i][e] = 0.f; sT[a][i][e] = 0.f; }
    union { bf16x8 v; uint32_t u[4]; } pf[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[a][i].u[e] = 0u;
    float l_run[2] = {0.f, 0.f};
    const int row_off = l31 * 128;
    const int r16 = lane & 15, g16 = (lane >> 4) & 1;
    const int v_row = (4 * hi + (r16 >> 2)) * 128 + g16 * 32 + (r16 & 3) * 8;
    const int v_off0 = v_row + ((r16 >> 3) << 6);
    const int v_off1 = v_row + ((1 - (r16 >> 3)) << 6);
#define W64_FENCE() __builtin_amdgcn_sched_barrier(0);
#define W64_PV(VA0, VA1, KK)                                                                                              \
    oT[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VA0, pf[0][KK].v, oT[0][0], 0, 0, 0);                              \
    oT[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VA0, pf[1][KK].v, oT[1][0], 0, 0, 0);                              \
    oT[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VA1, pf[0][KK].v, oT[0][1], 0, 0, 0);                              \
    oT[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VA1, pf[1][KK].v, oT[1][1], 0, 0, 0);
#define W64_QK(KF, KS, KB, ACC0, ACC1)                                                                                    \
    sT[0][KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KF, qf[0][KS], ACC0, 0, 0, 0);                                    \
    sT[1][KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KF, qf[1][KS], ACC1, 0, 0, 0);
    // matrix part of iteration t: PV of tile t - 1 (t > 0), QK^T of tile t (t < nt); fragment reads one group ahead of their MFMAs
    auto seg_x = [&](int t) {
        const char* sK = smem + (t % PP_SLOTS) * TILE_BYTES + row_off;
        auto kread = [&](int kb, int ks) { return *(const bf16x8*)(sK + kb * 4096 + (((ks * 2 + hi) ^ sw) * 16)); };
        bf16x8 k0, k1, k2, k3;
        if (t > 0) {
            const char* sV = smem + PP_SLOTS * TILE_BYTES + ((t - 1) % PP_SLOTS) * TILE_BYTES;
            bf16x8 va0 = tr_read_pair(sV + v_off0, sV + 8 * 128 + v_off0), va1 = tr_read_pair(sV + v_off1, sV + 8 * 128 + v_off1);
            bf16x8 vb0 = tr_read_pair(sV + 2048 + v_off0, sV + 2048 + 8 * 128 + v_off0), vb1 = tr_read_pair(sV + 2048 + v_off1, sV + 2048 + 8 * 128 + v_off1);
            W64_FENCE()
            W64_PV(va0, va1, 0)
            W64_FENCE()
            va0 = tr_read_pair(sV + 4096 + v_off0, sV + 4096 + 8 * 128 + v_off0); va1 = tr_read_pair(sV + 4096 + v_off1, sV + 4096 + 8 * 128 + v_off1);
            W64_FENCE()
            W64_PV(vb0, vb1, 1)
            W64_FENCE()
            vb0 = tr_read_pair(sV + 6144 + v_off0, sV + 6144 + 8 * 128 + v_off0); vb1 = tr_read_pair(sV + 6144 + v_off1, sV + 6144 + 8 * 128 + v_off1);
            W64_FENCE()
            W64_PV(va0, va1, 2)
            W64_FENCE()
            if (t < nt) { k0 = kread(0, 0); k1 = kread(0, 1); }
            W64_FENCE()
            W64_PV(vb0, vb1, 3)
            W64_FENCE()
            if (t < nt) { k2 = kread(0, 2); k3 = kread(0, 3); }
            W64_FENCE()
        } else {
            k0 = kread(0, 0); k1 = kread(0, 1); k2 = kread(0, 2); k3 = kread(0, 3);
            W64_FENCE()
        }
        if (t < nt) {
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            W64_QK(k0, 0, 0, z, z)
            W64_QK(k1, 1, 0, sT[0][0], sT[1][0])
            W64_FENCE()
            k0 = kread(1, 0); k1 = kread(1, 1);
            W64_FENCE()
            W64_QK(k2, 2, 0, sT[0][0], sT[1][0])
            W64_QK(k3, 3, 0, sT[0][0], sT[1][0])
            W64_FENCE()
            k2 = kread(1, 2); k3 = kread(1, 3);
            W64_FENCE()
            W64_QK(k0, 0, 1, z, z)
            W64_QK(k1, 1, 1, sT[0][1], sT[1][1])
            W64_QK(k2, 2, 1, sT[0][1], sT[1][1])
            W64_QK(k3, 3, 1, sT[0][1], sT[1][1])
            W64_FENCE()
        }
    };
    auto seg_y = [&](int t) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (t == nt - 1 && ragged) {
                const int kv0 = t * KV;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (key >= p.S) sT[qb][kb][r] = -INFINITY;
                    }
            }
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(sT[qb][kb][r]);
                    sT[qb][kb][r] = pv;
                    psum += pv;
                }
            l_run[qb] += psum;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int kb = kk >> 1, r0 = (kk & 1) * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[qb][kk].u[i] = pack2bf(sT[qb][kb][r0 + 2 * i], sT[qb][kb][r0 + 2 * i + 1]);
            }
        }
    };
#define W64_BAR()                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    __builtin_amdgcn_sched_barrier(0);
    // prologue: K_0 of both groups lands before anybody reads.  Barrier intervals: group 0 runs X_t in I_2t and Y_t in I_2t+1, group 1 one later.
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W64_BAR()
    if (grp == 1) { W64_BAR() }
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) stage(0, t + 1);                          // K_{t+1} -> the slot K_{t-1} left after X_{t-1}
        W64_FENCE()
        seg_x(t);
        W64_BAR()
        stage(1, t);                                              // V_t -> the slot V_{t-2} left after X_{t-1}
        W64_FENCE()
        seg_y(t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of K_{t+1} and V_t
        W64_BAR()
    }
    seg_x(nt);
    if (grp == 0) { W64_BAR() }
#undef W64_BAR

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = sum_with_partner_half(l_run[qb]);
        const int q = q0 + 32 * qb + l31;
        const float inv = 1.0f / l_tot;
        if (act && q < p.S) {
            bf16_t* op = p.out + (row0 + q) * p.ld_out + h * 64 + hi * 8;
            const long m_ = row0 + q;
            char* const pblk = (char*)p.out + (((m_ >> 4) * (p.ld_out >> 5) + 2 * h) << 10) + ((hi * 16 + (m_ & 15)) << 4);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint32_t a0 = pack2bf(oT[qb][db][(2 * u) * 4 + 0] * inv, oT[qb][db][(2 * u) * 4 + 1] * inv);
                    uint32_t a1 = pack2bf(oT[qb][db][(2 * u) * 4 + 2] * inv, oT[qb][db][(2 * u) * 4 + 3] * inv);
                    uint32_t b0 = pack2bf(oT[qb][db][(2 * u + 1) * 4 + 0] * inv, oT[qb][db][(2 * u + 1) * 4 + 1] * inv);
                    uint32_t b1 = pack2bf(oT[qb][db][(2 * u + 1) * 4 + 2] * inv, oT[qb][db][(2 * u + 1) * 4 + 3] * inv);
                    { const auto r = __builtin_amdgcn_permlane32_swap(a0, b0, false, false); a0 = r[0]; b0 = r[1]; }
                    { const auto r = __builtin_amdgcn_permlane32_swap(a1, b1, false, false); a1 = r[0]; b1 = r[1]; }
                    if (p.out_packed) *(uint4*)(pblk + db * 1024 + u * 512) = make_uint4(a0, a1, b0, b1);
                    else *(uint4*)(op + db * 32 + u * 16) = make_uint4(a0, a1, b0, b1);
                }
            if (p.lse && hi == 0) p.lse[((long)b * p.H + h) * p.S + q] = __log2f(l_tot) * 0.6931471805599453f;
        }
    }
#undef W64_FENCE
#undef W64_PV
#undef W64_QK
}

// ---------------------------------------------------------------------------------------------------------------
// The ping-pong kernel on v_mfma_f32_16x16x32_bf16 (fixed-shift softmax only).  Why: with nothing but MFMAs in the loop the part
// sustains 1968 TFLOP/s on random bf16 operands with the 16-wide shape and 1747 with the 32-wide one (profiles/
// r3_probe_mfma_shapes.txt) - 12.6 % more FLOPs per joule on a kernel that runs at the power cap (1.8 GHz).
// Lane l = (c = l & 15, g = l >> 4).  A wave still owns 32 query rows, now as two 16-row blocks qb: row q0 + 16 qb + c.
//   S^T block (kbk, qb) = K[16 kbk .. +15][:] . Q^T: the accumulator hands lane (c, g) the keys 16 kbk + 4 g + e (e < 4) of query
//     (qb, c): 8 blocks x 4 = 32 scores per lane as before, 16 per query row, a row's 64 keys spread over the four lane groups -
//     the row sum is reduced over g ONCE, after the last tile.
//   K fragments (A operand, 16 keys x 32 d): one ds_read_b128 per (kbk, k-step), used by both query blocks.
//   P^T (B operand, 32 keys x 16 queries) of key step s: k-group g carries keys 32 s + 4 g + e and 32 s + 16 + 4 g + e - the two
//     accumulator quads of blocks 2 s and 2 s + 1, packed, straight from the softmax.
//   V^T fragments (A operand, 16 d x 32 keys in THAT order): two ds_read_b64_tr_b16 per (db, s) - group g reads keys 32 s + 4 g .. + 3
//     and 32 s + 16 + 4 g .. + 3 of d block db -, used by both query blocks.  Same 8 + 16 fragment reads per tile as the 32-wide
//     form.  V image: the 32-byte segment of d block db of key row r sits at db ^ ((r >> 1) & 3), so the 8 key rows of half a wave's
//     transposing read fall into 8 different bank groups.
// Everything else (slots, DMA roles, barriers, the one-barrier phase shift of the second half) is attn_fwd_pp_kernel.
// STATUS (profiles/r3_attention_pingpong.txt, box 10-15): same results (same tests); the clock under the kernel rises from 1.71-1.80
// to 1.93-2.03 GHz - the power argument holds; with the row sums on the matrix pipe the tile loop is 6 % shorter than the 32-wide
// kernel's (1.71 vs 1.82 us per tile and wave) and the kernel is level with it standalone (0.356-0.359 vs 0.348-0.353 ms on one box,
// 0.358 vs 0.365 on another), but 4.5 % SLOWER in the model (0.337 vs 0.323 ms), so it stays opt-in (ORV_ATTN_M16=1).
__device__ __forceinline__ void m16_glds16(const char* sbase, unsigned voff, const void* lds_dst) {
    unsigned keep;
    const unsigned long long u = (unsigned long long)(uintptr_t)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const unsigned long long su = ((unsigned long long)hi_ << 32) | lo;
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(su), "s"(d) : "memory");
}
__device__ __forceinline__ float m16_sum_groups(float v) {          // sum over the four lane groups (lanes c, c + 16, c + 32, c + 48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__global__ __launch_bounds__(512, 4) void attn_fwd_m16_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * PP_SLOTS * TILE_BYTES];   // K slots | V slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int c = lane & 15, g = lane >> 4;
    const int nqt = (p.S + 255) / 256;
    const int item = orv_xcd_item(blockIdx.x, gridDim.x);
    const int bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q0 = (item % nqt) * 256 + wave * 32;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;
    const bool active = q0 < p.S;
#ifdef ORV_PP_TRACE
    const unsigned long long trace_t0 = wall_clock64();
#endif

    bf16x8 qf[2][2];                     // [query block][k-step]: d = 32 ks + 8 g .. + 7 of query row q0 + 16 qb + c
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qr = min(q0 + 16 * qb + c, p.S - 1);
        const bf16_t* qp = p.qkv + (row0 + qr) * p.ld + h * 64 + g * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qb][ks] = *(const bf16x8*)(qp + ks * 32);
    }

    // DMA source = wave-uniform tile base (scalar registers) + a 32-bit per-lane byte offset per piece: two VGPRs instead of two
    // 64-bit pointers (those were what the register allocator spilled inside the tile loop, and their reload put an
    // s_waitcnt vmcnt(0) between the two DMA instructions of a tile)
    unsigned loff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sr = wq * 16 + j * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = grp == 0 ? (slot ^ ((sr >> 1) & 7)) : (slot ^ (((sr >> 1) & 3) << 1));
        loff[j] = (unsigned)((sr * (int)p.ld + chunk * 8) * 2);
    }
    const char* const tbase = (const char*)(p.qkv + row0 * p.ld + (grp == 0 ? D : 2 * D) + h * 64);
    char* const sdst = smem + grp * PP_SLOTS * TILE_BYTES + wq * 2048;
    const int nt = (p.S + KV - 1) / KV;
    const bool ragged = (p.S & (KV - 1)) != 0;
    auto stage = [&](int t) {
        char* const d = sdst + (t % PP_SLOTS) * TILE_BYTES;
        if (__builtin_expect(ragged && t == nt - 1, 0)) {
            // keys past S read a valid row (their P is forced to 0): per-lane offsets from the batch base, rows clamped
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int sr = wq * 16 + j * 8 + (lane >> 3), slot = lane & 7;
                const int chunk = grp == 0 ? (slot ^ ((sr >> 1) & 7)) : (slot ^ (((sr >> 1) & 3) << 1));
                m16_glds16(tbase, (unsigned)((min(t * KV + sr, p.S - 1) * (int)p.ld + chunk * 8) * 2), d + j * 1024);
            }
        } else {
            const char* const tb = tbase + (long)t * KV * p.ld * 2;
            m16_glds16(tb, loff[0], d);
            m16_glds16(tb, loff[1], d + 1024);
        }
    };

    f32x4 oT[4][2], sT[4][2];            // O^T [d block][query block], S^T [key block][query block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { oT[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; sT[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    union { bf16x8 v; uint32_t u[4]; } pf[2][2];      // [key step][query block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[i][j].u[e] = 0u;
    float l_run[2] = {0.f, 0.f};
#ifndef ORV_M16_VALUSUM
    // Row sums on the matrix pipe: l^T (qb) += ones[16 x 32] . P^T (key step s, qb) - one extra MFMA per key step and query block
    // (+12.5 % MFMAs) instead of 32 v_add_f32 per tile.  The per-segment timing (tools/attn_seg_trace.sh) shows why: the vector
    // segment, not the matrix segment, is the long pole - the four waves of a SIMD need 4 x (32 quarter-rate v_exp_f32 + ~65 other
    // VALU instructions) = ~3100 VALU cycles per tile against 2048 MFMA cycles.  Every row of the product carries the sum, so
    // register 0 of any lane is the row sum of its query: no cross-lane reduction at the end.
    f32x4 lT[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    union { bf16x8 v; uint32_t u[4]; } ones;
#pragma unroll
    for (int e = 0; e < 4; ++e) ones.u[e] = 0x3f803f80u;
#define M16_LSUM(S_) { lT[0] = M16_MFMA(ones.v, pf[S_][0].v, lT[0]); lT[1] = M16_MFMA(ones.v, pf[S_][1].v, lT[1]); }
#else
#define M16_LSUM(S_)
#endif
    // K fragment of (kbk, ks): row 16 kbk + c, 16-byte chunk 4 ks + g, swizzled by (row >> 1) & 7 = (c >> 1) & 7
    const int k_row = c * 128, k_sw = (c >> 1) & 7;
    // V^T fragment pieces: lane (c, g) addresses key 4 g + (c >> 2) (+ 32 s + 16 half) and d 16 db + 4 (c & 3); segment db ^ fz
    const int v_fz = (2 * g + (c >> 3)) & 3;
    const int v_base = (4 * g + (c >> 2)) * 128 + 8 * (c & 3);
#define M16_BAR()                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define M16_FENCE() __builtin_amdgcn_sched_barrier(0);
#ifdef ORV_M16_ABL_NOMFMA
#define M16_MFMA(A, B, C) (C)
#else
#define M16_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
#endif
    auto seg_x = [&](int t, auto stage_k) {
        __builtin_amdgcn_s_setprio(1);
        const char* sK = smem + (t % PP_SLOTS) * TILE_BYTES + k_row;
        auto kread = [&](int kbk, int ks) {
#ifdef ORV_M16_ABL_NOK
            return qf[kbk & 1][ks];
#endif
            return *(const bf16x8*)(sK + kbk * 2048 + (((4 * ks + g) ^ k_sw) * 16)); };
        bf16x8 ka, kb_, kc, kd, ke, kf, kg, kh;          // K fragments: (key block 0..3, k-step 0) | (key block 0..3, k-step 1)
        if (t > 0) {
            const char* sV = smem + PP_SLOTS * TILE_BYTES + ((t - 1) % PP_SLOTS) * TILE_BYTES + v_base;
            auto vread = [&](int db, int s_) {
#ifdef ORV_M16_ABL_NOV
                return qf[db & 1][s_];
#endif
                const char* a = sV + s_ * 4096 + ((db ^ v_fz) * 32);
                return tr_read_pair(a, a + 2048);
            };
            bf16x8 v0 = vread(0, 0), v1 = vread(1, 0), v2 = vread(2, 0), v3 = vread(3, 0);
            M16_FENCE()
            oT[0][0] = M16_MFMA(v0, pf[0][0].v, oT[0][0]); oT[0][1] = M16_MFMA(v0, pf[0][1].v, oT[0][1]);
            oT[1][0] = M16_MFMA(v1, pf[0][0].v, oT[1][0]); oT[1][1] = M16_MFMA(v1, pf[0][1].v, oT[1][1]);
            M16_LSUM(0)
            M16_FENCE()
            v0 = vread(0, 1); v1 = vread(1, 1);
            M16_FENCE()
            oT[2][0] = M16_MFMA(v2, pf[0][0].v, oT[2][0]); oT[2][1] = M16_MFMA(v2, pf[0][1].v, oT[2][1]);
            oT[3][0] = M16_MFMA(v3, pf[0][0].v, oT[3][0]); oT[3][1] = M16_MFMA(v3, pf[0][1].v, oT[3][1]);
            M16_FENCE()
            v2 = vread(2, 1); v3 = vread(3, 1);
            M16_FENCE()
            stage_k();                                   // behind the last transposing read (see attn_fwd_pp_kernel)
            M16_FENCE()
            if (t < nt) { ka = kread(0, 0); kb_ = kread(1, 0); kc = kread(2, 0); kd = kread(3, 0); }
            M16_FENCE()
            oT[0][0] = M16_MFMA(v0, pf[1][0].v, oT[0][0]); oT[0][1] = M16_MFMA(v0, pf[1][1].v, oT[0][1]);
            oT[1][0] = M16_MFMA(v1, pf[1][0].v, oT[1][0]); oT[1][1] = M16_MFMA(v1, pf[1][1].v, oT[1][1]);
            M16_LSUM(1)
            M16_FENCE()
            if (t < nt) { ke = kread(0, 1); kf = kread(1, 1); }      // the dying V fragments / P make room for the second k-step
            M16_FENCE()
            oT[2][0] = M16_MFMA(v2, pf[1][0].v, oT[2][0]); oT[2][1] = M16_MFMA(v2, pf[1][1].v, oT[2][1]);
            oT[3][0] = M16_MFMA(v3, pf[1][0].v, oT[3][0]); oT[3][1] = M16_MFMA(v3, pf[1][1].v, oT[3][1]);
            M16_FENCE()
            if (t < nt) { kg = kread(2, 1); kh = kread(3, 1); }
            M16_FENCE()
        } else {
            stage_k();
            M16_FENCE()
            ka = kread(0, 0); kb_ = kread(1, 0); kc = kread(2, 0); kd = kread(3, 0);
            ke = kread(0, 1); kf = kread(1, 1); kg = kread(2, 1); kh = kread(3, 1);
            M16_FENCE()
        }
        if (t < nt) {
            // every K fragment of the tile is in flight before the first score MFMA: 16 MFMAs back to back
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            sT[0][0] = M16_MFMA(ka, qf[0][0], z); sT[0][1] = M16_MFMA(ka, qf[1][0], z);
            sT[1][0] = M16_MFMA(kb_, qf[0][0], z); sT[1][1] = M16_MFMA(kb_, qf[1][0], z);
            sT[2][0] = M16_MFMA(kc, qf[0][0], z); sT[2][1] = M16_MFMA(kc, qf[1][0], z);
            sT[3][0] = M16_MFMA(kd, qf[0][0], z); sT[3][1] = M16_MFMA(kd, qf[1][0], z);
            M16_FENCE()
            sT[0][0] = M16_MFMA(ke, qf[0][1], sT[0][0]); sT[0][1] = M16_MFMA(ke, qf[1][1], sT[0][1]);
            sT[1][0] = M16_MFMA(kf, qf[0][1], sT[1][0]); sT[1][1] = M16_MFMA(kf, qf[1][1], sT[1][1]);
            sT[2][0] = M16_MFMA(kg, qf[0][1], sT[2][0]); sT[2][1] = M16_MFMA(kg, qf[1][1], sT[2][1]);
            sT[3][0] = M16_MFMA(kh, qf[0][1], sT[3][0]); sT[3][1] = M16_MFMA(kh, qf[1][1], sT[3][1]);
            M16_FENCE()
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto seg_y = [&](int t) {
        if (t == nt - 1 && (p.S & (KV - 1)) != 0) {
            int kv0 = t * KV + 4 * g;
            asm volatile("" : "+v"(kv0));
#pragma unroll
            for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (kv0 + 16 * kbk + e >= p.S) { sT[kbk][0][e] = -INFINITY; sT[kbk][1][e] = -INFINITY; }
        }
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#ifdef ORV_M16_ABL_NOEXP
                const float a = sT[kbk][0][e], b_ = sT[kbk][1][e];
#else
                const float a = fast_exp2(sT[kbk][0][e]), b_ = fast_exp2(sT[kbk][1][e]);
#endif
                sT[kbk][0][e] = a; sT[kbk][1][e] = b_;
#ifdef ORV_M16_VALUSUM
                ps0 += a; ps1 += b_;
#endif
            }
        l_run[0] += ps0; l_run[1] += ps1;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                pf[s_][qb].u[0] = pack2bf(sT[2 * s_][qb][0], sT[2 * s_][qb][1]);
                pf[s_][qb].u[1] = pack2bf(sT[2 * s_][qb][2], sT[2 * s_][qb][3]);
                pf[s_][qb].u[2] = pack2bf(sT[2 * s_ + 1][qb][0], sT[2 * s_ + 1][qb][1]);
                pf[s_][qb].u[3] = pack2bf(sT[2 * s_ + 1][qb][2], sT[2 * s_ + 1][qb][3]);
            }
    };

    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    M16_BAR()
    const bool act = __builtin_amdgcn_readfirstlane((int)active) != 0;
    SEG_DECL
    if (grp == 0) {
        if (act) {
            for (int t = 0; t < nt; ++t) {
                SEG_T(0)
                seg_x(t, [&]() { if (t + 1 < nt) stage(t + 1); });
                SEG_T(1)
                M16_BAR()
                SEG_T(2)
                seg_y(t);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SEG_T(3)
                M16_BAR()
                SEG_T(4)
            }
            seg_x(nt, [&]() {});
        } else {
            for (int t = 0; t < nt; ++t) {
                if (t + 1 < nt) stage(t + 1);
                M16_BAR()
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                M16_BAR()
            }
        }
        M16_BAR()
    } else {
        M16_BAR()
        if (act) {
            for (int t = 0; t < nt; ++t) {
                SEG_T(0)
                seg_x(t, [&]() {});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SEG_T(1)
                M16_BAR()
                SEG_T(2)
                if (t + 1 < nt) stage(t + 1);
                __builtin_amdgcn_sched_barrier(0);
                seg_y(t);
                SEG_T(3)
                M16_BAR()
                SEG_T(4)
            }
            seg_x(nt, [&]() {});
        } else {
            for (int t = 0; t < nt; ++t) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                M16_BAR()
                if (t + 1 < nt) stage(t + 1);
                M16_BAR()
            }
        }
    }
#undef M16_BAR
#undef M16_FENCE
#undef M16_MFMA

    SEG_DUMP
#ifdef ORV_PP_TRACE
    if (p.lse && tid == 0) {
        unsigned long long* tr = (unsigned long long*)p.lse + (long)blockIdx.x * 4;
        tr[0] = trace_t0; tr[1] = wall_clock64();
        tr[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
        tr[3] = (unsigned long long)item;
    }
#endif
    // epilogue: lane (c, g) holds O^T[16 db + 4 g + e][query (qb, c)]; an even group keeps its d quad of db = 0 / 2 and takes the odd
    // partner's (lane ^ 16), the odd group the other way round for db = 1 / 3: every lane stores 8 consecutive d = 16 bytes
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#ifdef ORV_M16_VALUSUM
        const float l_tot = m16_sum_groups(l_run[qb]);
#else
        const float l_tot = lT[qb][0];
#endif
        const float inv = 1.0f / l_tot;
        const int q = q0 + 16 * qb + c;
        bf16_t* op = p.out + (row0 + min(q, p.S - 1)) * p.ld_out + h * 64;
#pragma unroll
        for (int dp = 0; dp < 2; ++dp) {          // d block pair (2 dp, 2 dp + 1)
            uint32_t x0 = pack2bf(oT[2 * dp][qb][0] * inv, oT[2 * dp][qb][1] * inv), x1 = pack2bf(oT[2 * dp][qb][2] * inv, oT[2 * dp][qb][3] * inv);
            uint32_t y0 = pack2bf(oT[2 * dp + 1][qb][0] * inv, oT[2 * dp + 1][qb][1] * inv), y1 = pack2bf(oT[2 * dp + 1][qb][2] * inv, oT[2 * dp + 1][qb][3] * inv);
            // v_permlane16_swap: lanes 16-31 (48-63) of the first operand <-> lanes 0-15 (32-47) of the second
            { const auto r = __builtin_amdgcn_permlane16_swap(x0, y0, false, false); x0 = r[0]; y0 = r[1]; }
            { const auto r = __builtin_amdgcn_permlane16_swap(x1, y1, false, false); x1 = r[0]; y1 = r[1]; }
            // even group: x = own quad of block 2 dp, y = partner's quad of block 2 dp        -> d 32 dp + 4 g .. + 7
            // odd group:  x = partner's quad of block 2 dp + 1, y = own quad of block 2 dp + 1 -> d 32 dp + 16 + 4 (g - 1) .. + 7
            const int d0 = 32 * dp + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
            if (q < p.S) {
                if (p.out_packed) {     // P16 layout: 16-byte slot (d0 % 32 / 8) * 16 + m % 16 of block (m / 16, 2 h + dp)
                    const long m_ = row0 + q;
                    *(uint4*)((char*)p.out + (((m_ >> 4) * (p.ld_out >> 5) + 2 * h + dp) << 10) + (((((d0 & 31) >> 3) << 4) + (m_ & 15)) << 4)) = make_uint4(x0, x1, y0, y1);
                } else *(uint4*)(op + d0) = make_uint4(x0, x1, y0, y1);
            }
        }
#if !defined(ORV_SEG_TRACE) && !defined(ORV_PP_TRACE)
        if (p.lse && g == 0 && q < p.S) p.lse[((long)b * p.H + h) * p.S + q] = __log2f(l_tot) * 0.6931471805599453f;
#endif
    }
}

}  // namespace

extern "C" int orv_attention_fwd(const void* qkv, int ld_qkv, const void* vT, void* out, int ld_out, float* lse, int B,
                                 int S, int H, int s_pad, float scale, void* stream) {
    ORV_REQUIRE(qkv && out, "orv_attention_fwd: null operand");
    ORV_REQUIRE(B > 0 && S > 0 && H > 0, "orv_attention_fwd: empty problem");
    ORV_REQUIRE(!vT || (s_pad % 64 == 0 && s_pad >= S), "orv_attention_fwd: s_pad=%d must be a multiple of 64 and >= S=%d", s_pad, S);
    ORV_REQUIRE(ld_qkv % 8 == 0 && ld_out % 4 == 0, "orv_attention_fwd: misaligned leading dimension");
    AttnArgs a;
    a.qkv = (const bf16_t*)qkv; a.ld = ld_qkv; a.vT = (const bf16_t*)vT; a.out = (bf16_t*)out; a.ld_out = ld_out;
    a.lse = lse; a.B = B; a.S = S; a.H = H; a.s_pad = s_pad;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f; a.shift = 0.f; a.guard_dev = nullptr; a.guard_limit = 0.f; a.ws_o = a.ws_l = nullptr; a.ws_cnt = nullptr; a.n_full = a.ks = 0;
    dim3 grid(((S + 255) / 256) * H * B);
    // q pre-multiplied by scale*log2(e) in orv_qkv_prep (q_premul) arrives here as scale == 1/log2(e): fused fast path
    const bool fused = fabsf(a.scale_log2 - 1.0f) < 1e-6f;
    hipStream_t st = (hipStream_t)stream;
    static int use_pp = -1;          // ORV_ATTN_PP=0: A/B switch (v2 instead of the ping-pong kernel)
    if (use_pp < 0) { const char* e = getenv("ORV_ATTN_PP"); use_pp = (e && atoi(e) == 0) ? 0 : 1; }
    if (!vT) {                       // V read in place from the packed projection (transposing LDS reads): the shipped path
        if (fused && use_pp && ld_out % 8 == 0 && ((uintptr_t)out & 15) == 0)
            hipLaunchKernelGGL((attn_fwd_pp_kernel<false>), grid, dim3(512), 0, st, a);
        else if (fused) hipLaunchKernelGGL((attn_fwd_v2_kernel<true, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_v2_kernel<true, false>), grid, dim3(512), 0, st, a);
    } else if (fused) hipLaunchKernelGGL((attn_fwd_v1_kernel<true, true>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((attn_fwd_v1_kernel<true, false>), grid, dim3(512), 0, st, a);
    return orv_check_launch("orv_attention_fwd");
}

// The same attention when the caller can BOUND the scores: |q . k| * scale * log2(e) <= score_bound for every (query, key) of the
// call (ORV: from the qk-LayerNorm affine parameters, orv_amd/cogvideox_control.py `score_bound`).  With the fused scale
// (q pre-multiplied, scale == 1 / log2 e) the softmax then needs no running max:
//   * attn_fwd_pp_kernel<STATIC> (16-byte aligned output) computes P = exp2(s) with NO shift at all, so P lies in
//     [2^-bound, 2^bound]: l <= S 2^bound and O <= l max|v| stay finite and every P stays a normal number for bound <= 90
//     (2^90 x 2^14 keys x 2^10 = 2^114 < 2^127; 2^-90 > 2^-126) - ORV_STATIC_LIMIT_PP.  At gamma = 1, beta = 0 the bound is
//     11.8, so the limit admits max|gamma_q| max|gamma_k| up to ~7.6 (40 was exceeded from ~3.4).
//   * attn_fwd_v2_kernel<.., STATIC> (unaligned outputs) shifts by the bound, P in [2^(-2 bound), 1]: valid to bound <= 60.
// Above the limit this is orv_attention_fwd (online softmax).  A bound that does not hold gives wrong results and, far enough
// off, inf: the caller owns the guarantee (tests/test_gpu_kernels.py exercises a violated bound to show it is a contract).
constexpr float ORV_STATIC_LIMIT_PP = 90.f, ORV_STATIC_LIMIT_V2 = 60.f;
// ORV_ATTN_W64: 1 = the 64-rows-per-wave kernel for the shift-free softmax, 0 = the 8-wave ping-pong kernel (A/B switch; default below)
static bool attn_use_w64() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ORV_ATTN_W64"); v = e ? (atoi(e) != 0) : ORV_ATTN_W64_DEFAULT; }
    return v != 0;
}
extern "C" int orv_attention_fwd_bounded(const void* qkv, int ld_qkv, void* out, int ld_out, float* lse, int B, int S, int H,
                                         float scale, float score_bound, void* stream) {
    const float scale_log2 = scale * 1.4426950408889634f;
    const bool fused = fabsf(scale_log2 - 1.0f) < 1e-6f;
    static int use_static = -1;      // ORV_ATTN_STATIC=0: A/B switch
    if (use_static < 0) { const char* e = getenv("ORV_ATTN_STATIC"); use_static = (e && atoi(e) == 0) ? 0 : 1; }
    static int use_pp_ = -1;
    if (use_pp_ < 0) { const char* e = getenv("ORV_ATTN_PP"); use_pp_ = (e && atoi(e) == 0) ? 0 : 1; }
    const bool aligned_out = ld_out % 8 == 0 && ((uintptr_t)out & 15) == 0;
    const float limit = (use_pp_ && aligned_out) ? ORV_STATIC_LIMIT_PP : ORV_STATIC_LIMIT_V2;
    if (!use_static || !fused || !(score_bound > 0.f) || score_bound > limit)
        return orv_attention_fwd(qkv, ld_qkv, nullptr, out, ld_out, lse, B, S, H, 0, scale, stream);
    ORV_REQUIRE(qkv && out, "orv_attention_fwd_bounded: null operand");
    ORV_REQUIRE(B > 0 && S > 0 && H > 0, "orv_attention_fwd_bounded: empty problem");
    ORV_REQUIRE(ld_qkv % 8 == 0 && ld_out % 4 == 0, "orv_attention_fwd_bounded: misaligned leading dimension");
    AttnArgs a;
    a.qkv = (const bf16_t*)qkv; a.ld = ld_qkv; a.vT = nullptr; a.out = (bf16_t*)out; a.ld_out = ld_out;
    a.lse = lse; a.B = B; a.S = S; a.H = H; a.s_pad = 0;
    a.scale = scale; a.scale_log2 = scale_log2; a.shift = score_bound; a.guard_dev = nullptr; a.guard_limit = 0.f; a.ws_o = a.ws_l = nullptr; a.ws_cnt = nullptr; a.n_full = a.ks = 0;
    dim3 grid(((S + 255) / 256) * H * B);
    static int use_pp = -1;
    if (use_pp < 0) { const char* e = getenv("ORV_ATTN_PP"); use_pp = (e && atoi(e) == 0) ? 0 : 1; }
    static int use_m16 = -1;         // ORV_ATTN_M16=1: the 16x16x32 form of the ping-pong kernel (A/B switch)
    if (use_m16 < 0) { const char* e = getenv("ORV_ATTN_M16"); use_m16 = e ? (atoi(e) != 0) : ORV_ATTN_M16_DEFAULT; }
    if (use_pp && attn_use_w64() && ld_out % 8 == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(attn_fwd_w64_kernel, dim3((grid.x + 1) / 2), dim3(512), 0, (hipStream_t)stream, a);
    else if (use_pp && use_m16 && ld_out % 8 == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(attn_fwd_m16_kernel, grid, dim3(512), 0, (hipStream_t)stream, a);
    else if (use_pp && ld_out % 8 == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL((attn_fwd_pp_kernel<true>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((attn_fwd_v2_kernel<true, true, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
    return orv_check_launch("orv_attention_fwd_bounded");
}

// orv_attention_fwd_bounded with the output in the packed P16 layout (orv_gemm_t.a_packed of the out-projection, cogvideox_control.py:263).
// Only the shift-free ping-pong kernel writes it: the caller checks 0 < score_bound <= orv_attention_static_limit(1) first (else: row-major
// output + the row-major GEMM).  `out` holds orv_packed_rows(B S) x (H 64) bf16.
extern "C" int orv_attention_fwd_packed(const void* qkv, int ld_qkv, void* out, float* lse, int B, int S, int H, float scale, float score_bound,
                                        void* stream) {
    const float scale_log2 = scale * 1.4426950408889634f;
    ORV_REQUIRE(qkv && out, "orv_attention_fwd_packed: null operand");
    ORV_REQUIRE(B > 0 && S > 0 && H > 0, "orv_attention_fwd_packed: empty problem");
    ORV_REQUIRE(ld_qkv % 8 == 0 && ((uintptr_t)out & 15) == 0, "orv_attention_fwd_packed: misaligned operand");
    ORV_REQUIRE(fabsf(scale_log2 - 1.0f) < 1e-6f && score_bound > 0.f && score_bound <= ORV_STATIC_LIMIT_PP,
                "orv_attention_fwd_packed: needs the fused scale (scale * log2 e == 1) and 0 < score_bound <= %g (got scale %g bound %g)",
                (double)ORV_STATIC_LIMIT_PP, (double)scale, (double)score_bound);
    AttnArgs a;
    a.qkv = (const bf16_t*)qkv; a.ld = ld_qkv; a.vT = nullptr; a.out = (bf16_t*)out; a.ld_out = (long)H * 64;
    a.lse = lse; a.B = B; a.S = S; a.H = H; a.s_pad = 0;
    a.scale = scale; a.scale_log2 = scale_log2; a.shift = score_bound; a.guard_dev = nullptr; a.guard_limit = 0.f; a.ws_o = a.ws_l = nullptr; a.ws_cnt = nullptr; a.n_full = a.ks = 0;
    a.out_packed = 1;
    static int use_m16p = -1;
    if (use_m16p < 0) { const char* e = getenv("ORV_ATTN_M16"); use_m16p = e ? (atoi(e) != 0) : ORV_ATTN_M16_DEFAULT; }
    if (attn_use_w64()) hipLaunchKernelGGL(attn_fwd_w64_kernel, dim3((((S + 255) / 256) * H * B + 1) / 2), dim3(512), 0, (hipStream_t)stream, a);
    else if (use_m16p) hipLaunchKernelGGL(attn_fwd_m16_kernel, dim3(((S + 255) / 256) * H * B), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_pp_kernel<true>), dim3(((S + 255) / 256) * H * B), dim3(512), 0, (hipStream_t)stream, a);
    return orv_check_launch("orv_attention_fwd_packed");
}

// Key-split tail (orv_attention_fwd_bounded_ws).  The grid of the ping-pong kernel is ceil(S / 256) H B workgroups on 2 x CUs slots: at
// the headline shape 1560 on 512 = 3.05 rounds, and the dispatcher's fourth "round" of 24 full-length workgroups costs ~10 % of the
// launch (profiles/r3_attention_pingpong.txt: span 3.79 workgroup lifetimes against 3.43 for an exact 3-round shape).  Plan: the
// items beyond the last full round - when they are few - are cut into ks key ranges each and dispatched last.
struct AttnSplit { int n_full, rest, ks; };
static AttnSplit attn_split_plan(int B, int S, int H) {
    AttnSplit sp{0, 0, 0};
    // OPT-IN (ORV_ATTN_SPLIT=1).  Measured (profiles/r4_attention_key_split.txt): standalone 0.393 -> 0.364 ms per launch, but inside
    // the graph-replayed step nothing (0.3215 vs 0.3231 ms, 41.1 ms / step either way, three interleaved rounds on one box) - and the
    // split items sum their key ranges in another fp32 order, which a 30-block random-init model amplifies from 3e-5 per call to
    // 1.1e-2 at the output, where the unsplit kernels make B = 4 bit-identical to four B = 1 calls.  Off by default.
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("ORV_ATTN_SPLIT"); enabled = (e && atoi(e) != 0) ? 1 : 0; }
    if (!enabled) return sp;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    static int slots_env = -1;       // ORV_ATTN_SLOTS: pretend the chip has this many workgroup slots (tests of the split path at small shapes)
    if (slots_env < 0) { const char* e = getenv("ORV_ATTN_SLOTS"); slots_env = e ? atoi(e) : 0; }
    const int slots = slots_env > 0 ? slots_env : 2 * cus, items = ((S + 255) / 256) * H * B, nt = (S + KV - 1) / KV;
    const int full = items / slots * slots, rest = items - full;
    if (full == 0 || rest == 0 || rest * 4 > slots) return sp;          // nothing to cut, or the last round is not small
    int ks = slots / (2 * rest);
    if (ks > 8) ks = 8;
    while (ks > 1 && nt < 4 * ks) --ks;                                  // every part keeps at least 4 key tiles
    if (ks < 2) return sp;
    sp.n_full = full; sp.rest = rest; sp.ks = ks;
    return sp;
}
// bytes of workspace orv_attention_fwd_bounded_ws wants for this shape (0: the shape is not split)
extern "C" size_t orv_attention_ws_bytes(int B, int S, int H) {
    const AttnSplit sp = attn_split_plan(B, S, H);
    if (!sp.ks) return 0;
    return 1024 + (size_t)sp.rest * sp.ks * (256 * 64 + 256) * sizeof(float);
}

// largest score bound (log2 units) the fixed-shift softmax of orv_attention_fwd_bounded(_dev) accepts: 16-byte aligned output
// (ping-pong kernel) / any output (v2 kernel)
extern "C" float orv_attention_static_limit(int aligned_out) { return aligned_out ? ORV_STATIC_LIMIT_PP : ORV_STATIC_LIMIT_V2; }

// orv_attention_fwd_bounded with a caller-provided workspace (>= orv_attention_ws_bytes(B, S, H), 256-byte aligned; reusable across
// calls on one stream): when the fixed-shift ping-pong kernel applies and the shape has a small last round, that round is key-split
// (above).  Without workspace, or where the plan does not apply, this IS orv_attention_fwd_bounded.  Results are deterministic
// (fixed part order) but differ in fp32 summation order from the unsplit kernel for the split items.
extern "C" int orv_attention_fwd_bounded_ws(const void* qkv, int ld_qkv, void* out, int ld_out, float* lse, int B, int S, int H,
                                            float scale, float score_bound, void* ws, size_t ws_bytes, void* stream) {
    const float scale_log2 = scale * 1.4426950408889634f;
    const bool fused = fabsf(scale_log2 - 1.0f) < 1e-6f;
    static int use_static = -1, use_pp = -1, use_m16 = -1;
    if (use_static < 0) { const char* e = getenv("ORV_ATTN_STATIC"); use_static = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_pp < 0) { const char* e = getenv("ORV_ATTN_PP"); use_pp = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_m16 < 0) { const char* e = getenv("ORV_ATTN_M16"); use_m16 = e ? (atoi(e) != 0) : ORV_ATTN_M16_DEFAULT; }
    const AttnSplit sp = (ws && B > 0 && S > 0 && H > 0) ? attn_split_plan(B, S, H) : AttnSplit{0, 0, 0};
    if (!sp.ks || !use_static || !use_pp || use_m16 || !fused || !(score_bound > 0.f) || score_bound > ORV_STATIC_LIMIT_PP ||
        ld_out % 8 != 0 || ((uintptr_t)out & 15) != 0 || ((uintptr_t)ws & 255) != 0 || ws_bytes < orv_attention_ws_bytes(B, S, H))
        return orv_attention_fwd_bounded(qkv, ld_qkv, out, ld_out, lse, B, S, H, scale, score_bound, stream);
    ORV_REQUIRE(qkv && out, "orv_attention_fwd_bounded_ws: null operand");
    ORV_REQUIRE(ld_qkv % 8 == 0, "orv_attention_fwd_bounded_ws: misaligned leading dimension");
    AttnArgs a;
    a.qkv = (const bf16_t*)qkv; a.ld = ld_qkv; a.vT = nullptr; a.out = (bf16_t*)out; a.ld_out = ld_out;
    a.lse = lse; a.B = B; a.S = S; a.H = H; a.s_pad = 0;
    a.scale = scale; a.scale_log2 = scale_log2; a.shift = 0.f; a.guard_dev = nullptr; a.guard_limit = 0.f;
    a.ws_cnt = (unsigned*)ws;
    a.ws_o = (float*)((char*)ws + 1024);
    a.ws_l = a.ws_o + (size_t)sp.rest * sp.ks * 256 * 64;
    a.n_full = sp.n_full; a.ks = sp.ks;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, 1024, st) != hipSuccess) { orv_set_error("orv_attention_fwd_bounded_ws: hipMemsetAsync failed"); return ORV_EDEVICE; }
    hipLaunchKernelGGL((attn_fwd_pp_kernel<true, true>), dim3(sp.n_full + sp.rest * sp.ks), dim3(512), 0, st, a);
    return orv_check_launch("orv_attention_fwd_bounded_ws");
}

// orv_attention_fwd_bounded with the bound in DEVICE memory (one fp32, e.g. element l of the per-layer bound vector the training
// step recomputes on the device after every optimizer update): no device -> host read.  The fixed-shift and the online form of
// the ping-pong kernel are both launched and every workgroup of the form the scalar does not select returns at once (~2 us of an
// empty grid per call).  Needs the fused scale and a 16-byte aligned output (else: orv_attention_fwd, online softmax).
extern "C" int orv_attention_fwd_bounded_dev(const void* qkv, int ld_qkv, void* out, int ld_out, float* lse, int B, int S, int H,
                                             float scale, const float* score_bound_dev, void* stream) {
    ORV_REQUIRE(qkv && out && score_bound_dev, "orv_attention_fwd_bounded_dev: null operand");
    ORV_REQUIRE(B > 0 && S > 0 && H > 0, "orv_attention_fwd_bounded_dev: empty problem");
    ORV_REQUIRE(ld_qkv % 8 == 0 && ld_out % 4 == 0, "orv_attention_fwd_bounded_dev: misaligned leading dimension");
    const float scale_log2 = scale * 1.4426950408889634f;
    const bool fused = fabsf(scale_log2 - 1.0f) < 1e-6f;
    static int use_static = -1, use_pp = -1;
    if (use_static < 0) { const char* e = getenv("ORV_ATTN_STATIC"); use_static = (e && atoi(e) == 0) ? 0 : 1; }
    if (use_pp < 0) { const char* e = getenv("ORV_ATTN_PP"); use_pp = (e && atoi(e) == 0) ? 0 : 1; }
    if (!use_static || !use_pp || !fused || ld_out % 8 != 0 || ((uintptr_t)out & 15) != 0)
        return orv_attention_fwd(qkv, ld_qkv, nullptr, out, ld_out, lse, B, S, H, 0, scale, stream);
    AttnArgs a;
    a.qkv = (const bf16_t*)qkv; a.ld = ld_qkv; a.vT = nullptr; a.out = (bf16_t*)out; a.ld_out = ld_out;
    a.lse = lse; a.B = B; a.S = S; a.H = H; a.s_pad = 0;
    a.scale = scale; a.scale_log2 = scale_log2; a.shift = 0.f; a.guard_dev = score_bound_dev; a.guard_limit = ORV_STATIC_LIMIT_PP; a.ws_o = a.ws_l = nullptr; a.ws_cnt = nullptr; a.n_full = a.ks = 0;
    dim3 grid(((S + 255) / 256) * H * B);
    hipLaunchKernelGGL((attn_fwd_pp_kernel<true>), grid, dim3(512), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL((attn_fwd_pp_kernel<false>), grid, dim3(512), 0, (hipStream_t)stream, a);
    return orv_check_launch("orv_attention_fwd_bounded_dev");
}
