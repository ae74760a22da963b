// Attention backward for gfx950 (adjoint of orv_attention_fwd; torch autograd of F.scaled_dot_product_attention at
// orv/models/cogvideox_control.py:256-258 in the reference's training step).  Two passes, no atomics:
//   pass A (one lane = one QUERY row, loops over key tiles)  ->  dQ
//   pass B (one lane = one KEY row,   loops over query tiles) ->  dK, dV
// Both recompute P = exp2(q'.k - lse2) from the saved log-sum-exp (q' = q * scale * log2 e as stored by orv_qkv_prep),
// keep the row they own on the lane axis (swapped-operand MFMAs, as in the forward) so that -lse2 and -delta enter through
// the accumulator init, and feed dS / P back into the next MFMA as its B operand straight from accumulator registers.
// Operands whose contraction index is the sequence (K^T, Q^T, dO^T) come from per-head transposed copies
// [B,H,64,s_pad] written by orv_head_transpose in the key order the accumulator layout produces (bits 2<->3 exchanged).
#include <mutex>
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int TILE = 8192;  // 64 rows x 128 B

__device__ __forceinline__ int seq_pos(int s) { return (s & ~12) | ((s & 4) << 1) | ((s & 8) >> 1); }

// ---- [B*S, ld] columns [col0 + h*64, +64) -> dst[b, h, d, pos(s)], zero for s >= S ----
__global__ __launch_bounds__(256) void head_transpose_kernel(const bf16_t* __restrict__ src, long ld, int col0,
                                                             bf16_t* __restrict__ dst, int S, int H, int s_pad) {
    __shared__ bf16_t t_s[64][66];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int sub = lane & 7;
    for (int it = 0; it < 2; ++it) {
        const int tl = wave * 16 + it * 8 + (lane >> 3);
        const int s = s0 + tl;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (s < S) u = *(const uint4*)(src + ((long)b * S + s) * ld + col0 + h * 64 + sub * 8);
        uint32_t* d = (uint32_t*)&t_s[tl][sub * 8];
        d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = u.w;
    }
    __syncthreads();
    const int d = tid >> 2, grp = tid & 3;
    bf16_t o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[seq_pos(k)] = t_s[grp * 16 + k][d];
    bf16_t* out = dst + ((long)(b * H + h) * 64 + d) * s_pad + s0 + grp * 16;
    uint4 a, c2;
    a.x = o[0] | ((uint32_t)o[1] << 16); a.y = o[2] | ((uint32_t)o[3] << 16);
    a.z = o[4] | ((uint32_t)o[5] << 16); a.w = o[6] | ((uint32_t)o[7] << 16);
    c2.x = o[8] | ((uint32_t)o[9] << 16); c2.y = o[10] | ((uint32_t)o[11] << 16);
    c2.z = o[12] | ((uint32_t)o[13] << 16); c2.w = o[14] | ((uint32_t)o[15] << 16);
    *(uint4*)out = a;
    *(uint4*)(out + 8) = c2;
}

// ---- neg_delta[b,h,s] = -sum_d dO*O ; neg_lse2[b,h,s] = -lse / ln 2 ; both [B,H,s_pad] fp32, 0 beyond S ----
__global__ __launch_bounds__(256) void bwd_prep_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, long ld,
                                                       const float* __restrict__ lse, float* __restrict__ neg_delta,
                                                       float* __restrict__ neg_lse2, int S, int H, int s_pad, int B) {
    // one 8-lane group per (row, head)
    const long gid = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const long total = (long)B * s_pad * H;
    if (gid >= total) return;
    const int h = (int)(gid % H);
    const long rs = gid / H;
    const int s = (int)(rs % s_pad), b = (int)(rs / s_pad);
    float acc = 0.f;
    if (s < S) {
        const uint4 uo = *(const uint4*)(o + ((long)b * S + s) * ld + h * 64 + sub * 8);
        const uint4 ud = *(const uint4*)(dout + ((long)b * S + s) * ld + h * 64 + sub * 8);
        const uint32_t wo[4] = {uo.x, uo.y, uo.z, uo.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += bf2f(wo[e] & 0xffff) * bf2f(wd[e] & 0xffff) + bf2f(wo[e] >> 16) * bf2f(wd[e] >> 16);
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    if (sub == 0) {
        const long oi = ((long)b * H + h) * s_pad + s;
        neg_delta[oi] = -acc;
        neg_lse2[oi] = s < S ? -lse[((long)b * H + h) * S + s] * 1.4426950408889634f : 0.f;
    }
}

// -DORV_SEG_TRACE (tools/attn_bwd_seg_trace.sh): per-wave time in the matrix segment, at the barrier behind it, in the vector segment
// and at the barrier behind that, summed over the tiles (s_memrealtime ticks of 10 ns); the buffer is handed over by
// orv_debug_attn_bwd_trace(), which exists in such builds only
#ifdef ORV_SEG_TRACE
__device__ unsigned long long* g_bwd_trace = nullptr;
#define SEG_T(I) { const unsigned long long now_ = wall_clock64(); if ((I) > 0) seg_acc[(I) - 1] += now_ - seg_last; seg_last = now_; }
#define SEG_DECL unsigned long long seg_acc[4] = {0, 0, 0, 0}, seg_last = 0;
#define SEG_DUMP(K) if (g_bwd_trace && lane == 0) { unsigned long long* tr_ = g_bwd_trace + ((long)(K) * gridDim.x * 8 + (long)blockIdx.x * 8 + wave) * 4; for (int i_ = 0; i_ < 4; ++i_) tr_[i_] = seg_acc[i_]; }
#else
#define SEG_T(I)
#define SEG_DECL
#define SEG_DUMP(K)
#endif
struct BwdArgs {
    const bf16_t* qkv; long ld;      // q' | k | v (after orv_qkv_prep)
    const bf16_t* qT; const bf16_t* kT; const bf16_t* doT;   // [B,H,64,s_pad]
    const bf16_t* dout; long ld_do;  // [B*S, H*64]
    const float* neg_lse2; const float* neg_delta;           // [B,H,s_pad]
    bf16_t* dqkv; long ld_dqkv;      // [B*S, 3*H*64]: dq | dk | dv
    int B, S, H, s_pad;
    float scale;
};

// ===== pass A: dQ =====
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_kernel(const BwdArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 3 * TILE];   // per stage: K | V | K^T
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int nqt = (p.S + 255) / 256;                      // head-major items per XCD, as in the forward
    const int item = orv_xcd_item(blockIdx.x, gridDim.x);
    const int bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q0 = (item % nqt) * 256 + wave * 32;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;
    const int qr = min(q0 + l31, p.S - 1);
    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qp = p.qkv + (row0 + qr) * p.ld + h * 64 + hi * 8;
        const bf16_t* dp = p.dout + (row0 + qr) * p.ld_do + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(dp + ks * 16); }
    }
    const long vecq = ((long)b * p.H + h) * p.s_pad + qr;
    f32x16 c_lse, c_del;
    {
        const float nl = p.neg_lse2[vecq], nd = p.neg_delta[vecq];
#pragma unroll
        for (int e = 0; e < 16; ++e) { c_lse[e] = nl; c_del[e] = nd; }
    }
    // staging: wave w moves rows 8w..8w+7 of each of the three tiles
    const int srow = wave * 8 + (lane >> 3), slot = lane & 7;
    const int schunk = slot ^ ((srow >> 1) & 7);
    const bf16_t* kbase = p.qkv + D + h * 64 + schunk * 8;
    const bf16_t* vbase = p.qkv + 2 * D + h * 64 + schunk * 8;
    const bf16_t* ktsrc = p.kT + ((long)(b * p.H + h) * 64 + srow) * p.s_pad + schunk * 8;
    auto stage_load = [&](int s, int kv0) {
        const int krow = min(kv0 + srow, p.S - 1);
        glds16(kbase + (row0 + krow) * p.ld, smem + s * 3 * TILE + wave * 1024);
        glds16(vbase + (row0 + krow) * p.ld, smem + s * 3 * TILE + TILE + wave * 1024);
        glds16(ktsrc + kv0, smem + s * 3 * TILE + 2 * TILE + wave * 1024);
    };
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[i][e] = 0.f;
    const int row_off = l31 * 128;
    const int nt = (p.S + 63) / 64;
    stage_load(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage_load((t + 1) & 1, (t + 1) * 64);
        if (q0 >= p.S) continue;      // all 32 query rows of this wave are past S: stage and synchronise only
        const char* sK = smem + (t & 1) * 3 * TILE;
        const char* sV = sK + TILE;
        const char* sKT = sK + 2 * TILE;
        f32x16 sT[2], dP[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bf16x8 k0 = *(const bf16x8*)(sK + kb * 4096 + row_off + (((0 * 2 + hi) ^ sw) * 16));
            const bf16x8 v0 = *(const bf16x8*)(sV + kb * 4096 + row_off + (((0 * 2 + hi) ^ sw) * 16));
            sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0], c_lse, 0, 0, 0);
            dP[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, dof[0], c_del, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(sK + kb * 4096 + row_off + (((ks * 2 + hi) ^ sw) * 16));
                const bf16x8 vf = *(const bf16x8*)(sV + kb * 4096 + row_off + (((ks * 2 + hi) ^ sw) * 16));
                sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sT[kb], 0, 0, 0);
                dP[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dP[kb], 0, 0, 0);
            }
        }
        const bool tail = (t == nt - 1) && (p.S & 63) != 0;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv = __builtin_amdgcn_exp2f(sT[kb][r]);
                if (tail) pv = (t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.S) ? 0.f : pv;
                sT[kb][r] = pv * dP[kb][r];      // dS^T = P (dP - delta)
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { bf16x8 v; uint32_t u[4]; } ds;
#pragma unroll
            for (int i = 0; i < 4; ++i) ds.u[i] = pack2bf(sT[kk >> 1][(kk & 1) * 8 + 2 * i], sT[kk >> 1][(kk & 1) * 8 + 2 * i + 1]);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const bf16x8 ktf = *(const bf16x8*)(sKT + db * 4096 + row_off + (((kk * 2 + hi) ^ sw) * 16));
                dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, ds.v, dq[db], 0, 0, 0);
            }
        }
    }
    const int q = q0 + l31;
    if (q < p.S) {
        bf16_t* op = p.dqkv + (row0 + q) * p.ld_dqkv + h * 64 + hi * 4;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                uint2 o;
                o.x = pack2bf(dq[db][qd * 4 + 0] * p.scale, dq[db][qd * 4 + 1] * p.scale);
                o.y = pack2bf(dq[db][qd * 4 + 2] * p.scale, dq[db][qd * 4 + 3] * p.scale);
                *(uint2*)(op + db * 32 + qd * 8) = o;
            }
    }
}

// ===== pass B: dK, dV =====
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_kernel(const BwdArgs p) {
    // per stage: Q' | dO | Q'^T | dO^T (8 KiB each) | neg_lse2[64] | neg_delta[64] (fp32)
    constexpr int STG = 4 * TILE + 512;
    __shared__ __attribute__((aligned(16))) char smem[2 * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5, sw = (lane >> 1) & 7;
    const int nkt = (p.S + 255) / 256;
    const int item = orv_xcd_item(blockIdx.x, gridDim.x);
    const int bh = item / nkt, h = bh % p.H, b = bh / p.H;
    const int k0 = (item % nkt) * 256 + wave * 32;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;
    const int kr = min(k0 + l31, p.S - 1);
    bf16x8 kf[4], vf[4];   // B operands: this lane's key row of K and V
    {
        const bf16_t* kp = p.qkv + (row0 + kr) * p.ld + D + h * 64 + hi * 8;
        const bf16_t* vp = p.qkv + (row0 + kr) * p.ld + 2 * D + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const bf16x8*)(kp + ks * 16); vf[ks] = *(const bf16x8*)(vp + ks * 16); }
    }
    const int srow = wave * 8 + (lane >> 3), slot = lane & 7;
    const int schunk = slot ^ ((srow >> 1) & 7);
    const bf16_t* qbase = p.qkv + h * 64 + schunk * 8;
    const bf16_t* dobase = p.dout + h * 64 + schunk * 8;
    const long hb = (long)(b * p.H + h);
    const bf16_t* qtsrc = p.qT + (hb * 64 + srow) * p.s_pad + schunk * 8;
    const bf16_t* dotsrc = p.doT + (hb * 64 + srow) * p.s_pad + schunk * 8;
    auto stage_load = [&](int s, int q0) {
        char* base = smem + s * STG;
        const int qrow = min(q0 + srow, p.S - 1);
        glds16(qbase + (row0 + qrow) * p.ld, base + wave * 1024);
        glds16(dobase + (row0 + qrow) * p.ld_do, base + TILE + wave * 1024);
        glds16(qtsrc + q0, base + 2 * TILE + wave * 1024);
        glds16(dotsrc + q0, base + 3 * TILE + wave * 1024);
        // the two 64-float vectors also travel by LDS-DMA (a register round trip here would put a vmcnt(0) wait right
        // behind the tile loads just issued: wave 0 then stalls a full memory latency per tile and everyone meets it at
        // the barrier - measured as 50 % of all wave cycles parked)
        if (wave == 0) glds4(p.neg_lse2 + hb * p.s_pad + q0 + lane, base + 4 * TILE);
        if (wave == 1) glds4(p.neg_delta + hb * p.s_pad + q0 + lane, base + 4 * TILE + 256);
    };
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[i][e] = 0.f; dv[i][e] = 0.f; }
    const int row_off = l31 * 128;
    const int nt = (p.S + 63) / 64;
    stage_load(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage_load((t + 1) & 1, (t + 1) * 64);
        if (k0 >= p.S) continue;      // all 32 key rows of this wave are past S: stage and synchronise only
        const char* sQ = smem + (t & 1) * STG;
        const char* sDO = sQ + TILE;
        const char* sQT = sQ + 2 * TILE;
        const char* sDOT = sQ + 3 * TILE;
        const float* sL = (const float*)(sQ + 4 * TILE);
        const float* sDl = sL + 64;
        f32x16 sS[2], dP[2];   // [q block][reg]: rows = q, cols = this lane's key
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 cl, cd;     // accumulator init: -lse2[q(r)] / -delta[q(r)], q(r) = qb*32 + (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = *(const float4*)(sL + qb * 32 + g * 8 + hi * 4);
                const float4 d = *(const float4*)(sDl + qb * 32 + g * 8 + hi * 4);
                cl[g * 4] = a.x; cl[g * 4 + 1] = a.y; cl[g * 4 + 2] = a.z; cl[g * 4 + 3] = a.w;
                cd[g * 4] = d.x; cd[g * 4 + 1] = d.y; cd[g * 4 + 2] = d.z; cd[g * 4 + 3] = d.w;
            }
            const bf16x8 q0f = *(const bf16x8*)(sQ + qb * 4096 + row_off + (((0 * 2 + hi) ^ sw) * 16));
            const bf16x8 d0f = *(const bf16x8*)(sDO + qb * 4096 + row_off + (((0 * 2 + hi) ^ sw) * 16));
            sS[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q0f, kf[0], cl, 0, 0, 0);
            dP[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0f, vf[0], cd, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                const bf16x8 qf_ = *(const bf16x8*)(sQ + qb * 4096 + row_off + (((ks * 2 + hi) ^ sw) * 16));
                const bf16x8 df_ = *(const bf16x8*)(sDO + qb * 4096 + row_off + (((ks * 2 + hi) ^ sw) * 16));
                sS[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf_, kf[ks], sS[qb], 0, 0, 0);
                dP[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df_, vf[ks], dP[qb], 0, 0, 0);
            }
        }
        const bool tail = (t == nt - 1) && (p.S & 63) != 0;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (rows of keys >= S compute garbage that is never stored; only query rows >= S must not contribute)
                float pv = __builtin_amdgcn_exp2f(sS[qb][r]);
                if (tail) pv = (t * 64 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.S) ? 0.f : pv;   // uniform branch + select
                sS[qb][r] = pv;                 // P
                dP[qb][r] = pv * dP[qb][r];     // dS
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            union { bf16x8 v; uint32_t u[4]; } pf, ds;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pf.u[i] = pack2bf(sS[kk >> 1][(kk & 1) * 8 + 2 * i], sS[kk >> 1][(kk & 1) * 8 + 2 * i + 1]);
                ds.u[i] = pack2bf(dP[kk >> 1][(kk & 1) * 8 + 2 * i], dP[kk >> 1][(kk & 1) * 8 + 2 * i + 1]);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const bf16x8 dot = *(const bf16x8*)(sDOT + db * 4096 + row_off + (((kk * 2 + hi) ^ sw) * 16));
                const bf16x8 qt = *(const bf16x8*)(sQT + db * 4096 + row_off + (((kk * 2 + hi) ^ sw) * 16));
                dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf.v, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, ds.v, dk[db], 0, 0, 0);
            }
        }
    }
    const int key = k0 + l31;
    if (key < p.S) {
        // dk = scale * dS_raw^T q = dS_raw^T q' / log2(e)
        const float ksc = 0.6931471805599453f;
        bf16_t* okp = p.dqkv + (row0 + key) * p.ld_dqkv + D + h * 64 + hi * 4;
        bf16_t* ovp = p.dqkv + (row0 + key) * p.ld_dqkv + 2 * D + h * 64 + hi * 4;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                uint2 o;
                o.x = pack2bf(dk[db][qd * 4 + 0] * ksc, dk[db][qd * 4 + 1] * ksc);
                o.y = pack2bf(dk[db][qd * 4 + 2] * ksc, dk[db][qd * 4 + 3] * ksc);
                *(uint2*)(okp + db * 32 + qd * 8) = o;
                o.x = pack2bf(dv[db][qd * 4 + 0], dv[db][qd * 4 + 1]);
                o.y = pack2bf(dv[db][qd * 4 + 2], dv[db][qd * 4 + 3]);
                *(uint2*)(ovp + db * 32 + qd * 8) = o;
            }
    }
}

// ===============================================================================================================
// Ping-pong forms of both passes (round 3; the structure of attn_fwd_pp_kernel, attention.hip): the two waves of every SIMD run
// half a tile out of phase - one in its matrix segment X_t, the other in its vector segment Y_t - separated by one s_barrier per
// segment, the second half of the workgroup one barrier behind the first.  The passes above have one workgroup per CU (256
// registers), i.e. two waves per SIMD in lock step with every ds_read right in front of its MFMA: ~690 TFLOP/s.
//   * No transposed copies any more: the operands whose contraction index is the sequence (K^T for dQ; Q'^T and dO^T for dK, dV)
//     are read with ds_read_b64_tr_b16 from the SAME row-major tile the score GEMMs read with ds_read_b128, one tile later
//     (X_t = { gradient GEMMs of tile t-1 ; score GEMMs of tile t }), so orv_head_transpose and the qT / kT / doT buffers are
//     gone.  One LDS image is conflict-free for both read kinds (swz below).
//   * dual-use tiles live two segments longer.  Round 5: the dual-use operand (K in the dQ pass, Q' in the dK / dV pass) has FOUR 8-KiB slots
//     (tile t in slot t & 3) and BOTH halves stage tile t + 2 at the top of their vector segment Y_t and wait for tile t + 1 at the end of X_t
//     (a whole segment after its issue) - an LDS-DMA issued from inside the matrix segment cost that half ~270 ns per tile
//     (profiles/r5_attention_bwd_slots.txt); the other operand keeps its two (V) / three (dO) slots.
// ===============================================================================================================
// One LDS image serves both read kinds without bank conflicts: 16-byte chunk c of tile row r lives in slot c ^ f(r),
//   f(r) = (bit 1 of r) << 2 | (bits 3..2 of r).
// ds_read_b128 (16-lane groups of rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}, one chunk column): even and odd rows use the two
// halves of the 256-byte bank row, and within either parity the eight rows of a group get eight distinct f.  ds_read_b64_tr_b16
// (32-lane halves = 4 consecutive rows x one 64-byte half row): rows r and r + 2 share a parity and must not share the half row,
// which bit 2 of f = bit 1 of r guarantees (the forward's V image is this rule alone).  Before: the ds_read_b128-optimal image
// c ^ ((r >> 1) & 7) made every transposing read 2-way conflicted and the gradient GEMMs LDS-bound (one 1-KiB fragment per MFMA).
// LDS-DMA from inline asm (cdna guide 5.7: M0 written in the same statement): hipcc tracks every __builtin_amdgcn_global_load_lds as
// a pending LDS write and puts s_waitcnt vmcnt(0) in front of the next ds_read_b64_tr_b16 (its builtin carries no memory operand
// to disambiguate).  Here the transposing reads of the NEXT matrix segment are issued early in the vector segment, i.e. right
// behind a DMA issue of the same wave: from asm the DMA is invisible to that pass, and its completion is owned by the explicit
// s_waitcnt vmcnt(0) + s_barrier pairs of the schedule.
__device__ __forceinline__ void glds16_asm(const void* gsrc, const void* lds_dst) {
    unsigned keep;
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(d) : "memory");
}
__device__ __forceinline__ void glds4_asm(const void* gsrc, const void* lds_dst) {
    unsigned keep;
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(d) : "memory");
}
// the same from a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: one VGPR per stream instead of a 64-bit pointer
__device__ __forceinline__ unsigned long long bw_uniform64(const void* q) {
    const unsigned long long u = (unsigned long long)(uintptr_t)q;
    const unsigned hi_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)), lo_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)u);
    return ((unsigned long long)hi_ << 32) | lo_;        // (the builtin returns int: unsigned first, or bit 31 of the low word smears)
}
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, const void* lds_dst) {
    unsigned keep;
    const unsigned long long su = bw_uniform64(sbase);
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(su), "s"(d) : "memory");
}
__device__ __forceinline__ void glds4_sv(const void* sbase, unsigned voff, const void* lds_dst) {
    unsigned keep;
    const unsigned long long su = bw_uniform64(sbase);
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(su), "s"(d) : "memory");
}
__device__ __forceinline__ int swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
typedef short v4i16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_pair(const char* a, const char* b) {
#ifdef ORV_BW_ABL_NOTR
    bf16x8 z; asm volatile("" : "=v"(z)); return z;
#endif
    const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)a);
    const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)b);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
#define BW_BAR()                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define BW_FENCE() __builtin_amdgcn_sched_barrier(0);
#ifdef ORV_BW_ABL_NOMFMA
#define BW_MFMA(A_, B_, C_) ([&]() { asm volatile("" ::"v"(A_), "v"(B_)); return C_; }())
#else
#define BW_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
#endif
// per-lane byte offsets of the transposing reads into a swizzled [64 rows][64 d] tile: contraction rows kk * 16 + half * 8 + 4 hi +
// (r16 >> 2), logical bytes db * 64 + g16 * 32 + (r16 & 3) * 8 (4 consecutive d); + kk * 2048 is an immediate
__device__ __forceinline__ int tr_off(int lane, int db, int half) {
    const int hi = lane >> 5, r16 = lane & 15, g16 = (lane >> 4) & 1;
    const int row = half * 8 + 4 * hi + (r16 >> 2);
    const int chunk = (db * 4 + g16 * 2 + ((r16 & 3) >> 1)) ^ swz(row);
    return row * 128 + chunk * 16 + (r16 & 1) * 8;
}
__device__ __forceinline__ void store_row16(bf16_t* op, const f32x16 (&acc)[2], float sc) {
    // lane holds d = 32 db + 8 qd + 4 hi + (0..3) of its row: column groups qd = 2u (lower half-wave) / 2u + 1 (upper) are exchanged
    // so that every lane stores one aligned 16-byte piece (op already includes + hi * 8)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint32_t a0 = pack2bf(acc[db][(2 * u) * 4 + 0] * sc, acc[db][(2 * u) * 4 + 1] * sc);
            uint32_t a1 = pack2bf(acc[db][(2 * u) * 4 + 2] * sc, acc[db][(2 * u) * 4 + 3] * sc);
            uint32_t b0 = pack2bf(acc[db][(2 * u + 1) * 4 + 0] * sc, acc[db][(2 * u + 1) * 4 + 1] * sc);
            uint32_t b1 = pack2bf(acc[db][(2 * u + 1) * 4 + 2] * sc, acc[db][(2 * u + 1) * 4 + 3] * sc);
            { const auto r = __builtin_amdgcn_permlane32_swap(a0, b0, false, false); a0 = r[0]; b0 = r[1]; }
            { const auto r = __builtin_amdgcn_permlane32_swap(a1, b1, false, false); a1 = r[0]; b1 = r[1]; }
            *(uint4*)(op + db * 32 + u * 16) = make_uint4(a0, a1, b0, b1);
        }
}

// ===== pass A, ping-pong: dQ.  X_t = { dQ^T += K_{t-1}^T dS_{t-1}^T (8 MFMAs, transposing reads) ; S_t^T = K_t Q^T - lse,
// dP_t^T = V_t dO^T - delta (16 MFMAs, ds_read_b128) } ; Y_t = { P = exp2(S), dS = P dP, bf16 pack }.
// K: four slots (read as tile t and as tile t-1), staged by waves 0-3; V: two slots, staged by waves 4-7; both two tiles ahead, from the
// vector segment (an LDS-DMA issue inside the matrix segment cost the first half ~270 ns per tile: profiles/r5_attention_bwd_slots.txt).
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_pp_kernel(const BwdArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[6 * TILE];          // K slots 0-3 | V slots 0-1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5, sw = swz(l31);
    const int nqt = (p.S + 255) / 256;
    const int item = orv_xcd_item(blockIdx.x, gridDim.x);
    const int bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q0 = (item % nqt) * 256 + wave * 32;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;
    const int qr = min(q0 + l31, p.S - 1);
    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qp = p.qkv + (row0 + qr) * p.ld + h * 64 + hi * 8;
        const bf16_t* dp = p.dout + (row0 + qr) * p.ld_do + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(dp + ks * 16); }
    }
    f32x16 c_lse, c_del;
    {
        const long vecq = ((long)b * p.H + h) * p.s_pad + qr;
        const float nl = p.neg_lse2[vecq], nd = p.neg_delta[vecq];
#pragma unroll
        for (int e = 0; e < 16; ++e) { c_lse[e] = nl; c_del[e] = nd; }
    }
    const int nt = (p.S + 63) / 64;
    const bool ragged = (p.S & 63) != 0;
    auto src_of = [&](int j, int t) {              // this lane's source line of piece j in tile t (keys past S: a valid row)
        const int sr = wq * 16 + j * 8 + (lane >> 3), slot = lane & 7;
        return p.qkv + (row0 + min(t * 64 + sr, p.S - 1)) * p.ld + (grp == 0 ? D : 2 * D) + h * 64 + (slot ^ swz(sr)) * 8;
    };
    const bf16_t* const sbase0 = src_of(0, 0);
    const bf16_t* const sbase1 = src_of(1, 0);
    char* const sdst = smem + (grp == 0 ? 0 : 4 * TILE) + wq * 2048;
    auto stage = [&](int t) {
        if (t >= nt) return;
#ifdef ORV_BW_ABL_NODMA
        if (t > 1) return;
#endif
        char* const d = sdst + (grp == 0 ? t & 3 : t & 1) * TILE;
        if (__builtin_expect(ragged && t == nt - 1, 0)) {
            glds16_asm(src_of(0, t), d);
            glds16_asm(src_of(1, t), d + 1024);
        } else {
            const long toff = (long)t * 64 * p.ld;
            glds16_asm(sbase0 + toff, d);
            glds16_asm(sbase1 + toff, d + 1024);
        }
    };
    f32x16 dq[2], sT[2], dP[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dq[i][e] = 0.f; sT[i][e] = 0.f; dP[i][e] = 0.f; }
    union { bf16x8 v; uint32_t u[4]; } dsf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) dsf[i].u[e] = 0u;
    const int row_off = l31 * 128;
    const int t00 = tr_off(lane, 0, 0), t01 = tr_off(lane, 0, 1), t10 = tr_off(lane, 1, 0), t11 = tr_off(lane, 1, 1);

    // transposing-read fragments of the gradient GEMMs, k-steps 0 and 1: fetched at the top of the vector segment Y_{t-1} (the tile
    // they come from is resident since X_{t-1}) and held across the barrier, so X_t opens with MFMAs instead of an LDS round trip
    // ALL four k-steps (32 registers): one wave per SIMD gets one LDS instruction per ~25 clocks whatever its width (profiles/
    // r5_attention_bwd_slots.txt), so the 16 transposing reads of a tile belong in the vector segment, where this wave's LDS port idles,
    // and the matrix segment keeps the 16 ds_read_b128 of the score GEMMs only
    bf16x8 ga0 = qf[0], ga1 = qf[0], gb0 = qf[0], gb1 = qf[0], gc0 = qf[0], gc1 = qf[0], gd0 = qf[0], gd1 = qf[0];
    auto prefetch_g = [&](int t) {                                 // fragments of tile t for the gradient part of X_{t+1}
        const char* sP = smem + (t & 3) * TILE;
        ga0 = tr_pair(sP + t00, sP + t01); ga1 = tr_pair(sP + t10, sP + t11);
        gb0 = tr_pair(sP + 2048 + t00, sP + 2048 + t01); gb1 = tr_pair(sP + 2048 + t10, sP + 2048 + t11);
        gc0 = tr_pair(sP + 4096 + t00, sP + 4096 + t01); gc1 = tr_pair(sP + 4096 + t10, sP + 4096 + t11);
        gd0 = tr_pair(sP + 6144 + t00, sP + 6144 + t01); gd1 = tr_pair(sP + 6144 + t10, sP + 6144 + t11);
    };
    auto seg_x = [&](int t, auto stage_k) {
        __builtin_amdgcn_s_setprio(1);
        const char* sK = smem + (t & 3) * TILE + row_off;
        const char* sV = smem + 4 * TILE + (t & 1) * TILE + row_off;
        auto rd = [&](const char* base, int kb, int ks) {
#ifdef ORV_BW_ABL_NOB128
            bf16x8 z; asm volatile("" : "=v"(z)); return z;
#endif
            return *(const bf16x8*)(base + kb * 4096 + (((ks * 2 + hi) ^ sw) * 16)); };
        bf16x8 ka0, va0, ka1, va1, kb0, vb0, kb1, vb1;
        if (t < nt) { ka0 = rd(sK, 0, 0); va0 = rd(sV, 0, 0); ka1 = rd(sK, 0, 1); va1 = rd(sV, 0, 1); }   // first score fragments: in flight under the gradient MFMAs
        BW_FENCE()
        stage_k();
        BW_FENCE()
        if (t > 0) {
            dq[0] = BW_MFMA(ga0, dsf[0].v, dq[0]); dq[1] = BW_MFMA(ga1, dsf[0].v, dq[1]);
            dq[0] = BW_MFMA(gb0, dsf[1].v, dq[0]); dq[1] = BW_MFMA(gb1, dsf[1].v, dq[1]);
            BW_FENCE()
            if (t < nt) { kb0 = rd(sK, 0, 2); vb0 = rd(sV, 0, 2); kb1 = rd(sK, 0, 3); vb1 = rd(sV, 0, 3); }
            BW_FENCE()
            dq[0] = BW_MFMA(gc0, dsf[2].v, dq[0]); dq[1] = BW_MFMA(gc1, dsf[2].v, dq[1]);
            dq[0] = BW_MFMA(gd0, dsf[3].v, dq[0]); dq[1] = BW_MFMA(gd1, dsf[3].v, dq[1]);
            BW_FENCE()
        } else {
            kb0 = rd(sK, 0, 2); vb0 = rd(sV, 0, 2); kb1 = rd(sK, 0, 3); vb1 = rd(sV, 0, 3);
            BW_FENCE()
        }
        if (t < nt) {
            // 16 score MFMAs in four groups of (K, V) x two k-steps; the reads of group g + 1 are in flight under group g
            sT[0] = BW_MFMA(ka0, qf[0], c_lse); dP[0] = BW_MFMA(va0, dof[0], c_del);
            sT[0] = BW_MFMA(ka1, qf[1], sT[0]); dP[0] = BW_MFMA(va1, dof[1], dP[0]);
            BW_FENCE()
            ka0 = rd(sK, 1, 0); va0 = rd(sV, 1, 0); ka1 = rd(sK, 1, 1); va1 = rd(sV, 1, 1);
            BW_FENCE()
            sT[0] = BW_MFMA(kb0, qf[2], sT[0]); dP[0] = BW_MFMA(vb0, dof[2], dP[0]);
            sT[0] = BW_MFMA(kb1, qf[3], sT[0]); dP[0] = BW_MFMA(vb1, dof[3], dP[0]);
            BW_FENCE()
            kb0 = rd(sK, 1, 2); vb0 = rd(sV, 1, 2); kb1 = rd(sK, 1, 3); vb1 = rd(sV, 1, 3);
            BW_FENCE()
            sT[1] = BW_MFMA(ka0, qf[0], c_lse); dP[1] = BW_MFMA(va0, dof[0], c_del);
            sT[1] = BW_MFMA(ka1, qf[1], sT[1]); dP[1] = BW_MFMA(va1, dof[1], dP[1]);
            sT[1] = BW_MFMA(kb0, qf[2], sT[1]); dP[1] = BW_MFMA(vb0, dof[2], dP[1]);
            sT[1] = BW_MFMA(kb1, qf[3], sT[1]); dP[1] = BW_MFMA(vb1, dof[3], dP[1]);
            BW_FENCE()
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto seg_y = [&](int t) {
        prefetch_g(t);
        BW_FENCE()
#ifdef ORV_BW_ABL_NOY
        asm volatile("" ::"v"(sT[0]), "v"(sT[1]), "v"(dP[0]), "v"(dP[1]));
        return;
#endif
        // keys past S (ragged last tile only): score -> -inf, P = 0.  A separate, uniform branch: folded into the element loop the
        // compare + select pair (and the s_nop between them) ran for EVERY element of EVERY tile - 190 instead of 64 VALU
        // instructions per tile in a segment that has the SIMD's VALU port to itself (tools/attn_bwd_seg_trace.py: 860-1070 ns)
        if (__builtin_expect(ragged && t == nt - 1, 0)) {
            int kv0 = t * 64 + 4 * hi;
            asm volatile("" : "+v"(kv0));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) >= p.S) sT[kb][r] = -INFINITY;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sT[kb][r] = __builtin_amdgcn_exp2f(sT[kb][r]) * dP[kb][r];      // dS^T = P (dP - delta)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) dsf[kk].u[i] = pack2bf(sT[kk >> 1][(kk & 1) * 8 + 2 * i], sT[kk >> 1][(kk & 1) * 8 + 2 * i + 1]);
    };

    // prologue: tiles 0 and 1 of both operands
    stage(0);
    stage(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BW_BAR()
    const bool act = __builtin_amdgcn_readfirstlane((int)(q0 < p.S)) != 0;
    SEG_DECL
    if (grp == 0) {
        if (act) {
            for (int t = 0; t < nt; ++t) {
                SEG_T(0)
                seg_x(t, [&]() {});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K_{t+1} landed (issued in Y_{t-1} / the prologue)
                SEG_T(1)
                BW_BAR()
                SEG_T(2)
                stage(t + 2);                                    // K_{t+2} -> slot of K_{t-2} (last read in the partners' X_{t-1}, two intervals ago)
                BW_FENCE()
                seg_y(t);
                SEG_T(3)
                BW_BAR()
                SEG_T(4)
            }
            seg_x(nt, [&]() {});
        } else {
            for (int t = 0; t < nt; ++t) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                BW_BAR()
                stage(t + 2);
                BW_BAR()
            }
        }
        BW_BAR()
    } else {
        BW_BAR()
        if (act) {
            for (int t = 0; t < nt; ++t) {
                SEG_T(0)
                seg_x(t, [&]() {});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // V_{t+1} landed (issued in Y_{t-1} / the prologue)
                SEG_T(1)
                BW_BAR()
                SEG_T(2)
                stage(t + 2);                                    // V_{t+2} -> slot of V_t (last read in this half's X_t)
                BW_FENCE()
                seg_y(t);
                SEG_T(3)
                BW_BAR()
                SEG_T(4)
            }
            seg_x(nt, [&]() {});
        } else {
            for (int t = 0; t < nt; ++t) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                BW_BAR()
                stage(t + 2);
                BW_BAR()
            }
        }
    }
    SEG_DUMP(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int q = q0 + l31;
    if (q < p.S) store_row16(p.dqkv + (row0 + q) * p.ld_dqkv + h * 64 + hi * 8, dq, p.scale);
}

// ===== pass B, ping-pong: dK, dV (one lane = one key row; loop over query tiles).
// X_t = { dV^T += dO_{t-1}^T P_{t-1}, dK^T += Q'_{t-1}^T dS_{t-1} (16 MFMAs, transposing reads) ; S_t = Q'_t K^T - lse, dP_t = dO_t V^T - delta
// (16 MFMAs, ds_read_b128; -lse / -delta of the tile's 64 query rows enter through the accumulator init, from LDS) } ;
// Y_t = { P = exp2(S), dS = P dP, bf16 pack of both }.  Q' tiles (+ the lse vector): four slots, staged by waves 0-3; dO tiles (+ the delta
// vector): three slots, staged by waves 4-7; both two tiles ahead, from the vector segment.
// Shared fragment buffer (round 6; -DORV_BW_NO_FRAGBUF builds the round-5 form for A/B): the transposed fragments (dO^T, Q'^T) of k-steps 1-3 are the SAME for all eight waves, and every wave
// fetched them with 24 ds_read_b64_tr_b16 in its matrix segment - the segment that is bound by the per-wave LDS instruction rate
// (tools/probe_lds_bcast.cpp).  Here the four waves of the FIRST half produce them once per tile in their vector segment (three fragments each:
// six transposing reads + three ds_write_b128 into a "fragment-major" slot, fragment f of lane l at f KiB + 16 l), and every wave reads them
// back in its matrix segment with 12 ds_read_b128: 64 -> 52 LDS instructions per tile and wave where it counts.  Two slots of 12 KiB (tile t - 1 is
// still read by the second half while the first half writes tile t).
#ifndef ORV_BW_NO_FRAGBUF
#define ORV_BW_FRAGBUF 1
#endif
#ifdef ORV_BW_FRAGBUF
constexpr int DKV_FRAG0 = 7 * TILE + 7 * 256, DKV_FRAG_SLOT = 12 * 1024, DKV_SMEM = DKV_FRAG0 + 2 * DKV_FRAG_SLOT;
#else
constexpr int DKV_SMEM = 0;
#endif
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_pp_kernel(const BwdArgs p) {
#ifdef ORV_BW_FRAGBUF
    extern __shared__ __attribute__((aligned(16))) char smem[];              // Q' slots 0-3 | dO slots 0-2 | lse x4 | delta x3 | fragment slots x2
#else
    __shared__ __attribute__((aligned(16))) char smem[7 * TILE + 7 * 256];   // Q' slots 0-3 | dO slots 0-2 | lse x4 | delta x3
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5, sw = swz(l31);
    const int nkt = (p.S + 255) / 256;
    const int item = orv_xcd_item(blockIdx.x, gridDim.x);
    const int bh = item / nkt, h = bh % p.H, b = bh / p.H;
    const int k0 = (item % nkt) * 256 + wave * 32;
    const int D = p.H * 64;
    const long row0 = (long)b * p.S;
    const int kr = min(k0 + l31, p.S - 1);
    bf16x8 kf[4], vf[4];   // B operands: this lane's key row of K and V
    {
        const bf16_t* kp = p.qkv + (row0 + kr) * p.ld + D + h * 64 + hi * 8;
        const bf16_t* vp = p.qkv + (row0 + kr) * p.ld + 2 * D + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const bf16x8*)(kp + ks * 16); vf[ks] = *(const bf16x8*)(vp + ks * 16); }
    }
    const int nt = (p.S + 63) / 64;
    const bool ragged = (p.S & 63) != 0;
    const long hb = (long)(b * p.H + h);
    // DMA sources: wave-uniform base + 32-bit per-lane byte offsets (the 64-bit lane pointers of the first form were what the
    // allocator spilled inside the tile loop, and their reload put s_waitcnt vmcnt(0) between the DMA instructions of a tile)
    const long ldx = grp == 0 ? p.ld : p.ld_do;
    const char* const bbase = (const char*)((grp == 0 ? p.qkv : p.dout) + row0 * ldx + h * 64);     // batch base of this head's columns
    unsigned loff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sr = wq * 16 + j * 8 + (lane >> 3), slot = lane & 7;
        loff[j] = (unsigned)((sr * (int)ldx + (slot ^ swz(sr)) * 8) * 2);
    }
    char* const sdst = smem + grp * 4 * TILE + wq * 2048;
    const char* const vbase = (const char*)((grp == 0 ? p.neg_lse2 : p.neg_delta) + hb * p.s_pad);
    char* const vdst = smem + 7 * TILE + grp * 4 * 256;
    auto stage = [&](int t) {
        if (t >= nt) return;
#ifdef ORV_BW_ABL_NODMA
        if (t > 1) return;
#endif
        const int slot_ = grp == 0 ? (t & 3) : t % 3;       // Q': four slots (staged two tiles ahead from the vector segment), dO: three
        char* const d = sdst + slot_ * TILE;
        if (__builtin_expect(ragged && t == nt - 1, 0)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {          // rows past S read a valid row (masked in the vector segment)
                const int sr = wq * 16 + j * 8 + (lane >> 3), slot = lane & 7;
                glds16_sv(bbase, (unsigned)((min(t * 64 + sr, p.S - 1) * (int)ldx + (slot ^ swz(sr)) * 8) * 2), d + j * 1024);
            }
        } else {
            const char* const tb = bbase + (long)t * 64 * ldx * 2;
            glds16_sv(tb, loff[0], d);
            glds16_sv(tb, loff[1], d + 1024);
        }
        // the tile's 64-float vector travels by LDS-DMA too (a register round trip would put a vmcnt(0) behind the tile loads)
        if (wq == 0) glds4_sv(vbase + (long)t * 256, (unsigned)(lane * 4), vdst + slot_ * 256);
    };
    f32x16 dk[2], dv[2], sS[2], dP[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[i][e] = 0.f; dv[i][e] = 0.f; sS[i][e] = 0.f; dP[i][e] = 0.f; }
    union { bf16x8 v; uint32_t u[4]; } pf[4], dsf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { pf[i].u[e] = 0u; dsf[i].u[e] = 0u; }
    const int row_off = l31 * 128;
    const int t00 = tr_off(lane, 0, 0), t01 = tr_off(lane, 0, 1), t10 = tr_off(lane, 1, 0), t11 = tr_off(lane, 1, 1);

    // transposing-read fragments of the gradient GEMMs, k-step 0 (dO^T and Q'^T, two d blocks each: 16 registers - k-step 1 as
    // well does not fit beside the 192 registers of state): fetched at the top of the vector segment Y_{t-1} and held across the
    // barrier, so X_t opens with MFMAs instead of an LDS round trip
    bf16x8 o0 = kf[0], o1 = kf[0], g0_ = kf[0], g1_ = kf[0];
    auto prefetch_g = [&](int t) {                                 // fragments of tile t for the gradient part of X_{t+1}
        const char* sQp = smem + (t & 3) * TILE;
        const char* sOp = smem + 4 * TILE + (t % 3) * TILE;
        o0 = tr_pair(sOp + t00, sOp + t01); o1 = tr_pair(sOp + t10, sOp + t11);
        g0_ = tr_pair(sQp + t00, sQp + t01); g1_ = tr_pair(sQp + t10, sQp + t11);
    };
    auto seg_x = [&](int t, auto stage_q) {
        __builtin_amdgcn_s_setprio(1);
        const char* sQ = smem + (t & 3) * TILE + row_off;
        const char* sO = smem + 4 * TILE + (t % 3) * TILE + row_off;
        const float* sL = (const float*)(smem + 7 * TILE + (t & 3) * 256);
        const float* sDl = (const float*)(smem + 7 * TILE + 4 * 256 + (t % 3) * 256);
        auto rd = [&](const char* base, int qb, int ks) {
#ifdef ORV_BW_ABL_NOB128
            bf16x8 z; asm volatile("" : "=v"(z)); return z;
#endif
            return *(const bf16x8*)(base + qb * 4096 + (((ks * 2 + hi) ^ sw) * 16)); };
        auto init = [&](const float* v, int qb) {   // accumulator init: v[q(r)], q(r) = qb * 32 + (r & 3) + 8 (r >> 2) + 4 hi
            f32x16 c;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = *(const float4*)(v + qb * 32 + g * 8 + hi * 4);
                c[g * 4] = a.x; c[g * 4 + 1] = a.y; c[g * 4 + 2] = a.z; c[g * 4 + 3] = a.w;
            }
            return c;
        };
        bf16x8 qa0, oa0, qa1, oa1, qb0, ob0, qb1, ob1, o2, o3, g2_, g3_;
        if (t > 0) {
            const char* sQp = smem + ((t - 1) & 3) * TILE;               // Q' and dO of the previous tile, transposed reads
            const char* sOp = smem + 4 * TILE + ((t - 1) % 3) * TILE;
#ifdef ORV_BW_FRAGBUF
            const char* fb = smem + DKV_FRAG0 + ((t - 1) & 1) * DKV_FRAG_SLOT + lane * 16;
            o2 = *(const bf16x8*)(fb + 0 * 1024); o3 = *(const bf16x8*)(fb + 3 * 1024);
            g2_ = *(const bf16x8*)(fb + 6 * 1024); g3_ = *(const bf16x8*)(fb + 9 * 1024);
#else
            o2 = tr_pair(sOp + 2048 + t00, sOp + 2048 + t01); o3 = tr_pair(sOp + 2048 + t10, sOp + 2048 + t11);
            g2_ = tr_pair(sQp + 2048 + t00, sQp + 2048 + t01); g3_ = tr_pair(sQp + 2048 + t10, sQp + 2048 + t11);
#endif
        }
        BW_FENCE()
        stage_q();
        BW_FENCE()
        if (t > 0) {
            const char* sQp = smem + ((t - 1) & 3) * TILE;
            const char* sOp = smem + 4 * TILE + ((t - 1) % 3) * TILE;
            // per k-step: dv[0], dv[1] += dO^T[db] P ; dk[0], dk[1] += Q'^T[db] dS   (4 fragments, 4 MFMAs)
            dv[0] = BW_MFMA(o0, pf[0].v, dv[0]); dv[1] = BW_MFMA(o1, pf[0].v, dv[1]);
            dk[0] = BW_MFMA(g0_, dsf[0].v, dk[0]); dk[1] = BW_MFMA(g1_, dsf[0].v, dk[1]);
            BW_FENCE()
#ifdef ORV_BW_FRAGBUF
            const char* fb = smem + DKV_FRAG0 + ((t - 1) & 1) * DKV_FRAG_SLOT + lane * 16;
            o0 = *(const bf16x8*)(fb + 1 * 1024); o1 = *(const bf16x8*)(fb + 4 * 1024);
            g0_ = *(const bf16x8*)(fb + 7 * 1024); g1_ = *(const bf16x8*)(fb + 10 * 1024);
#else
            o0 = tr_pair(sOp + 4096 + t00, sOp + 4096 + t01); o1 = tr_pair(sOp + 4096 + t10, sOp + 4096 + t11);
            g0_ = tr_pair(sQp + 4096 + t00, sQp + 4096 + t01); g1_ = tr_pair(sQp + 4096 + t10, sQp + 4096 + t11);
#endif
            BW_FENCE()
            dv[0] = BW_MFMA(o2, pf[1].v, dv[0]); dv[1] = BW_MFMA(o3, pf[1].v, dv[1]);
            dk[0] = BW_MFMA(g2_, dsf[1].v, dk[0]); dk[1] = BW_MFMA(g3_, dsf[1].v, dk[1]);
            BW_FENCE()
#ifdef ORV_BW_FRAGBUF
            o2 = *(const bf16x8*)(fb + 2 * 1024); o3 = *(const bf16x8*)(fb + 5 * 1024);
            g2_ = *(const bf16x8*)(fb + 8 * 1024); g3_ = *(const bf16x8*)(fb + 11 * 1024);
#else
            o2 = tr_pair(sOp + 6144 + t00, sOp + 6144 + t01); o3 = tr_pair(sOp + 6144 + t10, sOp + 6144 + t11);
            g2_ = tr_pair(sQp + 6144 + t00, sQp + 6144 + t01); g3_ = tr_pair(sQp + 6144 + t10, sQp + 6144 + t11);
#endif
            BW_FENCE()
            dv[0] = BW_MFMA(o0, pf[2].v, dv[0]); dv[1] = BW_MFMA(o1, pf[2].v, dv[1]);
            dk[0] = BW_MFMA(g0_, dsf[2].v, dk[0]); dk[1] = BW_MFMA(g1_, dsf[2].v, dk[1]);
            BW_FENCE()
            if (t < nt) { qa0 = rd(sQ, 0, 0); oa0 = rd(sO, 0, 0); qa1 = rd(sQ, 0, 1); oa1 = rd(sO, 0, 1); }   // first score fragments: under the last gradient MFMAs
            BW_FENCE()
            dv[0] = BW_MFMA(o2, pf[3].v, dv[0]); dv[1] = BW_MFMA(o3, pf[3].v, dv[1]);
            dk[0] = BW_MFMA(g2_, dsf[3].v, dk[0]); dk[1] = BW_MFMA(g3_, dsf[3].v, dk[1]);
            BW_FENCE()
        } else {
            qa0 = rd(sQ, 0, 0); oa0 = rd(sO, 0, 0); qa1 = rd(sQ, 0, 1); oa1 = rd(sO, 0, 1);
            BW_FENCE()
        }
        if (t < nt) {
            f32x16 cl = init(sL, 0), cd = init(sDl, 0);
            qb0 = rd(sQ, 0, 2); ob0 = rd(sO, 0, 2); qb1 = rd(sQ, 0, 3); ob1 = rd(sO, 0, 3);
            BW_FENCE()
            sS[0] = BW_MFMA(qa0, kf[0], cl); dP[0] = BW_MFMA(oa0, vf[0], cd);
            sS[0] = BW_MFMA(qa1, kf[1], sS[0]); dP[0] = BW_MFMA(oa1, vf[1], dP[0]);
            BW_FENCE()
            qa0 = rd(sQ, 1, 0); oa0 = rd(sO, 1, 0); qa1 = rd(sQ, 1, 1); oa1 = rd(sO, 1, 1);
            cl = init(sL, 1); cd = init(sDl, 1);
            BW_FENCE()
            sS[0] = BW_MFMA(qb0, kf[2], sS[0]); dP[0] = BW_MFMA(ob0, vf[2], dP[0]);
            sS[0] = BW_MFMA(qb1, kf[3], sS[0]); dP[0] = BW_MFMA(ob1, vf[3], dP[0]);
            BW_FENCE()
            qb0 = rd(sQ, 1, 2); ob0 = rd(sO, 1, 2); qb1 = rd(sQ, 1, 3); ob1 = rd(sO, 1, 3);
            BW_FENCE()
            sS[1] = BW_MFMA(qa0, kf[0], cl); dP[1] = BW_MFMA(oa0, vf[0], cd);
            sS[1] = BW_MFMA(qa1, kf[1], sS[1]); dP[1] = BW_MFMA(oa1, vf[1], dP[1]);
            sS[1] = BW_MFMA(qb0, kf[2], sS[1]); dP[1] = BW_MFMA(ob0, vf[2], dP[1]);
            sS[1] = BW_MFMA(qb1, kf[3], sS[1]); dP[1] = BW_MFMA(ob1, vf[3], dP[1]);
            BW_FENCE()
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // fragment (operand op = wq >> 1: 0 = dO^T, 1 = Q'^T; d block db = wq & 1; k-step ks) of tile t -> slot t & 1, index (op * 2 + db) * 3 + ks - 1
    auto produce_frags = [&](int t) {
#ifdef ORV_BW_FRAGBUF
        const int op = wq >> 1, db = wq & 1;
        const char* base = op == 0 ? smem + 4 * TILE + (t % 3) * TILE : smem + (t & 3) * TILE;
        const int ta = db ? t10 : t00, tb = db ? t11 : t01;
        char* fdst = smem + DKV_FRAG0 + (t & 1) * DKV_FRAG_SLOT + (op * 2 + db) * 3 * 1024 + lane * 16;
        const bf16x8 f1 = tr_pair(base + 2048 + ta, base + 2048 + tb), f2 = tr_pair(base + 4096 + ta, base + 4096 + tb),
                     f3 = tr_pair(base + 6144 + ta, base + 6144 + tb);
        *(bf16x8*)(fdst) = f1; *(bf16x8*)(fdst + 1024) = f2; *(bf16x8*)(fdst + 2048) = f3;
#endif
    };
    auto seg_y = [&](int t) {
        prefetch_g(t);
        BW_FENCE()
#ifdef ORV_BW_FRAGBUF
        if (grp == 0) { produce_frags(t); BW_FENCE() }
#endif
#ifdef ORV_BW_ABL_NOY
        asm volatile("" ::"v"(sS[0]), "v"(sS[1]), "v"(dP[0]), "v"(dP[1]));
        return;
#endif
        // (rows of keys >= S compute garbage that is never stored; only query rows >= S must not contribute: ragged last tile only,
        // as a uniform branch ahead of the element loop - see attn_bwd_dq_pp_kernel)
        if (__builtin_expect(ragged && t == nt - 1, 0)) {
            int qv0 = t * 64 + 4 * hi;
            asm volatile("" : "+v"(qv0));
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (qv0 + qb * 32 + (r & 3) + 8 * (r >> 2) >= p.S) sS[qb][r] = -INFINITY;
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sS[qb][r]);
                sS[qb][r] = pv;                 // P
                dP[qb][r] = pv * dP[qb][r];     // dS
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pf[kk].u[i] = pack2bf(sS[kk >> 1][(kk & 1) * 8 + 2 * i], sS[kk >> 1][(kk & 1) * 8 + 2 * i + 1]);
                dsf[kk].u[i] = pack2bf(dP[kk >> 1][(kk & 1) * 8 + 2 * i], dP[kk >> 1][(kk & 1) * 8 + 2 * i + 1]);
            }
    };

    // prologue: tiles 0 and 1 of both operands (Q' + lse by the first half, dO + delta by the second)
    stage(0);
    stage(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BW_BAR()
    const bool act = __builtin_amdgcn_readfirstlane((int)(k0 < p.S)) != 0;
    SEG_DECL
    if (grp == 0) {
        if (act) {
            for (int t = 0; t < nt; ++t) {
                SEG_T(0)
                seg_x(t, [&]() {});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Q'_{t+1} landed (issued in Y_{t-1} / the prologue)
                SEG_T(1)
                BW_BAR()
                SEG_T(2)
                stage(t + 2);                                    // Q'_{t+2} -> slot of Q'_{t-2} (last read in the partners' X_{t-1}, two intervals ago)
                BW_FENCE()
                seg_y(t);
                SEG_T(3)
                BW_BAR()
                SEG_T(4)
            }
            seg_x(nt, [&]() {});
        } else {
            for (int t = 0; t < nt; ++t) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                BW_BAR()
                stage(t + 2);
                produce_frags(t);                                // a wave without keys still owns its share of the fragment slot
                BW_BAR()
            }
        }
        BW_BAR()
    } else {
        BW_BAR()
        if (act) {
            for (int t = 0; t < nt; ++t) {
                SEG_T(0)
                seg_x(t, [&]() {});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SEG_T(1)
                BW_BAR()
                SEG_T(2)
                stage(t + 2);                                    // dO_{t+2} -> slot of dO_{t-1} (last read in this half's X_t)
                BW_FENCE()
                seg_y(t);
                SEG_T(3)
                BW_BAR()
                SEG_T(4)
            }
            seg_x(nt, [&]() {});
        } else {
            for (int t = 0; t < nt; ++t) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                BW_BAR()
                stage(t + 2);
                BW_BAR()
            }
        }
    }
    SEG_DUMP(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int key = k0 + l31;
    if (key < p.S) {
        // dk = scale * dS_raw^T q = dS_raw^T q' / log2(e)
        store_row16(p.dqkv + (row0 + key) * p.ld_dqkv + D + h * 64 + hi * 8, dk, 0.6931471805599453f);
        store_row16(p.dqkv + (row0 + key) * p.ld_dqkv + 2 * D + h * 64 + hi * 8, dv, 1.0f);
    }
}
#undef BW_BAR
#undef BW_FENCE
#undef BW_MFMA

// sum over the 8 row groups of a wave (lane ^ 8, ^ 16, ^ 32) on the VALU alone; the 8-lane sum is orv_sum8 (common.hpp)
__device__ __forceinline__ float sum_rows8(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128, 0xF, 0xF, true));    // row_ror:8 = lane ^ 8
    { const unsigned u = __float_as_uint(v); const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const unsigned u = __float_as_uint(v); const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return v;
}
// ---- adjoint of orv_qkv_prep on the q and k thirds, in place on dqkv (dq, dk arrive as gradients of the normalised,
//      rotated, un-premultiplied q / k): inverse RoPE rotation, LayerNorm(64) backward; norm_q / norm_k weight gradients.
__global__ __launch_bounds__(256) void qkv_prep_bwd_kernel(const bf16_t* __restrict__ qkv_raw, bf16_t* __restrict__ dqkv,
                                                           const bf16_t* __restrict__ gq, const bf16_t* __restrict__ gk,
                                                           const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                           float* __restrict__ partial, int S, int H,
                                                           int n_text, float eps) {
    // partial[block][which][0: dgamma | 1: dbeta][64]: per-block sums, reduced by reduce_partials_kernel (no atomics on
    // the 256 shared addresses - with 6000 blocks they serialised into milliseconds)
    __shared__ float red[4][2][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const long ld = 3L * H * 64;
    const int sub = lane & 7;
    for (int which = 0; which < 2; ++which) {
        const bf16_t* g = which ? gk : gq;
        float gam[8], ag[8], ab[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gam[e] = g ? bf2f(g[sub * 8 + e]) : 1.f; ag[e] = ab[e] = 0.f; }
        for (int it = 0; it < 2; ++it) {
            const int s = s0 + wave * 16 + it * 8 + (lane >> 3);
            const bool ok = s < S;
            const long off = ((long)b * S + (ok ? s : S - 1)) * ld + which * H * 64 + h * 64 + sub * 8;
            const uint4 ux = *(const uint4*)(qkv_raw + off);
            const uint4 ud = *(const uint4*)(dqkv + off);
            const uint32_t wx[4] = {ux.x, ux.y, ux.z, ux.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
            float x[8], dz[8], sum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[2 * e] = bf2f(wx[e] & 0xffff); x[2 * e + 1] = bf2f(wx[e] >> 16);
                dz[2 * e] = ok ? bf2f(wd[e] & 0xffff) : 0.f; dz[2 * e + 1] = ok ? bf2f(wd[e] >> 16) : 0.f;
                sum += x[2 * e] + x[2 * e + 1];
            }
            if (rcos && s >= n_text && ok) {   // inverse rotation (adjoint of pairs (2i,2i+1): x cos + rot(x) sin)
                const float* cp = rcos + (long)(s - n_text) * 64 + sub * 8;
                const float* sp = rsin + (long)(s - n_text) * 64 + sub * 8;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float a = dz[e], c2 = dz[e + 1];
                    dz[e] = a * cp[e] + c2 * sp[e + 1];
                    dz[e + 1] = c2 * cp[e + 1] - a * sp[e];
                }
            }
            sum = orv_sum8(sum);
            const float mean = sum * (1.f / 64.f);
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[e] -= mean; sq += x[e] * x[e]; }
            sq = orv_sum8(sq);
            const float rstd = rsqrtf(sq * (1.f / 64.f) + eps);
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] *= rstd;                     // xh
                ag[e] += dz[e] * x[e];
                ab[e] += dz[e];
                dz[e] *= gam[e];                  // dxh
                m1 += dz[e];
                m2 += dz[e] * x[e];
            }
            m1 = orv_sum8(m1); m2 = orv_sum8(m2);
            m1 *= (1.f / 64.f); m2 *= (1.f / 64.f);
            if (ok) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (dz[e] - m1 - x[e] * m2);
                *(uint4*)(dqkv + off) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
            }
        }
        // reduce the weight gradients over the 8 rows of the wave (lanes with equal `sub`), then one atomic per column
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = ag[e], c2 = ab[e];
            a = sum_rows8(a); c2 = sum_rows8(c2);
            if (lane < 8) {
                red[wave][which][0][sub * 8 + e] = a;
                red[wave][which][1][sub * 8 + e] = c2;
            }
        }
    }
    __syncthreads();
    {
        const int w2 = tid >> 7, gb = (tid >> 6) & 1, col = tid & 63;     // 256 threads = 2 x 2 x 64 outputs
        const float v = red[0][w2][gb][col] + red[1][w2][gb][col] + red[2][w2][gb][col] + red[3][w2][gb][col];
        const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[blk * 256 + tid] = v;
    }
}

// out[j] += sum_rows partial[row][j], j < 256: one block per 64-row slab, fp32 atomics only across slabs
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, long nrows,
                                                              float* __restrict__ o0, float* __restrict__ o1,
                                                              float* __restrict__ o2, float* __restrict__ o3) {
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * 64, r1 = min(nrows, r0 + 64);
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += partial[r * 256 + tid];
    float* dst = (tid < 64) ? o0 : (tid < 128) ? o1 : (tid < 192) ? o2 : o3;     // dgq | dbq | dgk | dbk
    if (dst) atomicAdd(dst + (tid & 63), s);
}

}  // namespace

extern "C" int orv_head_transpose(const void* src, int ld, int col0, void* dst, int B, int S, int H, int s_pad, void* stream) {
    ORV_REQUIRE(src && dst && B > 0 && S > 0 && H > 0, "orv_head_transpose: bad arguments");
    ORV_REQUIRE(s_pad == ((S + 63) / 64) * 64 && ld % 8 == 0 && col0 % 8 == 0, "orv_head_transpose: bad s_pad/ld");
    hipLaunchKernelGGL(head_transpose_kernel, dim3(s_pad / 64, H, B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (long)ld, col0, (bf16_t*)dst, S, H, s_pad);
    return orv_check_launch("orv_head_transpose");
}

// qT / kT / doT (per-head transposed copies written by orv_head_transpose) feed the two-pass kernels of rounds 1-2 only; with all
// three NULL - what the host passes since round 3 - the ping-pong kernels read K^T / Q'^T / dO^T with transposing LDS reads from the
// row-major tiles.  ORV_ATTN_BWD_PP=0 + non-NULL copies: the old kernels (A/B).
#ifdef ORV_SEG_TRACE
extern "C" int orv_debug_attn_bwd_trace(void* buf) {      // variant builds only: where the kernels leave their segment times
    unsigned long long* q = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_trace), &q, sizeof(q)) == hipSuccess ? 0 : 1;
}
#endif

namespace {
// per-device side stream + fork / join events of orv_attention_bwd (created on first use, kept)
struct BwdSide { hipStream_t stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; bool ok = false, tried = false; };
// Initialisation is serialised (two host threads may call orv_attention_bwd on one device); the events are shared per device, so the
// fork / join sequence itself runs under the same mutex: record + wait pairs of two callers must not interleave.
std::mutex g_bwd_side_mutex;
BwdSide* bwd_side() {
    static BwdSide sides[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    BwdSide& s = sides[dev];
    if (!s.tried) {
        s.tried = true;
        s.ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming) == hipSuccess;
    }
    return s.ok ? &s : nullptr;
}
}  // namespace

extern "C" int orv_attention_bwd(const void* qkv, int ld_qkv, const void* qT, const void* kT, const void* out,
                                 const void* dout, int ld_out, const void* doT, const float* lse, float* neg_lse2,
                                 float* neg_delta, void* dqkv, int ld_dqkv, int B, int S, int H, int s_pad, float scale,
                                 void* stream) {
    ORV_REQUIRE(qkv && out && dout && lse && neg_lse2 && neg_delta && dqkv, "orv_attention_bwd: null operand");
    ORV_REQUIRE(s_pad == ((S + 63) / 64) * 64, "orv_attention_bwd: s_pad must be S rounded up to 64");
    static int use_pp = -1;
    if (use_pp < 0) { const char* e = getenv("ORV_ATTN_BWD_PP"); use_pp = (e && atoi(e) == 0) ? 0 : 1; }
    const bool have_t = qT && kT && doT;
    const bool pp = (use_pp || !have_t) && ld_dqkv % 8 == 0 && ((uintptr_t)dqkv & 15) == 0;
    ORV_REQUIRE(pp || have_t, "orv_attention_bwd: the transposed-copy kernels need qT, kT and doT (or a 16-byte aligned dqkv for the ping-pong kernels)");
    hipStream_t st = (hipStream_t)stream;
    const long groups = (long)B * s_pad * H;
    hipLaunchKernelGGL(bwd_prep_kernel, dim3((unsigned)((groups * 8 + 255) / 256)), dim3(256), 0, st, (const bf16_t*)out,
                       (const bf16_t*)dout, (long)ld_out, lse, neg_delta, neg_lse2, S, H, s_pad, B);
    BwdArgs a;
    a.qkv = (const bf16_t*)qkv; a.ld = ld_qkv; a.qT = (const bf16_t*)qT; a.kT = (const bf16_t*)kT; a.doT = (const bf16_t*)doT;
    a.dout = (const bf16_t*)dout; a.ld_do = ld_out; a.neg_lse2 = neg_lse2; a.neg_delta = neg_delta;
    a.dqkv = (bf16_t*)dqkv; a.ld_dqkv = ld_dqkv; a.B = B; a.S = S; a.H = H; a.s_pad = s_pad; a.scale = scale;
    dim3 grid(((S + 255) / 256) * H * B);   // 1-D: orv_xcd_item hands head-major items to the XCDs
    if (pp && DKV_SMEM > 0) {
        static bool attr_done = false;
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_SMEM); attr_done = true; }
    }
    if (pp) {
        // The two passes are independent (dQ | dK, dV: disjoint columns of dqkv) and each is 1560 one-per-CU workgroups = 6.09 -> 7
        // rounds of 256 CUs at B = 4: launched on two streams the dispatcher fills the empty part of one pass's last round with
        // the other's workgroups (fork / join by events, legal under stream capture).  ORV_ATTN_BWD_FORK=0: back to back.
        static int fork = -1;
        if (fork < 0) { const char* e = getenv("ORV_ATTN_BWD_FORK"); fork = (e && atoi(e) == 0) ? 0 : 1; }
        std::lock_guard<std::mutex> lock(g_bwd_side_mutex);
        BwdSide* side = fork ? bwd_side() : nullptr;
        // a failed fork (event record / wait refused) must not let the dK / dV pass run unordered against its inputs: fall back to
        // the back-to-back launch on the caller's stream
        const bool forked = side && hipEventRecord(side->ev_fork, st) == hipSuccess &&
                            hipStreamWaitEvent(side->stream, side->ev_fork, 0) == hipSuccess;
        if (forked) {
            hipLaunchKernelGGL(attn_bwd_dkv_pp_kernel, grid, dim3(512), DKV_SMEM, side->stream, a);
            hipLaunchKernelGGL(attn_bwd_dq_pp_kernel, grid, dim3(512), 0, st, a);
            if (hipEventRecord(side->ev_join, side->stream) != hipSuccess || hipStreamWaitEvent(st, side->ev_join, 0) != hipSuccess) {
                // the join could not be queued: dqkv must not be consumed before the side pass is done
                (void)hipStreamSynchronize(side->stream);
                orv_set_error("orv_attention_bwd: joining the dK / dV side stream failed (%s)", hipGetErrorString(hipGetLastError()));
                return ORV_EDEVICE;
            }
        } else {
            (void)hipGetLastError();
            hipLaunchKernelGGL(attn_bwd_dq_pp_kernel, grid, dim3(512), 0, st, a);
            hipLaunchKernelGGL(attn_bwd_dkv_pp_kernel, grid, dim3(512), DKV_SMEM, st, a);
        }
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(512), 0, st, a);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(512), 0, st, a);
    }
    return orv_check_launch("orv_attention_bwd");
}

extern "C" int orv_qkv_prep_bwd(const void* qkv_raw, void* dqkv, const void* gq, const void* gk, const float* rope_cos,
                                const float* rope_sin, float* dgq, float* dbq, float* dgk, float* dbk, float* scratch,
                                int B, int S, int H, int n_text, float eps, void* stream) {
    ORV_REQUIRE(qkv_raw && dqkv && scratch && B > 0 && S > 0 && H > 0, "orv_qkv_prep_bwd: bad arguments");
    ORV_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), "orv_qkv_prep_bwd: cos and sin go together");
    hipLaunchKernelGGL(qkv_prep_bwd_kernel, dim3((S + 63) / 64, H, B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv_raw, (bf16_t*)dqkv, (const bf16_t*)gq, (const bf16_t*)gk, rope_cos, rope_sin, scratch,
                       S, H, n_text, eps);
    const long nblk = (long)((S + 63) / 64) * H * B;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((nblk + 63) / 64)), dim3(256), 0, (hipStream_t)stream, scratch, nblk,
                       dgq, dbq, dgk, dbk);
    return orv_check_launch("orv_qkv_prep_bwd");
}
