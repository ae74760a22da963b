// HBM-bound normalisation kernels for gfx950:
//   orv_layernorm_modulate : LayerNorm over D + per-token-group (1+scale)/shift  (AdaLN, cogvideox_control.py:117-145,155-197)
//   orv_qkv_prep           : per-head LayerNorm(64) on q,k (+RoPE) in place, V -> V^T  (cogvideox_control.py:239-254)
// One wave per row; 16-byte vector loads/stores; statistics in fp32 (two-pass in registers).
#include "common.hpp"

namespace {

// Round 4: every load of the kernel is UNCONDITIONAL (chunk index clamped, contributions of the lanes past the row end masked): with
// the loads inside `if (c < nchunk)` hipcc branched around each one and waited vmcnt(0) behind it - the four 16-byte loads of a row
// went out one dependent round trip after the other (cdna_hip_programming.md, "three .s-level traps" (c)) - and the wave reductions run
// on the VALU (DPP + half-swaps) instead of 2 x 6 dependent ds_bpermute.
// FULL: gamma, beta, scale and shift are all present (the two norms of every block): their loads carry no branch at all, so the compiler
// batches them (with the null checks of the general form every factor load sat behind a scalar branch + its own vmcnt(0): twelve
// dependent L2 round trips per row).
#ifndef ORV_LN_WAVES          // minimum waves per SIMD the register allocation must leave room for (A/B builds: tools/variants.sh)
#define ORV_LN_WAVES 1
#endif
template <int CH, bool FULL>
__global__ __launch_bounds__(256, ORV_LN_WAVES) void ln_mod_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y,
                                                     long ldy, const bf16_t* __restrict__ gamma,
                                                     const bf16_t* __restrict__ beta, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, long mod_b, long mod_g, int seq,
                                                     int n_text, int per_group, int rows, int D, float eps,
                                                     orv_rowmap_t xmap) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = D >> 3;
    const long xrow = xmap.rows > 0 ? (long)(row / xmap.rows) * xmap.bstride + xmap.off + row % xmap.rows : row;
    const bf16_t* xr = x + xrow * ldx;
    uint4 raw[CH];
    int ce[CH];
    bool ok[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        ok[i] = c < nchunk;
        ce[i] = min(c, nchunk - 1);
        raw[i] = *(const uint4*)(xr + ce[i] * 8);
    }
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const uint32_t w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[i][2 * e] = bf2f(w[e] & 0xffff);          // lanes past the row end carry the LAST chunk's real values (stored again
            v[i][2 * e + 1] = bf2f(w[e] >> 16);         // below, identical bytes); only the statistics mask them out
            s += ok[i] ? v[i][2 * e] + v[i][2 * e + 1] : 0.f;
        }
    }
    const float mean = wave_sum_valu(s) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = ok[i] ? v[i][e] - mean : 0.f;
            sq += d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum_valu(sq) / (float)D + eps);
    const float* sc = nullptr;
    const float* sh = nullptr;
    if (FULL || scale) {
        const int b = row / seq, sidx = row % seq;
        const long off = b * mod_b + orv_group_of(sidx, n_text, per_group) * mod_g;
        sc = scale + off;
        sh = shift + off;
    }
    bf16_t* yr = y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = ce[i];
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd;
        if (FULL || gamma) {
            const uint4 g = *(const uint4*)(gamma + c * 8);
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[2 * e] *= bf2f(gw[e] & 0xffff);
                o[2 * e + 1] *= bf2f(gw[e] >> 16);
            }
        }
        if (FULL || beta) {
            const uint4 g = *(const uint4*)(beta + c * 8);
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[2 * e] += bf2f(gw[e] & 0xffff);
                o[2 * e + 1] += bf2f(gw[e] >> 16);
            }
        }
        if (FULL || sc) {
            const float4 s0 = *(const float4*)(sc + c * 8), s1 = *(const float4*)(sc + c * 8 + 4);
            const float4 h0 = *(const float4*)(sh + c * 8), h1 = *(const float4*)(sh + c * 8 + 4);
            const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(o[e], 1.0f + ss[e], hh[e]);
        }
        uint4 u;
        u.x = pack2bf(o[0], o[1]); u.y = pack2bf(o[2], o[3]); u.z = pack2bf(o[4], o[5]); u.w = pack2bf(o[6], o[7]);
        // lanes past the row end hold the row's LAST chunk (clamped index: same input, same factors, same result): they store it too -
        // identical bytes to the same address - so the kernel has no divergent branch and its loads can be scheduled as one batch
        *(uint4*)(yr + c * 8) = u;
    }
}

// position of key s inside vT's key axis: bits 2 and 3 of s exchanged (see attention.hip header)
__device__ __forceinline__ int vt_pos(int s) { return (s & ~12) | ((s & 4) << 1) | ((s & 8) >> 1); }

// grid (ceil(S/64), H, B), 256 threads: 64 tokens of one head.
__global__ __launch_bounds__(256) void qkv_prep_kernel(const bf16_t* src, bf16_t* qkv, bf16_t* __restrict__ vT,
                                                       const bf16_t* __restrict__ gq, const bf16_t* __restrict__ bq,
                                                       const bf16_t* __restrict__ gk, const bf16_t* __restrict__ bk,
                                                       const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                       int S, int H, int n_text, int s_pad, float eps, float q_premul) {
    __shared__ bf16_t vt_s[64][66];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const long ld = 3L * H * 64;
    const int sub = lane & 7;  // 16-byte chunk of the 64-wide head row
    // q and k: 8 lanes per row, 8 rows per wave-instruction
    for (int which = 0; which < 2; ++which) {
        const bf16_t* g = which ? gk : gq;
        const bf16_t* be = which ? bk : bq;
        float gam[8], bet[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gam[e] = g ? bf2f(g[sub * 8 + e]) : 1.f;
            bet[e] = be ? bf2f(be[sub * 8 + e]) : 0.f;
        }
        for (int it = 0; it < 2; ++it) {
            const int s = s0 + wave * 16 + it * 8 + (lane >> 3);
            const bool ok = s < S;
            const long eoff = ((long)b * S + (ok ? s : S - 1)) * ld + which * H * 64 + h * 64 + sub * 8;
            bf16_t* ptr = qkv + eoff;
            const uint4 u = *(const uint4*)(src + eoff);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
            float v[8], sum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = bf2f(w[e] & 0xffff);
                v[2 * e + 1] = bf2f(w[e] >> 16);
                sum += v[2 * e] + v[2 * e + 1];
            }
            sum = orv_sum8(sum);
            const float mean = sum * (1.f / 64.f);
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] -= mean; sq += v[e] * v[e]; }
            sq = orv_sum8(sq);
            const float rstd = rsqrtf(sq * (1.f / 64.f) + eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * gam[e] + bet[e];
            if (rcos && s >= n_text && ok) {
                // the reference rounds the normalised q/k to the model dtype before apply_rotary_emb (cogvideox_control.py:243-254)
                const float* cp = rcos + (long)(s - n_text) * 64 + sub * 8;
                const float* sp = rsin + (long)(s - n_text) * 64 + sub * 8;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float a = bf2f(f2bf(v[e])), c2 = bf2f(f2bf(v[e + 1]));
                    v[e] = a * cp[e] - c2 * sp[e];
                    v[e + 1] = c2 * cp[e + 1] + a * sp[e + 1];
                }
            }
            if (which == 0 && q_premul != 1.0f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= q_premul;   // softmax scale (and log2 e) folded into q: one rounding
            }
            if (ok) {
                uint4 o;
                o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
                *(uint4*)ptr = o;
            }
        }
    }
    if (!vT) {                                   // attention reads V in place (attn_fwd_v2): only the out-of-place copy remains
        if (src != qkv)
            for (int it = 0; it < 2; ++it) {
                const int s = s0 + wave * 16 + it * 8 + (lane >> 3);
                if (s < S) {
                    const long voff = ((long)b * S + s) * ld + 2 * H * 64 + h * 64 + sub * 8;
                    *(uint4*)(qkv + voff) = *(const uint4*)(src + voff);
                }
            }
        return;
    }
    // v tile -> LDS [token][d]
    for (int it = 0; it < 2; ++it) {
        const int tl = wave * 16 + it * 8 + (lane >> 3);
        const int s = s0 + tl;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (s < S) {
            const long voff = ((long)b * S + s) * ld + 2 * H * 64 + h * 64 + sub * 8;
            u = *(const uint4*)(src + voff);
            if (src != qkv) *(uint4*)(qkv + voff) = u;      // out-of-place form: the v third travels too (attention bwd reads it)
        }
        uint32_t* dst = (uint32_t*)&vt_s[tl][sub * 8];
        dst[0] = u.x; dst[1] = u.y; dst[2] = u.z; dst[3] = u.w;
    }
    __syncthreads();
    // write V^T rows: thread -> (d, 16-key group); the 16 keys of a group are permuted by vt_pos
    {
        const int d = tid >> 2, grp = tid & 3;
        bf16_t o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) o[vt_pos(k)] = vt_s[grp * 16 + k][d];
        bf16_t* dst = vT + ((long)(b * H + h) * 64 + d) * s_pad + s0 + grp * 16;
        uint4 a, c2;
        a.x = o[0] | ((uint32_t)o[1] << 16); a.y = o[2] | ((uint32_t)o[3] << 16);
        a.z = o[4] | ((uint32_t)o[5] << 16); a.w = o[6] | ((uint32_t)o[7] << 16);
        c2.x = o[8] | ((uint32_t)o[9] << 16); c2.y = o[10] | ((uint32_t)o[11] << 16);
        c2.z = o[12] | ((uint32_t)o[13] << 16); c2.w = o[14] | ((uint32_t)o[15] << 16);
        *(uint4*)dst = a;
        *(uint4*)(dst + 8) = c2;
    }
}

// Round 4, rows-per-wave form of the FULL case (xmap-free): the one-row kernel moves 4 KB of x and 4 KB of y per row through the CU's texture
// path - and 24 KB of FACTOR loads (gamma, beta, scale, shift: L2 hits, but 24 of the wave's 32 vector-memory instructions).  Here a wave
// owns R consecutive rows, keeps the four factor vectors of the current token group in registers (reloaded when the group changes: text
// rows, then per_group rows per frame) FOLDED into two - y = xhat * [gamma (1 + scale)] + [beta (1 + scale) + shift], fp32: one rounding order
// away from ln_mod_kernel's ((xhat gamma + beta)(1 + scale) + shift), far below the bf16 output step - and prefetches row r + 1 while row r is
// normalised.
#ifndef ORV_LNR_WAVES
#define ORV_LNR_WAVES 3
#endif
// PACKED: y in the P16 layout (include/orv_mi355.h orv_gemm_t: the A operand of gemm_d8) - chunk c of row r is 16-byte slot (c & 3) * 16 + (r & 15)
// of block (r >> 4, c >> 2).  A wave instruction then writes 64 scattered 16-byte pieces (four per KiB block) instead of 1 KiB of one row.
// A/B switches (tools/variants.sh): -DORV_LNR_NT_LOAD / -DORV_LNR_NT_STORE = non-temporal x loads / y stores
__device__ __forceinline__ uint4 lnr_load(const bf16_t* p) {
#ifdef ORV_LNR_NT_LOAD
    typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
    const nt_u32x4 v = __builtin_nontemporal_load((const nt_u32x4*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *(const uint4*)p;
#endif
}
__device__ __forceinline__ void lnr_store(bf16_t* p, const uint4 u) {
#ifdef ORV_LNR_NT_STORE
    typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(nt_u32x4{u.x, u.y, u.z, u.w}, (nt_u32x4*)p);
#else
    *(uint4*)p = u;
#endif
}
template <int CH, bool PACKED = false>
__global__ __launch_bounds__(256, ORV_LNR_WAVES) void ln_mod_rows_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy,
                                                             const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, long mod_b,
                                                             long mod_g, int seq, int n_text, int per_group, int rows, int D, float eps, int R) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = __builtin_amdgcn_readfirstlane(w * R);
    if (r0 >= rows) return;
    const int r1 = min(r0 + R, rows);
    const int nchunk = D >> 3;
    int ce[CH];
    bool ok[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        ok[i] = c < nchunk;
        ce[i] = min(c, nchunk - 1);
    }
    float gg[CH][8], hh[CH][8];      // y = xhat * gg + hh with gg = gamma (1 + scale), hh = beta (1 + scale) + shift of the current token group
    long cur_off = -1;
    uint4 raw[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) raw[i] = lnr_load(x + (long)r0 * ldx + ce[i] * 8);
    for (int row = r0; row < r1; ++row) {
        const int b = row / seq, sidx = row % seq;
        const long off = b * mod_b + orv_group_of(sidx, n_text, per_group) * mod_g;
        if (off != cur_off) {        // wave-uniform
            cur_off = off;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const uint4 gm = *(const uint4*)(gamma + ce[i] * 8), bt = *(const uint4*)(beta + ce[i] * 8);
                const float4 s0 = *(const float4*)(scale + off + ce[i] * 8), s1 = *(const float4*)(scale + off + ce[i] * 8 + 4);
                const float4 h0 = *(const float4*)(shift + off + ce[i] * 8), h1 = *(const float4*)(shift + off + ce[i] * 8 + 4);
                const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                const uint32_t gw[4] = {gm.x, gm.y, gm.z, gm.w}, bw[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = bf2f((e & 1) ? gw[e >> 1] >> 16 : gw[e >> 1] & 0xffff);
                    const float be = bf2f((e & 1) ? bw[e >> 1] >> 16 : bw[e >> 1] & 0xffff);
                    gg[i][e] = g * (1.0f + sv[e]);
                    hh[i][e] = fmaf(be, 1.0f + sv[e], hv[e]);
                }
                __builtin_amdgcn_sched_barrier(0);      // one chunk's factor loads at a time (rare path: keeps the register peak down)
            }
        }
        float v[CH][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const uint32_t wv[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][2 * e] = bf2f(wv[e] & 0xffff);
                v[i][2 * e + 1] = bf2f(wv[e] >> 16);
                s += ok[i] ? v[i][2 * e] + v[i][2 * e + 1] : 0.f;
            }
        }
        {   // row r + 1 is fetched into the registers row r was just unpacked from
            const int rn = min(row + 1, rows - 1);
#pragma unroll
            for (int i = 0; i < CH; ++i) raw[i] = lnr_load(x + (long)rn * ldx + ce[i] * 8);
        }
        const float mean = wave_sum_valu(s) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = ok[i] ? v[i][e] - mean : 0.f;
                sq += d * d;
            }
        }
        const float rstd = rsqrtf(wave_sum_valu(sq) / (float)D + eps);
        bf16_t* yr = y + (long)row * ldy;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf((v[i][e] - mean) * rstd, gg[i][e], hh[i][e]);
            uint4 u;
            u.x = pack2bf(o[0], o[1]); u.y = pack2bf(o[2], o[3]); u.z = pack2bf(o[4], o[5]); u.w = pack2bf(o[6], o[7]);
            if (PACKED) *(uint4*)(y + ((((long)(row >> 4) * (D >> 5) + (ce[i] >> 2)) << 9) + ((((ce[i] & 3) << 4) + (row & 15)) << 3))) = u;
            else lnr_store(yr + ce[i] * 8, u);       // lanes past the row end re-store the last chunk's identical bytes
        }
    }
}

}  // namespace

static int layernorm_modulate_impl(const void* x, int ldx, orv_rowmap_t xmap, void* y, int ldy, const void* gamma,
                                   const void* beta, const float* scale, const float* shift, long mod_b, long mod_g,
                                   orv_groups_t grp, int batch, int D, float eps, void* stream, bool packed) {
    ORV_REQUIRE(x && y, "orv_layernorm_modulate: null operand");
    ORV_REQUIRE(D % 8 == 0 && D <= 4096, "orv_layernorm_modulate: D=%d must be a multiple of 8 and <= 4096", D);
    ORV_REQUIRE(grp.seq > 0 && batch > 0, "orv_layernorm_modulate: empty problem");
    ORV_REQUIRE((scale == nullptr) == (shift == nullptr), "orv_layernorm_modulate: scale and shift go together");
    ORV_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "orv_layernorm_modulate: misaligned rows");
    const int rows = batch * grp.seq;
    const int ch = (D / 8 + 63) / 64;
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool full = gamma && beta && scale && shift;
    // rows-per-wave form: big FULL launches without a row map; R = rows per wave so that one round of resident waves (12 per CU) covers
    // the problem.  ORV_LN_ROWS=0: the one-row kernel everywhere (A/B); ORV_LN_ROWS=n: fixed R.
    static int rows_env = -2;
    if (rows_env == -2) { const char* e = getenv("ORV_LN_ROWS"); rows_env = e ? atoi(e) : -1; }
    ORV_REQUIRE(!packed || (full && xmap.rows == 0 && ch <= 4 && D % 32 == 0 && ldy == D),
                "orv_layernorm_modulate_packed: needs gamma, beta, scale, shift, no row map, D %% 32 == 0, D <= 2048 and ldy == D");
    // Every full, map-free launch with D <= 2048 takes the rows kernel, whatever the row count (below 2048 rows the one-row kernel
    // rounded ((xhat gamma + beta)(1 + scale) + shift) in another order than the folded form, so a clip could differ in the last bit between
    // a batch size on the packed path and one on the row-major path).
    if (full && xmap.rows == 0 && (packed || rows_env != 0) && ch <= 4) {
        const int slots = 256 * 4 * ORV_LNR_WAVES;
        const int R = rows_env > 0 ? rows_env : max(rows >= 2048 ? 2 : 1, (rows + slots - 1) / slots);
        dim3 g2(((rows + R - 1) / R + 3) / 4);
#define ORV_LNR_CASE(C)                                                                                                \
        if (packed)                                                                                                    \
            hipLaunchKernelGGL((ln_mod_rows_kernel<C, true>), g2, block, 0, st, (const bf16_t*)x, (long)ldx, (bf16_t*)y, (long)ldy, \
                               (const bf16_t*)gamma, (const bf16_t*)beta, scale, shift, mod_b, mod_g, grp.seq, grp.n_text, \
                               grp.per_group, rows, D, eps, R);                                                        \
        else                                                                                                           \
            hipLaunchKernelGGL((ln_mod_rows_kernel<C>), g2, block, 0, st, (const bf16_t*)x, (long)ldx, (bf16_t*)y, (long)ldy, \
                               (const bf16_t*)gamma, (const bf16_t*)beta, scale, shift, mod_b, mod_g, grp.seq, grp.n_text, \
                               grp.per_group, rows, D, eps, R)
        if (ch <= 1) { ORV_LNR_CASE(1); }
        else if (ch <= 2) { ORV_LNR_CASE(2); }
        else { ORV_LNR_CASE(4); }
#undef ORV_LNR_CASE
        return orv_check_launch("orv_layernorm_modulate");
    }
#define ORV_LN_CASE(C)                                                                                                 \
    if (full)                                                                                                          \
        hipLaunchKernelGGL((ln_mod_kernel<C, true>), grid, block, 0, st, (const bf16_t*)x, (long)ldx, (bf16_t*)y, (long)ldy, \
                           (const bf16_t*)gamma, (const bf16_t*)beta, scale, shift, mod_b, mod_g, grp.seq, grp.n_text, \
                           grp.per_group, rows, D, eps, xmap);                                                         \
    else                                                                                                               \
        hipLaunchKernelGGL((ln_mod_kernel<C, false>), grid, block, 0, st, (const bf16_t*)x, (long)ldx, (bf16_t*)y, (long)ldy, \
                           (const bf16_t*)gamma, (const bf16_t*)beta, scale, shift, mod_b, mod_g, grp.seq, grp.n_text, \
                           grp.per_group, rows, D, eps, xmap)
    if (ch <= 1) { ORV_LN_CASE(1); }
    else if (ch <= 2) { ORV_LN_CASE(2); }
    else if (ch <= 4) { ORV_LN_CASE(4); }
    else if (ch <= 6) { ORV_LN_CASE(6); }
    else { ORV_LN_CASE(8); }
#undef ORV_LN_CASE
    return orv_check_launch("orv_layernorm_modulate");
}

extern "C" int orv_layernorm_modulate(const void* x, int ldx, orv_rowmap_t xmap, void* y, int ldy, const void* gamma,
                                      const void* beta, const float* scale, const float* shift, long mod_b, long mod_g,
                                      orv_groups_t grp, int batch, int D, float eps, void* stream) {
    return layernorm_modulate_impl(x, ldx, xmap, y, ldy, gamma, beta, scale, shift, mod_b, mod_g, grp, batch, D, eps, stream, false);
}
// y in the packed P16 layout (orv_packed_rows(batch * seq) x D; the A operand of the d8 GEMM): the full case only (gamma, beta, scale, shift,
// no row map), D % 32 == 0, D <= 2048
extern "C" int orv_layernorm_modulate_packed(const void* x, int ldx, void* y, const void* gamma, const void* beta, const float* scale,
                                             const float* shift, long mod_b, long mod_g, orv_groups_t grp, int batch, int D, float eps,
                                             void* stream) {
    orv_rowmap_t none{0, 0, 0};
    return layernorm_modulate_impl(x, ldx, none, y, D, gamma, beta, scale, shift, mod_b, mod_g, grp, batch, D, eps, stream, true);
}

extern "C" int orv_qkv_prep_from(const void* src, void* qkv, void* vT, const void* gq, const void* bq, const void* gk,
                                 const void* bk, const float* rope_cos, const float* rope_sin, int B, int S, int H, int n_text,
                                 int s_pad, float eps, float q_premul, void* stream);

extern "C" int orv_qkv_prep(void* qkv, void* vT, const void* gq, const void* bq, const void* gk, const void* bk,
                            const float* rope_cos, const float* rope_sin, int B, int S, int H, int n_text, int s_pad,
                            float eps, float q_premul, void* stream) {
    return orv_qkv_prep_from(qkv, qkv, vT, gq, bq, gk, bk, rope_cos, rope_sin, B, S, H, n_text, s_pad, eps, q_premul, stream);
}

extern "C" int orv_qkv_prep_from(const void* src, void* qkv, void* vT, const void* gq, const void* bq, const void* gk,
                                 const void* bk, const float* rope_cos, const float* rope_sin, int B, int S, int H, int n_text,
                                 int s_pad, float eps, float q_premul, void* stream) {
    ORV_REQUIRE(src && qkv, "orv_qkv_prep: null operand");      // vT may be NULL: no V^T copy (attention reads V in place)
    ORV_REQUIRE(B > 0 && S > 0 && H > 0, "orv_qkv_prep: empty problem");
    ORV_REQUIRE(s_pad % 64 == 0 && s_pad >= S && s_pad == ((S + 63) / 64) * 64,
                "orv_qkv_prep: s_pad=%d must be S=%d rounded up to 64", s_pad, S);
    ORV_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), "orv_qkv_prep: cos and sin go together");
    dim3 grid(s_pad / 64, H, B);
    hipLaunchKernelGGL(qkv_prep_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)qkv, (bf16_t*)vT,
                       (const bf16_t*)gq, (const bf16_t*)bq, (const bf16_t*)gk, (const bf16_t*)bk, rope_cos, rope_sin, S,
                       H, n_text, s_pad, eps, q_premul);
    return orv_check_launch("orv_qkv_prep");
}

// ---------------------------------------------------------------------------------------------------------------
// orv_modulation_tables: every AdaLN modulation linear of the forward in ONE launch.
//   out[tab][b][1+t][:] = W_tab[0:width]      . silu(temb[b] + a[b,t]) + bias[0:width]       (video rows, per frame)
//   out[tab][b][0][:]   = W_tab[width:2width] . silu(temb[b])          + bias[width:2width]  (text rows, if `text`)
// (cogvideox_control.py:117-130 "partially forward self.linear twice", :172).  temb/a are the same for all 61 norms of a
// step, so this is a [rows<=32, E] x [E, 61*6D] GEMM that streams ~0.7 GB of weights once: HBM-bound.  Each wave keeps
// the conditioning fragments (B operand, rows on the lane axis) in registers and walks 32-column weight blocks with
// v_mfma_f32_32x32x16_bf16, weight fragments loaded straight from global (each weight byte is used exactly once).
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct ModArgs {
    const bf16_t* temb;       // [B, E]
    const bf16_t* act;        // [B, T, E] or null (then T == 1 and the video rows use silu(temb))
    const bf16_t* const* W;   // n_tab pointers to [width * (1 + text), E]
    const bf16_t* const* bias;
    float* out;               // [n_tab, B, 1 + T, width]
    int n_tab, B, T, E, width, text;
    int vis_tiles;            // ceil(B*T / 32)
};

// conditioning fragments of a row: lane holds silu(temb[b] (+ act[b, t]))[ks * 16 + hi * 8 .. + 8].  Loads are UNCONDITIONAL (without an
// action embedding the second stream re-reads temb and is ignored) and issued eight k-steps at a time: with `if (ap) load` per k-step hipcc
// branched around every load and waited vmcnt(0) behind it - 32 dependent round trips (~40 us) at the head of every workgroup (round 4).
template <int KS>
__device__ __forceinline__ void mod_cond_frags(const ModArgs& p, int b, int t, bool is_text, bool rvalid, int hi, bf16x8 (&cf)[KS]) {
    const bool has_act = !is_text && p.act != nullptr;
    const bf16_t* tp = p.temb + (long)b * p.E + hi * 8;
    const bf16_t* ap = has_act ? p.act + ((long)b * p.T + t) * p.E + hi * 8 : tp;
    constexpr int BATCH = KS < 8 ? KS : 8;
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += BATCH) {
        uint4 tu[BATCH], au[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) { tu[j] = *(const uint4*)(tp + (k0 + j) * 16); au[j] = *(const uint4*)(ap + (k0 + j) * 16); }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const uint32_t tw[4] = {tu[j].x, tu[j].y, tu[j].z, tu[j].w}, aw[4] = {au[j].x, au[j].y, au[j].z, au[j].w};
            union { bf16x8 v; uint32_t u[4]; } c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = bf2f(tw[e] & 0xffff), hi2 = bf2f(tw[e] >> 16);
                // the reference adds in the model dtype (bf16) before SiLU
                const float lo_a = bf2f(f2bf(lo + bf2f(aw[e] & 0xffff))), hi_a = bf2f(f2bf(hi2 + bf2f(aw[e] >> 16)));
                lo = has_act ? lo_a : lo;
                hi2 = has_act ? hi_a : hi2;
                c.u[e] = rvalid ? pack2bf(silu(lo), silu(hi2)) : 0u;
            }
            cf[k0 + j] = c.v;
        }
    }
}

template <int KS>  // KS = E / 16 k-steps
__global__ __launch_bounds__(256) void mod_tables_kernel(const ModArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int tab = blockIdx.y;
    const bool is_text = (int)blockIdx.z >= p.vis_tiles;
    const int rtile = is_text ? blockIdx.z - p.vis_tiles : blockIdx.z;
    const int nrows = is_text ? p.B : p.B * p.T;
    const int r = rtile * 32 + l31;
    const bool rvalid = r < nrows;
    const int b = rvalid ? (is_text ? r : r / p.T) : 0;
    const int t = rvalid && !is_text ? r % p.T : 0;
    // conditioning fragments: lane holds cond[r][ks*16 + hi*8 .. +8]
    bf16x8 cf[KS];
    mod_cond_frags<KS>(p, b, t, is_text, rvalid, hi, cf);
    const bf16_t* Wt = p.W[tab] + (is_text ? (long)p.width * p.E : 0);
    const bf16_t* bt = p.bias ? (p.bias[tab] ? p.bias[tab] + (is_text ? p.width : 0) : nullptr) : nullptr;
    const int G = 1 + p.T;
    float* orow = p.out + (((long)tab * p.B + b) * G + (is_text ? 0 : 1 + t)) * p.width;
    const int nblocks = p.width / 32;
    for (int nb = blockIdx.x * 4 + wave; nb < nblocks; nb += gridDim.x * 4) {
        const bf16_t* wp = Wt + (long)(nb * 32 + l31) * p.E + hi * 8;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        constexpr int HALF = KS >= 16 ? 16 : KS;
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += HALF) {
            bf16x8 wf[HALF];
#pragma unroll
            for (int j = 0; j < HALF; ++j) wf[j] = *(const bf16x8*)(wp + (k0 + j) * 16);
#pragma unroll
            for (int j = 0; j < HALF; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], cf[k0 + j], acc, 0, 0, 0);
        }
        if (rvalid) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb * 32 + q * 8 + hi * 4;
                float4 o = make_float4(acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
                if (bt) {
                    const uint2 bb = *(const uint2*)(bt + n);
                    o.x += bf2f(bb.x & 0xffff); o.y += bf2f(bb.x >> 16); o.z += bf2f(bb.y & 0xffff); o.w += bf2f(bb.y >> 16);
                }
                *(float4*)(orow + n) = o;
            }
        }
    }
}

}  // namespace

namespace {
// Round 4, E = 512: the weight blocks go through LDS.  mod_tables_kernel loads a weight fragment straight into the MFMA layout - lane =
// (weight row, 16-byte k half): one wave instruction touches 32 rows x 32 B, i.e. 32 cache lines of which it uses a quarter, four
// instructions per line - and ran at 2.5 TB/s.  Here a wave streams its 32-row weight block in eight 64-k slabs of 32 rows x 128 B with the
// LDS-DMA (8 rows x one full 128-byte line per instruction; chunk ^ ((row >> 1) & 7) on the source: the image gemm_kernel reads conflict-free
// with ds_read_b128), four slabs in flight in a wave-private 16-KiB ring (no barrier: a wave reads only what it staged), and keeps the
// accumulators of its (up to four) blocks in registers until its stream has drained, so no store sits between the counted vmcnt waits.
// LDS-DMA from inline asm (M0 written in the same statement): behind the builtin hipcc treats every DMA as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of the next ds_read - the ring would drain at every slab.  The waits are the counted ones below.
__device__ __forceinline__ void mt_glds16(const void* gsrc, const void* lds_dst) {
    unsigned keep;
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(d) : "memory");
}
constexpr int MT_SLABS = 4;          // slabs in flight per wave
constexpr int MT_MAXBLK = 4;         // 32-column blocks per wave
__global__ __launch_bounds__(256) void mod_tables_lds_kernel(const ModArgs p) {
    constexpr int KS = 32, SPB = 8;                                   // E = 512: 32 k-steps of 16, 8 slabs of 64 k per block
    __shared__ __attribute__((aligned(16))) char smem[4 * MT_SLABS * 4096];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
    const int tab = blockIdx.y;
    const bool is_text = (int)blockIdx.z >= p.vis_tiles;
    const int rtile = is_text ? blockIdx.z - p.vis_tiles : blockIdx.z;
    const int nrows = is_text ? p.B : p.B * p.T;
    const int r = rtile * 32 + l31;
    const bool rvalid = r < nrows;
    const int b = rvalid ? (is_text ? r : r / p.T) : 0;
    const int t = rvalid && !is_text ? r % p.T : 0;
    bf16x8 cf[KS];
    mod_cond_frags<KS>(p, b, t, is_text, rvalid, hi, cf);
    const bf16_t* Wt = p.W[tab] + (is_text ? (long)p.width * p.E : 0);
    const bf16_t* bt = p.bias ? (p.bias[tab] ? p.bias[tab] + (is_text ? p.width : 0) : nullptr) : nullptr;
    const int G = 1 + p.T;
    float* orow = p.out + (((long)tab * p.B + b) * G + (is_text ? 0 : 1 + t)) * p.width;
    const int nblocks = p.width / 32;
    const int nb0 = blockIdx.x * 4 + wave, stride = gridDim.x * 4;
    const int nblk = nb0 < nblocks ? (nblocks - nb0 + stride - 1) / stride : 0;      // <= MT_MAXBLK (host)
    const int total = nblk * SPB;
    char* const ring = smem + wave * (MT_SLABS * 4096);
    // DMA source of this lane inside a slab: piece j = rows 8 j + (lane >> 3), 16-byte chunk (lane & 7) ^ ((row >> 1) & 7)
    int soff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * j + (lane >> 3);
        soff[j] = row * p.E + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    auto issue = [&](int sl) {
        const int bi = sl / SPB, ks = sl % SPB;
        const bf16_t* wp = Wt + (long)((nb0 + bi * stride) * 32) * p.E + ks * 64;
        char* const d = ring + (sl % MT_SLABS) * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) mt_glds16(wp + soff[j], d + j * 1024);
    };
    f32x16 acc[MT_MAXBLK];
#pragma unroll
    for (int i = 0; i < MT_MAXBLK; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int sl = 0; sl < MT_SLABS - 1; ++sl)
        if (sl < total) issue(sl);
    const int rd_off = l31 * 128;
#pragma unroll
    for (int bi = 0; bi < MT_MAXBLK; ++bi) {
        if (bi >= nblk) break;                    // wave-uniform
#pragma unroll
        for (int ks = 0; ks < SPB; ++ks) {
            const int sl = bi * SPB + ks;
            if (sl + MT_SLABS - 1 < total) {      // keep MT_SLABS - 1 slabs ahead; the slot being refilled was read an iteration ago
                issue(sl + MT_SLABS - 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (MT_SLABS - 1)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const char* sb = ring + (sl % MT_SLABS) * 4096 + rd_off;
            bf16x8 wf[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) wf[k4] = *(const bf16x8*)(sb + (((k4 * 2 + hi) ^ sw) << 4));
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) acc[bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k4], cf[ks * 4 + k4], acc[bi], 0, 0, 0);
            // the reads of this slab are complete (their values fed the MFMAs) before the NEXT iteration's DMA may overwrite slot (sl - 1) % 4
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (rvalid) {
#pragma unroll
        for (int bi = 0; bi < MT_MAXBLK; ++bi) {
            if (bi >= nblk) break;
            const int nb = nb0 + bi * stride;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb * 32 + q * 8 + hi * 4;
                float4 o = make_float4(acc[bi][q * 4], acc[bi][q * 4 + 1], acc[bi][q * 4 + 2], acc[bi][q * 4 + 3]);
                if (bt) {
                    const uint2 bb = *(const uint2*)(bt + n);
                    o.x += bf2f(bb.x & 0xffff); o.y += bf2f(bb.x >> 16); o.z += bf2f(bb.y & 0xffff); o.w += bf2f(bb.y >> 16);
                }
                *(float4*)(orow + n) = o;
            }
        }
    }
}
}  // namespace

extern "C" int orv_modulation_tables(const void* temb, const void* action_emb, const void* const* W,
                                     const void* const* bias, float* out, int n_tab, int B, int T, int E, int width,
                                     int text, void* stream) {
    ORV_REQUIRE(temb && W && out, "orv_modulation_tables: null operand");
    ORV_REQUIRE(n_tab > 0 && B > 0 && T > 0 && width > 0, "orv_modulation_tables: empty problem");
    ORV_REQUIRE(width % 32 == 0, "orv_modulation_tables: width=%d must be a multiple of 32", width);
    ORV_REQUIRE(E == 64 || E == 128 || E == 256 || E == 512, "orv_modulation_tables: E=%d unsupported (64/128/256/512)", E);
    ModArgs a;
    a.temb = (const bf16_t*)temb; a.act = (const bf16_t*)action_emb; a.W = (const bf16_t* const*)W;
    a.bias = (const bf16_t* const*)bias; a.out = out; a.n_tab = n_tab; a.B = B; a.T = T; a.E = E; a.width = width;
    a.text = text; a.vis_tiles = (B * T + 31) / 32;
    const int txt_tiles = text ? (B + 31) / 32 : 0;
    const int nblocks = width / 32;
    dim3 grid(max(1, min((nblocks + 11) / 12, 64)), n_tab, a.vis_tiles + txt_tiles);
    hipStream_t st = (hipStream_t)stream;
    static int lds_env = -1;     // ORV_MOD_TABLES_LDS=0: the direct-load kernel for E = 512 too (A/B)
    if (lds_env < 0) { const char* e = getenv("ORV_MOD_TABLES_LDS"); lds_env = (e && atoi(e) == 0) ? 0 : 1; }
    if (E == 512 && lds_env) {
        dim3 g2((nblocks + 4 * MT_MAXBLK - 1) / (4 * MT_MAXBLK), n_tab, a.vis_tiles + txt_tiles);
        hipLaunchKernelGGL(mod_tables_lds_kernel, g2, dim3(256), 0, st, a);
        return orv_check_launch("orv_modulation_tables");
    }
    switch (E) {
        case 64: hipLaunchKernelGGL(mod_tables_kernel<4>, grid, dim3(256), 0, st, a); break;
        case 128: hipLaunchKernelGGL(mod_tables_kernel<8>, grid, dim3(256), 0, st, a); break;
        case 256: hipLaunchKernelGGL(mod_tables_kernel<16>, grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(mod_tables_kernel<32>, grid, dim3(256), 0, st, a); break;
    }
    return orv_check_launch("orv_modulation_tables");
}
