// Backward (training) kernels that are HBM-bound: transposes feeding the dgrad/wgrad GEMMs, LayerNorm-modulate backward,
// gated-residual backward, bias/column sums, small-row weight/data gradients of the conditioning MLPs, fused AdamW.
// Gradients of the reference flow through torch autograd over the same ops (train_cogvideox_control_to_video_sft.py:1093);
// here every op of the forward path (cogvideox_control.py:394-445, :60-150) has its hand-written adjoint.
#include "common.hpp"

namespace {

// ---- dst[c, r] = src[r, c] for a [R, C] bf16 matrix (row stride lds), dst row stride ldd >= R; columns r in [R, ldd) of dst
//      are zero-filled (K padding for the wgrad GEMM).  64x64 tiles through LDS, 16-byte global accesses both ways.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, long lds_, bf16_t* __restrict__ dst,
                                                        long ldd, int R, int C) {
    __shared__ bf16_t tile[64][72];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    // load 64 rows x 64 cols: thread -> (row = tid/8 + 32*it, 8-col chunk = tid%8)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = (tid >> 3) + 32 * it, ch = tid & 7;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (r0 + r < R && c0 + ch * 8 < C) u = *(const uint4*)(src + (long)(r0 + r) * lds_ + c0 + ch * 8);
        *(uint4*)&tile[r][ch * 8] = u;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (tid >> 3) + 32 * it, ch = tid & 7;   // output row c (source column), 8 source rows per chunk
        if (c0 + c >= C) continue;
        bf16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = tile[ch * 8 + e][c];
        uint4 u;
        u.x = o[0] | ((uint32_t)o[1] << 16); u.y = o[2] | ((uint32_t)o[3] << 16);
        u.z = o[4] | ((uint32_t)o[5] << 16); u.w = o[6] | ((uint32_t)o[7] << 16);
        if (r0 + ch * 8 < ldd) *(uint4*)(dst + (long)(c0 + c) * ldd + r0 + ch * 8) = u;   // rows >= R were loaded as zeros
    }
}

// ---- the same transpose with the column sums of src taken on the way (bias gradient of a linear layer next to the dY^T its
//      weight gradient needs: one read of dY instead of two).  A workgroup walks RT consecutive 64-row tiles of one 64-column
//      strip, so a column gets ceil(R / (64 RT)) fp32 atomics - as many as orv_colsum's slabs.
template <int RT>
__global__ __launch_bounds__(256) void transpose_colsum_kernel(const bf16_t* __restrict__ src, long lds_, bf16_t* __restrict__ dst,
                                                               long ldd, int R, int C, float* __restrict__ csum) {
    __shared__ bf16_t tile[2][64][72];
    __shared__ float red[32][65];
    const int c0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lr = tid >> 3, ch = tid & 7;
    const bool col_ok = c0 + ch * 8 < C;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint4 u[2];
    auto fetch = [&](int t) {
        const int r0 = (blockIdx.x * RT + t) * 64;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = lr + 32 * it;
            u[it] = (r0 + r < R && col_ok) ? *(const uint4*)(src + (long)(r0 + r) * lds_ + c0 + ch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    fetch(0);
#pragma unroll 1
    for (int t = 0; t < RT; ++t) {
        const int r0 = (blockIdx.x * RT + t) * 64;
        if (r0 >= ldd) break;                              // whole-workgroup uniform
        bf16_t (*tl)[72] = tile[t & 1];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            *(uint4*)&tl[lr + 32 * it][ch * 8] = u[it];
            const uint32_t w[4] = {u[it].x, u[it].y, u[it].z, u[it].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[2 * e] += bf2f(w[e] & 0xffff); cs[2 * e + 1] += bf2f(w[e] >> 16); }
        }
        if (t + 1 < RT) fetch(t + 1);                      // next tile's loads fly over this tile's LDS round trip
        __syncthreads();                                   // (two LDS tiles: the previous tile's readers are past their reads)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = lr + 32 * it;
            if (c0 + c >= C) continue;
            bf16_t o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = tl[ch * 8 + e][c];
            uint4 v;
            v.x = o[0] | ((uint32_t)o[1] << 16); v.y = o[2] | ((uint32_t)o[3] << 16);
            v.z = o[4] | ((uint32_t)o[5] << 16); v.w = o[6] | ((uint32_t)o[7] << 16);
            if (r0 + ch * 8 < ldd) *(uint4*)(dst + (long)(c0 + c) * ldd + r0 + ch * 8) = v;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[lr][ch * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 64 && c0 + tid < C) {
        float s_ = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s_ += red[r][tid];
        atomicAdd(csum + c0 + tid, s_);
    }
}

// ---- out[c] += sum_r src[r, c]  (bias gradients).  One block = 512 columns x a slab of rows: thread -> (8-column chunk,
//      row phase 0..3), 16-byte loads, four rows in flight; LDS combine of the 4 phases, fp32 atomics across slabs.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ src, long ld, float* __restrict__ out, int R,
                                                     int C, int rows_per_block) {
    __shared__ float red[3][64][8];
    const int cc = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + cc) * 8;
    const int rb = blockIdx.y * rows_per_block, re = min(R, rb + rows_per_block);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < C) {
        for (int r = rb + ph; r < re; r += 16) {
            uint4 u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = (r + 4 * k < re) ? *(const uint4*)(src + (long)(r + 4 * k) * ld + c0) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t w[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[2 * e] += bf2f(w[e] & 0xffff); s[2 * e + 1] += bf2f(w[e] >> 16); }
            }
        }
    }
    if (ph > 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[ph - 1][cc][e] = s[e];
    __syncthreads();
    if (ph == 0 && c0 < C)
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(out + c0 + e, s[e] + red[0][cc][e] + red[1][cc][e] + red[2][cc][e]);
}

// out[b, g, j] += sum over the workgroups of (b, g) of part[blk * stride + sel * D + j]   (sel = sel0 -> out0, sel1 -> out1)
// The workgroups of one (batch, token group) are consecutive and each table entry has one writer: plain +=.
__global__ __launch_bounds__(256) void groups_reduce_kernel(const float* __restrict__ part, long stride, int sel0, int sel1,
                                                            int D, float* __restrict__ out0, float* __restrict__ out1,
                                                            long mod_b, long mod_g, int bt, int bg, int bpb, int ngroups) {
    // 64 columns x 4 slices of the workgroup list per block; the slices are combined through LDS in a fixed order
    // (deterministic, and four loads in flight per column instead of one thread walking ~38 partials alone)
    __shared__ float red[2][3][64];
    const int jl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const int b = blockIdx.y / (1 + ngroups), g = blockIdx.y % (1 + ngroups);
    const int first = b * bpb + (g == 0 ? 0 : bt + (g - 1) * bg), count = g == 0 ? bt : bg;
    if (count == 0) return;
    float s0 = 0.f, s1 = 0.f;
    if (j < D)
        for (int k = sl; k < count; k += 4) {
            s0 += part[(long)(first + k) * stride + (long)sel0 * D + j];
            if (out1) s1 += part[(long)(first + k) * stride + (long)sel1 * D + j];
        }
    if (sl > 0) { red[0][sl - 1][jl] = s0; red[1][sl - 1][jl] = s1; }
    __syncthreads();
    if (sl == 0 && j < D) {
        const long off = b * mod_b + g * mod_g + j;
        out0[off] += s0 + red[0][0][jl] + red[0][1][jl] + red[0][2][jl];
        if (out1) out1[off] += s1 + red[1][0][jl] + red[1][1][jl] + red[1][2][jl];
    }
}

// ---- gated residual backward (cogvideox_control.py:419-421,442-443):  out = x + gate[b,g] * y
//      dy = gate * dout (bf16);  dgate[b,g,:] += sum_{rows of group} dout * y.
//      Same decomposition as the LayerNorm adjoint below: a workgroup owns GRB rows of ONE (batch, token group), its 256
//      threads split the columns, four rows of loads in flight per thread, the column sums stay in registers and leave as
//      one fp32 atomic per column and workgroup.
constexpr int GRB = 32;
template <int CH>
__global__ __launch_bounds__(256) void gated_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                        const float* __restrict__ gate, float* __restrict__ dgate,
                                                        bf16_t* __restrict__ dy, long mod_b, long mod_g, int seq, int n_text,
                                                        int per_group, int D, int bt, int bg, int bpb, float* __restrict__ part) {
    const int tid = threadIdx.x, nchunk = D >> 3;
    const int b = blockIdx.x / bpb, r = blockIdx.x % bpb;
    int g, s0, s1;
    if (r < bt) { g = 0; s0 = r * GRB; s1 = min(n_text, s0 + GRB); }
    else {
        const int q = r - bt, gi = q / bg, ck = q % bg;
        const int gsz = per_group > 0 ? per_group : seq - n_text;
        const int gstart = n_text + gi * gsz;
        g = 1 + gi;
        s0 = gstart + ck * GRB;
        s1 = min(min(gstart + gsz, seq), s0 + GRB);
    }
    const long off = b * mod_b + g * mod_g;
    float gg[CH][8], acc[CH][8];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = tid + 256 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) { gg[i][e] = c < nchunk ? gate[off + c * 8 + e] : 0.f; acc[i][e] = 0.f; }
    }
    for (int rb = s0; rb < s1; rb += 4) {
        uint4 ud[4][CH], uy[4][CH];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                // unconditional, clamped (round 4): behind `if (c < nchunk)` hipcc branched around every load and waited for it - the eight
                // loads of a pass went out one round trip after the other (cdna guide, ".s-level traps" (c))
                const int c = min(tid + 256 * i, nchunk - 1);
                const long row = (long)b * seq + min(rb + k, s1 - 1);
                ud[k][i] = *(const uint4*)(dout + row * D + c * 8);
                uy[k][i] = *(const uint4*)(y + row * D + c * 8);
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (rb + k >= s1) continue;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = tid + 256 * i;
                if (c >= nchunk) continue;
                const uint32_t wd[4] = {ud[k][i].x, ud[k][i].y, ud[k][i].z, ud[k][i].w};
                const uint32_t wy[4] = {uy[k][i].x, uy[k][i].y, uy[k][i].z, uy[k][i].w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = bf2f(wd[e] & 0xffff), d1 = bf2f(wd[e] >> 16);
                    acc[i][2 * e] += d0 * bf2f(wy[e] & 0xffff);
                    acc[i][2 * e + 1] += d1 * bf2f(wy[e] >> 16);
                    o[e] = pack2bf(d0 * gg[i][2 * e], d1 * gg[i][2 * e + 1]);
                }
                *(uint4*)(dy + ((long)b * seq + rb + k) * D + c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = tid + 256 * i;
        if (c < nchunk) {
            float* pg = part + (long)blockIdx.x * D + c * 8;      // per-workgroup partial, summed by groups_reduce_kernel
            *(float4*)pg = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            *(float4*)(pg + 4) = make_float4(acc[i][4], acc[i][5], acc[i][6], acc[i][7]);
        }
    }
}

// ---- LayerNorm-modulate backward.  Forward: xh = (x - mu) rstd ; z = xh*gamma + beta ; y = z*(1+sc) + sh.
//      Given dy: dxh = dy*(1+sc)*gamma ; dx = rstd*(dxh - mean(dxh) - xh*mean(dxh*xh)) (+ dres).
//      All rows of a workgroup belong to ONE (batch, token group), so (1+sc) is a per-column constant of the block and only
//      two column sums are needed:  A1 = sum dy, A2 = sum dy*xh  ->  dshift = A1, dscale = gamma*A2 + beta*A1,
//      dbeta = (1+sc)*A1, dgamma = (1+sc)*A2.
//      Layout: the 256 threads split the COLUMNS (thread -> 8-column chunks tid, tid+256), every thread walks the block's rows
//      four at a time (8 independent 16-byte loads in flight per thread); the three row reductions (mean, variance,
//      the two dx means) go wave_sum -> LDS -> all threads, one barrier each for the four rows together.  Column sums stay
//      private to a thread: no cross-wave reduction.  dscale/dshift: fp32 atomics (~38 blocks per table row);
//      dgamma/dbeta: per-block partials + reduce kernel (every block would hit the same D addresses otherwise).
// rows per workgroup / rows in flight per pass, from a sweep at the 2B training shape (whole call incl. the two reductions):
// (16, 4) 105-117 us, (16, 2) 76 us, (16, 1) 76 us, (12, 1) 77 us, (20, 2) 78 us, (24, 2) 84 us, (8, 2) 108 us, (32, 4) 105 us
constexpr int LNB_RB = 16;
constexpr int LNB_R = 2;

struct LnBwdArgs {
    const bf16_t *dy, *x, *dres; bf16_t* dx;
    const bf16_t *gamma, *beta;
    const float* scale; float *dscale, *dshift;
    float* part;           // [blocks][4][D] partial column sums (dgamma | dbeta | dscale | dshift)
    int want_gb;
    long mod_b, mod_g;
    int seq, n_text, per_group, D;
    float eps;
    orv_rowmap_t xmap;
    int bt, bg, bpb;       // blocks per text group / per video group / per batch element
};

template <int CH>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const LnBwdArgs p) {
    __shared__ float red[3][2][LNB_R][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = p.D, nchunk = D >> 3;
    // ---- block -> (batch, group, rows [s0, s1) of the sequence) ----
    const int b = blockIdx.x / p.bpb, r = blockIdx.x % p.bpb;
    int g, s0, s1;
    if (r < p.bt) { g = 0; s0 = r * LNB_RB; s1 = min(p.n_text, s0 + LNB_RB); }
    else {
        const int q = r - p.bt, gi = q / p.bg, ck = q % p.bg;
        const int gsz = p.per_group > 0 ? p.per_group : p.seq - p.n_text;
        const int gstart = p.n_text + gi * gsz;
        g = 1 + gi;
        s0 = gstart + ck * LNB_RB;
        s1 = min(min(gstart + gsz, p.seq), s0 + LNB_RB);
    }
    const long off = p.scale ? b * p.mod_b + g * p.mod_g : 0;

    float gam[CH][8], osc[CH][8], a1[CH][8], a2[CH][8];   // gamma, 1 + scale, the two column sums
    bool live[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = tid + 256 * i;
        live[i] = c < nchunk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gam[i][e] = (p.gamma && live[i]) ? bf2f(p.gamma[c * 8 + e]) : 1.f;
            osc[i][e] = (p.scale && live[i]) ? 1.f + p.scale[off + c * 8 + e] : 1.f;
            a1[i][e] = a2[i][e] = 0.f;
        }
    }
    const float invD = 1.f / (float)D;

    for (int rb = s0; rb < s1; rb += LNB_R) {
        const int nr = min(LNB_R, s1 - rb);
        float xv[LNB_R][CH][8], dv[LNB_R][CH][8];
        long xrow[LNB_R];
        float ps[LNB_R];
#pragma unroll
        for (int k = 0; k < LNB_R; ++k) {
            const long row = (long)b * p.seq + rb + min(k, nr - 1);
            xrow[k] = p.xmap.rows > 0 ? (row / p.xmap.rows) * p.xmap.bstride + p.xmap.off + row % p.xmap.rows : row;
            ps[k] = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                // unconditional, clamped loads (round 4; the values of dead chunks / rows are zeroed afterwards): behind
                // `if (live && k < nr)` every load sat in its own branch with its own wait
                const int c = min(tid + 256 * i, nchunk - 1);
                const bool on = live[i] && k < nr;
                const uint4 ux = *(const uint4*)(p.x + xrow[k] * D + c * 8);
                const uint4 ud = *(const uint4*)(p.dy + row * D + c * 8);
                const uint32_t wx[4] = {ux.x, ux.y, ux.z, ux.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xv[k][i][2 * e] = on ? bf2f(wx[e] & 0xffff) : 0.f; xv[k][i][2 * e + 1] = on ? bf2f(wx[e] >> 16) : 0.f;
                    dv[k][i][2 * e] = on ? bf2f(wd[e] & 0xffff) : 0.f; dv[k][i][2 * e + 1] = on ? bf2f(wd[e] >> 16) : 0.f;
                    ps[k] += xv[k][i][2 * e] + xv[k][i][2 * e + 1];
                }
            }
        }
        // block-wide sums of LNB_R values per phase: wave_sum, one LDS slot per wave, one barrier (3 phase buffers, so the
        // next phase never overwrites what a slower wave is still reading)
#define ORV_BLOCK_SUM(PH, SLOT, V)                                                                   \
        {                                                                                            \
            _Pragma("unroll") for (int k = 0; k < LNB_R; ++k) {                                      \
                const float w_ = wave_sum_valu(V[k]);                                                     \
                if (lane == 0) red[PH][SLOT][k][wave] = w_;                                          \
            }                                                                                        \
        }
#define ORV_BLOCK_GET(PH, SLOT, k) (red[PH][SLOT][k][0] + red[PH][SLOT][k][1] + red[PH][SLOT][k][2] + red[PH][SLOT][k][3])
        ORV_BLOCK_SUM(0, 0, ps)
        __syncthreads();
        float mean[LNB_R], rstd[LNB_R];
#pragma unroll
        for (int k = 0; k < LNB_R; ++k) {
            mean[k] = ORV_BLOCK_GET(0, 0, k) * invD;
            ps[k] = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (live[i])
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = xv[k][i][e] - mean[k]; ps[k] += d * d; }
        }
        ORV_BLOCK_SUM(1, 0, ps)
        __syncthreads();
        float pm1[LNB_R], pm2[LNB_R];
#pragma unroll
        for (int k = 0; k < LNB_R; ++k) {
            rstd[k] = rsqrtf(ORV_BLOCK_GET(1, 0, k) * invD + p.eps);
            pm1[k] = pm2[k] = 0.f;
            if (k < nr) {
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if (live[i])
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xh = (xv[k][i][e] - mean[k]) * rstd[k];
                            const float d = dv[k][i][e];
                            a1[i][e] += d;
                            a2[i][e] += d * xh;
                            const float dxh = d * osc[i][e] * gam[i][e];
                            xv[k][i][e] = xh;
                            dv[k][i][e] = dxh;
                            pm1[k] += dxh;
                            pm2[k] += dxh * xh;
                        }
            }
        }
        ORV_BLOCK_SUM(2, 0, pm1)
        ORV_BLOCK_SUM(2, 1, pm2)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LNB_R; ++k) {
            if (k >= nr) continue;
            const float m1 = ORV_BLOCK_GET(2, 0, k) * invD, m2 = ORV_BLOCK_GET(2, 1, k) * invD;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (!live[i]) continue;
                const int c = tid + 256 * i;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd[k] * (dv[k][i][e] - m1 - xv[k][i][e] * m2);
                if (p.dres) {
                    const uint4 ur = *(const uint4*)(p.dres + xrow[k] * D + c * 8);
                    const uint32_t wr[4] = {ur.x, ur.y, ur.z, ur.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[2 * e] += bf2f(wr[e] & 0xffff); o[2 * e + 1] += bf2f(wr[e] >> 16); }
                }
                *(uint4*)(p.dx + xrow[k] * D + c * 8) =
                    make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
            }
        }
#undef ORV_BLOCK_SUM
#undef ORV_BLOCK_GET
    }
    // ---- column sums out: per-workgroup partials [blk][4][D] = dgamma | dbeta | dscale | dshift (plain stores; 3 M fp32
    //      atomics into the modulation tables cost 90 us per call), summed by the two reduce kernels below ----
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (!live[i]) continue;
        const int c = tid + 256 * i;
        float* pg = p.part + (long)blockIdx.x * 4 * D + c * 8;
        float o0[8], o1[8], o2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float bet = p.beta ? bf2f(p.beta[c * 8 + e]) : 0.f;
            o0[e] = osc[i][e] * a2[i][e];
            o1[e] = osc[i][e] * a1[i][e];
            o2[e] = gam[i][e] * a2[i][e] + bet * a1[i][e];
        }
        if (p.want_gb) {
            *(float4*)(pg) = make_float4(o0[0], o0[1], o0[2], o0[3]); *(float4*)(pg + 4) = make_float4(o0[4], o0[5], o0[6], o0[7]);
            *(float4*)(pg + D) = make_float4(o1[0], o1[1], o1[2], o1[3]); *(float4*)(pg + D + 4) = make_float4(o1[4], o1[5], o1[6], o1[7]);
        }
        if (p.scale) {
            *(float4*)(pg + 2 * D) = make_float4(o2[0], o2[1], o2[2], o2[3]); *(float4*)(pg + 2 * D + 4) = make_float4(o2[4], o2[5], o2[6], o2[7]);
            *(float4*)(pg + 3 * D) = make_float4(a1[i][0], a1[i][1], a1[i][2], a1[i][3]);
            *(float4*)(pg + 3 * D + 4) = make_float4(a1[i][4], a1[i][5], a1[i][6], a1[i][7]);
        }
    }
}

// out0[j] += sum_blocks part[blk][0][j] ; out1[j] += sum_blocks part[blk][1][j]   (grid: D/256 x slabs of 64 blocks)
constexpr int LNP_SLICE = 16;   // workgroup partials one thread adds before its atomic (64: 104 workgroups, 20 us; 16: 408)
__global__ __launch_bounds__(256) void ln_partials_reduce_kernel(const float* __restrict__ part, int nblk, int D,
                                                                 float* __restrict__ out0, float* __restrict__ out1) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= D) return;
    const int b0 = blockIdx.y * LNP_SLICE, b1 = min(nblk, b0 + LNP_SLICE);
    float s0 = 0.f, s1 = 0.f;
    for (int b = b0; b < b1; ++b) { s0 += part[(long)b * 4 * D + j]; s1 += part[(long)b * 4 * D + D + j]; }
    if (out0) atomicAdd(out0 + j, s0);
    if (out1) atomicAdd(out1 + j, s1);
}

// ---- small-row linear backward (rows <= 64; the conditioning MLPs and AdaLN linears):
//      dW[n, k] (+)= sum_r dy[r, n] * x[r, k]      (fp32 dy, bf16 x -> bf16 dW accumulate)
__global__ void small_wgrad_kernel(const float* __restrict__ dy, long ldy, const bf16_t* __restrict__ x, long ldx_,
                                   bf16_t* __restrict__ dW, float* __restrict__ db, int R, int N, int K, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += dy[(long)r * ldy + n] * bf2f(x[(long)r * ldx_ + k]);
    if (accumulate) s += bf2f(dW[i]);
    dW[i] = f2bf(s);
    if (db && k == 0) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += dy[(long)r * ldy + n];
        if (accumulate) t += db[n];
        db[n] = t;
    }
}
//      dx[r, k] += sum_n dy[r, n] * W[n, k]   (fp32 atomics; block = 256 columns n-slab x all rows)
__global__ __launch_bounds__(256) void small_dgrad_kernel(const float* __restrict__ dy, long ldy, const bf16_t* __restrict__ W,
                                                          float* __restrict__ dx, long lddx, int R, int N, int K, int nslab) {
    // block: 256 columns k x one n-slab; dy slab staged in LDS (rows x nslab), W streamed once, 32 rows per pass in registers
    extern __shared__ float dys[];   // [min(R,32)][nslab]
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n0 = blockIdx.y * nslab, n1 = min(N, n0 + nslab), nn = n1 - n0;
    for (int r0 = 0; r0 < R; r0 += 32) {
        const int rr = min(32, R - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < rr * nn; i += 256) dys[(i / nn) * nslab + i % nn] = dy[(long)(r0 + i / nn) * ldy + n0 + i % nn];
        __syncthreads();
        if (k < K) {
            float acc[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j] = 0.f;
            for (int n = 0; n < nn; ++n) {
                const float w = bf2f(W[(long)(n0 + n) * K + k]);
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (j < rr) acc[j] = fmaf(dys[j * nslab + n], w, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (j < rr) atomicAdd(dx + (long)(r0 + j) * lddx + k, acc[j]);
        }
    }
}

// ---- adjoint of orv_modulation_tables: ALL AdaLN linears of a backward in two launches -------------------------------
//      tab t, video rows:  out[t][b][1+f][n] = W_t[n,:] . cond_v[b*T+f,:] + bias      (n < width)
//      tab t, text rows:   out[t][b][0][n]   = W_t[width+n,:] . cond_t[b,:] + bias    (if text)
//      given dtab = d out (fp32, same layout):
//        gW_t[n,k]  = sum_rows dtab * cond   (bf16, OVERWRITTEN: one flat [n_tab, width*(1+text), E] buffer)
//        gb[t][n]   = sum_rows dtab          (fp32, overwritten)
//        d_cond_v[r,k] += sum_{t,n} dtab_v[t][r][n] * W_t[n,k]     d_cond_t likewise from the text halves
struct ModBwdArgs {
    const float* dtab; const bf16_t* cond_v; const bf16_t* cond_t; const bf16_t* const* W;
    bf16_t* gW; float* gb; float* d_cond_v; float* d_cond_t;
    int n_tab, B, T, E, width, text;
};

// weight / bias gradient: thread -> (n, 8 consecutive k); rows <= 32 per half (checked by the host)
__global__ __launch_bounds__(256) void mod_bwd_wgrad_kernel(const ModBwdArgs p) {
    const int tab = blockIdx.y, E = p.E, kc = E >> 3;
    const int ntot = p.width * (1 + p.text);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)ntot * kc) return;
    const int n = (int)(i / kc), k0 = (int)(i % kc) * 8;
    const bool is_text = n >= p.width;
    const int nn = is_text ? n - p.width : n;
    const int rows = is_text ? p.B : p.B * p.T, G = 1 + p.T;
    const bf16_t* cond = is_text ? p.cond_t : p.cond_v;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bs = 0.f;
    for (int r = 0; r < rows; ++r) {
        const int b = is_text ? r : r / p.T, g = is_text ? 0 : 1 + r % p.T;
        const float d = p.dtab[(((long)tab * p.B + b) * G + g) * p.width + nn];
        const uint4 u = *(const uint4*)(cond + (long)r * E + k0);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += d * bf2f(w[e] & 0xffff); acc[2 * e + 1] += d * bf2f(w[e] >> 16); }
        bs += d;
    }
    *(uint4*)(p.gW + ((long)tab * ntot + n) * E + k0) =
        make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
    if (k0 == 0) p.gb[(long)tab * ntot + n] = bs;
}

// input gradient: block = (64-wide k slice, tab, half); 256 threads = 64 k x 4 n-phases; the dtab slab of 128 n x rows is
// staged in LDS as [n][32 rows]; every thread streams its W column once (128-byte rows across the 64 k lanes).
// d cond[row, k] += sum_n dtab[row, n] W[n, k] over one quarter of the table's rows n.  thread -> FOUR consecutive k (one 8-byte load per weight
// row; with one bf16 per load the whole grid had ~1 MB in flight and streamed the 354 MB of weights at 0.5 TB/s: 600 us per launch), 64 threads
// cover 256 k, the four waves split the n rows (mod 4) and are summed through LDS; grid = (E / 256 x 4 n-quarters, tables, halves).
// RM = rows per pass (accumulators per thread and k): 32, or 8 when the half has at most 8 rows (the text half: B rows).
template <int RM>
__device__ __forceinline__ void mod_bwd_dgrad_body(const ModBwdArgs& p, float (*dys)[32], float (*red)[32][64]) {
    const int tab = blockIdx.y, half = blockIdx.z;          // half 0 = video rows, 1 = text rows
    const int E = p.E, G = 1 + p.T;
    const int rows = half ? p.B : p.B * p.T;
    const int kblocks = (E + 255) / 256;
    const int kb = blockIdx.x % kblocks, nq = blockIdx.x / kblocks;
    const int kk = threadIdx.x & 63, ph = threadIdx.x >> 6, k = kb * 256 + kk * 4;
    const int nchunk = ((p.width + 3) / 4 + 127) / 128 * 128;                 // rows n of this quarter: [nlo, nhi), a multiple of the 128-row tile
    const int nlo = nq * nchunk, nhi = min(p.width, nlo + nchunk);
    const bf16_t* W = p.W[tab] + (long)(half ? p.width : 0) * E;
    for (int r0 = 0; r0 < rows; r0 += RM) {                 // RM rows per pass (one pass up to B*T = 32; W is re-streamed beyond)
        const int rr = min(RM, rows - r0);
        float acc[RM][4];
#pragma unroll
        for (int j = 0; j < RM; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; acc[j][3] = 0.f; }
        for (int n0 = nlo; n0 < nhi; n0 += 128) {
            __syncthreads();
            for (int i = threadIdx.x; i < 128 * RM; i += 256) {
                const int r = i >> 7, nl = i & 127;          // consecutive threads -> consecutive n (coalesced dtab rows)
                float d = 0.f;
                if (r < rr && n0 + nl < nhi) {
                    const int row = r0 + r;
                    const int b = half ? row : row / p.T, g = half ? 0 : 1 + row % p.T;
                    d = p.dtab[(((long)tab * p.B + b) * G + g) * p.width + n0 + nl];
                }
                dys[nl][r] = d;
            }
            __syncthreads();
            if (k < E) {
                const int nend = min(128, nhi - n0);
                // eight weight loads in flight per thread (rows nl, nl + 4, ... ; rows past the tile end re-read its last row with weight 0)
                for (int nl0 = ph; nl0 < nend; nl0 += 32) {
                    float w8[8][4];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int nl = nl0 + 4 * u;
                        const uint2 wv = *(const uint2*)(W + (long)(n0 + min(nl, nend - 1)) * E + k);
                        const bool on = nl < nend;
                        w8[u][0] = on ? bf2f(wv.x & 0xffff) : 0.f; w8[u][1] = on ? bf2f(wv.x >> 16) : 0.f;
                        w8[u][2] = on ? bf2f(wv.y & 0xffff) : 0.f; w8[u][3] = on ? bf2f(wv.y >> 16) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int nl = min(nl0 + 4 * u, 127);
#pragma unroll
                        for (int j = 0; j < RM; j += 4) {
                            const float4 d4 = *(const float4*)&dys[nl][j];
                            const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int c = 0; c < 4; ++c) acc[j + jj][c] = fmaf(dd[jj], w8[u][c], acc[j + jj][c]);
                        }
                    }
                }
            }
        }
        // sum of the four waves (they took the n rows mod 4), one 4-column group at a time through LDS
        float* dx = half ? p.d_cond_t : p.d_cond_v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            __syncthreads();
            if (ph > 0)
#pragma unroll
                for (int j = 0; j < RM; ++j) red[ph - 1][j][kk] = acc[j][c];
            __syncthreads();
            if (ph == 0 && k + c < E)
#pragma unroll
                for (int j = 0; j < RM; ++j)
                    if (j < rr) atomicAdd(dx + (long)(r0 + j) * E + k + c, acc[j][c] + red[0][j][kk] + red[1][j][kk] + red[2][j][kk]);
        }
    }
}
__global__ __launch_bounds__(256) void mod_bwd_dgrad_kernel(const ModBwdArgs p) {
    __shared__ __attribute__((aligned(16))) float dys[128][32];
    __shared__ float red[3][32][64];
    const int rows = blockIdx.z ? p.B : p.B * p.T;          // uniform over the workgroup
    if (rows <= 8) mod_bwd_dgrad_body<8>(p, dys, red);
    else if (rows <= 16) mod_bwd_dgrad_body<16>(p, dys, red);
    else if (rows <= 24) mod_bwd_dgrad_body<24>(p, dys, red);           // B T = 20 at the training shape
    else mod_bwd_dgrad_body<32>(p, dys, red);
}

// ---- fused AdamW on bf16 parameters / bf16 gradients with fp32 moments (torch.optim.AdamW semantics, decoupled decay):
//      g *= clip ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p = p (1 - lr wd) - lr (m / bc1) / (sqrt(v / bc2) + eps)
__global__ void adamw_kernel(bf16_t* __restrict__ p, const bf16_t* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                             const float* __restrict__ clip) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = clip ? *clip : 1.f;
    const float gr = bf2f(g[i]) * c;
    const float mm = b1 * m[i] + (1.f - b1) * gr;
    const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mm; v[i] = vv;
    float pv = bf2f(p[i]);
    pv = pv * (1.f - lr * wd) - lr * (mm / bc1) / (sqrtf(vv / bc2) + eps);
    p[i] = f2bf(pv);
}

// ---- the same update over ONE flat buffer holding every parameter (segments padded to 2048 elements): a workgroup owns
//      2048 consecutive elements = one segment; segments whose parameter had no gradient this step are skipped
//      (torch.optim semantics: no decay, no moment update).  16-byte accesses throughout (22 bytes of traffic per element).
__global__ __launch_bounds__(256) void adamw_flat_kernel(bf16_t* __restrict__ p, const bf16_t* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const long* __restrict__ seg_start, const uint8_t* __restrict__ active,
                                                         int nseg, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                         float bc2, const float* __restrict__ clip,
                                                         const int* __restrict__ seg_step) {
    const long e0 = (long)blockIdx.x * 2048;
    int lo = 0, hi = nseg - 1;                 // last segment with seg_start <= e0 (wave-uniform binary search)
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg_start[mid] <= e0) lo = mid; else hi = mid - 1; }
    if (!active[lo]) return;
    if (seg_step) {                            // torch.optim.AdamW keeps one step count PER PARAMETER (bias correction)
        const float st = (float)seg_step[lo];
        bc1 = 1.f - powf(b1, st);
        bc2 = 1.f - powf(b2, st);
    }
    const long i = e0 + threadIdx.x * 8;
    const float c = clip ? *clip : 1.f;
    const uint4 up = *(const uint4*)(p + i), ug = *(const uint4*)(g + i);
    float4 m0 = *(const float4*)(m + i), m1 = *(const float4*)(m + i + 4), v0 = *(const float4*)(v + i), v1 = *(const float4*)(v + i + 4);
    float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    const uint32_t wp[4] = {up.x, up.y, up.z, up.w}, wg[4] = {ug.x, ug.y, ug.z, ug.w};
    float pv[8];
    const float decay = 1.f - lr * wd, ibc1 = 1.f / bc1, ibc2 = 1.f / bc2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float gr = bf2f((e & 1) ? wg[e >> 1] >> 16 : wg[e >> 1] & 0xffff) * c;
        const float pe = bf2f((e & 1) ? wp[e >> 1] >> 16 : wp[e >> 1] & 0xffff);
        mm[e] = b1 * mm[e] + (1.f - b1) * gr;
        vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
        pv[e] = pe * decay - lr * (mm[e] * ibc1) / (sqrtf(vv[e] * ibc2) + eps);
    }
    *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]); *(float4*)(m + i + 4) = make_float4(mm[4], mm[5], mm[6], mm[7]);
    *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]); *(float4*)(v + i + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
    *(uint4*)(p + i) = make_uint4(pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3]), pack2bf(pv[4], pv[5]), pack2bf(pv[6], pv[7]));
}

// ---- out[0] += sum g^2 (global gradient norm)
__global__ __launch_bounds__(256) void sumsq_kernel(const bf16_t* __restrict__ g, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 256 * 8) {
        if (i + 8 <= n) {
            const uint4 u = *(const uint4*)(g + i);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float a = bf2f(w[e] & 0xffff), b = bf2f(w[e] >> 16); s += a * a + b * b; }
        } else {
            for (long j = i; j < n; ++j) { const float a = bf2f(g[j]); s += a * a; }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

extern "C" int orv_transpose_bf16(const void* src, int ld_src, void* dst, int ld_dst, int R, int C, void* stream) {
    ORV_REQUIRE(src && dst && R > 0 && C > 0, "orv_transpose_bf16: bad arguments");
    ORV_REQUIRE(ld_src % 8 == 0 && ld_dst % 8 == 0 && ld_dst >= R && C % 8 == 0, "orv_transpose_bf16: misaligned");
    dim3 grid((ld_dst + 63) / 64, (C + 63) / 64);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (long)ld_src,
                       (bf16_t*)dst, (long)ld_dst, R, C);
    return orv_check_launch("orv_transpose_bf16");
}

extern "C" int orv_transpose_colsum_bf16(const void* src, int ld_src, void* dst, int ld_dst, int R, int C, float* colsum,
                                         void* stream) {
    ORV_REQUIRE(src && dst && colsum && R > 0 && C > 0, "orv_transpose_colsum_bf16: bad arguments");
    ORV_REQUIRE(ld_src % 8 == 0 && ld_dst % 8 == 0 && ld_dst >= R && C % 8 == 0, "orv_transpose_colsum_bf16: misaligned");
    // 16-byte loads from src / stores to dst (a column slice of a wider tensor is a legal view and would fault here); colsum is
    // ACCUMULATED with fp32 atomics: the caller zeroes it (ops.transpose_colsum does)
    ORV_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "orv_transpose_colsum_bf16: src and dst must be 16-byte aligned");
    constexpr int RT = 4;
    const int row_tiles = (ld_dst + 63) / 64;
    dim3 grid((row_tiles + RT - 1) / RT, (C + 63) / 64);
    hipLaunchKernelGGL(transpose_colsum_kernel<RT>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (long)ld_src,
                       (bf16_t*)dst, (long)ld_dst, R, C, colsum);
    return orv_check_launch("orv_transpose_colsum_bf16");
}

extern "C" int orv_colsum(const void* src, int ld, float* out, int R, int C, void* stream) {
    ORV_REQUIRE(src && out && R > 0 && C > 0, "orv_colsum: bad arguments");
    ORV_REQUIRE(C % 8 == 0 && ld % 8 == 0 && ((uintptr_t)src & 15) == 0, "orv_colsum: C / ld must be multiples of 8, src 16-byte aligned");
    const int cblocks = (C + 511) / 512;
    // about one workgroup per CU: fewer, longer slabs mean fewer fp32 atomics on the same C addresses (measured at R = 12904:
    // C = 1920 46 -> 20 us with 204 instead of 808 workgroups, C = 5760 / 7680 32 / 35 -> 27 / 32 us)
    int rpb = 512;
    while (rpb > 32 && (long)cblocks * ((R + rpb - 1) / rpb) < 200) rpb >>= 1;
    dim3 grid(cblocks, (R + rpb - 1) / rpb);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (long)ld, out, R, C, rpb);
    return orv_check_launch("orv_colsum");
}

static void group_blocks(orv_groups_t grp, int rb, int& bt, int& bg, int& bpb) {
    const int vid = grp.seq - grp.n_text;
    const int gsz = grp.per_group > 0 ? grp.per_group : vid;
    const int ngroups = gsz > 0 ? (vid + gsz - 1) / gsz : 0;
    bt = (grp.n_text + rb - 1) / rb;
    bg = gsz > 0 ? (gsz + rb - 1) / rb : 0;
    bpb = bt + ngroups * bg;
}

extern "C" long orv_gated_residual_bwd_scratch(orv_groups_t grp, int batch, int D) {
    int bt, bg, bpb;
    group_blocks(grp, GRB, bt, bg, bpb);
    return (long)batch * bpb * D;
}

extern "C" int orv_gated_residual_bwd(const void* dout, const void* y, const float* gate, float* dgate, void* dy, float* scratch,
                                      long mod_b, long mod_g, orv_groups_t grp, int batch, int D, void* stream) {
    ORV_REQUIRE(dout && y && gate && dgate && dy && scratch, "orv_gated_residual_bwd: null operand");
    ORV_REQUIRE(D % 8 == 0 && D <= 4096, "orv_gated_residual_bwd: D=%d unsupported", D);
    ORV_REQUIRE(grp.per_group <= 0 || (grp.seq - grp.n_text) % grp.per_group == 0,
                "orv_gated_residual_bwd: video rows must be a whole number of groups");
    int bt, bg, bpb;
    group_blocks(grp, GRB, bt, bg, bpb);
    hipStream_t st = (hipStream_t)stream;
#define ORV_CASE(C)                                                                                                    \
    hipLaunchKernelGGL(gated_bwd_kernel<C>, dim3(batch * bpb), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)y, gate, \
                       dgate, (bf16_t*)dy, mod_b, mod_g, grp.seq, grp.n_text, grp.per_group, D, bt, bg, bpb, scratch)
    if (D <= 2048) ORV_CASE(1); else ORV_CASE(2);
#undef ORV_CASE
    const int ngroups = bg > 0 ? (bpb - bt) / bg : 0;
    hipLaunchKernelGGL(groups_reduce_kernel, dim3((D + 63) / 64, batch * (1 + ngroups)), dim3(256), 0, st, scratch, (long)D, 0, 0, D,
                       dgate, (float*)nullptr, mod_b, mod_g, bt, bg, bpb, ngroups);
    return orv_check_launch("orv_gated_residual_bwd");
}

static void ln_bwd_blocks(orv_groups_t grp, int& bt, int& bg, int& bpb) { group_blocks(grp, LNB_RB, bt, bg, bpb); }

extern "C" long orv_layernorm_modulate_bwd_scratch(orv_groups_t grp, int batch, int D) {
    int bt, bg, bpb;
    ln_bwd_blocks(grp, bt, bg, bpb);
    return (long)batch * bpb * 4 * D;
}

extern "C" int orv_layernorm_modulate_bwd(const void* dy, const void* x, orv_rowmap_t xmap, const void* dres, void* dx,
                                          const void* gamma, const void* beta, const float* scale, float* dscale,
                                          float* dshift, float* dgamma, float* dbeta, float* scratch, long mod_b, long mod_g,
                                          orv_groups_t grp, int batch, int D, float eps, void* stream) {
    ORV_REQUIRE(dy && x && dx, "orv_layernorm_modulate_bwd: null operand");
    ORV_REQUIRE(D % 8 == 0 && D <= 4096, "orv_layernorm_modulate_bwd: D=%d unsupported", D);
    ORV_REQUIRE(!scale || (dscale && dshift), "orv_layernorm_modulate_bwd: dscale/dshift required with scale");
    ORV_REQUIRE(!(dgamma || dbeta || scale) || scratch, "orv_layernorm_modulate_bwd: scratch required for the column sums");
    ORV_REQUIRE(grp.per_group <= 0 || (grp.seq - grp.n_text) % grp.per_group == 0,
                "orv_layernorm_modulate_bwd: video rows must be a whole number of groups");
    LnBwdArgs a;
    a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x; a.dres = (const bf16_t*)dres; a.dx = (bf16_t*)dx;
    a.gamma = (const bf16_t*)gamma; a.beta = (const bf16_t*)beta; a.scale = scale; a.dscale = dscale; a.dshift = dshift;
    a.part = scratch; a.want_gb = (dgamma || dbeta) ? 1 : 0;
    a.mod_b = mod_b; a.mod_g = mod_g; a.seq = grp.seq; a.n_text = grp.n_text; a.per_group = grp.per_group; a.D = D;
    a.eps = eps; a.xmap = xmap;
    ln_bwd_blocks(grp, a.bt, a.bg, a.bpb);
    const int nblk = batch * a.bpb;
    hipStream_t st = (hipStream_t)stream;
    if (D <= 2048) hipLaunchKernelGGL(ln_mod_bwd_kernel<1>, dim3(nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(ln_mod_bwd_kernel<2>, dim3(nblk), dim3(256), 0, st, a);
    if (a.want_gb)
        hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3((D + 255) / 256, (nblk + LNP_SLICE - 1) / LNP_SLICE), dim3(256), 0, st, scratch, nblk, D,
                           dgamma, dbeta);
    if (scale) {
        const int ngroups = a.bg > 0 ? (a.bpb - a.bt) / a.bg : 0;
        hipLaunchKernelGGL(groups_reduce_kernel, dim3((D + 63) / 64, batch * (1 + ngroups)), dim3(256), 0, st, scratch, 4L * D, 2,
                           3, D, dscale, dshift, mod_b, mod_g, a.bt, a.bg, a.bpb, ngroups);
    }
    return orv_check_launch("orv_layernorm_modulate_bwd");
}

extern "C" int orv_small_linear_bwd(const float* dy, int ldy, const void* x, int ldx, const void* W, void* dW, float* db,
                                    float* dx, int lddx, int R, int N, int K, int accumulate, void* stream) {
    ORV_REQUIRE(dy && R > 0 && R <= 4096 && N > 0 && K > 0, "orv_small_linear_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dW) {
        ORV_REQUIRE(x, "orv_small_linear_bwd: x required for dW");
        const long total = (long)N * K;
        hipLaunchKernelGGL(small_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, (long)ldy,
                           (const bf16_t*)x, (long)ldx, (bf16_t*)dW, db, R, N, K, accumulate);
    }
    if (dx) {
        ORV_REQUIRE(W, "orv_small_linear_bwd: W required for dx");
        const int nslab = 64;
        dim3 grid((K + 255) / 256, (N + nslab - 1) / nslab);
        hipLaunchKernelGGL(small_dgrad_kernel, grid, dim3(256), 32 * nslab * sizeof(float), st, dy, (long)ldy, (const bf16_t*)W, dx,
                           (long)lddx, R, N, K, nslab);
    }
    return orv_check_launch("orv_small_linear_bwd");
}

extern "C" int orv_modulation_tables_bwd(const float* dtab, const void* cond_v, const void* cond_t, const void* const* W,
                                         void* gW, float* gb, float* d_cond_v, float* d_cond_t, int n_tab, int B, int T, int E,
                                         int width, int text, void* stream) {
    ORV_REQUIRE(dtab && cond_v && W && gW && gb && d_cond_v, "orv_modulation_tables_bwd: null operand");
    ORV_REQUIRE(!text || (cond_t && d_cond_t), "orv_modulation_tables_bwd: text rows need cond_t / d_cond_t");
    ORV_REQUIRE(n_tab > 0 && B > 0 && T > 0 && B * T <= 4096 && E % 64 == 0 && width % 8 == 0,
                "orv_modulation_tables_bwd: unsupported shape (B*T=%d, E=%d %% 64, width=%d %% 8)", B * T, E, width);
    ModBwdArgs a{dtab, (const bf16_t*)cond_v, (const bf16_t*)cond_t, (const bf16_t* const*)W, (bf16_t*)gW, gb, d_cond_v, d_cond_t,
                 n_tab, B, T, E, width, text ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const long per_tab = (long)width * (1 + a.text) * (E / 8);
    hipLaunchKernelGGL(mod_bwd_wgrad_kernel, dim3((unsigned)((per_tab + 255) / 256), n_tab), dim3(256), 0, st, a);
    hipLaunchKernelGGL(mod_bwd_dgrad_kernel, dim3(((E + 255) / 256) * 4, n_tab, 1 + a.text), dim3(256), 0, st, a);
    return orv_check_launch("orv_modulation_tables_bwd");
}

extern "C" int orv_adamw(void* p, const void* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, const float* clip_coef, void* stream) {
    ORV_REQUIRE(p && g && m && v && n > 0 && step > 0, "orv_adamw: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p,
                       (const bf16_t*)g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, clip_coef);
    return orv_check_launch("orv_adamw");
}

extern "C" int orv_adamw_flat_steps(void* p, const void* g, float* m, float* v, long n, const long* seg_start,
                                    const unsigned char* seg_active, const int* seg_step, int nseg, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int step, const float* clip_coef, void* stream) {
    ORV_REQUIRE(p && g && m && v && seg_start && seg_active && nseg > 0 && step > 0, "orv_adamw_flat: bad arguments");
    ORV_REQUIRE(n > 0 && n % 2048 == 0, "orv_adamw_flat: n=%ld must be a multiple of 2048 (pad every segment)", n);
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)(n / 2048)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p,
                       (const bf16_t*)g, m, v, seg_start, seg_active, nseg, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                       clip_coef, seg_step);
    return orv_check_launch("orv_adamw_flat");
}

extern "C" int orv_adamw_flat(void* p, const void* g, float* m, float* v, long n, const long* seg_start,
                              const unsigned char* seg_active, int nseg, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, const float* clip_coef, void* stream) {
    return orv_adamw_flat_steps(p, g, m, v, n, seg_start, seg_active, nullptr, nseg, lr, beta1, beta2, eps, weight_decay, step,
                                clip_coef, stream);
}

// ---- dst_s[j] = bf16(src[off_s + j]) for a list of segments s: the small fp32-accumulated gradients (biases, LayerNorm affine, qk-norm)
//      leave the backward's arena straight into their places of the optimizer's flat bf16 gradient buffer - one launch.
__global__ __launch_bounds__(256) void scatter_cast_kernel(const float* __restrict__ src, const long* __restrict__ off,
                                                           const long* __restrict__ dst, const int* __restrict__ len) {
    const int s_ = blockIdx.y;
    const int n = len[s_];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    ((bf16_t*)dst[s_])[j] = f2bf(src[off[s_] + j]);
}

extern "C" int orv_scatter_f32_to_bf16(const float* src, const long* src_off, const long* dst_ptr, const int* len, int nseg,
                                       int max_len, void* stream) {
    ORV_REQUIRE(src && src_off && dst_ptr && len && nseg > 0 && max_len > 0, "orv_scatter_f32_to_bf16: bad arguments");
    ORV_REQUIRE(nseg <= 65535, "orv_scatter_f32_to_bf16: %d segments (at most 65535 per call)", nseg);
    hipLaunchKernelGGL(scatter_cast_kernel, dim3((max_len + 255) / 256, nseg), dim3(256), 0, (hipStream_t)stream, src, src_off,
                       dst_ptr, len);
    return orv_check_launch("orv_scatter_f32_to_bf16");
}

extern "C" int orv_sumsq(const void* g, long n, float* out, void* stream) {
    ORV_REQUIRE(g && out && n > 0, "orv_sumsq: bad arguments");
    const int blocks = (int)min((long)2048, (n + 2047) / 2048);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, n, out);
    return orv_check_launch("orv_sumsq");
}
