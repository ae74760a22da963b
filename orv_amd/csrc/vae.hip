// CogVideoX VAE (AutoencoderKLCogVideoX) building blocks for gfx950 - SURVEY.md §8(f) rank 1: the decode the reference calls at
// orv/models/cogvideox_control.py:1476-1479 and the reference-frame encode at :1161-1166 (arithmetic: diffusers, absent).
//
// Data layout: activations are CHANNELS-LAST bf16 [B, T, H, W, C]: a voxel's channels are one contiguous line, so
//   * a causal 3-D convolution is a GEMM  out[voxel, co] = sum_{tap, ci} patch[voxel, tap*C + ci] * Wp[co, tap*C + ci]
//     whose A operand (orv_vae_im2col) gathers, per tap, whole C-channel lines: 16-byte accesses, coalesced on both sides.
//     The gather also folds what sits in front of the convolution in the reference: replicate-first-frame temporal padding
//     (CogVideoXCausalConv3d), zero spatial padding, nearest-neighbour x2 upsampling in space / time (CogVideoXUpsample3D:
//     the up-sampled tensor is never materialised) and the stride-2 / pad (0,1,0,1) form of CogVideoXDownsample3D.
//     The GEMM itself is orv_gemm_bf16 (bias / residual-add epilogues).
//   * GroupNorm statistics (32 groups over C/32 channels x all voxels of a frame batch) are one streaming pass with fp32
//     partial sums per workgroup, combined in a fixed order (orv_vae_groupnorm_stats: no atomics, bit-reproducible);
//     normalisation, the SpatialNorm
//     modulation by the latent (norm(f) * conv_y(zq) + conv_b(zq), zq looked up by nearest-neighbour index, never resized
//     in memory) and SiLU are ONE elementwise pass (orv_vae_norm_apply).
// All three kernels are HBM-bound: algorithmic bytes are stated at each entry point.
#include "common.hpp"

namespace {

struct Im2colArgs {
    const bf16_t* src; bf16_t* dst;
    int B, Ts, Hs, Ws, C;        // source tensor [B, Ts, Hs, Ws, C]
    int T, H, W;                 // output voxel grid of the convolution
    int kt, kh, kw;              // kernel taps
    int stride;                  // spatial stride (1, or 2 for the down-sampling conv)
    int pad_lo;                  // zero padding in front (top / left); the far side is implied by the output size
    int ups_s;                   // 1: conv input = source nearest-upsampled x2 in H, W
    int ups_t;                   // 0 none, 1 all frames doubled, 2 first frame kept + the others doubled (odd clips)
    int t_shift;                 // frames of context the caller put in FRONT of src (the previous chunk's conv_cache): output frame
                                 // t then reads source frames t + t_shift - (kt-1) + dt (0: replicate the first frame instead)
    int Kpad;                    // row pitch of dst in elements (>= taps * C, multiple of 64; tail zero-filled)
    long m0, mc;                 // rows [m0, m0 + mc) of the [B*T*H*W] voxel list are produced
};

__global__ __launch_bounds__(256) void vae_im2col_kernel(const Im2colArgs p) {
    const int cpr = p.Kpad >> 3;                                   // 16-byte chunks per dst row
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.mc * cpr) return;
    const long r = i / cpr;
    const int ch = (int)(i % cpr);
    const int k = ch * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    const int taps = p.kt * p.kh * p.kw;
    if (k < taps * p.C) {
        const int tap = k / p.C, c = k % p.C;
        const int dx = tap % p.kw, dy = (tap / p.kw) % p.kh, dt = tap / (p.kw * p.kh);
        long m = p.m0 + r;
        const int x = (int)(m % p.W); m /= p.W;
        const int y = (int)(m % p.H); m /= p.H;
        const int t = (int)(m % p.T);
        const int b = (int)(m / p.T);
        // conv-input coordinates (after padding / upsampling)
        int ti = t + p.t_shift - (p.kt - 1) + dt;                       // causal: (kt-1) frames in front
        if (ti < 0) ti = 0;                                             // ... filled with copies of the first frame
        const int yi = y * p.stride - p.pad_lo + dy, xi = x * p.stride - p.pad_lo + dx;
        const int Hi = p.ups_s ? p.Hs * 2 : p.Hs, Wi = p.ups_s ? p.Ws * 2 : p.Ws;
        if (yi >= 0 && yi < Hi && xi >= 0 && xi < Wi) {
            const int ys = p.ups_s ? yi >> 1 : yi, xs = p.ups_s ? xi >> 1 : xi;
            int ts = ti;
            if (p.ups_t == 1) ts = ti >> 1;
            else if (p.ups_t == 2) ts = ti == 0 ? 0 : 1 + ((ti - 1) >> 1);
            v = *(const uint4*)(p.src + ((((long)b * p.Ts + ts) * p.Hs + ys) * p.Ws + xs) * p.C + c);
        }
    }
    *(uint4*)(p.dst + r * p.Kpad + k) = v;
}

// per (batch, group): sum and sum of squares over N voxels x cpg channels, DETERMINISTIC (no atomics: run-to-run identical
// frames).  Block = 256 threads over a slab of voxels; thread -> 8-channel chunk (tid % (C/8)), voxel lane tid / (C/8); per-thread
// per-channel partials go through LDS, one thread per group adds them in a fixed order, block partials land in `scratch`
// [B, nblk, G, 2] and a second tiny launch adds those in block order.
__global__ __launch_bounds__(256) void vae_gn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ scratch, long N,
                                                           int C, int G, long vox_per_block) {
    __shared__ float part[256][17];                // [thread][8 sums | 8 sums of squares] (+1: bank spread)
    const int b = blockIdx.y;
    const int cc = C >> 3;                         // chunks per voxel
    const int chunk = threadIdx.x % cc, vl = threadIdx.x / cc, vstep = blockDim.x / cc;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    const long v0 = (long)blockIdx.x * vox_per_block, v1 = min(N, v0 + vox_per_block);
    auto add = [&](const uint4& u) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bf2f(w[e] & 0xffff), c2 = bf2f(w[e] >> 16);
            s[2 * e] += a; q[2 * e] += a * a;
            s[2 * e + 1] += c2; q[2 * e + 1] += c2 * c2;
        }
    };
    const bf16_t* xb = x + (long)b * N * C + chunk * 8;
    long v = v0 + vl;
    for (; v + 3 * vstep < v1; v += 4 * vstep) {       // four independent 16-byte loads in flight; summed in voxel order
        const uint4 u0 = *(const uint4*)(xb + v * C), u1 = *(const uint4*)(xb + (v + vstep) * C);
        const uint4 u2 = *(const uint4*)(xb + (v + 2 * vstep) * C), u3 = *(const uint4*)(xb + (v + 3 * vstep) * C);
        add(u0); add(u1); add(u2); add(u3);
    }
    for (; v < v1; v += vstep) add(*(const uint4*)(xb + v * C));
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[threadIdx.x][e] = s[e]; part[threadIdx.x][8 + e] = q[e]; }
    __syncthreads();
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float ts = 0.f, tq = 0.f;
        for (int ch = g * cpg; ch < (g + 1) * cpg; ++ch)
            for (int j = 0; j < vstep; ++j) { ts += part[(ch >> 3) + cc * j][ch & 7]; tq += part[(ch >> 3) + cc * j][8 + (ch & 7)]; }
        float* o = scratch + (((long)b * gridDim.x + blockIdx.x) * G + g) * 2;
        o[0] = ts; o[1] = tq;
    }
}
// sums[b, i] = sum over blocks of scratch[b, blk, i]: one workgroup per (b, i), 256 threads stride over the blocks, fixed-order
// LDS tree - deterministic and parallel (a full-resolution clip has ~10^4 block partials per batch element)
__global__ __launch_bounds__(256) void vae_gn_reduce_kernel(const float* __restrict__ scratch, float* __restrict__ sums, int nblk,
                                                            int G2) {
    __shared__ float red[256];
    const int b = blockIdx.y, i = blockIdx.x;
    float t = 0.f;
    for (int k = threadIdx.x; k < nblk; k += 256) t += scratch[((long)b * nblk + k) * G2 + i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[(long)b * G2 + i] = red[0];
}

struct NormArgs {
    const bf16_t* x; bf16_t* out;
    const float* sums;                 // [B, G, 2]
    const bf16_t *gamma, *beta;        // [C]
    const bf16_t *zy, *zb;             // conv_y(zq), conv_b(zq) at latent resolution [B, Tz, hz, wz, C] (null: plain GroupNorm)
    int B, T, H, W, C, G;
    int Tz, hz, wz;
    float eps;
    int silu;
    int out_lead;                      // out is [B, out_lead + T, H, W, C]: frames [0, out_lead) are left for the caller (conv_cache)
};

// grid = (ceil(W * C/8 / 256), H, B * T): the frame / row indices are block-uniform, a thread divides once (x position |
// 8-channel chunk) - the first version decomposed a flat 64-bit index per thread and spent its time in integer division.
__global__ __launch_bounds__(256) void vae_norm_apply_kernel(const NormArgs p) {
    const unsigned cc = (unsigned)p.C >> 3;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)p.W * cc) return;
    const unsigned xw = idx / cc, chunk = idx - xw * cc;
    const int yh = blockIdx.y;
    const int b = (int)blockIdx.z / p.T, t = (int)blockIdx.z - b * p.T;
    const int c0 = (int)chunk * 8, cpg = p.C / p.G;
    const float inv_n = 1.f / ((float)p.T * (float)p.H * (float)p.W * (float)cpg);
    const long bv = (((long)b * p.T + t) * p.H + yh) * p.W + xw;
    const uint4 u = *(const uint4*)(p.x + bv * p.C + c0);
    const uint4 ug = *(const uint4*)(p.gamma + c0), ub = *(const uint4*)(p.beta + c0);
    uint4 uy = make_uint4(0, 0, 0, 0), uz = make_uint4(0, 0, 0, 0);
    if (p.zy) {
        // F.interpolate(nearest): src = floor(dst * in / out); SpatialNorm resizes the first frame of an odd-length clip apart
        int tz;
        if (p.T > 1 && (p.T & 1)) tz = t == 0 ? 0 : 1 + ((t - 1) * (p.Tz - 1)) / (p.T - 1);
        else tz = (t * p.Tz) / p.T;
        const int yz = (yh * p.hz) / p.H;
        const unsigned xz = (xw * (unsigned)p.wz) / (unsigned)p.W;
        const long zi = ((((long)b * p.Tz + tz) * p.hz + yz) * p.wz + xz) * p.C + c0;
        uy = *(const uint4*)(p.zy + zi);
        uz = *(const uint4*)(p.zb + zi);
    }
    const uint32_t wx[4] = {u.x, u.y, u.z, u.w}, wg[4] = {ug.x, ug.y, ug.z, ug.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
    const uint32_t wy[4] = {uy.x, uy.y, uy.z, uy.w}, wz[4] = {uz.x, uz.y, uz.z, uz.w};
    float o[8];
    int g = c0 / cpg, r = c0 - g * cpg;       // group of channel c0 and the position inside it; statistics reloaded when it changes
    float mean = 0.f, rstd = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e == 0 || r == 0) {
            const float2 sq = *(const float2*)(p.sums + ((long)b * p.G + g) * 2);
            mean = sq.x * inv_n;
            rstd = rsqrtf(fmaxf(sq.y * inv_n - mean * mean, 0.f) + p.eps);
        }
        auto pick = [&](const uint32_t (&w)[4]) { return bf2f((e & 1) ? w[e >> 1] >> 16 : w[e >> 1] & 0xffff); };
        float val = (pick(wx) - mean) * rstd * pick(wg) + pick(wb);
        if (p.zy) val = val * pick(wy) + pick(wz);
        if (p.silu) val = silu(val);
        o[e] = val;
        if (++r == cpg) { r = 0; ++g; }
    }
    const long ov = (((long)b * (p.T + p.out_lead) + t + p.out_lead) * p.H + yh) * p.W + xw;
    *(uint4*)(p.out + ov * p.C + c0) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
}

}  // namespace

// Patch matrix of a causal / strided / up-sampling convolution: dst[mc, Kpad] rows for voxels [m0, m0 + mc).
// Algorithmic bytes: mc * Kpad * 2 written + the same read (27 x the activation for a 3x3x3 kernel).
extern "C" int orv_vae_im2col(const void* src, void* dst, int B, int Ts, int Hs, int Ws, int C, int T, int H, int W, int kt,
                              int kh, int kw, int stride, int pad_lo, int ups_s, int ups_t, int t_shift, int Kpad, long m0, long mc,
                              void* stream) {
    ORV_REQUIRE(src && dst && B > 0 && Ts > 0 && Hs > 0 && Ws > 0 && T > 0 && H > 0 && W > 0, "orv_vae_im2col: bad shape");
    ORV_REQUIRE(C % 8 == 0 && Kpad % 64 == 0 && Kpad >= kt * kh * kw * C, "orv_vae_im2col: C=%d must be a multiple of 8 and Kpad=%d a multiple of 64 covering %d taps", C, Kpad, kt * kh * kw);
    ORV_REQUIRE(kt >= 1 && kh >= 1 && kw >= 1 && (stride == 1 || stride == 2) && pad_lo >= 0 && ups_t >= 0 && ups_t <= 2 &&
                    t_shift >= 0 && t_shift <= kt - 1 && (t_shift == 0 || ups_t == 0), "orv_vae_im2col: bad kernel geometry");
    ORV_REQUIRE(m0 >= 0 && mc > 0 && m0 + mc <= (long)B * T * H * W, "orv_vae_im2col: row range outside the voxel list");
    Im2colArgs a{(const bf16_t*)src, (bf16_t*)dst, B, Ts, Hs, Ws, C, T, H, W, kt, kh, kw, stride, pad_lo, ups_s, ups_t, t_shift, Kpad, m0, mc};
    const long total = mc * (Kpad >> 3);
    hipLaunchKernelGGL(vae_im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return orv_check_launch("orv_vae_im2col");
}

static long gn_vox_per_block(int C) { return (long)(256 / (C / 8)) * 64; }   // 64 voxel rows per thread
// floats of scratch orv_vae_groupnorm_stats needs for x[B, N, C] with G groups
extern "C" long orv_vae_groupnorm_scratch(int B, long N, int C, int G) {
    if (B <= 0 || N <= 0 || C <= 0 || C % 8 || G <= 0) return 0;
    const long vpb = gn_vox_per_block(C);
    return (long)B * ((N + vpb - 1) / vpb) * G * 2;
}
// sums[B, G, 2] = (sum, sum of squares) per group over x[B, N, C], deterministic.  Algorithmic bytes: B*N*C*2 read.
extern "C" int orv_vae_groupnorm_stats(const void* x, float* sums, float* scratch, int B, long N, int C, int G, void* stream) {
    ORV_REQUIRE(x && sums && scratch && B > 0 && N > 0 && G > 0 && C % 8 == 0 && C % G == 0 && C / 8 <= 256 && 256 % (C / 8) == 0,
                "orv_vae_groupnorm_stats: bad shape C=%d G=%d", C, G);
    const long vpb = gn_vox_per_block(C);
    const long nblk = (N + vpb - 1) / vpb;
    hipLaunchKernelGGL(vae_gn_stats_kernel, dim3((unsigned)nblk, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scratch, N,
                       C, G, vpb);
    hipLaunchKernelGGL(vae_gn_reduce_kernel, dim3(2 * G, B), dim3(256), 0, (hipStream_t)stream, scratch, sums, (int)nblk, 2 * G);
    return orv_check_launch("orv_vae_groupnorm_stats");
}

// out = act(GroupNorm(x; sums) [* zy[zmap] + zb[zmap]]): GroupNorm affine, SpatialNorm modulation by the latent-resolution
// tensors zy / zb (nearest-neighbour lookup) and SiLU in one pass.  Algorithmic bytes: B*N*C*2 read + written.
// out is [B, out_lead + T, H, W, C]; the first out_lead frames of every clip are not written (the caller puts the causal
// context of the next convolution there: its conv_cache frames or copies of the first frame).
extern "C" int orv_vae_norm_apply(const void* x, void* out, const float* sums, const void* gamma, const void* beta,
                                  const void* zy, const void* zb, int B, int T, int H, int W, int C, int G, int Tz, int hz,
                                  int wz, float eps, int silu_act, int out_lead, void* stream) {
    ORV_REQUIRE(x && out && sums && gamma && beta && B > 0 && T > 0 && H > 0 && W > 0 && C % 8 == 0 && G > 0 && C % G == 0,
                "orv_vae_norm_apply: bad arguments");
    ORV_REQUIRE(out_lead >= 0 && out_lead <= 8, "orv_vae_norm_apply: out_lead=%d outside [0, 8]", out_lead);
    ORV_REQUIRE((zy == nullptr) == (zb == nullptr) && (!zy || (Tz > 0 && hz > 0 && wz > 0)), "orv_vae_norm_apply: zy / zb go together");
    NormArgs a{(const bf16_t*)x, (bf16_t*)out, sums, (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)zy,
               (const bf16_t*)zb, B, T, H, W, C, G, Tz, hz, wz, eps, silu_act, out_lead};
    ORV_REQUIRE(H <= 65535 && (long)B * T <= 65535 && (long)W * (C >> 3) < (1L << 31) && (long)T * Tz < (1L << 31) &&
                    (long)H * hz < (1L << 31) && (long)W * wz < (1L << 31), "orv_vae_norm_apply: clip too large for one launch");
    hipLaunchKernelGGL(vae_norm_apply_kernel, dim3((unsigned)(((long)W * (C >> 3) + 255) / 256), H, B * T), dim3(256), 0,
                       (hipStream_t)stream, a);
    return orv_check_launch("orv_vae_norm_apply");
}

// ---- tile seams of the tiled decode / encode (diffusers AutoencoderKLCogVideoX.blend_v / blend_h) ----
// b[.., y, :] = a[.., Ha - e + y, :] (1 - y / e) + b[.., y, :] (y / e)  for y < e  (vertical: a is the tile above, same width)
// b[.., :, x] = a[.., :, Wa - e + x] (1 - x / e) + b[.., :, x] (x / e)  for x < e  (horizontal: a is the tile to the left, same height)
// channels-last [outer = B * T, H, W, C] bf16, in place on b; one thread per element of the blended strip.
__global__ void vae_blend_kernel(const bf16_t* __restrict__ a, bf16_t* __restrict__ b, int outer, int Ha, int Wa, int Hb, int Wb, int C,
                                 int extent, int horizontal) {
    const int sh = horizontal ? Hb : extent, sw = horizontal ? extent : Wb;   // strip of b that changes
    const long n = (long)outer * sh * sw * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int x = (int)(r % sw); r /= sw;
        const int y = (int)(r % sh);
        const long o = r / sh;
        const int ya = horizontal ? y : Ha - extent + y, xa = horizontal ? Wa - extent + x : x;
        const float w = (float)(horizontal ? x : y) / (float)extent;
        const long ib = ((o * Hb + y) * Wb + x) * C + c, ia = ((o * Ha + ya) * Wa + xa) * C + c;
        b[ib] = f2bf(bf2f(a[ia]) * (1.f - w) + bf2f(b[ib]) * w);
    }
}
extern "C" int orv_vae_blend(const void* a, void* b, int outer, int Ha, int Wa, int Hb, int Wb, int C, int extent, int horizontal,
                             void* stream) {
    ORV_REQUIRE(a && b && outer > 0 && Ha > 0 && Wa > 0 && Hb > 0 && Wb > 0 && C > 0, "orv_vae_blend: bad shape");
    ORV_REQUIRE(extent > 0 && (horizontal ? (extent <= Wa && extent <= Wb && Ha == Hb) : (extent <= Ha && extent <= Hb && Wa == Wb)),
                "orv_vae_blend: extent %d does not fit the tiles (a %dx%d, b %dx%d, %s)", extent, Ha, Wa, Hb, Wb,
                horizontal ? "horizontal" : "vertical");
    const long n = (long)outer * (horizontal ? (long)Hb * extent : (long)extent * Wb) * C;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 65535 ? 65535 : (n + 255) / 256);
    hipLaunchKernelGGL(vae_blend_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (bf16_t*)b, outer, Ha, Wa, Hb,
                       Wb, C, extent, horizontal);
    return orv_check_launch("orv_vae_blend");
}
