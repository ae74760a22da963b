// Small kernels around the DiT trunk (all HBM/latency-bound, no MFMA):
//   orv_timestep_embedding, orv_skinny_linear, orv_patchify, orv_unpatchify, orv_sched_step, orv_gaussian_sample
#include "common.hpp"

namespace {

__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int batch, int dim,
                                          int flip, float shift) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * half) return;
    const int b = i / half, k = i % half;
    const float freq = expf(-logf(10000.f) * (float)k / ((float)half - shift));
    const float a = t[b] * freq;
    const float s = sinf(a), c = cosf(a);
    bf16_t* o = out + (long)b * dim;
    if (flip) { o[k] = f2bf(c); o[half + k] = f2bf(s); }
    else { o[k] = f2bf(s); o[half + k] = f2bf(c); }
    if ((dim & 1) && k == 0) o[dim - 1] = 0;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    return act == 1 ? silu(v) : (act == 2 ? gelu_tanh(v) : v);
}

// Block = 256 threads (4 waves), handles MT=16 rows x NPB output columns. x tile staged once in LDS as fp32.
constexpr int SK_MT = 16;
constexpr int SK_NPW = 8;   // output columns per wave at most (the launcher lowers it until the grid covers the CUs)
__global__ __launch_bounds__(256) void skinny_linear_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ xb,
                                                            int xb_rep, const bf16_t* __restrict__ W,
                                                            const bf16_t* __restrict__ bias, void* __restrict__ out,
                                                            int M, int N, int K, int act_in, int act_out, int out_f32,
                                                            long ldo, orv_rowmap_t omap, int npw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = (bf16_t*)smem;  // [SK_MT][K] bf16, act_in applied
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * SK_MT;
    const int mt = min(SK_MT, M - m0);
    for (int r = 0; r < SK_MT; ++r)
        for (int k = tid; k < K; k += 256) {
            float v = 0.f;
            if (r < mt) {
                v = bf2f(x[(long)(m0 + r) * K + k]);
                if (xb) v = bf2f(f2bf(v + bf2f(xb[(long)((m0 + r) / xb_rep) * K + k])));  // the reference adds in the model dtype
                v = apply_act(v, act_in);
            }
            xs[r * K + k] = f2bf(v);
        }
    __syncthreads();
    const int nchunk = (K % 8 == 0) ? (K >> 3) : 0;  // rows are 16-byte aligned only when K % 8 == 0
    for (int j = 0; j < npw; ++j) {
        const int n = (blockIdx.x * 4 + wave) * npw + j;
        if (n >= N) break;
        float acc[SK_MT];
#pragma unroll
        for (int r = 0; r < SK_MT; ++r) acc[r] = 0.f;
        const bf16_t* wr = W + (long)n * K;
        for (int c = lane; c < nchunk; c += 64) {
            const uint4 u = *(const uint4*)(wr + c * 8);
            const uint32_t ww[4] = {u.x, u.y, u.z, u.w};
            float wv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { wv[2 * e] = bf2f(ww[e] & 0xffff); wv[2 * e + 1] = bf2f(ww[e] >> 16); }
#pragma unroll
            for (int r = 0; r < SK_MT; ++r) {
                const uint4 xu = *(const uint4*)(xs + r * K + c * 8);
                const uint32_t xw[4] = {xu.x, xu.y, xu.z, xu.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[r] = fmaf(bf2f(xw[e] & 0xffff), wv[2 * e], acc[r]);
                    acc[r] = fmaf(bf2f(xw[e] >> 16), wv[2 * e + 1], acc[r]);
                }
            }
        }
        // scalar tail when K is not a multiple of 8 (ActionEmbed: K = 28)
        for (int k = (nchunk << 3) + lane; k < K; k += 64) {
            const float wv = bf2f(wr[k]);
#pragma unroll
            for (int r = 0; r < SK_MT; ++r) acc[r] = fmaf(bf2f(xs[r * K + k]), wv, acc[r]);
        }
        const float bv = bias ? bf2f(bias[n]) : 0.f;
#pragma unroll
        for (int r = 0; r < SK_MT; ++r) {
            const float v = wave_sum(acc[r]);
            if (lane == 0 && r < mt) {
                const float o = apply_act(v + bv, act_out);
                const int m = m0 + r;
                const long orow = omap.rows > 0 ? (long)(m / omap.rows) * omap.bstride + omap.off + m % omap.rows : m;
                if (out_f32) ((float*)out)[orow * ldo + n] = o;
                else ((bf16_t*)out)[orow * ldo + n] = f2bf(o);
            }
        }
    }
}

// tokens[b, tok, f] <- src[b, t, c, y, x]; pure index map, one thread per element
__global__ void patchify_kernel(const bf16_t* __restrict__ s0, int c0, const bf16_t* __restrict__ s1, int c1,
                                bf16_t* __restrict__ tok, int B, int T, int H, int W, int p, int pt, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int C = c0 + c1, h = H / p, w = W / p, ptt = pt > 0 ? pt : 1;
    const int F = C * ptt * p * p;
    const int f = (int)(i % F);
    long r = i / F;
    const int ntok = (T / ptt) * h * w;
    const int tk = (int)(r % ntok), b = (int)(r / ntok);
    const int xx = tk % w, yy = (tk / w) % h, tq = tk / (w * h);
    const int bb = f % p, a = (f / p) % p, tt = (f / (p * p)) % ptt, c = f / (p * p * ptt);
    const int t = tq * ptt + tt, y = yy * p + a, x = xx * p + bb;
    bf16_t v;
    if (c < c0) v = s0[((((long)b * T + t) * c0 + c) * H + y) * W + x];
    else v = s1[((((long)b * T + t) * c1 + (c - c0)) * H + y) * W + x];
    tok[i] = v;
}

// out[b, t, c, y, x] <- x[b, tok, f]; one thread per output element
__global__ void unpatchify_kernel(const bf16_t* __restrict__ xin, bf16_t* __restrict__ out, int B, int T, int C, int H,
                                  int W, int p, int pt, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int h = H / p, w = W / p, ptt = pt > 0 ? pt : 1;
    const int x = (int)(i % W);
    long r = i / W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % C); r /= C;
    const int t = (int)(r % T), b = (int)(r / T);
    const int xx = x / p, bb = x % p, yy = y / p, a = y % p, tq = t / ptt, tt = t % ptt;
    const int ntok = ((T + ptt - 1) / ptt) * h * w;
    const int F = C * ptt * p * p;
    const int tk = (tq * h + yy) * w + xx;
    const int f = ((c * ptt + tt) * p + a) * p + bb;
    out[i] = xin[((long)b * ntok + tk) * F + f];
}

__global__ void sched_step_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ vc,
                                  const bf16_t* __restrict__ vu, float gs, const float* __restrict__ old_x0,
                                  const float* __restrict__ noise, bf16_t* __restrict__ xo, float* __restrict__ x0o,
                                  float sa, float sb, float m3, float m4, float cx, float cd, float cn, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = bf2f(x[i]);
    float v = bf2f(vc[i]);
    if (vu) { const float u = bf2f(vu[i]); v = u + gs * (v - u); }
    const float x0 = sa * xv - sb * v;
    const float d = old_x0 ? m3 * x0 - m4 * old_x0[i] : x0;
    float r = cx * xv + cd * d;
    if (noise) r += cn * noise[i];
    xo[i] = f2bf(r);
    if (x0o) x0o[i] = x0;
}

// out[b, f, c, hw] = (mean + exp(0.5 clamp(logvar)) eps) * scale ; moments [b, 2C, f, hw]
__global__ void gaussian_sample_kernel(const bf16_t* __restrict__ mom, const float* __restrict__ eps,
                                       bf16_t* __restrict__ out, int B, int C, int F, int HW, float scale, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int hw = (int)(i % HW);
    long r = i / HW;
    const int c = (int)(r % C); r /= C;
    const int f = (int)(r % F), b = (int)(r / F);
    const long mi = (((long)b * 2 * C + c) * F + f) * HW + hw;
    const long li = (((long)b * 2 * C + C + c) * F + f) * HW + hw;
    const long ei = (((long)b * C + c) * F + f) * HW + hw;
    const float lv = fminf(fmaxf(bf2f(mom[li]), -30.f), 20.f);
    out[i] = f2bf((bf2f(mom[mi]) + expf(0.5f * lv) * eps[ei]) * scale);
}

// out[m, col_off + d] = a[amap(m), d] + b[m, d]  (16-byte vectors)
__global__ void add_rows_kernel(const bf16_t* __restrict__ a, long lda, orv_rowmap_t amap, const bf16_t* __restrict__ b,
                                long ldb, bf16_t* __restrict__ out, long ldo, int col_off, int M, int D) {
    const int nchunk = D >> 3;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * nchunk) return;
    const int m = (int)(i / nchunk), c = (int)(i % nchunk);
    const long ar = amap.rows > 0 ? (long)(m / amap.rows) * amap.bstride + amap.off + m % amap.rows : m;
    const uint4 ua = *(const uint4*)(a + ar * lda + c * 8);
    const uint4 ub = *(const uint4*)(b + (long)m * ldb + c * 8);
    const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        o[e] = pack2bf(bf2f(wa[e] & 0xffff) + bf2f(wb[e] & 0xffff), bf2f(wa[e] >> 16) + bf2f(wb[e] >> 16));
    *(uint4*)(out + (long)m * ldo + col_off + c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

// dst[r, :] = src[idx[r], :]
__global__ void gather_rows_kernel(const bf16_t* __restrict__ src, long lds_, const int* __restrict__ idx,
                                   bf16_t* __restrict__ dst, long ldd, int R, int D) {
    const int nchunk = D >> 3;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)R * nchunk) return;
    const int r = (int)(i / nchunk), c = (int)(i % nchunk);
    *(uint4*)(dst + (long)r * ldd + c * 8) = *(const uint4*)(src + (long)idx[r] * lds_ + c * 8);
}
// x[idx[r], :] += gate[(idx[r] / seq) * gate_b + :] * y[r, :]   for rows whose target is a video row ((idx[r] % seq) >= n_text)
__global__ void scatter_gated_rows_kernel(const bf16_t* __restrict__ y, long ldy, const int* __restrict__ idx,
                                          const float* __restrict__ gate, long gate_b, bf16_t* __restrict__ x, long ldx_,
                                          int R, int D, int seq, int n_text) {
    const int nchunk = D >> 3;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)R * nchunk) return;
    const int r = (int)(i / nchunk), c = (int)(i % nchunk);
    const int t = idx[r];
    if (t % seq < n_text) return;
    const float* g = gate + (long)(t / seq) * gate_b + c * 8;
    const uint4 uy = *(const uint4*)(y + (long)r * ldy + c * 8);
    bf16_t* xp = x + (long)t * ldx_ + c * 8;
    const uint4 ux = *(const uint4*)xp;
    const uint32_t wy[4] = {uy.x, uy.y, uy.z, uy.w}, wx[4] = {ux.x, ux.y, ux.z, ux.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        o[e] = pack2bf(bf2f(wx[e] & 0xffff) + g[2 * e] * bf2f(wy[e] & 0xffff), bf2f(wx[e] >> 16) + g[2 * e + 1] * bf2f(wy[e] >> 16));
    *(uint4*)xp = make_uint4(o[0], o[1], o[2], o[3]);
}

}  // namespace

extern "C" int orv_gather_rows(const void* src, int ld_src, const int* idx, void* dst, int ld_dst, int R, int D, void* stream) {
    ORV_REQUIRE(src && idx && dst && R > 0 && D > 0 && D % 8 == 0, "orv_gather_rows: bad arguments");
    const long total = (long)R * (D / 8);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (long)ld_src, idx, (bf16_t*)dst, (long)ld_dst, R, D);
    return orv_check_launch("orv_gather_rows");
}

extern "C" int orv_scatter_gated_rows(const void* y, int ldy, const int* idx, const float* gate, long gate_b, void* x, int ldx,
                                      int R, int D, int seq, int n_text, void* stream) {
    ORV_REQUIRE(y && idx && gate && x && R > 0 && D > 0 && D % 8 == 0 && seq > 0, "orv_scatter_gated_rows: bad arguments");
    const long total = (long)R * (D / 8);
    hipLaunchKernelGGL(scatter_gated_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)y, (long)ldy, idx, gate, gate_b, (bf16_t*)x, (long)ldx, R, D, seq, n_text);
    return orv_check_launch("orv_scatter_gated_rows");
}

extern "C" int orv_add_rows(const void* a, int lda, orv_rowmap_t amap, const void* b, int ldb, void* out, int ldo,
                            int col_off, int M, int D, void* stream) {
    ORV_REQUIRE(a && b && out && M > 0 && D > 0, "orv_add_rows: bad arguments");
    ORV_REQUIRE(D % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0 && col_off % 8 == 0, "orv_add_rows: misaligned");
    const long total = (long)M * (D / 8);
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)a, (long)lda, amap, (const bf16_t*)b, (long)ldb, (bf16_t*)out, (long)ldo, col_off, M, D);
    return orv_check_launch("orv_add_rows");
}

extern "C" int orv_timestep_embedding(const float* t, void* out_bf16, int batch, int dim, int flip_sin_to_cos,
                                      float freq_shift, void* stream) {
    ORV_REQUIRE(t && out_bf16 && batch > 0 && dim >= 2, "orv_timestep_embedding: bad arguments");
    const int n = batch * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t,
                       (bf16_t*)out_bf16, batch, dim, flip_sin_to_cos, freq_shift);
    return orv_check_launch("orv_timestep_embedding");
}

extern "C" int orv_skinny_linear(const void* x, const void* xb, int xb_rep, const void* W, const void* bias, void* out,
                                 int M, int N, int K, int act_in, int act_out, int out_f32, int ldo, orv_rowmap_t omap,
                                 void* stream) {
    ORV_REQUIRE(x && W && out, "orv_skinny_linear: null operand");
    ORV_REQUIRE(M > 0 && M <= 4096 && N > 0 && K > 0, "orv_skinny_linear: bad shape M=%d N=%d K=%d", M, N, K);
    ORV_REQUIRE(K % 4 == 0 && (K % 8 == 0 || K <= 64), "orv_skinny_linear: K=%d unsupported", K);
    ORV_REQUIRE(SK_MT * K * 2 <= 160 * 1024, "orv_skinny_linear: K=%d too large for the LDS tile", K);
    ORV_REQUIRE(!xb || xb_rep > 0, "orv_skinny_linear: xb_rep must be > 0");
    const int smem = SK_MT * K * 2;
    static int smem_max = 0;
    if (smem > smem_max && smem > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)skinny_linear_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        smem_max = smem;
    }
    // rows of W must be 16-byte aligned for the vector path: K % 8 == 0; otherwise the scalar tail handles K < 64
    // columns per wave: as many as keep >= 256 workgroups in flight (N = 512: one column per wave, 128 workgroups instead of 16)
    const int mblocks = (M + SK_MT - 1) / SK_MT;
    int npw = SK_NPW;
    while (npw > 1 && (long)((N + 4 * npw - 1) / (4 * npw)) * mblocks < 256) npw >>= 1;
    dim3 grid((N + 4 * npw - 1) / (4 * npw), mblocks);
    hipLaunchKernelGGL(skinny_linear_kernel, grid, dim3(256), smem, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)xb, xb_rep, (const bf16_t*)W, (const bf16_t*)bias, out, M, N, K, act_in, act_out,
                       out_f32, (long)ldo, omap, npw);
    return orv_check_launch("orv_skinny_linear");
}

extern "C" int orv_patchify(const void* src0, int c0, const void* src1, int c1, void* tokens, int B, int T, int H, int W,
                            int p, int pt, void* stream) {
    ORV_REQUIRE(src0 && tokens && c0 > 0 && (c1 == 0 || src1), "orv_patchify: null operand");
    ORV_REQUIRE(B > 0 && T > 0 && p > 0 && H % p == 0 && W % p == 0, "orv_patchify: bad shape");
    ORV_REQUIRE(pt == 0 || T % pt == 0, "orv_patchify: T=%d not divisible by patch_size_t=%d", T, pt);
    const long total = (long)B * T * (c0 + c1) * H * W;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src0, c0, (const bf16_t*)src1, c1, (bf16_t*)tokens, B, T, H, W, p, pt, total);
    return orv_check_launch("orv_patchify");
}

extern "C" int orv_unpatchify(const void* x, void* out, int B, int T, int C, int H, int W, int p, int pt, void* stream) {
    ORV_REQUIRE(x && out, "orv_unpatchify: null operand");
    ORV_REQUIRE(B > 0 && T > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, "orv_unpatchify: bad shape");
    ORV_REQUIRE(pt == 0 || T % pt == 0, "orv_unpatchify: T=%d not divisible by patch_size_t=%d", T, pt);
    const long total = (long)B * T * C * H * W;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)out, B, T, C, H, W, p, pt, total);
    return orv_check_launch("orv_unpatchify");
}

extern "C" int orv_sched_step(const void* x, const void* v_c, const void* v_u, float guidance_scale, const float* old_x0,
                              const float* noise, void* x_out, float* x0_out, float sa, float sb, float m3, float m4,
                              float cx, float cd, float cn, long n, void* stream) {
    ORV_REQUIRE(x && v_c && x_out && n > 0, "orv_sched_step: bad arguments");
    hipLaunchKernelGGL(sched_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)v_c, (const bf16_t*)v_u, guidance_scale, old_x0, noise,
                       (bf16_t*)x_out, x0_out, sa, sb, m3, m4, cx, cd, cn, n);
    return orv_check_launch("orv_sched_step");
}

extern "C" int orv_gaussian_sample(const void* moments, const float* eps, void* out, int B, int C, int F, int HW,
                                   float scale, void* stream) {
    ORV_REQUIRE(moments && eps && out && B > 0 && C > 0 && F > 0 && HW > 0, "orv_gaussian_sample: bad arguments");
    const long total = (long)B * F * C * HW;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)moments, eps, (bf16_t*)out, B, C, F, HW, scale, total);
    return orv_check_launch("orv_gaussian_sample");
}
