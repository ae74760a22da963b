// CONTROL for the GEMM work: the 256 x 256 8-phase plain-HIP template of
// /opt/skills/guides/cdna_hip_programming.md (5, "The 256^2 8-phase template") written out literally from its description -
//   16x16x32 bf16 MFMA, 8 waves as 2 (M) x 4 (N), per-wave output 128 x 64, BK = 64, 128 KiB LDS = 2 buffers x {A, B} x 2
//   half-tiles of 128 rows x 64 k, st_16x32 subtiles (16 rows x 32 k = 1024 B, byte ^= ((byte >> 9) & 1) << 5), the swizzle on the
//   DMA source address and again on the ds_read address, 2 global_load_lds_dwordx4 per thread and half-tile, 8 phases per loop
//   trip (2 K-tiles), each {ds_read subtile | one half-tile of DMA | s_barrier | lgkmcnt(0) | setprio 1 | 16 MFMA | setprio 0 |
//   s_barrier}, vmcnt(6) in phases 4 and 8 only, the wr == 1 waves one barrier behind -
// and A/B'd IN ONE PROCESS, interleaved, against the product's gemm_ph_kernel<256, 0> (orv_gemm_bf16 with ORV_GEMM_TILE=2,256,256).
// Not part of the product library.  C = A[M,K] . W[N,K]^T, bf16 in / bf16 out, fp32 accumulate.
//   usage: probe_gemm_template [rounds]      (KB_ZERO=1: zero-filled operands)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "../include/orv_mi355.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} }while(0)

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef float orv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 orv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    orv_f32x2 v; v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, orv_bf16x2));
}
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// MAP: 0 = the product's XCD-aware super-tile order (GM = 4), 1 = the guide's plain remap (XCD chunk, row-major tiles)
template <int MAP>
__global__ __launch_bounds__(512) void gemm_tpl_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                       int M, int N, int K, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HALF = 16384, BUF = 65536;          // half-tile bytes; one buffer = A0 | A1 | B0 | B1
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    int tm, tn;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = b & 7, j = b >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        if (MAP == 0) {
            const int per = 4 * tiles_n, gid = L / per, rem = L % per, first_m = gid * 4, gsize = min(tiles_m - first_m, 4);
            tm = first_m + rem % gsize; tn = rem / gsize;
        } else { tm = L / tiles_n; tn = L % tiles_n; }
    }
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = K / 64;

    // ---- DMA source pointers.  Wave w moves 16-row block w of every half-tile (two subtiles: k halves 0 / 1).  Lane l writes the
    // 16 bytes at physical offset 16 l of the subtile, which hold logical byte 16 l ^ (((16 l >> 9) & 1) << 5) = 16 (l ^ ((l >> 5) << 1)).
    const int lsw = lane ^ ((lane >> 5) << 1);
    const int srow = lsw >> 2, schunk = lsw & 3;
    const int q = wave * 16 + srow;                   // row inside the 128-row half-tile
    // A half h: rows {wr' * 128 + h * 64 + (q & 63)}, wr' = q >> 6 ;  B half h: n = (q >> 5) * 64 + h * 32 + (q & 31)
    const bf16_t *pA[2], *pB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int arow = (q >> 6) * 128 + h * 64 + (q & 63);
        pA[h] = A + (long)min(m0 + arow, M - 1) * K + schunk * 8;
        const int brow = (q >> 5) * 64 + h * 32 + (q & 31);
        pB[h] = W + (long)(n0 + brow) * K + schunk * 8;
    }
    char* const dma_dst = smem + wave * 2048;         // + buffer * BUF + {0, HALF, 2 HALF, 3 HALF} + kh * 1024
    int kB1 = 0, kB0 = 0, kA0 = 0, kA1 = 0;           // K-tile cursor of each half-tile stream
#define ISSUE(PTR, KC, REG, S)                                                                    \
    {                                                                                             \
        const long ko_ = (long)min(KC, nk - 1) * 64;   /* past the end: re-fetch (never read) */  \
        glds16(PTR + ko_, dma_dst + (S) * BUF + (REG) * HALF);                                    \
        glds16(PTR + ko_ + 32, dma_dst + (S) * BUF + (REG) * HALF + 1024);                        \
        ++KC;                                                                                     \
    }

    // ---- fragment read addresses: logical byte (l & 15) * 64 + (l >> 4) * 16 of a subtile, bit 5 flipped for rows 8-15
    const int fro = ((lane & 15) * 64 + (lane >> 4) * 16) ^ (((lane >> 3) & 1) << 5);
    const char* const rdA = smem + (wr * 4) * 2048 + fro;             // + S*BUF + mh*HALF + mb*2048 + kh*1024
    const char* const rdB = smem + 2 * HALF + (wc * 2) * 2048 + fro;  // + S*BUF + nh*HALF + nb*2048 + kh*1024

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4][2], fb[2][2][2];                  // [mh][mb][kh], [nh][nb][kh]

#define READ_A(MH, S)                                                                             \
    _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)                                              \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                          \
            fa[MH][mb][kh] = *(const bf16x8*)(rdA + (S) * BUF + (MH) * HALF + mb * 2048 + kh * 1024);
#define READ_B(NH, S)                                                                             \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                              \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                          \
            fb[NH][nb][kh] = *(const bf16x8*)(rdB + (S) * BUF + (NH) * HALF + nb * 2048 + kh * 1024);
    // C^T form (W fragment as the A operand): D row = n, D column = m  ->  4 consecutive n per lane (8-byte bf16 stores)
#define MFMA_Q(MH, NH)                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                              \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)                                          \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                      \
                acc[MH][NH][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[NH][nb][kh], fa[MH][mb][kh], acc[MH][NH][mb][nb], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
#define BAR()                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define LGKM0()                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    __builtin_amdgcn_sched_barrier(0);
#define KTILE(S)                                                                                  \
    {                                                                                             \
        /* phase 1: 4 x B, 8 x A | B1 of the next K-tile | (m0, n0) */                            \
        READ_B(0, S)                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        READ_A(0, S)                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        ISSUE(pB[1], kB1, 3, (S) ^ 1)                                                             \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                        \
        BAR() LGKM0() MFMA_Q(0, 0) BAR()                                                          \
        /* phase 2: 8 x A | B0 of K-tile + 2 (its reads were retired by the lgkmcnt(8)) | (m1, n0) */ \
        READ_A(1, S)                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        ISSUE(pB[0], kB0, 2, S)                                                                   \
        BAR() LGKM0() MFMA_Q(1, 0) BAR()                                                          \
        /* phase 3: 4 x B | A0 of K-tile + 2 | (m1, n1) */                                        \
        READ_B(1, S)                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        ISSUE(pA[0], kA0, 0, S)                                                                   \
        BAR() LGKM0() MFMA_Q(1, 1) BAR()                                                          \
        /* phase 4: - | A1 of K-tile + 2 | (m0, n1); the one counted vmcnt per K-tile */          \
        ISSUE(pA[1], kA1, 1, S)                                                                   \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                          \
        BAR() MFMA_Q(0, 1) BAR()                                                                  \
    }

    // prologue: K-tile 0 complete into buffer 0; B0, A0, A1 of K-tile 1 into buffer 1
    ISSUE(pA[0], kA0, 0, 0) ISSUE(pB[0], kB0, 2, 0) ISSUE(pA[1], kA1, 1, 0) ISSUE(pB[1], kB1, 3, 0)
    ISSUE(pB[0], kB0, 2, 1) ISSUE(pA[0], kA0, 0, 1) ISSUE(pA[1], kA1, 1, 1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    BAR()
    if (wr == 1) { BAR() }
    for (int kt = 0; kt < nk; kt += 2) {
        KTILE(0)
        KTILE(1)
    }
    if (wr == 0) { BAR() }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // epilogue: lane holds C[m = ... + (l & 15)][n = ... + (l >> 4) * 4 + 0..3]
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int m = m0 + wr * 128 + mh * 64 + mb * 16 + (lane & 15);
            if (m >= M) continue;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int n = n0 + wc * 64 + nh * 32 + nb * 16 + (lane >> 4) * 4;
                    const f32x4 v = acc[mh][nh][mb][nb];
                    *(uint2*)(C + (long)m * N + n) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
        }
}

static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); u += 0x7fff + ((u>>16)&1); return (uint16_t)(u>>16);}
static inline float bf2f(uint16_t h){ uint32_t u=((uint32_t)h)<<16; float f; memcpy(&f,&u,4); return f;}
static std::vector<uint16_t> rnd_bf(size_t n, float scale, uint32_t seed, bool zero){
    if (zero) return std::vector<uint16_t>(n, 0);
    std::mt19937 g(seed); std::uniform_real_distribution<float> d(-1.f,1.f); std::vector<uint16_t> v(n); for(auto& x:v) x=f2bf(d(g)*scale); return v; }
template<class T> static T* up(const std::vector<T>& h){ T* d; CK(hipMalloc(&d,h.size()*sizeof(T))); CK(hipMemcpy(d,h.data(),h.size()*sizeof(T),hipMemcpyHostToDevice)); return d; }

static int g_map = 0;
static void launch_tpl(const uint16_t* A, const uint16_t* W, uint16_t* C, int M, int N, int K) {
    const int tiles_m = (M + 255) / 256, tiles_n = N / 256;
    static bool done = false;
    if (!done) {
        CK(hipFuncSetAttribute((const void*)gemm_tpl_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        CK(hipFuncSetAttribute((const void*)gemm_tpl_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        done = true;
    }
    if (g_map == 0) hipLaunchKernelGGL(gemm_tpl_kernel<0>, dim3(tiles_m * tiles_n), dim3(512), 131072, 0, A, W, C, M, N, K, tiles_m, tiles_n);
    else hipLaunchKernelGGL(gemm_tpl_kernel<1>, dim3(tiles_m * tiles_n), dim3(512), 131072, 0, A, W, C, M, N, K, tiles_m, tiles_n);
}
static void launch_ph(const uint16_t* A, const uint16_t* W, uint16_t* C, int M, int N, int K, int ring = 2) {
    orv_gemm_force_tile(ring, 256, 256);
    orv_gemm_t g{}; g.A = A; g.lda = K; g.W = W; g.ldw = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.epilogue = 0;
    if (orv_gemm_bf16(&g, nullptr)) { printf("orv_gemm_bf16: %s\n", orv_last_error()); exit(1); }
}

static int check(int M, int N, int K) {
    auto A = rnd_bf((size_t)M*K, 1.f, 11, false), W = rnd_bf((size_t)N*K, 0.05f, 12, false);
    uint16_t *dA = up(A), *dW = up(W), *dC; CK(hipMalloc(&dC, (size_t)M*N*2)); CK(hipMemset(dC, 0xff, (size_t)M*N*2));
    launch_tpl(dA, dW, dC, M, N, K); CK(hipDeviceSynchronize());
    std::vector<uint16_t> C((size_t)M*N); CK(hipMemcpy(C.data(), dC, C.size()*2, hipMemcpyDeviceToHost));
    std::mt19937 rg(7); double maxerr = 0, maxref = 0; const int nsamp = 40000;
    for (int s = 0; s < nsamp; ++s) {
        int m = rg() % M, n = rg() % N; if (s < 4000) { m = M - 1 - (s % 300) % M; }
        double acc = 0; for (int k = 0; k < K; ++k) acc += (double)bf2f(A[(size_t)m*K+k]) * bf2f(W[(size_t)n*K+k]);
        const double got = bf2f(C[(size_t)m*N+n]); maxerr = fmax(maxerr, fabs(got-acc)); maxref = fmax(maxref, fabs(acc));
    }
    const bool ok = maxerr <= 0.01 * maxref + 1e-3;
    printf("check template M=%d N=%d K=%d map=%d: maxerr=%.4g maxref=%.4g %s\n", M, N, K, g_map, maxerr, maxref, ok ? "OK" : "FAIL");
    hipFree(dA); hipFree(dW); hipFree(dC); return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    const bool pmc = getenv("TPL_PMC") != nullptr;     // under rocprofv3 --pmc: a handful of launches only
    if (orv_device_check(0)) { printf("%s\n", orv_last_error()); return 2; }
    int bad = 0;
    if (!pmc) for (int mp = 0; mp < 2; ++mp) { g_map = mp; bad += check(256, 256, 128); bad += check(1000, 512, 256); bad += check(3226, 1792, 1920); }
    if (bad) { printf("TEMPLATE CORRECTNESS FAILURES %d\n", bad); return 1; }
    g_map = 0;
    struct Shape { int M, N, K; } shapes[] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {12904, 7680, 1920}};
    for (int zero = 0; zero < 2; ++zero)
        for (const Shape& s : shapes) {
            auto A = rnd_bf((size_t)s.M*s.K, 1.f, 1, zero), W = rnd_bf((size_t)s.N*s.K, 0.05f, 2, zero);
            uint16_t *dA = up(A), *dW = up(W), *dC; CK(hipMalloc(&dC, (size_t)s.M*s.N*2));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int iters = pmc ? 2 : (s.K >= 8192 ? 8 : 30);
            std::vector<double> tf[4];
            for (int r = 0; r < (pmc ? 1 : rounds); ++r)
                for (int v = 0; v < 4; ++v) {           // 0 = template (product tile map), 1 = template (guide's map), 2 = gemm_ph_kernel<256,0>, 3 = gemm_t8_kernel<256,0>
                    g_map = v == 1 ? 1 : 0;
                    auto go = [&]() { if (v >= 2) launch_ph(dA, dW, dC, s.M, s.N, s.K, v); else launch_tpl(dA, dW, dC, s.M, s.N, s.K); };
                    for (int i = 0; i < (pmc ? 0 : 3); ++i) go();
                    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
                    tf[v].push_back(2.0 * s.M * s.N * s.K / ms / 1e9);
                }
            const char* names[4] = {"template(map=product)", "template(map=guide)  ", "gemm_ph_kernel<256,0>", "gemm_t8_kernel<256,0>"};
            for (int v = 0; v < 4; ++v) {
                std::sort(tf[v].begin(), tf[v].end());
                printf("%s M=%5d N=%5d K=%5d %s: median %.0f  min %.0f  max %.0f TFLOP/s (%zu rounds)\n", zero ? "zero  " : "random", s.M, s.N, s.K,
                       names[v], tf[v][tf[v].size()/2], tf[v].front(), tf[v].back(), tf[v].size());
            }
            hipFree(dA); hipFree(dW); hipFree(dC);
        }
    return 0;
}
