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
#include "common.hpp"

namespace {

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
    int tiles_m, tiles_n;
};

constexpr int BK = 64;
constexpr int GM = 4;  // super-tile height in tiles

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(512) void gemm_kernel(const GemmArgs p) {
    constexpr int MB = BM / 128;       // 32-row blocks per wave along M (4 M-waves)
    constexpr int NB = BN / 64;        // 32-col blocks per wave along N (2 N-waves)
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
    auto stage_load = [&](int s) {
        char* sa = smem + s * STAGE + wave * A_LD * 1024;
        char* sb = smem + s * STAGE + A_BYTES + wave * B_LD * 1024;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) { glds16(a_src[j], sa + j * 1024); a_src[j] += BK; }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) { glds16(b_src[j], sb + j * 1024); b_src[j] += BK; }
    };

    // ---- fragment read offsets ----
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

    const int nk = p.K / BK;
    stage_load(0);
    for (int t = 0; t < nk; ++t) {
        // tile t has landed for this wave; after the barrier it has landed for all waves and nobody still
        // reads stage (t+1)&1 (those reads were consumed by the MFMAs of iteration t-1).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nk) stage_load((t + 1) & 1);
        const char* sbase = smem + (t & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = ((ks * 2 + hi) ^ sw) * 16;
            bf16x8 af[MB], bf[NB];
#pragma unroll
            for (int j = 0; j < MB; ++j) af[j] = *(const bf16x8*)(sbase + a_row_off + j * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < NB; ++i) bf[i] = *(const bf16x8*)(sbase + b_row_off + i * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < MB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[i], af[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: acc[i][j][4q+e] = C[m][n], m = m0 + wm*BM/4 + j*32 + l31, n = n0 + wn*BN/2 + i*32 + 8q + 4hi + e
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int m = m0 + wm * (BM / 4) + j * 32 + l31;
        if (m >= p.M) continue;
        long orow = m;
        if (p.c_rows > 0) orow = (long)(m / p.c_rows) * p.c_bstride + p.c_off + m % p.c_rows;
        bf16_t* crow = p.C + orow * p.ldc;
        const bf16_t* rrow = nullptr;
        const float* grow = nullptr;
        if (EPI == 2) {
            const long rr = p.r_mod > 0 ? m % p.r_mod : orow;
            rrow = p.R + rr * p.ldr;
            if (p.gate) {
                const int bidx = (int)(orow / p.seq), s = (int)(orow % p.seq);
                grow = p.gate + bidx * p.gate_b + orv_group_of(s, p.n_text, p.per_group) * p.gate_g;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * (BN / 2) + i * 32 + q * 8 + hi * 4;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
                if (p.bias) {
                    const uint2 bb = *(const uint2*)(p.bias + n);
                    v[0] += bf2f(bb.x & 0xffff); v[1] += bf2f(bb.x >> 16);
                    v[2] += bf2f(bb.y & 0xffff); v[3] += bf2f(bb.y >> 16);
                }
                if (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
                }
                if (EPI == 2) {
                    const uint2 rr = *(const uint2*)(rrow + n);
                    float g[4] = {1.f, 1.f, 1.f, 1.f};
                    if (grow) { const float4 gg = *(const float4*)(grow + n); g[0] = gg.x; g[1] = gg.y; g[2] = gg.z; g[3] = gg.w; }
                    v[0] = bf2f(rr.x & 0xffff) + g[0] * v[0]; v[1] = bf2f(rr.x >> 16) + g[1] * v[1];
                    v[2] = bf2f(rr.y & 0xffff) + g[2] * v[2]; v[3] = bf2f(rr.y >> 16) + g[3] * v[3];
                }
                uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
                *(uint2*)(crow + n) = o;
            }
        }
    }
}

template <int BM, int BN>
int launch(const GemmArgs& a, int epi, hipStream_t st) {
    const int smem = 2 * (BM + BN) * 128;
    const int grid = a.tiles_m * a.tiles_n;
#define ORV_GEMM_CASE(E)                                                                                    \
    case E: {                                                                                               \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, E>,                                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem);                    \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((gemm_kernel<BM, BN, E>), dim3(grid), dim3(512), smem, st, a);                   \
        break;                                                                                              \
    }
    switch (epi) {
        ORV_GEMM_CASE(0)
        ORV_GEMM_CASE(1)
        ORV_GEMM_CASE(2)
        default: orv_set_error("orv_gemm_bf16: bad epilogue %d", epi); return ORV_EINVAL;
    }
#undef ORV_GEMM_CASE
    return orv_check_launch("orv_gemm_bf16");
}

}  // namespace

// Tile choice.  BN must divide N: 192 divides 1920/5760/7680 (the 2B model) exactly, 128 covers 3072-wide (5B) and
// the tiny test widths, 64 the 64-wide proj_out.  BM = 256 unless the grid would leave most CUs idle.
extern "C" int orv_gemm_bf16(const orv_gemm_t* g, void* stream) {
    ORV_REQUIRE(g && g->A && g->W && g->C, "orv_gemm_bf16: null operand");
    ORV_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, "orv_gemm_bf16: empty problem M=%d N=%d K=%d", g->M, g->N, g->K);
    ORV_REQUIRE(g->K % 64 == 0, "orv_gemm_bf16: K=%d must be a multiple of 64", g->K);
    ORV_REQUIRE(g->N % 64 == 0, "orv_gemm_bf16: N=%d must be a multiple of 64", g->N);
    ORV_REQUIRE(g->lda % 8 == 0 && g->ldw % 8 == 0 && g->ldc % 4 == 0, "orv_gemm_bf16: misaligned leading dimension");
    ORV_REQUIRE(g->epilogue != 2 || g->R, "orv_gemm_bf16: epilogue 2 needs R");
    ORV_REQUIRE(g->epilogue != 2 || !g->gate || g->grp.seq > 0, "orv_gemm_bf16: gate needs grp.seq");
    GemmArgs a;
    a.A = (const bf16_t*)g->A; a.lda = g->lda; a.W = (const bf16_t*)g->W; a.ldw = g->ldw;
    a.bias = (const bf16_t*)g->bias; a.C = (bf16_t*)g->C; a.ldc = g->ldc;
    a.M = g->M; a.N = g->N; a.K = g->K;
    a.R = (const bf16_t*)g->R; a.ldr = g->ldr; a.r_mod = g->r_mod;
    a.gate = g->gate; a.gate_b = g->gate_b; a.gate_g = g->gate_g;
    a.seq = g->grp.seq; a.n_text = g->grp.n_text; a.per_group = g->grp.per_group;
    a.c_rows = g->cmap.rows; a.c_bstride = g->cmap.bstride; a.c_off = g->cmap.off;
    hipStream_t st = (hipStream_t)stream;
    const int bn = (g->N % 192 == 0) ? 192 : (g->N % 128 == 0 ? 128 : 64);
    a.tiles_n = g->N / bn;
    const int tiles256 = ((g->M + 255) / 256) * a.tiles_n;
    const bool big = tiles256 >= 224;  // ~one full wave of workgroups over the 256 CUs
    a.tiles_m = big ? (g->M + 255) / 256 : (g->M + 127) / 128;
    if (bn == 192) return big ? launch<256, 192>(a, g->epilogue, st) : launch<128, 192>(a, g->epilogue, st);
    if (bn == 128) return big ? launch<256, 128>(a, g->epilogue, st) : launch<128, 128>(a, g->epilogue, st);
    return big ? launch<256, 64>(a, g->epilogue, st) : launch<128, 64>(a, g->epilogue, st);
}
