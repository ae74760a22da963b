// Probe (round 5, gemm_d8 design): what does the SHAPE of a global_load_dwordx4 cost on the CU's texture path?  One wave instruction moves
// 1 KiB in all three shapes; rows are ROWSTRIDE bytes apart (a K-contiguous GEMM operand), the workgroup walks along K, its window is L2-resident.
//   shape 0  "fragment": lane l -> row l & 15, 16-byte chunk l >> 4   (the MFMA operand layout: the 4 lanes of a quad sit in 4 different rows)
//   shape 1  "quad":     lane l -> row l >> 2, chunk l & 3            (16 rows x 64 B, a quad = 64 contiguous bytes)
//   shape 2  "line":     lane l -> row l >> 3, chunk l & 7            (8 rows x one full 128-byte line)
// Two instructions per row block in shapes 0 / 1 (the two 64-byte halves of the lines), one per 8 rows in shape 2: the same bytes, the same lines.
// Also: ds_bpermute_b32 throughput (the in-register fix-up from shape 1 to shape 0 costs 4 per loaded 16 bytes), 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void kload(const char* src, long rowstride, int kbytes, int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // every workgroup of an XCD reads the same 256 rows (panel sharing as in a GEMM); a wave owns 32 of them
    const long row0 = (long)(blockIdx.x & 7) * 256 + wave * 32;
    unsigned off[4];
    if (SHAPE == 0) { for (int j = 0; j < 4; ++j) off[j] = (unsigned)((row0 + (j >> 1) * 16 + (lane & 15)) * rowstride + (j & 1) * 64 + (lane >> 4) * 16); }
    if (SHAPE == 1) { for (int j = 0; j < 4; ++j) off[j] = (unsigned)((row0 + (j >> 1) * 16 + (lane >> 2)) * rowstride + (j & 1) * 64 + (lane & 3) * 16); }
    if (SHAPE == 2) { for (int j = 0; j < 4; ++j) off[j] = (unsigned)((row0 + j * 8 + (lane >> 3)) * rowstride + (lane & 7) * 16); }
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
        for (int ko = 0; ko < kbytes; ko += 256) {          // two K-tiles per trip: 8 loads in flight per wave
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const u32x4*)(src + off[j & 3] + ko + (j >> 2) * 128);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j];
        }
    if (acc[0] == 0x12345678u && acc[1] == 77u) sink[0] = acc[2] ^ acc[3];
}
template <int N>
__global__ __launch_bounds__(512) void kperm(int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    const int addr = 4 * (4 * (lane & 15) + (lane >> 4));
    unsigned v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = lane * 7 + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_ds_bpermute(addr, v[j]);
    }
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) s ^= v[j];
    if (s == 0x12345678u) sink[0] = s;
}
template <int SHAPE> void run(const char* src, long rowstride, int kbytes, int iters, unsigned* sink, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kload<SHAPE><<<256, 512>>>(src, rowstride, kbytes, 2, sink);
    CK(hipEventRecord(e0)); kload<SHAPE><<<256, 512>>>(src, rowstride, kbytes, iters, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * 8 * 32 * kbytes * iters;       // per CU: 8 waves x 32 rows x kbytes
    printf("%-10s rowstride %6ld: %.3f ms  %.1f GB/s per CU  (%.1f B/clk/CU at 2.0 GHz)  chip %.2f TB/s\n", name, rowstride, ms, bytes / 256 / ms / 1e6,
           bytes / 256 / ms / 1e6 / 2.0, bytes / ms / 1e9);
}
int main() {
    const long rowstride = 3840; const int kbytes = 3840;      // K = 1920 bf16
    char* src; CK(hipMalloc(&src, (size_t)2048 * rowstride + 4096)); CK(hipMemset(src, 1, (size_t)2048 * rowstride + 4096));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    for (int r = 0; r < 2; ++r) {
        run<0>(src, rowstride, kbytes, 200, sink, "fragment");
        run<1>(src, rowstride, kbytes, 200, sink, "quad");
        run<2>(src, rowstride, kbytes, 200, sink, "line");
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) {
        const int iters = 20000;
        kperm<16><<<256, 512>>>(10, sink);
        CK(hipEventRecord(e0)); kperm<16><<<256, 512>>>(iters, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("ds_bpermute_b32: %.3f ms for %d x 16 per wave, 8 waves per CU: %.2f ns per wave-instruction and CU (%.1f cycles at 2.0 GHz)\n", ms, iters,
               ms * 1e6 / ((double)iters * 16 * 8), ms * 1e6 / ((double)iters * 16 * 8) * 2.0);
    }
    return 0;
}
