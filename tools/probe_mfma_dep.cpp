// Probe: cost of DEPENDENT v_mfma_f32_32x32x16_bf16 chains as a function of the distance between two MFMAs on the same
// accumulator (NACC accumulators used round-robin: distance NACC), one wave per SIMD (256 threads / CU) so nothing else hides
// the stall.  Prints cycles per MFMA (s_memtime ticks are shader cycles).  Motivation: the attention kernels alternate two
// accumulators (distance 2) in their matrix segments.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k(const bf16x8* in, float* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    bf16x8 a = in[(t * 7) & 4095], b = in[(t * 5 + 64) & 4095];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[t] = s;
    if (t == 0) ((unsigned long long*)out)[70000] = t1 - t0;
}
template <int NACC> void run(const bf16x8* in, float* out) {
    const int iters = 20000;
    k<NACC><<<256, 256>>>(in, out, iters);
    hipDeviceSynchronize();
    unsigned long long mt; hipMemcpy(&mt, (char*)out + 70000 * 8, 8, hipMemcpyDeviceToHost);
    printf("accumulators round-robin = %d (dependent distance %d): %.1f cycles per MFMA\n", NACC, NACC, (double)mt / iters / 48.0);
}
int main() {
    bf16x8* in; float* out; hipMalloc(&in, 4096 * 16); hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20);
    unsigned short h[4096 * 8]; for (int i = 0; i < 4096 * 8; i++) h[i] = 0x3c00 + (rand() & 0x3ff);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<1>(in, out); run<2>(in, out); run<3>(in, out); run<4>(in, out); run<6>(in, out); run<8>(in, out);
    return 0;
}
