// Probe: sustained bf16 MFMA rate of the two shapes under the power cap, operands in registers only (no LDS, no global traffic in
// the loop): v_mfma_f32_32x32x16_bf16 (wave tile 128 x 64 as 4 x 2 blocks: 6 fragments, 8 MFMAs per k = 16) against
// v_mfma_f32_16x16x32_bf16 (the same wave tile as 8 x 4 blocks: 12 fragments, 32 MFMAs per k = 32), random and zero operands.
// Which shape gives more FLOPs per joule on this part decides the GEMM kernels' inner product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(512) void k32(const bf16x8* in, float* out, int iters) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; i++) a[i] = in[(t * 7 + i) & 4095];
  for (int i = 0; i < 2; i++) b[i] = in[(t * 5 + i + 64) & 4095];
  f32x16 acc[8];
  for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++)          // two k = 16 steps = one k = 32 step of the other shape
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[n], a[m], acc[m * 2 + n], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
  out[t] = s;
}
__global__ __launch_bounds__(512) void k16(const bf16x8* in, float* out, int iters) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  bf16x8 a[8], b[4];
  for (int i = 0; i < 8; i++) a[i] = in[(t * 7 + i) & 4095];
  for (int i = 0; i < 4; i++) b[i] = in[(t * 5 + i + 64) & 4095];
  f32x4 acc[32];
  for (int i = 0; i < 32; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 8; m++)
#pragma unroll
      for (int n = 0; n < 4; n++) acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[m * 4 + n], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 32; i++) for (int e = 0; e < 4; e++) s += acc[i][e];
  out[t] = s;
}
template <class K> double run(K kern, int iters, const bf16x8* in, float* out, int threads = 512) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<256, threads>>>(in, out, iters);
  hipEventRecord(e0); for (int i = 0; i < 3; i++) kern<<<256, threads>>>(in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  return 256.0 * (threads / 64) * iters * 32.0 * 16384.0 / ms / 1e9;      // both kernels: 128 x 64 x 32 MACs x 2 per iteration and wave
}
int main() {
  bf16x8* in; float* out; hipMalloc(&in, 4096 * 16); hipMalloc(&out, 256 * 512 * 4);
  static unsigned short h[4096 * 8];
  for (int i = 0; i < 4096 * 8; i++) h[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15) - ((rand() & 3) << 7);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; rep++)
    printf("random operands: 32x32x16 %.0f TFLOP/s | 16x16x32 %.0f TFLOP/s\n", run(k32, 60000, in, out), run(k16, 60000, in, out));
  printf("one wave per SIMD (256 threads), random: 32x32x16 %.0f TFLOP/s | 16x16x32 %.0f TFLOP/s\n", run(k32, 60000, in, out, 256), run(k16, 60000, in, out, 256));
  hipMemset(in, 0, 4096 * 16);
  printf("one wave per SIMD (256 threads), zero  : 32x32x16 %.0f TFLOP/s | 16x16x32 %.0f TFLOP/s\n", run(k32, 60000, in, out, 256), run(k16, 60000, in, out, 256));
  printf("zero operands  : 32x32x16 %.0f TFLOP/s | 16x16x32 %.0f TFLOP/s\n", run(k32, 60000, in, out), run(k16, 60000, in, out));
  return 0;
}
