// LDS read rate of one CU: ds_read_b128 with every lane at its own address, with two distinct addresses per wave (the half-wave
// broadcast of the dK / dV pass's accumulator init) and ds_read_b64_tr_b16; 4 and 8 waves per CU.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, int Q>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)smem)[i] = (float)i;
    __syncthreads();
    unsigned off;
    if (MODE == 0) off = (wave & 7) * 4096 + lane * 16;                 // distinct, conflict-free
    else if (MODE == 1) off = (wave & 7) * 4096 + (lane >> 5) * 16;     // two addresses per wave
    else off = (wave & 7) * 4096 + lane * 8;                            // b64 (transposing) reads, contiguous
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + off;
    f4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 2) {
                float2 v;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(u * 512));
                asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(Q) : "memory");
                acc[0] += v.x;
            } else {
                f4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(u * 16 * (MODE == 0 ? 0 : 1)));
                asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(Q) : "memory");
                acc[0] += v[0];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc[0] == 12345.678f) out[0] = 0;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16 * 8);
    const int iters = 2000;
    const char* names[3] = {"ds_read_b128 distinct", "ds_read_b128 2 addresses/wave", "ds_read_b64_tr_b16"};
    for (int q = 15; q >= 3; q = (q - 1) / 2)
    for (int waves = 4; waves <= 16; waves *= 2)
        for (int m = 0; m < 3; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0 && q == 15) hipLaunchKernelGGL((k<0, 15>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                if (m == 1 && q == 15) hipLaunchKernelGGL((k<1, 15>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                if (m == 2 && q == 15) hipLaunchKernelGGL((k<2, 15>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                if (m == 0 && q == 7) hipLaunchKernelGGL((k<0, 7>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                if (m == 1 && q == 7) hipLaunchKernelGGL((k<1, 7>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                if (m == 2 && q == 7) hipLaunchKernelGGL((k<2, 7>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                if (m == 0 && q == 3) hipLaunchKernelGGL((k<0, 3>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                if (m == 1 && q == 3) hipLaunchKernelGGL((k<1, 3>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                if (m == 2 && q == 3) hipLaunchKernelGGL((k<2, 3>), dim3(1), dim3(waves * 64), 0, 0, d, iters);
                hipDeviceSynchronize();
            }
            unsigned long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
            const double per = mx / (double)(iters * 16);      // clock64 ticks (100 MHz? shader clock?) per read per wave
            printf("outstanding <= %2d  %-32s %2d waves: %.2f ticks per instruction and wave, %.2f per instruction over the CU\n", q + 1, names[m], waves, per, per / waves);
        }
    return 0;
}
