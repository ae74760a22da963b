// Probe: semantics of ds_read_b64_tr_b16 (gfx950).  LDS holds u16 values = their own element index; every lane passes an address
// and we print what each lane receives, for a few address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, int mode) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int addr_elems;
    if (mode == 0) addr_elems = (l & 15) * 4;                         // 16 lanes: consecutive 8-byte pieces of one row
    else if (mode == 1) addr_elems = (l & 15) * 64 + (l >> 4) * 4;    // lane i of a 16-group -> row i (row pitch 64 elems), col block = l>>4
    else if (mode == 2) addr_elems = (l & 3) * 64 + ((l >> 2) & 3) * 4 + (l >> 4) * 16;  // 4 rows x 4 col-pieces per 16 lanes
    else addr_elems = l * 4;
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + addr_elems));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 4; ++mode) {
        k<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l == 19 && mode != 3) { printf("  ...\n"); l = 47; } }
    }
    return 0;
}
