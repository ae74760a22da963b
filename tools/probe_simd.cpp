// Probe: which SIMD does wave i of a 512-thread workgroup run on? (HW_REG_HW_ID: simd_id bits [5:4] on gfx9-family)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out){
  extern __shared__ char smem[];
  unsigned hw = __builtin_amdgcn_s_getreg((31<<11)|(0<<6)|4);
  if((threadIdx.x&63)==0) out[blockIdx.x*8 + (threadIdx.x>>6)] = hw;
}
int main(){ unsigned* d; hipMalloc(&d, 64*8*4); hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140*1024); k<<<64,512,140*1024>>>(d); unsigned h[512]; hipMemcpy(h,d,sizeof(h),hipMemcpyDeviceToHost);
  for(int b=0;b<12;b++){ printf("block %2d: ",b); for(int w=0;w<8;w++){ unsigned v=h[b*8+w]; printf("w%d[simd=%u wave=%u cu=%u] ", w, (v>>4)&3, v&15, (v>>8)&15);} printf("\n"); }
  return 0; }
