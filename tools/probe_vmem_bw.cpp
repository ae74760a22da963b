// Probe: per-CU bandwidth of the two ways to bring an L2-resident tile into LDS on gfx950, same addresses, no compute:
//   mode 0: global_load_lds (LDS-DMA, 16 B per lane, lane-linear destination)
//   mode 1: global_load_dwordx4 into registers, then ds_write_b128 (register staging), NL loads in flight per lane
// Each workgroup (8 waves) walks its own WIN-byte window (larger than the 32 KB vector L1, resident in the 4 MB L2) ITER times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<int MODE,int NL>
__global__ __launch_bounds__(512) void k(const char* src, int win, int iters, float* sink){
  __shared__ __attribute__((aligned(16))) char smem[8*NL*1024*2];
  const int lane=threadIdx.x&63, wave=__builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  const char* base=src+(size_t)blockIdx.x*win;
  const int step=8*NL*1024;                      // bytes the workgroup moves per inner iteration
  float acc=0.f;
  for(int it=0;it<iters;it++){
    for(int off=0;off+step<=win;off+=step){
      const int buf=((off/step)&1)*8*NL*1024;
      if(MODE==0){
        #pragma unroll
        for(int j=0;j<NL;j++) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base+off+(wave*NL+j)*1024+lane*16), (__attribute__((address_space(3))) void*)(smem+buf+(wave*NL+j)*1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)":::"memory");
      } else {
        uint4 v[NL];
        #pragma unroll
        for(int j=0;j<NL;j++) v[j]=*(const uint4*)(base+off+(wave*NL+j)*1024+lane*16);
        #pragma unroll
        for(int j=0;j<NL;j++) *(uint4*)(smem+buf+(wave*NL+j)*1024+lane*16)=v[j];
      }
    }
  }
  __syncthreads();
  acc=*(float*)(smem+lane*4);
  if(acc==12345.678f) sink[0]=acc;
}
template<int MODE,int NL> void run(const char* src,int grid,int win,int iters,float* sink){
  for(int i=0;i<2;i++) k<MODE,NL><<<grid,512>>>(src,win,iters,sink);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for(int i=0;i<5;i++) k<MODE,NL><<<grid,512>>>(src,win,iters,sink); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5;
  const int step=8*NL*1024; double bytes=(double)grid*iters*(win/step)*step;
  printf("mode %d (%s) NL=%d grid=%d win=%d KB: %.3f ms  %.2f TB/s  %.1f B/clk/CU @2.1 GHz (256 CUs)\n",MODE,MODE?"global_load + ds_write":"global_load_lds",NL,grid,win/1024,ms,bytes/ms/1e9,bytes/ms/1e-3/256/2.1e9);
  CK(hipGetLastError());
}
int main(){
  const int win=(getenv("WIN_KB")?atoi(getenv("WIN_KB")):64)*1024; const int grid_max=512; char* src; float* sink; CK(hipMalloc(&src,(size_t)grid_max*win)); CK(hipMemset(src,1,(size_t)grid_max*win)); CK(hipMalloc(&sink,16));
  for(int grid: {256,512}){
    run<0,2>(src,grid,win,160,sink); run<0,4>(src,grid,win,160,sink); run<0,8>(src,grid,win,160,sink);
    run<1,2>(src,grid,win,160,sink); run<1,4>(src,grid,win,160,sink); run<1,8>(src,grid,win,160,sink);
  }
  return 0;
}
