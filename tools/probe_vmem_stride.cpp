// Probe: global_load_lds bandwidth per CU for the GEMM operand pattern - one wave instruction = 8 rows x 128 B, rows ROWSTRIDE bytes
// apart (a K-contiguous operand), the workgroup walking along K - against the contiguous pattern (8 consecutive lines).
// Every workgroup owns ROWS rows; ROWS x KB_PER_ROW is sized to stay L2-resident (32 workgroups per XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<int NL>
__global__ __launch_bounds__(512) void k(const char* src, long rowstride, int kbytes, int iters, int shared_rows, float* sink){
  __shared__ __attribute__((aligned(16))) char smem[8*NL*1024*2];
  const int lane=threadIdx.x&63, wave=__builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  constexpr int ROWS=8*NL*8;
  // shared_rows: all workgroups of an XCD read the SAME rows (panel sharing as in a GEMM), else each its own
  const long row0 = shared_rows ? (long)(blockIdx.x&7)*ROWS : (long)blockIdx.x*ROWS;
  const char* p[NL];
  #pragma unroll
  for(int j=0;j<NL;j++) p[j]=src+(row0+(wave*NL+j)*8+(lane>>3))*rowstride+(lane&7)*16;
  for(int it=0;it<iters;it++)
    for(int ko=0;ko<kbytes;ko+=128){
      const int buf=((ko>>7)&1)*8*NL*1024;
      #pragma unroll
      for(int j=0;j<NL;j++) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p[j]+ko), (__attribute__((address_space(3))) void*)(smem+buf+(wave*NL+j)*1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)":::"memory");
    }
  __syncthreads();
  float acc=*(float*)(smem+lane*4);
  if(acc==12345.678f) sink[0]=acc;
}
template<int NL> void run(const char* src,long rowstride,int kbytes,int iters,int shared,float* sink){
  const int grid=256;
  for(int i=0;i<2;i++) k<NL><<<grid,512>>>(src,rowstride,kbytes,iters,shared,sink);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for(int i=0;i<5;i++) k<NL><<<grid,512>>>(src,rowstride,kbytes,iters,shared,sink); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5;
  double bytes=(double)grid*iters*(kbytes/128)*8*NL*1024;
  printf("rows/wg=%d rowstride=%ld B k=%d B %s: %.3f ms  %.2f TB/s  %.1f B/clk/CU @2.1 GHz\n",8*NL*8,rowstride,kbytes,shared?"shared panel per XCD":"own rows",ms,bytes/ms/1e9,bytes/ms/1e-3/256/2.1e9);
  CK(hipGetLastError());
}
int main(){
  char* src; float* sink; size_t sz=(size_t)3<<30; CK(hipMalloc(&src,sz)); CK(hipMemset(src,1,sz)); CK(hipMalloc(&sink,16));
  // NL=5: 320 rows per workgroup (the 192x128 tile).  k = 512 B per row keeps 320 x 512 = 160 KB per workgroup: 5 MB per XCD (over L2), 256 B: 2.5 MB
  for(long rs: {256L, 3840L, 15360L, 15360L+128, 4096L, 16384L}){
    run<5>(src,rs,256,400,0,sink);
    run<5>(src,rs,256,400,1,sink);
  }
  return 0;
}
