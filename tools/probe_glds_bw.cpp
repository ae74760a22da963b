// Probe: sustainable L2->LDS bandwidth of global_load_lds in a GEMM-like panel access pattern with deep prefetch.
// Each workgroup (8 waves) plays a GEMM tile (tm, tn): streams A rows [tm*BM, +BM) and W rows [tn*BN, +BN) over K,
// 16 B per lane, DEPTH K-slices (BK=32) in flight per wave (counted vmcnt), no compute.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<int BM,int BN,int DEPTH>
__global__ __launch_bounds__(512) void k(const char* A, const char* W, int K2 /*row bytes*/, int tiles_n, int nk, int xcd_group){
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane=threadIdx.x&63, wave=__builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  int b=blockIdx.x, nb=gridDim.x; int tm,tn;
  if(xcd_group){ int q=nb>>3, xcd=b&7, j=b>>3; int L=xcd*q+j; int per=4*tiles_n; int gid=L/per, rem=L%per; tm=gid*4+rem%4; tn=rem/4; }
  else { tm=b/tiles_n; tn=b%tiles_n; }
  constexpr int ROWS=(BM+BN); constexpr int PIECES=ROWS/16; constexpr int PPW=PIECES/8; // 16 rows x 64B per piece (BK=32)
  const char* src[PPW];
  for(int j=0;j<PPW;j++){ int row=(wave*PPW+j)*16+(lane>>2); int chunk=lane&3; const char* base = row<BM ? A+(size_t)(tm*BM+row)*K2 : W+(size_t)(tn*BN+row-BM)*K2; src[j]=base+chunk*16; }
  constexpr int SLOT=ROWS*64;
  int issued=0;
  for(int t=0;t<nk+DEPTH;t++){
    if(t<nk){
      #pragma unroll
      for(int j=0;j<PPW;j++){ __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j]+(size_t)t*64), (__attribute__((address_space(3))) void*)(smem+(t%(DEPTH+1))*SLOT+(wave*PPW+j)*1024),16,0,0); }
    }
    if(t>=DEPTH){
      if(t<nk){ if constexpr(PPW*DEPTH<=63) asm volatile("s_waitcnt vmcnt(%0)"::"n"(PPW*DEPTH):"memory"); else asm volatile("s_waitcnt vmcnt(0)":::"memory"); }
      else asm volatile("s_waitcnt vmcnt(0)":::"memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}
template<int BM,int BN,int DEPTH> void run(const char* A,const char* W,int M,int N,int K,int xg){
  int tiles_m=M/BM, tiles_n=N/BN; int nk=K/32; int grid=tiles_m*tiles_n; int smem=(DEPTH+1)*(BM+BN)*64;
  CK(hipFuncSetAttribute((const void*)k<BM,BN,DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int i=0;i<2;i++) k<BM,BN,DEPTH><<<grid,512,smem>>>(A,W,K*2,tiles_n,nk,xg);
  hipEventRecord(e0); for(int i=0;i<5;i++) k<BM,BN,DEPTH><<<grid,512,smem>>>(A,W,K*2,tiles_n,nk,xg); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5;
  double bytes=(double)grid*nk*(BM+BN)*64; printf("BM=%d BN=%d depth=%d xcdgroup=%d M=%d N=%d K=%d: %.3f ms  %.2f TB/s  (%.1f B/clk/CU @2.1GHz)  equiv GEMM TF at this load rate: %.0f\n",BM,BN,DEPTH,xg,M,N,K,ms,bytes/ms/1e9,bytes/ms/1e9*1e12/1e3/256/2.1e9*1e-6*1e6/1e3, 2.0*M*N*K/ms/1e9);
  CK(hipGetLastError());
}
int main(){
  int M=12288,N=7680,K=1920; char *A,*W; CK(hipMalloc(&A,(size_t)M*K*2)); CK(hipMalloc(&W,(size_t)N*K*2)); CK(hipMemset(A,1,(size_t)M*K*2)); CK(hipMemset(W,1,(size_t)N*K*2));
  for(int xg=0;xg<2;xg++){
    run<256,192,1>(A,W,M,N,K,xg); run<256,192,2>(A,W,M,N,K,xg); run<256,192,4>(A,W,M,N,K,xg);
    run<256,256,3>(A,W,M,N,K,xg); run<256,384,3>(A,W,M,N,K,xg); run<256,128,4>(A,W,M,N,K,xg); run<128,128,6>(A,W,M,N,K,xg);
  }
  return 0;
}
