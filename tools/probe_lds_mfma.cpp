// Probe: what MFMA rate does an LDS-fed 32x32x16 bf16 main loop sustain for a given wave tile (MI x NI blocks of 32x32),
// waves per workgroup and concurrent global_load_lds refill traffic?  No epilogue, operands random, LDS layout = the GEMM's
// (64-byte rows of 32 K, chunk ^ ((row>>2)&3) swizzle).  Answers "is the ring GEMM LDS-port bound, and would 4 waves x
// (128x128) fix it" before rewriting the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

template<int MI,int NI,int WM,int WN,int NSLOT,int GLDS,int NP=0,int CM=0,int PM=0>
__global__ __launch_bounds__((WM*WN+NP)*64) void k(const char* G, float* out, int iters, size_t gspan){
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const long long c0=clock64(), w0=wall_clock64();
  constexpr int BM=MI*32*WM, BN=NI*32*WN, ROWS=BM+BN, SLOT=ROWS*64, NW=WM*WN;
  constexpr int PIECES=ROWS/16, PPW=PIECES/NW;
  const int lane=threadIdx.x&63, wave=__builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  const int wm=wave%WM, wn=wave/WM;
  if constexpr (NP>0) {
    if (wave>=NW) {   // producer wave: only DMA + barrier
      constexpr int PPP=PIECES/NP; const int pw=wave-NW;
      const char* gs=G+((size_t)blockIdx.x*ROWS*64)%gspan + (size_t)pw*PPP*1024 + lane*16;
      for(int i=threadIdx.x;i<NSLOT*SLOT/4;i+=blockDim.x) ((uint32_t*)smem)[i]=0x3f803f80u ^ ((i*2654435761u)&0x007f007fu);
      __syncthreads();
      if constexpr (PM==0) {
      for(int it=0;it<iters;it++){
        char* dst=smem+((it+NSLOT-1)%NSLOT)*SLOT+pw*PPP*1024;
        #pragma unroll
        for(int p=0;p<PPP;p++)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gs+((size_t)it*SLOT+p*1024)%gspan),
                                           (__attribute__((address_space(3))) void*)(dst+p*1024),16,0,0);
        asm volatile("s_waitcnt vmcnt(%0)"::"n"(PPP*(NSLOT-2)>63?63:PPP*(NSLOT-2)):"memory");
        __builtin_amdgcn_s_barrier();
      }
      } else {
      // register staging: 2 sub-stages in flight (two register sets), ds_write one set per iteration
      u4 ra[PPP], rb[PPP];
      #pragma unroll
      for(int p=0;p<PPP;p++){ ra[p]=*(const u4*)(gs+p*1024); rb[p]=*(const u4*)(gs+SLOT+p*1024); }
      for(int it=0;it<iters;it+=2){
        char* dst=smem+((it+NSLOT-1)%NSLOT)*SLOT+pw*PPP*1024+lane*16;
        #pragma unroll
        for(int p=0;p<PPP;p++){ if(PM==2) *(u4*)(dst+p*1024)=ra[p]; else asm volatile(""::"v"(ra[p])); }
        #pragma unroll
        for(int p=0;p<PPP;p++) ra[p]=*(const u4*)(gs+((size_t)(it+2)*SLOT+p*1024)%gspan);
        __builtin_amdgcn_s_barrier();
        dst=smem+((it+NSLOT)%NSLOT)*SLOT+pw*PPP*1024+lane*16;
        #pragma unroll
        for(int p=0;p<PPP;p++){ if(PM==2) *(u4*)(dst+p*1024)=rb[p]; else asm volatile(""::"v"(rb[p])); }
        #pragma unroll
        for(int p=0;p<PPP;p++) rb[p]=*(const u4*)(gs+((size_t)(it+3)*SLOT+p*1024)%gspan);
        __builtin_amdgcn_s_barrier();
      }
      }
      asm volatile("s_waitcnt vmcnt(0)":::"memory");
      return;
    }
  }
  // fill LDS with something non-trivial
  for(int i=threadIdx.x;i<NSLOT*SLOT/4;i+=blockDim.x) ((uint32_t*)smem)[i]=0x3f803f80u ^ ((i*2654435761u)&0x007f007fu);
  __syncthreads();
#ifdef PROBE_PRIO
  __builtin_amdgcn_s_setprio(PROBE_PRIO);
#endif
  f16v acc[MI][NI];
  for(int i=0;i<MI;i++) for(int j=0;j<NI;j++) for(int e=0;e<16;e++) acc[i][j][e]=0.f;
  const int r=lane&31, c=lane>>5;
  int aoff[MI], boff[NI];
  for(int i=0;i<MI;i++){ int row=wm*MI*32+i*32+r; aoff[i]=row*64; }
  for(int j=0;j<NI;j++){ int row=BM+wn*NI*32+j*32+r; boff[j]=row*64; }
  const int sw=((r>>2)&3);
  const char* gsrc=G+((size_t)blockIdx.x*ROWS*64)%gspan + (size_t)wave*PPW*1024 + lane*16;
  bf8 keepa[2][MI], keepb[2][NI];
  for(int it=0;it<iters;it++){
    const int slot=it%NSLOT; char* base=smem+slot*SLOT;
    if(GLDS){
      char* dst=smem+((it+NSLOT-1)%NSLOT)*SLOT+wave*PPW*1024;
      #pragma unroll
      for(int p=0;p<PPW;p++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc+((size_t)it*SLOT+p*1024)%gspan),
                                         (__attribute__((address_space(3))) void*)(dst+p*1024),16,0,0);
    }
    #pragma unroll
    for(int ks=0;ks<2;ks++){
      bf8 a[MI], b[NI];
      const int ch=((ks*2+c)^sw)*16;
      if (CM!=1 || it==0) {
      #pragma unroll
      for(int i=0;i<MI;i++) a[i]=*(const bf8*)(base+aoff[i]+ch);
      #pragma unroll
      for(int j=0;j<NI;j++) b[j]=*(const bf8*)(base+boff[j]+ch);
      if (CM==1) { for(int i=0;i<MI;i++) keepa[ks][i]=a[i]; for(int j=0;j<NI;j++) keepb[ks][j]=b[j]; }
      } else { for(int i=0;i<MI;i++) a[i]=keepa[ks][i]; for(int j=0;j<NI;j++) b[j]=keepb[ks][j]; }
      if (CM==2) {
        #pragma unroll
        for(int i=0;i<MI;i++) asm volatile(""::"v"(a[i]));
        #pragma unroll
        for(int j=0;j<NI;j++) asm volatile(""::"v"(b[j]));
      } else {
      #pragma unroll
      for(int i=0;i<MI;i++)
        #pragma unroll
        for(int j=0;j<NI;j++) acc[i][j]=__builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i],b[j],acc[i][j],0,0,0);
      }
    }
    if(GLDS){ asm volatile("s_waitcnt vmcnt(%0)"::"n"(PPW*(NSLOT-2)>63?63:PPW*(NSLOT-2)):"memory"); }
    if(GLDS==2 || GLDS==1 || NP>0) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)":::"memory");
  if(blockIdx.x==0 && threadIdx.x==0){ ((long long*)(out+1024))[0]=clock64()-c0; ((long long*)(out+1024))[1]=wall_clock64()-w0; }
  float s=0; for(int i=0;i<MI;i++) for(int j=0;j<NI;j++) for(int e=0;e<16;e++) s+=acc[i][j][e];
  if(s==123.456f) out[threadIdx.x]=s;
}
template<int MI,int NI,int WM,int WN,int NSLOT,int GLDS,int NP=0,int CM=0,int PM=0> void run(const char* G,float* out,size_t gspan,const char* name){
  constexpr int BM=MI*32*WM, BN=NI*32*WN; int smem=NSLOT*(BM+BN)*64; int iters=600;
  CK(hipFuncSetAttribute((const void*)k<MI,NI,WM,WN,NSLOT,GLDS,NP,CM,PM>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  int grid=256*4;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int i=0;i<2;i++) k<MI,NI,WM,WN,NSLOT,GLDS,NP,CM,PM><<<grid,(WM*WN+NP)*64,smem>>>(G,out,iters,gspan);
  hipEventRecord(e0); for(int i=0;i<5;i++) k<MI,NI,WM,WN,NSLOT,GLDS,NP,CM,PM><<<grid,(WM*WN+NP)*64,smem>>>(G,out,iters,gspan); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5; CK(hipGetLastError());
  long long hc[2]; CK(hipMemcpy(hc,out+1024,16,hipMemcpyDeviceToHost));
  double fl=(double)grid*iters*2.0*BM*BN*32; 
  printf("%-34s NP=%d tile %3dx%3d waves %d (wave tile %3dx%3d) slots %d smem %3dKB glds %d : %.3f ms  %.0f TF  clk %.0f MHz\n",name,NP,BM,BN,WM*WN,MI*32,NI*32,NSLOT,smem/1024,GLDS,ms,fl/ms/1e9,(double)hc[0]/((double)hc[1]/100.0));
}

typedef __attribute__((ext_vector_type(4))) float f4v;
// same loop with v_mfma_f32_16x16x32_bf16: wave tile (MI*32) x (NI*32) as 16x16 blocks, one K=32 step per sub-stage
template<int MI,int NI,int WM,int WN,int NSLOT,int GLDS,int DEPTH=NSLOT-2>
__global__ __launch_bounds__(WM*WN*64) void k16(const char* G, float* out, int iters, size_t gspan){
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const long long c0=clock64(), w0=wall_clock64();
  constexpr int BM=MI*32*WM, BN=NI*32*WN, ROWS=BM+BN, SLOT=ROWS*64, NW=WM*WN;
  constexpr int PIECES=ROWS/16, PPW=PIECES/NW;
  constexpr int MB=MI*2, NB=NI*2;   // 16-row blocks
  const int lane=threadIdx.x&63, wave=__builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  const int wm=wave%WM, wn=wave/WM;
  for(int i=threadIdx.x;i<NSLOT*SLOT/4;i+=blockDim.x) ((uint32_t*)smem)[i]=0x3f803f80u ^ ((i*2654435761u)&0x007f007fu);
  __syncthreads();
  f4v acc[MB][NB];
  for(int i=0;i<MB;i++) for(int j=0;j<NB;j++) for(int e=0;e<4;e++) acc[i][j][e]=0.f;
  const int r=lane&15, c=lane>>4;
  int aoff[MB], boff[NB];
  for(int i=0;i<MB;i++){ int row=wm*MI*32+i*16+r; aoff[i]=row*64+((c^((row>>2)&3))*16); }
  for(int j=0;j<NB;j++){ int row=BM+wn*NI*32+j*16+r; boff[j]=row*64+((c^((row>>2)&3))*16); }
  const char* gsrc=G+((size_t)blockIdx.x*ROWS*64)%gspan + (size_t)wave*PPW*1024 + lane*16;
  for(int it=0;it<iters;it++){
    char* base=smem+(it%NSLOT)*SLOT;
    if(GLDS){
      char* dst=smem+((it+NSLOT-1)%NSLOT)*SLOT+wave*PPW*1024;
      #pragma unroll
      for(int p=0;p<PPW;p++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc+((size_t)it*SLOT+p*1024)%gspan),
                                         (__attribute__((address_space(3))) void*)(dst+p*1024),16,0,0);
    }
    bf8 a[MB], b[NB];
    #pragma unroll
    for(int i=0;i<MB;i++) a[i]=*(const bf8*)(base+aoff[i]);
    #pragma unroll
    for(int j=0;j<NB;j++) b[j]=*(const bf8*)(base+boff[j]);
    #pragma unroll
    for(int i=0;i<MB;i++)
      #pragma unroll
      for(int j=0;j<NB;j++) acc[i][j]=__builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i],b[j],acc[i][j],0,0,0);
    if(GLDS){ asm volatile("s_waitcnt vmcnt(%0)"::"n"(PPW*DEPTH>63?63:PPW*DEPTH):"memory"); __builtin_amdgcn_s_barrier(); }
  }
  asm volatile("s_waitcnt vmcnt(0)":::"memory");
  if(blockIdx.x==0 && threadIdx.x==0){ ((long long*)(out+1024))[0]=clock64()-c0; ((long long*)(out+1024))[1]=wall_clock64()-w0; }
  float s=0; for(int i=0;i<MB;i++) for(int j=0;j<NB;j++) for(int e=0;e<4;e++) s+=acc[i][j][e];
  if(s==123.456f) out[threadIdx.x]=s;
}
template<int MI,int NI,int WM,int WN,int NSLOT,int GLDS,int DEPTH=NSLOT-2> void run16(const char* G,float* out,size_t gspan,const char* name){
  constexpr int BM=MI*32*WM, BN=NI*32*WN; int smem=NSLOT*(BM+BN)*64; int iters=600;
  CK(hipFuncSetAttribute((const void*)k16<MI,NI,WM,WN,NSLOT,GLDS,DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  int grid=256*4;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int i=0;i<2;i++) k16<MI,NI,WM,WN,NSLOT,GLDS,DEPTH><<<grid,WM*WN*64,smem>>>(G,out,iters,gspan);
  hipEventRecord(e0); for(int i=0;i<5;i++) k16<MI,NI,WM,WN,NSLOT,GLDS,DEPTH><<<grid,WM*WN*64,smem>>>(G,out,iters,gspan); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5; CK(hipGetLastError());
  long long hc[2]; CK(hipMemcpy(hc,out+1024,16,hipMemcpyDeviceToHost));
  double fl=(double)grid*iters*2.0*BM*BN*32;
  printf("%-34s [16x16x32] depth %d tile %3dx%3d waves %d glds %d : %.3f ms  %.0f TF  clk %.0f MHz\n",name,DEPTH,BM,BN,WM*WN,GLDS,ms,fl/ms/1e9,(double)hc[0]/((double)hc[1]/100.0));
}

int main(){
  size_t gspan=(size_t)256<<20; char* G; CK(hipMalloc(&G,gspan+(4<<20))); 
  { std::vector<uint16_t> h((gspan+(4<<20))/2); for(size_t i=0;i<h.size();i++) h[i]=0x3f80 ^ (uint16_t)((i*2654435761u)>>25); CK(hipMemcpy(G,h.data(),h.size()*2,hipMemcpyHostToDevice)); }
  float* out; CK(hipMalloc(&out,8192));
  for(size_t span : {(size_t)2<<20, (size_t)64<<20}){
    printf("--- span %zu MB\n", span>>20);
    run16<2,3,4,2,5,1,0>(G,out,span,"256x192 depth0");
    run16<2,3,4,2,5,1,1>(G,out,span,"256x192 depth1");
    run16<2,3,4,2,5,1,2>(G,out,span,"256x192 depth2");
    run16<2,3,4,2,5,1,3>(G,out,span,"256x192 depth3");
    run16<2,2,4,2,5,1,3>(G,out,span,"256x128 depth3");
    run16<2,2,4,2,6,1,4>(G,out,span,"256x128 6 slots depth4");
    run16<2,2,4,2,6,1,2>(G,out,span,"256x128 6 slots depth2");
  }
  return 0;
}
