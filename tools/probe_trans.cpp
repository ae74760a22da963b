// Probe: do transcendental VALU ops (v_exp_f32, quarter rate) overlap with ordinary VALU ops of the same wave / other waves on
// a SIMD, or do they occupy the same issue slots?  time(exp only), time(fma only), time(both interleaved).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} }while(0)
template<int NEXP,int NFMA>
__global__ __launch_bounds__(256) void k(float* out, int iters){
  float e[8], f[8];
  for(int i=0;i<8;i++){ e[i]=-0.001f*(threadIdx.x+i); f[i]=1.0f+0.001f*i; }
  for(int it=0;it<iters;it++){
    #pragma unroll
    for(int r=0;r<4;r++){
      if(NEXP){
        #pragma unroll
        for(int i=0;i<8;i++) e[i]=__builtin_amdgcn_exp2f(e[i])-1.0f;     // exp + 1 full-rate op
      }
      if(NFMA){
        #pragma unroll
        for(int j=0;j<NFMA;j++)
          #pragma unroll
          for(int i=0;i<8;i++) f[i]=fmaf(f[i],0.9999f,0.0001f);
      }
    }
  }
  float s=0; for(int i=0;i<8;i++) s+=e[i]+f[i];
  if(s==123.456f) out[threadIdx.x]=s;
}
template<int NEXP,int NFMA> float run(float* out,const char* name){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters=2000; k<NEXP,NFMA><<<256*8,256>>>(out,iters);
  hipEventRecord(e0); k<NEXP,NFMA><<<256*8,256>>>(out,iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  printf("%-28s %.3f ms\n",name,ms); return ms;
}
int main(){ float* out; CK(hipMalloc(&out,4096));
  run<1,0>(out,"8 exp(+sub) x4");
  run<0,1>(out,"8 fma x4");
  run<0,4>(out,"32 fma x4");
  run<1,1>(out,"8 exp + 8 fma x4");
  run<1,4>(out,"8 exp + 32 fma x4");
  return 0; }
