// Probe: practical MFMA ceiling (32x32x16 bf16) with random-ish operands, 1 or 2 waves per SIMD, with/without a
// workgroup barrier every NB MFMAs.  Calibrates the DVFS clock under matrix load for the roofline discussion.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template<int BAR>
__global__ __launch_bounds__(512) void k(const bf16x8* in, float* out, int iters){
  int t=threadIdx.x+blockIdx.x*blockDim.x; bf16x8 a[4],b[3];
  for(int i=0;i<4;i++) a[i]=in[(t*7+i)&4095]; for(int i=0;i<3;i++) b[i]=in[(t*5+i+64)&4095];
  unsigned long long t0=__builtin_amdgcn_s_memtime(), w0=wall_clock64();
  f32x16 acc[6]; for(int i=0;i<6;i++) for(int e=0;e<16;e++) acc[i][e]=0.f;
  for(int it=0;it<iters;it++){
    #pragma unroll
    for(int r=0;r<2;r++){
      #pragma unroll
      for(int i=0;i<3;i++){ acc[2*i]=__builtin_amdgcn_mfma_f32_32x32x16_bf16(b[i],a[r*2],acc[2*i],0,0,0); acc[2*i+1]=__builtin_amdgcn_mfma_f32_32x32x16_bf16(b[i],a[r*2+1],acc[2*i+1],0,0,0); }
    }
    if(BAR) __builtin_amdgcn_s_barrier();
  }
  float s=0; for(int i=0;i<6;i++) for(int e=0;e<16;e++) s+=acc[i][e]; out[t]=s;
  if(t==0){ ((unsigned long long*)out)[70000]=__builtin_amdgcn_s_memtime()-t0; ((unsigned long long*)out)[70001]=wall_clock64()-w0; }
}
template<int BAR> void run(int threads,int iters,const bf16x8* in,float* out){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<BAR><<<256,threads>>>(in,out,iters);
  hipEventRecord(e0); for(int i=0;i<3;i++) k<BAR><<<256,threads>>>(in,out,iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=3;
  double fl=256.0*(threads/64)*iters*12*32768.0; unsigned long long mt[2]; hipMemcpy(mt,(char*)out+70000*8,16,hipMemcpyDeviceToHost); printf("[memtime ticks/iter %.1f, memtime MHz %.0f, wallclock MHz %.0f] ", (double)mt[0]/iters, mt[0]/ms/1e3, mt[1]/ms/1e3); printf("threads=%d barrier=%d: %.3f ms  %.0f TFLOP/s  (=> %.2f GHz if 100%% issue)\n",threads,BAR,ms,fl/ms/1e9, fl/ms/1e9/2500.0*2.4);
}
int main(){ bf16x8* in; float* out; hipMalloc(&in,4096*16); hipMalloc(&out,256*512*4); unsigned short h[4096*8]; for(int i=0;i<4096*8;i++){ h[i]=0x3c00+(rand()&0x3ff)+((rand()&1)<<15)-((rand()&3)<<7);} hipMemcpy(in,h,sizeof(h),hipMemcpyHostToDevice);
  for(int rep=0;rep<2;rep++){ run<0>(256,200000,in,out); run<0>(512,100000,in,out); run<1>(512,100000,in,out); }
  unsigned short z[4096*8]={0}; hipMemcpy(in,z,sizeof(z),hipMemcpyHostToDevice); printf("zero operands:\n"); run<0>(512,100000,in,out);
  return 0; }
