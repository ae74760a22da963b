// Probe: issue rate of v_exp_f32 vs v_exp_f16 (and v_exp_legacy) on gfx950: cycles per wave instruction, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template<int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters){
  float e[8]; _Float16 h[8];
  for(int i=0;i<8;i++){ e[i]=-0.001f*(threadIdx.x+i); h[i]=(_Float16)e[i]; }
  for(int it=0;it<iters;it++){
    #pragma unroll
    for(int r=0;r<8;r++){
      #pragma unroll
      for(int i=0;i<8;i++){
        if(MODE==0) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
        else if(MODE==1) asm volatile("v_exp_f16 %0, %0" : "+v"(h[i]));
        else asm volatile("v_mul_f32 %0, %0, %0" : "+v"(e[i]));
      }
    }
  }
  float s=0; for(int i=0;i<8;i++) s+=e[i]+(float)h[i];
  if(s==123.456f) out[threadIdx.x]=s;
}
template<int MODE> void run(float* out,const char* name){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters=4000; k<MODE><<<256*4,256>>>(out,iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<256*4,256>>>(out,iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
  // per SIMD: 4 WGs per CU x 4 waves / 4 SIMDs = 4 waves per SIMD; instructions per wave = iters*64
  double instr_per_simd = 4.0 * iters * 64; printf("%-10s %.3f ms  -> %.1f cycles per wave instruction at 2.4 GHz (4 waves / SIMD)\n", name, ms, ms*1e-3*2.4e9/instr_per_simd);
}
int main(){ float* out; hipMalloc(&out, 4096); run<2>(out,"v_mul_f32"); run<0>(out,"v_exp_f32"); run<1>(out,"v_exp_f16"); return 0; }
