// GPU probe: verifies gfx950 MFMA fragment layouts, global_load_lds semantics and permlane32_swap.
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_layouts probe_layouts.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); u += 0x7fff + ((u>>16)&1); return (uint16_t)(u>>16);} 
static inline float bf2f(uint16_t h){ uint32_t u=((uint32_t)h)<<16; float f; memcpy(&f,&u,4); return f;}

// hypothesis: A lane l holds A[i=l%32][k=8*(l/32)+e], B lane l holds B[k=8*(l/32)+e][j=l%32]
__global__ void k32(const uint16_t* A, const uint16_t* B, float* C){ // A[32][16] row-major, B[16][32] row-major
  int l=threadIdx.x; bf16x8 a,b;
  for(int e=0;e<8;e++){ a[e]=A[(l%32)*16 + 8*(l/32)+e]; b[e]=B[(8*(l/32)+e)*32 + (l%32)]; }
  f32x16 c={0}; c=__builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,c,0,0,0);
  for(int r=0;r<16;r++){ int row=(r&3)+8*(r>>2)+4*(l>>5); int col=l&31; C[row*32+col]=c[r]; }
}
__global__ void k16(const uint16_t* A, const uint16_t* B, float* C){ // A[16][32], B[32][16]
  int l=threadIdx.x; bf16x8 a,b;
  for(int e=0;e<8;e++){ a[e]=A[(l%16)*32 + 8*(l/16)+e]; b[e]=B[(8*(l/16)+e)*16 + (l%16)]; }
  f32x4 c={0}; c=__builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0);
  for(int r=0;r<4;r++){ int row=(l>>4)*4+r; int col=l&15; C[row*16+col]=c[r]; }
}
// glds: each lane loads 16B from its own global ptr; LDS dest = uniform base + lane*16
__global__ void kglds(const uint32_t* src, uint32_t* out){
  __shared__ __attribute__((aligned(16))) uint32_t lds[64*4*2];
  int l=threadIdx.x;
  // lane l loads 16B chunk index perm(l) = l^5 from src
  const uint32_t* g = src + ((l^5)*4);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(lds+256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for(int i=0;i<4;i++) out[l*4+i]=lds[256+l*4+i];
}
__global__ void kperm(uint32_t* out){
  int l=threadIdx.x; unsigned a=1000+l, b=2000+l;
  auto r=__builtin_amdgcn_permlane32_swap(a,b,false,false);
  out[l*2]=r[0]; out[l*2+1]=r[1];
}
// simple HBM copy bandwidth
__global__ void kcopy(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n){
  size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=st) d[i]=s[i];
}
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0)); printf("device %s arch %s CUs %d clock %d kHz mem %zu GB lds/block %zu\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem>>30, p.sharedMemPerBlock);
  { // 32x32x16
    std::vector<uint16_t> A(32*16),B(16*32); std::vector<float> Af(32*16),Bf(16*32);
    for(int i=0;i<32*16;i++){ float v=(float)((i*7+3)%13-6)/4.f; A[i]=f2bf(v); Af[i]=bf2f(A[i]); }
    for(int i=0;i<16*32;i++){ float v=(float)((i*5+1)%17-8)/8.f; B[i]=f2bf(v); Bf[i]=bf2f(B[i]); }
    uint16_t *dA,*dB; float* dC; CK(hipMalloc(&dA,A.size()*2)); CK(hipMalloc(&dB,B.size()*2)); CK(hipMalloc(&dC,32*32*4));
    CK(hipMemcpy(dA,A.data(),A.size()*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,B.data(),B.size()*2,hipMemcpyHostToDevice));
    k32<<<1,64>>>(dA,dB,dC); std::vector<float> C(32*32); CK(hipMemcpy(C.data(),dC,32*32*4,hipMemcpyDeviceToHost));
    double maxerr=0; for(int i=0;i<32;i++)for(int j=0;j<32;j++){ double r=0; for(int k=0;k<16;k++) r+=Af[i*16+k]*Bf[k*32+j]; maxerr=fmax(maxerr,fabs(r-C[i*32+j])); }
    printf("mfma_32x32x16_bf16 layout check maxerr=%g %s\n",maxerr,maxerr<1e-3?"OK":"MISMATCH");
  }
  { // 16x16x32
    std::vector<uint16_t> A(16*32),B(32*16); std::vector<float> Af(16*32),Bf(32*16);
    for(int i=0;i<16*32;i++){ float v=(float)((i*7+3)%13-6)/4.f; A[i]=f2bf(v); Af[i]=bf2f(A[i]); }
    for(int i=0;i<32*16;i++){ float v=(float)((i*5+1)%17-8)/8.f; B[i]=f2bf(v); Bf[i]=bf2f(B[i]); }
    uint16_t *dA,*dB; float* dC; CK(hipMalloc(&dA,A.size()*2)); CK(hipMalloc(&dB,B.size()*2)); CK(hipMalloc(&dC,16*16*4));
    CK(hipMemcpy(dA,A.data(),A.size()*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,B.data(),B.size()*2,hipMemcpyHostToDevice));
    k16<<<1,64>>>(dA,dB,dC); std::vector<float> C(16*16); CK(hipMemcpy(C.data(),dC,16*16*4,hipMemcpyDeviceToHost));
    double maxerr=0; for(int i=0;i<16;i++)for(int j=0;j<16;j++){ double r=0; for(int k=0;k<32;k++) r+=Af[i*32+k]*Bf[k*16+j]; maxerr=fmax(maxerr,fabs(r-C[i*16+j])); }
    printf("mfma_16x16x32_bf16 layout check maxerr=%g %s\n",maxerr,maxerr<1e-3?"OK":"MISMATCH");
  }
  { std::vector<uint32_t> s(256); for(int i=0;i<256;i++) s[i]=i; uint32_t *ds,*dd; CK(hipMalloc(&ds,1024)); CK(hipMalloc(&dd,1024)); CK(hipMemcpy(ds,s.data(),1024,hipMemcpyHostToDevice));
    kglds<<<1,64>>>(ds,dd); std::vector<uint32_t> o(256); CK(hipMemcpy(o.data(),dd,1024,hipMemcpyDeviceToHost)); int bad=0; for(int l=0;l<64;l++)for(int i=0;i<4;i++) if(o[l*4+i]!=(uint32_t)((l^5)*4+i)) bad++;
    printf("global_load_lds(16B) lane-linear dest, per-lane src: %s (bad=%d) sample o[0..7]=%u %u %u %u %u %u %u %u\n", bad?"MISMATCH":"OK", bad,o[0],o[1],o[2],o[3],o[4],o[5],o[6],o[7]); }
  { uint32_t* dd; CK(hipMalloc(&dd,64*8)); kperm<<<1,64>>>(dd); std::vector<uint32_t> o(128); CK(hipMemcpy(o.data(),dd,512,hipMemcpyDeviceToHost));
    printf("permlane32_swap(a=1000+l,b=2000+l): lane0 r=(%u,%u) lane5 r=(%u,%u) lane32 r=(%u,%u) lane37 r=(%u,%u)\n",o[0],o[1],o[10],o[11],o[64],o[65],o[74],o[75]); }
  { size_t n=(size_t)1<<30; uint4 *s,*d; CK(hipMalloc(&s,n)); CK(hipMalloc(&d,n)); CK(hipMemset(s,1,n)); hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for(int it=0;it<2;it++){ hipEventRecord(e0); for(int r=0;r<5;r++) kcopy<<<2048,256>>>(s,d,n/16); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); printf("copy 1GiB x5: %.3f ms -> %.2f TB/s (r+w)\n", ms, 5*2.0*n/ms/1e9); } }
  return 0;
}
