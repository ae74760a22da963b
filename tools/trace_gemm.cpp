// Per-tile s_memtime trace of the ring GEMM (needs the ORV_GEMM_TRACE build of the library, tools/abl.sh builds it):
// workgroup 0 / wave 0 stamps [tile start, main loop end, epilogue end, tile end] for its first 16 tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/orv_mi355.h"
int main(int argc,char**argv){ int M=12904,N=7680,K=1920,epi=1; if(argc>4){M=atoi(argv[1]);N=atoi(argv[2]);K=atoi(argv[3]);epi=atoi(argv[4]);}
  uint16_t *A,*W,*C,*b; hipMalloc(&A,(size_t)M*K*2); hipMalloc(&W,(size_t)N*K*2); hipMalloc(&C,(size_t)M*N*2); hipMalloc(&b,N*2); hipMemset(b,0,N*2);
  std::vector<uint16_t> h((size_t)M*K); for(size_t i=0;i<h.size();i++) h[i]=0x3c00+(rand()&0x3ff)+((rand()&1)<<15); hipMemcpy(A,h.data(),h.size()*2,hipMemcpyHostToDevice);
  h.resize((size_t)N*K); for(size_t i=0;i<h.size();i++) h[i]=0x3800+(rand()&0x3ff)+((rand()&1)<<15); hipMemcpy(W,h.data(),h.size()*2,hipMemcpyHostToDevice);
  unsigned long long* T; hipMalloc(&T,16*4*8); hipMemset(T,0,16*4*8);
  orv_gemm_t g{}; g.A=A; g.lda=K; g.W=W; g.ldw=K; g.bias=b; g.C=C; g.ldc=N; g.M=M; g.N=N; g.K=K; g.epilogue=epi; g.R=T; g.ldr=N;
  for(int i=0;i<3;i++) if(orv_gemm_bf16(&g,nullptr)){ printf("%s\n",orv_last_error()); return 1; } hipDeviceSynchronize();
  unsigned long long t[16*4]; hipMemcpy(t,T,sizeof(t),hipMemcpyDeviceToHost);
  printf("M=%d N=%d K=%d epi=%d  (ticks of s_memtime = 100 MHz constant clock -> x10 ns)\n",M,N,K,epi);
  for(int i=0;i<16 && t[i*4+3];i++){ unsigned long long* r=t+i*4; printf(" tile %2d: main loop %6llu  epilogue %5llu  tail %4llu | total %6llu   gap to next %llu\n", i, r[1]-r[0], r[2]-r[1], r[3]-r[2], r[3]-r[0], (i<15&&t[(i+1)*4])? t[(i+1)*4]-r[3]:0ULL); }
  return 0; }
