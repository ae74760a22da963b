// Per-phase s_memtime trace of the ping-pong GEMM (needs the ORV_GEMM_TRACE build of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/orv_mi355.h"
int main(int argc,char**argv){ int M=12904,N=1920,K=7680; if(argc>3){M=atoi(argv[1]);N=atoi(argv[2]);K=atoi(argv[3]);}
  uint16_t *A,*W,*C; hipMalloc(&A,(size_t)M*K*2); hipMalloc(&W,(size_t)N*K*2); hipMalloc(&C,(size_t)M*N*2);
  std::vector<uint16_t> h((size_t)M*K); for(size_t i=0;i<h.size();i++) h[i]=0x3c00+(rand()&0x3ff)+((rand()&1)<<15); hipMemcpy(A,h.data(),h.size()*2,hipMemcpyHostToDevice);
  h.resize((size_t)N*K); for(size_t i=0;i<h.size();i++) h[i]=0x3c00+(rand()&0x3ff)+((rand()&1)<<15); hipMemcpy(W,h.data(),h.size()*2,hipMemcpyHostToDevice);
  unsigned long long* T; hipMalloc(&T,2*64*4*8); hipMemset(T,0,2*64*4*8);
  orv_gemm_t g{}; g.A=A; g.lda=K; g.W=W; g.ldw=K; g.C=C; g.ldc=N; g.M=M; g.N=N; g.K=K; g.epilogue=0; g.R=T;
  for(int i=0;i<3;i++) orv_gemm_bf16(&g,nullptr); hipDeviceSynchronize();
  unsigned long long t[2*64*4]; hipMemcpy(t,T,sizeof(t),hipMemcpyDeviceToHost);
  printf("group0 (compute phase first): per iteration j: [lgk+mfma issue] [wait_dma] [barrier] [reads+barrier -> next iter]\n");
  for(int j=0;j<24;j++){ unsigned long long* r=t+(0*64+j)*4; unsigned long long* nx=t+(0*64+j+1)*4; printf(" j=%2d  mfma %5llu  dma %4llu  bar %4llu  load-phase %5llu   | total %5llu\n", j, r[1]-r[0], r[2]-r[1], r[3]-r[2], nx[0]-r[3], nx[0]-r[0]); }
  printf("group1 (load phase first): [reads+wait_dma] [barrier] [lgk+mfma issue] [barrier->next]\n");
  for(int j=0;j<24;j++){ unsigned long long* r=t+(1*64+j)*4; unsigned long long* nx=t+(1*64+j+1)*4; printf(" j=%2d  reads %5llu  bar %4llu  mfma %5llu  bar %5llu   | total %5llu\n", j, r[1]-r[0], r[2]-r[1], r[3]-r[2], nx[0]-r[3], nx[0]-r[0]); }
  return 0; }
