// Per-tile wall-clock trace of gemm_t8_kernel (needs an ORV_T8_TRACE build of the library: tools/t8_trace.sh): waves 0 and 4 of
// workgroups 0, 8 and 100 stamp [loop start, after K-tiles 0-1, after K-tiles 2-3, loop end, epilogue start, epilogue end, (drained)]
// for their first 8 tiles.  EPI 0 / 1 only (R is the trace buffer).   trace_t8 M N K epi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstring>
#include "../include/orv_mi355.h"
static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); u += 0x7fff + ((u>>16)&1); return (uint16_t)(u>>16);}
int main(int argc,char**argv){ int M=12904,N=7680,K=1920,epi=1; if(argc>4){M=atoi(argv[1]);N=atoi(argv[2]);K=atoi(argv[3]);epi=atoi(argv[4]);}
  uint16_t *A,*W,*C,*b; hipMalloc(&A,(size_t)M*K*2); hipMalloc(&W,(size_t)N*K*2); hipMalloc(&C,(size_t)M*N*2); hipMalloc(&b,N*2); hipMemset(b,0,N*2);
  std::mt19937 g(1); std::uniform_real_distribution<float> d(-1.f,1.f);
  std::vector<uint16_t> h((size_t)M*K); for(auto& x:h) x=f2bf(d(g)); hipMemcpy(A,h.data(),h.size()*2,hipMemcpyHostToDevice);
  h.resize((size_t)N*K); for(auto& x:h) x=f2bf(d(g)*0.05f); hipMemcpy(W,h.data(),h.size()*2,hipMemcpyHostToDevice);
  const int NS=3*2*8*8; unsigned long long* T; hipMalloc(&T,NS*8); hipMemset(T,0,NS*8);
  orv_gemm_t gg{}; gg.A=A; gg.lda=K; gg.W=W; gg.ldw=K; gg.bias=b; gg.C=C; gg.ldc=N; gg.M=M; gg.N=N; gg.K=K; gg.epilogue=epi; gg.R=T; gg.ldr=N;
  for(int i=0;i<3;i++) if(orv_gemm_bf16(&gg,nullptr)){ printf("%s\n",orv_last_error()); return 1; } hipDeviceSynchronize();
  std::vector<unsigned long long> t(NS); hipMemcpy(t.data(),T,NS*8,hipMemcpyDeviceToHost);
  printf("M=%d N=%d K=%d epi=%d   (us; 10-ns wall clock)\n",M,N,K,epi);
  const char* wgs[3]={"wg 0","wg 8","wg 100"};
  for(int w=0;w<3;w++) for(int half=0;half<2;half++){ printf("%s wave %d:\n",wgs[w],half*4);
    for(int i=0;i<8;i++){ unsigned long long* r=&t[((w*2+half)*8+i)*8]; if(!r[5]) break; auto us=[&](int a,int b){ return (double)(r[b]-r[a])*0.01; };
      unsigned long long* nx = i<7 ? &t[((w*2+half)*8+i+1)*8] : nullptr;
      printf("  tile %d: K-tiles 0-1 %5.2f  2-3 %5.2f  rest %6.2f | loop %6.2f  resync %5.2f  epilogue %5.2f", i, us(0,1), us(1,2), us(2,3), us(0,3), us(3,4), us(4,5));
      if(r[6]) printf("  store drain %5.2f", us(5,6));
      if(nx && nx[0]) printf("  -> next loop start +%5.2f", (double)(nx[0]-(r[6]?r[6]:r[5]))*0.01);
      printf("\n"); } }
  return 0; }
