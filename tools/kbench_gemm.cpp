// Standalone GPU check + micro-benchmark of orv_gemm_bf16 (links liborv_mi355.so; no torch).
//   ./kbench_gemm            correctness on a few shapes + timing of the CogVideoX-2B GEMM shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "../include/orv_mi355.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} }while(0)
static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); u += 0x7fff + ((u>>16)&1); return (uint16_t)(u>>16);} 
static inline float bf2f(uint16_t h){ uint32_t u=((uint32_t)h)<<16; float f; memcpy(&f,&u,4); return f;}
static float gelu(float x){ float u=0.7978845608028654f*(x+0.044715f*x*x*x); return 0.5f*x*(1.f+tanhf(u)); }

struct Dev { void* p=nullptr; size_t n=0; };
static std::vector<uint16_t> rnd_bf(size_t n, float scale, uint32_t seed){ if(getenv("KB_ZERO")) return std::vector<uint16_t>(n,0); if(getenv("KB_UNIT")) scale=1.f; std::mt19937 g(seed); std::uniform_real_distribution<float> d(-1.f,1.f); std::vector<uint16_t> v(n); for(auto& x:v) x=f2bf(d(g)*scale); return v; }
template<class T> static T* up(const std::vector<T>& h){ T* d; CK(hipMalloc(&d,h.size()*sizeof(T))); CK(hipMemcpy(d,h.data(),h.size()*sizeof(T),hipMemcpyHostToDevice)); return d; }

static int check(int M,int N,int K,int epi,int seq,int ntext,int pg){
  auto A=rnd_bf((size_t)M*K,1.f,1), W=rnd_bf((size_t)N*K,0.05f,2), bias=rnd_bf(N,0.5f,3), R=rnd_bf((size_t)M*N,1.f,4);
  int B=(M+seq-1)/seq; int G=1+(pg? (seq-ntext+pg-1)/pg:1);
  std::vector<float> gate((size_t)B*G*N); { std::mt19937 g(5); std::uniform_real_distribution<float> d(-1.f,1.f); for(auto&x:gate) x=d(g);} 
  uint16_t *dA=up(A),*dW=up(W),*db=up(bias),*dR=up(R); float* dg=up(gate); uint16_t* dC; CK(hipMalloc(&dC,(size_t)M*N*2)); CK(hipMemset(dC,0xff,(size_t)M*N*2));
  orv_gemm_t g{}; g.A=dA; g.lda=K; g.W=dW; g.ldw=K; g.bias=db; g.C=dC; g.ldc=N; g.M=M; g.N=N; g.K=K; g.epilogue=epi; g.R=dR; g.ldr=N; g.r_mod=0; g.gate=dg; g.gate_b=(long)G*N; g.gate_g=N; g.grp={seq,ntext,pg};
  int rc=orv_gemm_bf16(&g,nullptr); if(rc){ printf("rc=%d %s\n",rc,orv_last_error()); return 1; }
  CK(hipDeviceSynchronize()); std::vector<uint16_t> C((size_t)M*N); CK(hipMemcpy(C.data(),dC,C.size()*2,hipMemcpyDeviceToHost));
  // sampled reference
  std::mt19937 rg(7); double maxerr=0, maxref=0; int nsamp = (size_t)M*N<=(1<<18) ? M*N : 20000;
  for(int s=0;s<nsamp;s++){ int m,n; if(nsamp==M*N){ m=s/N; n=s%N; } else { m=rg()%M; n=rg()%N; if(s<2000) m=M-1-(s%64); }
    double acc=0; for(int k=0;k<K;k++) acc+=(double)bf2f(A[(size_t)m*K+k])*bf2f(W[(size_t)n*K+k]); acc+=bf2f(bias[n]);
    if(epi==1) acc=gelu((float)acc); if(epi==2){ int b=m/seq, sq=m%seq; int grp= sq<ntext?0:(pg?1+(sq-ntext)/pg:1); acc=bf2f(R[(size_t)m*N+n])+gate[((size_t)b*G+grp)*N+n]*acc; }
    double got=bf2f(C[(size_t)m*N+n]); maxerr=fmax(maxerr,fabs(got-acc)); maxref=fmax(maxref,fabs(acc)); }
  bool ok = maxerr <= 0.01*maxref+1e-3; printf("check M=%d N=%d K=%d epi=%d: maxerr=%.4g maxref=%.4g %s\n",M,N,K,epi,maxerr,maxref,ok?"OK":"FAIL");
  hipFree(dA);hipFree(dW);hipFree(db);hipFree(dR);hipFree(dg);hipFree(dC); return ok?0:1;
}
static void bench(int M,int N,int K,int epi,int iters){
  auto A=rnd_bf((size_t)M*K,1.f,1), W=rnd_bf((size_t)N*K,0.05f,2), bias=rnd_bf(N,0.5f,3);
  uint16_t *dA=up(A),*dW=up(W),*db=up(bias); uint16_t* dC; CK(hipMalloc(&dC,(size_t)M*N*2)); CK(hipMemset(dC,0,(size_t)M*N*2));
  int seq=3226; int B=(M+seq-1)/seq; int G=6; std::vector<float> gate((size_t)B*G*N,0.5f); float* dg=up(gate);
  orv_gemm_t g{}; g.A=dA; g.lda=K; g.W=dW; g.ldw=K; g.bias=db; g.C=dC; g.ldc=N; g.M=M; g.N=N; g.K=K; g.epilogue=epi; g.R=dC; g.ldr=N; g.gate=dg; g.gate_b=(long)G*N; g.gate_g=N; g.grp={seq,226,600};
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int i=0;i<3;i++) orv_gemm_bf16(&g,nullptr);
  CK(hipEventRecord(e0)); for(int i=0;i<iters;i++) orv_gemm_bf16(&g,nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=iters;
  printf("bench M=%5d N=%5d K=%5d epi=%d: %.3f ms  %.1f TFLOP/s\n",M,N,K,epi,ms,2.0*M*N*K/ms/1e9);
  hipFree(dA);hipFree(dW);hipFree(db);hipFree(dC);hipFree(dg);
}
// same-process interleaved A/B of forced tile candidates: ab M N K epi rounds "ring,bm,bn" ["ring,bm,bn" ...]
static void ab(int M,int N,int K,int epi,int rounds,int nt,char** tiles){
  auto A=rnd_bf((size_t)M*K,1.f,1), W=rnd_bf((size_t)N*K,0.05f,2), bias=rnd_bf(N,0.5f,3);
  uint16_t *dA=up(A),*dW=up(W),*db=up(bias); uint16_t *dC,*dR; CK(hipMalloc(&dC,(size_t)M*N*2)); CK(hipMalloc(&dR,(size_t)M*N*2)); CK(hipMemset(dC,0,(size_t)M*N*2)); CK(hipMemset(dR,0,(size_t)M*N*2));
  int seq=3226; int B=(M+seq-1)/seq; int G=6; std::vector<float> gate((size_t)B*G*N,0.5f); float* dg=up(gate);
  orv_gemm_t g{}; g.A=dA; g.lda=K; g.W=dW; g.ldw=K; g.bias=db; g.C=dC; g.ldc=N; g.M=M; g.N=N; g.K=K; g.epilogue=epi; g.R=dR; g.ldr=N; g.gate=dg; g.gate_b=(long)G*N; g.gate_g=N; g.grp={seq,226,600};
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<std::vector<double>> tf(nt);
  const int iters = 2.0*M*N*K > 5e11 ? 8 : 30;
  for(int r=0;r<rounds;r++) for(int t=0;t<nt;t++){
    int a,b,c; sscanf(tiles[t],"%d,%d,%d",&a,&b,&c); orv_gemm_force_tile(a,b,c);
    if(orv_gemm_bf16(&g,nullptr)){ printf("tile %s: %s\n",tiles[t],orv_last_error()); tf[t].push_back(0); continue; }
    for(int i=0;i<2;i++) orv_gemm_bf16(&g,nullptr);
    CK(hipEventRecord(e0)); for(int i=0;i<iters;i++) orv_gemm_bf16(&g,nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=iters;
    tf[t].push_back(2.0*M*N*K/ms/1e9);
  }
  for(int t=0;t<nt;t++){ std::sort(tf[t].begin(),tf[t].end()); printf("ab M=%5d N=%5d K=%5d epi=%d tile %-10s: median %.0f  min %.0f  max %.0f TFLOP/s  (%.4f ms)\n",M,N,K,epi,tiles[t],tf[t][tf[t].size()/2],tf[t].front(),tf[t].back(),2.0*M*N*K/tf[t][tf[t].size()/2]/1e9); }
  orv_gemm_force_tile(0,0,0);
  hipFree(dA);hipFree(dW);hipFree(db);hipFree(dC);hipFree(dR);hipFree(dg);
}

// packed-A d8 kernel against the row-major t8 kernel, same process, interleaved: abp M N K epi rounds bn [cpacked]
//   A is packed once with orv_pack_rows16; outputs compared bit for bit (same accumulation order).  cpacked = 1: d8 writes C packed (epi 0 / 1), unpacked for the check.
static int abp(int M,int N,int K,int epi,int rounds,int bn,int cpacked){
  auto A=rnd_bf((size_t)M*K,1.f,1), W=rnd_bf((size_t)N*K,0.05f,2), bias=rnd_bf(N,0.5f,3), R=rnd_bf((size_t)M*N,1.f,4);
  uint16_t *dA=up(A),*dW=up(W),*db=up(bias),*dR=up(R); uint16_t *dC0,*dC1,*dCp,*dAp; const long Mp=orv_packed_rows(M);
  CK(hipMalloc(&dC0,(size_t)M*N*2)); CK(hipMalloc(&dC1,(size_t)M*N*2)); CK(hipMalloc(&dCp,(size_t)Mp*N*2)); CK(hipMalloc(&dAp,(size_t)Mp*K*2));
  CK(hipMemset(dC0,0,(size_t)M*N*2)); CK(hipMemset(dC1,0xff,(size_t)M*N*2));
  if(orv_pack_rows16(dA,K,dAp,M,K,nullptr)){ printf("%s\n",orv_last_error()); return 1; }
  int seq=3226; int B=(M+seq-1)/seq; int G=6; std::vector<float> gate((size_t)B*G*N); { std::mt19937 g(5); std::uniform_real_distribution<float> d(-1.f,1.f); for(auto&x:gate) x=d(g);} float* dg=up(gate);
  orv_gemm_t g0{}; g0.A=dA; g0.lda=K; g0.W=dW; g0.ldw=K; g0.bias=db; g0.C=dC0; g0.ldc=N; g0.M=M; g0.N=N; g0.K=K; g0.epilogue=epi; g0.R=dR; g0.ldr=N; g0.gate=dg; g0.gate_b=(long)G*N; g0.gate_g=N; g0.grp={seq,226,600};
  orv_gemm_t g1=g0; g1.A=dAp; g1.a_packed=1; g1.C=cpacked?dCp:dC1; g1.c_packed=cpacked;
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> tf[2]; const int iters = 2.0*M*N*K > 5e11 ? 8 : 30;
  for(int r=0;r<rounds;r++) for(int t=0;t<2;t++){
    if(t) orv_gemm_force_tile(5,256,bn); else if(getenv("KB_RM_FREE")) orv_gemm_force_tile(0,0,0); else orv_gemm_force_tile(3,256,bn);   // KB_RM_FREE: the row-major side takes whatever the cost model picks
    const orv_gemm_t* g=t?&g1:&g0;
    if(orv_gemm_bf16(g,nullptr)){ printf("%s: %s\n",t?"d8":"t8",orv_last_error()); return 1; }
    for(int i=0;i<2;i++) orv_gemm_bf16(g,nullptr);
    CK(hipEventRecord(e0)); for(int i=0;i<iters;i++) orv_gemm_bf16(g,nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=iters;
    tf[t].push_back(2.0*M*N*K/ms/1e9);
  }
  orv_gemm_force_tile(0,0,0);
  if(cpacked){ if(orv_unpack_rows16(dCp,dC1,N,M,N,nullptr)){ printf("%s\n",orv_last_error()); return 1; } }
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> c0((size_t)M*N), c1((size_t)M*N); CK(hipMemcpy(c0.data(),dC0,c0.size()*2,hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(),dC1,c1.size()*2,hipMemcpyDeviceToHost));
  size_t diff=0; double maxd=0; for(size_t i=0;i<c0.size();i++) if(c0[i]!=c1[i]){ diff++; maxd=fmax(maxd,fabs(bf2f(c0[i])-bf2f(c1[i]))); }
  for(int t=0;t<2;t++){ std::sort(tf[t].begin(),tf[t].end()); printf("abp M=%5d N=%5d K=%5d epi=%d bn=%d %s: median %.0f  min %.0f  max %.0f TFLOP/s  (%.4f ms)\n",M,N,K,epi,bn,t?(cpacked?"d8 packed A, packed C":"d8 packed A          "):(getenv("KB_RM_FREE")?"row-major (model)    ":"t8 row-major         "),tf[t][tf[t].size()/2],tf[t].front(),tf[t].back(),2.0*M*N*K/tf[t][tf[t].size()/2]/1e9); }
  printf("abp M=%5d N=%5d K=%5d epi=%d bn=%d: %zu of %zu outputs differ from t8 (max |diff| %.4g) %s\n",M,N,K,epi,bn,diff,c0.size(),maxd,diff?"MISMATCH":"BIT-IDENTICAL");
  hipFree(dA);hipFree(dW);hipFree(db);hipFree(dR);hipFree(dC0);hipFree(dC1);hipFree(dCp);hipFree(dAp);hipFree(dg); return diff?1:0;
}
// warm vs cold operands: cold M N K epi iters   - the same GEMM with (a) one operand set reused, (b) the weights rotating through 32
// buffers (every layer of the model has its own: HBM- and TLB-cold at each launch), (c) every operand rotating through 8 sets
static void cold(int M,int N,int K,int epi,int iters){
  const int NW=32, NS=8;
  auto A=rnd_bf((size_t)M*K,1.f,1), W=rnd_bf((size_t)N*K,0.05f,2), bias=rnd_bf(N,0.5f,3);
  std::vector<uint16_t*> dA(NS),dC(NS),dR(NS),dW(NW);
  for(int i=0;i<NS;i++){ dA[i]=up(A); CK(hipMalloc(&dC[i],(size_t)M*N*2)); CK(hipMalloc(&dR[i],(size_t)M*N*2)); CK(hipMemset(dC[i],0,(size_t)M*N*2)); CK(hipMemset(dR[i],0,(size_t)M*N*2)); }
  for(int i=0;i<NW;i++) dW[i]=up(W);
  uint16_t* db=up(bias); int seq=3226; int B=(M+seq-1)/seq; int G=6; std::vector<float> gate((size_t)B*G*N,0.5f); float* dg=up(gate);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[6]={"one operand set","32 weight buffers","32 weights + 8 activation sets","only A rotates (8)","only R rotates (8)","only C rotates (8)"};
  for(int round=0;round<2;round++) for(int mode=0;mode<6;mode++){
    auto run=[&](int i){ orv_gemm_t g{}; int s= mode==2 ? i%NS : 0; int w= (mode==1||mode==2) ? i%NW : 0;
      g.A=dA[mode==3? i%NS : s]; g.lda=K; g.W=dW[w]; g.ldw=K; g.bias=db; g.C=dC[mode==5? i%NS : s]; g.ldc=N; g.M=M; g.N=N; g.K=K; g.epilogue=epi; g.R=dR[mode==4? i%NS : s]; g.ldr=N; g.gate=dg; g.gate_b=(long)G*N; g.gate_g=N; g.grp={seq,226,600};
      if(orv_gemm_bf16(&g,nullptr)){ printf("%s\n",orv_last_error()); exit(1);} };
    for(int i=0;i<4;i++) run(i);
    CK(hipEventRecord(e0)); for(int i=0;i<iters;i++) run(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=iters;
    if(mode==3){   // A rotates AND is rewritten on the device right before each launch (what the model does): is fresh data warm?
      float tot=0; for(int i=0;i<iters;i++){ CK(hipMemcpyAsync(dA[i%NS],dA[(i+1)%NS],(size_t)M*K*2,hipMemcpyDeviceToDevice,0)); CK(hipEventRecord(e0)); run(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t,e0,e1)); tot+=t; }
      printf("cold M=%5d N=%5d K=%5d epi=%d %-28s: %.4f ms (per-launch events; loop-timed %.4f)\n",M,N,K,epi,"A rotates, rewritten just before",tot/iters,ms); }
    printf("cold M=%5d N=%5d K=%5d epi=%d %-28s: %.4f ms  %.0f TFLOP/s\n",M,N,K,epi,names[mode],ms,2.0*M*N*K/ms/1e9);
  }
}
int main(int argc,char**argv){
  if(orv_device_check(0)){ printf("%s\n",orv_last_error()); return 2; }
  if(argc>=7 && !strcmp(argv[1],"cold")){ cold(atoi(argv[2]),atoi(argv[3]),atoi(argv[4]),atoi(argv[5]),atoi(argv[6])); return 0; }
  if(argc>=9 && !strcmp(argv[1],"check")){ return check(atoi(argv[2]),atoi(argv[3]),atoi(argv[4]),atoi(argv[5]),atoi(argv[6]),atoi(argv[7]),atoi(argv[8])); }
  if(argc>=8 && !strcmp(argv[1],"abp")){ return abp(atoi(argv[2]),atoi(argv[3]),atoi(argv[4]),atoi(argv[5]),atoi(argv[6]),atoi(argv[7]),argc>8?atoi(argv[8]):0); }
  if(argc>=8 && !strcmp(argv[1],"ab")){ ab(atoi(argv[2]),atoi(argv[3]),atoi(argv[4]),atoi(argv[5]),atoi(argv[6]),argc-7,argv+7); return 0; }
  if(argc>=7 && !strcmp(argv[1],"bench")){ bench(atoi(argv[2]),atoi(argv[3]),atoi(argv[4]),atoi(argv[5]),atoi(argv[6])); return 0; }
  int bad=0;
  bad+=check(64,128,128,0,64,8,0); bad+=check(100,192,256,1,50,8,14); bad+=check(300,64,1920,0,300,0,0);
  bad+=check(700,384,512,2,350,30,64); bad+=check(3226,1920,1920,2,3226,226,600); bad+=check(3226,7680,1920,1,3226,226,600);
  bad+=check(12904,5760,1920,0,3226,226,600); bad+=check(12904,7680,256,1,3226,226,600); bad+=check(13000,5120,128,2,3250,250,600);
  if(bad){ printf("CORRECTNESS FAILURES: %d\n",bad); return 1; }
  for(int M: {12904, 3226}){ bench(M,5760,1920,0,20); bench(M,1920,1920,2,20); bench(M,7680,1920,1,20); bench(M,1920,7680,2,20); }
  bench(8192,8192,8192,0,5);
  return 0;
}
