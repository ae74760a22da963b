// Standalone micro-benchmark of orv_qkv_prep + orv_attention_fwd at the CogVideoX-2B shape (no torch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "../include/orv_mi355.h"
int main(int argc,char**argv){ int B=argc>1?atoi(argv[1]):4, S=argc>2?atoi(argv[2]):3226, H=argc>3?atoi(argv[3]):30; int s_pad=(S+63)/64*64; size_t n=(size_t)B*S*3*H*64;
  std::vector<uint16_t> h(n); for(size_t i=0;i<n;i++){ float f=((rand()&0xffff)/32768.f-1.f); uint32_t u; memcpy(&u,&f,4); h[i]=u>>16; }
  uint16_t *qkv,*vT,*out; hipMalloc(&qkv,n*2); hipMalloc(&vT,(size_t)B*H*64*s_pad*2); hipMalloc(&out,(size_t)B*S*H*64*2); hipMemcpy(qkv,h.data(),n*2,hipMemcpyHostToDevice); hipMemset(vT,0,(size_t)B*H*64*s_pad*2);
  float premul = getenv("FUSED") ? 0.125f*1.4426950408889634f : 1.0f; float sc = getenv("FUSED") ? 1.0f/1.4426950408889634f : 0.125f;
  orv_qkv_prep(qkv,vT,nullptr,nullptr,nullptr,nullptr,nullptr,nullptr,B,S,H,226,s_pad,1e-6f,premul,nullptr);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint16_t* vt_arg = getenv("ATTN_V1") ? vT : nullptr;   // default: V in place (attn_fwd_v2)
  if(getenv("BOUND")){ float bd=atof(getenv("BOUND"));   // bounded scores: orv_attention_fwd_bounded (fixed-shift / ping-pong kernels)
    for(int i=0;i<3;i++) orv_attention_fwd_bounded(qkv,3*H*64,out,H*64,nullptr,B,S,H,sc,bd,nullptr);
    int it=getenv("ITERS")?atoi(getenv("ITERS")):20;
    hipEventRecord(e0); for(int i=0;i<it;i++) orv_attention_fwd_bounded(qkv,3*H*64,out,H*64,nullptr,B,S,H,sc,bd,nullptr); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=it;
    printf("attention(bounded %.1f) B=%d S=%d H=%d: %.4f ms  %.1f TFLOP/s\n",bd,B,S,H,ms,4.0*B*H*(double)S*S*64/ms/1e9);
    if(getenv("TRACE")){   // library built with -DORV_PP_TRACE: per-workgroup (start, end, HW_ID | XCC_ID << 32, item) through the lse pointer
      int nwg=((S+255)/256)*H*B; const int words = getenv("TRACE_WORDS") ? atoi(getenv("TRACE_WORDS")) : 4;   // u64 per workgroup
      unsigned long long* tr; hipMalloc(&tr,(size_t)B*H*S*4+(size_t)nwg*words*8); hipMemset(tr,0,(size_t)nwg*words*8);
      orv_attention_fwd_bounded(qkv,3*H*64,out,H*64,(float*)tr,B,S,H,sc,bd,nullptr); hipDeviceSynchronize();
      std::vector<unsigned long long> t((size_t)nwg*words); hipMemcpy(t.data(),tr,(size_t)nwg*words*8,hipMemcpyDeviceToHost);
      FILE* f=fopen(getenv("TRACE"),"w"); for(int i=0;i<nwg;i++){ fprintf(f,"%d",i); for(int w=0;w<words;w++) fprintf(f," %llu",t[(size_t)i*words+w]); fprintf(f,"\n"); } fclose(f); }
    return 0; }
  for(int i=0;i<3;i++) orv_attention_fwd(qkv,3*H*64,vt_arg,out,H*64,nullptr,B,S,H,s_pad,sc,nullptr);
  hipEventRecord(e0); for(int i=0;i<20;i++) orv_attention_fwd(qkv,3*H*64,vt_arg,out,H*64,nullptr,B,S,H,s_pad,sc,nullptr); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=20;
  printf("attention B=%d S=%d H=%d: %.4f ms  %.1f TFLOP/s\n",B,S,H,ms,4.0*B*H*(double)S*S*64/ms/1e9);
  return 0; }
