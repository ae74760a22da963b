// Probe: what does a GEMM epilogue's store pattern cost per CU?  One workgroup per CU (512 threads) writes its 256 x 256 bf16 tile
// (128 KB, row stride LD bytes) `iters` times in three lane maps, 16 bytes per lane and instruction:
//   scattered : lane (r = l & 15, g = l >> 4) -> row r of a 16-row block, bytes [64 P + 16 g, +16)  (the MFMA C^T accumulator layout of
//               gemm_t8.hip: the 4 lanes of a row's 64-byte segment are 16 lanes apart, an instruction touches 16 rows)
//   quad      : lane l -> row l >> 2, bytes [64 P + 16 (l & 3), +16): the same 16 rows x 64 B per instruction, adjacent lanes adjacent
//   line      : lane l -> row l >> 3, bytes [16 (l & 7), +16): 8 rows x one full 128-byte line per instruction
// and the same three maps for LOADS (residual operand of the gated epilogue).  Tiles are distinct per workgroup and rotate through
// `nbuf` buffers so the data does not sit in L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<int MODE, bool LOAD>
__global__ __launch_bounds__(512) void k(char* base, long ld, int iters, long bufstride, int nbuf, float* sink){
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  const int wr=wave>>2, wc=wave&3;          // wave tile: 128 rows x 64 columns (128 B)
  uint4 v=make_uint4(lane,wave,blockIdx.x,1); float acc=0.f;
  for(int it=0;it<iters;it++){
    char* tile = base + (long)(it%nbuf)*bufstride + (long)blockIdx.x*256*ld /*rows*/ ;
    char* wt = tile + (long)(wr*128)*ld + wc*128;
    #pragma unroll
    for(int blk=0;blk<8;blk++){             // 16-row blocks of the wave
      char* bp = wt + (long)(blk*16)*ld;
      #pragma unroll
      for(int j=0;j<2;j++){
        char* a;
        if(MODE==0) a = bp + (long)(lane&15)*ld + 64*j + 16*(lane>>4);
        else if(MODE==1) a = bp + (long)(lane>>2)*ld + 64*j + 16*(lane&3);
        else a = bp + (long)(8*j+(lane>>3))*ld + 16*(lane&7);
        if(LOAD){ uint4 r=*(const uint4*)a; acc+=__uint_as_float(r.x)+__uint_as_float(r.w); }
        else *(uint4*)a = v;
      }
    }
    v.w++;
  }
  if(LOAD && acc==12345.678f) sink[0]=acc;
}
static int GRID=256;
template<int MODE,bool LOAD> void run(char* base,long ld,int iters,long bufstride,int nbuf,float* sink,const char* name){
  for(int i=0;i<2;i++) k<MODE,LOAD><<<GRID,512>>>(base,ld,iters,bufstride,nbuf,sink);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for(int i=0;i<5;i++) k<MODE,LOAD><<<GRID,512>>>(base,ld,iters,bufstride,nbuf,sink); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5;
  double bytes=(double)GRID*iters*131072;
  printf("grid %3d %-5s %-9s ld=%6ld: %.3f ms  %.2f us per 128-KB tile and CU  %.2f TB/s  %.1f B/clk/CU @2.1 GHz\n",GRID,LOAD?"load":"store",name,ld,ms,ms*1e3/iters,bytes/ms/1e9,bytes/ms/1e-3/GRID/2.1e9);
  CK(hipGetLastError());
}
int main(){
  // 256 workgroups x 256 rows = 65536 rows; ld = 15360 B (N = 7680 bf16) -> 1 GB per buffer; 2 buffers
  const long ld=15360; const long bufstride=65536L*ld; const int nbuf=2; char* base; float* sink;
  CK(hipMalloc(&base,bufstride*nbuf)); CK(hipMemset(base,0,bufstride*nbuf)); CK(hipMalloc(&sink,16));
  for(int g: {256, 64, 16}) { GRID=g;
  for(int rep=0;rep<2;rep++){
    run<0,false>(base,ld,24,bufstride,nbuf,sink,"scattered"); run<1,false>(base,ld,24,bufstride,nbuf,sink,"quad"); run<2,false>(base,ld,24,bufstride,nbuf,sink,"line");
    run<0,true>(base,ld,24,bufstride,nbuf,sink,"scattered"); run<1,true>(base,ld,24,bufstride,nbuf,sink,"quad"); run<2,true>(base,ld,24,bufstride,nbuf,sink,"line");
  }
  // N = 1920 row stride (3840 B): out-projection / FFN2 outputs
  const long ld2=3840;
  for(int rep=0;rep<1;rep++){
    run<0,false>(base,ld2,24,65536L*ld2,nbuf,sink,"scattered"); run<2,false>(base,ld2,24,65536L*ld2,nbuf,sink,"line");
    run<0,true>(base,ld2,24,65536L*ld2,nbuf,sink,"scattered"); run<2,true>(base,ld2,24,65536L*ld2,nbuf,sink,"line");
  }
  }
  return 0;
}
