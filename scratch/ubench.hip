// Micro-benchmarks: what can gfx950 do for random 256-B row traffic?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

__device__ __forceinline__ uint32_t hash32(uint32_t x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

__global__ void copy_k(const f4* __restrict__ a, f4* __restrict__ b, int64_t n){
  int64_t s=(int64_t)gridDim.x*blockDim.x; for(int64_t i=(int64_t)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=s) b[i]=a[i]; }

// random row ops: rows of 64 floats, 16 lanes per row, UN rows in flight per lane-group
// mode 0: read only (sum), 1: RMW in place, 2: write only
template<int UN,int MODE>
__global__ __launch_bounds__(256) void rand_rows(float* __restrict__ W, const int* __restrict__ ids, int64_t nrows_tab, int64_t nrefs, float* out, int use_ids){
  const int lane=threadIdx.x&63, sub=lane&15, grp=lane>>4;
  const int64_t gw=(int64_t)blockIdx.x*4+(threadIdx.x>>6);
  const int64_t nw=(int64_t)gridDim.x*4;
  f4 acc={0,0,0,0};
  for(int64_t base=gw*4*UN; base<nrefs; base+=nw*4*UN){
    f4 r[UN]; int64_t row[UN];
    #pragma unroll
    for(int j=0;j<UN;j++){ int64_t t=base+j*4+grp; uint32_t id = use_ids? (uint32_t)ids[t] : hash32((uint32_t)t*2654435761u+12345u)%(uint32_t)nrows_tab; row[j]=id; }
    if(MODE!=2){
      #pragma unroll
      for(int j=0;j<UN;j++) r[j]=*reinterpret_cast<const f4*>(W+row[j]*64+sub*4);
    }
    #pragma unroll
    for(int j=0;j<UN;j++){
      if(MODE==0) acc+=r[j];
      else if(MODE==1) *reinterpret_cast<f4*>(W+row[j]*64+sub*4)=r[j]*0.999f+0.001f;
      else { f4 v={1.f,2.f,3.f,(float)j}; *reinterpret_cast<f4*>(W+row[j]*64+sub*4)=v; }
    }
  }
  if(MODE==0 && acc.x==12345.678f) out[0]=acc.y;
}

// random 4-byte ops. mode 0: atomicAdd no return, 1: plain store, 2: load+sum, 3: atomicAdd returning
template<int MODE>
__global__ __launch_bounds__(256) void rand_words(int* __restrict__ C, int64_t n_tab, int64_t nrefs, int* out){
  int64_t s=(int64_t)gridDim.x*blockDim.x; int acc=0;
  for(int64_t t=(int64_t)blockIdx.x*blockDim.x+threadIdx.x;t<nrefs;t+=s){
    uint32_t id=hash32((uint32_t)t*2654435761u+777u)%(uint32_t)n_tab;
    if(MODE==0) atomicAdd(C+id,1); else if(MODE==1) C[id]=(int)t; else if(MODE==2) acc+=C[id]; else acc+=atomicAdd(C+id,1);
  }
  if((MODE>=2) && acc==123456789) out[0]=acc;
}

template<class F> float timeit(F f,int reps){ hipEvent_t a,b; CK(hipEventCreate(&a));CK(hipEventCreate(&b)); f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); for(int i=0;i<reps;i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps; }

int main(int argc,char**argv){
  int64_t NROWS = argc>1? atoll(argv[1]) : 1000000;   // rows per table
  int64_t NREF = 65536*3;
  float* W; CK(hipMalloc(&W,(size_t)NROWS*64*4)); CK(hipMemset(W,0,(size_t)NROWS*64*4));
  float* W2; CK(hipMalloc(&W2,(size_t)NROWS*64*4));
  int* C; CK(hipMalloc(&C,(size_t)NROWS*4)); CK(hipMemset(C,0,(size_t)NROWS*4));
  int* ids; CK(hipMalloc(&ids,NREF*4)); std::vector<int> h(NREF); for(auto&x:h) x=(int)(((uint64_t)rand()*RAND_MAX+rand())%NROWS); CK(hipMemcpy(ids,h.data(),NREF*4,hipMemcpyHostToDevice));
  float* out; CK(hipMalloc(&out,64)); int* iout=(int*)out;
  printf("table rows %lld (%.0f MB)\n",(long long)NROWS,NROWS*256.0/1e6);
  { int64_t n=NROWS*16; float ms=timeit([&]{ copy_k<<<256*8,256>>>((f4*)W,(f4*)W2,n); },5); printf("stream copy: %.2f TB/s (R+W)\n", 2.0*n*16/ms/1e9); }
  // per "step" = 196608 row refs (like one BPR batch), more reps
  for(int big=0; big<2; big++){
    int64_t nref = big? NREF*16 : NREF;   // big: hash ids only
    printf("--- nrefs=%lld %s\n",(long long)nref, big?"(hash ids)":"(loaded ids)");
    int use_ids = big?0:1;
    #define RUN(UN,MODE,name,bytes) { int64_t blocks=(nref+16*UN-1)/(16*UN); float ms=timeit([&]{ rand_rows<UN,MODE><<<dim3(blocks),256>>>(W,ids,NROWS,nref,out,use_ids); },20); printf("%s UN=%d: %.1f us, %.2f TB/s, %.1f rows/ns\n",name,UN,ms*1e3,(double)nref*bytes/ms/1e9,(double)nref/ms/1e6); }
    RUN(1,0,"read ",256) RUN(2,0,"read ",256) RUN(4,0,"read ",256) RUN(8,0,"read ",256)
    RUN(1,1,"rmw  ",512) RUN(2,1,"rmw  ",512) RUN(4,1,"rmw  ",512) RUN(8,1,"rmw  ",512)
    RUN(1,2,"write",256) RUN(4,2,"write",256)
  }
  // persistent variants: fixed grid
  { int64_t nref=NREF*16; for(int bpc: {4,8}){ int blocks=256*bpc; float ms=timeit([&]{ rand_rows<4,1><<<blocks,256>>>(W,ids,NROWS,nref,out,0); },20); printf("rmw persistent grid=%d UN=4: %.1f us %.2f TB/s\n",blocks,ms*1e3,(double)nref*512/ms/1e9);} }
  for(int64_t nref: {NREF, NREF*16}){
    printf("--- words nrefs=%lld\n",(long long)nref);
    #define RW(MODE,name) { int64_t blocks=(nref+255)/256; float ms=timeit([&]{ rand_words<MODE><<<dim3(blocks),256>>>(C,NROWS,nref,iout); },20); printf("%s: %.1f us, %.1f ops/ns\n",name,ms*1e3,(double)nref/ms/1e6); }
    RW(0,"atomic noret") RW(3,"atomic ret  ") RW(1,"plain store ") RW(2,"plain load  ")
  }
  return 0;
}
