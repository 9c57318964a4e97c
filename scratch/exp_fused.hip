// Experiment: cost of each component of the fused BPR step (D=64, B=65536, 1M x 1M tables)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
template <int CTRL> __device__ __forceinline__ float dpp_f(float x){ return __builtin_bit_cast(float,__builtin_amdgcn_update_dpp(0,__builtin_bit_cast(int,x),CTRL,0xF,0xF,true)); }
__device__ __forceinline__ float red16(float x){ x+=dpp_f<0xB1>(x); x+=dpp_f<0x4E>(x); x+=dpp_f<0x141>(x); x+=dpp_f<0x140>(x); return x; }
__device__ __forceinline__ float dot4(f4 a,f4 b){return a.x*b.x+a.y*b.y+a.z*b.z+a.w*b.w;}

struct A { float*U,*V,*b; int*cU,*cV; int*c2U,*c2V; const int*uid,*pid,*nid; const int*uid2,*pid2,*nid2; unsigned char* dm; float* part; int B; float lr, invB; };

// flags: 1 bias, 2 cnt read, 4 cnt reset, 8 embedded atomics for next step, 16 tag store (plain) for next step, 32 extra tag load (pass B), 64 loss math
template<int F, int UN, int BS>
__global__ __launch_bounds__(BS) void fused(A a){
  const int lane=threadIdx.x&63, sub=lane&15, grp=lane>>4;
  const int64_t gw=(int64_t)blockIdx.x*(BS/64)+(threadIdx.x>>6);
  float lacc=0, sacc=0;
  const int base=(int)gw*4*UN;
  #pragma unroll
  for(int j=0;j<UN;j++){
    int t=base+j*4+grp; if(t>=a.B) break;
    int u=a.uid[t],p=a.pid[t],n=a.nid[t];
    int cu=1,cp=1,cn=1;
    if(F&2){cu=a.cU[u];cp=a.cV[p];cn=a.cV[n];}
    if(F&32){ cu+=a.c2U[u]>>30; cp+=a.c2V[p]>>30; cn+=a.c2V[n]>>30; }
    float*Up=a.U+(size_t)u*64+sub*4,*Pp=a.V+(size_t)p*64+sub*4,*Np=a.V+(size_t)n*64+sub*4;
    f4 ru,rp,rn;
    if(F&256){ ru=__builtin_nontemporal_load((f4*)Up); rp=__builtin_nontemporal_load((f4*)Pp); rn=__builtin_nontemporal_load((f4*)Np);} else { ru=*(f4*)Up; rp=*(f4*)Pp; rn=*(f4*)Np; }
    float bp=0,bn=0; if(F&1){bp=a.b[p];bn=a.b[n];}
    if(F&8){ if(sub==0){ atomicAdd(a.c2U+a.uid2[t],1); atomicAdd(a.c2V+a.pid2[t],1); atomicAdd(a.c2V+a.nid2[t],1);} }
    if(F&16){ if(sub==0){ a.c2U[a.uid2[t]]=t; a.c2V[a.pid2[t]]=t|(1<<20); a.c2V[a.nid2[t]]=t|(2<<20);} }
    float x=red16(dot4(ru,rp-rn))+bp-bn;
    float g;
    if(F&64){ float m=fmaxf(x,-30.f); float e=__expf(-fabsf(m)); lacc+= (sub==0)?(fmaxf(-m,0.f)+log1pf(e))*a.invB:0.f; float sig=(x>=0)?e/(1+e):1.f/(1+e); g=(x>=-30.f)?-sig*a.invB:0.f; sacc+=dot4(ru,ru)+dot4(rp,rp)+dot4(rn,rn);} else g=x*1e-6f;
    f4 gu=g*(rp-rn)+ru, gp=g*ru+rp, gn=-g*ru+rn;
    if(F&128){
      if(cu==1){ __builtin_nontemporal_store(ru-a.lr*gu,(f4*)Up); }
      if(cp==1){ __builtin_nontemporal_store(rp-a.lr*gp,(f4*)Pp); if(sub==0){ if(F&1) a.b[p]=bp-a.lr*g; } }
      if(cn==1){ __builtin_nontemporal_store(rn-a.lr*gn,(f4*)Np); if(sub==0){ if(F&1) a.b[n]=bn+a.lr*g; } }
    } else {
    if(cu==1){ *(f4*)Up=ru-a.lr*gu; if((F&4)&&sub==0) a.cU[u]=0; }
    if(cp==1){ *(f4*)Pp=rp-a.lr*gp; if(sub==0){ if(F&1) a.b[p]=bp-a.lr*g; if(F&4) a.cV[p]=0; } }
    if(cn==1){ *(f4*)Np=rn-a.lr*gn; if(sub==0){ if(F&1) a.b[n]=bn+a.lr*g; if(F&4) a.cV[n]=0; } }
    }
    if((F&2)&&sub==0) a.dm[t]=(cu!=1)|((cp!=1)<<1)|((cn!=1)<<2);
  }
  if(F&64){ for(int o=32;o>0;o>>=1){ lacc+=__shfl_xor(lacc,o); sacc+=__shfl_xor(sacc,o);} if(lane==0){a.part[2*gw]=lacc;a.part[2*gw+1]=sacc;} }
}
template<class Fn> float timeit(Fn f,int reps){ hipEvent_t a,b; CK(hipEventCreate(&a));CK(hipEventCreate(&b)); for(int i=0;i<3;i++) f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); for(int i=0;i<reps;i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps; }
int main(int argc,char**argv){ bool do_sort = argc>1;
  const int N=1000000,B=65536,K=16; A a;
  CK(hipMalloc(&a.U,(size_t)N*256)); CK(hipMalloc(&a.V,(size_t)N*256)); CK(hipMalloc(&a.b,N*4));
  CK(hipMemset(a.U,0,(size_t)N*256)); CK(hipMemset(a.V,0,(size_t)N*256)); CK(hipMemset(a.b,0,N*4));
  CK(hipMalloc(&a.cU,N*4)); CK(hipMalloc(&a.cV,N*4)); CK(hipMalloc(&a.c2U,N*4)); CK(hipMalloc(&a.c2V,N*4));
  std::vector<int> ones(N,1); CK(hipMemcpy(a.cU,ones.data(),N*4,hipMemcpyHostToDevice)); CK(hipMemcpy(a.cV,ones.data(),N*4,hipMemcpyHostToDevice)); CK(hipMemset(a.c2U,0,N*4)); CK(hipMemset(a.c2V,0,N*4));
  int* ids; CK(hipMalloc(&ids,(size_t)3*K*B*4)); std::vector<int> h((size_t)3*K*B); srand(1); for(auto&x:h) x=(int)(((uint64_t)rand()*2147483647ull+rand())%N); if(do_sort){
    for(int s=0;s<K;s++){ std::vector<int> perm(B); for(int i=0;i<B;i++) perm[i]=i; int* u=&h[(size_t)s*B]; int* p=&h[(size_t)(K+s)*B]; int* n=&h[(size_t)(2*K+s)*B];
      std::sort(perm.begin(),perm.end(),[&](int a,int b){return u[a]<u[b];}); std::vector<int> uu(B),pp(B),nn(B); for(int i=0;i<B;i++){uu[i]=u[perm[i]];pp[i]=p[perm[i]];nn[i]=n[perm[i]];} for(int i=0;i<B;i++){u[i]=uu[i];p[i]=pp[i];n[i]=nn[i];} } printf("triplets sorted by user id\n"); }
  CK(hipMemcpy(ids,h.data(),h.size()*4,hipMemcpyHostToDevice));
  CK(hipMalloc(&a.dm,B)); CK(hipMalloc(&a.part,65536*8)); a.B=B; a.lr=0.05f; a.invB=1.f/B;
  int step=0;
  auto setids=[&](){ int s=step%K, s2=(step+1)%K; a.uid=ids+(size_t)s*B; a.pid=ids+(size_t)(K+s)*B; a.nid=ids+(size_t)(2*K+s)*B; a.uid2=ids+(size_t)s2*B; a.pid2=ids+(size_t)(K+s2)*B; a.nid2=ids+(size_t)(2*K+s2)*B; step++; };
  #define RUN(F,UN,name) { float ms=timeit([&]{ setids(); hipLaunchKernelGGL((fused<F,UN,256>),dim3(B/(16*UN)),dim3(256),0,0,a); },48); printf("%-46s UN=%d: %.1f us  (%.2f TB/s alg)\n",name,UN,ms*1e3,B*1564.0/ms/1e9); CK(hipMemcpy(a.cU,ones.data(),N*4,hipMemcpyHostToDevice)); CK(hipMemcpy(a.cV,ones.data(),N*4,hipMemcpyHostToDevice)); }
  #define RUNB(F,BS,name) { float ms=timeit([&]{ setids(); hipLaunchKernelGGL((fused<F,1,BS>),dim3(B*16/BS),dim3(BS),0,0,a); },48); printf("%-30s BS=%d: %.1f us  (%.2f TB/s alg)\n",name,BS,ms*1e3,B*1564.0/ms/1e9); }
  RUNB(65,64,"rows+bias+loss")
  RUNB(65,128,"rows+bias+loss")
  RUNB(65,256,"rows+bias+loss")
  RUNB(65,512,"rows+bias+loss")
  RUNB(65,1024,"rows+bias+loss")
  RUNB(0,64,"rows only")
  RUNB(0,256,"rows only")
  RUNB(0,1024,"rows only")
  return 0;
}
