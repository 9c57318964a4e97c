// Round-5 experiment (VERDICT r4 "next" #2): what the four scattered 4-byte item-bias requests per triplet cost the fused BPR step
// (D = 64, B = 65536, 1M x 1M tables, uniform ids), and what any "side pass" that takes them out of the fused launch has to beat.
//   A  racy rows + bias in its own [N,1] table + loss math           (= the floor of the product kernel's access pattern)
//   B  racy rows + loss math, biases arrive in a COALESCED per-triplet record (float2), bias gradients leave as ONE coalesced float
//      per triplet: zero scattered 4-byte requests in the launch     (= the fused kernel if a side pass delivered the biases for free)
//   C  rows + loss math, no bias at all
//   G  the side pass's gather alone: bpn[t] = (b[p[t]], b[n[t]])     (131 k scattered 4-byte reads, coalesced 8-byte stores)
//   S  the side pass's apply alone: b[p[t]] -= lr g[t], b[n[t]] += lr g[t]   (131 k scattered 4-byte read-modify-writes; plain and atomic)
// A side pass pays G + S (+ two kernel boundaries per step) to turn A into B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
template <int CTRL> __device__ __forceinline__ float dpp_f(float x){ return __builtin_bit_cast(float,__builtin_amdgcn_update_dpp(0,__builtin_bit_cast(int,x),CTRL,0xF,0xF,true)); }
__device__ __forceinline__ float red16(float x){ x+=dpp_f<0xB1>(x); x+=dpp_f<0x4E>(x); x+=dpp_f<0x141>(x); x+=dpp_f<0x140>(x); return x; }
__device__ __forceinline__ float dot4(f4 a,f4 b){return a.x*b.x+a.y*b.y+a.z*b.z+a.w*b.w;}

struct A3 { float *U, *V, *b; const int *uid, *pid, *nid; const float2* bpn; float* gout; float* part; int B; float lr, invB; };

// BIAS: 0 none, 1 own table (scattered), 2 coalesced record in / coalesced gradient out,
//   3 own table, reads only; 4 own table, writes only (split of A's cost);
//   5 bias rows PADDED to 64 bytes (pitch 16 floats): 4-byte read, the 16 lanes of the group write the whole 64-byte sector;
//   6 padded to 128 bytes (pitch 32): the group writes the whole line (8 bytes per lane)
template <int BIAS>
__global__ __launch_bounds__(256) void step_k(A3 a) {
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float lacc = 0, sacc = 0;
  const int t = (int)gw * 4 + grp;
  if (t < a.B) {
    const int u = a.uid[t], p = a.pid[t], n = a.nid[t];
    float* Up = a.U + (size_t)u * 64 + sub * 4; float* Pp = a.V + (size_t)p * 64 + sub * 4; float* Np = a.V + (size_t)n * 64 + sub * 4;
    f4 ru = *(f4*)Up, rp = *(f4*)Pp, rn = *(f4*)Np;
    float bp = 0, bn = 0;
    if (BIAS == 1 || BIAS == 3) { bp = a.b[p]; bn = a.b[n]; }
    if (BIAS == 5) { bp = a.b[(size_t)p * 16]; bn = a.b[(size_t)n * 16]; }
    if (BIAS == 6) { bp = a.b[(size_t)p * 32]; bn = a.b[(size_t)n * 32]; }
    if (BIAS == 2) { const float2 v = a.bpn[t]; bp = v.x; bn = v.y; }
    const float x = red16(dot4(ru, rp - rn)) + bp - bn;
    const float m = fmaxf(x, -30.f); const float e = __expf(-fabsf(m));
    lacc += (sub == 0) ? (fmaxf(-m, 0.f) + log1pf(e)) * a.invB : 0.f;
    const float sig = (x >= 0) ? e / (1 + e) : 1.f / (1 + e); const float g = (x >= -30.f) ? -sig * a.invB : 0.f;
    sacc += dot4(ru, ru) + dot4(rp, rp) + dot4(rn, rn);
    const f4 gu = g * (rp - rn) + ru, gp = g * ru + rp, gn = -g * ru + rn;
    *(f4*)Up = ru - a.lr * gu; *(f4*)Pp = rp - a.lr * gp; *(f4*)Np = rn - a.lr * gn;
    if ((BIAS == 1 || BIAS == 4) && sub == 0) { a.b[p] = bp - a.lr * g; a.b[n] = bn + a.lr * g; }
    if (BIAS == 3 && sub == 0) lacc += (bp + bn) * 1e-30f;
    if (BIAS == 5) { a.b[(size_t)p * 16 + sub] = bp - a.lr * g; a.b[(size_t)n * 16 + sub] = bn + a.lr * g; }
    if (BIAS == 6) { *(float2*)(a.b + (size_t)p * 32 + 2 * sub) = make_float2(bp - a.lr * g, 0.f); *(float2*)(a.b + (size_t)n * 32 + 2 * sub) = make_float2(bn + a.lr * g, 0.f); }
    if (BIAS == 2 && sub == 0) a.gout[t] = g;
  }
  for (int o = 32; o > 0; o >>= 1) { lacc += __shfl_xor(lacc, o); sacc += __shfl_xor(sacc, o); }
  if (lane == 0) { a.part[2 * gw] = lacc; a.part[2 * gw + 1] = sacc; }
}
__global__ __launch_bounds__(256) void gather_k(A3 a, float2* out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < a.B) out[t] = make_float2(a.b[a.pid[t]], a.b[a.nid[t]]);
}
template <int ATOMIC>
__global__ __launch_bounds__(256) void scatter_k(A3 a, const float* g) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < a.B) {
    const float d = a.lr * g[t];
    if (ATOMIC) { atomicAdd(a.b + a.pid[t], -d); atomicAdd(a.b + a.nid[t], d); }
    else { a.b[a.pid[t]] -= d; a.b[a.nid[t]] += d; }
  }
}
template <class Fn> float timeit(Fn f, int reps) { hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); for (int i = 0; i < 3; i++) f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps; }

int main() {
  const int N = 1000000, B = 65536, K = 16;
  std::vector<int> h((size_t)3 * K * B); srand(1); for (auto& x : h) x = (int)(((uint64_t)rand() * 2147483647ull + rand()) % N);
  int* d_raw; CK(hipMalloc(&d_raw, h.size() * 4)); CK(hipMemcpy(d_raw, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  A3 a; memset(&a, 0, sizeof(a));
  CK(hipMalloc(&a.U, (size_t)N * 256)); CK(hipMalloc(&a.V, (size_t)N * 256)); CK(hipMalloc(&a.b, (size_t)N * 128)); CK(hipMemset(a.b, 0, (size_t)N * 128));
  std::vector<float> init((size_t)N * 64); { uint32_t st = 12345u; for (auto& x : init) { st = st * 1664525u + 1013904223u; x = ((st >> 8) / 16777216.0f - 0.5f) * 0.1f; } }
  CK(hipMemcpy(a.U, init.data(), init.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(a.V, init.data(), init.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(a.b, init.data(), (size_t)N * 4, hipMemcpyHostToDevice));
  float2* bpn; float* gout; CK(hipMalloc(&bpn, (size_t)B * 8)); CK(hipMalloc(&gout, (size_t)B * 4)); CK(hipMemset(bpn, 0, (size_t)B * 8)); CK(hipMemset(gout, 0, (size_t)B * 4));
  CK(hipMalloc(&a.part, 65536 * 8)); a.B = B; a.lr = 0.05f; a.invB = 1.f / B; a.bpn = bpn; a.gout = gout;
  int step = 0;
  auto setids = [&]() { const int s = step % K; step++; a.uid = d_raw + (size_t)s * B; a.pid = d_raw + (size_t)(K + s) * B; a.nid = d_raw + (size_t)(2 * K + s) * B; };
#define RUN(name, launch) { step = 0; float ms = timeit([&] { setids(); launch; }, 64); printf("%-78s %6.2f us\n", name, ms * 1e3); }
  for (int rep = 0; rep < 3; ++rep) {
    printf("=== pass %d\n", rep);
    RUN("A  rows + bias table (4 scattered 4-byte requests per triplet) + loss math", (step_k<1><<<B / 16, 256>>>(a)));
    RUN("B  rows + loss math, biases in a coalesced record, gradient out coalesced", (step_k<2><<<B / 16, 256>>>(a)));
    RUN("C  rows + loss math, no bias", (step_k<0><<<B / 16, 256>>>(a)));
    RUN("A3 as A, bias READS only", (step_k<3><<<B / 16, 256>>>(a)));
    RUN("A4 as A, bias WRITES only", (step_k<4><<<B / 16, 256>>>(a)));
    RUN("A5 bias rows padded to 64 B: 4-byte read, whole-sector write by the lane group", (step_k<5><<<B / 16, 256>>>(a)));
    RUN("A6 bias rows padded to 128 B: 4-byte read, whole-line write by the lane group", (step_k<6><<<B / 16, 256>>>(a)));
    RUN("G  side pass, gather alone (131 k scattered reads -> coalesced record)", (gather_k<<<B / 256, 256>>>(a, bpn)));
    RUN("S  side pass, apply alone, plain read-modify-write", (scatter_k<0><<<B / 256, 256>>>(a, gout)));
    RUN("S' side pass, apply alone, fp32 atomics", (scatter_k<1><<<B / 256, 256>>>(a, gout)));
    auto three = [&] { gather_k<<<B / 256, 256>>>(a, bpn); step_k<2><<<B / 16, 256>>>(a); scatter_k<0><<<B / 256, 256>>>(a, gout); };
    RUN("G + B + S as three launches per step (what a side pass makes of A)", three());
  }
  return 0;
}
