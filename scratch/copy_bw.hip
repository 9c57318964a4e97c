// Streaming-copy yardstick for the HBM roofline (the guide quotes ~6.3 TB/s for a float4 copy;
// scratch/exp2.hip reached 5.8-5.9).  Variants: bytes per buffer, workgroups per CU, loads in flight, plain vs nontemporal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
template <int UN, int NT>
__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  // block-contiguous: every workgroup owns a contiguous run, UN float4 in flight per thread
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)256 * UN) {
    f4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) { const size_t j = i + (size_t)u * 256; if (j < hi) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j]; }
#pragma unroll
    for (int u = 0; u < UN; ++u) { const size_t j = i + (size_t)u * 256; if (j < hi) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; } }
  }
}
template <int UN, int NT>
__global__ __launch_bounds__(256) void copy_strided_k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
  // grid-stride: consecutive workgroups touch consecutive 4-KB pieces
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256 * UN) {
    f4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) { const size_t j = i + (size_t)u * gridDim.x * 256; if (j < n) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j]; }
#pragma unroll
    for (int u = 0; u < UN; ++u) { const size_t j = i + (size_t)u * gridDim.x * 256; if (j < n) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; } }
  }
}
template <typename F> static float timeit(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1000.f / reps;
}
int main() {
  for (size_t mb : {256, 1024, 4096}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    f4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    const int reps = mb >= 4096 ? 5 : 20;
    auto rep = [&](const char* name, float us) { printf("%5zu MB  %-34s %9.1f us  %6.2f TB/s (read + write)\n", mb, name, us, 2.0 * bytes / us * 1e-6); };
    for (int wg : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
      char nm[64];
      snprintf(nm, 64, "contig un4 plain  wg=%d", wg); rep(nm, timeit([&] { copy_k<4, 0><<<wg, 256>>>(a, b, n); }, reps));
      snprintf(nm, 64, "contig un8 plain  wg=%d", wg); rep(nm, timeit([&] { copy_k<8, 0><<<wg, 256>>>(a, b, n); }, reps));
      snprintf(nm, 64, "contig un8 nontmp wg=%d", wg); rep(nm, timeit([&] { copy_k<8, 1><<<wg, 256>>>(a, b, n); }, reps));
    }
    rep("grid-stride un4 plain wg=8192", timeit([&] { copy_strided_k<4, 0><<<8192, 256>>>(a, b, n); }, reps));
    rep("grid-stride un4 nontmp wg=8192", timeit([&] { copy_strided_k<4, 1><<<8192, 256>>>(a, b, n); }, reps));
    rep("one float4 per thread", timeit([&] { copy_strided_k<1, 0><<<(unsigned)(n / 256), 256>>>(a, b, n); }, reps));
    { float us = timeit([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, reps); rep("hipMemcpyAsync D2D", us); }
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
