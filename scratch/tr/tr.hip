#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int mode) {
  __shared__ __attribute__((aligned(16))) short S[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) S[i] = (short)i;      // S[row][col], 64 cols: value = row*64+col
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // group g reads the block rows 4g..4g+3, cols 0..15: lane i supplies &S[4g + i/4][(i%4)*4]
  const short* p = mode == 0 ? &S[(4 * g + i / 4) * 64 + (i % 4) * 4] : &S[(4 * g + i % 4) * 64 + (i / 4) * 4];
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(d, mode); int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64); printf("\n"); }
  }
}
