// Round-2 experiments for the DLRM top-MLP products in fp16-MFMA mode (MI355X): C[M][N] = A16[M][K] * B16[N][K]^T (+bias, relu,
// fp32 store + fp16 copy), M = 8192 (batch), N, K <= 1024.  Variants:
//   base   : the round-1 kernel (128x128x32, one LDS buffer, two barriers per K step, 4-byte epilogue stores)
//   pipe   : BK = 64, swapped MFMA operands (each lane owns 4 consecutive columns: 16-byte stores), optional LDS double buffer
//   NOEPI  : the same without the epilogue's global traffic (what the main loop costs)
// and the weight-gradient product dW[Kin][Nout] = X16[B][Kin]^T * dZ16[B][Nout] straight from the batch-major fp16 copies
// (LDS transpose reads, ds_read_b64_tr_b16) with split-K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

struct G {
  const _Float16* A; int64_t lda;
  const _Float16* B; int64_t ldb;
  float* C; int64_t ldc;
  _Float16* C16; int64_t ldc16;
  const float* bias;
  int M, N, K, act;
};

// ---------------------------------------------------------------- round-1 kernel
struct TileLoader16 {
  h8 v[2];
  __device__ __forceinline__ void load(const _Float16* P, int64_t ld, int row0, int nrows, int k0) {
    const int t = threadIdx.x; const int r = t >> 1, kb = (t & 1) * 16; const bool rok = row0 + r < nrows;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + kb + 8 * h;
      h8 z; for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
      v[h] = (rok && k < ld) ? *reinterpret_cast<const h8*>(P + (int64_t)(row0 + r) * ld + k) : z;
    }
  }
  __device__ __forceinline__ void store(_Float16* T, int LD) const {
    const int t = threadIdx.x; const int r = t >> 1, kb = (t & 1) * 16;
    *reinterpret_cast<h8*>(T + r * LD + kb) = v[0];
    *reinterpret_cast<h8*>(T + r * LD + kb + 8) = v[1];
  }
};
template <int NOEPI>
__global__ __launch_bounds__(256) void base_k(G g) {
  constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 8;
  __shared__ __attribute__((aligned(16))) _Float16 As[BM * LD];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[BN * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
  TileLoader16 la, lb;
  la.load(g.A, g.lda, bm, g.M, 0);
  lb.load(g.B, g.ldb, bn, g.N, 0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    la.store(As, LD); lb.store(Bs, LD);
    __syncthreads();
    if (k0 + BK < g.K) { la.load(g.A, g.lda, bm, g.M, k0 + BK); lb.load(g.B, g.ldb, bn, g.N, k0 + BK); }
    h8 a[4], b[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const h8*>(&As[(wm + mi * 16 + (lane & 15)) * LD + (lane >> 4) * 8]);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const h8*>(&Bs[(wn + ni * 16 + (lane & 15)) * LD + (lane >> 4) * 8]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int col = bn + wn + ni * 16 + (lane & 15);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bm + wm + mi * 16 + (lane >> 4) * 4 + r;
        if (row < g.M && col < g.N) {
          float v = acc[mi][ni][r];
          if (NOEPI) { if (v == 12345.678f) g.C[0] = v; continue; }
          if (g.bias) v += g.bias[col];
          if (g.act == 1) v = fmaxf(v, 0.0f);
          g.C[(int64_t)row * g.ldc + col] = v;
          if (g.C16) g.C16[(int64_t)row * g.ldc16 + col] = (_Float16)v;
        }
      }
  }
}

// ---------------------------------------------------------------- pipelined kernel
// WM x WN wavefronts, each TM x TN tiles of 16x16; BK halves per stage; DBUF: two LDS stages, one barrier per K step
template <int WM, int WN, int TM, int TN, int BK, int DBUF, int NOEPI, int MINB, int REMAP = 0, int LDPAD = 8, int WMAP = 0>
__global__ __launch_bounds__(64 * WM * WN, MINB) void pipe_k(G g) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, LD = BK + LDPAD, NT = 64 * WM * WN;
  constexpr int CPR = BK / 8;                                  // 16-byte chunks per tile row
  constexpr int NA = BM * CPR / NT, NB = BN * CPR / NT;        // chunks per thread
  static_assert(BM * CPR % NT == 0 && BN * CPR % NT == 0, "tile/thread mismatch");
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  constexpr int STAGE = (BM + BN) * LD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bx = blockIdx.x, by = blockIdx.y;
  if (REMAP) {             // workgroup b runs on XCD b % 8: give every XCD a contiguous run of tiles (whole row blocks of A)
    const int nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const int per = nb >> 3;                                   // nb % 8 == 0 assumed here
    const int t = (b & 7) * per + (b >> 3);
    if (REMAP == 1) { bx = t % gridDim.x; by = t / gridDim.x; }
    else { by = t % gridDim.y; bx = t / gridDim.y; }          // column-major inside the XCD's run
  }
  const int bm = by * BM, bn = bx * BN;
  const int wm = (wave / WN) * TM * 16, wn = (wave % WN) * TN * 16;
  const int r16 = lane & 15, q = lane >> 4;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
  h8 ra[NA], rb[NB];
  h8 zero; for (int e = 0; e < 8; ++e) zero[e] = (_Float16)0.0f;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = threadIdx.x + NT * i, row = WMAP ? c % BM : c / CPR, k = k0 + (WMAP ? c / BM : c % CPR) * 8;
      ra[i] = (bm + row < g.M && k < g.lda) ? *reinterpret_cast<const h8*>(g.A + (int64_t)(bm + row) * g.lda + k) : zero;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = threadIdx.x + NT * i, row = WMAP ? c % BN : c / CPR, k = k0 + (WMAP ? c / BN : c % CPR) * 8;
      rb[i] = (bn + row < g.N && k < g.ldb) ? *reinterpret_cast<const h8*>(g.B + (int64_t)(bn + row) * g.ldb + k) : zero;
    }
  };
  auto lstore = [&](_Float16* S) {
#pragma unroll
    for (int i = 0; i < NA; ++i) { const int c = threadIdx.x + NT * i; *reinterpret_cast<h8*>(S + (WMAP ? c % BM : c / CPR) * LD + (WMAP ? c / BM : c % CPR) * 8) = ra[i]; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { const int c = threadIdx.x + NT * i; *reinterpret_cast<h8*>(S + (BM + (WMAP ? c % BN : c / CPR)) * LD + (WMAP ? c / BN : c % CPR) * 8) = rb[i]; }
  };
  auto compute = [&](const _Float16* S) {
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      h8 a[TM], b[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const h8*>(S + (wm + mi * 16 + r16) * LD + kk * 32 + q * 8);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const h8*>(S + (BM + wn + ni * 16 + r16) * LD + kk * 32 + q * 8);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);   // swapped: D[n][m]
    }
  };
  const int nk = (g.K + BK - 1) / BK;
  gload(0);
  if (DBUF) {
    lstore(lds);
    if (nk > 1) gload(BK);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      _Float16* cur = lds + (t & 1) * STAGE; _Float16* nxt = lds + ((t + 1) & 1) * STAGE;
      if (t + 1 < nk) lstore(nxt);
      if (t + 2 < nk) gload((t + 2) * BK);
      compute(cur);
      __syncthreads();
    }
  } else {
    for (int t = 0; t < nk; ++t) {
      lstore(lds);
      __syncthreads();
      if (t + 1 < nk) gload((t + 1) * BK);
      compute(lds);
      __syncthreads();
    }
  }
  // epilogue: lane owns C[m = .. + r16][n = .. + 4q .. 4q+3]
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int row = bm + wm + mi * 16 + r16;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int col = bn + wn + ni * 16 + q * 4;
      f32x4 v = acc[mi][ni];
      if (NOEPI) { if (v.x == 12345.678f) g.C[0] = v.x; continue; }
      if (row < g.M && col < g.N) {                      // N % 4 == 0
        if (g.bias) { const f32x4 bb = *reinterpret_cast<const f32x4*>(g.bias + col); v += bb; }
        if (g.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (g.C) *reinterpret_cast<f32x4*>(g.C + (int64_t)row * g.ldc + col) = v;
        if (g.C16) { h4 h; h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
                     *reinterpret_cast<h4*>(g.C16 + (int64_t)row * g.ldc16 + col) = h; }
      }
    }
  }
}


// ---------------------------------------------------------------- weight-gradient product from batch-major fp16 copies
// C[M][N] (+)= sum_k A[k][m] * B[k][n]: A = X16 [Kred][lda], B = dZ16 [Kred][ldb]; LDS keeps the tiles k-major as they
// arrive; MFMA fragments come from ds_read_b64_tr_b16 (lane i of a 16-lane group supplies &blk[i/4][4*(i%4)] of a [4][16]
// block and receives column i).  k order inside a 32-deep MFMA step: lane group q holds rows 4q..4q+3 and 16+4q..16+4q+3
// (the same for both operands), which keeps the two 16-lane groups of an LDS cycle on different banks at a 288-byte row pitch.
typedef short s4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt4(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 load_coh4(const float* p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ h4 tr_read(const _Float16* p) {
  s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p);
  return __builtin_bit_cast(h4, v);
}
struct GT { const _Float16* A; int64_t lda; const _Float16* B; int64_t ldb; float* C; int64_t ldc; int M, N, K, kchunk; float* slab; int* cnt; };
template <int WM, int WN, int TM, int TN, int BK, int NOEPI, int MINB, int REMAP = 0>
__global__ __launch_bounds__(64 * WM * WN, MINB) void tn_k(GT g) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = 64 * WM * WN;
  constexpr int LDM = BM + 16, LDN = BN + 16;                    // +32 bytes per k row
  constexpr int CA = BM / 8, CB = BN / 8;                        // 16-byte chunks per k row
  constexpr int NA = BK * CA / NT, NB = BK * CB / NT;
  static_assert(BK * CA % NT == 0 && BK * CB % NT == 0, "tile/thread mismatch");
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  constexpr int STAGE = BK * (LDM + LDN);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (REMAP) {                    // workgroup b runs on XCD b % 8: one XCD takes whole K slices (all tiles of a slice share nothing with other slices)
    const int nt = gridDim.x * gridDim.y, nb = nt * gridDim.z, b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int t = (b & 7) * (nb >> 3) + (b >> 3);
    bz = t / nt; const int r = t - bz * nt; by = r / gridDim.x; bx = r - by * gridDim.x;
  }
  const int bm = by * BM, bn = bx * BN;
  const int wm = (wave / WN) * TM * 16, wn = (wave % WN) * TN * 16;
  const int i16 = lane & 15, q = lane >> 4;
  const int kbeg = bz * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
  h8 ra[NA], rb[NB];
  h8 zero; for (int e = 0; e < 8; ++e) zero[e] = (_Float16)0.0f;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = threadIdx.x + NT * i, kr = c / CA, col = bm + (c % CA) * 8;
      ra[i] = (k0 + kr < kend && col < g.lda) ? *reinterpret_cast<const h8*>(g.A + (int64_t)(k0 + kr) * g.lda + col) : zero;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = threadIdx.x + NT * i, kr = c / CB, col = bn + (c % CB) * 8;
      rb[i] = (k0 + kr < kend && col < g.ldb) ? *reinterpret_cast<const h8*>(g.B + (int64_t)(k0 + kr) * g.ldb + col) : zero;
    }
  };
  auto lstore = [&](_Float16* S) {
#pragma unroll
    for (int i = 0; i < NA; ++i) { const int c = threadIdx.x + NT * i; *reinterpret_cast<h8*>(S + (c / CA) * LDM + (c % CA) * 8) = ra[i]; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { const int c = threadIdx.x + NT * i; *reinterpret_cast<h8*>(S + BK * LDM + (c / CB) * LDN + (c % CB) * 8) = rb[i]; }
  };
  // this lane's address inside a [4][16] block: row i/4, columns 4*(i%4)..
  const int tr_a = (q * 4 + i16 / 4) * LDM + (i16 % 4) * 4, tr_b = (q * 4 + i16 / 4) * LDN + (i16 % 4) * 4;
  auto compute = [&](const _Float16* S) {
    const _Float16* SA = S; const _Float16* SB = S + BK * LDM;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      h8 a[TM], b[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const h4 lo = tr_read(SA + (kk * 32) * LDM + tr_a + wm + mi * 16), hi = tr_read(SA + (kk * 32 + 16) * LDM + tr_a + wm + mi * 16);
        a[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const h4 lo = tr_read(SB + (kk * 32) * LDN + tr_b + wn + ni * 16), hi = tr_read(SB + (kk * 32 + 16) * LDN + tr_b + wn + ni * 16);
        b[ni] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
    }
  };
  const int nk = (kend - kbeg + BK - 1) / BK;
  gload(kbeg);
  lstore(lds);
  if (nk > 1) gload(kbeg + BK);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    _Float16* cur = lds + (t & 1) * STAGE; _Float16* nxt = lds + ((t + 1) & 1) * STAGE;
    if (t + 1 < nk) lstore(nxt);
    if (t + 2 < nk) gload(kbeg + (t + 2) * BK);
    compute(cur);
    __syncthreads();
  }
  if (NOEPI >= 2) {             // split-K through slabs: plain 16-byte stores, the last block of a tile sums the slices in order
    const int S = gridDim.z, tile = by * gridDim.x + bx;
    float* mine = g.slab + ((size_t)tile * S + bz) * (BM * BN);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
      { float* p = mine + (wm + mi * 16 + i16) * BN + wn + ni * 16 + q * 4;
        if (NOEPI == 3) store_wt4(p, acc[mi][ni]); else *reinterpret_cast<f32x4*>(p) = acc[mi][ni]; }
    if (NOEPI == 4) return;
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (NOEPI == 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      const int tk = __hip_atomic_fetch_add(g.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = tk == S - 1;
      if (tk == S - 1) { g.cnt[tile] = 0; if (NOEPI == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    }
    __syncthreads();
    if (!s_last) return;
    const float* base = g.slab + (size_t)tile * S * (BM * BN);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int row = bm + wm + mi * 16 + i16;
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = bn + wn + ni * 16 + q * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (NOEPI == 3) {
          for (int z0 = 0; z0 < S; z0 += 4) {
            f32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (z0 + u < S) w[u] = load_coh4(base + (size_t)(z0 + u) * (BM * BN) + (wm + mi * 16 + i16) * BN + wn + ni * 16 + q * 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) if (z0 + u < S) v += w[u];
          }
        } else
        for (int z = 0; z < S; ++z) v += *reinterpret_cast<const f32x4*>(base + (size_t)z * (BM * BN) + (wm + mi * 16 + i16) * BN + wn + ni * 16 + q * 4);
        if (row < g.M && col < g.N) { float* p = g.C + (int64_t)row * g.ldc + col; f32x4 o = *reinterpret_cast<f32x4*>(p); o += v; *reinterpret_cast<f32x4*>(p) = o; }
      }
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int row = bm + wm + mi * 16 + i16;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int col = bn + wn + ni * 16 + q * 4;
      const f32x4 v = acc[mi][ni];
      if (NOEPI == 1) { if (v.x == 12345.678f) g.C[0] = v.x; continue; }
      if (row < g.M && col < g.N) {
        float* p = g.C + (int64_t)row * g.ldc + col;
        if (gridDim.z > 1) { unsafeAtomicAdd(p, v.x); unsafeAtomicAdd(p + 1, v.y); unsafeAtomicAdd(p + 2, v.z); unsafeAtomicAdd(p + 3, v.w); }
        else *reinterpret_cast<f32x4*>(p) = v;
      }
    }
  }
}
// slabs [tile][S][BM*BN] -> C += sum_z
template <int BM, int BN>
__global__ __launch_bounds__(256) void slab_reduce_k(GT g, int S, int ntx) {
  const int tile = blockIdx.x, by = tile / ntx, bx = tile - by * ntx;
  const float* base = g.slab + (size_t)tile * S * (BM * BN);
  for (int e = threadIdx.x + 256 * blockIdx.y; e < BM * BN / 4; e += 256 * gridDim.y) {
    const int r = (e * 4) / BN, c = (e * 4) % BN;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < S; ++z) v += *reinterpret_cast<const f32x4*>(base + (size_t)z * (BM * BN) + e * 4);
    const int row = by * BM + r, col = bx * BN + c;
    if (row < g.M && col < g.N) { float* p = g.C + (int64_t)row * g.ldc + col; f32x4 o = *reinterpret_cast<f32x4*>(p); o += v; *reinterpret_cast<f32x4*>(p) = o; }
  }
}
__global__ void ref_tn_k(GT g, float* out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
  if (col >= g.N) return;
  float s = 0.f;
  for (int k = 0; k < g.K; ++k) s += (float)g.A[(int64_t)k * g.lda + row] * (float)g.B[(int64_t)k * g.ldb + col];
  out[(int64_t)row * g.N + col] = s;
}

// ---------------------------------------------------------------- reference (one thread per output)
__global__ void ref_k(G g, float* out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
  if (col >= g.N) return;
  float s = 0.f;
  for (int k = 0; k < g.K; ++k) s += (float)g.A[(int64_t)row * g.lda + k] * (float)g.B[(int64_t)col * g.ldb + k];
  if (g.bias) s += g.bias[col];
  if (g.act == 1) s = fmaxf(s, 0.f);
  out[(int64_t)row * g.N + col] = s;
}
__global__ void diff_k(const float* a, const float* b, int64_t n, float* out) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(a[i] - b[i]));
  atomicMax((int*)out, __float_as_int(m));
}
__global__ void diff16_k(const _Float16* a, const float* b, int64_t n, float* out) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf((float)a[i] - (float)(_Float16)b[i]));
  atomicMax((int*)out, __float_as_int(m));
}

template <typename F> static float timeit(F f, int reps = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
  printf("M=%d N=%d K=%d  (%.2f GFLOP)\n", M, N, K, 2.0 * M * N * K * 1e-9);
  std::vector<_Float16> hA((size_t)M * K), hB((size_t)N * K); std::vector<float> hb(N);
  uint32_t s = 12345; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& x : hA) x = (_Float16)rnd(); for (auto& x : hB) x = (_Float16)(rnd() * 0.1f); for (auto& x : hb) x = rnd();
  _Float16 *A, *B, *C16; float *C, *Cr, *bias, *dmax;
  CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C16, (size_t)M * N * 2));
  CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&Cr, (size_t)M * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&dmax, 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  G g{A, K, B, K, C, N, C16, N, bias, M, N, K, 1};
  ref_k<<<dim3((N + 255) / 256, M), 256>>>(g, Cr); CK(hipDeviceSynchronize());
  auto check = [&](const char* name, float us) {
    float z = 0.f, d32, d16; CK(hipMemcpy(dmax, &z, 4, hipMemcpyHostToDevice));
    diff_k<<<1024, 256>>>(C, Cr, (int64_t)M * N, dmax); CK(hipMemcpy(&d32, dmax, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dmax, &z, 4, hipMemcpyHostToDevice));
    diff16_k<<<1024, 256>>>(C16, Cr, (int64_t)M * N, dmax); CK(hipMemcpy(&d16, dmax, 4, hipMemcpyDeviceToHost));
    printf("%-34s %8.2f us  %7.1f TFLOP/s   maxdiff fp32 %.2e fp16 %.2e\n", name, us, 2.0 * M * N * K / us * 1e-6, d32, d16);
    CK(hipMemset(C, 0, (size_t)M * N * 4)); CK(hipMemset(C16, 0, (size_t)M * N * 2));
  };
  auto noepi = [&](const char* name, float us) { printf("%-34s %8.2f us  %7.1f TFLOP/s   (no epilogue)\n", name, us, 2.0 * M * N * K / us * 1e-6); };
  const bool lds_only = argc > 6 && !strcmp(argv[6], "lds");
  if (lds_only) {
#define RUNL(PAD, WMAP_, NOEPI, NAME) do { constexpr int BM_ = 256, BN_ = 128; const size_t sh = (size_t)2 * (BM_ + BN_) * (64 + PAD) * 2; \
    auto kern = pipe_k<4, 2, 4, 4, 64, 1, NOEPI, 1, 1, PAD, WMAP_>; \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
    float us = timeit([&] { hipLaunchKernelGGL(kern, dim3((N + BN_ - 1) / BN_, (M + BM_ - 1) / BM_), dim3(512), sh, 0, g); }); \
    CK(hipGetLastError()); if (NOEPI) noepi(NAME, us); else check(NAME, us); } while (0)
    RUNL(8, 0, 1, "256x128 pad8 (current) NOEPI"); RUNL(16, 0, 1, "256x128 pad16 NOEPI"); RUNL(24, 0, 1, "256x128 pad24 NOEPI"); RUNL(40, 0, 1, "256x128 pad40 NOEPI");
    RUNL(8, 1, 1, "256x128 pad8 rowwise-writes NOEPI"); RUNL(8, 0, 0, "256x128 pad8 (current)"); RUNL(24, 0, 0, "256x128 pad24"); RUNL(8, 1, 0, "256x128 pad8 rowwise-writes");
    return 0;
  }
  check("base 128x128x32", timeit([&] { base_k<0><<<dim3((N + 127) / 128, (M + 127) / 128), 256>>>(g); }));
  noepi("base 128x128x32 NOEPI", timeit([&] { base_k<1><<<dim3((N + 127) / 128, (M + 127) / 128), 256>>>(g); }));
#define RUN(WM, WN, TM, TN, BK, DB, NOEPI, MINB, NAME) RUNR(WM, WN, TM, TN, BK, DB, NOEPI, MINB, 0, NAME)
#define RUNR(WM, WN, TM, TN, BK, DB, NOEPI, MINB, REMAP, NAME) do { \
    constexpr int BM_ = WM * TM * 16, BN_ = WN * TN * 16; const size_t sh = (size_t)(DB ? 2 : 1) * (BM_ + BN_) * (BK + 8) * 2; \
    auto kern = pipe_k<WM, WN, TM, TN, BK, DB, NOEPI, MINB, REMAP>; \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
    float us = timeit([&] { hipLaunchKernelGGL(kern, dim3((N + BN_ - 1) / BN_, (M + BM_ - 1) / BM_), dim3(64 * WM * WN), sh, 0, g); }); \
    CK(hipGetLastError()); if (NOEPI) noepi(NAME, us); else check(NAME, us); } while (0)
  RUN(2, 2, 4, 4, 32, 0, 0, 2, "pipe 128x128x32 sbuf");
  RUN(2, 2, 4, 4, 64, 0, 0, 2, "pipe 128x128x64 sbuf");
  RUN(2, 2, 4, 4, 64, 1, 0, 2, "pipe 128x128x64 dbuf");
  RUN(2, 2, 4, 4, 64, 1, 1, 2, "pipe 128x128x64 dbuf NOEPI");
  RUN(2, 2, 4, 4, 32, 1, 0, 2, "pipe 128x128x32 dbuf");
  RUN(4, 2, 4, 4, 64, 1, 0, 1, "pipe 256x128x64 dbuf 8w");
  RUN(4, 2, 4, 4, 64, 1, 1, 1, "pipe 256x128x64 dbuf 8w NOEPI");
  RUN(2, 4, 4, 4, 64, 1, 0, 1, "pipe 128x256x64 dbuf 8w");
  RUN(2, 2, 8, 4, 64, 1, 0, 1, "pipe 256x128x64 dbuf 4w(8x4)");
  RUN(4, 4, 4, 4, 64, 1, 0, 1, "pipe 256x256x64 dbuf 16w");
  RUN(2, 2, 2, 4, 64, 1, 0, 2, "pipe 64x128x64 dbuf");
  RUNR(2, 2, 4, 4, 64, 1, 0, 2, 1, "pipe 128x128x64 dbuf remap1");
  RUNR(2, 2, 4, 4, 64, 1, 1, 2, 1, "pipe 128x128x64 dbuf remap1 NOEPI");
  RUNR(2, 2, 4, 4, 64, 1, 0, 2, 2, "pipe 128x128x64 dbuf remap2");
  RUNR(2, 2, 4, 4, 64, 1, 1, 2, 2, "pipe 128x128x64 dbuf remap2 NOEPI");
  RUNR(4, 2, 4, 4, 64, 1, 0, 1, 1, "pipe 256x128x64 8w remap1");
  RUNR(4, 2, 4, 4, 64, 1, 1, 1, 1, "pipe 256x128x64 8w remap1 NOEPI");
  RUNR(4, 2, 4, 4, 64, 1, 1, 1, 2, "pipe 256x128x64 8w remap2 NOEPI");
  g.C = nullptr;      // fp16 output only
  { auto kern = pipe_k<2, 2, 4, 4, 64, 1, 0, 2, 0>; const size_t sh = 2 * 256 * 72 * 2;
    float us = timeit([&] { hipLaunchKernelGGL(kern, dim3(N / 128, M / 128), dim3(256), sh, 0, g); });
    g.C = C; hipLaunchKernelGGL(kern, dim3(N / 128, M / 128), dim3(256), sh, 0, g); g.C = nullptr; check("pipe 128x128x64 dbuf fp16-only out", us); }
  { auto kern = pipe_k<4, 2, 4, 4, 64, 1, 0, 1, 0>; const size_t sh = 2 * 384 * 72 * 2;
    float us = timeit([&] { hipLaunchKernelGGL(kern, dim3(N / 128, M / 256), dim3(512), sh, 0, g); });
    g.C = C; hipLaunchKernelGGL(kern, dim3(N / 128, M / 256), dim3(512), sh, 0, g); g.C = nullptr; check("pipe 256x128x64 8w fp16-only out", us); }
  { auto kern = pipe_k<4, 2, 4, 4, 64, 1, 0, 1, 1>; const size_t sh = 2 * 384 * 72 * 2;
    float us = timeit([&] { hipLaunchKernelGGL(kern, dim3(N / 128, M / 256), dim3(512), sh, 0, g); });
    g.C = C; hipLaunchKernelGGL(kern, dim3(N / 128, M / 256), dim3(512), sh, 0, g); g.C = nullptr; check("pipe 256x128x64 8w remap1 fp16-only", us); }

  // ---- weight gradient: dW[Kin][Nout] = X16[B][Kin]^T dZ16[B][Nout]
  {
    const int Bt = 8192, Kin = argc > 4 ? atoi(argv[4]) : 1024, Nout = argc > 5 ? atoi(argv[5]) : 1024;
    printf("dW: batch %d, Kin %d, Nout %d (%.2f GFLOP)\n", Bt, Kin, Nout, 2.0 * Bt * Kin * Nout * 1e-9);
    std::vector<_Float16> hX((size_t)Bt * Kin), hZ((size_t)Bt * Nout);
    for (auto& x : hX) x = (_Float16)rnd(); for (auto& x : hZ) x = (_Float16)(rnd() * 0.1f);
    _Float16 *X, *Z; float *W, *Wr;
    CK(hipMalloc(&X, hX.size() * 2)); CK(hipMalloc(&Z, hZ.size() * 2)); CK(hipMalloc(&W, (size_t)Kin * Nout * 4)); CK(hipMalloc(&Wr, (size_t)Kin * Nout * 4));
    CK(hipMemcpy(X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Z, hZ.data(), hZ.size() * 2, hipMemcpyHostToDevice));
    float* slab; int* cnt; CK(hipMalloc(&slab, (size_t)64 << 20)); CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMemset(cnt, 0, 4096 * 4));
    GT t{X, Kin, Z, Nout, W, Nout, Kin, Nout, Bt, Bt, slab, cnt};
    ref_tn_k<<<dim3((Nout + 255) / 256, Kin), 256>>>(t, Wr); CK(hipDeviceSynchronize());
#define RUNT(WM, WN, TM, TN, BK, NOEPI, MINB, SPLIT, NAME) RUNTR(WM, WN, TM, TN, BK, NOEPI, MINB, SPLIT, 0, NAME)
#define RUNTR(WM, WN, TM, TN, BK, NOEPI, MINB, SPLIT, REMAP, NAME) do { \
      constexpr int BM_ = WM * TM * 16, BN_ = WN * TN * 16; const size_t sh = (size_t)2 * BK * (BM_ + BN_ + 32) * 2; \
      auto kern = tn_k<WM, WN, TM, TN, BK, NOEPI, MINB, REMAP>; \
      CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
      t.kchunk = ((Bt / SPLIT + BK - 1) / BK) * BK; \
      const dim3 grid((Nout + BN_ - 1) / BN_, (Kin + BM_ - 1) / BM_, SPLIT); \
      auto go = [&] { hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), sh, 0, t); \
                      if (NOEPI == 4) hipLaunchKernelGGL((slab_reduce_k<BM_, BN_>), dim3(grid.x * grid.y, 8), dim3(256), 0, 0, t, SPLIT, (int)grid.x); }; \
      float us = timeit(go); CK(hipGetLastError()); \
      CK(hipMemset(W, 0, (size_t)Kin * Nout * 4)); go(); \
      float z = 0.f, d32; CK(hipMemcpy(dmax, &z, 4, hipMemcpyHostToDevice)); \
      diff_k<<<1024, 256>>>(W, Wr, (int64_t)Kin * Nout, dmax); CK(hipMemcpy(&d32, dmax, 4, hipMemcpyDeviceToHost)); \
      printf("%-38s %8.2f us  %7.1f TFLOP/s   maxdiff %.2e%s\n", NAME, us, 2.0 * Bt * Kin * Nout / us * 1e-6, d32, NOEPI == 1 ? " (no epilogue)" : ""); } while (0)
    RUNT(2, 2, 4, 4, 64, 1, 2, 8, "tn 128x128x64 split8 NOEPI");
    RUNTR(2, 2, 4, 4, 64, 1, 2, 8, 1, "tn 128x128x64 split8 NOEPI remap");
    RUNT(2, 2, 4, 4, 64, 2, 2, 8, "tn 128x128x64 split8 slabs");
    RUNTR(2, 2, 4, 4, 64, 2, 2, 8, 1, "tn 128x128x64 split8 slabs remap");
    RUNTR(2, 2, 4, 4, 64, 3, 2, 8, 1, "tn 128x128x64 split8 slabs sc1 remap");
    RUNTR(2, 2, 4, 4, 64, 3, 2, 8, 0, "tn 128x128x64 split8 slabs sc1");
    RUNTR(2, 2, 4, 4, 64, 4, 2, 8, 1, "tn 128x128x64 split8 slabs+reduce remap");
    RUNTR(2, 2, 4, 4, 64, 4, 2, 4, 1, "tn 128x128x64 split4 slabs+reduce remap");
    RUNTR(2, 2, 4, 4, 64, 3, 2, 4, 1, "tn 128x128x64 split4 slabs sc1 remap");
    RUNTR(4, 2, 4, 4, 64, 3, 1, 8, 1, "tn 256x128x64 8w split8 slabs sc1 remap");
    RUNTR(4, 2, 4, 4, 64, 4, 1, 8, 1, "tn 256x128x64 8w split8 slabs+reduce remap");
    RUNTR(2, 2, 4, 4, 64, 3, 2, 16, 1, "tn 128x128x64 split16 slabs sc1 remap");
    RUNTR(2, 2, 4, 4, 64, 4, 2, 16, 1, "tn 128x128x64 split16 slabs+reduce remap");
    RUNTR(4, 2, 4, 4, 64, 1, 1, 8, 1, "tn 256x128x64 8w split8 NOEPI remap");
  }
  return 0;
}
