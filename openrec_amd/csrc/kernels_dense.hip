// Dense-side kernels of the DLRM step (recommenders/dlrm.py:63-100): fp32 GEMM on
// the f32 MFMA (exact fp32: v_mfma_f32_16x16x4_f32 == an fmaf chain), bias/
// activation epilogues, activation backward, column sums, the second-order
// feature interaction (modules/second_order_feature_interaction.py:12-34,
// forward and backward, with the reference's triangle bug as an option), the
// loss (MSE / BCE on probabilities, Keras semantics) and index helpers.
#include <cstring>
#include "orx_csr_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------- GEMM ---
// C[M,N] = act(A[M,K] * B[K,N] + bias[N]);  A(i,k) = A[i*sa0 + k*sa1],
// B(k,j) = B[k*sb0 + j*sb1]  (all three MLP products -- X*W, dY*W^T, X^T*dY --
// are expressed through the strides).  64x64 block tile, 4 wavefronts each
// computing 32x32 as 2x2 MFMA 16x16x4 tiles, K staged through LDS 16 at a time.
struct GemmArgs {
    const float* A; int64_t sa0, sa1;
    const float* B; int64_t sb0, sb1;
    float* C; int64_t ldc;
    const float* bias;
    int M, N, K;
    int act;                 // 0 none, 1 relu, 2 sigmoid
    int kchunk;              // split-K: K range per blockIdx.z; the partial products go to ws [splits][M][N] (plain stores) and
    float* ws;               // splitk_reduce_kernel adds them in split order (no atomics: the sum is reproducible)
    float out_scale;         // the product is multiplied by this (1 / loss scale of the fp16 mode's weight gradients; else 1)
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float As[64][17];
    __shared__ float Bs[16][65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm = blockIdx.y * 64, bn = blockIdx.x * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = tid + 256 * r;
            const int i = idx >> 4, k = idx & 15;
            As[i][k] = (bm + i < g.M && k0 + k < g.K) ? g.A[(int64_t)(bm + i) * g.sa0 + (int64_t)(k0 + k) * g.sa1] : 0.0f;
            const int kb = idx >> 6, j = idx & 63;
            Bs[kb][j] = (k0 + kb < g.K && bn + j < g.N) ? g.B[(int64_t)(k0 + kb) * g.sb0 + (int64_t)(bn + j) * g.sb1] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = As[wm + mi * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = Bs[kk + (lane >> 4)][wn + ni * 16 + (lane & 15)];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm + wm + mi * 16 + (lane >> 4) * 4 + r;
                const int col = bn + wn + ni * 16 + (lane & 15);
                if (row < g.M && col < g.N) {
                    float v = acc[mi][ni][r] * g.out_scale;
                    if (g.bias) v += g.bias[col];
                    if (g.act == 1) v = fmaxf(v, 0.0f);
                    else if (g.act == 2) v = 1.0f / (1.0f + __expf(-v));
                    g.C[(int64_t)row * g.ldc + col] = v;
                }
            }
}

// ---- fp16 MFMA variant (performance mode of the DLRM MLPs) -------------------------------
// Same contract as gemm_f32_kernel, but the operands are rounded to fp16 while they are staged
// into LDS and multiplied with v_mfma_f32_16x16x32_f16 (fp32 accumulate, fp32 output).
// 128x128 block tile, 4 wavefronts x (4x4 tiles of 16x16), K staged 32 at a time.  Both LDS
// tiles are stored K-contiguous ([m][k] and [n][k]) so that every MFMA operand is one 16-byte
// LDS read.  A_KC / B_NC tell which global dimension is contiguous so that the staging loads
// coalesce in all three MLP products (X*W, dY*W^T, X^T*dY).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// 4 consecutive floats starting at p (element i valid iff i < nvalid); one 16-byte load when possible
__device__ __forceinline__ f32x4 load4c(const float* p, int nvalid, bool vec_ok) {
    f32x4 v; v.x = v.y = v.z = v.w = 0.0f;
    if (nvalid >= 4 && vec_ok) return *reinterpret_cast<const f32x4*>(p);
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}

// One 128 x 32 operand tile per block, 16 elements per thread, held in registers between the
// global loads and the LDS stores (so the loads of tile t+1 fly while tile t is multiplied).
//   KC (K contiguous in memory): thread = (row, 16 consecutive k)  -> two 16-byte LDS stores
//   otherwise (row index contiguous): thread = 4 rows x 4 k         -> four 8-byte LDS stores
template <bool KC>
struct TileLoader {
    f32x4 v[4];
    __device__ __forceinline__ void load(const float* P, int64_t s_row, int64_t s_k, int row0, int nrows, int k0, int K, bool vec_ok) {
        const int t = threadIdx.x;
        if (KC) {
            const int r = t >> 1, kb = (t & 1) * 16;
            const bool rok = row0 + r < nrows;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + kb + 4 * q;
                v[q] = load4c(P + (int64_t)(row0 + r) * s_row + k, rok ? K - k : 0, vec_ok);
            }
        } else {
            const int rb = (t & 31) * 4, kb = (t >> 5) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + kb + q;
                v[q] = load4c(P + (int64_t)k * s_k + (row0 + rb), k < K ? nrows - (row0 + rb) : 0, vec_ok);
            }
        }
    }
    // fp32 tile T[row][k] (exact-fp32 kernel)
    __device__ __forceinline__ void storef(float* T, int LD) const {
        const int t = threadIdx.x;
        if (KC) {
            const int r = t >> 1, kb = (t & 1) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(T + r * LD + kb + 4 * q) = v[q];
        } else {
            const int rb = (t & 31) * 4, kb = (t >> 5) * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 w; w.x = v[0][m]; w.y = v[1][m]; w.z = v[2][m]; w.w = v[3][m];
                *reinterpret_cast<f32x4*>(T + (rb + m) * LD + kb) = w;
            }
        }
    }
    __device__ __forceinline__ void store(_Float16* T, int LD) const {
        const int t = threadIdx.x;
        if (KC) {
            const int r = t >> 1, kb = (t & 1) * 16;
            h8 lo, hi;
            lo[0] = (_Float16)v[0].x; lo[1] = (_Float16)v[0].y; lo[2] = (_Float16)v[0].z; lo[3] = (_Float16)v[0].w;
            lo[4] = (_Float16)v[1].x; lo[5] = (_Float16)v[1].y; lo[6] = (_Float16)v[1].z; lo[7] = (_Float16)v[1].w;
            hi[0] = (_Float16)v[2].x; hi[1] = (_Float16)v[2].y; hi[2] = (_Float16)v[2].z; hi[3] = (_Float16)v[2].w;
            hi[4] = (_Float16)v[3].x; hi[5] = (_Float16)v[3].y; hi[6] = (_Float16)v[3].z; hi[7] = (_Float16)v[3].w;
            *reinterpret_cast<h8*>(T + r * LD + kb) = lo;
            *reinterpret_cast<h8*>(T + r * LD + kb + 8) = hi;
        } else {
            const int rb = (t & 31) * 4, kb = (t >> 5) * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                h4 w;
                w[0] = (_Float16)v[0][m]; w[1] = (_Float16)v[1][m]; w[2] = (_Float16)v[2][m]; w[3] = (_Float16)v[3][m];
                *reinterpret_cast<h4*>(T + (rb + m) * LD + kb) = w;
            }
        }
    }
};

template <bool A_KC, bool B_NC>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs g, int vec_a, int vec_b) {
    constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 8;      // 80-byte rows: 16-B aligned, spread over banks
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
    TileLoader<A_KC> la;       // A(i,k): rows = m
    TileLoader<!B_NC> lb;      // B(k,j): rows = n; K contiguous iff NOT n-contiguous
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);                // kchunk is a multiple of BK
    la.load(g.A, g.sa0, g.sa1, bm, g.M, kbeg, kend, vec_a != 0);
    lb.load(g.B, g.sb1, g.sb0, bn, g.N, kbeg, kend, vec_b != 0);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        la.store(As, LD);
        lb.store(Bs, LD);
        __syncthreads();
        if (k0 + BK < kend) {                                  // next tile's loads overlap the MFMAs below
            la.load(g.A, g.sa0, g.sa1, bm, g.M, k0 + BK, kend, vec_a != 0);
            lb.load(g.B, g.sb1, g.sb0, bn, g.N, k0 + BK, kend, vec_b != 0);
        }
        h8 a[4], b[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const h8*>(&As[(wm + mi * 16 + (lane & 15)) * LD + (lane >> 4) * 8]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const h8*>(&Bs[(wn + ni * 16 + (lane & 15)) * LD + (lane >> 4) * 8]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm + wm + mi * 16 + (lane >> 4) * 4 + r;
                const int col = bn + wn + ni * 16 + (lane & 15);
                if (row < g.M && col < g.N) {
                    float v = acc[mi][ni][r] * g.out_scale;
                    if (gridDim.z > 1) { g.ws[((int64_t)blockIdx.z * g.M + row) * g.N + col] = v; continue; }
                    if (g.bias) v += g.bias[col];
                    if (g.act == 1) v = fmaxf(v, 0.0f);
                    else if (g.act == 2) v = 1.0f / (1.0f + __expf(-v));
                    g.C[(int64_t)row * g.ldc + col] = v;
                }
            }
}

// Exact-fp32 product with the structure of gemm_f16_kernel: 128x128 block tile, 4 wavefronts x (4x4 tiles of
// 16x16), K staged 32 at a time through registers into fp32 LDS tiles [row][k], v_mfma_f32_16x16x4_f32.
// K order inside a staged tile: MFMA step s of lane group q = lane >> 4 multiplies physical column 8q + s,
// for A and B alike, so every lane fetches its 8 operands of a row with two 16-byte LDS reads (the sum over
// k is a permutation of the same terms).
template <bool A_KC, bool B_NC>
__global__ __launch_bounds__(256) void gemm_f32v_kernel(GemmArgs g, int vec_a, int vec_b) {
    constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 4;      // 144-byte rows: 16-B aligned, rotate over the banks
    __shared__ __attribute__((aligned(16))) float As[BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
    TileLoader<A_KC> la;
    TileLoader<!B_NC> lb;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    la.load(g.A, g.sa0, g.sa1, bm, g.M, kbeg, kend, vec_a != 0);
    lb.load(g.B, g.sb1, g.sb0, bn, g.N, kbeg, kend, vec_b != 0);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        la.storef(As, LD);
        lb.storef(Bs, LD);
        __syncthreads();
        if (k0 + BK < kend) {
            la.load(g.A, g.sa0, g.sa1, bm, g.M, k0 + BK, kend, vec_a != 0);
            lb.load(g.B, g.sb1, g.sb0, bn, g.N, k0 + BK, kend, vec_b != 0);
        }
        const int q8 = (lane >> 4) * 8;
#pragma unroll
        for (int half = 0; half < 2; ++half) {                   // 4 of the 8 k-steps at a time (register pressure)
            f32x4 a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(&As[(wm + mi * 16 + (lane & 15)) * LD + q8 + 4 * half]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(&Bs[(wn + ni * 16 + (lane & 15)) * LD + q8 + 4 * half]);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][st], b[ni][st], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm + wm + mi * 16 + (lane >> 4) * 4 + r;
                const int col = bn + wn + ni * 16 + (lane & 15);
                if (row < g.M && col < g.N) {
                    float v = acc[mi][ni][r] * g.out_scale;
                    if (gridDim.z > 1) { g.ws[((int64_t)blockIdx.z * g.M + row) * g.N + col] = v; continue; }
                    if (g.bias) v += g.bias[col];
                    if (g.act == 1) v = fmaxf(v, 0.0f);
                    else if (g.act == 2) v = 1.0f / (1.0f + __expf(-v));
                    g.C[(int64_t)row * g.ldc + col] = v;
                }
            }
}

// C[M][N] = sum over the splits of ws [splits][M][N], in split order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, float* C, int64_t ldc, int M, int N, int splits) {
    const int64_t total = (int64_t)M * N, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        float v = 0.0f;
        int z = 0;
        for (; z + 4 <= splits; z += 4) {
            const float w0 = ws[(int64_t)z * total + i], w1 = ws[(int64_t)(z + 1) * total + i], w2 = ws[(int64_t)(z + 2) * total + i], w3 = ws[(int64_t)(z + 3) * total + i];
            v += w0; v += w1; v += w2; v += w3;
        }
        for (; z < splits; ++z) v += ws[(int64_t)z * total + i];
        C[(i / N) * ldc + (i % N)] = v;
    }
}

static int splitk_finish(orx_ctx* ctx, const GemmArgs& g, int splits) {
    int64_t gx = ((int64_t)g.M * g.N + 255) / 256; if (gx > 4096) gx = 4096;
    ORX_LAUNCH(ctx, splitk_reduce_kernel, dim3((unsigned)gx), dim3(256), 0, (const float*)g.ws, g.C, g.ldc, g.M, g.N, splits);
    return ORX_OK;
}

// ---- fp16-resident operands ("shadows") -------------------------------------------------------------
// gemm_f16_kernel reads fp32 operands and is bound by L2 -> CU bandwidth (a 1024x1024 layer at batch 8192 moves
// 512 MB for 17 GFLOP: ~8 TB/s at the measured 66 us).  The forward product X*W and the input-gradient product
// dY*W^T can read fp16 copies that their producers write anyway (the previous layer's epilogue, the activation
// backward, the optimizer): half the bytes, no conversion in the loop.  Both operands K-contiguous:
//   A16 [M][lda] (k contiguous), B16 [N][ldb] (k contiguous), leading dims multiples of 8 halves, zero padded
// to them.  Same rounding point as gemm_f16_kernel (operand -> fp16 once), so the products are identical.
struct Gemm16Args {
    const _Float16* A; int64_t lda;
    const _Float16* B; int64_t ldb;
    float* C; int64_t ldc;
    _Float16* C16; int64_t ldc16;      // optional fp16 copy of the output (next layer's operand)
    const float* bias;
    int M, N, K;
    int act;
    // fused activation backward of the layer BELOW (input-gradient product dX = dZ * W^T): the epilogue turns dX into
    // that layer's dZ = dX * act'(Y), adds its column sums to the bias gradient gb (zero before) -- the separate
    // act_bwd_colsum pass over [batch, width] disappears
    const float* actY; int64_t ldy; int act_y; float* gb;      // gb: partial column sums [row blocks of 128][N] (plain stores)
};

struct TileLoader16 {
    h8 v[2];
    __device__ __forceinline__ void load(const _Float16* P, int64_t ld, int row0, int nrows, int k0) {
        const int t = threadIdx.x;
        const int r = t >> 1, kb = (t & 1) * 16;
        const bool rok = row0 + r < nrows;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + kb + 8 * h;
            h8 z; for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
            v[h] = (rok && k < ld) ? *reinterpret_cast<const h8*>(P + (int64_t)(row0 + r) * ld + k) : z;
        }
    }
    __device__ __forceinline__ void store(_Float16* T, int LD) const {
        const int t = threadIdx.x;
        const int r = t >> 1, kb = (t & 1) * 16;
        *reinterpret_cast<h8*>(T + r * LD + kb) = v[0];
        *reinterpret_cast<h8*>(T + r * LD + kb + 8) = v[1];
    }
};

__global__ __launch_bounds__(256) void gemm_f16s_kernel(Gemm16Args g) {
    constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 8;
    __shared__ __attribute__((aligned(16))) _Float16 As[BM * LD];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[BN * LD];
    __shared__ float colred[2][BN];
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
        la.store(As, LD);
        lb.store(Bs, LD);
        __syncthreads();
        if (k0 + BK < g.K) {
            la.load(g.A, g.lda, bm, g.M, k0 + BK);
            lb.load(g.B, g.ldb, bn, g.N, k0 + BK);
        }
        h8 a[4], b[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const h8*>(&As[(wm + mi * 16 + (lane & 15)) * LD + (lane >> 4) * 8]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const h8*>(&Bs[(wn + ni * 16 + (lane & 15)) * LD + (lane >> 4) * 8]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int col = bn + wn + ni * 16 + (lane & 15);
        float csum = 0.0f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm + wm + mi * 16 + (lane >> 4) * 4 + r;
                if (row < g.M && col < g.N) {
                    float v = acc[mi][ni][r];
                    if (g.bias) v += g.bias[col];
                    if (g.act == 1) v = fmaxf(v, 0.0f);
                    else if (g.act == 2) v = 1.0f / (1.0f + __expf(-v));
                    if (g.actY) {
                        const float y = g.actY[(int64_t)row * g.ldy + col];
                        v = g.act_y == 1 ? (y > 0.0f ? v : 0.0f) : (g.act_y == 2 ? v * y * (1.0f - y) : v);
                        csum += v;
                    }
                    g.C[(int64_t)row * g.ldc + col] = v;
                    if (g.C16) g.C16[(int64_t)row * g.ldc16 + col] = (_Float16)v;
                }
            }
        if (g.gb) {                                      // the 4 lane groups hold different rows of the same column
            csum += __shfl_xor(csum, 16);
            csum += __shfl_xor(csum, 32);
            if (lane < 16) colred[wave >> 1][wn + ni * 16 + lane] = csum;
        }
    }
    if (g.gb) {                                          // the two wavefronts that share the columns, then one plain store per column
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < BN; cidx += 256)
            if (bn + cidx < g.N) g.gb[(int64_t)blockIdx.y * g.N + bn + cidx] = colred[0][cidx] + colred[1][cidx];
    }
}

int orx_launch_gemm_f16s(orx_ctx* ctx, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                         void* C16, int64_t ldc16, const float* bias, int M, int N, int K, int act,
                         const float* actY, int64_t ldy, int act_y, ColPart* gbp) {
    if (M == 0 || N == 0) return ORX_OK;
    float* gb = gbp ? gbp->parts : nullptr;
    if (gbp) gbp->P = (M + 127) / 128;
    ORX_ARG(lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)A16 | (uintptr_t)B16) & 15) == 0, "gemm_f16s: operands need 16-byte rows");
    ProfScope ps(ctx, ORX_K_GEMM);
    Gemm16Args g{(const _Float16*)A16, lda, (const _Float16*)B16, ldb, C, ldc, (_Float16*)C16, ldc16, bias, M, N, K, act, actY, ldy, act_y, gb};
    ORX_LAUNCH(ctx, gemm_f16s_kernel, dim3((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128)), dim3(256), 0, g);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// fp16 copies of the dense kernels after an optimizer step: W16 [in][ld16] (same layout, operand of dY*W^T)
// and W16T [out][ld16t] (transposed, operand of X*W); one launch for all layers (grid.y = layer)
__global__ __launch_bounds__(256) void dense_shadow_kernel(const ShadowParam* ps) {
    const ShadowParam p = ps[blockIdx.y];
    const int64_t n = (int64_t)p.in * p.out;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int i = (int)(e / p.out), j = (int)(e - (int64_t)i * p.out);
        const _Float16 h = (_Float16)p.w[e];
        ((_Float16*)p.w16)[(int64_t)i * p.ld16 + j] = h;
        ((_Float16*)p.w16t)[(int64_t)j * p.ld16t + i] = h;
    }
}

int orx_launch_dense_shadow(orx_ctx* ctx, const ShadowParam* ps_dev, int count, int64_t max_n) {
    if (count == 0) return ORX_OK;
    int64_t gx = (max_n + 255) / 256; if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
    ORX_LAUNCH(ctx, dense_shadow_kernel, dim3((unsigned)gx, (unsigned)count), dim3(256), 0, ps_dev);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// grid, split-K factor and vectorization flags shared by the two 128x128 kernels
static int gemm_plan(const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1, float* C, int64_t ldc,
                     const float* bias, int M, int N, int K, int act, GemmArgs* g, dim3* grid, int* va, int* vb, bool* akc, bool* bnc) {
    const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
    int splits = 1;
    if (bias == nullptr && act == 0 && ldc == N && tiles < 256 && K >= 256) {
        splits = (512 + tiles - 1) / tiles;
        const int max_splits = K >= 4096 ? K / 256 : K / 64;           // small batches: down to two 32-wide iterations per split
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    g->kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
    splits = (K + g->kchunk - 1) / g->kchunk;
    *grid = dim3((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128), (unsigned)splits);
    *akc = sa1 == 1; *bnc = sb1 == 1;
    // 16-byte loads need an aligned base and a leading stride that keeps every row aligned
    *va = (((uintptr_t)A & 15) == 0) && ((*akc ? sa0 : sa1) % 4 == 0);
    *vb = (((uintptr_t)B & 15) == 0) && ((*bnc ? sb0 : sb1) % 4 == 0);
    return splits;
}

int orx_launch_gemm_f16(orx_ctx* ctx, const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1,
                        float* C, int64_t ldc, const float* bias, int M, int N, int K, int act, bool c_zero, float out_scale) {
    if (M == 0 || N == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_GEMM);
    GemmArgs g{A, sa0, sa1, B, sb0, sb1, C, ldc, bias, M, N, K, act, 0, nullptr, out_scale};
    // split-K when the output has too few tiles to fill the chip (the X^T*dY weight-gradient products:
    // small M x N, K = batch); the partial products leave through a workspace and are added in split order
    const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
    int splits = 1;
    if (bias == nullptr && act == 0 && ldc == N && tiles < 256 && K >= 256) {
        splits = (512 + tiles - 1) / tiles;
        const int max_splits = K >= 4096 ? K / 256 : K / 64;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    g.kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
    splits = (K + g.kchunk - 1) / g.kchunk;
    (void)c_zero;
    if (splits > 1) {
        if (orx_ensure((void**)&ctx->d_splitk, &ctx->d_splitk_cap, sizeof(float) * (size_t)splits * M * N) != ORX_OK) return ORX_ERR_OOM;
        g.ws = ctx->d_splitk;
    }
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128), (unsigned)splits);
    const bool akc = sa1 == 1, bnc = sb1 == 1;
    ORX_ARG((akc || sa0 == 1) && (bnc || sb0 == 1), "gemm_f16: every operand needs one unit stride");
    // 16-byte loads need an aligned base and a leading stride that keeps every row aligned
    const int va = (((uintptr_t)A & 15) == 0) && ((akc ? sa0 : sa1) % 4 == 0);
    const int vb = (((uintptr_t)B & 15) == 0) && ((bnc ? sb0 : sb1) % 4 == 0);
    if (akc && bnc) ORX_LAUNCH(ctx, (gemm_f16_kernel<true, true>), grid, dim3(256), 0, g, va, vb);
    else if (akc) ORX_LAUNCH(ctx, (gemm_f16_kernel<true, false>), grid, dim3(256), 0, g, va, vb);
    else if (bnc) ORX_LAUNCH(ctx, (gemm_f16_kernel<false, true>), grid, dim3(256), 0, g, va, vb);
    else ORX_LAUNCH(ctx, (gemm_f16_kernel<false, false>), grid, dim3(256), 0, g, va, vb);
    if (splits > 1) splitk_finish(ctx, g, splits);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_gemm(orx_ctx* ctx, const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1,
                    float* C, int64_t ldc, const float* bias, int M, int N, int K, int act, bool c_zero, float out_scale) {
    if (M == 0 || N == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_GEMM);
    GemmArgs g{A, sa0, sa1, B, sb0, sb1, C, ldc, bias, M, N, K, act, 0, nullptr, out_scale};
    (void)c_zero;
    if ((sa1 == 1 || sa0 == 1) && (sb1 == 1 || sb0 == 1) && getenv("ORX_GEMM_F32_SIMPLE") == nullptr) {
        dim3 grid; int va, vb; bool akc, bnc;
        const int splits = gemm_plan(A, sa0, sa1, B, sb0, sb1, C, ldc, bias, M, N, K, act, &g, &grid, &va, &vb, &akc, &bnc);
        if (splits > 1) {
            if (orx_ensure((void**)&ctx->d_splitk, &ctx->d_splitk_cap, sizeof(float) * (size_t)splits * M * N) != ORX_OK) return ORX_ERR_OOM;
            g.ws = ctx->d_splitk;
        }
        if (akc && bnc) ORX_LAUNCH(ctx, (gemm_f32v_kernel<true, true>), grid, dim3(256), 0, g, va, vb);
        else if (akc) ORX_LAUNCH(ctx, (gemm_f32v_kernel<true, false>), grid, dim3(256), 0, g, va, vb);
        else if (bnc) ORX_LAUNCH(ctx, (gemm_f32v_kernel<false, true>), grid, dim3(256), 0, g, va, vb);
        else ORX_LAUNCH(ctx, (gemm_f32v_kernel<false, false>), grid, dim3(256), 0, g, va, vb);
        if (splits > 1) splitk_finish(ctx, g, splits);
        ORX_HIP(hipGetLastError());
        return ORX_OK;
    }
    ORX_LAUNCH(ctx, gemm_f32_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), dim3(256), 0, g);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// dZ = dY * act'(Y), in place on dY.  Y may be strided (ldy), dY is contiguous [M,N].
__global__ void act_bwd_kernel(float* dY, const float* Y, int64_t ldy, int M, int N, int act) {
    const int64_t total = (int64_t)M * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / N; const int c = (int)(i - r * N);
        const float y = Y[r * ldy + c];
        float d = dY[i];
        if (act == 1) d = y > 0.0f ? d : 0.0f;
        else if (act == 2) d = d * y * (1.0f - y);
        dY[i] = d;
    }
}

int orx_launch_act_bwd(orx_ctx* ctx, float* dY, const float* Y, int64_t ldy, int M, int N, int act) {
    if (act == 0 || M == 0) return ORX_OK;
    int64_t g = ((int64_t)M * N + 255) / 256; if (g > 8192) g = 8192;
    ORX_LAUNCH(ctx, act_bwd_kernel, dim3((unsigned)g), dim3(256), 0, dY, Y, ldy, M, N, act);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// dZ = dY * act'(Y) in place AND the column sums of dZ (the bias gradient) of every row slab in the same pass:
// parts[slab][c], plain stores; colparts_reduce_kernel adds the slabs in order.  grid = (columns / 64, row slabs).
// src != NULL: the incoming gradient is src[r * ld_src + c] * src_scale instead of dY (the bottom MLP's dY is a strided slice of dZ:
// no copy launch of its own)
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(float* dY, const float* Y, int64_t ldy, int M, int N, int act, float* gb,
                                                             _Float16* d16, int64_t ld16, int slab, const float* src, int64_t ld_src, float src_scale) {
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int r0 = blockIdx.y * slab, r1 = min(M, r0 + slab);
    float s = 0.0f;
    if (c < N) {
        for (int r = r0 + part; r < r1; r += 4) {
            const float y = Y[(int64_t)r * ldy + c];
            float d = src ? src[(int64_t)r * ld_src + c] * src_scale : dY[(int64_t)r * N + c];
            if (act == 1) d = y > 0.0f ? d : 0.0f;
            else if (act == 2) d = d * y * (1.0f - y);
            dY[(int64_t)r * N + c] = d;
            if (d16) d16[(int64_t)r * ld16 + c] = (_Float16)d;
            s += d;
        }
    }
    sh[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < N) gb[(int64_t)blockIdx.y * N + c] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

int orx_launch_act_bwd_colsum(orx_ctx* ctx, float* dY, const float* Y, int64_t ldy, int M, int N, int act, ColPart* gbp, void* d16, int64_t ld16,
                              const float* src, int64_t ld_src, float src_scale) {
    if (M == 0 || N == 0) return ORX_OK;
    const int slab = (int64_t)M * N >= (4 << 20) ? 256 : 32;          // small layers: more, shorter slabs (the pass is latency-bound)
    float* gb = gbp->parts;
    gbp->P = (M + slab - 1) / slab;
    ORX_LAUNCH(ctx, act_bwd_colsum_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + slab - 1) / slab)), dim3(256), 0, dY, Y, ldy, M, N, act, gb,
               (_Float16*)d16, ld16, slab, src, ld_src, src_scale);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// out[c] = sum over the P row blocks of parts[p][c], in block order: the second stage of every bias-gradient column sum
// (act_bwd_colsum_kernel, the fused epilogues of the fp16 products, the head kernel).  One launch serves all layers of an
// MLP backward: grid = (columns / 256, jobs).
__global__ __launch_bounds__(256) void colparts_reduce_kernel(ColJobs jobs) {
    // 64 columns per workgroup; the P partial rows in four contiguous segments (one per wavefront, 8 loads in flight), combined in
    // segment order: the summation order is a function of P alone
    __shared__ float seg[4][64];
    const ColJob j = jobs.j[blockIdx.y];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    if (blockIdx.x * 64 >= j.N) return;
    const int per = (j.P + 3) / 4, p0 = w * per, p1 = p0 + per < j.P ? p0 + per : j.P;
    float s = 0.0f;
    if (c < j.N) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = j.parts[(int64_t)(p + u) * j.N + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += a[u];
        }
        for (; p < p1; ++p) s += j.parts[(int64_t)p * j.N + c];
    }
    seg[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < j.N) j.out[c] = (((seg[0][threadIdx.x] + seg[1][threadIdx.x]) + seg[2][threadIdx.x]) + seg[3][threadIdx.x]) * j.scale;
}

int orx_launch_colparts_reduce(orx_ctx* ctx, const ColJob* jobs, int n) {
    for (int base = 0; base < n; base += ORX_COLJOBS_MAX) {
        ColJobs cj;
        const int cnt = n - base < ORX_COLJOBS_MAX ? n - base : ORX_COLJOBS_MAX;
        int maxN = 1;
        for (int i = 0; i < cnt; ++i) { cj.j[i] = jobs[base + i]; if (cj.j[i].N > maxN) maxN = cj.j[i].N; }
        ORX_LAUNCH(ctx, colparts_reduce_kernel, dim3((unsigned)((maxN + 63) / 64), (unsigned)cnt), dim3(256), 0, cj);
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// strided copy: dst[r, 0:n] = src[r*lds + 0:n]
__global__ void copy2d_kernel(float* dst, int64_t ldd, const float* src, int64_t lds_, int M, int N, float scale) {
    const int64_t total = (int64_t)M * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / N; const int c = (int)(i - r * N);
        dst[r * ldd + c] = src[r * lds_ + c] * scale;
    }
}

int orx_launch_copy2d(orx_ctx* ctx, float* dst, int64_t ldd, const float* src, int64_t lds_, int M, int N, float scale) {
    if (M == 0 || N == 0) return ORX_OK;
    int64_t g = ((int64_t)M * N + 255) / 256; if (g > 8192) g = 8192;
    ORX_LAUNCH(ctx, copy2d_kernel, dim3((unsigned)g), dim3(256), 0, dst, ldd, src, lds_, M, N, scale);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ------------------------------------------------------- feature interaction ---
// Z [B, F, d] (slot F-1 = bottom-MLP output).  R [B, d + P]: R[:, 0:d] = Z[:, F-1, :],
// R[:, d + k] = k-th selected element of Z Z^T in tf.boolean_mask (row-major) order.
//   compat = 1 : lower triangle kept, (strictly) UPPER triangle selected -> zeros,
//                only the diagonal survives when `itself`  (the reference's behaviour)
//   compat = 0 : (strictly) lower triangle selected, values z_i . z_j
__device__ __forceinline__ bool pair_selected(int i, int j, int compat, int itself) {
    if (compat) return itself ? (j >= i) : (j > i);
    return itself ? (j <= i) : (j < i);
}

__global__ __launch_bounds__(256) void interact_fwd_kernel(const float* Z, int F, int d, int compat, int itself,
                                                           float* R, int P, int64_t B, int ldR) {
    extern __shared__ float zs[];            // [F][d + 1]: the +1 keeps rows i and j off the same LDS bank
    const int zd = d + 1;
    const int64_t b = blockIdx.x;
    if (b >= B) return;
    const float* zb = Z + b * F * d;
    for (int k = threadIdx.x; k < F * d; k += 256) zs[(k / d) * zd + k % d] = zb[k];
    __syncthreads();
    float* rb = R + b * ldR;
    for (int k = threadIdx.x; k < d; k += 256) rb[k] = zs[(F - 1) * zd + k];
    for (int e = threadIdx.x; e < F * F; e += 256) {
        const int i = e / F, j = e % F;
        if (!pair_selected(i, j, compat, itself)) continue;
        // rank of (i, j) among the selected elements in row-major order
        int rank;
        if (compat) rank = itself ? (i * F - i * (i - 1) / 2 + (j - i)) : (i * F - i * (i + 1) / 2 + (j - i - 1));
        else rank = itself ? (i * (i + 1) / 2 + j) : (i * (i - 1) / 2 + j);
        float v = 0.0f;
        if (!compat || i == j) { for (int k = 0; k < d; ++k) v += zs[i * zd + k] * zs[j * zd + k]; }
        rb[d + rank] = v;
    }
}

// dZ [B, F, d] from dR [B, d + P]
__global__ __launch_bounds__(256) void interact_bwd_kernel(const float* Z, const float* dR, int F, int d, int compat,
                                                           int itself, float* dZ, int P, int64_t B, int ldR, float scale) {
    extern __shared__ float sm[];            // zs [F][d + 1], gs [F][F]
    const int zd = d + 1;
    float* zs = sm;
    float* gs = sm + F * zd;
    const int64_t b = blockIdx.x;
    if (b >= B) return;
    const float* zb = Z + b * F * d;
    const float* rb = dR + b * ldR;
    for (int k = threadIdx.x; k < F * d; k += 256) zs[(k / d) * zd + k % d] = zb[k];
    for (int e = threadIdx.x; e < F * F; e += 256) {
        const int i = e / F, j = e % F;
        float g = 0.0f;
        if (pair_selected(i, j, compat, itself) && (!compat || i == j)) {
            int rank;
            if (compat) rank = itself ? (i * F - i * (i - 1) / 2 + (j - i)) : (i * F - i * (i + 1) / 2 + (j - i - 1));
            else rank = itself ? (i * (i + 1) / 2 + j) : (i * (i - 1) / 2 + j);
            g = rb[d + rank];
        }
        gs[e] = g;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < F * d; e += 256) {
        const int i = e / d, k = e % d;
        float acc = (i == F - 1) ? rb[k] : 0.0f;
        for (int j = 0; j < F; ++j) acc += (gs[i * F + j] + gs[j * F + i]) * zs[j * zd + k];  // d(z_i.z_j): both orders
        dZ[b * F * d + e] = acc * scale;
    }
}

// ---- MFMA versions (compat = 0, F <= 32, d % 32 == 0): one wavefront per sample, exact fp32 --------------
// forward: G = Z Z^T on the lower triangle of 16x16 tiles (v_mfma_f32_16x16x4_f32; the B operand of tile
// (ti, tj) is the A operand of row tile tj: both are "lane (i, q) holds Z[16 t + i][k]").  K order inside a
// 32-wide block: step s of lane group q multiplies column 8q + s (two 16-byte loads per row and block).
// Where the feature rows of a sample come from: the gathered copy Z [B][F][d], or -- `emb` set -- straight from the combined
// embedding table through the step's row ids (slot F-1, the bottom MLP's output, always lives in Z).  Reading the table
// directly removes the gather pass: 109 MB written and read again per step at the C5 shapes.
struct RowSrc {
    const float* Z; const float* emb; const int32_t* idx; int64_t rows; int* err;
    float* gdst;       // backward: the gradient of embedding slot f of sample b goes to row idx[b F + f] of this array instead of dZ
                       // (hybrid-parallel step: straight into the buffer that travels back to the rows' owners); NULL: dZ
    int xcd;           // workgroup -> samples the way the fp16 products map tiles (orx_device.h xcd_slot): XCD x takes one contiguous eighth of
                       // the batch, the eighth whose rows of R the top MLP's first product reads on the same XCD (and whose bottom-MLP
                       // outputs were written there).  Measured (round 6, scripts/gpu_r6_u.sh): the C5 step 0.4364 against 0.4384 ms with the
                       // round-robin order (ORX_INTERACT_NO_XCD=1), three alternating runs each; no launch of the trace moves by more than 1 %
    __device__ __forceinline__ int64_t first_sample() const {
        return (int64_t)(xcd ? xcd_slot((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * 4 + (threadIdx.x >> 6);
    }
    __device__ __forceinline__ const float* row(int64_t b, int f, int F, int d) const {
        if (emb == nullptr || f == F - 1) return Z + (b * F + f) * d;
        const int r = idx[b * F + f];
        if ((uint32_t)r >= (uint64_t)rows) { *err = 1; return nullptr; }
        return emb + (size_t)r * d;
    }
};

__global__ __launch_bounds__(256) void interact_fwd_mfma_kernel(RowSrc src, int F, int d, int itself, float* R, int64_t B, int ldR,
                                                                _Float16* R16, int ldR16) {
    const int lane = threadIdx.x & 63;
    const int64_t b = src.first_sample();
    if (b >= B) return;
    const int i = lane & 15, q = lane >> 4;
    const float* z0 = i < F ? src.row(b, i, F, d) : nullptr;
    const float* z1 = 16 + i < F ? src.row(b, 16 + i, F, d) : nullptr;
    const float* zlast = src.row(b, F - 1, F, d);
    f32x4 a00, a10, a11;
    a00.x = a00.y = a00.z = a00.w = 0.0f; a10 = a00; a11 = a00;
    for (int kb = 0; kb < d; kb += 32) {
        f32x4 t0[2], t1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 z; z.x = z.y = z.z = z.w = 0.0f;
            t0[h] = z0 ? *reinterpret_cast<const f32x4*>(z0 + kb + 8 * q + 4 * h) : z;
            t1[h] = z1 ? *reinterpret_cast<const f32x4*>(z1 + kb + 8 * q + 4 * h) : z;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                a00 = __builtin_amdgcn_mfma_f32_16x16x4f32(t0[h][st], t0[h][st], a00, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1[h][st], t0[h][st], a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1[h][st], t1[h][st], a11, 0, 0, 0);
            }
    }
    // the sample's output row is put together in LDS and leaves in whole 16-byte pieces (the pair products land at triangular
    // offsets: written straight to memory they are 24 wavefront stores of 4 scattered bytes per lane, twice with the fp16 copy)
    extern __shared__ __attribute__((aligned(16))) float r_lds[];
    const int rowlen = ldR > ldR16 ? ldR : ldR16;        // (R16 == NULL: ldR16 = 0)
    float* rs = r_lds + (threadIdx.x >> 6) * rowlen;
    const int P = itself ? F * (F + 1) / 2 : F * (F - 1) / 2;
    for (int k = lane; k < d; k += 64) rs[k] = zlast[k];
    for (int k = d + P + lane; k < rowlen; k += 64) rs[k] = 0.0f;      // padding columns
    auto emit = [&](const f32x4& acc, int r0, int c0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = r0 + q * 4 + r, gj = c0 + i;
            if (gi < F && (itself ? gj <= gi : gj < gi)) rs[d + (itself ? gi * (gi + 1) / 2 + gj : gi * (gi - 1) / 2 + gj)] = acc[r];
        }
    };
    emit(a00, 0, 0); emit(a10, 16, 0); emit(a11, 16, 16);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* rb = R + b * ldR;
    for (int k = lane * 4; k < ldR; k += 256) *reinterpret_cast<f32x4*>(rb + k) = *reinterpret_cast<const f32x4*>(rs + k);
    if (R16) {
        _Float16* rh = R16 + b * ldR16;
        for (int k = lane * 8; k < ldR16; k += 512) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(rs + k), v1 = *reinterpret_cast<const f32x4*>(rs + k + 4);
            typedef _Float16 h8v __attribute__((ext_vector_type(8)));
            h8v o;
            o[0] = (_Float16)v0.x; o[1] = (_Float16)v0.y; o[2] = (_Float16)v0.z; o[3] = (_Float16)v0.w;
            o[4] = (_Float16)v1.x; o[5] = (_Float16)v1.y; o[6] = (_Float16)v1.z; o[7] = (_Float16)v1.w;
            *reinterpret_cast<h8v*>(rh + k) = o;
        }
    }
}

// backward: dZ = (Gs + Gs^T) Z (+ dR[0:d] on the dense slot), Gs = the pair gradients scattered back to [F][F].
// A operand: lane (i, q) holds S[16 ti + i][4 s + q], built straight from dR.  B operand: lane (j, q) owns the
// CPL = d/16 consecutive columns CPL*j .. CPL*j + CPL-1 of Z row 4 s + q (float4 loads); column tile t of the
// MFMA grid is {CPL*j + t}, so every lane ends up with CPL consecutive columns of its output rows (float4 stores).
// SPLIT wavefronts may share a sample, each taking d / SPLIT of the columns.  Measured at d = 128, B = 8192: two wavefronts of 64
// columns (16 384 wavefronts of ~90 VGPRs) are SLOWER than one of 128 (8192 of ~170): step 0.6285 against 0.6125 ms -- every
// wavefront re-reads the sample's dR and the 256-byte half rows lose the 512-byte bursts.  No launch uses SPLIT > 1.
// (round 4: NSAMP samples per wavefront in turn, the next sample's rows requested before the current sample's products -- 2 samples:
// no change, 0.5547 against 0.5546 ms per step; 4 / 8 samples: +2 % / +9 %, too few wavefronts.  The launch is not the sum of one
// wavefront's phases; the random 512-byte rows set its pace.)
// FUSE (round 6; north_star's "one pass": dlrm.py:83-85 + second_order_feature_interaction.py:20-32 + tf2_examples/dlrm_criteo.py:45-46):
// the sparse SGD / Adagrad rule is applied HERE to every embedding row that no other lookup of the step references (`single`, made
// from the step's sorted id list: orx_rows_single_flags) -- the wavefront that forms the row's gradient is the only reader and
// the only writer of the row in this step, so w - lr * g goes straight into the table and the gradient never reaches HBM (no dZ row
// written, none read back by the sorted apply, whose table-row read goes too).  Rows with several references keep the
// deterministic sorted path (csr_apply_kernel with skip_single).
struct FusedRows {
    const unsigned char* single;       // [B][F] flags
    float* W; float* A;                // the table (= src.emb) and the Adagrad accumulators
    int kind; float lr, eps;           // ORX_SGD | ORX_ADAGRAD
};

// OCC: wavefronts per SIMD the register allocation is held to (d = 128: 138 registers and 3 wavefronts without a bound, 118 and 4 with it,
// no spills; 5 spills).  0: no bound.
template <int CPL, int SPLIT, bool FUSE, int OCC = 0>
__global__ __launch_bounds__(256, OCC > 0 ? OCC : 1) void interact_bwd_mfma_kernel(RowSrc src, const float* dR, int F, int itself, float* dZ,
                                                                int64_t B, int ldR, float scale, FusedRows fr) {
    constexpr int d = 16 * CPL * SPLIT;
    const int lane = threadIdx.x & 63;
    const int64_t wv = src.first_sample();
    const int64_t b = wv / SPLIT;
    const int coff = (int)(wv % SPLIT) * 16 * CPL;          // first column of this wavefront's share
    if (b >= B) return;
    const int i = lane & 15, q = lane >> 4;
    // (round 6: the sample's rows are requested BEFORE its dR row goes through LDS -- behind the wavefront fence below the compiler could not hoist them, and the
    // two round trips were taken one after the other: 68.8-69.2 -> 67.2 us at the C5 shapes)
    // (tried: lane i taking columns (t / 4) * 64 + 4 i + t % 4, so that a 16-byte access of sixteen lanes covers 256 contiguous bytes instead
    // of every other 16 bytes of 512 -- 72.1 against 68.8 us, no gain)
    float zr[8][CPL];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k = 4 * s + q;
#pragma unroll
        for (int c4 = 0; c4 < CPL; c4 += 4) {
            f32x4 v; v.x = v.y = v.z = v.w = 0.0f;
            const float* zk = k < F ? src.row(b, k, F, d) : nullptr;
            if (zk) {
                if (CPL >= 4) v = *reinterpret_cast<const f32x4*>(zk + coff + CPL * i + c4);
                else { v.x = zk[coff + CPL * i]; v.y = zk[coff + CPL * i + 1]; }
            }
            zr[s][c4] = v.x; if (c4 + 1 < CPL) zr[s][c4 + 1] = v.y;
            if (c4 + 2 < CPL) zr[s][c4 + 2] = v.z; if (c4 + 3 < CPL) zr[s][c4 + 3] = v.w;
        }
    }
    // the sample's dR row goes through LDS: two coalesced 16-byte loads per lane instead of the sixteen 4-byte gathers that build
    // the A operand (each a wavefront instruction with 64 different addresses)
    extern __shared__ __attribute__((aligned(16))) float dr_lds[];
    float* rb = dr_lds + (threadIdx.x >> 6) * ldR;
    {
        const float* rg = dR + b * ldR;                  // (ldR is a multiple of 4 floats, the rows are 16-byte aligned)
        for (int k = lane * 4; k < ldR; k += 256) *reinterpret_cast<f32x4*>(rb + k) = *reinterpret_cast<const f32x4*>(rg + k);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    auto sval = [&](int r, int c) -> float {             // (Gs + Gs^T)[r][c]
        if (r >= F || c >= F) return 0.0f;
        if (r == c) return itself ? 2.0f * rb[d + r * (r + 1) / 2 + r] : 0.0f;
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        return rb[d + (itself ? hi * (hi + 1) / 2 + lo : hi * (hi - 1) / 2 + lo)];
    };
    float sa[2][8];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int s = 0; s < 8; ++s) sa[ti][s] = sval(16 * ti + i, 4 * s + q);
    // FUSE: which of the sample's lookups are applied here (bit g of the wavefront-uniform mask) and their rows
    unsigned fmask = 0u; int myrow = -1;
    if (FUSE) {
        if (lane < F - 1) {
            myrow = src.idx[b * F + lane];
            if (fr.single[b * F + lane] && (uint32_t)myrow < (uint64_t)src.rows) fmask = 1u;
        }
        fmask = (unsigned)__ballot(fmask != 0u);
    }
    f32x4 acc[2][CPL];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int t = 0; t < CPL; ++t) { acc[ti][t].x = acc[ti][t].y = acc[ti][t].z = acc[ti][t].w = 0.0f; }
#pragma unroll
    for (int t = 0; t < CPL; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][s], zr[s][t], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][s], zr[s][t], acc[1][t], 0, 0, 0);
        }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = 16 * ti + q * 4 + r;
            const int rowg = FUSE ? __shfl(myrow, g) : 0;                // (all lanes still active here: lane g is a valid source)
            if (g >= F) continue;
            float* out = dZ + (b * F + g) * d + coff + CPL * i;
            if (src.gdst != nullptr && g != F - 1) {
                const int r = src.idx[b * F + g];
                if ((uint32_t)r >= (uint64_t)src.rows) continue;         // (reported by the forward pass)
                out = src.gdst + (size_t)r * d + coff + CPL * i;
            }
            float o[CPL];
#pragma unroll
            for (int t = 0; t < CPL; ++t) o[t] = (acc[ti][t][r] + (g == F - 1 ? rb[coff + CPL * i + t] : 0.0f)) * scale;
            if (FUSE && CPL >= 4 && ((fmask >> g) & 1u)) {
                // the row is this wavefront's alone: its CPL columns come back from L2 (the B operand above holds the same row in
                // another lane), the rule of csr_rule (kernels_rowsort.hip) is applied, the new row is written in place
                const size_t at = (size_t)rowg * d + coff + CPL * i;
                float* wp = fr.W + at;
#pragma unroll
                for (int t = 0; t < CPL; t += 4) {
                    f32x4 w = *reinterpret_cast<const f32x4*>(wp + t);
                    if (fr.kind == ORX_ADAGRAD) {
                        f32x4 av = *reinterpret_cast<const f32x4*>(fr.A + at + t);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a2 = av[e] + o[t + e] * o[t + e];
                            av[e] = a2;
                            w[e] = w[e] - fr.lr * o[t + e] / (sqrtf(a2) + fr.eps);
                        }
                        *reinterpret_cast<f32x4*>(fr.A + at + t) = av;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = w[e] - fr.lr * o[t + e];
                    }
                    *reinterpret_cast<f32x4*>(wp + t) = w;
                }
                continue;
            }
            if (CPL >= 4) {
#pragma unroll
                for (int t = 0; t < CPL; t += 4) { f32x4 v; v.x = o[t]; v.y = o[t + 1]; v.z = o[t + 2]; v.w = o[t + 3]; *reinterpret_cast<f32x4*>(out + t) = v; }
            } else {
                out[0] = o[0]; out[1] = o[1];
            }
        }
}

// can the interaction read its embedding rows straight from the table (both passes on the MFMA kernels)?
bool orx_interact_direct_ok(int F, int d, int compat) {
    return !compat && F <= 32 && (d == 32 || d == 64 || d == 128 || d == 256) && getenv("ORX_INTERACT_SIMPLE") == nullptr &&
           getenv("ORX_DLRM_NO_DIRECT") == nullptr;
}

// can orx_launch_interact's backward apply the rows referenced once in place (see FusedRows)?
bool orx_interact_fuse_ok(int F, int d, int compat) {
    return orx_interact_direct_ok(F, d, compat) && d >= 64 && getenv("ORX_DLRM_NO_FUSED_SPARSE") == nullptr;
}

int orx_launch_interact(orx_ctx* ctx, bool fwd, const float* Z, const float* dR, int F, int d, int compat, int itself,
                        float* out, int P, int64_t B, int ldR, void* R16, int ldR16, bool* wrote16,
                        const float* emb, const int32_t* idx, int64_t emb_rows, float scale, float* gdst,
                        const unsigned char* single, int opt_kind, float lr, float eps, float* acc_rows) {
    if (wrote16) *wrote16 = false;
    if (B == 0) return ORX_OK;
    const bool mfma = !compat && F <= 32 && d % 32 == 0 && ((uintptr_t)Z & 15) == 0 && getenv("ORX_INTERACT_SIMPLE") == nullptr;
    const bool bwd_ok = d == 32 || d == 64 || d == 128 || d == 256;
    ORX_ARG(emb == nullptr || (mfma && bwd_ok), "interact: direct table rows need the MFMA kernels");
    static const bool no_xcd = getenv("ORX_INTERACT_NO_XCD") != nullptr;
    const RowSrc src{Z, emb, idx, emb_rows, ctx->d_err, fwd ? nullptr : gdst, (!no_xcd && ((B + 3) / 4) % 8 == 0) ? 1 : 0};
    ORX_ARG(gdst == nullptr || (emb != nullptr && mfma && bwd_ok), "interact: a gradient destination needs the direct rows");
    if (mfma && (fwd || bwd_ok)) {
        const dim3 g((unsigned)((B + 3) / 4));
        if (fwd) {
            ORX_ARG(ldR % 4 == 0 && (R16 == nullptr || ldR16 % 8 == 0), "interact: rows of R need 16-byte strides");
            ORX_LAUNCH(ctx, interact_fwd_mfma_kernel, g, dim3(256), (size_t)4 * std::max(ldR, R16 ? ldR16 : 0) * sizeof(float), src, F, d, itself, out, B, ldR, (_Float16*)R16, R16 ? ldR16 : 0);
            if (wrote16 && R16) *wrote16 = true;
        }
        else {
            FusedRows fr{single, const_cast<float*>(emb), acc_rows, opt_kind, lr, eps};
            const bool fuse = single != nullptr;
            ORX_ARG(!fuse || (emb != nullptr && gdst == nullptr && d >= 64 && (opt_kind == ORX_SGD || (opt_kind == ORX_ADAGRAD && acc_rows != nullptr))),
                    "interact: the in-place apply needs direct table rows, d >= 64 and SGD / Adagrad");
            const size_t shm = (size_t)4 * ldR * sizeof(float);
            static const bool occ4 = getenv("ORX_INTERACT_BWD_OCC") == nullptr || atoi(getenv("ORX_INTERACT_BWD_OCC")) == 4;
            if (d == 32) ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<2, 1, false>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
            else if (d == 64 && fuse) ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<4, 1, true>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
            else if (d == 64) ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<4, 1, false>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
            else if (d == 128 && fuse && occ4) ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<8, 1, true, 4>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
            else if (d == 128 && occ4) ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<8, 1, false, 4>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
            else if (d == 128 && fuse) ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<8, 1, true>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
            else if (d == 128) ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<8, 1, false>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
            else if (fuse) ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<16, 1, true>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
            else ORX_LAUNCH(ctx, (interact_bwd_mfma_kernel<16, 1, false>), g, dim3(256), shm, src, dR, F, itself, out, B, ldR, scale, fr);
        }
        ORX_HIP(hipGetLastError());
        return ORX_OK;
    }
    if (fwd) ORX_LAUNCH(ctx, interact_fwd_kernel, dim3((unsigned)B), dim3(256), (size_t)F * (d + 1) * sizeof(float), Z, F, d, compat, itself, out, P, B, ldR);
    else ORX_LAUNCH(ctx, interact_bwd_kernel, dim3((unsigned)B), dim3(256), (size_t)(F * (d + 1) + F * F) * sizeof(float), Z, dR, F, d, compat, itself, out, P, B, ldR, scale);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ------------------------------------------------------------------- loss ---
// pred [B] = (clipped) top-MLP output; writes loss (fp64) and dP [B] = dLoss/d(pre-clip p).
// n_mean: the batch the loss mean runs over (= B, or the global batch of a data-parallel step);
// accumulate: add this launch's share to *loss_out instead of overwriting it
__global__ __launch_bounds__(1024) void dlrm_loss_kernel(float* P, const float* y, int64_t B, int bce, float thr,
                                                         float* dP, double* loss_out, int64_t n_mean, int accumulate, float gscale) {
    __shared__ double sh[16];
    double s = 0.0;
    const float invB = 1.0f / (float)n_mean;
    for (int64_t i = threadIdx.x; i < B; i += 1024) {
        float p = P[i];
        float mask = 1.0f;
        if (thr > 0.0f && thr < 1.0f) {                         // dlrm.py:97-98
            mask = (p >= thr && p <= 1.0f - thr) ? 1.0f : 0.0f;
            p = fminf(fmaxf(p, thr), 1.0f - thr);
            P[i] = p;
        }
        const float t = y[i];
        float g;
        if (!bce) {                                             // keras.losses.MeanSquaredError
            const float r = t - p;
            s += (double)(r * r);
            g = 2.0f * (p - t) * invB;
        } else {                                                // keras.losses.BinaryCrossentropy (probabilities)
            const float eps = 1e-7f;
            const float pc = fminf(fmaxf(p, eps), 1.0f - eps);
            s += -(double)(t * logf(pc + eps) + (1.0f - t) * logf(1.0f - pc + eps));
            const float inside = (p >= eps && p <= 1.0f - eps) ? 1.0f : 0.0f;
            g = -(t / (pc + eps) - (1.0f - t) / (1.0f - pc + eps)) * inside * invB;
        }
        if (dP) dP[i] = g * mask * gscale;           // (gscale: the fp16 mode's loss scale, a power of two)
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += sh[k];
        if (accumulate) loss_out[0] += t / (double)n_mean;
        else loss_out[0] = t / (double)n_mean;
    }
}

int orx_launch_dlrm_loss(orx_ctx* ctx, float* P, const float* y, int64_t B, int bce, float thr, float* dP, double* loss_out,
                         int64_t n_mean, int accumulate, float gscale) {
    ORX_LAUNCH(ctx, dlrm_loss_kernel, dim3(1), dim3(1024), 0, P, y, B, bce, thr, dP, loss_out, n_mean > 0 ? n_mean : B, accumulate, gscale);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ---------------------------------------------------------------- id helper ---
// Row ids into the COMBINED embedding table: idx[b*F + f] = offset[f] + sparse[b, f]
// (f < F-1), -1 for the dense slot f = F-1 and for out-of-range ids (flagged).
__global__ void dlrm_ids_kernel(const int32_t* sparse, const int64_t* offset, const int64_t* rows, int nf, int64_t B,
                                int32_t* idx, int* err) {
    const int F = nf + 1;
    const int64_t total = B * F;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t b = i / F; const int f = (int)(i - b * F);
        int32_t v = -1;
        if (f < nf) {
            const int s = sparse[b * nf + f];
            if ((uint32_t)s >= (uint64_t)rows[f]) *err = 1; else v = (int32_t)(offset[f] + s);
        }
        idx[i] = v;
    }
}

int orx_launch_dlrm_ids(orx_ctx* ctx, const int32_t* sparse, const int64_t* offset, const int64_t* rows, int nf, int64_t B,
                        int32_t* idx) {
    int64_t g = (B * (nf + 1) + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
    ORX_LAUNCH(ctx, dlrm_ids_kernel, dim3((unsigned)g), dim3(256), 0, sparse, offset, rows, nf, B, idx, ctx->d_err);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// sparse gradient rows -> gsum (Adam on sharded/DLRM tables: accumulate, the dense sweep follows)
__global__ void rows_accum_kernel(float* G, const int32_t* ids, const float* grads, int64_t g_stride, int64_t n, int D, int64_t rows, int* err) {
    const int64_t total = n * D;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t k = i / D; const int e = (int)(i - k * D);
        const int r = ids[k];
        if (r < 0) continue;
        if ((int64_t)r >= rows) { *err = 1; continue; }
        unsafeAtomicAdd(G + (size_t)r * D + e, grads[k * g_stride + e]);
    }
}

int orx_launch_rows_accum(orx_ctx* ctx, float* G, const int32_t* ids, const float* grads, int64_t g_stride, int64_t n, int D, int64_t rows) {
    if (n == 0) return ORX_OK;
    int64_t g = (n * D + 255) / 256; if (g > 65536) g = 65536;
    ORX_LAUNCH(ctx, rows_accum_kernel, dim3((unsigned)g), dim3(256), 0, G, ids, grads, g_stride, n, D, rows, ctx->d_err);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ------------------------------------------------------- multi-tensor dense apply ---
// The SGD / Adagrad / Adam rule of every dense parameter of a model in ONE launch (16 launches of a few microseconds
// each otherwise): grid.y = parameter, grid.x covers the largest one.  Gradients are zeroed behind.
__global__ __launch_bounds__(256) void dense_apply_multi_kernel(const DenseParam* ps, int optkind, float lr, float eps, float b1, float b2) {
    const DenseParam p = ps[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
        const float gi = p.g[i];
        p.g[i] = 0.0f;
        if (optkind == ORX_ADAM) {          // lr = lr_t of this step; acc = m, acc2 = v
            float w = p.w[i], m = p.acc[i], v = p.acc2[i];
            adam_elem(w, m, v, gi, lr, b1, b2, eps);
            p.w[i] = w; p.acc[i] = m; p.acc2[i] = v;
        } else if (optkind == ORX_ADAGRAD) {
            const float a2 = p.acc[i] + gi * gi;
            p.acc[i] = a2;
            p.w[i] = p.w[i] - lr * gi / (sqrtf(a2) + eps);
        } else {
            p.w[i] = p.w[i] - lr * gi;
        }
    }
}

int orx_launch_dense_apply_multi(orx_ctx* ctx, const DenseParam* ps_dev, int count, int64_t max_n, int optkind, float lr, float eps,
                                 float b1, float b2) {
    if (count == 0) return ORX_OK;
    int64_t gx = (max_n + 255) / 256; if (gx > 1024) gx = 1024; if (gx < 1) gx = 1;
    ORX_LAUNCH(ctx, dense_apply_multi_kernel, dim3((unsigned)gx, (unsigned)count), dim3(256), 0, ps_dev, optkind, lr, eps, b1, b2);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// One 16 x 64 tile of one dense parameter per workgroup (thread = 4 consecutive columns of one row; ~2400 workgroups for the C5 MLPs:
// 64 x 64 tiles gave 600 and a launch of 96 us, latency-bound): gradient = g + slab_scale * (sum of the S split-K slices, in slice
// order, 8 in flight), the optimizer rule, g re-zeroed, and the fp16 copies of the new weights -- w16 row-major straight from the
// registers, w16t through an LDS transpose.
constexpr int DF_ROWS = 16;
// (round 6) The first `fin.blocks` workgroups of the launch are not tiles of a dense parameter: they finish the embedding table's sorted apply -- the
// runs of equal rows that cross a 64-entry block, csr_finish_block (orx_csr_device.h), until now a launch of its own between csr_apply_kernel and
// this one (6.8 us + a launch gap of the C5 step for a few hundred wavefronts with work).  Neither side reads what the other writes; first in the
// grid, their short dependent chains run under the whole launch.
template <int MODE>
__device__ __forceinline__ void csr_finish_any(const CsrArgs& a, int64_t b, int lane) {
    switch ((a.D + 63) / 64) {
        case 1: csr_finish_block<1, MODE>(a, b, lane); break;
        case 2: csr_finish_block<2, MODE>(a, b, lane); break;
        case 3: csr_finish_block<3, MODE>(a, b, lane); break;
        default: csr_finish_block<4, MODE>(a, b, lane); break;
    }
}

__global__ __launch_bounds__(256) void dense_apply_fused_kernel(const DenseFused* ps, DenseFusedTiles tt, int optkind, float lr, float eps, float b1, float b2,
                                                                float slab_scale, CsrFinish fin) {
    if ((int)blockIdx.x < fin.blocks) {
        const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (fin.mode == CSR_SGD) csr_finish_any<CSR_SGD>(fin.a, b, threadIdx.x & 63);
        else csr_finish_any<CSR_ADAGRAD>(fin.a, b, threadIdx.x & 63);
        return;
    }
    const int bx = (int)blockIdx.x - fin.blocks;
    __shared__ _Float16 tr[DF_ROWS][64 + 2];
    __shared__ float cseg[4][64];
    int pi = 0;                                              // (the tile table travels in the kernel arguments: scalar loads, no dependent memory chain)
    while (pi + 1 < tt.count && bx >= tt.tile0[pi + 1]) ++pi;
    const DenseFused p = ps[pi];
    const int t = bx - p.tile0;
    const int r0 = (t / p.tiles_x) * DF_ROWS, c0 = (t % p.tiles_x) * 64;
    // a vector parameter's gradient in partial rows (bias gradients, the 1-unit head's weight gradient): the tile's <= 64 elements are
    // summed exactly as colparts_reduce_kernel does it -- wavefront w takes the w-th quarter of the P rows in order, 8 loads in
    // flight, the quarters are combined in order -- by all 256 threads of the workgroup, before the (few) owning threads go on
    const int cP = p.cparts != nullptr ? tt.cP[pi] : 0;
    const int cbase = p.rows == 1 ? c0 : r0;                // (rows == 1: element = column; cols == 1: element = row)
    if (cP > 0) {
        const int k = threadIdx.x & 63, w = threadIdx.x >> 6, c = cbase + k;
        const int per = (cP + 3) / 4, p0 = w * per, p1 = p0 + per < cP ? p0 + per : cP;
        float sacc = 0.0f;
        if (c < p.cN) {
            int q = p0;
            for (; q + 8 <= p1; q += 8) {
                float a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = p.cparts[(int64_t)(q + u) * p.cN + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) sacc += a[u];
            }
            for (; q < p1; ++q) sacc += p.cparts[(int64_t)q * p.cN + c];
        }
        cseg[w][k] = sacc;
        __syncthreads();
    }
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const bool vec = (p.cols & 3) == 0;
    const int row = r0 + ty, col = c0 + 4 * tx;
    const int nv = col < p.cols ? (p.cols - col < 4 ? p.cols - col : 4) : 0;
    const bool live = row < p.rows && nv > 0;
    float gi[4] = {0.f, 0.f, 0.f, 0.f}, wv[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t i = (int64_t)row * p.cols + col;
    if (live) {
        if (vec) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.w + i);
            // (a gradient that arrives in split-K slices never touches g: it is zero and stays zero -- not read, not re-zeroed)
            if (p.slab == nullptr) { const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.g + i); gi[0] = g4.x; gi[1] = g4.y; gi[2] = g4.z; gi[3] = g4.w; }
            wv[0] = w4.x; wv[1] = w4.y; wv[2] = w4.z; wv[3] = w4.w;
            if (optkind != ORX_SGD) { const f32x4 q = *reinterpret_cast<const f32x4*>(p.acc + i); a1[0] = q.x; a1[1] = q.y; a1[2] = q.z; a1[3] = q.w; }
            if (optkind == ORX_ADAM) { const f32x4 q = *reinterpret_cast<const f32x4*>(p.acc2 + i); a2[0] = q.x; a2[1] = q.y; a2[2] = q.z; a2[3] = q.w; }
        } else {
            for (int e = 0; e < nv; ++e) { gi[e] = p.g[i + e]; wv[e] = p.w[i + e]; if (optkind != ORX_SGD) a1[e] = p.acc[i + e]; if (optkind == ORX_ADAM) a2[e] = p.acc2[i + e]; }
        }
        if (p.slab != nullptr) {           // the gradient's split-K slices: tile (row / 128, col / 128), 128 x 128 floats per slice, slice order
            const float* base = p.slab + ((size_t)((row >> 7) * p.ntn + (col >> 7)) * p.S) * ORX_SLAB_STRIDE + (row & 127) * 128 + (col & 127);
            float sv[4] = {0.f, 0.f, 0.f, 0.f};
            for (int z0 = 0; z0 < p.S; z0 += 8) {
                f32x4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    q[u].x = q[u].y = q[u].z = q[u].w = 0.f;
                    if (z0 + u < p.S) {
                        const float* sp = base + (size_t)(z0 + u) * ORX_SLAB_STRIDE;
                        if (vec) q[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp));
                        else { q[u].x = sp[0]; if (nv > 1) q[u].y = sp[1]; if (nv > 2) q[u].z = sp[2]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { sv[0] += q[u].x; sv[1] += q[u].y; sv[2] += q[u].z; sv[3] += q[u].w; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) gi[e] += sv[e] * slab_scale;
        }
        if (cP > 0) {
            for (int e = 0; e < nv; ++e) {
                const int k = (p.rows == 1 ? col + e : row) - cbase;
                gi[e] += (((cseg[0][k] + cseg[1][k]) + cseg[2][k]) + cseg[3][k]) * slab_scale;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (optkind == ORX_ADAM) adam_elem(wv[e], a1[e], a2[e], gi[e], lr, b1, b2, eps);
            else if (optkind == ORX_ADAGRAD) { a1[e] += gi[e] * gi[e]; wv[e] -= lr * gi[e] / (sqrtf(a1[e]) + eps); }
            else wv[e] -= lr * gi[e];
        }
        if (vec) {
            f32x4 o; o.x = wv[0]; o.y = wv[1]; o.z = wv[2]; o.w = wv[3];
            *reinterpret_cast<f32x4*>(p.w + i) = o;
            f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            if (p.slab == nullptr) *reinterpret_cast<f32x4*>(p.g + i) = z4;
            if (optkind != ORX_SGD) { f32x4 q; q.x = a1[0]; q.y = a1[1]; q.z = a1[2]; q.w = a1[3]; *reinterpret_cast<f32x4*>(p.acc + i) = q; }
            if (optkind == ORX_ADAM) { f32x4 q; q.x = a2[0]; q.y = a2[1]; q.z = a2[2]; q.w = a2[3]; *reinterpret_cast<f32x4*>(p.acc2 + i) = q; }
        } else {
            for (int e = 0; e < nv; ++e) { p.w[i + e] = wv[e]; p.g[i + e] = 0.0f; if (optkind != ORX_SGD) p.acc[i + e] = a1[e]; if (optkind == ORX_ADAM) p.acc2[i + e] = a2[e]; }
        }
        if (p.w16 != nullptr) {
            _Float16* q = (_Float16*)p.w16 + (int64_t)row * p.ld16 + col;
            if (vec) { h4 h; h[0] = (_Float16)wv[0]; h[1] = (_Float16)wv[1]; h[2] = (_Float16)wv[2]; h[3] = (_Float16)wv[3]; *reinterpret_cast<h4*>(q) = h; }
            else for (int e = 0; e < nv; ++e) q[e] = (_Float16)wv[e];
        }
    }
    if (p.w16t == nullptr) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) tr[ty][4 * tx + e] = (_Float16)(live && e < nv ? wv[e] : 0.0f);
    __syncthreads();
    // w16t [cols][ld16t]: thread = column c0 + t / 4, 4 consecutive rows r0 + 4 (t % 4) ..
    const int c = threadIdx.x >> 2, rb = (threadIdx.x & 3) * 4;
    if (c0 + c < p.cols) {
        _Float16* q = (_Float16*)p.w16t + (int64_t)(c0 + c) * p.ld16t + r0 + rb;
        if (r0 + rb + 3 < p.rows && (p.ld16t & 3) == 0) {
            h4 h; h[0] = tr[rb][c]; h[1] = tr[rb + 1][c]; h[2] = tr[rb + 2][c]; h[3] = tr[rb + 3][c];
            *reinterpret_cast<h4*>(q) = h;
        } else {
            for (int e = 0; e < 4; ++e) if (r0 + rb + e < p.rows) q[e] = tr[rb + e][c];
        }
    }
}

int orx_launch_dense_apply_fused(orx_ctx* ctx, const DenseFused* ps_dev, const DenseFusedTiles& tt, int total_tiles, int optkind, float lr, float eps,
                                 float b1, float b2, float slab_scale, const CsrFinish* finish) {
    CsrFinish fin;
    memset(&fin, 0, sizeof(fin));
    if (finish != nullptr) fin = *finish;
    if ((tt.count == 0 || total_tiles == 0) && fin.blocks == 0) return ORX_OK;
    ORX_ARG(fin.blocks == 0 || fin.mode == CSR_SGD || fin.mode == CSR_ADAGRAD, "dense_apply_fused: the finish pass it carries is SGD / Adagrad");
    ORX_LAUNCH(ctx, dense_apply_fused_kernel, dim3((unsigned)(total_tiles + fin.blocks)), dim3(256), 0, ps_dev, tt, optkind, lr, eps, b1, b2, slab_scale, fin);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ------------------------------------------------- tiny embedding tables (SGD) ---
// Criteo has tables of 3 .. 27 rows: a batch of 8192 samples puts thousands of gradient rows on each of
// their rows, and fp32 atomics on one address serialize (~60-100 ns each).  One block per (tiny table, slab
// of samples) sums its slab into an LDS copy of the table's gradient ([rows][d], LDS atomics), then adds
// -lr * sum to the table with rows*d global atomics.  Exact scatter_add semantics (only the fp32 summation
// order differs).  apply_rows sees these slots as padding (idx_big = -1).
__global__ __launch_bounds__(256) void dlrm_tiny_apply_kernel(const int32_t* idx, const float* dZ, const int* tiny_f, const int64_t* offset,
                                                              const int64_t* rows, int F, int d, int64_t B, int slab, float lr, float* W) {
    extern __shared__ float acc[];                       // [R][d]
    const int f = tiny_f[blockIdx.x];
    const int R = (int)rows[f];
    const int64_t off = offset[f];
    for (int i = threadIdx.x; i < R * d; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.y * slab, b1 = min(B, b0 + slab);
    if ((d & 3) == 0 && d <= 1024) {
        // a thread owns 4 columns: 16-byte loads, 256 / (d / 4) samples side by side, four rounds in flight -- the pass is bound
        // by bytes in flight per CU (the scalar version kept 8 KB in flight per workgroup: 48 us for 29 MB)
        const int tpr = d / 4, per = 256 / tpr > 0 ? 256 / tpr : 1;
        const int e4 = (threadIdx.x % tpr) * 4, sub = threadIdx.x / tpr;
        if (sub < per) {
            constexpr int UN = 8;
            for (int64_t bb = b0 + sub; bb < b1; bb += UN * per) {
                int r[UN]; f32x4 v[UN];
#pragma unroll
                for (int k = 0; k < UN; ++k) {               // unconditional loads (clamped), so that all UN are in flight together
                    const int64_t x = bb + (int64_t)k * per, xc = x < b1 ? x : b1 - 1;
                    r[k] = idx[xc * F + f];
                    v[k] = *reinterpret_cast<const f32x4*>(dZ + (xc * F + f) * d + e4);
                    if (x >= b1) r[k] = -1;
                }
#pragma unroll
                for (int k = 0; k < UN; ++k)
                    if (r[k] >= 0) {
                        // LDS layout [row][component c][d / 4]: the lanes of one ds_add_f32 hit consecutive banks (with [row][d] they
                        // hit every 4th: 16 banks for 64 lanes, and the pass spent 38 of its 48 us in LDS atomics)
                        float* p = &acc[(int)(r[k] - off) * d + (e4 >> 2)];
                        atomicAdd(p, v[k].x); atomicAdd(p + tpr, v[k].y); atomicAdd(p + 2 * tpr, v[k].z); atomicAdd(p + 3 * tpr, v[k].w);
                    }
            }
        }
    } else {
        const int per = blockDim.x / d > 0 ? blockDim.x / d : 1;   // samples handled concurrently (d <= 256)
        const int e = threadIdx.x % d, sub = threadIdx.x / d;
        if (sub < per) {
            for (int64_t bb = b0 + sub; bb < b1; bb += 4 * per) {
                int r[4]; float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t x = bb + (int64_t)k * per;
                    r[k] = x < b1 ? idx[x * F + f] : -1;
                    v[k] = r[k] >= 0 ? dZ[(x * F + f) * d + e] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (r[k] >= 0) atomicAdd(&acc[(int)(r[k] - off) * d + e], v[k]);
            }
        }
    }
    __syncthreads();
    const bool planar = (d & 3) == 0 && d <= 1024;
    for (int i = threadIdx.x; i < R * d; i += blockDim.x) {
        const int r = i / d, col = i - r * d;
        const float v = planar ? acc[r * d + (col & 3) * (d >> 2) + (col >> 2)] : acc[i];
        if (v != 0.0f) unsafeAtomicAdd(W + off * d + i, -lr * v);
    }
}

// The same sums as a one-hot product on the fp32 MFMA: sum[r][:] = sum_b [row(b) == r] * dZ[b][:], i.e. O^T dZ with O the
// [samples][R] one-hot matrix of the slot's row ids -- exact (products with 0 / 1; fp32 accumulation in MFMA order).  LDS fp32
// atomics turned out to be the whole cost of the LDS version: 38 of its 48 us at the C5 shapes, whatever the slab size or the
// bank layout (7.3 M ds_add_f32 lane-ops at ~3 cycles each per CU).  One wavefront per slab of samples; A operand: lane (i, q)
// holds [row(sample q) == 16 ti + i]; B operand: lane (j, q) owns the CPL = d / 16 consecutive columns CPL j .. of sample q's
// gradient row (16-byte loads; MFMA column tile t = {CPL j + t}), so a lane ends with CPL consecutive columns of rows 4 q' + r.
template <int CPL, int RT>
__global__ __launch_bounds__(256) void dlrm_tiny_apply_mfma_kernel(const int32_t* idx, const float* dZ, const int* tiny_f, const int64_t* offset,
                                                                   const int64_t* rows, int F, int64_t B, int slab, float lr, float* W) {
    constexpr int d = 16 * CPL, UN = 4;
    const int f = tiny_f[blockIdx.x];
    const int R = (int)rows[f];
    const int64_t off = offset[f];
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    extern __shared__ float red[];                          // [R][d]: the block's four wavefronts add their sums in turn
    const int wave = threadIdx.x >> 6;
    const int64_t b0 = ((int64_t)blockIdx.y * 4 + wave) * slab, b1 = min(B, b0 + slab);
    f32x4 acc[RT][CPL];
#pragma unroll
    for (int ti = 0; ti < RT; ++ti)
#pragma unroll
        for (int t = 0; t < CPL; ++t) { acc[ti][t].x = acc[ti][t].y = acc[ti][t].z = acc[ti][t].w = 0.0f; }
    for (int64_t s = b0; s < b1; s += 4 * UN) {
        int r[UN]; float z[UN][CPL];
#pragma unroll
        for (int u = 0; u < UN; ++u) {                       // unconditional (clamped) loads: all UN groups of 4 samples in flight
            const int64_t x = s + 4 * u + q, xc = x < b1 ? x : b1 - 1;
            r[u] = (int)(idx[xc * F + f] - off);
            const float* zp = dZ + (xc * F + f) * d + CPL * i;
            if (CPL >= 4) {
#pragma unroll
                for (int c4 = 0; c4 < CPL; c4 += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(zp + c4);
                    z[u][c4] = v.x; z[u][c4 + 1] = v.y; z[u][c4 + 2] = v.z; z[u][c4 + 3] = v.w;
                }
            } else { z[u][0] = zp[0]; z[u][1] = zp[1]; }
            if (x >= b1) r[u] = -1;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int ti = 0; ti < RT; ++ti) {
                if (R <= 16 * ti) continue;                      // (wave-uniform)
                const float a = r[u] == 16 * ti + i ? 1.0f : 0.0f;
#pragma unroll
                for (int t = 0; t < CPL; ++t) acc[ti][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, z[u][t], acc[ti][t], 0, 0, 0);
            }
        }
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ti = 0; ti < RT; ++ti)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int g = 16 * ti + 4 * q + rr;
                    if (g >= R) continue;
                    float* p = red + g * d + CPL * i;
#pragma unroll
                    for (int t = 0; t < CPL; ++t) p[t] = (w == 0 ? 0.0f : p[t]) + acc[ti][t][rr];
                }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < R * d; e += 256) {
        const float v = red[e];
        if (v != 0.0f) unsafeAtomicAdd(W + off * d + e, -lr * v);
    }
}

int orx_launch_dlrm_tiny_apply(orx_ctx* ctx, const int32_t* idx, const float* dZ, const int* tiny_f_dev, int n_tiny, int max_rows,
                               const int64_t* offset, const int64_t* rows, int F, int d, int64_t B, float lr, float* W) {
    if (n_tiny == 0 || B == 0) return ORX_OK;
    static const int slab_env = getenv("ORX_TINY_SLAB") ? atoi(getenv("ORX_TINY_SLAB")) : 0;
    if (max_rows <= 64 && (d == 32 || d == 64 || d == 128 || d == 256) && getenv("ORX_TINY_LDS") == nullptr) {
        const int slab = slab_env > 0 ? slab_env : 64;                   // samples per wavefront (4 wavefronts per workgroup)
        const dim3 g((unsigned)n_tiny, (unsigned)((B + 4 * slab - 1) / (4 * slab)));
#define ORX_TINY_GO(CPL, RT) ORX_LAUNCH(ctx, (dlrm_tiny_apply_mfma_kernel<CPL, RT>), g, dim3(256), (size_t)max_rows * d * sizeof(float), idx, dZ, tiny_f_dev, offset, rows, F, B, slab, lr, W)
        if (max_rows <= 32) {
            if (d == 32) ORX_TINY_GO(2, 2); else if (d == 64) ORX_TINY_GO(4, 2); else if (d == 128) ORX_TINY_GO(8, 2); else ORX_TINY_GO(16, 2);
        } else {
            if (d == 32) ORX_TINY_GO(2, 4); else if (d == 64) ORX_TINY_GO(4, 4); else if (d == 128) ORX_TINY_GO(8, 4); else ORX_TINY_GO(16, 4);
        }
#undef ORX_TINY_GO
        ORX_HIP(hipGetLastError());
        return ORX_OK;
    }
    const int slab = slab_env > 0 ? slab_env : 128;
    ORX_LAUNCH(ctx, dlrm_tiny_apply_kernel, dim3((unsigned)n_tiny, (unsigned)((B + slab - 1) / slab)), dim3(256),
               (size_t)max_rows * d * sizeof(float), idx, dZ, tiny_f_dev, offset, rows, F, d, B, slab, lr, W);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// idx_big[i] = idx[i] unless slot i belongs to a tiny table (then -1)
__global__ void dlrm_mask_tiny_kernel(const int32_t* idx, const unsigned char* is_tiny, int F, int64_t total, int32_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int f = (int)(i % F);
        out[i] = is_tiny[f] ? -1 : idx[i];
    }
}

int orx_launch_dlrm_mask_tiny(orx_ctx* ctx, const int32_t* idx, const unsigned char* is_tiny_dev, int F, int64_t total, int32_t* out) {
    int64_t g = (total + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
    ORX_LAUNCH(ctx, dlrm_mask_tiny_kernel, dim3((unsigned)g), dim3(256), 0, idx, is_tiny_dev, F, total, out);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
