// fp16-MFMA products of the DLRM top MLP on fp16-resident operands (ORX_DLRM_FP16_MLP; recommenders/dlrm.py:76-100,
// modules/multi_layer_perceptron.py:5-18), second generation (scratch/exp_gemm.hip has the experiments):
//
//   gemm16_nt_kernel   C[M][N] = A16[M][K] * B16[N][K]^T  -- forward (X16 * W16T) and input gradient (dZ16 * W16): both
//                      operands K-contiguous.  BK = 64, two LDS stages (one barrier per K step; the next tile is written to
//                      LDS after the barrier while the tile after it is already in flight to registers), MFMA operands
//                      swapped so a lane owns 4 consecutive columns of a row (16-byte epilogue accesses), workgroup ids
//                      remapped so that an XCD (workgroup b runs on XCD b % 8) owns whole row blocks of A: its L2 then
//                      holds that slice of A and all of B instead of all of A (128x128 tiles: 36.2 -> 32.7 us on
//                      8192x1024x1024; 256x128 tiles of 8 wavefronts: 31.0; the round-1 kernel: 46.0).
//                      Epilogue: + bias, activation, optional fp32 store, optional fp16 copy; or the fused activation
//                      backward of the layer below (dZ = dX * act'(Y), Y from its fp32 or fp16 copy) with the bias-gradient
//                      column sums.
//   gemm16_tn_kernel   C[M][N] += A16[K][M]^T * B16[K][N] -- weight gradient X16^T * dZ16 straight from the BATCH-major fp16
//                      copies the other two products already use (no transposed shadows, no fp32 re-read): the tiles stay
//                      k-major in LDS and the MFMA fragments come from ds_read_b64_tr_b16.  Split-K over the batch; the
//                      slices leave through fp32 slabs (plain 16-byte stores) that slab_reduce_kernel adds into the
//                      gradient: fp32 atomics run at ~85 G/s here (+95 us for a 1024x1024 layer), an in-kernel last-arriver
//                      reduction pays an L2 write-back + invalidate per workgroup (+50 us), slabs + one reduce launch +14 us.
#include "orx_device.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));

struct Nt16Args {
    const _Float16* A; int64_t lda;
    const _Float16* B; int64_t ldb;
    float* C; int64_t ldc;              // optional
    _Float16* C16; int64_t ldc16;       // optional
    const float* bias;
    int M, N, K, act;
    const float* actY; const _Float16* actY16; int64_t ldy; int act_y; float* gb;      // fused activation backward (see above)
    // relu masks (round 6): ONE 64-bit word per lane and tile -- bit (mi * (TN / 2) + np) * 8 + e says whether the fp16 output of the lane's
    // element (row tile mi, column pair np, element e) is > 0 -- written by the forward launch of a relu layer (mask_out), read by the input-
    // gradient launch above it (mask_in) in place of the eight 16-byte requests per lane that fetched the layer's fp16 output for the fused
    // activation backward (scratch/exp_k512.hip: those requests cost ~5 us of a 18-27 us launch; one request ~1 us).  Word index =
    // tile * threads + thread: forward and backward launch must use the same tile configuration (the host checks: orx_gemm16_nt_config).
    unsigned long long* mask_out; const unsigned long long* mask_in;
};

// (XCD-aware tile order: xcd_slot, orx_device.h)

// the products' common epilogue: + bias, activation, stores, or the fused activation backward with its column sums (see the head
// of the file); `red` = LDS that is free once the main loop is over, [WM][BN] floats
// NTS: how the outputs are stored: 0 = ordinary stores, 1 = nontemporal, 2 = write-through (sc1: past the XCD's L2 as they are issued,
// nothing left dirty for the end of the kernel)
// What the epilogue reads -- the bias of the lane's columns and, for the fused activation backward, the layer below's fp16 output at the
// lane's elements -- requested at the START of the kernel, ahead of the first tiles (round 6).  scratch/exp_k512.hip: with the loads issued inside
// the epilogue's (row tile, column pair) loop, each of its eight iterations waited for its own round trip: the backward form's epilogue took
// 11 us of a 20.5 us launch at 8192 x 1024 x 512 (forward form: 5 of 14.5).  Only for tiles that lie inside the matrices and have
// 16-byte rows (`fast`, uniform over the workgroup); edge tiles take the element-wise path as before.
// (plain arrays handed on by reference: a struct behind a pointer went to scratch memory -- the "prefetched" operands then came back from HBM)
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ size_t nt_mask_index(const Nt16Args& g, int bm, int bn) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = 64 * WM * WN;
    const int ntn = (g.N + BN - 1) / BN;
    return ((size_t)(bm / BM) * ntn + (size_t)(bn / BN)) * NT + threadIdx.x;
}
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ bool nt_epilogue_prefetch(const Nt16Args& g, int bm, int bn, int wm, int wn, h8 (&py)[TM][TN / 2], f32x4 (&pb)[TN / 2][2],
                                                     unsigned long long& pmask) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    const bool use_mask = g.mask_in != nullptr && g.act_y == 1;
    pmask = use_mask ? g.mask_in[nt_mask_index<WM, WN, TM, TN>(g, bm, bn)] : 0ull;
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, q = lane >> 4;
    const bool fast = bm + BM <= g.M && bn + BN <= g.N && (g.N & 7) == 0 && (g.actY16 == nullptr || (g.ldy & 7) == 0) && g.actY == nullptr &&
                      (g.C == nullptr || (g.ldc & 3) == 0) && (g.C16 == nullptr || (g.ldc16 & 7) == 0);
#pragma unroll
    for (int np = 0; np < TN / 2; ++np) {
        const int col = bn + wn + np * 32 + q * 8;
        f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        pb[np][0] = z4; pb[np][1] = z4;
        if (fast && g.bias) { pb[np][0] = *reinterpret_cast<const f32x4*>(g.bias + col); pb[np][1] = *reinterpret_cast<const f32x4*>(g.bias + col + 4); }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int e = 0; e < 8; ++e) py[mi][np][e] = (_Float16)0.0f;
#ifdef ORX_EXP_Y1          // (scratch/exp_k512.hip: what ONE request per lane costs instead of eight)
            if (fast && g.actY16 && mi == 0 && np == 0) py[mi][np] = *reinterpret_cast<const h8*>(g.actY16 + (int64_t)(bm + wm + r16) * g.ldy + col);
#else
            if (fast && g.actY16 && !use_mask) py[mi][np] = *reinterpret_cast<const h8*>(g.actY16 + (int64_t)(bm + wm + mi * 16 + r16) * g.ldy + col);
#endif
        }
    }
    return fast;
}

// the column sums' second half (shared by both epilogue forms): over the block's WM wavefronts through LDS, one plain store per (workgroup, column)
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void nt_colsums(const Nt16Args& g, float (&cs)[TN / 2][8], int bm, int bn, int wn, float* red) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = 64 * WM * WN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int np = 0; np < TN / 2; ++np)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = group_allreduce<16>(cs[np][e]);
            if (r16 == 0) red[(wave / WN) * BN + wn + np * 32 + q * 8 + e] = s;
        }
    __syncthreads();
    for (int cidx = threadIdx.x; cidx < BN; cidx += NT) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < WM; ++w) s += red[w * BN + cidx];
        if (bn + cidx < g.N) g.gb[(int64_t)(bm / BM) * g.N + bn + cidx] = s;
    }
}

// The epilogue of a tile that lies INSIDE the matrices with 16-byte rows everywhere (NtEpiPre::fast, every tile of the C5 shapes): 16-byte
// accesses only, no bounds checks, no element-wise fallbacks -- a few hundred instructions.  scratch/exp_k512.hip: the general form below,
// unrolled eight times with its fallbacks, ran the backward form's epilogue 8 us longer than its memory traffic explains (its code alone is
// most of a 10 000-line kernel: instruction fetch, not data).
template <int WM, int WN, int TM, int TN, int NTS>
__device__ __forceinline__ void nt_epilogue_fast(const Nt16Args& g, f32x4 (&acc)[TM][TN], int bm, int bn, int wm, int wn, float* red,
                                                 const h8 (&py)[TM][TN / 2], const f32x4 (&pb)[TN / 2][2], unsigned long long pmask) {
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, q = lane >> 4;
    float cs[TN / 2][8];
#pragma unroll
    for (int np = 0; np < TN / 2; ++np)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[np][e] = 0.0f;
    const bool has_y = g.actY16 != nullptr;
    const bool use_mask = g.mask_in != nullptr && g.act_y == 1;
    unsigned long long omask = 0ull;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int64_t row = bm + wm + mi * 16 + r16;
#pragma unroll
        for (int np = 0; np < TN / 2; ++np) {
            const int col = bn + wn + np * 32 + q * 8;
            float v[8] = {acc[mi][2 * np].x, acc[mi][2 * np].y, acc[mi][2 * np].z, acc[mi][2 * np].w,
                          acc[mi][2 * np + 1].x, acc[mi][2 * np + 1].y, acc[mi][2 * np + 1].z, acc[mi][2 * np + 1].w};
            if (g.bias) {
                const f32x4 b0 = pb[np][0], b1 = pb[np][1];
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (g.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
            } else if (g.act == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 1.0f / (1.0f + __expf(-v[e]));
            }
            if (use_mask) {
                const unsigned bits = (unsigned)(pmask >> ((mi * (TN / 2) + np) * 8)) & 0xffu;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = ((bits >> e) & 1u) ? v[e] : 0.0f;
                    cs[np][e] += v[e];
                }
            } else if (has_y) {
                const h8 t8 = py[mi][np];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = (float)t8[e];
                    v[e] = g.act_y == 1 ? (y > 0.0f ? v[e] : 0.0f) : (g.act_y == 2 ? v[e] * y * (1.0f - y) : v[e]);
                    cs[np][e] += v[e];
                }
            }
            if (g.mask_out) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((_Float16)v[e] > (_Float16)0.0f) omask |= 1ull << ((mi * (TN / 2) + np) * 8 + e);
            }
            if (g.C) {
                float* p = g.C + row * g.ldc + col;
                f32x4 o0, o1; o0.x = v[0]; o0.y = v[1]; o0.z = v[2]; o0.w = v[3]; o1.x = v[4]; o1.y = v[5]; o1.z = v[6]; o1.w = v[7];
                if (NTS == 2) { store_sc1(reinterpret_cast<f32x4*>(p), o0); store_sc1(reinterpret_cast<f32x4*>(p + 4), o1); }
                else if (NTS == 1) { __builtin_nontemporal_store(o0, reinterpret_cast<f32x4*>(p)); __builtin_nontemporal_store(o1, reinterpret_cast<f32x4*>(p + 4)); }
                else { *reinterpret_cast<f32x4*>(p) = o0; *reinterpret_cast<f32x4*>(p + 4) = o1; }
            }
            if (g.C16) {
                _Float16* p = g.C16 + row * g.ldc16 + col;
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
                if (NTS == 2) store_sc1(reinterpret_cast<h8*>(p), o); else if (NTS == 1) __builtin_nontemporal_store(o, reinterpret_cast<h8*>(p)); else *reinterpret_cast<h8*>(p) = o;
            }
        }
    }
    if (g.mask_out) g.mask_out[nt_mask_index<WM, WN, TM, TN>(g, bm, bn)] = omask;
    if (NTS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g.gb) nt_colsums<WM, WN, TM, TN>(g, cs, bm, bn, wn, red);
}

template <int WM, int WN, int TM, int TN, int NTS = 0>
__device__ __forceinline__ void nt_epilogue(const Nt16Args& g, f32x4 (&acc)[TM][TN], int bm, int bn, int wm, int wn, float* red) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT = 64 * WM * WN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    // ---- epilogue: the lane holds C[row = .. + r16][col = 32 * np + 8q .. + 7] for every tile pair np
    static_assert(TN % 2 == 0, "tile pairs");
    const bool vec = (g.N & 7) == 0;
    float cs[TN / 2][8];
    unsigned long long gen_mask = 0ull;
#pragma unroll
    for (int np = 0; np < TN / 2; ++np)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[np][e] = 0.0f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = bm + wm + mi * 16 + r16;
#pragma unroll
        for (int np = 0; np < TN / 2; ++np) {
            const int col = bn + wn + np * 32 + q * 8;
            if (row >= g.M || col >= g.N) continue;
            float v[8] = {acc[mi][2 * np].x, acc[mi][2 * np].y, acc[mi][2 * np].z, acc[mi][2 * np].w,
                          acc[mi][2 * np + 1].x, acc[mi][2 * np + 1].y, acc[mi][2 * np + 1].z, acc[mi][2 * np + 1].w};
            const bool full = vec && col + 7 < g.N;
            if (g.bias) {
                if (full) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + col), b1 = *reinterpret_cast<const f32x4*>(g.bias + col + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (col + e < g.N) v[e] += g.bias[col + e];
                }
            }
            if (g.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
            } else if (g.act == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 1.0f / (1.0f + __expf(-v[e]));
            }
            if (g.actY || g.actY16) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = 0.0f;
                if (g.actY16) {
                    if (full && (g.ldy & 7) == 0) {
                        const h8 t8 = *reinterpret_cast<const h8*>(g.actY16 + (int64_t)row * g.ldy + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] = (float)t8[e];
                    } else {
                        for (int e = 0; e < 8; ++e) if (col + e < g.N) y[e] = (float)g.actY16[(int64_t)row * g.ldy + col + e];
                    }
                } else {
                    if (full && (g.ldy & 3) == 0) {
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(g.actY + (int64_t)row * g.ldy + col), t1 = *reinterpret_cast<const f32x4*>(g.actY + (int64_t)row * g.ldy + col + 4);
                        y[0] = t0.x; y[1] = t0.y; y[2] = t0.z; y[3] = t0.w; y[4] = t1.x; y[5] = t1.y; y[6] = t1.z; y[7] = t1.w;
                    } else {
                        for (int e = 0; e < 8; ++e) if (col + e < g.N) y[e] = g.actY[(int64_t)row * g.ldy + col + e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = g.act_y == 1 ? (y[e] > 0.0f ? v[e] : 0.0f) : (g.act_y == 2 ? v[e] * y[e] * (1.0f - y[e]) : v[e]);
                    if (col + e < g.N) cs[np][e] += v[e];
                }
            }
            if (g.C) {
                float* p = g.C + (int64_t)row * g.ldc + col;
                if (full && (g.ldc & 3) == 0) {
                    f32x4 o0, o1; o0.x = v[0]; o0.y = v[1]; o0.z = v[2]; o0.w = v[3]; o1.x = v[4]; o1.y = v[5]; o1.z = v[6]; o1.w = v[7];
                    if (NTS == 2) { store_sc1(reinterpret_cast<f32x4*>(p), o0); store_sc1(reinterpret_cast<f32x4*>(p + 4), o1); }
                    else if (NTS == 1) { __builtin_nontemporal_store(o0, reinterpret_cast<f32x4*>(p)); __builtin_nontemporal_store(o1, reinterpret_cast<f32x4*>(p + 4)); }
                    else { *reinterpret_cast<f32x4*>(p) = o0; *reinterpret_cast<f32x4*>(p + 4) = o1; }
                } else {
                    for (int e = 0; e < 8; ++e) if (col + e < g.N) p[e] = v[e];
                }
            }
            if (g.mask_out) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < g.N && (_Float16)v[e] > (_Float16)0.0f) gen_mask |= 1ull << ((mi * (TN / 2) + np) * 8 + e);
            }
            if (g.C16) {
                _Float16* p = g.C16 + (int64_t)row * g.ldc16 + col;
                if (full && (g.ldc16 & 7) == 0) {
                    h8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
                    if (NTS == 2) store_sc1(reinterpret_cast<h8*>(p), o); else if (NTS == 1) __builtin_nontemporal_store(o, reinterpret_cast<h8*>(p)); else *reinterpret_cast<h8*>(p) = o;
                } else {
                    for (int e = 0; e < 8; ++e) if (col + e < g.N) p[e] = (_Float16)v[e];
                }
            }
        }
    }
    if (g.mask_out) g.mask_out[nt_mask_index<WM, WN, TM, TN>(g, bm, bn)] = gen_mask;
    if (NTS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the write-through stores are acknowledged before the wavefront ends)
    if (g.gb) {
        // column sums: over the 16 rows of a lane group (DPP), over the block's WM wavefronts through LDS (the stages are
        // free now), then ONE plain store per (workgroup, column) into the row block's partial row: colparts_reduce_kernel adds
        // the row blocks in order (atomics would make the sum depend on arrival order)
#pragma unroll
        for (int np = 0; np < TN / 2; ++np)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = group_allreduce<16>(cs[np][e]);
                if (r16 == 0) red[(wave / WN) * BN + wn + np * 32 + q * 8 + e] = s;
            }
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < BN; cidx += NT) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += red[w * BN + cidx];
            if (bn + cidx < g.N) g.gb[(int64_t)(bm / BM) * g.N + bn + cidx] = s;
        }
    }
}

// PAD: halves of padding per LDS tile row.  8 (144-byte rows) leaves 36 % of the LDS-active cycles in bank conflicts
// (profiles/r2_dlrm_fp16_pmc_mfma_lds.csv); 16 (160-byte rows) removes them: main loop 21.1 -> 20.0 us on 8192 x 1024 x 1024
// (profiles/r2_exp_gemm_lds_pitch.log).  The 128 x 128 configuration keeps 8: two workgroups of it must share a CU's 160 KB.
template <int WM, int WN, int TM, int TN, int MINB, int PAD>
__global__ __launch_bounds__(64 * WM * WN, MINB) void gemm16_nt_kernel(Nt16Args g) {
    constexpr int BK = 64, BM = WM * TM * 16, BN = WN * TN * 16, LD = BK + PAD, NT = 64 * WM * WN;
    constexpr int CPR = BK / 8;                                  // 16-byte chunks per tile row
    constexpr int NA = BM * CPR / NT, NB = BN * CPR / NT;
    static_assert(BM * CPR % NT == 0 && BN * CPR % NT == 0, "tile / thread mismatch");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds16[];
    constexpr int STAGE = (BM + BN) * LD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = xcd_slot(blockIdx.x, gridDim.x);
    const int ntn = (g.N + BN - 1) / BN;
    const int bm = (t / ntn) * BM, bn = (t % ntn) * BN;
    const int wm = (wave / WN) * TM * 16, wn = (wave % WN) * TN * 16;
    const int r16 = lane & 15, q = lane >> 4;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
    h8 ra[NA], rb[NB];
    h8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (_Float16)0.0f;
    auto gload = [&](int k0) {                                   // leading dims are multiples of 8 halves, zero padded
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = threadIdx.x + NT * i, row = c / CPR, k = k0 + (c % CPR) * 8;
            ra[i] = (bm + row < g.M && k < g.lda) ? *reinterpret_cast<const h8*>(g.A + (int64_t)(bm + row) * g.lda + k) : zero;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = threadIdx.x + NT * i, row = c / CPR, k = k0 + (c % CPR) * 8;
            rb[i] = (bn + row < g.N && k < g.ldb) ? *reinterpret_cast<const h8*>(g.B + (int64_t)(bn + row) * g.ldb + k) : zero;
        }
    };
    // B rows are permuted on their way into LDS: column n = 32a + 8b + 4c + d of the tile sits in LDS row 32a + 16c + 4b + d, so
    // MFMA tile ni = 2a + c, fragment row i = 4b + d is column 32a + 8b + 4c + d: lane group q = b then owns, over the tile
    // pair (2a, 2a + 1), the 8 CONSECUTIVE columns 32a + 8q .. + 7 of its row (16-byte fp16 / 32-byte fp32 epilogue accesses)
    auto lstore = [&](_Float16* S) {
#pragma unroll
        for (int i = 0; i < NA; ++i) { const int c = threadIdx.x + NT * i; *reinterpret_cast<h8*>(S + (c / CPR) * LD + (c % CPR) * 8) = ra[i]; }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = threadIdx.x + NT * i, n = c / CPR;
            const int row = (n & ~31) | ((n & 4) << 2) | ((n & 24) >> 1) | (n & 3);
            *reinterpret_cast<h8*>(S + (BM + row) * LD + (c % CPR) * 8) = rb[i];
        }
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
                for (int ni = 0; ni < TN; ++ni)       // operands swapped: D[n][m], the lane holds C[m = r16][n = 4q .. 4q + 3]
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
    };
    const int nk = (g.K + BK - 1) / BK;
    gload(0);
    lstore(lds16);
    if (nk > 1) gload(BK);
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
        _Float16* cur = lds16 + (s & 1) * STAGE; _Float16* nxt = lds16 + ((s + 1) & 1) * STAGE;
        if (s + 1 < nk) lstore(nxt);
        if (s + 2 < nk) gload((s + 2) * BK);
        compute(cur);
        __syncthreads();
    }
    nt_epilogue<WM, WN, TM, TN>(g, acc, bm, bn, wm, wn, reinterpret_cast<float*>(lds16));
}

// ---- the same product with the tiles brought into LDS by the LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write pass -- the 16-byte LDS stores of the kernel above move 48 KB per K step at ~79 B/clk, more LDS time than the
// fragment reads of the step).  The DMA writes lane-linear (wave-uniform base + 16 * lane), so a tile row is 128 bytes without
// padding and the bank spread comes from the SOURCE side: LDS row r holds the eight 16-byte chunks of its K slice at slot
// (chunk ^ (r & 7)); a ds_read_b128 lane group (16 rows, two of the four k-chunks) then covers all 64 banks once.
// NS LDS stages: tile s + NS is requested while the second half of tile s is multiplied; one raw s_barrier per K step, the waits
// on the DMA are counted (vmcnt) so that NS - 2 tiles stay in flight across the barrier.
// Chunks beyond a row's leading dimension (K tails) and rows beyond M / N come from a 16-byte block of zeros.
__device__ __attribute__((aligned(16))) const _Float16 g_zero_chunk[8] = {};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// TAIL: a leading dimension is not a multiple of 64 halves (the last tile's chunks beyond it come from the zeros, through per-lane
// 64-bit addresses).  All other requests are `global_load_lds_dwordx4 v_offset, s[base]`: a scalar base that advances with the tile
// and 32-bit per-lane offsets made once -- no vector arithmetic per request.
// DBG (scratch/exp_dma.hip only): 1 = no epilogue, 2 = no DMA inside the loop, 4 = no fragment reads / MFMA, 16 = nontemporal stores, 32 = write-through stores,
// 8 = with 1: the main loop's duration in shader cycles and in 100 MHz ticks goes to g.C as [workgroup][2] 64-bit counts
// (the body is a device function of (workgroup index, number of workgroups of the product): gemm16_group_kernel runs several products in one grid)
template <int WM, int WN, int TM, int TN, int NS, bool TAIL, int DBG = 0>
__device__ __forceinline__ void gemm16_nt_dma_body(const Nt16Args& g, const int bid, const int nblk) {
    constexpr int BK = 64, BM = WM * TM * 16, BN = WN * TN * 16, NT = 64 * WM * WN;
    constexpr int NA = BM * 8 / NT, NB = BN * 8 / NT, NL = NA + NB;            // DMA instructions per wavefront and tile
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile / thread mismatch");
    static_assert(NS == 2 || NS == 3, "two or three stages");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds16[];
    constexpr int STAGE = (BM + BN) * BK;                                      // halves
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = xcd_slot(bid, nblk);
    const int ntn = (g.N + BN - 1) / BN;
    const int bm = (t / ntn) * BM, bn = (t % ntn) * BN;
    const int wm = (wave / WN) * TM * 16, wn = (wave % WN) * TN * 16;
    const int r16 = lane & 15, q = lane >> 4;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
    // the lane's source chunk of DMA instruction i: LDS chunk c = 64 * wave + lane + NT * i -> LDS row c / 8, slot c % 8.  Rows
    // beyond M / N are clamped to the last one (their products are never stored).
    // B rows are permuted on their way into LDS exactly as in the kernel above (column n = 32a + 8b + 4c + d sits in LDS row
    // 32a + 16c + 4b + d): LDS row r = 32a + 16c + 4b + d is column 32a + 8b + 4c + d
    unsigned off[NL];                                                          // byte offset of the chunk from A / B at k0 = 0
    int kc[NL];                                                                // TAIL: the chunk's k offset inside a tile (halves)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int c = wave * 64 + lane + NT * (i < NA ? i : i - NA), r = c >> 3;
        kc[i] = ((c & 7) ^ (r & 7)) * 8;
        if (i < NA) {
            off[i] = (unsigned)(((int64_t)min(bm + r, g.M - 1) * g.lda + kc[i]) * 2);
        } else {
            const int n = (r & ~31) | ((r & 12) << 1) | ((r & 16) >> 2) | (r & 3);
            off[i] = (unsigned)(((int64_t)min(bn + n, g.N - 1) * g.ldb + kc[i]) * 2);
        }
    }
    const char* const baseA = reinterpret_cast<const char*>(g.A);
    const char* const baseB = reinterpret_cast<const char*>(g.B);
    const int lda = (int)g.lda, ldb = (int)g.ldb;
    auto issue_one = [&](int k0, _Float16* S, int i) {                       // request i of the NL of a tile (i: a constant after unrolling)
        const int j = i < NA ? i : i - NA;
        _Float16* dst = S + (i < NA ? 0 : BM * BK) + (NT * j + wave * 64) * 8;
        if (TAIL && k0 + BK > (i < NA ? lda : ldb)) {                          // (uniform: the last tile of a row that ends inside it)
            const char* p = (i < NA ? baseA : baseB) + (size_t)k0 * 2 + off[i];
            if (k0 + kc[i] >= (i < NA ? lda : ldb)) p = reinterpret_cast<const char*>(g_zero_chunk);
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)dst, 16, 0, 0);
        } else {
            // (written out: the compiler folds the tile's advance into per-lane 64-bit addresses, two vector adds per request)
            const char* sb = (i < NA ? baseA : baseB) + (size_t)k0 * 2;
            const unsigned m0v = (unsigned)(uintptr_t)(lptr_t)dst;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off[i]), "s"(sb), "s"(m0v) : "memory");
        }
    };
    auto issue = [&](int k0, _Float16* S) {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_one(k0, S, i);
    };
    // fragment addresses: row * 64 halves + ((4 kk + q) ^ (row & 7)) * 8, and (row & 7) == (r16 & 7) for every fragment row
    const int fa0 = (wm + r16) * BK + ((q ^ (r16 & 7)) * 8), fb0 = (BM + wn + r16) * BK + ((q ^ (r16 & 7)) * 8);
    h8 a0[TM], b0[TN], a1[TM], b1[TN];                                         // fragments of the two 32-deep halves of a tile
    auto fetch = [&](const _Float16* S, int kk, h8 (&a)[TM], h8 (&b)[TN]) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) a[mi] = *reinterpret_cast<const h8*>(S + ((fa0 + mi * 16 * BK) ^ (kk * 32)));
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const h8*>(S + ((fb0 + ni * 16 * BK) ^ (kk * 32)));
    };
    auto arrived = [&](h8 (&a)[TM], h8 (&b)[TN]) {                             // (see the end of `step`)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) asm volatile("" : "+v"(a[mi]));
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) asm volatile("" : "+v"(b[ni]));
    };
    // The loop is software-pipelined around ONE barrier per tile: the fragments of half 1 are read while half 0 is multiplied;
    // then the barrier (every wavefront has read all of tile s: its stage is free; tile s + 1 has landed for everyone); then
    // the fragments of tile s + 1's half 0 are read and the NL DMA requests of tile s + NS go out BETWEEN the products of half 1
    // (a wavefront that issues its requests back to back stands still for ~150 cycles each).
    const int nk = (g.K + BK - 1) / BK;
    int cur = 0;                                                               // stage of tile s
    // DMA: tile s + NS exists and is requested; MORE: tile s + 1 exists; INFLIGHT: tile s + 2 has been requested (NS = 3)
    auto step = [&](auto dma_c, auto more_c, auto inflight_c, int s) {
        constexpr bool DMA = decltype(dma_c)::value && !(DBG & 2), MORE = decltype(more_c)::value, INFLIGHT = decltype(inflight_c)::value;
        _Float16* S = lds16 + cur * STAGE;
        cur = cur + 1 == NS ? 0 : cur + 1;
        if (!(DBG & 4)) {
            fetch(S, 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);                                 // (the reads go out first ...)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
        }
        // my reads of tile s are complete (their data is in a1 / b1); tile s + 1 has landed for me
        __builtin_amdgcn_sched_barrier(0);                                     // (... and the products above stay above: they cover the reads)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(DBG & 4)) arrived(a1, b1);                                       // (the compiler's own count: nothing pending behind the barrier)
        if (MORE) { if (NS == 3 && INFLIGHT) wait_vm<NL>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        if (!(DBG & 4) && MORE) fetch(lds16 + cur * STAGE, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        const int k0 = (s + NS) * BK;
        if (DBG & 4) {
            if (DMA) issue(k0, S);
            return;
        }
        constexpr int EVERY = TM * TN / NL > 0 ? TM * TN / NL : 1;             // one request after every EVERY-th product
        static_assert(TM * TN / EVERY >= NL, "not every request of a tile finds a place");
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
                const int done = mi * TN + ni + 1;
                if (DMA && done % EVERY == 0 && done / EVERY <= NL) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_one(k0, S, done / EVERY - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        // the fragments read behind the barrier have arrived long before these products end; saying so HERE leaves no LDS read
        // pending across the back edge (the compiler's wait counts then let the next half's reads fly under the products above)
        if (MORE) arrived(a0, b0);
    };
    // The epilogue's operands are requested FIRST, ahead of the first tiles: the memory counter the loop's waits count (vmcnt) retires in
    // order, so requests OLDER than every tile are long complete wherever the loop waits, while requests issued later -- before the last
    // K steps, say -- would sit among the newest ones and be waited for with the tiles (measured: no gain that way).
    h8 pre_y[TM][TN / 2]; f32x4 pre_b[TN / 2][2];
    unsigned long long pre_mask = 0ull;
    bool pre_fast = false;
    if (!(DBG & 1)) pre_fast = nt_epilogue_prefetch<WM, WN, TM, TN>(g, bm, bn, wm, wn, pre_y, pre_b, pre_mask);
#pragma unroll
    for (int s = 0; s < NS; ++s) if (s < nk) issue(s * BK, lds16 + s * STAGE);
    if (nk > 2 && NS == 3) wait_vm<2 * NL>(); else if (nk > 1) wait_vm<NL>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (!(DBG & 4)) { fetch(lds16, 0, a0, b0); arrived(a0, b0); }
    unsigned long long dbg_c0 = 0, dbg_w0 = 0;
    if (DBG & 8) { dbg_c0 = __builtin_amdgcn_s_memtime(); dbg_w0 = wall_clock64(); }
    using T = std::true_type; using F = std::false_type;
    int s = 0;
    for (; s + NS < nk; ++s) step(T(), T(), T(), s);
    if (NS == 3 && s + 2 < nk) { step(F(), T(), T(), s); ++s; }                // (tile s + 2 is the last one: in flight, nothing more to request)
    if (s + 1 < nk) { step(F(), T(), F(), s); ++s; }
    if (s < nk) step(F(), F(), F(), s);
    if (DBG & 8) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
        if (threadIdx.x == 0) { unsigned long long* o = reinterpret_cast<unsigned long long*>(g.C) + 2 * bid; o[0] = c1 - dbg_c0; o[1] = w1 - dbg_w0; }
    }
    if (DBG & 1) {
        float x = 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) x += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
        if (x == 12345.678f) g.C16[0] = (_Float16)x;
        return;
    }
    if (g.gb) __syncthreads();                                                 // (the column sums go through the stages' LDS)
    if (pre_fast) nt_epilogue_fast<WM, WN, TM, TN, (DBG & 32) ? 2 : (DBG & 16) ? 1 : 0>(g, acc, bm, bn, wm, wn, reinterpret_cast<float*>(lds16), pre_y, pre_b, pre_mask);
    else nt_epilogue<WM, WN, TM, TN, (DBG & 32) ? 2 : (DBG & 16) ? 1 : 0>(g, acc, bm, bn, wm, wn, reinterpret_cast<float*>(lds16));
}

template <int WM, int WN, int TM, int TN, int MINB, int NS, bool TAIL, int DBG = 0>
__global__ __launch_bounds__(64 * WM * WN, MINB) void gemm16_nt_dma_kernel(Nt16Args g) {
    gemm16_nt_dma_body<WM, WN, TM, TN, NS, TAIL, DBG>(g, (int)blockIdx.x, (int)gridDim.x);
}

template <int WM, int WN, int TM, int TN, int MINB, int NS>
static int launch_nt_dma(orx_ctx* ctx, const Nt16Args& g) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr size_t shm = (size_t)NS * (BM + BN) * 64 * 2;
    const bool tail = (g.lda & 63) != 0 || (g.ldb & 63) != 0;
    // How the outputs leave (ORX_GEMM16_NTS): 2 (default) = write-through stores, nothing left dirty in the XCD's L2 for the end of the
    // kernel: alone, 8192 x 1024 x 1024 goes from 24.3 to 21.0 us (scratch/exp_dma.hip), in the DLRM step the products gain 1 %;
    // 1 = nontemporal stores: 19.9 us alone, but the NEXT product then reads its operand from HBM and the step is 2 % slower;
    // 0 = ordinary stores.
    static const int nts = getenv("ORX_GEMM16_NTS") != nullptr ? atoi(getenv("ORX_GEMM16_NTS")) : 2;
    using K = void (*)(Nt16Args);
    const K kerns[3][2] = {{gemm16_nt_dma_kernel<WM, WN, TM, TN, MINB, NS, false, 0>, gemm16_nt_dma_kernel<WM, WN, TM, TN, MINB, NS, true, 0>},
                           {gemm16_nt_dma_kernel<WM, WN, TM, TN, MINB, NS, false, 16>, gemm16_nt_dma_kernel<WM, WN, TM, TN, MINB, NS, true, 16>},
                           {gemm16_nt_dma_kernel<WM, WN, TM, TN, MINB, NS, false, 32>, gemm16_nt_dma_kernel<WM, WN, TM, TN, MINB, NS, true, 32>}};
    ORX_ONCE_PER_DEVICE(ctx, {
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 2; ++b)
            ORX_HIP(hipFuncSetAttribute((const void*)kerns[a][b], hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    });
    const K kern = kerns[nts >= 0 && nts <= 2 ? nts : 0][tail ? 1 : 0];
    const unsigned nb = (unsigned)(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN));
    ORX_LAUNCH(ctx, kern, dim3(nb), dim3(64 * WM * WN), shm, g);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

template <int WM, int WN, int TM, int TN, int MINB, int PAD>
static int launch_nt(orx_ctx* ctx, const Nt16Args& g) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr size_t shm = (size_t)2 * (BM + BN) * (64 + PAD) * 2;
    auto kern = gemm16_nt_kernel<WM, WN, TM, TN, MINB, PAD>;
    ORX_ONCE_PER_DEVICE(ctx, ORX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)));
    const unsigned nb = (unsigned)(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN));
    ORX_LAUNCH(ctx, kern, dim3(nb), dim3(64 * WM * WN), shm, g);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

bool orx_gemm16_nt_ok(int64_t lda, int64_t ldb, int N, int K) { return lda % 8 == 0 && ldb % 8 == 0 && N >= 32 && K >= 8; }

// which tile configuration the [M, N] product takes (the relu-mask words of Nt16Args are laid out per configuration): 1 = 256 x 128 on eight
// wavefronts, 2 = 128 x 128, 3 = 128 x 64, each in its LDS-DMA form; 0 = any other form (forced tiles, register staging, wave-tile experiment): no masks.
// words_out: 64-bit words a mask of that product needs
int orx_gemm16_nt_config(orx_ctx* ctx, int M, int N, int64_t* words_out) {
    static const bool off = getenv("ORX_GEMM16_NO_MASK") != nullptr;
    static const int force = getenv("ORX_GEMM16_TILE") ? atoi(getenv("ORX_GEMM16_TILE")) : 0;
    static const int dma_env = getenv("ORX_GEMM16_DMA") ? atoi(getenv("ORX_GEMM16_DMA")) : 3;
    static const int wave_tile = getenv("ORX_GEMM16_WAVE_TILE") ? atoi(getenv("ORX_GEMM16_WAVE_TILE")) : 64;
    if (words_out) *words_out = 0;
    if (off || force != 0 || dma_env != 3 || wave_tile != 64) return 0;
    const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
    auto blocks = [&](int bm, int bn) { return (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    int cfg, bm, bn, nt;
    if (blocks(256, 128) >= cus) { cfg = 1; bm = 256; bn = 128; nt = 512; }
    else if (blocks(128, 128) >= 2 * cus) { cfg = 2; bm = 128; bn = 128; nt = 256; }
    else { cfg = 3; bm = 128; bn = 64; nt = 256; }
    if (words_out) *words_out = blocks(bm, bn) * nt;
    return cfg;
}

int orx_launch_gemm16_nt(orx_ctx* ctx, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                         void* C16, int64_t ldc16, const float* bias, int M, int N, int K, int act,
                         const float* actY, const void* actY16, int64_t ldy, int act_y, ColPart* gbp,
                         unsigned long long* mask_out, const unsigned long long* mask_in) {
    if (M == 0 || N == 0) return ORX_OK;
    float* gb = gbp ? gbp->parts : nullptr;
    ORX_ARG(lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)A16 | (uintptr_t)B16) & 15) == 0, "gemm16_nt: operands need 16-byte rows");
    ProfScope ps(ctx, ORX_K_GEMM);
    Nt16Args g{(const _Float16*)A16, lda, (const _Float16*)B16, ldb, C, ldc, (_Float16*)C16, ldc16, bias, M, N, K, act,
               actY, (const _Float16*)actY16, ldy, act_y, gb, mask_out, mask_in};
    ORX_ARG((mask_out == nullptr && mask_in == nullptr) || ((int64_t)M * lda < (1LL << 30) && (int64_t)N * ldb < (1LL << 30) && orx_gemm16_nt_config(ctx, M, N, nullptr) != 0),
            "gemm16_nt: relu masks need the LDS-DMA tile forms");
    // the largest tile that still gives every CU a workgroup (256 CUs)
    auto blocks = [&](int bm, int bn) { return (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    static const int force = getenv("ORX_GEMM16_TILE") ? atoi(getenv("ORX_GEMM16_TILE")) : 0;
    const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
    // (the 4-wavefront configurations want two workgroups per CU: with one, every load and barrier latency of the short K
    // loops of the narrow layers is exposed -- a fused 8192 x 512 x 256 product took 18 us on 256 tiles of 128 x 128)
    // ORX_GEMM16_DMA: 0 = the register-staged kernels, 2 / 3 = LDS-DMA staging with that many stages (default 3 for the 256 x 128 tile)
    static const int dma_env = getenv("ORX_GEMM16_DMA") ? atoi(getenv("ORX_GEMM16_DMA")) : 3;
    const int dma = ((int64_t)M * lda < (1LL << 30) && (int64_t)N * ldb < (1LL << 30)) ? dma_env : 0;      // (32-bit byte offsets from A and B)
    // ORX_GEMM16_WAVE_TILE=128: the 256 x 128 tile on FOUR wavefronts of 128 x 64 instead of eight of 64 x 64.  The main loop of these
    // products is bound by LDS bandwidth, not by the MFMA pipes (fragment reads 8.7 us + DMA writes 3.3 us of a 13.4 us loop at
    // 8192 x 1024 x 1024, against ~5 us of products): what a wavefront reads per K step is (rows of A + rows of B) of ITS tile, and
    // 4 x (128 + 64) rows are 25 % fewer than 8 x (64 + 64).
    static const int wave_tile = getenv("ORX_GEMM16_WAVE_TILE") ? atoi(getenv("ORX_GEMM16_WAVE_TILE")) : 64;
    if (force == 1 || (force == 0 && blocks(256, 128) >= cus)) {
        if (gbp) gbp->P = (M + 255) / 256;
        if (dma == 3 && wave_tile == 128) return launch_nt_dma<2, 2, 8, 4, 1, 3>(ctx, g);
        if (dma == 3) return launch_nt_dma<4, 2, 4, 4, 1, 3>(ctx, g);
        if (dma == 2) return launch_nt_dma<4, 2, 4, 4, 1, 2>(ctx, g);
        return launch_nt<4, 2, 4, 4, 1, 16>(ctx, g);
    }
    if (gbp) gbp->P = (M + 127) / 128;
    if (force == 2 || (force == 0 && blocks(128, 128) >= 2 * cus)) return dma ? launch_nt_dma<2, 2, 4, 4, 2, 2>(ctx, g) : launch_nt<2, 2, 4, 4, 2, 8>(ctx, g);
    if (dma == 3) return launch_nt_dma<2, 2, 4, 2, 2, 3>(ctx, g);
    if (dma == 2) return launch_nt_dma<2, 2, 4, 2, 2, 2>(ctx, g);
    return launch_nt<2, 2, 4, 2, 2, 16>(ctx, g);
}

// ------------------------------------------------------------------------------------------------ weight gradient
__device__ __forceinline__ h4 lds_tr_read(const _Float16* p) {
    // lane i of a 16-lane group supplies &blk[i / 4][4 * (i % 4)] of a [4][16] fp16 block and receives column i
    s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p);
    return __builtin_bit_cast(h4, v);
}

// one slice of one tile in the split-K workspace: 128 x 128 floats + 256 bytes, so that the S slices of a tile (read together by
// the reduce kernel) do not all start on the same memory channel
constexpr size_t SLAB_STRIDE = ORX_SLAB_STRIDE;

struct Tn16Args {
    const _Float16* A; int64_t lda;      // [K][lda]: rows = reduction index (samples), M columns used
    const _Float16* B; int64_t ldb;      // [K][ldb]
    float* C; int64_t ldc;               // S == 1: C += tile
    float* slab;                         // S > 1: [tile][S][BM * BN]
    int M, N, K, kchunk;
    float out_scale;                     // S == 1: C += out_scale * tile (the slabs are scaled by slab_reduce_kernel)
};

// k order inside a 32-deep MFMA step: lane group q holds rows 4q .. 4q + 3 and 16 + 4q .. 16 + 4q + 3 of the stage (the same
// for both operands, so the sum is a permutation of the same terms); with a 288-byte row pitch the two 16-lane groups of
// an LDS cycle then sit on different halves of the banks.
template <int WM, int WN, int TM, int TN, int MINB>
__global__ __launch_bounds__(64 * WM * WN, MINB) void gemm16_tn_kernel(Tn16Args g) {
    constexpr int BK = 64, BM = WM * TM * 16, BN = WN * TN * 16, NT = 64 * WM * WN;
    constexpr int LDM = BM + 16, LDN = BN + 16;
    constexpr int CA = BM / 8, CB = BN / 8;
    constexpr int NA = BK * CA / NT, NB = BK * CB / NT;
    static_assert(BK * CA % NT == 0 && BK * CB % NT == 0, "tile / thread mismatch");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds16[];
    constexpr int STAGE = BK * (LDM + LDN);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM, nt = ntn * ntm;
    // an XCD takes whole K slices: the tiles of a slice share its rows of A and B, different slices share nothing
    const int t = xcd_slot(blockIdx.x, gridDim.x);
    const int bz = t / nt, tile = t - bz * nt;
    const int bm = (tile / ntn) * BM, bn = (tile % ntn) * BN;
    const int wm = (wave / WN) * TM * 16, wn = (wave % WN) * TN * 16;
    const int i16 = lane & 15, q = lane >> 4;
    const int kbeg = bz * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
    h8 ra[NA], rb[NB];
    h8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (_Float16)0.0f;
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
    const int tr_a = (q * 4 + i16 / 4) * LDM + (i16 % 4) * 4, tr_b = (q * 4 + i16 / 4) * LDN + (i16 % 4) * 4;
    auto compute = [&](const _Float16* S) {
        const _Float16* SA = S; const _Float16* SB = S + BK * LDM;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            h8 a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const h4 lo = lds_tr_read(SA + (kk * 32) * LDM + tr_a + wm + mi * 16), hi = lds_tr_read(SA + (kk * 32 + 16) * LDM + tr_a + wm + mi * 16);
                a[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const h4 lo = lds_tr_read(SB + (kk * 32) * LDN + tr_b + wn + ni * 16), hi = lds_tr_read(SB + (kk * 32 + 16) * LDN + tr_b + wn + ni * 16);
                b[ni] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
    };
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        gload(kbeg);
        lstore(lds16);
        if (nk > 1) gload(kbeg + BK);
        __syncthreads();
        for (int s = 0; s < nk; ++s) {
            _Float16* cur = lds16 + (s & 1) * STAGE; _Float16* nxt = lds16 + ((s + 1) & 1) * STAGE;
            if (s + 1 < nk) lstore(nxt);
            if (s + 2 < nk) gload(kbeg + (s + 2) * BK);
            compute(cur);
            __syncthreads();
        }
    }
    const int S = gridDim.x / nt;
    if (S > 1) {
        float* mine = g.slab + ((size_t)tile * S + bz) * SLAB_STRIDE;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                if (bm + wm + mi * 16 + i16 < g.M)               // (rows beyond M are never read: a 13 x 512 gradient stores 13 of a tile's 128 rows)
                    store_sc1(reinterpret_cast<f32x4*>(mine + (wm + mi * 16 + i16) * BN + wn + ni * 16 + q * 4), acc[mi][ni]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // (the write-through stores are acknowledged before the wavefront ends)
        return;
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = bm + wm + mi * 16 + i16;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = bn + wn + ni * 16 + q * 4;
            if (row >= g.M) continue;
            const float v[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) if (col + e < g.N) g.C[(int64_t)row * g.ldc + col + e] += v[e] * g.out_scale;
        }
    }
}

// ---- the weight gradient with LDS-DMA staging (see gemm16_nt_dma_kernel): a stage holds 64 k-rows x 128 columns of each operand,
// 256-byte rows without padding.  The transposing reads take 32 contiguous bytes of 8 different rows per LDS cycle: row r keeps
// the 16-byte chunk c of its 256 bytes at slot c ^ (2 * (r & 7)) -- chunk PAIRS stay adjacent and the eight rows of a cycle sit on
// eight different quarters of the banks (what the 288-byte pitch does for the register-staged kernel).
// TAIL: a tile reaches beyond a leading dimension or the slice ends inside a stage (zero chunk through per-lane addresses).
// KG = 2 (round 6): a workgroup of EIGHT wavefronts, two K groups of four -- each group runs the schedule below on its own half of the slice's
// k-rows with its own LDS stages, and at the end group 1 hands its accumulators to group 0 through LDS.  One workgroup per CU then keeps two
// wavefronts on every SIMD (the four-wavefront form leaves each SIMD one wavefront: nothing to issue while it waits for its fragments) WITHOUT
// the second set of split-K slices that two four-wavefront workgroups per CU cost (their slabs are written here and read again by the optimizer
// launch).  The sum of a slice becomes (first half) + (second half): another fixed order.
template <int NS, bool TAIL, int KG = 1>
__device__ __forceinline__ void gemm16_tn_dma_body(const Tn16Args& g, const int bid, const int nblk) {
    constexpr int BK = 64, BM = 128, BN = 128, NT = 256, TM = 4, TN = 4, WN = 2;
    constexpr int NA = BK * (BM / 8) / NT, NB = BK * (BN / 8) / NT, NL = NA + NB;      // 4 + 4 requests per wavefront and stage
    extern __shared__ __attribute__((aligned(16))) _Float16 lds16_all[];
    constexpr int STAGE = BK * (BM + BN);                                      // halves
    const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = KG == 2 ? (wave_all >> 2) : 0, wave = wave_all & 3;          // K group, wavefront inside it
    _Float16* const lds16 = lds16_all + kg * (NS * STAGE);
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM, nt = ntn * ntm;
    const int t = xcd_slot(bid, nblk);
    const int bz = t / nt, tile = t - bz * nt;
    const int bm = (tile / ntn) * BM, bn = (tile % ntn) * BN;
    const int wm = (wave / WN) * TM * 16, wn = (wave % WN) * TN * 16;
    const int i16 = lane & 15, q = lane >> 4;
    const int kbeg_s = bz * g.kchunk, kend = min(g.K, kbeg_s + g.kchunk);
    // (KG = 2: both groups run the same number of K steps -- the barriers are the workgroup's -- group 1's steps beyond the slice read zeros)
    const int nk_all = kend > kbeg_s ? (kend - kbeg_s + BK - 1) / BK : 0;
    const int nk_g = KG == 2 ? (nk_all + 1) / 2 : nk_all;
    const int kbeg = kbeg_s + kg * nk_g * BK;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc[i][j].x = acc[i][j].y = acc[i][j].z = acc[i][j].w = 0.0f; }
    // request i of a wavefront covers LDS chunks c = 64 * wave + lane + 256 * i: row c / 16, slot c % 16
    unsigned off[NL]; int kr[NL], col[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int c = wave * 64 + lane + NT * (i < NA ? i : i - NA), r = c >> 4, ch = (c & 15) ^ (2 * (r & 7));
        kr[i] = r; col[i] = (i < NA ? bm : bn) + ch * 8;
        off[i] = (unsigned)(((int64_t)r * (i < NA ? g.lda : g.ldb) + col[i]) * 2);
    }
    const int lda = (int)g.lda, ldb = (int)g.ldb;
    auto issue_one = [&](int k0, _Float16* S, int i) {                       // the k-rows k0 .. k0 + 63 of the slice
        const int j = i < NA ? i : i - NA;
        _Float16* dst = S + (i < NA ? 0 : BK * BM) + (NT * j + wave * 64) * 8;
        const char* sb = reinterpret_cast<const char*>(i < NA ? g.A : g.B) + (size_t)k0 * (i < NA ? lda : ldb) * 2;
        if (TAIL) {
            const char* p = (k0 + kr[i] < kend && col[i] < (i < NA ? lda : ldb)) ? sb + off[i] : reinterpret_cast<const char*>(g_zero_chunk);
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)dst, 16, 0, 0);
        } else {
            const unsigned m0v = (unsigned)(uintptr_t)(lptr_t)dst;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off[i]), "s"(sb), "s"(m0v) : "memory");
        }
    };
    auto issue = [&](int k0, _Float16* S) {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_one(k0, S, i);
    };
    // transposing reads: lane i of a 16-lane group supplies &blk[i / 4][4 * (i % 4)] of a [4 rows][16 columns] block; lane group q
    // takes rows 4q .. 4q + 3 and 16 + 4q .. (the k order inside a 32-deep product, the same for both operands)
    const int rr = q * 4 + i16 / 4, x = rr & 7;
    int fa[TM], fb[TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) fa[mi] = rr * BM + ((((wm >> 4) + mi) ^ x) * 16) + (i16 % 4) * 4;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) fb[ni] = BK * BM + rr * BN + ((((wn >> 4) + ni) ^ x) * 16) + (i16 % 4) * 4;
    h8 a0[TM], b0[TN], a1[TM], b1[TN];
    auto fetch = [&](const _Float16* S, int kk, h8 (&a)[TM], h8 (&b)[TN]) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const h4 lo = lds_tr_read(S + fa[mi] + (kk * 32) * BM), hi = lds_tr_read(S + fa[mi] + (kk * 32 + 16) * BM);
            a[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const h4 lo = lds_tr_read(S + fb[ni] + (kk * 32) * BN), hi = lds_tr_read(S + fb[ni] + (kk * 32 + 16) * BN);
            b[ni] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    auto arrived = [&](h8 (&a)[TM], h8 (&b)[TN]) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) asm volatile("" : "+v"(a[mi]));
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) asm volatile("" : "+v"(b[ni]));
    };
    const int nk = nk_g;
    int cur = 0;
    auto step = [&](auto dma_c, auto more_c, auto inflight_c, int s) {          // (the schedule of gemm16_nt_dma_kernel)
        constexpr bool DMA = decltype(dma_c)::value, MORE = decltype(more_c)::value, INFLIGHT = decltype(inflight_c)::value;
        _Float16* S = lds16 + cur * STAGE;
        cur = cur + 1 == NS ? 0 : cur + 1;
        fetch(S, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        arrived(a1, b1);
        if (MORE) { if (NS == 3 && INFLIGHT) wait_vm<NL>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        if (MORE) fetch(lds16 + cur * STAGE, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        const int k0 = kbeg + (s + NS) * BK;
        constexpr int EVERY = TM * TN / NL;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
                const int done = mi * TN + ni + 1;
                if (DMA && done % EVERY == 0 && done / EVERY <= NL) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_one(k0, S, done / EVERY - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        if (MORE) arrived(a0, b0);
    };
    if (nk > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) if (s < nk) issue(kbeg + s * BK, lds16 + s * STAGE);
        if (nk > 2 && NS == 3) wait_vm<2 * NL>(); else if (nk > 1) wait_vm<NL>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        fetch(lds16, 0, a0, b0); arrived(a0, b0);
        using T = std::true_type; using F = std::false_type;
        int s = 0;
        for (; s + NS < nk; ++s) step(T(), T(), T(), s);
        if (NS == 3 && s + 2 < nk) { step(F(), T(), T(), s); ++s; }
        if (s + 1 < nk) { step(F(), T(), F(), s); ++s; }
        if (s < nk) step(F(), F(), F(), s);
    }
    if (KG == 2) {
        // group 1's accumulators go to group 0 through LDS (the stages are free: 2 x NS x 32 KB >= 128 rows x 132 floats; the 4-float pad
        // spreads the sixteen rows of a store over the banks)
        constexpr int XLD = BN + 4;
        float* xb = reinterpret_cast<float*>(lds16_all);
        __syncthreads();
        if (kg == 1) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    *reinterpret_cast<f32x4*>(xb + (wm + mi * 16 + i16) * XLD + wn + ni * 16 + q * 4) = acc[mi][ni];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] += *reinterpret_cast<const f32x4*>(xb + (wm + mi * 16 + i16) * XLD + wn + ni * 16 + q * 4);
    }
    const int S = nblk / nt;
    if (S > 1) {
        float* mine = g.slab + ((size_t)tile * S + bz) * SLAB_STRIDE;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                if (bm + wm + mi * 16 + i16 < g.M)               // (rows beyond M are never read: a 13 x 512 gradient stores 13 of a tile's 128 rows)
                    store_sc1(reinterpret_cast<f32x4*>(mine + (wm + mi * 16 + i16) * BN + wn + ni * 16 + q * 4), acc[mi][ni]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // (the write-through stores are acknowledged before the wavefront ends)
        return;
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = bm + wm + mi * 16 + i16;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int c0 = bn + wn + ni * 16 + q * 4;
            if (row >= g.M) continue;
            const float v[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c0 + e < g.N) g.C[(int64_t)row * g.ldc + c0 + e] += v[e] * g.out_scale;
        }
    }
}

template <int MINB, int NS, bool TAIL>
__global__ __launch_bounds__(256, MINB) void gemm16_tn_dma_kernel(Tn16Args g) {
    gemm16_tn_dma_body<NS, TAIL>(g, (int)blockIdx.x, (int)gridDim.x);
}

// eight wavefronts, two K groups (see the body): one workgroup per CU, two stages per group = 128 KB of LDS
template <bool TAIL>
__global__ __launch_bounds__(512, 1) void gemm16_tn_dma_kg2_kernel(Tn16Args g) {
    gemm16_tn_dma_body<2, TAIL, 2>(g, (int)blockIdx.x, (int)gridDim.x);
}

// ---- grouped launch (round 6): the weight gradient X16^T dZ16 (tn) and the input gradient dZ16 W16^T (nt) of ONE layer -- independent
// given dZ -- in one grid, for the layers whose products cannot fill the chip on their own (a [8192, 256] output is 64 tiles of
// 256 x 128 on 256 CUs; each of the two launches cost 7-10 us for <= 1 GFLOP).  The weight-gradient workgroups come first (a split-K
// slice of 128 x 128 is the longer piece), both bodies are the kernels above: same tiles, same summation order, same bits.
// Two stages of LDS for the tn part (64 KB) and three for the 128 x 64 nt tile (72 KB): two workgroups per CU.
struct Group16Args { Tn16Args tn; Nt16Args nt; int n_tn, n_nt; };
template <bool TN_TAIL, bool NT_TAIL, int NTS>
__global__ __launch_bounds__(256, 2) void gemm16_group_kernel(Group16Args g) {
    const int b = (int)blockIdx.x;
    if (b < g.n_tn) gemm16_tn_dma_body<2, TN_TAIL>(g.tn, b, g.n_tn);
    else gemm16_nt_dma_body<2, 2, 4, 2, 3, NT_TAIL, NTS == 2 ? 32 : NTS == 1 ? 16 : 0>(g.nt, b - g.n_tn, g.n_nt);
}

// C += sum over the S slices of every tile; one launch serves all layers of an MLP backward.  grid = (max tiles, 8 parts, layers)
template <int BM, int BN>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const SlabReduce* jobs, float out_scale) {
    const SlabReduce j = jobs[blockIdx.z];
    const int tile = blockIdx.x;
    if (tile >= j.tiles) return;
    const int by = tile / j.ntn, bx = tile - by * j.ntn;
    const float* base = j.slab + (size_t)tile * j.S * SLAB_STRIDE;
    for (int e = threadIdx.x + 256 * blockIdx.y; e < BM * BN / 4; e += 256 * gridDim.y) {
        const int r = (e * 4) / BN, c = (e * 4) % BN;
        const int row = by * BM + r, col = bx * BN + c;
        if (row >= j.M || col >= j.N) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= j.S; z += 8) {                       // 8 slices in flight, summed in slice order
            f32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)(z + u) * SLAB_STRIDE + e * 4));
#pragma unroll
            for (int u = 0; u < 8; ++u) v += w[u];
        }
        for (; z < j.S; ++z) v += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)z * SLAB_STRIDE + e * 4));
        v *= out_scale;
        float* p = j.C + (int64_t)row * j.ldc + col;
        if (col + 3 < j.N && (j.ldc & 3) == 0) { f32x4 o = *reinterpret_cast<f32x4*>(p); o += v; *reinterpret_cast<f32x4*>(p) = o; continue; }
        const float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (col + k < j.N) p[k] += o[k];
    }
}

// the split the launcher will use for a [M x N] gradient over K samples (slab floats needed = tiles * S * 128 * 128)
void orx_gemm16_tn_plan(orx_ctx* ctx, int M, int N, int K, int* S_out, int* tiles_out, int* kchunk_out) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
    // S depends on the SHAPE of the gradient only, never on the number of samples: the workspace and the reduce descriptors
    // are made once per model, and a later call with fewer samples (the last batch of an epoch) must write -- and the
    // reduce must add -- the same S slices; slices beyond the samples write zeros
    // ONE workgroup per CU (three LDS stages): half the slices of two per CU -- the slabs are written here and read again by the
    // optimizer launch, 146 -> 73 MB each way at the C5 shapes; the products take the same time (0.290 ms), the step 0.558 -> 0.543 ms.
    // ORX_GEMM16_TN_PER_CU=2 with ORX_GEMM16_TN_DMA=2: the two-per-CU form
    static const int per_cu = getenv("ORX_GEMM16_TN_PER_CU") ? atoi(getenv("ORX_GEMM16_TN_PER_CU")) : 1;
    // (round 6) gradients of <= 16 tiles used to take 32 slices to fill the chip: a 512 x 256 gradient (0.5 MB) left 16.8 MB of slabs for the
    // optimizer launch to read back.  Their weight-gradient workgroups now share a launch with the layer's input gradient
    // (gemm16_group_kernel), which fills the chip: fewer slices are enough (ORX_GEMM16_TN_SMALL_S; 32: the round-5 plan).
    // Measured at the C5 shapes on one box (profiles/r6_tn_slices.txt): 8 slices 499 us per step (the optimizer launch 33 -> 20 us, but the
    // grouped launches +3-4 us each and the 13 x 512 gradient, which has no input gradient to share a launch with, 8 -> 19 us), 16: 487,
    // 32: 493.  So: 16 for gradients of 5 .. 16 tiles; smaller ones keep 32 (their slabs are small, and rows beyond M are not stored).
    static const int small_s = getenv("ORX_GEMM16_TN_SMALL_S") ? atoi(getenv("ORX_GEMM16_TN_SMALL_S")) : 16;
    int S = std::max(1, std::min(32, (per_cu * cus) / tiles));
    if (tiles > 4 && tiles <= 16 && small_s >= 1) S = std::min(S, small_s);
    const int kchunk = std::max(64, (((K + S - 1) / S + 63) / 64) * 64);
    *S_out = S; *tiles_out = tiles; *kchunk_out = kchunk;
}

bool orx_gemm16_tn_ok(int64_t lda, int64_t ldb, int N) { return lda % 8 == 0 && ldb % 8 == 0 && N % 8 == 0; }

int orx_launch_gemm16_tn(orx_ctx* ctx, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                         float* slab, int M, int N, int K, float out_scale) {
    if (M == 0 || N == 0 || K == 0) return ORX_OK;
    ORX_ARG(lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)A16 | (uintptr_t)B16) & 15) == 0, "gemm16_tn: operands need 16-byte rows");
    ProfScope ps(ctx, ORX_K_GEMM);
    int S, tiles, kchunk;
    orx_gemm16_tn_plan(ctx, M, N, K, &S, &tiles, &kchunk);
    ORX_ARG(S == 1 || slab != nullptr, "gemm16_tn: split-K needs a slab workspace");
    Tn16Args g{(const _Float16*)A16, lda, (const _Float16*)B16, ldb, C, ldc, slab, M, N, K, kchunk, out_scale};
    // ORX_GEMM16_TN_DMA: 0 = the register-staged kernel, 2 = LDS-DMA staging with two stages and two workgroups per CU,
    // 3 (default) = three stages, one workgroup per CU
    static const int dma = getenv("ORX_GEMM16_TN_DMA") ? atoi(getenv("ORX_GEMM16_TN_DMA")) : 4;
    if (dma == 4) {
        // 4 (default, round 6) = eight wavefronts in two K groups, one workgroup per CU (gemm16_tn_dma_kg2_kernel); a slice whose K steps do not
        // split evenly, or whose tiles reach beyond a leading dimension, takes the tail form
        const int nk_all = (std::min(K, kchunk) + 63) / 64;
        const bool tail = (K & 63) != 0 || kchunk % 64 != 0 || (nk_all & 1) != 0 || (K % kchunk) != 0 ||
                          (int64_t)((M + 127) / 128) * 128 > lda || (int64_t)((N + 127) / 128) * 128 > ldb;
        constexpr size_t shm = (size_t)2 * 2 * 64 * 256 * 2;
        static_assert(shm >= (size_t)128 * 132 * 4, "LDS of the accumulator exchange");
        auto kern = tail ? gemm16_tn_dma_kg2_kernel<true> : gemm16_tn_dma_kg2_kernel<false>;
        ORX_ONCE_PER_DEVICE(ctx, {
            ORX_HIP(hipFuncSetAttribute((const void*)gemm16_tn_dma_kg2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            ORX_HIP(hipFuncSetAttribute((const void*)gemm16_tn_dma_kg2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        });
        ORX_LAUNCH(ctx, kern, dim3((unsigned)(tiles * S)), dim3(512), shm, g);
        ORX_HIP(hipGetLastError());
        return ORX_OK;
    }
    if (dma == 2 || dma == 3) {
        const bool tail = (K & 63) != 0 || kchunk % 64 != 0 || (int64_t)((M + 127) / 128) * 128 > lda || (int64_t)((N + 127) / 128) * 128 > ldb;
        const size_t shm = (size_t)dma * 64 * 256 * 2;
        auto kern = dma == 2 ? (tail ? gemm16_tn_dma_kernel<2, 2, true> : gemm16_tn_dma_kernel<2, 2, false>)
                             : (tail ? gemm16_tn_dma_kernel<1, 3, true> : gemm16_tn_dma_kernel<1, 3, false>);
        ORX_ONCE_PER_DEVICE(ctx, {
            ORX_HIP(hipFuncSetAttribute((const void*)gemm16_tn_dma_kernel<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 256 * 2));
            ORX_HIP(hipFuncSetAttribute((const void*)gemm16_tn_dma_kernel<2, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 256 * 2));
            ORX_HIP(hipFuncSetAttribute((const void*)gemm16_tn_dma_kernel<1, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 256 * 2));
            ORX_HIP(hipFuncSetAttribute((const void*)gemm16_tn_dma_kernel<1, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 256 * 2));
        });
        ORX_LAUNCH(ctx, kern, dim3((unsigned)(tiles * S)), dim3(256), shm, g);
        ORX_HIP(hipGetLastError());
        return ORX_OK;
    }
    constexpr size_t shm = (size_t)2 * 64 * (128 + 16 + 128 + 16) * 2;
    auto kern = gemm16_tn_kernel<2, 2, 4, 4, 2>;
    ORX_ONCE_PER_DEVICE(ctx, ORX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)));
    ORX_LAUNCH(ctx, kern, dim3((unsigned)(tiles * S)), dim3(256), shm, g);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// Can the two backward products of a layer [B, out] -> [B, in] share a launch (gemm16_group_kernel)?  Only where the input gradient takes
// the 128 x 64 tile with three DMA stages and the weight gradient the DMA kernel -- the forms the grouped kernel carries -- i.e. where
// neither product fills the chip alone.  ORX_GEMM16_NO_GROUP=1: separate launches (A/B measurements, the bit-identity test).
bool orx_gemm16_group_ok(orx_ctx* ctx, int B, int in, int out, int64_t ldx16, int64_t ldw16) {
    static const bool off = getenv("ORX_GEMM16_NO_GROUP") != nullptr;
    static const int dma_nt = getenv("ORX_GEMM16_DMA") ? atoi(getenv("ORX_GEMM16_DMA")) : 3;
    static const int dma_tn = getenv("ORX_GEMM16_TN_DMA") ? atoi(getenv("ORX_GEMM16_TN_DMA")) : 3;
    static const bool forced = getenv("ORX_GEMM16_TILE") != nullptr;
    if (off || forced || dma_nt != 3 || (dma_tn != 2 && dma_tn != 3)) return false;
    const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
    auto blocks = [&](int bm, int bn) { return (int64_t)((B + bm - 1) / bm) * ((in + bn - 1) / bn); };
    if (blocks(256, 128) >= cus || blocks(128, 128) >= 2 * cus) return false;                   // (the larger tiles: the product fills the chip)
    // 32-bit byte offsets of the DMA requests: dZ16 [B][out], W16 [in][ldw16], X16 [B][ldx16]
    if ((int64_t)B * out >= (1LL << 29) || (int64_t)in * ldw16 >= (1LL << 29) || (int64_t)B * ldx16 >= (1LL << 29)) return false;
    return orx_gemm16_nt_ok(out, ldw16, in, out) && orx_gemm16_tn_ok(ldx16, out, out);
}

// dW [in][out] (+)= X16^T dZ16 (slabs when the plan splits K) and dX = dZ16 W16^T with the fused activation backward of the layer below,
// one launch: the arguments of orx_launch_gemm16_tn followed by those of orx_launch_gemm16_nt
int orx_launch_gemm16_group(orx_ctx* ctx, const void* X16, int64_t ldx, const void* dZ16, int64_t lddz, float* gW, int64_t ldgw, float* slab,
                            int in, int out, int B, float out_scale,
                            const void* W16, int64_t ldw, float* C, int64_t ldc, void* C16, int64_t ldc16,
                            const float* actY, const void* actY16, int64_t ldy, int act_y, ColPart* gbp, const unsigned long long* mask_in, int nt_cols) {
    if (B == 0 || in == 0 || out == 0) return ORX_OK;
    const int in_nt = nt_cols > 0 ? nt_cols : in;          // columns of the input-gradient product (>= in: the operand's zero padding rows)
    ORX_ARG(ldx % 8 == 0 && lddz % 8 == 0 && ldw % 8 == 0 && (((uintptr_t)X16 | (uintptr_t)dZ16 | (uintptr_t)W16) & 15) == 0, "gemm16_group: operands need 16-byte rows");
    ProfScope ps(ctx, ORX_K_GEMM);
    int S, tiles, kchunk;
    orx_gemm16_tn_plan(ctx, in, out, B, &S, &tiles, &kchunk);
    ORX_ARG(S == 1 || slab != nullptr, "gemm16_group: split-K needs a slab workspace");
    Group16Args g;
    g.tn = Tn16Args{(const _Float16*)X16, ldx, (const _Float16*)dZ16, lddz, gW, ldgw, slab, in, out, B, kchunk, out_scale};
    g.nt = Nt16Args{(const _Float16*)dZ16, lddz, (const _Float16*)W16, ldw, C, ldc, (_Float16*)C16, ldc16, nullptr, B, in_nt, out, 0,
                    actY, (const _Float16*)actY16, ldy, act_y, gbp ? gbp->parts : nullptr, nullptr, mask_in};
    if (gbp) gbp->P = (B + 127) / 128;
    g.n_tn = tiles * S;
    g.n_nt = ((B + 127) / 128) * ((in_nt + 63) / 64);
    const bool tn_tail = (B & 63) != 0 || kchunk % 64 != 0 || (int64_t)((in + 127) / 128) * 128 > ldx || (int64_t)((out + 127) / 128) * 128 > lddz;
    const bool nt_tail = (lddz & 63) != 0 || (ldw & 63) != 0;
    static const int nts_env = getenv("ORX_GEMM16_NTS") != nullptr ? atoi(getenv("ORX_GEMM16_NTS")) : 2;
    const int nts = nts_env >= 0 && nts_env <= 2 ? nts_env : 0;
    using K = void (*)(Group16Args);
    static const K kerns[3][2][2] = {
        {{gemm16_group_kernel<false, false, 0>, gemm16_group_kernel<false, true, 0>}, {gemm16_group_kernel<true, false, 0>, gemm16_group_kernel<true, true, 0>}},
        {{gemm16_group_kernel<false, false, 1>, gemm16_group_kernel<false, true, 1>}, {gemm16_group_kernel<true, false, 1>, gemm16_group_kernel<true, true, 1>}},
        {{gemm16_group_kernel<false, false, 2>, gemm16_group_kernel<false, true, 2>}, {gemm16_group_kernel<true, false, 2>, gemm16_group_kernel<true, true, 2>}}};
    constexpr size_t shm = 3 * (128 + 64) * 64 * 2;                // (>= the tn part's two stages of 64 x 256 halves)
    static_assert(shm >= 2 * 64 * 256 * 2, "LDS of the grouped launch");
    ORX_ONCE_PER_DEVICE(ctx, {
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 2; ++b) for (int c2 = 0; c2 < 2; ++c2)
            ORX_HIP(hipFuncSetAttribute((const void*)kerns[a][b][c2], hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    });
    ORX_LAUNCH(ctx, kerns[nts][tn_tail ? 1 : 0][nt_tail ? 1 : 0], dim3((unsigned)(g.n_tn + g.n_nt)), dim3(256), shm, g);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_slab_reduce(orx_ctx* ctx, const void* jobs_dev, int n_jobs, int max_tiles, float out_scale) {
    if (n_jobs == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_GEMM);                            // (part of the weight-gradient products' cost)
    ORX_LAUNCH(ctx, (slab_reduce_kernel<128, 128>), dim3((unsigned)max_tiles, 8, (unsigned)n_jobs), dim3(256), 0, (const SlabReduce*)jobs_dev, out_scale);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// fp16 copy of an fp32 activation [M][N] (row stride lds_) into [M][ld16], padding columns zeroed: the interaction kernels
// other than the MFMA one leave R in fp32 only
__global__ __launch_bounds__(256) void cast16_kernel(const float* src, int64_t lds_, _Float16* dst, int64_t ld16, int M, int N) {
    const int64_t total = (int64_t)M * ld16, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / ld16; const int c = (int)(i - r * ld16);
        dst[i] = c < N ? (_Float16)src[r * lds_ + c] : (_Float16)0.0f;
    }
}

int orx_launch_cast16(orx_ctx* ctx, const float* src, int64_t lds_, void* dst16, int64_t ld16, int M, int N) {
    if (M == 0 || N == 0) return ORX_OK;
    int64_t g = ((int64_t)M * ld16 + 255) / 256; if (g > 8192) g = 8192;
    ProfScope ps(ctx, ORX_K_GEMM);
    ORX_LAUNCH(ctx, cast16_kernel, dim3((unsigned)g), dim3(256), 0, src, lds_, (_Float16*)dst16, ld16, M, N);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ------------------------------------------------------------------------------------------------ the 1-unit head
// The top MLP ends in one unit (dlrm.py:57-61: ln_top[-1] == 1): its three "products" are a GEMV, an outer product and a
// weighted column sum over [B][K] -- 4 MB at K = 256 -- which the tiled kernels ran as three launches of ~15-23 us each.
// head_fwd: pred[b] = act(b0 + sum_k X16[b][k] * w16[k]).  head_bwd, one pass over X16 (= the fp16 copy of the layer below's
// output Y): dz[b] = dy[b] * act'(pred[b]); gb += sum dz; gW[k] += sum_b X16[b][k] * dz16[b]; and the input gradient with the
// layer below's activation backward folded in: dZb[b][k] = dz16[b] * w16[k] * act_b'(Y[b][k]) -> fp16 (and fp32 on request),
// gbb[k] += sum_b dZb[b][k].  Operands are rounded to fp16 where the MFMA path rounds them (dz, w, X).
// Half a wavefront per row (lane j owns chunks j, j + 32, ... of 8 halves), HEAD_ROWS rows per workgroup, one atomic per
// (workgroup, column) after an LDS reduction.
constexpr int HEAD_KMAX = 1024, HEAD_J = HEAD_KMAX / 256;

__global__ __launch_bounds__(256) void head_fwd_kernel(const _Float16* X, int64_t ldx, const _Float16* w, const float* bias, int act,
                                                       float* pred, int B, int K) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    float s = 0.0f;
    if (row < B) {
        for (int c = l32; c * 8 < K; c += 32) {
            const h8 x = *reinterpret_cast<const h8*>(X + (int64_t)row * ldx + c * 8), ww = *reinterpret_cast<const h8*>(w + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)x[e] * (float)ww[e];
        }
    }
    s = group_allreduce<32>(s);
    if (row < B && l32 == 0) {
        float v = s + bias[0];
        if (act == 1) v = fmaxf(v, 0.0f); else if (act == 2) v = 1.0f / (1.0f + __expf(-v));
        pred[row] = v;
    }
}

struct HeadBwdArgs {
    const _Float16* X; int64_t ldx;      // [B][ldx]: input of the head = fp16 copy of the layer below's output
    const _Float16* w;                   // [K] fp16 copy of the head's weights
    const float* dy; const float* pred;  // [B]
    int act, act_below;
    float* gW; float* gb;                // head gradients: partial sums per workgroup, [blocks][K] and [blocks]
    _Float16* dZ16; int64_t ld16;        // [B][ld16] gradient w.r.t. the layer below's pre-activation
    float* dZ32; int64_t ld32;           // optional
    float* gb_below;                     // [blocks][K] partial sums
    int B, K, rows_per_block;
    // round 6: the loss folded in (label != NULL; no dlrm_loss_kernel launch in the step) -- dy is formed here from pred and the label
    // (dlrm.py:72-73, :97-98; the formulas of dlrm_loss_kernel), the loss terms leave as one fp64 partial per workgroup
    const float* label; int bce; float thr, invB, gscale; double* loss_part;
    // ... and the head's FORWARD too (fwd_bias != NULL; no head_fwd_kernel launch in the step): the row is read once, pred[b] is formed as
    // head_fwd_kernel forms it (same lanes, same order of additions) and written to pred_out
    const float* fwd_bias; float* pred_out;
};

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
    __shared__ float red[8][HEAD_KMAX / 8][8 + 1];             // [half-wave][chunk][e]  (padded), used twice
    __shared__ float red_b[8];
    __shared__ double red_l[8];
    double lsum = 0.0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    const int hw = wave * 2 + half;                            // 8 half-wavefronts, each takes every 8th row of the block's slab
    const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.B, r0 + a.rows_per_block);
    float gw[HEAD_J][8], gbb[HEAD_J][8];
    h8 wv[HEAD_J];
#pragma unroll
    for (int j = 0; j < HEAD_J; ++j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { gw[j][e] = 0.0f; gbb[j][e] = 0.0f; wv[j][e] = (_Float16)0.0f; }
        const int c = l32 + 32 * j;
        if (c * 8 < a.K) wv[j] = *reinterpret_cast<const h8*>(a.w + c * 8);
    }
    float gb = 0.0f;
    h8 xn[HEAD_J];                                                 // the NEXT row of this half-wavefront, requested one iteration ahead (the forward
    auto request = [&](int row) {                                  // inside makes a row's chain load -> reduce -> gradients: nothing else hides its load)
#pragma unroll
        for (int j = 0; j < HEAD_J; ++j) {
            const int c = l32 + 32 * j;
            if (c * 8 < a.K && row < r1) xn[j] = *reinterpret_cast<const h8*>(a.X + (int64_t)row * a.ldx + c * 8);
        }
    };
    request(r0 + hw);
    for (int row = r0 + hw; row < r1; row += 8) {
        h8 xr[HEAD_J];                                             // the row: the forward's operand and the backward's
#pragma unroll
        for (int j = 0; j < HEAD_J; ++j) xr[j] = xn[j];
        request(row + 8);
        float p;
        if (a.fwd_bias != nullptr) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < HEAD_J; ++j) {
                if ((l32 + 32 * j) * 8 >= a.K) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)xr[j][e] * (float)wv[j][e];
            }
            s = group_allreduce<32>(s);
            float v = s + a.fwd_bias[0];
            if (a.act == 1) v = fmaxf(v, 0.0f); else if (a.act == 2) v = 1.0f / (1.0f + __expf(-v));
            if (l32 == 0) a.pred_out[row] = v;
            p = v;
        } else {
            p = a.pred[row];
        }
        float dz;
        if (a.label != nullptr) {
            float mask = 1.0f;
            if (a.thr > 0.0f && a.thr < 1.0f) {                     // dlrm.py:97-98
                mask = (p >= a.thr && p <= 1.0f - a.thr) ? 1.0f : 0.0f;
                p = fminf(fmaxf(p, a.thr), 1.0f - a.thr);
            }
            const float t = a.label[row];
            float g;
            if (!a.bce) {                                           // keras.losses.MeanSquaredError
                const float r = t - p;
                if (l32 == 0) lsum += (double)(r * r);
                g = 2.0f * (p - t) * a.invB;
            } else {                                                // keras.losses.BinaryCrossentropy (probabilities)
                const float eps = 1e-7f;
                const float pc = fminf(fmaxf(p, eps), 1.0f - eps);
                if (l32 == 0) lsum += -(double)(t * logf(pc + eps) + (1.0f - t) * logf(1.0f - pc + eps));
                const float inside = (p >= eps && p <= 1.0f - eps) ? 1.0f : 0.0f;
                g = -(t / (pc + eps) - (1.0f - t) / (1.0f - pc + eps)) * inside * a.invB;
            }
            dz = g * mask * a.gscale;
        } else {
            dz = a.dy[row];
        }
        dz = a.act == 1 ? (p > 0.0f ? dz : 0.0f) : (a.act == 2 ? dz * p * (1.0f - p) : dz);
        if (l32 == 0) gb += dz;
        const float dzh = (float)(_Float16)dz;
#pragma unroll
        for (int j = 0; j < HEAD_J; ++j) {
            const int c = l32 + 32 * j;
            if (c * 8 >= a.K) continue;
            const h8 x = xr[j];
            h8 o;
            float o32[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = (float)x[e];
                gw[j][e] += y * dzh;
                float d = dzh * (float)wv[j][e];
                d = a.act_below == 1 ? (y > 0.0f ? d : 0.0f) : (a.act_below == 2 ? d * y * (1.0f - y) : d);
                gbb[j][e] += d; o[e] = (_Float16)d; o32[e] = d;
            }
            *reinterpret_cast<h8*>(a.dZ16 + (int64_t)row * a.ld16 + c * 8) = o;
            if (a.dZ32) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a.dZ32[(int64_t)row * a.ld32 + c * 8 + e] = o32[e];
            }
        }
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int j = 0; j < HEAD_J; ++j) {
            const int c = l32 + 32 * j;
            if (c * 8 >= a.K) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) red[hw][c][e] = pass == 0 ? gw[j][e] : gbb[j][e];
        }
        if (pass == 0 && l32 == 0) { red_b[hw] = gb; red_l[hw] = lsum; }
        __syncthreads();
        for (int k = threadIdx.x; k < a.K; k += 256) {
            float s0 = 0.0f;
#pragma unroll
            for (int h = 0; h < 8; ++h) s0 += red[h][k >> 3][k & 7];
            (pass == 0 ? a.gW : a.gb_below)[(int64_t)blockIdx.x * a.K + k] = s0;
        }
        if (pass == 0 && threadIdx.x == 0) {
            float s = 0.0f;
            for (int h = 0; h < 8; ++h) s += red_b[h];
            a.gb[blockIdx.x] = s;
            if (a.loss_part != nullptr) {
                double t = 0.0;
                for (int h = 0; h < 8; ++h) t += red_l[h];
                a.loss_part[blockIdx.x] = t;
            }
        }
        __syncthreads();
    }
}

bool orx_head16_ok(int K, int64_t ldx) { return K % 8 == 0 && K >= 8 && K <= HEAD_KMAX && ldx % 8 == 0; }

int orx_launch_head_fwd(orx_ctx* ctx, const void* X16, int64_t ldx, const void* w16, const float* bias, int act, float* pred, int B, int K) {
    if (B == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_GEMM);
    ORX_LAUNCH(ctx, head_fwd_kernel, dim3((unsigned)((B + 7) / 8)), dim3(256), 0, (const _Float16*)X16, ldx, (const _Float16*)w16, bias, act, pred, B, K);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// sum of the loss partials head_bwd_kernel left, for all steps of a call: loss_out[s] = sum_b parts[s][b] / n_mean
__global__ __launch_bounds__(64) void head_loss_finish_kernel(const double* parts, int64_t stride, int n, double n_mean, double* loss_out) {
    const double* p = parts + (int64_t)blockIdx.x * stride;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) loss_out[blockIdx.x] = s / n_mean;
}

int orx_launch_head_loss_finish(orx_ctx* ctx, const double* parts, int64_t stride, int n, int64_t K, int64_t n_mean, double* loss_out) {
    if (K == 0) return ORX_OK;
    ORX_LAUNCH(ctx, head_loss_finish_kernel, dim3((unsigned)K), dim3(64), 0, parts, stride, n, (double)n_mean, loss_out);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_head_bwd_blocks(orx_ctx* ctx, int B) {
    const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
    int rows = std::max(32, (int)((B + cus - 1) / cus));
    rows = (rows + 7) / 8 * 8;
    return (B + rows - 1) / rows;
}

int orx_launch_head_bwd(orx_ctx* ctx, const void* X16, int64_t ldx, const void* w16, const float* dy, const float* pred, int act, int act_below,
                        ColPart* gW, ColPart* gb, void* dZ16, int64_t ld16, float* dZ32, int64_t ld32, ColPart* gb_below, int B, int K,
                        const HeadLoss* hl) {
    if (B == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_GEMM);
    const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
    int rows = std::max(32, (int)((B + cus - 1) / cus));           // <= one workgroup per CU, >= 32 rows each (partial rows = blocks)
    rows = (rows + 7) / 8 * 8;
    gW->P = gb->P = gb_below->P = (B + rows - 1) / rows;
    HeadBwdArgs a{(const _Float16*)X16, ldx, (const _Float16*)w16, dy, pred, act, act_below, gW->parts, gb->parts, (_Float16*)dZ16, ld16, dZ32, ld32, gb_below->parts, B, K, rows,
                  hl ? hl->label : nullptr, hl ? hl->bce : 0, hl ? hl->thr : 0.f, hl ? 1.0f / (float)hl->n_mean : 0.f, hl ? hl->gscale : 1.f, hl ? hl->loss_part : nullptr,
                  hl ? hl->fwd_bias : nullptr, hl ? hl->pred_out : nullptr};
    ORX_ARG(a.fwd_bias == nullptr || a.pred_out != nullptr, "head_bwd: the folded forward needs a place for pred");
    ORX_LAUNCH(ctx, head_bwd_kernel, dim3((unsigned)((B + rows - 1) / rows)), dim3(256), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
