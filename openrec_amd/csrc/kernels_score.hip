// All-item scorer of `Recommender.inference` on the fp32 matrix cores (gfx950):
//   BPR / WRMF   out[q, j] =  U[uid[q]] . V[j] + b[j]                      (bpr.py:39-43, wrmf.py:36-40)
//   UCML         out[q, j] = -|U[uid[q]] - V[j]|^2 + b[j]                   (ucml.py:50-53)
//   GMF          out[q, j] =  sum_e w[e] U[uid[q], e] V[j, e] + b[j]        (gmf.py:36-41: Dense(1, no bias) of the products)
// The reference materialises a [B, N, D] broadcast for UCML / GMF and a GEMM for BPR; the caller on the other side is
// eval_step (tf2_examples/bpr_citeulike.py:41-46), 1000 users x every item per call.
//
// v_mfma_f32_16x16x4_f32 multiplies exact fp32 products into an fp32 accumulator (an fmaf chain), so the scores are fp32
// dot products with a different summation order, nothing less.  A workgroup (4 wavefronts) owns 64 UW users -- wavefront w
// UW groups of 16, their rows in LDS once -- and walks a run of items in tiles of TI: the tile's rows are one contiguous
// piece of V, fetched with 16-byte loads into registers while the previous tile is multiplied, then laid into LDS at a row
// pitch of Dp + 4 floats (Dp = D rounded up to 16).  The sum over k does not care about its order, so within every block of
// 16 columns lane group q = lane / 16 takes columns 4q .. 4q+3 for four consecutive MFMA steps, for the item and the user
// operand alike: ONE ds_read_b128 per operand and four steps instead of four ds_read_b32.  The ITEMS are the M side of the
// product, so a lane ends up with four CONSECUTIVE items of one user: one 16-byte store per lane and 16 x 16 tile.
// UCML uses |u - v|^2 = |u|^2 + |v|^2 - 2 u.v with the norms taken from the same LDS tiles.  The output, nq x NI floats, is
// the traffic that bounds the kernel (4 GB for 1000 users x 1 M items against 256 MB of item rows).
#include "orx_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ScoreArgs {
    const float* U; const float* V; const float* b; const float* w;
    const int32_t* uid; int64_t nq; int64_t NU; int64_t NI;
    int D; int Dp; int TI; int64_t chunk;        // Dp = D rounded up to 16; TI items per LDS tile; `chunk` items per workgroup
    float* out; int* err;
};

template <int KIND, int NSUB, int UW, int KB>
__global__ __launch_bounds__(256) void score_mfma_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sc_lds[];
    const int D = a.D;
    constexpr int Dp = 16 * KB, pitch = Dp + 4;           // the k loop is unrolled completely: with a back-edge the compiler drains the
                                                          // next tile's global loads (s_waitcnt vmcnt(0)) before the first MFMA
    constexpr int TI = 16 * NSUB;
    constexpr int UB = 64 * UW;                             // users per workgroup: wavefront w takes UW groups of 16
    float* As = sc_lds;
    auto Bsel = [&](int i) -> float* { return sc_lds + (UB + i * TI) * pitch; };      // the two item-tile buffers
    float* un2 = sc_lds + (UB + 2 * TI) * pitch;           // [UB]   |u|^2       (UCML)
    float* vn2 = un2 + UB;                                 // [2][TI] |v|^2
    float* bt = vn2 + 2 * TI;                              // [2][TI] the tile's biases (a global load in the epilogue would
                                                           //         serialise every store behind a memory round trip)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q0 = (int64_t)blockIdx.y * UB;
    const int64_t jbeg = (int64_t)blockIdx.x * a.chunk;
    const int64_t jend = jbeg + a.chunk < a.NI ? jbeg + a.chunk : a.NI;
    // the 64 user rows (GMF: pre-multiplied by the Dense kernel), zero beyond nq / D
    for (int idx = tid; idx < UB * Dp; idx += 256) {
        const int r = idx / Dp, c = idx - r * Dp;
        float v = 0.0f;
        if (q0 + r < a.nq && c < D) {
            const int u = a.uid[q0 + r];
            if ((uint32_t)u >= (uint64_t)a.NU) *a.err = 1;
            else { v = a.U[(size_t)u * D + c]; if (KIND == 2) v *= a.w[c]; }
        }
        As[r * pitch + c] = v;
    }
    const bool vec = (D & 3) == 0;                          // the tile is a contiguous run of float4 in memory
    const int nv = vec ? (TI * D) / 4 : TI * Dp;            // elements (float4 or float) of a tile
    const int per = (nv + 255) / 256;                       // <= 8 float4 (TI = 64, D = 128) or 16 floats per thread
    f32x4 stage[8];
    float bstage = 0.f;
    auto fetch = [&](int64_t j0) {
        bstage = (tid < TI && j0 + tid < a.NI) ? a.b[j0 + tid] : 0.f;
        if (vec) {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.V + (size_t)j0 * D);
            const int64_t lim = (a.NI - j0) * (int64_t)(D / 4);   // float4 of the rows that exist
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = tid + 256 * k;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (k < per && e < nv && e < lim) z = src[e];
                stage[k] = z;
            }
        }
    };
    auto put = [&](float* B, int64_t j0, int buf) {
        if (tid < TI) bt[buf * TI + tid] = bstage;
        if (vec) {
            const int q4 = D / 4;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = tid + 256 * k;
                if (k < per && e < nv) {
                    const int r = e / q4, c = (e - r * q4) * 4;
                    *reinterpret_cast<f32x4*>(B + r * pitch + c) = stage[k];        // (pitch and c multiples of 4: 16-byte aligned)
                }
            }
        } else {
            for (int idx = tid; idx < TI * Dp; idx += 256) {
                const int r = idx / Dp, c = idx - r * Dp;
                B[r * pitch + c] = (j0 + r < a.NI && c < D) ? a.V[(size_t)(j0 + r) * D + c] : 0.0f;
            }
        }
    };
    if (vec && Dp > D) for (int idx = tid; idx < 2 * TI * (Dp - D); idx += 256) {      // columns D .. Dp-1 of both tile buffers stay zero
        const int r = idx / (Dp - D), c = D + idx % (Dp - D);
        Bsel(0)[r * pitch + c] = 0.0f;
    }
    fetch(jbeg);
    put(Bsel(0), jbeg, 0);
    __syncthreads();
    if (KIND == 1) {
        for (int r = tid; r < UB; r += 256) { float s = 0.f; for (int c = 0; c < Dp; ++c) { const float x = As[r * pitch + c]; s += x * x; } un2[r] = s; }
        if (tid < TI) { const int r = tid; float s = 0.f; for (int c = 0; c < Dp; ++c) { const float x = Bsel(0)[r * pitch + c]; s += x * x; } vn2[r] = s; }
        __syncthreads();
    }
    const float* urow = As + (16 * UW * wave + (lane & 15)) * pitch + 4 * (lane >> 4);   // this lane's first user row, its 4 columns of a block
    int t = 0;
    for (int64_t j0 = jbeg; j0 < jend; j0 += TI, ++t) {
        const float* B = Bsel(t & 1);
        const bool more = j0 + TI < jend;
        if (more) fetch(j0 + TI);
        f32x4 acc[UW][NSUB];
#pragma unroll
        for (int g = 0; g < UW; ++g)
#pragma unroll
            for (int s = 0; s < NSUB; ++s) acc[g][s] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* irow = B + (lane & 15) * pitch + 4 * (lane >> 4);
#pragma unroll
        for (int kb = 0; kb < Dp; kb += 16) {
            f32x4 uv[UW], iv[NSUB];
#pragma unroll
            for (int g = 0; g < UW; ++g) uv[g] = *reinterpret_cast<const f32x4*>(urow + g * 16 * pitch + kb);
#pragma unroll
            for (int s = 0; s < NSUB; ++s) iv[s] = *reinterpret_cast<const f32x4*>(irow + s * 16 * pitch + kb);
            // UW x NSUB independent accumulators, interleaved: consecutive MFMAs never wait for each other's result
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < UW; ++g)
#pragma unroll
                    for (int s = 0; s < NSUB; ++s) acc[g][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(iv[s][c], uv[g][c], acc[g][s], 0, 0, 0);
        }
        // acc[g][s][r] = score(item j0 + 16 s + 4 (lane / 16) + r, user q0 + 16 (UW wave + g) + lane % 16)
#pragma unroll
        for (int g = 0; g < UW; ++g) {
            const int ul = 16 * (UW * wave + g) + (lane & 15);
            const int64_t q = q0 + ul;
            if (q >= a.nq) continue;
            const float un = KIND == 1 ? un2[ul] : 0.f;
            float* orow = a.out + (size_t)q * a.NI;
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                const int64_t j = j0 + 16 * s + 4 * (lane >> 4);
                f32x4 v = acc[g][s];
                if (j + 3 < jend && ((a.NI | j) & 3) == 0) {           // four consecutive items, 16-byte aligned in the output row
                    const f32x4 bj = *reinterpret_cast<const f32x4*>(bt + (t & 1) * TI + 16 * s + 4 * (lane >> 4));
                    if (KIND == 1) { const f32x4 vn = *reinterpret_cast<const f32x4*>(vn2 + (t & 1) * TI + 16 * s + 4 * (lane >> 4)); v = 2.0f * v - un - vn; }
                    *reinterpret_cast<f32x4*>(orow + j) = v + bj;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (j + r < jend) {
                            float x = v[r];
                            if (KIND == 1) x = 2.0f * x - un - vn2[(t & 1) * TI + 16 * s + 4 * (lane >> 4) + r];
                            orow[j + r] = x + bt[(t & 1) * TI + 16 * s + 4 * (lane >> 4) + r];
                        }
                    }
                }
            }
        }
        if (more) {
            put(Bsel((t + 1) & 1), j0 + TI, (t + 1) & 1);
            if (KIND == 1) {
                __syncthreads();
                if (tid < TI) { float s = 0.f; const float* Bn = Bsel((t + 1) & 1); for (int c = 0; c < Dp; ++c) { const float x = Bn[tid * pitch + c]; s += x * x; } vn2[((t + 1) & 1) * TI + tid] = s; }
            }
        }
        __syncthreads();
    }
}

// returns ORX_OK and *launched = true when the MFMA scorer took the job (it needs D + tiles to fit the LDS)
int orx_launch_score_mfma(orx_ctx* ctx, const float* U, const float* V, const float* b, const float* w, const int32_t* uid,
                          int64_t nq, int64_t NU, int64_t NI, int D, int kind, float* out, bool* launched) {
    *launched = false;
    ScoreArgs a;
    a.U = U; a.V = V; a.b = b; a.w = w; a.uid = uid; a.nq = nq; a.NU = NU; a.NI = NI; a.D = D; a.Dp = 0;
    a.out = out; a.err = ctx->d_err;
    // k blocks of 16 columns (a template parameter: 1, 2, 4, 8 or 16), users per workgroup 64 UW, items per tile 16 NSUB.
    // Every workgroup streams its run of item rows from L2 / HBM, so the item table is read nq / (64 UW) times: 128 users per
    // workgroup halve that traffic (it equals the output's at 64) where there are that many users.
    int KB = 1;
    while (16 * KB < D) KB *= 2;
    if (KB > 16) return ORX_OK;
    a.Dp = 16 * KB;
    const int pitch = a.Dp + 4;
    static const char* uw_env = getenv("ORX_SCORE_UW");
    const int UW = KB == 16 ? 1 : (uw_env ? (atoi(uw_env) >= 2 ? 2 : 1) : (nq > 64 ? 2 : 1));
    const int TI = KB == 16 ? 32 : 64;
    a.TI = TI;
    if ((D & 3) == 0 && (TI * D) / 4 > 8 * 256) return ORX_OK;      // (the register stage of the vector path: never with these tiles)
    const int64_t nqt = (nq + 64 * UW - 1) / (64 * UW);
    // items per workgroup: enough workgroups to fill the chip, few enough to amortise the user rows
    int64_t chunk = (NI * nqt + 1023) / 1024;
    chunk = ((chunk + TI - 1) / TI) * TI;
    if (chunk < 4 * TI) chunk = 4 * TI;
    if (chunk > 4096) chunk = 4096;
    a.chunk = chunk;
    const size_t lds = ((size_t)(64 * UW + 2 * TI) * pitch + 64 * UW + 4 * TI) * sizeof(float);
    const dim3 g((unsigned)((NI + chunk - 1) / chunk), (unsigned)nqt);
#define ORX_SC(K, N, W, B) do { \
        ORX_ONCE_PER_DEVICE(ctx, ORX_HIP(hipFuncSetAttribute((const void*)score_mfma_kernel<K, N, W, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024))); \
        ORX_LAUNCH(ctx, (score_mfma_kernel<K, N, W, B>), g, dim3(256), lds, a); } while (0)
#define ORX_SCW(K, B) do { if (UW == 2) ORX_SC(K, 4, 2, B); else ORX_SC(K, 4, 1, B); } while (0)
#define ORX_SCK(K) do { switch (KB) { case 1: ORX_SCW(K, 1); break; case 2: ORX_SCW(K, 2); break; case 4: ORX_SCW(K, 4); break; \
                                      case 8: ORX_SCW(K, 8); break; default: ORX_SC(K, 2, 1, 16); break; } } while (0)
    if (kind == 0) ORX_SCK(0); else if (kind == 1) ORX_SCK(1); else ORX_SCK(2);
#undef ORX_SCK
#undef ORX_SCW
#undef ORX_SC
    ORX_HIP(hipGetLastError());
    *launched = true;
    return ORX_OK;
}
