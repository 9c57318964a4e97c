// EXPERIMENT (round 5, not part of the library: profiles/r5_exp_chain.txt has the verdict).
// The DLRM bottom MLP as ONE launch per direction (ORX_DLRM_FP16_MLP; recommenders/dlrm.py:76-79,87, modules/multi_layer_perceptron.py:5-18).
//
// At the C5 batch the bottom MLP (13-512-256-128, 8.4 GFLOP of the step's 116) cost ten launches of 7-10 us each: every one of its
// products is a few hundred MFLOP per CU-microsecond short of mattering, and what the launches paid for was launch + first tile +
// write-back.  A sample's way through an MLP touches no other sample, so here a workgroup OWNS 32 samples and walks them through all the
// layers: the activations of the 32 rows stay in LDS (fp16, [32][width + 8]: the pitch puts the 16 rows of a fragment read on
// different banks), a wavefront owns whole 16-column tiles of a layer's output for all 32 rows, and its B fragments -- the weights --
// come straight from global memory (L2: the MLP's kernels are 340 KB) into registers, four 32-deep K steps ahead: no wavefront shares
// a weight fragment with another one, so staging them through LDS would buy nothing.
//   forward : X (fp32 rows) -> fp16 -> [ Y = act(X W + b) ] per layer; by-products: the fp16 copies of the input and of every layer's
//             output that the backward pass reads, fp32 outputs where somebody wants them (the last layer: slot F-1 of Z).
//   backward: dY (fp32 slice of dZ, times the loss scale) -> dZ_L = dY * act'(Y_L) -> [ dZ_{l-1} = (dZ_l W_l^T) * act'(Y_{l-1}) ]; by-products:
//             the fp16 copy of every dZ_l (operand of the weight-gradient products) and the bias gradients' partial rows (one per
//             workgroup: colparts_reduce_kernel adds them in order).
// Both are the same kernel: a stage is `out = epilogue(in * W^T)` with W [N][K] K-contiguous (forward: the transposed fp16 kernel,
// backward: the fp16 kernel itself).
#include "orx_device.h"

// a whole MLP per launch on fp16-resident kernels, 32 samples per workgroup (kernels_chain16.hip): stage s is
// out = epilogue(in * W^T), W [N][ldw] K-contiguous; forward epilogue = + bias, activation; backward = activation backward with the fp16
// activations M16 + the column sums (partial row `workgroup` of gb)
constexpr int ORX_CHAIN_MAX = 6;
struct ChainStage {
    const _Float16* W; int ldw, N, K;
    const float* bias; int act;
    const _Float16* M16; int ldm, act_y;
    float* gb;
    _Float16* Y16; int ldy16;            // fp16 copy of the stage's output (optional)
    float* Y32; int64_t ldy32;           // fp32 copy (optional)
};
struct ChainArgs {
    int B, n_stages, pitch, backward;
    const float* X32; int64_t ldx; int K0; float xscale;       // the input rows
    const float* XM32; int64_t ldxm; int x_act; float* xgb;    // backward: activation backward of the input with these outputs, its column sums
    _Float16* X16out; int ldx16;                               // fp16 copy of the input as the first stage reads it
    ChainStage S[ORX_CHAIN_MAX];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CH_ROWS = 32;                  // samples per workgroup (two 16-row MFMA tiles)
constexpr int CH_WAVES = 8;
constexpr int CH_NT = 64 * CH_WAVES;
constexpr int CH_INFLIGHT = 16;              // weight fragments (1 KB per wavefront each) in flight per wavefront: K steps ahead = 16 / CT


// LDS row pitch (halves) for rows of `w` halves
__host__ __device__ constexpr int ch_pitch(int w) { return ((w + 31) & ~31) + 8; }

// one stage over the workgroup's 32 rows.  CT = column tiles of 16 the wavefront works on at once; BWD: the epilogue is the activation
// backward with the stage's mask (else + bias, activation).
// Every guarded load here clamps its ADDRESS and loads unconditionally: a load under a condition becomes a branch with
// `s_waitcnt vmcnt(0)` behind it -- one memory round trip per load instead of one per batch of loads (the first version of this
// kernel spent 33 us that way).  A weight chunk beyond the row's end (K not a multiple of 32) re-reads the row's first chunk: its
// partner columns of the activations are zero in LDS.
// DBG (scratch/exp_chain.hip only): 1 = no global stores, 2 = every weight request reads the row's first chunk of tile 0 (L1 hits),
// 4 = no MFMA, 8 = no input loads, 16 = weights as if packed in fragment order (one contiguous KB per request; wrong results)
template <int CT, bool BWD, int DBG>
__device__ __forceinline__ void chain_pass(const ChainStage& S, const _Float16* in, int ldin, _Float16* out, int ldout,
                                           int row0, int B, int blk, int first_tile, int tiles) {
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, q = lane >> 4;
    constexpr int CH_PD = CH_INFLIGHT / CT;
    const int nk = (S.K + 31) >> 5;
    f32x4 acc[2][CT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < CT; ++j) { acc[m][j].x = acc[m][j].y = acc[m][j].z = acc[m][j].w = 0.0f; }
    // the lane's row of W for tile j: n = 16 * (first_tile + CH_WAVES * j) + r16 (a tile beyond the last one: the last one again, never stored)
    const _Float16* wrow[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) wrow[j] = S.W + (int64_t)(16 * min(first_tile + CH_WAVES * j, tiles - 1) + r16) * S.ldw;
    const int ldw = S.ldw;
    auto wload = [&](int j, int s) -> h8 {
        const int k = s * 32 + q * 8;
        if (DBG & 2) return *reinterpret_cast<const h8*>(S.W + r16 * ldw + q * 8);
        if (DBG & 16) return *reinterpret_cast<const h8*>(S.W + ((int64_t)(min(first_tile + CH_WAVES * j, tiles - 1) * nk + s) * 64 + lane) * 8);
        return *reinterpret_cast<const h8*>(wrow[j] + (k < ldw ? k : 0));
    };
    // the mask / the bias of the epilogue: requested before the products, used after them
    h4 msk[2][CT];
    f32x4 bias[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int t = min(first_tile + CH_WAVES * j, tiles - 1);
        if (BWD) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
                msk[m][j] = *reinterpret_cast<const h4*>(S.M16 + (int64_t)min(row0 + 16 * m + r16, B - 1) * S.ldm + 16 * t + 4 * q);
        } else {
            bias[j] = *reinterpret_cast<const f32x4*>(S.bias + 16 * t + 4 * q);
        }
    }
    h8 b[CH_PD][CT];
#pragma unroll
    for (int u = 0; u < CH_PD; ++u)
#pragma unroll
        for (int j = 0; j < CT; ++j) b[u][j] = wload(j, min(u, nk - 1));
    const _Float16* a0 = in + r16 * ldin + q * 8;
    const _Float16* a1 = a0 + 16 * ldin;
    auto kstep = [&](int s, int u) {
        const h8 x0 = *reinterpret_cast<const h8*>(a0 + s * 32), x1 = *reinterpret_cast<const h8*>(a1 + s * 32);
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if (DBG & 4) { acc[0][j].x += (float)b[u][j][0] + (float)x0[0]; acc[1][j].x += (float)b[u][j][1] + (float)x1[0]; continue; }
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][j], x0, acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][j], x1, acc[1][j], 0, 0, 0);
        }
    };
    int s0 = 0;
    for (; s0 + CH_PD <= nk; s0 += CH_PD) {          // whole groups: no branch between a request and its use
#pragma unroll
        for (int u = 0; u < CH_PD; ++u) {
            kstep(s0 + u, u);
#pragma unroll
            for (int j = 0; j < CT; ++j) b[u][j] = wload(j, min(s0 + u + CH_PD, nk - 1));      // (past the end: a re-read nobody uses)
            __builtin_amdgcn_sched_barrier(0);       // (the refill stays behind its step: moved to the end of the group it would be one round trip per group)
        }
    }
#pragma unroll
    for (int u = 0; u < CH_PD; ++u)
        if (s0 + u < nk) kstep(s0 + u, u);
    // ---- epilogue: the lane holds out[row = 16 m + r16][col = 16 t + 4 q .. + 3]
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int t = first_tile + CH_WAVES * j, col = 16 * t + 4 * q;
        if (t >= tiles) continue;
        float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float v[4] = {acc[m][j].x, acc[m][j].y, acc[m][j].z, acc[m][j].w};
            if (BWD) {
                if (S.act_y == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)msk[m][j][e] > 0.0f ? v[e] : 0.0f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[e] += v[e];
            } else {
                v[0] += bias[j].x; v[1] += bias[j].y; v[2] += bias[j].z; v[3] += bias[j].w;
                if (S.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                }
            }
            const int lrow = 16 * m + r16, row = row0 + lrow;
            h4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
            *reinterpret_cast<h4*>(out + lrow * ldout + col) = o;
            if (!(DBG & 1) && S.Y32 != nullptr && row < B) {
                f32x4 w; w.x = v[0]; w.y = v[1]; w.z = v[2]; w.w = v[3];
                *reinterpret_cast<f32x4*>(S.Y32 + (int64_t)row * S.ldy32 + col) = w;
            }
        }
        if (!(DBG & 1) && BWD && S.gb != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = group_allreduce<16>(cs[e]);
                if (r16 == 0) S.gb[(int64_t)blk * S.N + col + e] = s;
            }
        }
    }
}

template <bool BWD, int DBG>
__device__ __forceinline__ void chain_stage(const ChainStage& S, const _Float16* in, _Float16* out, int ld, int row0, int B, int blk, int wave) {
    const int tiles = S.N >> 4;
    const int tpw = (tiles + CH_WAVES - 1) / CH_WAVES;           // tiles of this wavefront (at most)
    // wavefront w owns tiles w, w + 8, w + 16, ...; CT of them per pass over the rows' K
    int done = 0;
    while (done < tpw) {
        const int left = tpw - done, first = wave + CH_WAVES * done;
        if (left >= 4) { chain_pass<4, BWD, DBG>(S, in, ld, out, ld, row0, B, blk, first, tiles); done += 4; }
        else if (left >= 2) { chain_pass<2, BWD, DBG>(S, in, ld, out, ld, row0, B, blk, first, tiles); done += 2; }
        else { chain_pass<1, BWD, DBG>(S, in, ld, out, ld, row0, B, blk, first, tiles); done += 1; }
    }
}

template <int DBG>
__global__ __launch_bounds__(CH_NT, 1) void mlp_chain_kernel(ChainArgs g) {
    extern __shared__ __attribute__((aligned(16))) _Float16 chain_lds[];
    const int blk = blockIdx.x, row0 = blk * CH_ROWS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    _Float16* buf0 = chain_lds;
    _Float16* buf1 = chain_lds + CH_ROWS * g.pitch;
    const int ld = g.pitch;
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(g.xgb) + (size_t)blk * 8;     // DBG 32: s_memtime at the stage boundaries (xgb is a scratch array then)
    int n_st = 0;
    auto stamp = [&]() { if ((DBG & 32) && threadIdx.x == 0) stamps[n_st++] = __builtin_readcyclecounter(); };
    stamp();
    // ---- the input: fp32 rows -> (scale, activation backward + its column sums) -> fp16 in LDS (+ the fp16 copy in global memory)
    {
        const int K0 = g.K0, K0p = (K0 + 31) & ~31;
        for (int c = threadIdx.x; c < K0p; c += CH_NT) {
            float sum = 0.0f;
            if (c < K0) {
                float v[CH_ROWS];
#pragma unroll
                for (int r = 0; r < CH_ROWS; ++r) v[r] = (DBG & 8) ? (float)(r + c) : g.X32[(int64_t)min(row0 + r, g.B - 1) * g.ldx + c] * g.xscale;
                if (!(DBG & 8) && g.XM32 != nullptr && g.x_act == 1) {
                    float y[CH_ROWS];
#pragma unroll
                    for (int r = 0; r < CH_ROWS; ++r) y[r] = g.XM32[(int64_t)min(row0 + r, g.B - 1) * g.ldxm + c];
#pragma unroll
                    for (int r = 0; r < CH_ROWS; ++r) v[r] = y[r] > 0.0f ? v[r] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < CH_ROWS; ++r) v[r] = row0 + r < g.B ? v[r] : 0.0f;
#pragma unroll
                for (int r = 0; r < CH_ROWS; ++r) { sum += v[r]; buf0[r * ld + c] = (_Float16)v[r]; }
                if (!(DBG & 1) && g.X16out != nullptr) {
#pragma unroll
                    for (int r = 0; r < CH_ROWS; ++r) if (row0 + r < g.B) g.X16out[(int64_t)(row0 + r) * g.ldx16 + c] = (_Float16)v[r];
                }
                if (!(DBG & 33) && g.xgb != nullptr) g.xgb[(int64_t)blk * K0 + c] = sum;
            } else {
#pragma unroll
                for (int r = 0; r < CH_ROWS; ++r) buf0[r * ld + c] = (_Float16)0.0f;
                if (!(DBG & 1) && g.X16out != nullptr && c < g.ldx16)
                    for (int r = 0; r < CH_ROWS; ++r) if (row0 + r < g.B) g.X16out[(int64_t)(row0 + r) * g.ldx16 + c] = (_Float16)0.0f;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (not __syncthreads: on gfx9 stores count in vmcnt, and the barrier would wait for their acknowledgements)
    stamp();
    _Float16* in = buf0; _Float16* out = buf1;
    for (int si = 0; si < g.n_stages; ++si) {
        const ChainStage& S = g.S[si];
        if (g.backward) chain_stage<true, DBG>(S, in, out, ld, row0, g.B, blk, wave);
        else chain_stage<false, DBG>(S, in, out, ld, row0, g.B, blk, wave);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!(DBG & 1) && S.Y16 != nullptr) {                         // the fp16 copy leaves in whole 16-byte chunks of a row
            const int cpr = S.N >> 3;
            for (int c = threadIdx.x; c < CH_ROWS * cpr; c += CH_NT) {
                const int r = c / cpr, k = (c - r * cpr) * 8;
                if (row0 + r < g.B) *reinterpret_cast<h8*>(S.Y16 + (int64_t)(row0 + r) * S.ldy16 + k) = *reinterpret_cast<const h8*>(out + r * ld + k);
            }
        }
        _Float16* t = in; in = out; out = t;
        stamp();
    }
}

}  // namespace

