// The sorted sparse apply's device pieces that more than one translation unit uses (kernels_rowsort.hip: the apply and finish launches;
// kernels_dense.hip: the finish pass riding in the dense optimizer's launch, round 6).
#pragma once
#include "orx_device.h"

constexpr uint32_t CSR_KEY_NONE = 0xffffffffu;

enum { CSR_SGD = 0, CSR_ADAGRAD = 1, CSR_ADAM = 2, CSR_ACCUM = 3 };

struct CsrArgs {
    const uint2* sorted; int64_t n; uint32_t rows; int D;
    const float* grads; int64_t g_stride;
    float* W; float* A; float* V; int* last; float* G;      // table; Adagrad acc / Adam m; Adam v; lazy stamps; gsum (ACCUM)
    float lr, eps, b1, b2, lr_T; const float* lrt; int T, newton; AdamCFParams cf;
    float* part_lo; float* part_hi;                          // [blocks][Dp]: sums of the runs open at a block's start / end
    int Dp;
    int skip_single;                                         // rows referenced once are NOT applied here: the kernel that formed their
                                                             // gradient updated them in place (orx_rows_single_flags, interact_bwd_mfma_kernel)
};

// the finish pass as a passenger of another launch (dense_apply_fused_kernel): blocks = workgroups of 4 wavefronts, one wavefront per 64-entry block
struct CsrFinish { CsrArgs a; int blocks; int mode; };

// the row's state, loaded together with the gradient rows (no dependent round trip when the rule is applied)
template <int NE, int MODE>
struct RowState {
    float w[NE], a[NE], v[NE]; int last;
    __device__ __forceinline__ void load(const CsrArgs& c, uint32_t row, int lane) {
        const bool live = row < c.rows;
        last = 0;
        if (MODE == CSR_ADAM) last = live ? c.last[row] : 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int col = lane + 64 * e;
            const size_t i = (size_t)row * c.D + col;
            const bool ok = live && col < c.D;
            w[e] = a[e] = v[e] = 0.0f;
            if (MODE == CSR_ACCUM) { if (ok) w[e] = c.G[i]; continue; }
            if (ok) w[e] = c.W[i];
            if (MODE == CSR_ADAGRAD || MODE == CSR_ADAM) { if (ok) a[e] = c.A[i]; }
            if (MODE == CSR_ADAM) { if (ok) v[e] = c.V[i]; }
        }
    }
};

// the optimizer rule on row `row` with the summed gradient s[e] of column lane + 64 e
template <int NE, int MODE>
__device__ __forceinline__ void csr_rule(const CsrArgs& a, uint32_t row, const float (&s)[NE], const RowState<NE, MODE>& st, int lane) {
    int from = 0;
    if (MODE == CSR_ADAM) from = __builtin_amdgcn_readfirstlane(st.last);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int col = lane + 64 * e;
        if (col >= a.D) continue;
        const size_t i = (size_t)row * a.D + col;
        if (MODE == CSR_SGD) a.W[i] = st.w[e] - a.lr * s[e];
        else if (MODE == CSR_ADAGRAD) {
            const float acc = st.a[e] + s[e] * s[e];
            a.A[i] = acc;
            a.W[i] = st.w[e] - a.lr * s[e] / (sqrtf(acc) + a.eps);
        }
        else if (MODE == CSR_ACCUM) a.G[i] = st.w[e] + s[e];
        else {
            float w = st.w[e], m = st.a[e], v = st.v[e];
            adam_replay1<true>(w, m, v, from, a.T - 1, a.lrt, a.b1, a.b2, a.eps, a.newton != 0, a.cf);
            adam_elem(w, m, v, s[e], a.lr_T, a.b1, a.b2, a.eps);
            a.W[i] = w; a.A[i] = m; a.V[i] = v;
        }
    }
    if (MODE == CSR_ADAM && lane == 0) a.last[row] = a.T;
}

// runs that cross block boundaries: the wavefront of the block a run STARTS in adds the partial sums in block order
template <int NE, int MODE>
__device__ __forceinline__ void csr_finish_block(const CsrArgs& a, int64_t b, int lane) {
    const int64_t i0 = b * 64;
    if (i0 + 64 >= a.n) return;                              // the last block's runs end in it
    const uint32_t row = a.sorted[i0 + 63].x;
    if (row >= a.rows || a.sorted[i0 + 64].x != row) return;           // the block's last run ends here
    if (a.sorted[i0].x == row && b > 0 && a.sorted[i0 - 1].x == row) return;   // ... or started in an earlier block
    // how many blocks the run continues into: lane l looks at the first key of block b + 2 + l (64 blocks per round)
    int64_t last = b + 1;                                    // the last block that holds a piece of the run
    for (;;) {
        const int64_t nx = (last + 1 + lane) * 64;
        const bool cont = nx < a.n && a.sorted[nx].x == row;
        const unsigned long long m = __ballot(cont);
        const int run = m == ~0ull ? 64 : __builtin_ctzll(~m);         // consecutive continuing blocks
        last += run;
        if (run < 64) break;
    }
    float s[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) s[e] = lane + 64 * e < a.D ? a.part_hi[(size_t)b * a.Dp + lane + 64 * e] : 0.0f;
    constexpr int UN = 8;
    for (int64_t c0 = b + 1; c0 <= last; c0 += UN) {         // UN partial rows in flight, added in block order
        float q[UN][NE];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int e = 0; e < NE; ++e) q[u][e] = (c0 + u <= last && lane + 64 * e < a.D) ? a.part_lo[(size_t)(c0 + u) * a.Dp + lane + 64 * e] : 0.0f;
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int e = 0; e < NE; ++e) if (c0 + u <= last) s[e] += q[u][e];
    }
    RowState<NE, MODE> st;
    st.load(a, row, lane);
    csr_rule<NE, MODE>(a, row, s, st, lane);
}

