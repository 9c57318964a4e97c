// Fused pairwise (BPR / UCML) train-step kernels for gfx950 (MI355X, CDNA4).
//
// Replaces, for one batch of B (user, pos item, neg item) triplets, the TF op
// sequence that the reference makes TensorFlow run (paths relative to
// /root/reference):
//   5 Embedding gathers                      recommenders/bpr.py:23-27, ucml.py:23-27
//   dot / L2-distance score, bias add        modules/pairwise_log_loss.py:19-30, ucml.py:29-37
//   clamp, log-sigmoid mean | hinge sum      pairwise_log_loss.py:32 | ucml.py:39
//   l2_loss                                  bpr.py:35, ucml.py:40
//   GradientTape over (loss, l2_loss)        tf2_examples/bpr_citeulike.py:35-37
//   Keras optimizer sparse apply             tf2_examples/bpr_citeulike.py:38
//
// Design:
//   * HBM-bound: 3 random row reads + 3 random row writes per triplet, no reuse.
//     A row of D fp32 is owned by LPR = D/4 adjacent lanes, one float4 each, so a
//     D=64 row is one coalesced 256-B segment and a 64-lane wavefront carries
//     64/LPR triplets; dot products reduce inside the lane group with DPP
//     (quad_perm / row_half_mirror / row_mirror), never through LDS.
//   * TF semantics need every gradient of a step to be taken on the PRE-step
//     tables.  A one-pass in-place kernel violates that only for rows referenced
//     more than once in the batch, so the step is three launches:
//       count_kernel  : cnt[row] += 1 per reference            (4-B atomics)
//       fused_kernel  : rows with cnt == 1 are updated in place (exact: nobody
//                       else reads or writes them); references to rows with
//                       cnt != 1 add their gradient to gsum[row] with fp32
//                       atomics and leave the table row untouched
//       dup_kernel    : per duplicate reference cnt[row] -= 1; the reference
//                       that brings it to 0 applies the optimizer rule with the
//                       summed gradient (TF dedup-sum semantics for Adagrad,
//                       identical result for SGD) and re-zeroes gsum[row]
//     cnt[] and gsum[] are all-zero again after every step.
#include "orx_internal.h"

typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------ lane helpers ---
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

// sum over the LPR adjacent lanes that own one row; every lane gets the total
template <int LPR>
__device__ __forceinline__ float group_allreduce(float x) {
    if (LPR >= 2) x += dpp_f<0xB1>(x);    // quad_perm [1,0,3,2]
    if (LPR >= 4) x += dpp_f<0x4E>(x);    // quad_perm [2,3,0,1]
    if (LPR >= 8) x += dpp_f<0x141>(x);   // row_half_mirror
    if (LPR >= 16) x += dpp_f<0x140>(x);  // row_mirror
    if (LPR >= 32) x += __shfl_xor(x, 16);
    if (LPR >= 64) x += __shfl_xor(x, 32);
    return x;
}

__device__ __forceinline__ float wave_sum(float x) {
    x = group_allreduce<16>(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

__device__ __forceinline__ float dot4(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__device__ __forceinline__ void atomic_add_f4(float* p, f4 v) {
    unsafeAtomicAdd(p + 0, v.x);
    unsafeAtomicAdd(p + 1, v.y);
    unsafeAtomicAdd(p + 2, v.z);
    unsafeAtomicAdd(p + 3, v.w);
}

// ------------------------------------------------------------ score / loss ---
// Returns the per-triplet loss term and the gradient coefficient `g`.
//   BPR : x = s+ - s-,  term = -log_sigmoid(max(x,-30))/B,  g = dJ/dx
//   UCML: h = margin - diff, term = max(h,0), g = [h >= 0]
template <int MODEL>
__device__ __forceinline__ void score(float red, float bp, float bn, float invB, float margin,
                                      float& term, float& g) {
    if (MODEL == ORX_BPR) {
        const float x = red + bp - bn;                       // pairwise_log_loss.py:19-30
        const float m = fmaxf(x, -30.0f);                    // :32
        const float e = __expf(-fabsf(m));
        term = (fmaxf(-m, 0.0f) + log1pf(e)) * invB;         // -log_sigmoid(m) / B
        const float sig = (x >= 0.0f) ? e / (1.0f + e) : 1.0f / (1.0f + e);   // sigmoid(-x)
        g = (x >= -30.0f) ? -sig * invB : 0.0f;              // Maximum: gradient to arg 0 on >=
    } else {
        const float diff = red + bp - bn;                    // ucml.py:35-37, red = d(u,n) - d(u,p)
        const float h = margin - diff;
        term = fmaxf(h, 0.0f);                               // ucml.py:39 (sum)
        g = (h >= 0.0f) ? 1.0f : 0.0f;
    }
}

template <int MODEL>
__device__ __forceinline__ float score_partial(f4 u, f4 p, f4 n) {
    if (MODEL == ORX_BPR) {
        return dot4(u, p - n);
    } else {
        const f4 a = u - p, c = u - n;
        return dot4(c, c) - dot4(a, a);
    }
}

// per-occurrence gradients of J = loss + l2w * l2_loss w.r.t. the gathered rows
template <int MODEL>
__device__ __forceinline__ void row_grads(f4 u, f4 p, f4 n, float g, float l2w, f4& gu, f4& gp, f4& gn,
                                          float& gbp, float& gbn) {
    if (MODEL == ORX_BPR) {
        gu = g * (p - n) + l2w * u;
        gp = g * u + l2w * p;
        gn = -g * u + l2w * n;
        gbp = g; gbn = -g;
    } else {
        const float a2 = 2.0f * g;
        gu = -a2 * (p - n) + l2w * u;
        gp = -a2 * (u - p) + l2w * p;
        gn = a2 * (u - n) + l2w * n;
        gbp = -g; gbn = g;
    }
}

// ---------------------------------------------------------- optimizer rule ---
template <int OPT>
__device__ __forceinline__ void opt_apply4(float* w_ptr, float* a_ptr, f4 w_old, f4 grad, float lr, float eps) {
    if (OPT == ORX_ADAGRAD) {
        f4 acc = *reinterpret_cast<f4*>(a_ptr);
        acc = acc + grad * grad;
        *reinterpret_cast<f4*>(a_ptr) = acc;
        f4 den;
        den.x = sqrtf(acc.x) + eps; den.y = sqrtf(acc.y) + eps; den.z = sqrtf(acc.z) + eps; den.w = sqrtf(acc.w) + eps;
        *reinterpret_cast<f4*>(w_ptr) = w_old - lr * grad / den;
    } else {
        *reinterpret_cast<f4*>(w_ptr) = w_old - lr * grad;
    }
}

template <int OPT>
__device__ __forceinline__ void opt_apply1(float* w_ptr, float* a_ptr, float w_old, float grad, float lr, float eps) {
    if (OPT == ORX_ADAGRAD) {
        const float acc = *a_ptr + grad * grad;
        *a_ptr = acc;
        *w_ptr = w_old - lr * grad / (sqrtf(acc) + eps);
    } else {
        *w_ptr = w_old - lr * grad;
    }
}

// ------------------------------------------------------------ count kernel ---
__global__ __launch_bounds__(256) void count_kernel(PairArgs a) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.B) return;
    const int u = a.uid[t], p = a.pid[t], n = a.nid[t];
    if ((uint32_t)u >= (uint64_t)a.NU || (uint32_t)p >= (uint64_t)a.NI || (uint32_t)n >= (uint64_t)a.NI) {
        *a.err = 1;
        return;
    }
    atomicAdd(a.cntU + u, 1);
    atomicAdd(a.cntV + p, 1);
    atomicAdd(a.cntV + n, 1);
}

// -------------------------------------------------- loss partial reduction ---
__device__ __forceinline__ void reduce_partials(const PairArgs& a) {
    // one block: sum nwaves x {loss, l2} fp32 partials in fp64 -> loss_out[0..1]
    __shared__ double sh[2][4];
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < a.nwaves; i += blockDim.x) {
        s0 += (double)a.partial[2 * i];
        s1 += (double)a.partial[2 * i + 1];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        s1 += __shfl_xor(s1, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s0; sh[1][w] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.loss_out[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        a.loss_out[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(PairArgs a) { reduce_partials(a); }

// ------------------------------------------------------------ fused kernel ---
// LPR lanes own one row (D = 4*LPR).  MODE: see orx_internal.h.
template <int LPR, int MODEL, int OPT, int MODE>
__global__ __launch_bounds__(256) void fused_kernel(PairArgs a) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4 * TPW;
    float loss_acc = 0.0f, sq_acc = 0.0f;

    for (int64_t t = wave_global * TPW + grp; t < a.B; t += stride) {
        const int u = a.uid[t], p = a.pid[t], n = a.nid[t];
        if ((uint32_t)u >= (uint64_t)a.NU || (uint32_t)p >= (uint64_t)a.NI || (uint32_t)n >= (uint64_t)a.NI) {
            if (sub == 0) { *a.err = 1; if (MODE == MODE_EXACT) a.dupmask[t] = 0; }
            continue;
        }
        int cu = 1, cp = 1, cn = 1;
        if (MODE == MODE_EXACT) { cu = a.cntU[u]; cp = a.cntV[p]; cn = a.cntV[n]; }
        float* Up = a.U + (size_t)u * D + 4 * sub;
        float* Pp = a.V + (size_t)p * D + 4 * sub;
        float* Np = a.V + (size_t)n * D + 4 * sub;
        const f4 ru = *reinterpret_cast<const f4*>(Up);
        const f4 rp = *reinterpret_cast<const f4*>(Pp);
        const f4 rn = *reinterpret_cast<const f4*>(Np);
        const float bp = a.b[p], bn = a.b[n];

        const float red = group_allreduce<LPR>(score_partial<MODEL>(ru, rp, rn));
        float term, g;
        score<MODEL>(red, bp, bn, a.invB, a.margin, term, g);
        sq_acc += dot4(ru, ru) + dot4(rp, rp) + dot4(rn, rn);
        if (sub == 0) loss_acc += term;
        if (MODE == MODE_LOSS) continue;

        f4 gu, gp, gn; float gbp, gbn;
        row_grads<MODEL>(ru, rp, rn, g, a.l2w, gu, gp, gn, gbp, gbn);

        if (MODE == MODE_ACCUM) {
            atomic_add_f4(a.gU + (size_t)u * D + 4 * sub, gu);
            atomic_add_f4(a.gV + (size_t)p * D + 4 * sub, gp);
            atomic_add_f4(a.gV + (size_t)n * D + 4 * sub, gn);
            if (sub == 0) { unsafeAtomicAdd(a.gb + p, gbp); unsafeAtomicAdd(a.gb + n, gbn); }
            continue;
        }
        // user row
        if (cu == 1) {
            opt_apply4<OPT>(Up, a.aU + (size_t)u * D + 4 * sub, ru, gu, a.lr, a.eps);
            if (MODE == MODE_EXACT && sub == 0) a.cntU[u] = 0;
        } else {
            atomic_add_f4(a.gU + (size_t)u * D + 4 * sub, gu);
        }
        // positive item row + bias
        if (cp == 1) {
            opt_apply4<OPT>(Pp, a.aV + (size_t)p * D + 4 * sub, rp, gp, a.lr, a.eps);
            if (sub == 0) {
                opt_apply1<OPT>(a.b + p, a.ab + p, bp, gbp, a.lr, a.eps);
                if (MODE == MODE_EXACT) a.cntV[p] = 0;
            }
        } else {
            atomic_add_f4(a.gV + (size_t)p * D + 4 * sub, gp);
            if (sub == 0) unsafeAtomicAdd(a.gb + p, gbp);
        }
        // negative item row + bias
        if (cn == 1) {
            opt_apply4<OPT>(Np, a.aV + (size_t)n * D + 4 * sub, rn, gn, a.lr, a.eps);
            if (sub == 0) {
                opt_apply1<OPT>(a.b + n, a.ab + n, bn, gbn, a.lr, a.eps);
                if (MODE == MODE_EXACT) a.cntV[n] = 0;
            }
        } else {
            atomic_add_f4(a.gV + (size_t)n * D + 4 * sub, gn);
            if (sub == 0) unsafeAtomicAdd(a.gb + n, gbn);
        }
        if (MODE == MODE_EXACT && sub == 0)
            a.dupmask[t] = (unsigned char)((cu != 1) | ((cp != 1) << 1) | ((cn != 1) << 2));
    }
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        a.partial[2 * wave_global] = ls;
        a.partial[2 * wave_global + 1] = 0.5f * sq;
    }
}

// Any D: one triplet per wavefront, scalar elements strided by 64 lanes, two
// passes over the (L1/L2-resident) rows.  Used for dims without a float4 path
// (e.g. the example's dim_embed = 50, tf2_examples/bpr_citeulike.py:12).
template <int MODEL, int OPT, int MODE>
__global__ __launch_bounds__(256) void fused_generic_kernel(PairArgs a) {
    const int lane = threadIdx.x & 63;
    const int D = a.D;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    float loss_acc = 0.0f, sq_acc = 0.0f;
    for (int64_t t = wave_global; t < a.B; t += stride) {
        const int u = a.uid[t], p = a.pid[t], n = a.nid[t];
        if ((uint32_t)u >= (uint64_t)a.NU || (uint32_t)p >= (uint64_t)a.NI || (uint32_t)n >= (uint64_t)a.NI) {
            if (lane == 0) { *a.err = 1; if (MODE == MODE_EXACT) a.dupmask[t] = 0; }
            continue;
        }
        int cu = 1, cp = 1, cn = 1;
        if (MODE == MODE_EXACT) { cu = a.cntU[u]; cp = a.cntV[p]; cn = a.cntV[n]; }
        float* Ur = a.U + (size_t)u * D;
        float* Pr = a.V + (size_t)p * D;
        float* Nr = a.V + (size_t)n * D;
        const float bp = a.b[p], bn = a.b[n];
        float part = 0.0f;
        for (int e = lane; e < D; e += 64) {
            const float x = Ur[e], y = Pr[e], z = Nr[e];
            if (MODEL == ORX_BPR) part += x * (y - z);
            else part += (x - z) * (x - z) - (x - y) * (x - y);
            sq_acc += x * x + y * y + z * z;
        }
        const float red = wave_sum(part);
        float term, g;
        score<MODEL>(red, bp, bn, a.invB, a.margin, term, g);
        if (lane == 0) loss_acc += term;
        if (MODE == MODE_LOSS) continue;
        float gbp, gbn;
        for (int e = lane; e < D; e += 64) {
            const float x = Ur[e], y = Pr[e], z = Nr[e];
            float gu, gp, gn;
            if (MODEL == ORX_BPR) {
                gu = g * (y - z) + a.l2w * x; gp = g * x + a.l2w * y; gn = -g * x + a.l2w * z;
            } else {
                const float a2 = 2.0f * g;
                gu = -a2 * (y - z) + a.l2w * x; gp = -a2 * (x - y) + a.l2w * y; gn = a2 * (x - z) + a.l2w * z;
            }
            if (MODE != MODE_ACCUM && cu == 1) opt_apply1<OPT>(Ur + e, a.aU + (size_t)u * D + e, x, gu, a.lr, a.eps);
            else unsafeAtomicAdd(a.gU + (size_t)u * D + e, gu);
            if (MODE != MODE_ACCUM && cp == 1) opt_apply1<OPT>(Pr + e, a.aV + (size_t)p * D + e, y, gp, a.lr, a.eps);
            else unsafeAtomicAdd(a.gV + (size_t)p * D + e, gp);
            if (MODE != MODE_ACCUM && cn == 1) opt_apply1<OPT>(Nr + e, a.aV + (size_t)n * D + e, z, gn, a.lr, a.eps);
            else unsafeAtomicAdd(a.gV + (size_t)n * D + e, gn);
        }
        if (MODEL == ORX_BPR) { gbp = g; gbn = -g; } else { gbp = -g; gbn = g; }
        if (lane == 0) {
            if (MODE != MODE_ACCUM && cp == 1) opt_apply1<OPT>(a.b + p, a.ab + p, bp, gbp, a.lr, a.eps);
            else unsafeAtomicAdd(a.gb + p, gbp);
            if (MODE != MODE_ACCUM && cn == 1) opt_apply1<OPT>(a.b + n, a.ab + n, bn, gbn, a.lr, a.eps);
            else unsafeAtomicAdd(a.gb + n, gbn);
            if (MODE == MODE_EXACT) {
                if (cu == 1) a.cntU[u] = 0;
                if (cp == 1) a.cntV[p] = 0;
                if (cn == 1) a.cntV[n] = 0;
                a.dupmask[t] = (unsigned char)((cu != 1) | ((cp != 1) << 1) | ((cn != 1) << 2));
            }
        }
    }
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        a.partial[2 * wave_global] = ls;
        a.partial[2 * wave_global + 1] = 0.5f * sq;
    }
}

// -------------------------------------------------------------- dup kernel ---
// Finalize one duplicate reference: decrement the row's reference count; the
// reference that reaches zero applies the optimizer with the summed gradient.
template <int LPR, int OPT>
__device__ __forceinline__ void finalize_row(float* W, float* G, int* cnt, float* A,
                                             float* bW, float* bG, float* bA,
                                             int row, int sub, int leader_lane, float lr, float eps) {
    constexpr int D = 4 * LPR;
    int old = 0;
    if (sub == 0) old = atomicSub(cnt + row, 1);
    old = __shfl(old, leader_lane);
    if (old != 1) return;
    float* gp = G + (size_t)row * D + 4 * sub;
    float* wp = W + (size_t)row * D + 4 * sub;
    const f4 g = *reinterpret_cast<const f4*>(gp);
    const f4 w = *reinterpret_cast<const f4*>(wp);
    f4 z; z.x = z.y = z.z = z.w = 0.0f;
    *reinterpret_cast<f4*>(gp) = z;
    opt_apply4<OPT>(wp, A + (size_t)row * D + 4 * sub, w, g, lr, eps);
    if (bW != nullptr && sub == 0) {
        const float gb = bG[row];
        bG[row] = 0.0f;
        opt_apply1<OPT>(bW + row, bA + row, bW[row], gb, lr, eps);
    }
}

template <int LPR, int OPT>
__global__ __launch_bounds__(256) void dup_kernel(PairArgs a) {
    constexpr int TPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4 * TPW;
    for (int64_t t = wave_global * TPW + grp; t < a.B; t += stride) {
        const int m = a.dupmask[t];
        if (m == 0) continue;
        const int lead = grp * LPR;
        if (m & 1) finalize_row<LPR, OPT>(a.U, a.gU, a.cntU, a.aU, nullptr, nullptr, nullptr, a.uid[t], sub, lead, a.lr, a.eps);
        if (m & 2) finalize_row<LPR, OPT>(a.V, a.gV, a.cntV, a.aV, a.b, a.gb, a.ab, a.pid[t], sub, lead, a.lr, a.eps);
        if (m & 4) finalize_row<LPR, OPT>(a.V, a.gV, a.cntV, a.aV, a.b, a.gb, a.ab, a.nid[t], sub, lead, a.lr, a.eps);
    }
    if (blockIdx.x == 0) reduce_partials(a);
}

template <int OPT>
__device__ __forceinline__ void finalize_row_generic(float* W, float* G, int* cnt, float* A,
                                                     float* bW, float* bG, float* bA,
                                                     int row, int lane, int D, float lr, float eps) {
    int old = 0;
    if (lane == 0) old = atomicSub(cnt + row, 1);
    old = __shfl(old, 0);
    if (old != 1) return;
    for (int e = lane; e < D; e += 64) {
        const size_t i = (size_t)row * D + e;
        const float g = G[i];
        G[i] = 0.0f;
        opt_apply1<OPT>(W + i, A + i, W[i], g, lr, eps);
    }
    if (bW != nullptr && lane == 0) {
        const float gb = bG[row];
        bG[row] = 0.0f;
        opt_apply1<OPT>(bW + row, bA + row, bW[row], gb, lr, eps);
    }
}

template <int OPT>
__global__ __launch_bounds__(256) void dup_generic_kernel(PairArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t t = wave_global; t < a.B; t += stride) {
        const int m = a.dupmask[t];
        if (m == 0) continue;
        if (m & 1) finalize_row_generic<OPT>(a.U, a.gU, a.cntU, a.aU, nullptr, nullptr, nullptr, a.uid[t], lane, a.D, a.lr, a.eps);
        if (m & 2) finalize_row_generic<OPT>(a.V, a.gV, a.cntV, a.aV, a.b, a.gb, a.ab, a.pid[t], lane, a.D, a.lr, a.eps);
        if (m & 4) finalize_row_generic<OPT>(a.V, a.gV, a.cntV, a.aV, a.b, a.gb, a.ab, a.nid[t], lane, a.D, a.lr, a.eps);
    }
    if (blockIdx.x == 0) reduce_partials(a);
}

// ---------------------------------------------------------------- launchers ---
static inline int lpr_for_dim(int D) {
    switch (D) {
        case 16: return 4;
        case 32: return 8;
        case 64: return 16;
        case 128: return 32;
        case 256: return 64;
        default: return 0;      // generic path
    }
}

static inline int64_t fused_grid(int D, int64_t B) {
    const int lpr = lpr_for_dim(D);
    const int64_t tpb = lpr ? 4 * (64 / lpr) : 4;      // triplets per 256-thread block per pass
    int64_t g = (B + tpb - 1) / tpb;
    const int64_t cap = 1 << 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return g;
}

int orx_fused_nwaves(int D, int64_t B) { return (int)(fused_grid(D, B) * 4); }

int orx_launch_count(orx_ctx* ctx, const PairArgs& a) {
    ProfScope ps(ctx, ORX_K_COUNT);
    const int64_t g = (a.B + 255) / 256;
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

template <int LPR, int MODEL, int OPT>
static void launch_fused_mode(int mode, dim3 g, hipStream_t s, const PairArgs& a) {
    switch (mode) {
        case MODE_EXACT: hipLaunchKernelGGL((fused_kernel<LPR, MODEL, OPT, MODE_EXACT>), g, dim3(256), 0, s, a); break;
        case MODE_HOGWILD: hipLaunchKernelGGL((fused_kernel<LPR, MODEL, OPT, MODE_HOGWILD>), g, dim3(256), 0, s, a); break;
        case MODE_ACCUM: hipLaunchKernelGGL((fused_kernel<LPR, MODEL, ORX_SGD, MODE_ACCUM>), g, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((fused_kernel<LPR, MODEL, ORX_SGD, MODE_LOSS>), g, dim3(256), 0, s, a); break;
    }
}

template <int MODEL, int OPT>
static void launch_generic_mode(int mode, dim3 g, hipStream_t s, const PairArgs& a) {
    switch (mode) {
        case MODE_EXACT: hipLaunchKernelGGL((fused_generic_kernel<MODEL, OPT, MODE_EXACT>), g, dim3(256), 0, s, a); break;
        case MODE_HOGWILD: hipLaunchKernelGGL((fused_generic_kernel<MODEL, OPT, MODE_HOGWILD>), g, dim3(256), 0, s, a); break;
        case MODE_ACCUM: hipLaunchKernelGGL((fused_generic_kernel<MODEL, ORX_SGD, MODE_ACCUM>), g, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((fused_generic_kernel<MODEL, ORX_SGD, MODE_LOSS>), g, dim3(256), 0, s, a); break;
    }
}

template <int MODEL, int OPT>
static void launch_fused_lpr(int lpr, int mode, dim3 g, hipStream_t s, const PairArgs& a) {
    switch (lpr) {
        case 4: launch_fused_mode<4, MODEL, OPT>(mode, g, s, a); break;
        case 8: launch_fused_mode<8, MODEL, OPT>(mode, g, s, a); break;
        case 16: launch_fused_mode<16, MODEL, OPT>(mode, g, s, a); break;
        case 32: launch_fused_mode<32, MODEL, OPT>(mode, g, s, a); break;
        case 64: launch_fused_mode<64, MODEL, OPT>(mode, g, s, a); break;
        default: launch_generic_mode<MODEL, OPT>(mode, g, s, a); break;
    }
}

int orx_launch_fused(orx_ctx* ctx, int model, int optkind, int mode, const PairArgs& a, int* nwaves_out) {
    ProfScope ps(ctx, ORX_K_FUSED);
    const int lpr = lpr_for_dim(a.D);
    const dim3 g((unsigned)fused_grid(a.D, a.B));
    if (nwaves_out) *nwaves_out = (int)g.x * 4;
    const int ok = (optkind == ORX_ADAGRAD) ? ORX_ADAGRAD : ORX_SGD;
    if (model == ORX_BPR) {
        if (ok == ORX_ADAGRAD) launch_fused_lpr<ORX_BPR, ORX_ADAGRAD>(lpr, mode, g, ctx->stream, a);
        else launch_fused_lpr<ORX_BPR, ORX_SGD>(lpr, mode, g, ctx->stream, a);
    } else {
        if (ok == ORX_ADAGRAD) launch_fused_lpr<ORX_UCML, ORX_ADAGRAD>(lpr, mode, g, ctx->stream, a);
        else launch_fused_lpr<ORX_UCML, ORX_SGD>(lpr, mode, g, ctx->stream, a);
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

template <int OPT>
static void launch_dup_lpr(int lpr, dim3 g, hipStream_t s, const PairArgs& a) {
    switch (lpr) {
        case 4: hipLaunchKernelGGL((dup_kernel<4, OPT>), g, dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((dup_kernel<8, OPT>), g, dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((dup_kernel<16, OPT>), g, dim3(256), 0, s, a); break;
        case 32: hipLaunchKernelGGL((dup_kernel<32, OPT>), g, dim3(256), 0, s, a); break;
        case 64: hipLaunchKernelGGL((dup_kernel<64, OPT>), g, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((dup_generic_kernel<OPT>), g, dim3(256), 0, s, a); break;
    }
}

// optkind < 0: only reduce the loss partials (modes without duplicate handling)
int orx_launch_dup(orx_ctx* ctx, int optkind, const PairArgs& a) {
    ProfScope ps(ctx, ORX_K_DUP);
    if (optkind < 0) {
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, a);
    } else {
        const int lpr = lpr_for_dim(a.D);
        const dim3 g((unsigned)fused_grid(a.D, a.B));
        if (optkind == ORX_ADAGRAD) launch_dup_lpr<ORX_ADAGRAD>(lpr, g, ctx->stream, a);
        else launch_dup_lpr<ORX_SGD>(lpr, g, ctx->stream, a);
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
