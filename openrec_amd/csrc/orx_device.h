// Device-side helpers shared by the kernel translation units (gfx950).
#pragma once
#include "orx_internal.h"

typedef float f4 __attribute__((ext_vector_type(4)));

// 16-byte write-through store (sc1: past the XCD's L2 as it is issued).  For outputs the kernel never reads again: nothing of them is
// left dirty for the write-back at the end of the kernel (kernels_gemm16.hip has the measurement)
template <typename V> __device__ __forceinline__ void store_sc1(V* p, V v) {
    static_assert(sizeof(V) == 16, "16-byte stores");
    // (s_nop 1: the two wait states a store of more than 64 bits needs before a VALU instruction may overwrite its data registers --
    // the compiler's hazard recognizer does not see through inline asm, and `v_accvgpr_read v0, ..` right behind the store corrupted it)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

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

// XCD-aware tile order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so id b -> slot
// (b % 8) * (n / 8) + b / 8 hands every XCD a contiguous run of the row-major tile list
__device__ __forceinline__ int xcd_slot(int b, int n) { return (n & 7) ? b : (b & 7) * (n >> 3) + (b >> 3); }

__device__ __forceinline__ float dot4(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__device__ __forceinline__ void atomic_add_f4(float* p, f4 v) {
    unsafeAtomicAdd(p + 0, v.x);
    unsafeAtomicAdd(p + 1, v.y);
    unsafeAtomicAdd(p + 2, v.z);
    unsafeAtomicAdd(p + 3, v.w);
}

// device-coherent load: bypasses the per-CU L1 and the non-coherent per-XCD L2,
// i.e. observes fp32 atomics performed by any CU of the device.
__device__ __forceinline__ float load_coherent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool id_ok(int id, int64_t rows) { return (uint32_t)id < (uint64_t)rows; }

// ------------------------------------------------------------ score / loss ---
// Returns the per-triplet loss term and the gradient coefficient `g`.
//   BPR : x = s+ - s-,  term = -log_sigmoid(max(x,-30))/B,  g = dJ/dx
//   UCML: h = margin - diff, term = max(h,0), g = [h >= 0]
// log1p(e) for e in [0, 1] (e = exp(-|x|) of a log-sigmoid): hardware log2 of u = 1 + e, corrected by e / (u - 1) for the
// rounding of the sum (Kahan's log1p) -- within a few ulp (3e-7 relative) over the whole range, against the ~150 instructions
// and five hoisted constant registers of libdevice's log1pf
__device__ __forceinline__ float log1p_unit(float e) {
    const float u = 1.0f + e;
    const float d = u - 1.0f;
    return d == 0.0f ? e : __logf(u) * (e * __builtin_amdgcn_rcpf(d));
}

template <int MODEL>
__device__ __forceinline__ void score(float red, float bp, float bn, float invB, float margin,
                                      float& term, float& g) {
    if (MODEL == ORX_BPR) {
        const float x = red + bp - bn;                       // pairwise_log_loss.py:19-30
        const float m = fmaxf(x, -30.0f);                    // :32
        const float e = __expf(-fabsf(m));
        term = (fmaxf(-m, 0.0f) + log1p_unit(e)) * invB;     // -log_sigmoid(m) / B
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
// gradient of one row of a triplet: c * A + l2w * row, as ONE expression (explicit fma) wherever it is formed -- the pairing
// exchange of the fused kernel forms the gradient of the shared row from selected operands and must round like row_grads
__device__ __forceinline__ f4 grad4(float c, f4 A, float l2w, f4 row) {
    f4 r;
    r.x = fmaf(c, A.x, l2w * row.x); r.y = fmaf(c, A.y, l2w * row.y); r.z = fmaf(c, A.z, l2w * row.z); r.w = fmaf(c, A.w, l2w * row.w);
    return r;
}

template <int MODEL>
__device__ __forceinline__ void row_grads(f4 u, f4 p, f4 n, float g, float l2w, f4& gu, f4& gp, f4& gn,
                                          float& gbp, float& gbn) {
    if (MODEL == ORX_BPR) {
        gu = grad4(g, p - n, l2w, u);
        gp = grad4(g, u, l2w, p);
        gn = grad4(-g, u, l2w, n);
        gbp = g; gbn = -g;
    } else {
        const float a2 = 2.0f * g;
        gu = grad4(-a2, p - n, l2w, u);
        gp = grad4(-a2, u - p, l2w, p);
        gn = grad4(a2, u - n, l2w, n);
        gbp = -g; gbn = g;
    }
}

// the same gradient for ONE slot of the triplet (0 user, 1 pos item, 2 neg item), operands selected at run time: row = the slot's row
template <int MODEL>
__device__ __forceinline__ f4 slot_grad(int slot, f4 u, f4 p, f4 n, f4 row, float g, float l2w) {
    const f4 pn = p - n;
    if (MODEL == ORX_BPR) {
        const f4 A = slot == 0 ? pn : u;
        return grad4(slot == 2 ? -g : g, A, l2w, row);
    }
    const float a2 = 2.0f * g;
    const f4 A = slot == 0 ? pn : u - row;
    return grad4(slot == 2 ? a2 : -a2, A, l2w, row);
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

// the updated row is returned instead of stored (the caller post-processes it); accumulator stored
template <int OPT>
__device__ __forceinline__ f4 opt_new4(float* a_ptr, f4 w_old, f4 grad, float lr, float eps) {
    if (OPT == ORX_ADAGRAD) {
        f4 acc = *reinterpret_cast<f4*>(a_ptr);
        acc = acc + grad * grad;
        *reinterpret_cast<f4*>(a_ptr) = acc;
        f4 den;
        den.x = sqrtf(acc.x) + eps; den.y = sqrtf(acc.y) + eps; den.z = sqrtf(acc.z) + eps; den.w = sqrtf(acc.w) + eps;
        return w_old - lr * grad / den;
    }
    return w_old - lr * grad;
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

// ------------------------------------------------------------------ Adam ---
// One element of keras.optimizers.Adam in TF 2.0 (dense and sparse apply share it): the SAME expression is used by
// the whole-table sweep, the lazy catch-up (g = 0) and the fused step, so a replayed step rounds like a swept one.
__device__ __forceinline__ void adam_elem(float& w, float& m, float& v, float g, float lr_t, float b1, float b2, float eps) {
    const float mi = b1 * m + (1.0f - b1) * g;
    const float vi = b2 * v + (1.0f - b2) * g * g;
    m = mi;
    v = vi;
    w = w - lr_t * mi / (sqrtf(vi) + eps);
}

__device__ __forceinline__ void adam_elem4(f4& w, f4& m, f4& v, f4 g, float lr_t, float b1, float b2, float eps) {
    float wx = w.x, wy = w.y, wz = w.z, ww = w.w, mx = m.x, my = m.y, mz = m.z, mw = m.w, vx = v.x, vy = v.y, vz = v.z, vw = v.w;
    adam_elem(wx, mx, vx, g.x, lr_t, b1, b2, eps); adam_elem(wy, my, vy, g.y, lr_t, b1, b2, eps);
    adam_elem(wz, mz, vz, g.z, lr_t, b1, b2, eps); adam_elem(ww, mw, vw, g.w, lr_t, b1, b2, eps);
    w.x = wx; w.y = wy; w.z = wz; w.w = ww; m.x = mx; m.y = my; m.z = mz; m.w = mw; v.x = vx; v.y = vy; v.z = vz; v.w = vw;
}

// Replay of the gradient-free steps from+1 .. to of a row slice.  With g = 0 the rule is m <- b1*m, v <- b2*v,
// w <- w - lr_k * m / (sqrt(v) + eps): m and v take the same multiplies as the whole-table sweep (bit-identical
// slots); sqrt(v) follows as s <- sqrt(b2)*s from one square root, and the quotient uses the hardware reciprocal
// (1 ulp).  The replayed var differs from the sweep's by rounding of the update term only (~1e-7 of an update).
__device__ __forceinline__ f4 sqrt4(f4 v) { f4 r; r.x = sqrtf(v.x); r.y = sqrtf(v.y); r.z = sqrtf(v.z); r.w = sqrtf(v.w); return r; }
__device__ __forceinline__ f4 rcp4(f4 v) {
    f4 r; r.x = __builtin_amdgcn_rcpf(v.x); r.y = __builtin_amdgcn_rcpf(v.y); r.z = __builtin_amdgcn_rcpf(v.z); r.w = __builtin_amdgcn_rcpf(v.w);
    return r;
}

__device__ __forceinline__ void adam_catchup4(f4& w, f4& m, f4& v, int from, int to, const float* lrt, float b1, float b2, float eps) {
    if (from >= to) return;
    f4 s = sqrt4(v);
    const float sb2 = sqrtf(b2);
    for (int k = from + 1; k <= to; ++k) {
        m = m * b1; v = v * b2; s = s * sb2;
        w = w - (lrt[k] * m) * rcp4(s + eps);
    }
}

__device__ __forceinline__ void adam_catchup1(float& w, float& m, float& v, int from, int to, const float* lrt, float b1, float b2, float eps) {
    if (from >= to) return;
    float s = sqrtf(v);
    const float sb2 = sqrtf(b2);
    for (int k = from + 1; k <= to; ++k) {
        m *= b1; v *= b2; s *= sb2;
        w -= (lrt[k] * m) * __builtin_amdgcn_rcpf(s + eps);
    }
}

// Scalar (SMEM) load of a wave-uniform entry of a table that no kernel writes (lr_t per step): read through the
// constant address space, the compiler emits s_load_dword and tracks its completion itself.  (A hand-written
// s_load / s_waitcnt pair in inline asm is NOT safe: the compiler may copy the destination SGPR while the load is
// still in flight -- it did, in the row kernels: every replay of two or more steps went wrong.)
typedef const __attribute__((address_space(4))) float* orx_const_fp;
__device__ __forceinline__ float sload(const float* p, int idx) { return ((orx_const_fp)p)[idx]; }

// Replay of ONE element per lane over a wave-uniform range (one row per wavefront: kernels_sharded.hip), and of
// one element with its own range (the table flush).  Rows of a large table wait hundreds or thousands of steps
// between references; the replay stays bounded because the update term lr_k*m/(sqrt(v)+eps) shrinks by >= 9 % per
// step (m by beta_1, the denominator by at most sqrt(beta_2), lr_k grows < 0.5 %/step): once 4x the term no longer
// changes w in fp32 it never will again, and the remaining steps only decay m and v, which is done in closed form
// (m*b1^n, v*b2^n by exp2; relative error ~1e-7 * n*|log2 b|).  At the defaults the loop ends after ~170 steps.
__device__ __forceinline__ void adam_decay_tail(float& m, float& v, int rem, float b1, float b2) {
    if (rem <= 0) return;
    m *= exp2f((float)rem * log2f(b1));
    v *= exp2f((float)rem * log2f(b2));
}

template <bool UNIFORM>
__device__ __forceinline__ void adam_replay1(float& w, float& m, float& v, int from, int to, const float* lrt, float b1, float b2, float eps,
                                             bool newton) {
    if (from >= to) return;
    const float sb2 = sqrtf(b2);
    const float ce = eps * (1.0f - sb2);
    float d = sqrtf(v) + eps;
    float q = __builtin_amdgcn_rcpf(d);
    int k = from + 1;
    if (UNIFORM) {
        for (; k <= to; ++k) {
            const float lr = sload(lrt, k);
            m *= b1; v *= b2; d = d * sb2 + ce;
            q = newton ? q * (2.0f - d * q) : __builtin_amdgcn_rcpf(d);
            const float u = (lr * m) * q;
            w -= u;
            if ((k & 7) == 0 && !__any((w - 4.0f * u) != w)) { ++k; break; }
        }
    } else {
        for (; k <= to; ++k) {
            m *= b1; v *= b2; d = d * sb2 + ce;
            q = newton ? q * (2.0f - d * q) : __builtin_amdgcn_rcpf(d);
            const float u = (lrt[k] * m) * q;
            w -= u;
            if ((w - 4.0f * u) == w) { ++k; break; }
        }
    }
    adam_decay_tail(m, v, to - k + 1, b1, b2);
}

// One row slice (4 elements per lane, plus the item bias carried along) whose stamp is FAR behind: replayed on its
// own, bounded like adam_replay1 (the loop ends once no element of the lane moves any more, the remaining steps decay
// m and v in closed form).  Rows of a large table wait hundreds of steps between references; in the merged loop
// below they would run -- and make the whole wavefront run -- the entire gap.  Costs ~58 VGPRs when inlined, so it
// lives in its own instantiation of the fused kernel (LONGGAP), chosen by the host for tables that are large
// relative to the batch.
constexpr int ORX_ADAM_LONG_GAP = 256;
__device__ __forceinline__ void adam_replay4_bounded(f4& w, f4& m, f4& v, float& bw, float& bm, float& bv, int from, int to,
                                                     const float* lrt, float b1, float b2, float eps) {
    if (from >= to) return;
    const float sb2 = sqrtf(b2);
    const float ce = eps * (1.0f - sb2);
    f4 d = sqrt4(v) + eps;
    float db = sqrtf(bv) + eps;
    int k = from + 1;
    for (; k <= to; ++k) {
        const float lr = lrt[k];
        m = m * b1; v = v * b2; d = d * sb2 + ce;
        bm *= b1; bv *= b2; db = db * sb2 + ce;
        const f4 u = (lr * m) * rcp4(d);
        const float ub = (lr * bm) * __builtin_amdgcn_rcpf(db);
        const f4 wn = w - u;
        const float bn = bw - ub;
        const f4 t = wn - 4.0f * u;
        const bool still = (t.x != wn.x) | (t.y != wn.y) | (t.z != wn.z) | (t.w != wn.w) | ((bn - 4.0f * ub) != bn);
        w = wn; bw = bn;
        if (!still) { ++k; break; }
    }
    const int rem = to - k + 1;
    if (rem > 0) {
        const float f1 = exp2f((float)rem * log2f(b1)), f2 = exp2f((float)rem * log2f(b2));
        m = m * f1; v = v * f2; bm *= f1; bv *= f2;
    }
}

// ---- closed-form replay (round 6).  The gradient-free steps from+1 .. to of an element are
//     m_j = m b1^j,   sqrt(v_j) = a0 rho^j  (a0 = sqrt(v), rho = sqrt(b2)),   w -= sum_{j=1..n} lr_{from+j} m b1^j F(j),   F(j) = 1 / (a0 rho^j + eps).
// b1^j lr_{from+j} decays by ~10 % per step while F moves by at most delta = -ln(rho) = 5e-4 per step: F is expanded to second order at
// j = 0 -- F(j) ~ F0 (1 + delta x j + delta^2 x (2x - 1) j^2 / 2), x = a0 F0 -- and the sum needs only the three moments
// W_q = sum_{j=1..n} b1^j j^q lr_{from+j}, which come from the per-step table V_q[k] = sum_{j>=1} b1^j j^q lr_{k+j} (host, double; orx_adam_lrt):
//     W_0 = V_0[from] - b1^n V_0[to],  W_1 = V_1[from] - b1^n (n V_0[to] + V_1[to]),  W_2 = V_2[from] - b1^n (n^2 V_0[to] + 2 n V_1[to] + V_2[to]).
// Truncation: the third-order term is delta^3 * sum j^3 b1^j / 6 ~ 1e-6 of the sum; cancellation in W_q costs <= 10 ulp.  No loop, no
// divergence between the rows of a wavefront (the merged loop ran every wavefront to the LONGEST gap of its twelve rows: ~3x the mean).
// lr_k itself is not expanded (it moves by per cents per step early in training): the table carries it exactly.
struct AdamCF {
    float W0, W1, W2, p1, p2;                   // the row's moments and decay factors b1^n, b2^n
    __device__ __forceinline__ void setup(const float4* lrv, int from, int to, float4 Vt, float lb1, float lb2) {
        const float n = (float)(to - from);
        const float4 Vf = lrv[from];
        p1 = exp2f(n * lb1); p2 = exp2f(n * lb2);
        W0 = Vf.x - p1 * Vt.x;
        W1 = Vf.y - p1 * (n * Vt.x + Vt.y);
        W2 = Vf.z - p1 * ((n * n) * Vt.x + (2.0f * n) * Vt.y + Vt.z);
    }
    __device__ __forceinline__ void elem(float& w, float& m, float& v, float eps, float delta) const {
        const float a0 = sqrtf(v);
        const float F0 = __builtin_amdgcn_rcpf(a0 + eps);
        const float x = a0 * F0;
        w -= (m * F0) * (W0 + (delta * x) * (W1 + (0.5f * delta) * (2.0f * x - 1.0f) * W2));
        m *= p1; v *= p2;
    }
    __device__ __forceinline__ void row4(f4& w, f4& m, f4& v, float eps, float delta) const {
        float wx = w.x, wy = w.y, wz = w.z, ww = w.w, mx = m.x, my = m.y, mz = m.z, mw = m.w, vx = v.x, vy = v.y, vz = v.z, vw = v.w;
        elem(wx, mx, vx, eps, delta); elem(wy, my, vy, eps, delta); elem(wz, mz, vz, eps, delta); elem(ww, mw, vw, eps, delta);
        w.x = wx; w.y = wy; w.z = wz; w.w = ww; m.x = mx; m.y = my; m.z = mz; m.w = mw; v.x = vx; v.y = vy; v.z = vz; v.w = vw;
    }
};

// adam_replay1 with the closed form where the caller has the moments table (cf.lrv != NULL)
template <bool UNIFORM>
__device__ __forceinline__ void adam_replay1(float& w, float& m, float& v, int from, int to, const float* lrt, float b1, float b2, float eps,
                                             bool newton, const AdamCFParams& cf) {
    if (cf.lrv != nullptr) {
        if (from >= to) return;
        AdamCF c;
        c.setup(cf.lrv, from, to, cf.lrv[to], cf.lb1, cf.lb2);
        c.elem(w, m, v, eps, cf.delta);
        return;
    }
    adam_replay1<UNIFORM>(w, m, v, from, to, lrt, b1, b2, eps, newton);
}

// the three rows (and the two item biases, which share their item row's stamp) of one triplet in ONE loop: its trip
// count is the longest of the three gaps, not their sum, and the three chains interleave.  A row whose gap is
// shorter is masked out of the early iterations.
// The denominator d = sqrt(v) + eps follows d <- sqrt(b2)*d + eps*(1 - sqrt(b2)) (one fma).  NEWTON: its reciprocal
// is carried along by one Newton step per replayed step, r <- r*(2 - d*r): d moves by 1 - sqrt(b2) (5e-4 at the
// default beta_2) per step, so the carried reciprocal stays within (5e-4)^2 = 2.5e-7 of exact with no
// quarter-rate v_rcp in the loop; the caller selects it only for 1 - sqrt(b2) <= 1e-3.
// The loop counter is wave-uniform (it starts at the smallest stamp of the wavefront's active lane groups: the
// hardware loop runs to the longest gap of the wavefront either way), which makes lr_k a scalar load.
template <bool NEWTON, int LPR>
__device__ __forceinline__ void adam_catchup_triplet(f4& wu, f4& mu, f4& vu, int lu, f4& wp, f4& mp, f4& vp, int lp,
                                                     f4& wn, f4& mn, f4& vn, int ln, float& bp, float& mbp, float& vbp,
                                                     float& bn, float& mbn, float& vbn, int to, const float* lrt,
                                                     float b1, float b2, float eps) {
    const int mine = min(lu, min(lp, ln));
    const unsigned long long act = __ballot(1);
    int first = to;
#pragma unroll
    for (int g = 0; g < 64 / LPR; ++g) {
        const int other = __builtin_amdgcn_readlane(mine, g * LPR);
        if ((act >> (g * LPR)) & 1) first = min(first, other);
    }
    if (first >= to) return;
    const float sb2 = sqrtf(b2);
    const float ce = eps * (1.0f - sb2);
    f4 du = sqrt4(vu) + eps, dp = sqrt4(vp) + eps, dn = sqrt4(vn) + eps;
    float dbp = sqrtf(vbp) + eps, dbn = sqrtf(vbn) + eps;
    f4 qu = rcp4(du), qp = rcp4(dp), qn = rcp4(dn);
    float qbp = __builtin_amdgcn_rcpf(dbp), qbn = __builtin_amdgcn_rcpf(dbn);
    for (int k = first + 1; k <= to; ++k) {
        const float lr = sload(lrt, k);
        // a row whose stamp is later than k sits this iteration out (EXEC mask; skipped when no lane group needs it)
        if (k > lu) {
            mu = mu * b1; vu = vu * b2; du = du * sb2 + ce;
            if (NEWTON) qu = qu * (2.0f - du * qu); else qu = rcp4(du);
            wu = wu - (lr * mu) * qu;
        }
        if (k > lp) {
            mp = mp * b1; vp = vp * b2; dp = dp * sb2 + ce;
            mbp *= b1; vbp *= b2; dbp = dbp * sb2 + ce;
            if (NEWTON) { qp = qp * (2.0f - dp * qp); qbp = qbp * (2.0f - dbp * qbp); } else { qp = rcp4(dp); qbp = __builtin_amdgcn_rcpf(dbp); }
            wp = wp - (lr * mp) * qp; bp -= (lr * mbp) * qbp;
        }
        if (k > ln) {
            mn = mn * b1; vn = vn * b2; dn = dn * sb2 + ce;
            mbn *= b1; vbn *= b2; dbn = dbn * sb2 + ce;
            if (NEWTON) { qn = qn * (2.0f - dn * qn); qbn = qbn * (2.0f - dbn * qbn); } else { qn = rcp4(dn); qbn = __builtin_amdgcn_rcpf(dbn); }
            wn = wn - (lr * mn) * qn; bn -= (lr * mbn) * qbn;
        }
    }
}

// the two rows (and the item bias) of one pointwise sample: same scheme as adam_catchup_triplet
template <bool NEWTON, int LPR>
__device__ __forceinline__ void adam_catchup_pair(f4& wu, f4& mu, f4& vu, int lu, f4& wi, f4& mi, f4& vi, int li,
                                                  float& bi, float& mbi, float& vbi, int to, const float* lrt,
                                                  float b1, float b2, float eps) {
    const int mine = min(lu, li);
    const unsigned long long act = __ballot(1);
    int first = to;
#pragma unroll
    for (int g = 0; g < 64 / LPR; ++g) {
        const int other = __builtin_amdgcn_readlane(mine, g * LPR);
        if ((act >> (g * LPR)) & 1) first = min(first, other);
    }
    if (first >= to) return;
    const float sb2 = sqrtf(b2);
    const float ce = eps * (1.0f - sb2);
    f4 du = sqrt4(vu) + eps, di = sqrt4(vi) + eps;
    float dbi = sqrtf(vbi) + eps;
    f4 qu = rcp4(du), qi = rcp4(di);
    float qbi = __builtin_amdgcn_rcpf(dbi);
    for (int k = first + 1; k <= to; ++k) {
        const float lr = sload(lrt, k);
        if (k > lu) {
            mu = mu * b1; vu = vu * b2; du = du * sb2 + ce;
            if (NEWTON) qu = qu * (2.0f - du * qu); else qu = rcp4(du);
            wu = wu - (lr * mu) * qu;
        }
        if (k > li) {
            mi = mi * b1; vi = vi * b2; di = di * sb2 + ce;
            mbi *= b1; vbi *= b2; dbi = dbi * sb2 + ce;
            if (NEWTON) { qi = qi * (2.0f - di * qi); qbi = qbi * (2.0f - dbi * qbi); } else { qi = rcp4(di); qbi = __builtin_amdgcn_rcpf(dbi); }
            wi = wi - (lr * mi) * qi; bi -= (lr * mbi) * qbi;
        }
    }
}

// ------------------------------------------------- duplicated-row deposits ---
// Gradient of a duplicated reference.  Rows referenced exactly twice in the batch get one PLAIN
// store per reference, into scratch row 1 (role 0) or scratch row 2 (role 1): no atomics and a
// summation that is bitwise reproducible.  Rows referenced three or more times (role 2) use
// fp32 atomics into scratch row 1.  dup_apply_kernel adds the two scratch rows.
__device__ __forceinline__ void dup_store4(float* G1, float* G2, size_t off, f4 g, int role) {
    if (role == 0) *reinterpret_cast<f4*>(G1 + off) = g;
    else if (role == 1) *reinterpret_cast<f4*>(G2 + off) = g;
    else atomic_add_f4(G1 + off, g);
}

__device__ __forceinline__ void dup_store1(float* G1, float* G2, size_t off, float g, int role) {
    if (role == 0) G1[off] = g;
    else if (role == 1) G2[off] = g;
    else unsafeAtomicAdd(G1 + off, g);
}

// with a staging plan (slot >= 0) a role-2 reference owns one staging slot: plain store, no atomics
__device__ __forceinline__ void dup_store4s(float* G1, float* G2, size_t off, f4 g, int role, float* stage, int slot, int D, int sub) {
    if (role == 3) return;          // pairing: the partner lane group writes the row (fused_kernel)
    if (role == 2 && slot >= 0) *reinterpret_cast<f4*>(stage + (size_t)slot * D + 4 * sub) = g;
    else dup_store4(G1, G2, off, g, role);
}

__device__ __forceinline__ void dup_store1s(float* G1, float* G2, size_t off, float g, int role, float* stageb, int slot) {
    if (role == 3) return;
    if (role == 2 && slot >= 0) stageb[slot] = g;
    else dup_store1(G1, G2, off, g, role);
}

