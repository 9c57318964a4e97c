// Device-side helpers shared by the kernel translation units (gfx950).
#pragma once
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
    if (role == 2 && slot >= 0) *reinterpret_cast<f4*>(stage + (size_t)slot * D + 4 * sub) = g;
    else dup_store4(G1, G2, off, g, role);
}

__device__ __forceinline__ void dup_store1s(float* G1, float* G2, size_t off, float g, int role, float* stageb, int slot) {
    if (role == 2 && slot >= 0) stageb[slot] = g;
    else dup_store1(G1, G2, off, g, role);
}

