// Device code shared by the fused pairwise and pointwise kernels: the in-launch application of the previous step's duplicated
// rows (apply blocks at the front of a fused launch) with its write-through hand-off, the staging-segment sums, the censor.
#pragma once
#include "orx_device.h"

// sum of a staging segment, one float4 column slice per lane (LPR lanes per row)
template <int D>
__device__ __forceinline__ f4 segment_sum4(const float* stage, int seg, int cnt, int sub) {
    const float* p = stage + (size_t)seg * D + 4 * sub;
    f4 s0, s1; s0.x = s0.y = s0.z = s0.w = 0.0f; s1 = s0;
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {
        const f4 a0 = *reinterpret_cast<const f4*>(p + (size_t)(k + 0) * D);
        const f4 a1 = *reinterpret_cast<const f4*>(p + (size_t)(k + 1) * D);
        const f4 a2 = *reinterpret_cast<const f4*>(p + (size_t)(k + 2) * D);
        const f4 a3 = *reinterpret_cast<const f4*>(p + (size_t)(k + 3) * D);
        s0 = s0 + (a0 + a1); s1 = s1 + (a2 + a3);
    }
    for (; k < cnt; ++k) s0 = s0 + *reinterpret_cast<const f4*>(p + (size_t)k * D);
    return s0 + s1;
}

// sum of the staged bias gradients of a segment, spread over the LPR lanes of the row's group
template <int LPR>
__device__ __forceinline__ float segment_sum1(const float* stageb, int seg, int cnt, int sub) {
    float s = 0.0f;
    for (int k = sub; k < cnt; k += LPR) s += stageb[seg + k];
    return group_allreduce<LPR>(s);
}

// ---- in-launch application of the previous step's duplicated rows ------------------------
// The first `n_apply_blocks` blocks of a fused launch of step s apply the summed gradients
// that step s-1 left in the scratch rows (what dup_apply_kernel does as a launch of its own).
// Hand-off to the references of step s that read such a row (marked "urgent" by
// urgent_kernel), following the producer/consumer recipe of the CDNA guide:
//   producer: row / accumulator / scratch-zero stores WRITE-THROUGH (agent-scope stores),
//             s_waitcnt vmcnt(0), then ONE lane stores the row's ready flag (= epoch of this launch)
//   consumer: one lane polls the flag with relaxed agent-scope loads, then an agent-scope acquire
//             fence, then plain loads.
// Apply blocks have the lowest block indices (dispatched first) and never wait, so a consumer
// cannot starve them.
__device__ __forceinline__ void store_wt(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_wt4(float* p, f4 v) {
    // one 16-byte write-through (sc1) store; the caller drains it with s_waitcnt vmcnt(0).
    // The s_nop covers the hardware hazard "VALU write of the data VGPRs right after a >64-bit VMEM
    // store" (2 wait states): the compiler's hazard recognizer does not look inside inline asm, and
    // without it a following v_mov into the same registers corrupted the stored row.
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

template <int OPT>
__device__ __forceinline__ float opt_rule(float w, float g, float& acc, float lr, float eps) {
    if (OPT == ORX_ADAGRAD) { acc = acc + g * g; return w - lr * g / (sqrtf(acc) + eps); }
    return w - lr * g;
}

// LatentFactor.censor (latent_factor.py:17-23) of a row held by LPR lanes: row / max(||row||, min_norm)
template <int LPR>
__device__ __forceinline__ f4 censor4(f4 w, float min_norm) {
    const float m = fmaxf(sqrtf(group_allreduce<LPR>(dot4(w, w))), min_norm);
    f4 r; r.x = w.x / m; r.y = w.y / m; r.z = w.z / m; r.w = w.w / m;
    return r;
}

// was this item row referenced as a positive AND as a negative in the step of epoch `ep`?
// (censor_vec censors the row once per id list, ucml.py:46-48)
__device__ __forceinline__ bool censored_twice(const PairArgs& a, size_t row, int ep) {
    const int2 m = *reinterpret_cast<const int2*>(a.sideV + 2 * row);
    return m.x == ep && m.y == ep;
}

template <int LPR, int OPT, bool CENSOR, bool STAGED>
__device__ __forceinline__ void inline_apply(const PairArgs& a) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int n = *a.prev_dcount;
    const int64_t stride = (int64_t)a.n_apply_blocks * 4 * TPW;
    for (int64_t e = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * TPW + grp; e < n; e += stride) {
        const uint32_t ent = a.prev_dlist[e];
        if (ent == ORX_DLIST_DEAD) continue;            // (the row was paired after all: updated in place by step s-1's launch)
        const bool item = (ent >> 31) != 0;
        const size_t row = ent & 0x7fffffffu;
        float* W = item ? a.V : a.U;
        float* A = item ? a.aV : a.aU;
        float* g1 = (item ? a.gV : a.gU) + row * D + 4 * sub;
        float* g2 = (item ? a.gV2 : a.gU2) + row * D + 4 * sub;
        float* wp = W + row * D + 4 * sub;
        // staged row (>= 3 references): sum its segment; the scratch rows were not used
        const int scnt = STAGED ? a.prev_dcnt[e] : 0;
        const int sseg = STAGED && scnt > 0 ? a.prev_dseg[e] : 0;
        f4 g;
        if (STAGED && scnt > 0) g = segment_sum4<D>(a.prev_stage, sseg, scnt, sub);
        else g = *reinterpret_cast<const f4*>(g1) + *reinterpret_cast<const f4*>(g2);
        const f4 w = *reinterpret_cast<const f4*>(wp);
        f4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
        if (OPT == ORX_ADAGRAD || OPT == ORX_ADAM) acc = *reinterpret_cast<const f4*>(A + row * D + 4 * sub);
        float ac[4] = {acc.x, acc.y, acc.z, acc.w};
        f4 wn;
        f4 vv; vv.x = vv.y = vv.z = vv.w = 0.0f;           // Adam v (acc holds m)
        if (OPT == ORX_ADAM) {
            // lazy Adam: the rows of step s-1 (optimizer step step_t - 1) are first caught up, then take their step
            float* A2 = item ? a.a2V : a.a2U;
            int* L = item ? a.lastV : a.lastU;
            vv = *reinterpret_cast<const f4*>(A2 + row * D + 4 * sub);
            wn = w;
            const int tp = a.step_t - 1;
            adam_catchup4(wn, acc, vv, L[row], tp - 1, a.lrt, a.b1, a.b2, a.eps);
            adam_elem4(wn, acc, vv, g, a.lrt[tp], a.b1, a.b2, a.eps);
            store_wt4(A2 + row * D + 4 * sub, vv);
            if (sub == 0) __hip_atomic_store(L + row, tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
        wn.x = opt_rule<OPT>(w.x, g.x, ac[0], a.lr, a.eps); wn.y = opt_rule<OPT>(w.y, g.y, ac[1], a.lr, a.eps);
        wn.z = opt_rule<OPT>(w.z, g.z, ac[2], a.lr, a.eps); wn.w = opt_rule<OPT>(w.w, g.w, ac[3], a.lr, a.eps);
        acc.x = ac[0]; acc.y = ac[1]; acc.z = ac[2]; acc.w = ac[3];
        }
        if (CENSOR) {                   // the rows of step s-1 are censored where they are applied
            wn = censor4<LPR>(wn, a.min_norm);
            if (item && censored_twice(a, row, a.epoch - 1)) wn = censor4<LPR>(wn, a.min_norm);
        }
        f4 z; z.x = z.y = z.z = z.w = 0.0f;
        store_wt4(wp, wn);
        if (!STAGED || scnt <= 0) { store_wt4(g1, z); store_wt4(g2, z); }
        if (OPT == ORX_ADAGRAD || OPT == ORX_ADAM) store_wt4(A + row * D + 4 * sub, acc);
        float gbs = 0.0f;
        if (STAGED && scnt > 0 && item) gbs = segment_sum1<LPR>(a.prev_stageb, sseg, scnt, sub);
        if (item && sub == 0) {
            const float gb = STAGED && scnt > 0 ? gbs : a.gb[row] + a.gb2[row];
            float ab = (OPT == ORX_ADAGRAD || OPT == ORX_ADAM) ? a.ab[row] : 0.0f;
            float bn;
            if (OPT == ORX_ADAM) {
                float bw = a.b[row], bv = a.a2b[row];
                const int tp = a.step_t - 1;
                adam_catchup1(bw, ab, bv, a.lastb[row], tp - 1, a.lrt, a.b1, a.b2, a.eps);
                adam_elem(bw, ab, bv, gb, a.lrt[tp], a.b1, a.b2, a.eps);
                bn = bw;
                store_wt(a.a2b + row, bv);
                __hip_atomic_store(a.lastb + row, tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                bn = opt_rule<OPT>(a.b[row], gb, ab, a.lr, a.eps);
            }
            store_wt(a.b + row, bn);
            if (!STAGED || scnt <= 0) { store_wt(a.gb + row, 0.0f); store_wt(a.gb2 + row, 0.0f); }
            if (OPT == ORX_ADAGRAD || OPT == ORX_ADAM) store_wt(a.ab + row, ab);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this wave has been written through
        if (sub == 0) __hip_atomic_store((item ? a.readyV : a.readyU) + row, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void wait_ready(const int* flag, int epoch) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(4);
}

