// Fused pointwise (GMF / WRMF) train step and the all-item scorer (inference).
//
//   GMF.call   recommenders/gmf.py:22-34   logit = Dense(1,no bias)(u*i) + b_i,
//                                          mean BCE-with-logits, l2 over u, i and the kernel w
//   WRMF.call  recommenders/wrmf.py:21-34 + modules/pointwise_mse_loss.py:18-31
//                                          pred = u.i + b_i, sum of c*(y-pred)^2, c = (a-b)*y + b
//   inference  bpr.py:39-43, wrmf.py:36-40 (dot), ucml.py:50-53 (-L2), gmf.py:36-41 (weighted dot)
//
// Same structure as kernels_pairwise.hip: LPR = D/4 lanes own one row, unique rows
// are updated in place, duplicated rows (flags from dedup_kernel) accumulate into
// gsum and are finished by dup_apply_kernel.  GMF's dense kernel gradient
// (a [D] vector summed over the batch) leaves the fused kernel as one partial per
// wavefront and is reduced + applied by dense_apply_kernel.
#include "orx_apply_device.h"

// per-sample loss term and d(loss)/d(score)
template <int MODEL>
__device__ __forceinline__ void point_score(float s, float y, float invB, float a_w, float b_w, float& term, float& gs, int sigmoid = 0) {
    if (MODEL == ORX_GMF) {
        const float e = __expf(-fabsf(s));
        term = (fmaxf(s, 0.0f) - s * y + log1pf(e)) * invB;            // BCE with logits, mean
        const float sig = (s >= 0.0f) ? 1.0f / (1.0f + e) : e / (1.0f + e);
        gs = (sig - y) * invB;
    } else {
        const float c = (a_w - b_w) * y + b_w;                         // pointwise_mse_loss.py:30
        float pred = s, dpred = 1.0f;
        if (sigmoid) {                                                 // :24-25  pred = sigmoid(dot + bias)
            const float e = __expf(-fabsf(s));
            pred = (s >= 0.0f) ? 1.0f / (1.0f + e) : e / (1.0f + e);
            dpred = pred * (1.0f - pred);
        }
        const float r = y - pred;
        term = c * r * r;                                              // :31 (sum)
        gs = -2.0f * c * r * dpred;
    }
}

// GMF's Dense(1) gradient INSIDE the launch (round 6).  Until now: one partial row per workgroup, then dense_reduce1_kernel and
// dense_reduce_kernel between this step's launch and the next one's -- 4.7 + 5.7 us of a 32.8 us step at B = 65 536, on the critical path
// (the next step reads the updated kernel).  Now the launch carries ceil(workgroups / 64) + 1 REDUCER workgroups behind its sample
// workgroups (highest block indices: dispatched last, when everything they wait for is resident or done).  Reducer g adds the rows of
// sample workgroups 64 g .. 64 g + 63 in index order into one group row; the last reducer adds the group rows and applies the dense rule
// (what dense_reduce_kernel did).  Fixed order of additions: the step stays bit-reproducible.
// Hand-off without any wait on the producers' side: a slot that holds WT_EMPTY (all ones: a NaN no arithmetic produces) is "not there
// yet"; a producer stores its row write-through and leaves; a reducer polls the slots with agent-scope loads and hands them back empty.
// What was tried first, per launch at B = 65 536 (23.8 us without any of it): last-arriver counters with release / acquire fences 580 us
// (an agent-scope release on gfx950 writes the whole L2 back, and this kernel keeps tens of MB of updated rows there); counters behind
// s_waitcnt vmcnt(0) 46.7 us (every wavefront then sits on its slot until its scattered row stores are acknowledged).
constexpr int WT_GROUP = 64;
constexpr uint32_t WT_EMPTY = 0xffffffffu;
constexpr int WT_SPIN_MAX = 1 << 22;         // (a value that IS all ones -- only a NaN fed in from outside can be -- is taken after this many polls)

template <int LPR>
__device__ __forceinline__ void dense_tail_reducer(const PointArgs& a, int red, int nwg) {
    constexpr int D = 4 * LPR, NSL = 256 / LPR, RB = 4;
    __shared__ f4 wt_red[256];
    const int sub = threadIdx.x % LPR, sl = threadIdx.x / LPR;
    const int ngrp = (nwg + WT_GROUP - 1) / WT_GROUP;
    // rows first .. first + n - 1 of src are awaited, added in a fixed order (thread slices of every NSL-th row, RB rows in flight, then the
    // slices in order) and handed back empty; the sum is valid in the threads of slice 0
    auto take_rows = [&](float* src, int first, int n) -> f4 {
        f4 s; s.x = s.y = s.z = s.w = 0.0f;
        for (int r0 = sl; r0 < n; r0 += NSL * RB) {
            uint32_t v[RB][4];
#pragma unroll
            for (int k = 0; k < RB; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = WT_EMPTY;
            for (int spins = 0; spins < WT_SPIN_MAX; ++spins) {
                bool again = false;
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const int r = r0 + k * NSL;
                    if (r >= n) continue;
                    uint32_t* q = reinterpret_cast<uint32_t*>(src + (size_t)(first + r) * D + 4 * sub);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (v[k][e] == WT_EMPTY) {
                            v[k][e] = __hip_atomic_load(q + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            again |= v[k][e] == WT_EMPTY;
                        }
                }
                if (!again) break;
                __builtin_amdgcn_s_sleep(8);
            }
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const int r = r0 + k * NSL;
                if (r >= n) continue;
                f4 x, e4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = __uint_as_float(v[k][e]); e4[e] = __uint_as_float(WT_EMPTY); }
                s += x;
                store_wt4(src + (size_t)(first + r) * D + 4 * sub, e4);      // (write-through like the producers' stores: rows of 64 bytes share a line)
            }
        }
        wt_red[threadIdx.x] = s;
        __syncthreads();
        f4 t; t.x = t.y = t.z = t.w = 0.0f;
        if (sl == 0) for (int k = 0; k < NSL; ++k) t += wt_red[k * LPR + sub];
        __syncthreads();
        return t;
    };
    if (red < ngrp) {
        const int in_grp = nwg - red * WT_GROUP < WT_GROUP ? nwg - red * WT_GROUP : WT_GROUP;
        const f4 t = take_rows(a.wpartial, red * WT_GROUP, in_grp);
        if (sl == 0) store_wt4(a.wt_rows + (size_t)red * D + 4 * sub, t);
        return;
    }
    const f4 t = take_rows(a.wt_rows, 0, ngrp);
    if (threadIdx.x < LPR) {                       // (slice 0 = the first LPR lanes of wavefront 0) the rule of dense_reduce_kernel
        float* wp = const_cast<float*>(a.w) + 4 * sub;
        const f4 we = *reinterpret_cast<const f4*>(wp);
        float wsq = dot4(we, we);
        const f4 g = t + a.l2w * we;
        if (a.wt_optkind == ORX_SGD) {
            *reinterpret_cast<f4*>(wp) = we - a.lr * g;
        } else if (a.wt_optkind == ORX_ADAGRAD) {
            f4 a2 = *reinterpret_cast<const f4*>(a.wt_acc + 4 * sub);
            f4 wn;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a2[e] = a2[e] + g[e] * g[e]; wn[e] = we[e] - a.lr * g[e] / (sqrtf(a2[e]) + a.eps); }
            *reinterpret_cast<f4*>(a.wt_acc + 4 * sub) = a2;
            *reinterpret_cast<f4*>(wp) = wn;
        } else if (a.wt_gout != nullptr) {
            *reinterpret_cast<f4*>(a.wt_gout + 4 * sub) = g;
        }
        for (int off = LPR / 2; off > 0; off >>= 1) wsq += __shfl_xor(wsq, off);
        if (threadIdx.x == 0 && a.wt_l2slot != nullptr) { a.wt_l2slot[0] = 0.0f; a.wt_l2slot[1] = 0.5f * wsq; }
    }
}

// PAIR: the instantiation with the pairing tail (97 against 73 VGPRs for GMF at D = 64 -- two wavefronts of occupancy: the host launches it
// only for steps whose plan paired, a.ids4 != NULL)
template <int LPR, int MODEL, int OPT, int MODE, bool PAIR = false>
__global__ __launch_bounds__(256) void point_fused_kernel(PointArgs a) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    // pairing (round 6; the pairwise step's machinery, kernels_plan.hip + fused_kernel): the two samples of a row referenced exactly
    // twice in the step meet in a wavefront, leave their gradients of the row in LDS, and ONE of them writes the row -- no deposit,
    // no apply of that row.  SGD only (as there); a sample has two id slots (0 user, 1 item), its label rides in the record's word z.
    constexpr bool PAIRS = PAIR && MODE == MODE_EXACT && OPT == ORX_SGD && TPW > 1;
    __shared__ f4 pair_xg[PAIRS ? 256 : 1];            // gradient exchange, one slot per lane
    __shared__ float pair_xb[PAIRS ? 256 / LPR : 1];   // ... and per lane group (item bias)
    __shared__ f4 pair_xw[PAIRS ? 256 : 1];            // the writer's copy of the shared row as read
    __shared__ float pair_xwb[PAIRS ? 256 / LPR : 1];
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int nab = MODE == MODE_EXACT ? a.n_apply_blocks : 0;
    if (MODE == MODE_EXACT && (int)blockIdx.x < nab) {          // apply role (block-uniform): step s-1's duplicated rows
        if (a.ap.prev_dcnt != nullptr) inline_apply<LPR, OPT, false, true>(a.ap);
        else inline_apply<LPR, OPT, false, false>(a.ap);
        return;
    }
    const int nwg = (int)gridDim.x - nab - (MODEL == ORX_GMF ? a.wt_nred : 0);      // sample workgroups
    if (MODEL == ORX_GMF && MODE != MODE_LOSS && (int)blockIdx.x - nab >= nwg) {        // reducer role (block-uniform): see dense_tail_reducer
        dense_tail_reducer<LPR>(a, (int)blockIdx.x - nab - nwg, nwg);
        return;
    }
    const int64_t wave_global = (int64_t)(blockIdx.x - nab) * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)nwg * 4 * TPW;
    float loss_acc = 0.0f, sq_acc = 0.0f;
    f4 wv; wv.x = wv.y = wv.z = wv.w = 1.0f;
    if (MODEL == ORX_GMF) wv = *reinterpret_cast<const f4*>(a.w + 4 * sub);
    f4 gw_acc; gw_acc.x = gw_acc.y = gw_acc.z = gw_acc.w = 0.0f;
    for (int64_t t = wave_global * TPW + grp; t < a.B; t += stride) {
        int u, i; float y;
        uint32_t pinfo = 0u;                    // pairing word of the sample processed at position t
        int64_t t0 = t;                         // where the sample stood (its staging records are indexed by that)
        bool packed = false;
        if (PAIRS) packed = a.ids4 != nullptr;
        if (packed) {
            const int4 v = a.ids4[t];
            u = v.x; i = v.y; y = __int_as_float(v.z);
            pinfo = (uint32_t)v.w & 0x3ffu; t0 = (int64_t)((uint32_t)v.w >> 10);
        } else {
            u = a.uid[t]; i = a.iid[t]; y = a.label[t];
        }
        int du = 0, di = 0;
        int ku = 2, ki = 2;                     // duplicate role: 0 / 1 = plain store into scratch row 1 / 2, 2 = staged or atomics
        int urgent = 0;
        if (MODE == MODE_EXACT) {
            if (a.role_bits) {                  // ids rewritten by dedup_kernel
                du = (uint32_t)u >> 31; di = (uint32_t)i >> 31;
                ku = ((uint32_t)u >> 29) & 3; ki = ((uint32_t)i >> 29) & 3;
                urgent = (((uint32_t)u >> 28) & 1) | (((uint32_t)i >> 27) & 2);
                u &= 0x0fffffff; i &= 0x0fffffff;
            } else {
                du = a.dflag[t]; di = a.dflag[a.B + t];
            }
        }
        if (MODE == MODE_ACCUM) { du = di = 1; }
        if (!(id_ok(u, a.NU) & id_ok(i, a.NI))) {
            if (sub == 0) *a.err = 1;
            if (PAIRS) {
                if (pinfo & ORX_PAIR_VALID) {   // (its partner must not add what an earlier iteration left in LDS)
                    f4 z; z.x = z.y = z.z = z.w = 0.0f;
                    pair_xg[threadIdx.x] = z;
                    if (sub == 0) pair_xb[threadIdx.x / LPR] = 0.0f;
                }
            }
            continue;
        }
        if (MODE == MODE_EXACT && urgent && nab) {      // a row of this sample is being updated by an apply block of this launch
            if (sub == 0) {
                if (urgent & 1) wait_ready(a.readyU + u, a.epoch);
                if (urgent & 2) wait_ready(a.readyV + i, a.epoch);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        float* Up = a.U + (size_t)u * D + 4 * sub;
        float* Ip = a.V + (size_t)i * D + 4 * sub;
        f4 ru = *reinterpret_cast<const f4*>(Up);
        f4 ri = *reinterpret_cast<const f4*>(Ip);
        float bi = a.b[i];
        // lazy Adam: (w, m, v) of the two rows and the bias, replayed up to the step before this one
        f4 mu, vu, mi, vi;
        float mbi = 0.f, vbi = 0.f;
        if (OPT == ORX_ADAM) {
            mu = *reinterpret_cast<const f4*>(a.aU + (size_t)u * D + 4 * sub); vu = *reinterpret_cast<const f4*>(a.a2U + (size_t)u * D + 4 * sub);
            mi = *reinterpret_cast<const f4*>(a.aV + (size_t)i * D + 4 * sub); vi = *reinterpret_cast<const f4*>(a.a2V + (size_t)i * D + 4 * sub);
            mbi = a.ab[i]; vbi = a.a2b[i];
            const int lu = a.lastU[u], li = a.lastV[i];
            if (a.lrv != nullptr) {                 // closed-form replay (no loop over the skipped steps)
                const int T1 = a.step_t - 1;
                const float4 Vt = a.lrv[T1];
                AdamCF cf;
                if (lu < T1) { cf.setup(a.lrv, lu, T1, Vt, a.cf_lb1, a.cf_lb2); cf.row4(ru, mu, vu, a.eps, a.cf_delta); }
                if (li < T1) { cf.setup(a.lrv, li, T1, Vt, a.cf_lb1, a.cf_lb2); cf.row4(ri, mi, vi, a.eps, a.cf_delta); cf.elem(bi, mbi, vbi, a.eps, a.cf_delta); }
            }
            else if (a.newton) adam_catchup_pair<true, LPR>(ru, mu, vu, lu, ri, mi, vi, li, bi, mbi, vbi, a.step_t - 1, a.lrt, a.b1, a.b2, a.eps);
            else adam_catchup_pair<false, LPR>(ru, mu, vu, lu, ri, mi, vi, li, bi, mbi, vbi, a.step_t - 1, a.lrt, a.b1, a.b2, a.eps);
        }
        const f4 ui = ru * ri;
        const float s = group_allreduce<LPR>(dot4(ui, wv)) + bi;
        float term, gs;
        point_score<MODEL>(s, y, a.invB, a.a_w, a.b_w, term, gs, a.sigmoid);
        sq_acc += dot4(ru, ru) + dot4(ri, ri);
        if (sub == 0) loss_acc += term;
        if (MODE == MODE_LOSS) continue;
        const f4 gu = gs * (ri * wv) + a.l2w * ru;
        const f4 gi = gs * (ru * wv) + a.l2w * ri;
        if (MODEL == ORX_GMF) gw_acc += gs * ui;
        if (PAIRS) {
            if (pinfo & ORX_PAIR_VALID) {
                const int myslot = (pinfo >> 4) & 3;
                pair_xg[threadIdx.x] = myslot == 0 ? gu : gi;
                if (pinfo & ORX_PAIR_WRITER) pair_xw[threadIdx.x] = myslot == 0 ? ru : ri;
                if (sub == 0) {
                    pair_xb[threadIdx.x / LPR] = gs;
                    if (pinfo & ORX_PAIR_WRITER) pair_xwb[threadIdx.x / LPR] = bi;
                }
                if (myslot == 0) { du = 1; ku = 3; } else { di = 1; ki = 3; }      // the slot is settled: no store below
            }
        }
        // staged references: slot = segment start of the row + rank of the reference (refinfo (-1, 0): no plan)
        auto slot_of = [&](int64_t ref) -> int {
            if (MODE != MODE_EXACT || a.stage == nullptr) return -1;
            const int2 ri2 = a.refinfo[ref];
            return ri2.x < 0 ? -1 : a.segstart[ri2.x] + ri2.y;
        };
        if (OPT == ORX_ADAM) {          // a row referenced once takes its step here, a duplicated one deposits its gradient
            const float lrT = a.lrt[a.step_t];
            if (du == 0) {
                adam_elem4(ru, mu, vu, gu, lrT, a.b1, a.b2, a.eps);
                *reinterpret_cast<f4*>(Up) = ru; *reinterpret_cast<f4*>(a.aU + (size_t)u * D + 4 * sub) = mu;
                *reinterpret_cast<f4*>(a.a2U + (size_t)u * D + 4 * sub) = vu;
                if (sub == 0) a.lastU[u] = a.step_t;
            } else dup_store4s(a.gU, a.gU2, (size_t)u * D + 4 * sub, gu, ku, a.stage, ku == 2 ? slot_of(t) : -1, D, sub);
            if (di == 0) {
                adam_elem4(ri, mi, vi, gi, lrT, a.b1, a.b2, a.eps);
                *reinterpret_cast<f4*>(Ip) = ri; *reinterpret_cast<f4*>(a.aV + (size_t)i * D + 4 * sub) = mi;
                *reinterpret_cast<f4*>(a.a2V + (size_t)i * D + 4 * sub) = vi;
                if (sub == 0) {
                    adam_elem(bi, mbi, vbi, gs, lrT, a.b1, a.b2, a.eps);
                    a.b[i] = bi; a.ab[i] = mbi; a.a2b[i] = vbi; a.lastV[i] = a.step_t; a.lastb[i] = a.step_t;
                }
            } else {
                const int si = ki == 2 ? slot_of((a.iid - a.uid) + t) : -1;
                dup_store4s(a.gV, a.gV2, (size_t)i * D + 4 * sub, gi, ki, a.stage, si, D, sub);
                if (sub == 0) dup_store1s(a.gb, a.gb2, i, gs, ki, a.stageb, si);
            }
            continue;
        }
        if (du == 0) opt_apply4<OPT>(Up, a.aU + (size_t)u * D + 4 * sub, ru, gu, a.lr, a.eps);
        else dup_store4s(a.gU, a.gU2, (size_t)u * D + 4 * sub, gu, ku, a.stage, ku == 2 ? slot_of(t0) : -1, D, sub);
        if (di == 0) {
            opt_apply4<OPT>(Ip, a.aV + (size_t)i * D + 4 * sub, ri, gi, a.lr, a.eps);
            if (sub == 0) opt_apply1<OPT>(a.b + i, a.ab + i, bi, gs, a.lr, a.eps);
        } else {
            const int si = ki == 2 ? slot_of((a.iid - a.uid) + t0) : -1;
            dup_store4s(a.gV, a.gV2, (size_t)i * D + 4 * sub, gi, ki, a.stage, si, D, sub);
            if (sub == 0) dup_store1s(a.gb, a.gb2, i, gs, ki, a.stageb, si);
        }
        if (PAIRS) {
            // the writer of a pair: TF sums the gradients of duplicate indices before the sparse apply -- the two gradients
            // are added and the rule is applied once, to the row as this lane group read it (a wavefront's LDS operations execute in order)
            if (pinfo & ORX_PAIR_WRITER) {
                const int myslot = (pinfo >> 4) & 3;
                const int xsrc = (int)(threadIdx.x & ~63u) + (int)(pinfo & 15u) * LPR + sub;
                const int id = myslot == 0 ? u : i;
                const size_t off = (size_t)id * D + 4 * sub;
                const f4 gsum2 = pair_xg[threadIdx.x] + pair_xg[xsrc];
                opt_apply4<OPT>((myslot == 0 ? a.U : a.V) + off, (myslot == 0 ? a.aU : a.aV) + off, pair_xw[threadIdx.x], gsum2, a.lr, a.eps);
                if (myslot != 0 && sub == 0)
                    opt_apply1<OPT>(a.b + id, a.ab + id, pair_xwb[threadIdx.x / LPR], pair_xb[threadIdx.x / LPR] + pair_xb[xsrc / LPR], a.lr, a.eps);
            }
        }
    }
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        float2 v; v.x = ls; v.y = 0.5f * sq;
        *reinterpret_cast<float2*>(a.partial + 2 * wave_global) = v;
    }
    if (MODEL == ORX_GMF && MODE != MODE_LOSS) {
        // sum the lane groups of the wavefront (lanes with equal `sub`), one [D] partial per wave
        for (int off = LPR; off < 64; off <<= 1) {
            gw_acc.x += __shfl_xor(gw_acc.x, off); gw_acc.y += __shfl_xor(gw_acc.y, off);
            gw_acc.z += __shfl_xor(gw_acc.z, off); gw_acc.w += __shfl_xor(gw_acc.w, off);
        }
        // ... and the workgroup's four wavefronts through LDS: one [D] partial per WORKGROUP (1 MB instead of 4 MB per step at B = 65 536 for the
        // reduce launches to read; every wavefront of a sample block arrives here)
        __shared__ f4 gwsh[4][LPR];
        if (grp == 0) gwsh[threadIdx.x >> 6][sub] = gw_acc;
        __syncthreads();
        if (threadIdx.x < LPR) {
            const f4 t = (gwsh[0][sub] + gwsh[1][sub]) + (gwsh[2][sub] + gwsh[3][sub]);
            if (a.wt_nred > 0) store_wt4(a.wpartial + (size_t)(blockIdx.x - nab) * D + 4 * sub, t);      // (a reducer of this launch is waiting for it)
            else *reinterpret_cast<f4*>(a.wpartial + (size_t)(blockIdx.x - nab) * D + 4 * sub) = t;
        }
    }
}

// any D: one sample per wavefront, scalar elements
template <int MODEL, int OPT, int MODE>
__global__ __launch_bounds__(256) void point_generic_kernel(PointArgs a) {
    const int lane = threadIdx.x & 63;
    const int D = a.D;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    float loss_acc = 0.0f, sq_acc = 0.0f;
    float* wp = a.wpartial + (size_t)wave_global * D;
    if (MODEL == ORX_GMF && MODE != MODE_LOSS) for (int e = lane; e < D; e += 64) wp[e] = 0.0f;
    for (int64_t t = wave_global; t < a.B; t += stride) {
        const int u = a.uid[t], i = a.iid[t];
        const float y = a.label[t];
        int du = 0, di = 0;
        if (MODE == MODE_EXACT) { du = a.dflag[t]; di = a.dflag[a.B + t]; }
        if (MODE == MODE_ACCUM) { du = di = 1; }
        if (!(id_ok(u, a.NU) & id_ok(i, a.NI))) { if (lane == 0) *a.err = 1; continue; }
        float* Ur = a.U + (size_t)u * D;
        float* Ir = a.V + (size_t)i * D;
        const float bi = a.b[i];
        float part = 0.0f;
        for (int e = lane; e < D; e += 64) {
            const float x = Ur[e], z = Ir[e];
            part += x * z * (MODEL == ORX_GMF ? a.w[e] : 1.0f);
            sq_acc += x * x + z * z;
        }
        const float s = wave_sum(part) + bi;
        float term, gs;
        point_score<MODEL>(s, y, a.invB, a.a_w, a.b_w, term, gs, a.sigmoid);
        if (lane == 0) loss_acc += term;
        if (MODE == MODE_LOSS) continue;
        for (int e = lane; e < D; e += 64) {
            const float x = Ur[e], z = Ir[e];
            const float we = MODEL == ORX_GMF ? a.w[e] : 1.0f;
            const float gu = gs * z * we + a.l2w * x, gi = gs * x * we + a.l2w * z;
            if (MODEL == ORX_GMF) wp[e] += gs * x * z;
            if (du == 0) opt_apply1<OPT>(Ur + e, a.aU + (size_t)u * D + e, x, gu, a.lr, a.eps);
            else unsafeAtomicAdd(a.gU + (size_t)u * D + e, gu);
            if (di == 0) opt_apply1<OPT>(Ir + e, a.aV + (size_t)i * D + e, z, gi, a.lr, a.eps);
            else unsafeAtomicAdd(a.gV + (size_t)i * D + e, gi);
        }
        if (lane == 0) {
            if (di == 0) opt_apply1<OPT>(a.b + i, a.ab + i, bi, gs, a.lr, a.eps);
            else unsafeAtomicAdd(a.gb + i, gs);
        }
    }
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        float2 v; v.x = ls; v.y = 0.5f * sq;
        *reinterpret_cast<float2*>(a.partial + 2 * wave_global) = v;
    }
}

// ------------------------------------------------------- dense kernel update ---
// GMF's Dense(1, use_bias=False) kernel w[D]: gradient = sum of the per-wave
// partials + l2w * w (gmf.py:31-32), then the optimizer's dense rule.
// stage 1: block g sums the wave partials r = g, g + gridDim.x, ... -> out[g][D]  (a single block over
// the 16 384 partials of a 65 536-sample batch took 250 us)
__global__ __launch_bounds__(256) void dense_reduce1_kernel(const float* wpartial, int nwaves, int D, float* out) {
    __shared__ float sh[256];
    for (int e0 = 0; e0 < D; e0 += 64) {
        const int e = e0 + (threadIdx.x & 63);
        const int part = threadIdx.x >> 6;          // 4 row slices
        float s = 0.0f;
        if (e < D) for (int r = blockIdx.x + part * gridDim.x; r < nwaves; r += 4 * gridDim.x) s += wpartial[(size_t)r * D + e];
        sh[threadIdx.x] = s;
        __syncthreads();
        if (part == 0 && e < D) out[(size_t)blockIdx.x * D + e] = sh[threadIdx.x] + sh[64 + threadIdx.x] + sh[128 + threadIdx.x] + sh[192 + threadIdx.x];
        __syncthreads();
    }
}

// final stage: gradient = sum of the partial rows + l2w * w; 0.5*||w||^2 of the PRE-step kernel joins
// l2_loss (gmf.py:31-32); with optkind >= 0 the optimizer's dense rule is applied here as well (SGD /
// Adagrad), otherwise the gradient is left in gout (Adam: adam_sweep_kernel follows).
__global__ __launch_bounds__(1024) void dense_reduce_kernel(const float* wpartial, int nwaves, int D, float* w,
                                                            float l2w, float* gout, float* l2slot,
                                                            float* acc, int optkind, float lr, float eps) {
    // blockDim = 1024 threads; column e is summed by threads e, e + 64, ... (16 row slices)
    __shared__ float sh[1024];
    float wsq = 0.0f;
    for (int e0 = 0; e0 < D; e0 += 64) {
        const int e = e0 + (threadIdx.x & 63);
        const int part = threadIdx.x >> 6;
        float s = 0.0f;
        if (e < D) for (int r = part; r < nwaves; r += 16) s += wpartial[(size_t)r * D + e];
        sh[threadIdx.x] = s;
        __syncthreads();
        if (part == 0 && e < D) {
            float t = 0.0f;
            for (int k = 0; k < 16; ++k) t += sh[k * 64 + (threadIdx.x & 63)];
            const float we = w[e];
            wsq += we * we;
            const float g = t + l2w * we;
            if (optkind == ORX_SGD) {
                w[e] = we - lr * g;
            } else if (optkind == ORX_ADAGRAD) {
                const float a2 = acc[e] + g * g;
                acc[e] = a2;
                w[e] = we - lr * g / (sqrtf(a2) + eps);
            } else if (gout != nullptr) {
                gout[e] = g;
            }
        }
        __syncthreads();
    }
    if (l2slot != nullptr && threadIdx.x < 64) {
        for (int off = 32; off > 0; off >>= 1) wsq += __shfl_xor(wsq, off);
        if (threadIdx.x == 0) { l2slot[0] = 0.0f; l2slot[1] = 0.5f * wsq; }
    }
}

__global__ __launch_bounds__(256) void dense_apply_kernel(float* w, float* acc, float* g, int n, int optkind, float lr, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    g[i] = 0.0f;
    if (optkind == ORX_ADAGRAD) {
        const float a2 = acc[i] + gi * gi;
        acc[i] = a2;
        w[i] = w[i] - lr * gi / (sqrtf(a2) + eps);
    } else {
        w[i] = w[i] - lr * gi;
    }
}

// ------------------------------------------------------------- all-item scorer ---
// out[q, j] = score(U[uid[q]], V[j]) + b[j]; block = 64 items x 16 users, the 16
// user rows staged in LDS, every thread walks its own item row.
// kind 0: dot, 1: -squared L2 distance, 2: GMF weighted dot
__global__ __launch_bounds__(256) void score_all_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                        const float* __restrict__ b, const float* __restrict__ w,
                                                        const int32_t* __restrict__ uid, int64_t nq, int64_t NU,
                                                        int64_t NI, int D, int kind, float* __restrict__ out, int* err) {
    extern __shared__ float urow[];                 // [16][D]
    const int64_t q0 = (int64_t)blockIdx.y * 16;
    const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int qs = threadIdx.x >> 6;                // this thread scores users qs, qs+4, qs+8, qs+12
    for (int k = threadIdx.x; k < 16 * D; k += 256) {
        const int64_t q = q0 + k / D;
        float v = 0.0f;
        if (q < nq) {
            const int u = uid[q];
            if ((uint32_t)u >= (uint64_t)NU) *err = 1; else v = U[(size_t)u * D + k % D];
        }
        urow[k] = v;
    }
    __syncthreads();
    if (j >= NI) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* vr = V + (size_t)j * D;
    for (int e = 0; e < D; ++e) {
        const float v = vr[e];
        const float we = kind == 2 ? w[e] : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = urow[(qs + 4 * k) * D + e];
            if (kind == 1) { const float d = x - v; acc[k] -= d * d; }
            else acc[k] += x * v * we;
        }
    }
    const float bj = b[j];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t q = q0 + qs + 4 * k;
        if (q < nq) out[q * NI + j] = acc[k] + bj;
    }
}

// ---------------------------------------------------------------- launchers ---
static inline int lpr_for_dim_p(int D) {
    switch (D) { case 16: return 4; case 32: return 8; case 64: return 16; case 128: return 32; case 256: return 64; default: return 0; }
}

static inline int64_t point_grid(int D, int64_t B) {
    const int lpr = lpr_for_dim_p(D);
    const int64_t tpb = lpr ? 4 * (64 / lpr) : 4;
    int64_t g = (B + tpb - 1) / tpb;
    // ORX_POINT_GRID_MAX (A/B): workgroups that loop over several sample blocks leave fewer partial rows for GMF's Dense(1) gradient
    // (<= 1024: dense_reduce_kernel alone, no dense_reduce1_kernel launch)
    static const int64_t cap = getenv("ORX_POINT_GRID_MAX") ? std::max(64, atoi(getenv("ORX_POINT_GRID_MAX"))) : 16384;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return g;
}

int orx_point_nwaves(int D, int64_t B) { return (int)(point_grid(D, B) * 4); }
// rows of GMF's dense-gradient partials: one per workgroup from the float4 kernels, one per wavefront from the generic one
int orx_point_wparts(int D, int64_t B) { return (int)(point_grid(D, B) * (lpr_for_dim_p(D) ? 1 : 4)); }

// does the float4 kernel of this dim reduce and apply GMF's Dense(1) gradient itself (dense_tail)?  ORX_POINT_NO_WTAIL=1: the two
// reduce launches of rounds 2-5 (A/B)
// reducer workgroups behind the sample workgroups of a GMF launch: one per 64 partial rows + the one that applies the rule
int orx_point_reducers(int D, int64_t B) { return (int)((point_grid(D, B) + WT_GROUP - 1) / WT_GROUP) + 1; }
bool orx_point_dense_tail_ok(int D) {
    return lpr_for_dim_p(D) != 0 && getenv("ORX_POINT_NO_WTAIL") == nullptr;
}

template <int LPR, int MODEL, int OPT>
static void launch_point_mode(int mode, dim3 g, orx_ctx* c, const PointArgs& a) {
    switch (mode) {
        case MODE_EXACT:
            if (OPT == ORX_SGD && LPR < 64 && a.ids4 != nullptr) ORX_LAUNCH(c, (point_fused_kernel<LPR, MODEL, OPT, MODE_EXACT, true>), g, dim3(256), 0, a);
            else ORX_LAUNCH(c, (point_fused_kernel<LPR, MODEL, OPT, MODE_EXACT>), g, dim3(256), 0, a);
            break;
        case MODE_HOGWILD: ORX_LAUNCH(c, (point_fused_kernel<LPR, MODEL, OPT, MODE_HOGWILD>), g, dim3(256), 0, a); break;
        case MODE_ACCUM: ORX_LAUNCH(c, (point_fused_kernel<LPR, MODEL, ORX_SGD, MODE_ACCUM>), g, dim3(256), 0, a); break;
        default: ORX_LAUNCH(c, (point_fused_kernel<LPR, MODEL, ORX_SGD, MODE_LOSS>), g, dim3(256), 0, a); break;
    }
}

template <int MODEL, int OPT>
static void launch_point_generic(int mode, dim3 g, orx_ctx* c, const PointArgs& a) {
    switch (mode) {
        case MODE_EXACT: ORX_LAUNCH(c, (point_generic_kernel<MODEL, OPT, MODE_EXACT>), g, dim3(256), 0, a); break;
        case MODE_HOGWILD: ORX_LAUNCH(c, (point_generic_kernel<MODEL, OPT, MODE_HOGWILD>), g, dim3(256), 0, a); break;
        case MODE_ACCUM: ORX_LAUNCH(c, (point_generic_kernel<MODEL, ORX_SGD, MODE_ACCUM>), g, dim3(256), 0, a); break;
        default: ORX_LAUNCH(c, (point_generic_kernel<MODEL, ORX_SGD, MODE_LOSS>), g, dim3(256), 0, a); break;
    }
}

template <int MODEL, int OPT>
static void launch_point_lpr(int lpr, int mode, dim3 g, orx_ctx* c, const PointArgs& a) {
    switch (lpr) {
        case 4: launch_point_mode<4, MODEL, OPT>(mode, g, c, a); break;
        case 8: launch_point_mode<8, MODEL, OPT>(mode, g, c, a); break;
        case 16: launch_point_mode<16, MODEL, OPT>(mode, g, c, a); break;
        case 32: launch_point_mode<32, MODEL, OPT>(mode, g, c, a); break;
        case 64: launch_point_mode<64, MODEL, OPT>(mode, g, c, a); break;
        default: launch_point_generic<MODEL, OPT>(mode, g, c, a); break;
    }
}

// lazy Adam: exact mode on the float4 dims only
template <int MODEL>
static void launch_point_adam(int lpr, dim3 g, orx_ctx* c, const PointArgs& a) {
    switch (lpr) {
        case 4: ORX_LAUNCH(c, (point_fused_kernel<4, MODEL, ORX_ADAM, MODE_EXACT>), g, dim3(256), 0, a); break;
        case 8: ORX_LAUNCH(c, (point_fused_kernel<8, MODEL, ORX_ADAM, MODE_EXACT>), g, dim3(256), 0, a); break;
        case 16: ORX_LAUNCH(c, (point_fused_kernel<16, MODEL, ORX_ADAM, MODE_EXACT>), g, dim3(256), 0, a); break;
        case 32: ORX_LAUNCH(c, (point_fused_kernel<32, MODEL, ORX_ADAM, MODE_EXACT>), g, dim3(256), 0, a); break;
        default: ORX_LAUNCH(c, (point_fused_kernel<64, MODEL, ORX_ADAM, MODE_EXACT>), g, dim3(256), 0, a); break;
    }
}

int orx_launch_point_fused(orx_ctx* ctx, int model, int optkind, int mode, const PointArgs& a) {
    ProfScope ps(ctx, ORX_K_POINT);
    const int lpr = lpr_for_dim_p(a.D);
    ORX_ARG(a.wt_nred == 0 || (model == ORX_GMF && lpr != 0 && mode != MODE_LOSS && a.wt_nred == orx_point_reducers(a.D, a.B)),
            "point_fused: reducer workgroups are for GMF's float4 kernels");
    const dim3 g((unsigned)(point_grid(a.D, a.B) + (mode == MODE_EXACT && lpr != 0 ? a.n_apply_blocks : 0) + a.wt_nred));
    if (optkind == ORX_ADAM && mode == MODE_EXACT) {
        ORX_ARG(lpr != 0 && a.lrt != nullptr && a.role_bits, "point_fused: the lazy Adam path needs a float4 dim and the exact-step plan");
        if (model == ORX_GMF) launch_point_adam<ORX_GMF>(lpr, g, ctx, a);
        else launch_point_adam<ORX_WRMF>(lpr, g, ctx, a);
        ORX_HIP(hipGetLastError());
        return ORX_OK;
    }
    const bool ada = optkind == ORX_ADAGRAD;
    if (model == ORX_GMF) {
        if (ada) launch_point_lpr<ORX_GMF, ORX_ADAGRAD>(lpr, mode, g, ctx, a);
        else launch_point_lpr<ORX_GMF, ORX_SGD>(lpr, mode, g, ctx, a);
    } else {
        if (ada) launch_point_lpr<ORX_WRMF, ORX_ADAGRAD>(lpr, mode, g, ctx, a);
        else launch_point_lpr<ORX_WRMF, ORX_SGD>(lpr, mode, g, ctx, a);
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_dense_reduce(orx_ctx* ctx, const float* wpartial, int nwaves, int D, float* w, float l2w, float* gout,
                            float* l2slot, float* acc, int optkind, float lr, float eps) {
    constexpr int G = 256;      // wpartial has room for G more rows behind the nwaves partials (orx_dense_reduce_rows)
    if (nwaves > 4 * G) {
        float* stage1 = const_cast<float*>(wpartial) + (size_t)nwaves * D;
        ORX_LAUNCH(ctx, dense_reduce1_kernel, dim3(G), dim3(256), 0, wpartial, nwaves, D, stage1);
        wpartial = stage1; nwaves = G;
    }
    ORX_LAUNCH(ctx, dense_reduce_kernel, dim3(1), dim3(1024), 0, wpartial, nwaves, D, w, l2w, gout, l2slot, acc, optkind, lr, eps);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_dense_apply(orx_ctx* ctx, float* w, float* acc, float* g, int n, int optkind, float lr, float eps) {
    ORX_LAUNCH(ctx, dense_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w, acc, g, n, optkind, lr, eps);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_score_all(orx_ctx* ctx, const float* U, const float* V, const float* b, const float* w,
                         const int32_t* uid, int64_t nq, int64_t NU, int64_t NI, int D, int kind, float* out) {
    ProfScope ps(ctx, ORX_K_GEMM);
    if (getenv("ORX_SCORE_SIMPLE") == nullptr) {         // the matrix-core scorer (kernels_score.hip); this kernel remains for
        bool launched = false;                           // dims whose tiles do not fit the LDS and as the A/B reference
        const int rc = orx_launch_score_mfma(ctx, U, V, b, w, uid, nq, NU, NI, D, kind, out, &launched);
        if (rc != ORX_OK || launched) return rc;
    }
    const dim3 g((unsigned)((NI + 63) / 64), (unsigned)((nq + 15) / 16));
    ORX_LAUNCH(ctx, score_all_kernel, g, dim3(256), (size_t)16 * D * sizeof(float), U, V, b, w, uid, nq, NU, NI, D, kind, out, ctx->d_err);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
