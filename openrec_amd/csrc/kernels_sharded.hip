// Building blocks of the row-sharded multi-GPU step (one process per GPU; the
// exchange between them is RCCL all-to-all driven from openrec_amd/sharded.py).
// After the exchange every rank holds, per live triplet slot, the user row
// (local shard) and the two item rows (received), all as dense [T, stride]
// buffers with the bias at column D:
//   pair_grads : score + loss + per-occurrence gradients (same math as the
//                fused single-GPU kernel, kernels_pairwise.hip)
//   apply_rows : optimizer sparse apply of per-occurrence gradient rows onto the
//                local shard.  Reads (gather) and writes (apply) are separate
//                phases here, so duplicates are no hazard for SGD: every
//                occurrence is accumulated with fp32 atomics (TF scatter_add).
//                Adagrad sums duplicates first (dedup flags + gsum + dup_apply).
#include "orx_device.h"

// ------------------------------------------------------------- pair_grads ---
template <int LPR, int MODEL>
__global__ __launch_bounds__(256) void pair_grads_kernel(GradArgs a) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4 * TPW;
    float loss_acc = 0.0f, sq_acc = 0.0f;
    for (int64_t t = wave_global * TPW + grp; t < a.T; t += stride) {
        if (a.valid != nullptr && a.valid[t] < 0) continue;
        const f4 ru = *reinterpret_cast<const f4*>(a.u + t * a.row_stride + 4 * sub);
        const f4 rp = *reinterpret_cast<const f4*>(a.p + t * a.row_stride + 4 * sub);
        const f4 rn = *reinterpret_cast<const f4*>(a.n + t * a.row_stride + 4 * sub);
        const float bp = a.p[t * a.row_stride + D], bn = a.n[t * a.row_stride + D];
        const float red = group_allreduce<LPR>(score_partial<MODEL>(ru, rp, rn));
        float term, g;
        score<MODEL>(red, bp, bn, a.invB, a.margin, term, g);
        sq_acc += dot4(ru, ru) + dot4(rp, rp) + dot4(rn, rn);
        if (sub == 0) loss_acc += term;
        f4 gu, gp, gn; float gbp, gbn;
        row_grads<MODEL>(ru, rp, rn, g, a.l2w, gu, gp, gn, gbp, gbn);
        *reinterpret_cast<f4*>(a.gu + t * a.g_stride + 4 * sub) = gu;
        *reinterpret_cast<f4*>(a.gp + t * a.g_stride + 4 * sub) = gp;
        *reinterpret_cast<f4*>(a.gn + t * a.g_stride + 4 * sub) = gn;
        if (sub == 0) { a.gp[t * a.g_stride + D] = gbp; a.gn[t * a.g_stride + D] = gbn; }
    }
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        float2 v; v.x = ls; v.y = 0.5f * sq;
        *reinterpret_cast<float2*>(a.partial + 2 * wave_global) = v;
    }
}

template <int MODEL>
__global__ __launch_bounds__(256) void pair_grads_generic_kernel(GradArgs a) {
    const int lane = threadIdx.x & 63;
    const int D = a.D;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    float loss_acc = 0.0f, sq_acc = 0.0f;
    for (int64_t t = wave_global; t < a.T; t += stride) {
        if (a.valid != nullptr && a.valid[t] < 0) continue;
        const float* u = a.u + t * a.row_stride;
        const float* p = a.p + t * a.row_stride;
        const float* n = a.n + t * a.row_stride;
        float part = 0.0f;
        for (int e = lane; e < D; e += 64) {
            const float x = u[e], y = p[e], z = n[e];
            if (MODEL == ORX_BPR) part += x * (y - z);
            else part += (x - z) * (x - z) - (x - y) * (x - y);
            sq_acc += x * x + y * y + z * z;
        }
        const float red = wave_sum(part);
        float term, g;
        score<MODEL>(red, p[D], n[D], a.invB, a.margin, term, g);
        if (lane == 0) loss_acc += term;
        for (int e = lane; e < D; e += 64) {
            const float x = u[e], y = p[e], z = n[e];
            float gu, gp, gn;
            if (MODEL == ORX_BPR) {
                gu = g * (y - z) + a.l2w * x; gp = g * x + a.l2w * y; gn = -g * x + a.l2w * z;
            } else {
                const float a2 = 2.0f * g;
                gu = -a2 * (y - z) + a.l2w * x; gp = -a2 * (x - y) + a.l2w * y; gn = a2 * (x - z) + a.l2w * z;
            }
            a.gu[t * a.g_stride + e] = gu; a.gp[t * a.g_stride + e] = gp; a.gn[t * a.g_stride + e] = gn;
        }
        if (lane == 0) {
            const float gbp = MODEL == ORX_BPR ? g : -g;
            a.gp[t * a.g_stride + D] = gbp; a.gn[t * a.g_stride + D] = -gbp;
        }
    }
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        float2 v; v.x = ls; v.y = 0.5f * sq;
        *reinterpret_cast<float2*>(a.partial + 2 * wave_global) = v;
    }
}

static inline int lpr_for(int D, int64_t s1, int64_t s2) {
    if (s1 % 4 || s2 % 4) return 0;
    switch (D) { case 16: return 4; case 32: return 8; case 64: return 16; case 128: return 32; case 256: return 64; default: return 0; }
}

static inline unsigned grid_for_rows(int lpr, int64_t n) {
    const int64_t per = lpr ? 4 * (64 / lpr) : 4;
    int64_t g = (n + per - 1) / per;
    if (g > 65536) g = 65536;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int orx_launch_pair_grads(orx_ctx* ctx, int model, const GradArgs& a, int* nwaves) {
    ProfScope ps(ctx, ORX_K_FUSED);
    const int lpr = lpr_for(a.D, a.row_stride, a.g_stride);
    const dim3 g(grid_for_rows(lpr, a.T));
    if (nwaves) *nwaves = (int)g.x * 4;
#define PG(L) (model == ORX_BPR ? (void)ORX_LAUNCH(ctx, (pair_grads_kernel<L, ORX_BPR>), g, dim3(256), 0, a) \
                                : (void)ORX_LAUNCH(ctx, (pair_grads_kernel<L, ORX_UCML>), g, dim3(256), 0, a))
    switch (lpr) {
        case 4: PG(4); break;
        case 8: PG(8); break;
        case 16: PG(16); break;
        case 32: PG(32); break;
        case 64: PG(64); break;
        default:
            if (model == ORX_BPR) ORX_LAUNCH(ctx, (pair_grads_generic_kernel<ORX_BPR>), g, dim3(256), 0, a);
            else ORX_LAUNCH(ctx, (pair_grads_generic_kernel<ORX_UCML>), g, dim3(256), 0, a);
    }
#undef PG
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// --------------------------------------------------------------- apply_rows ---
// SGD: var.scatter_add(ids, -lr * grad), every occurrence accumulated.
__global__ __launch_bounds__(256) void apply_rows_sgd_kernel(RowsArgs a) {
    const int D = a.D;
    const int64_t total = a.n * (D + 1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t k = i / (D + 1);
        const int e = (int)(i - k * (D + 1));
        const int r = a.ids[k];
        if (r < 0) continue;
        if ((int64_t)r >= a.rows) { *a.err = 1; continue; }
        const float g = a.grads[k * a.g_stride + e];
        if (e < D) unsafeAtomicAdd(a.W + (size_t)r * D + e, -a.lr * g);
        else if (a.bias != nullptr) unsafeAtomicAdd(a.bias + r, -a.lr * g);
    }
}

// The same for D % 64 == 0 without a bias column: one wavefront per gradient row, lane -> columns lane, lane + 64, ... (each
// atomic instruction covers 256 contiguous bytes, no index division), four rows in flight per wavefront.
__global__ __launch_bounds__(256) void apply_rows_sgd_wave_kernel(RowsArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t k0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; k0 < a.n; k0 += nw * 4) {
        int r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = k0 + u < a.n ? a.ids[k0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r[u] < 0) continue;
            if ((int64_t)r[u] >= a.rows) { if (lane == 0) *a.err = 1; r[u] = -1; }
        }
        for (int e = lane; e < a.D; e += 64) {
            float g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = r[u] >= 0 ? a.grads[(k0 + u) * a.g_stride + e] : 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (r[u] >= 0) unsafeAtomicAdd(a.W + (size_t)r[u] * a.D + e, -a.lr * g[u]);
        }
    }
}

// SGD with duplicate flags (computed once for all steps of a planned K-step call): a row referenced
// once in the list is a plain float4 read-modify-write by its lane group, only duplicated rows take
// the 64 atomics per reference.  LPR lanes per reference.
template <int LPR>
__device__ __forceinline__ void apply_rows_sgd_flagged_body(const RowsArgs a, int block, int nblocks) {      // (by value: through a reference the fields of the kernel argument end up in vector registers -- 34 VGPRs against 24, 35 -> 49 us)
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t stride = (int64_t)nblocks * 4 * TPW;
    for (int64_t k = ((int64_t)block * 4 + (threadIdx.x >> 6)) * TPW + grp; k < a.n; k += stride) {
        const int r = a.ids[k];
        if (r < 0) continue;
        if ((int64_t)r >= a.rows) { if (sub == 0) *a.err = 1; continue; }
        const bool dup = a.dflag[k] != 0;
        const f4 g = *reinterpret_cast<const f4*>(a.grads + k * a.g_stride + 4 * sub);
        float* w = a.W + (size_t)r * D + 4 * sub;
        if (dup) {
            atomic_add_f4(w, -a.lr * g);
        } else {
            *reinterpret_cast<f4*>(w) = *reinterpret_cast<const f4*>(w) - a.lr * g;
        }
        if (a.bias != nullptr && sub == 0) {
            const float gb = a.gbias ? a.gbias[k] : a.grads[k * a.g_stride + D];
            if (dup) unsafeAtomicAdd(a.bias + r, -a.lr * gb);
            else a.bias[r] = a.bias[r] - a.lr * gb;
        }
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void apply_rows_sgd_flagged_kernel(RowsArgs a) { apply_rows_sgd_flagged_body<LPR>(a, (int)blockIdx.x, (int)gridDim.x); }

// two lists of two tables in one launch (the sharded step's user rows and item rows: one launch and the gap behind it fewer per step --
// the user list is short work, 7.5 us as a launch of its own): the first blocks_a workgroups take list a, the others list b
template <int LPR>
__global__ __launch_bounds__(256) void apply_rows_sgd_flagged2_kernel(RowsArgs a, RowsArgs b, int blocks_a) {
    if ((int)blockIdx.x < blocks_a) apply_rows_sgd_flagged_body<LPR>(a, (int)blockIdx.x, blocks_a);
    else apply_rows_sgd_flagged_body<LPR>(b, (int)blockIdx.x - blocks_a, (int)gridDim.x - blocks_a);
}

// Planned apply: one lane group per reference.  The row's duplicate status / role comes from the ids rewritten by
// dedup_kernel: referenced once -> the optimizer rule in place (plain float4 read-modify-write); twice -> one plain
// store per reference into the two scratch rows; more -> the reference's private staging slot (or atomics where the
// range made no plan).  dup_apply_kernel (+ hot_reduce_kernel) finishes the duplicated rows.
template <int LPR, int OPT>
__global__ __launch_bounds__(256) void rows_planned_kernel(RowsArgs a) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t stride = (int64_t)gridDim.x * 4 * TPW;
    for (int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * TPW + grp; k < a.n; k += stride) {
        const int r0 = a.ids[k];
        if (r0 < 0) continue;                                                  // padding
        if ((int64_t)r0 >= a.rows) { if (sub == 0) *a.err = 1; continue; }
        const uint32_t v = (uint32_t)a.ids2[k];
        const int dup = v >> 31, role = (v >> 29) & 3;
        const size_t off = (size_t)r0 * D + 4 * sub;
        const f4 g = *reinterpret_cast<const f4*>(a.grads + k * a.g_stride + 4 * sub);
        const float gb = (a.bias != nullptr && sub == 0) ? a.grads[k * a.g_stride + D] : 0.0f;
        if (!dup) {
            opt_apply4<OPT>(a.W + off, a.A + off, *reinterpret_cast<const f4*>(a.W + off), g, a.lr, a.eps);
            if (a.bias != nullptr && sub == 0) opt_apply1<OPT>(a.bias + r0, a.ab + r0, a.bias[r0], gb, a.lr, a.eps);
        } else {
            int slot = -1;
            if (role == 2 && a.stage != nullptr) { const int2 ri = a.refinfo[k]; slot = ri.x < 0 ? -1 : a.segstart[ri.x] + ri.y; }
            dup_store4s(a.G, a.G2, off, g, role, a.stage, slot, D, sub);
            if (a.bias != nullptr && sub == 0) dup_store1s(a.gb, a.gb2, r0, gb, role, a.stageb, slot);
        }
    }
}

template <int OPT>
static void launch_rows_planned(orx_ctx* ctx, int lpr, const RowsArgs& a) {
    const dim3 g(grid_for_rows(lpr, a.n));
    switch (lpr) {
        case 4: ORX_LAUNCH(ctx, (rows_planned_kernel<4, OPT>), g, dim3(256), 0, a); break;
        case 8: ORX_LAUNCH(ctx, (rows_planned_kernel<8, OPT>), g, dim3(256), 0, a); break;
        case 16: ORX_LAUNCH(ctx, (rows_planned_kernel<16, OPT>), g, dim3(256), 0, a); break;
        case 32: ORX_LAUNCH(ctx, (rows_planned_kernel<32, OPT>), g, dim3(256), 0, a); break;
        default: ORX_LAUNCH(ctx, (rows_planned_kernel<64, OPT>), g, dim3(256), 0, a); break;
    }
}

int orx_launch_rows_planned(orx_ctx* ctx, int optkind, const RowsArgs& a) {
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    int lpr = 0;
    switch (a.D) { case 16: lpr = 4; break; case 32: lpr = 8; break; case 64: lpr = 16; break; case 128: lpr = 32; break; case 256: lpr = 64; break; }
    ORX_ARG(lpr != 0 && a.g_stride % 4 == 0, "rows_planned: dim must be 16/32/64/128/256 and the gradient rows 16-byte aligned");
    if (optkind == ORX_ADAGRAD) launch_rows_planned<ORX_ADAGRAD>(ctx, lpr, a);
    else launch_rows_planned<ORX_SGD>(ctx, lpr, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// Adagrad: rows referenced once are updated in place, duplicated rows sum into
// gsum first (dup_apply_kernel finishes them).  One wavefront per reference.
__global__ __launch_bounds__(256) void apply_rows_adagrad_kernel(RowsArgs a) {
    const int lane = threadIdx.x & 63;
    const int D = a.D;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < a.n; k += stride) {
        const int r = a.ids[k];
        if (r < 0) continue;
        if ((int64_t)r >= a.rows) { if (lane == 0) *a.err = 1; continue; }
        const bool dup = a.dflag[k] != 0;
        const float* g = a.grads + k * a.g_stride;
        for (int e = lane; e < D; e += 64) {
            const size_t i = (size_t)r * D + e;
            if (dup) unsafeAtomicAdd(a.G + i, g[e]);
            else opt_apply1<ORX_ADAGRAD>(a.W + i, a.A + i, a.W[i], g[e], a.lr, a.eps);
        }
        if (a.bias != nullptr && lane == 0) {
            if (dup) unsafeAtomicAdd(a.gb + r, g[D]);
            else opt_apply1<ORX_ADAGRAD>(a.bias + r, a.ab + r, a.bias[r], g[D], a.lr, a.eps);
        }
    }
}

// TF-2.0 Adam on gradient rows, applied lazily: a row replays its gradient-free steps when it is
// next referenced.  One wavefront per reference (STEP) or per row to bring up to date (touch, !STEP); lane e
// owns elements e, e+64, ...  `T` is the step being taken (STEP: rows are replayed to T-1, then step T with the
// summed gradient) or the step to replay to (touch).  Rows referenced once are finished here, duplicated rows
// (dflag) by adam_rows_dup_kernel over the dedup list: STEP sums their gradients into gsum first.
template <bool STEP>
__global__ __launch_bounds__(256) void adam_rows_kernel(AdamRowsArgs a) {
    const int lane = threadIdx.x & 63;
    const int D = a.D;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < a.n; k += stride) {
        const int r = a.ids[k];
        if (r < 0) continue;
        if ((int64_t)r >= a.rows) { if (lane == 0) *a.err = 1; continue; }
        const bool dup = a.dflag[k] != 0;
        if (dup) {
            if (STEP) {
                const float* g = a.grads + k * a.g_stride;
                for (int e = lane; e < D; e += 64) unsafeAtomicAdd(a.G + (size_t)r * D + e, g[e]);
            }
            continue;
        }
        const int from = __builtin_amdgcn_readfirstlane(a.last[r]);
        if (!STEP && from >= a.T) continue;
        for (int e = lane; e < D; e += 64) {
            const size_t i = (size_t)r * D + e;
            float w = a.W[i], m = a.M[i], v = a.V[i];
            adam_replay1<true>(w, m, v, from, STEP ? a.T - 1 : a.T, a.lrt, a.b1, a.b2, a.eps, a.newton != 0, a.cf);
            if (STEP) adam_elem(w, m, v, a.grads[k * a.g_stride + e], a.lr_T, a.b1, a.b2, a.eps);
            a.W[i] = w; a.M[i] = m; a.V[i] = v;
        }
        if (lane == 0) a.last[r] = a.T;
    }
}

template <bool STEP>
__global__ __launch_bounds__(256) void adam_rows_dup_kernel(AdamRowsArgs a) {
    const int lane = threadIdx.x & 63;
    const int D = a.D;
    const int n = *a.dcount;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < n; k += stride) {
        const size_t r = a.dlist[k] & 0x7fffffffu;
        const int from = __builtin_amdgcn_readfirstlane(a.last[r]);
        if (!STEP && from >= a.T) continue;
        for (int e = lane; e < D; e += 64) {
            const size_t i = r * D + e;
            float w = a.W[i], m = a.M[i], v = a.V[i];
            adam_replay1<true>(w, m, v, from, STEP ? a.T - 1 : a.T, a.lrt, a.b1, a.b2, a.eps, a.newton != 0, a.cf);
            if (STEP) { adam_elem(w, m, v, a.G[i], a.lr_T, a.b1, a.b2, a.eps); a.G[i] = 0.0f; }
            a.W[i] = w; a.M[i] = m; a.V[i] = v;
        }
        if (lane == 0) a.last[r] = a.T;
    }
}

int orx_launch_adam_rows(orx_ctx* ctx, bool step, const AdamRowsArgs& a, int64_t max_dups) {
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    if (a.n == 0) return ORX_OK;
    const dim3 g(grid_for_rows(0, a.n)), gd(grid_for_rows(0, max_dups > 0 ? max_dups : 1));
    if (step) {
        ORX_LAUNCH(ctx, adam_rows_kernel<true>, g, dim3(256), 0, a);
        ORX_LAUNCH(ctx, adam_rows_dup_kernel<true>, gd, dim3(256), 0, a);
    } else {
        ORX_LAUNCH(ctx, adam_rows_kernel<false>, g, dim3(256), 0, a);
        ORX_LAUNCH(ctx, adam_rows_dup_kernel<false>, gd, dim3(256), 0, a);
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_apply_rows(orx_ctx* ctx, int optkind, bool use_dflag, const RowsArgs& a) {
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    if (a.n == 0) return ORX_OK;
    int lpr = 0;
    switch (a.D) { case 16: lpr = 4; break; case 32: lpr = 8; break; case 64: lpr = 16; break; case 128: lpr = 32; break; case 256: lpr = 64; break; }
    if (optkind == ORX_SGD && use_dflag && lpr != 0 && a.g_stride % 4 == 0) {
        const dim3 g(grid_for_rows(lpr, a.n));
        switch (lpr) {
            case 4: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged_kernel<4>), g, dim3(256), 0, a); break;
            case 8: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged_kernel<8>), g, dim3(256), 0, a); break;
            case 16: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged_kernel<16>), g, dim3(256), 0, a); break;
            case 32: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged_kernel<32>), g, dim3(256), 0, a); break;
            default: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged_kernel<64>), g, dim3(256), 0, a); break;
        }
    } else if (optkind == ORX_SGD && a.bias == nullptr && a.D % 64 == 0 && getenv("ORX_APPLY_SCALAR") == nullptr) {
        int64_t g = (a.n + 15) / 16;
        if (g > 16384) g = 16384;
        ORX_LAUNCH(ctx, apply_rows_sgd_wave_kernel, dim3((unsigned)g), dim3(256), 0, a);
    } else if (optkind == ORX_SGD) {
        int64_t g = (a.n * (a.D + 1) + 255) / 256;
        if (g > 65536) g = 65536;
        ORX_LAUNCH(ctx, apply_rows_sgd_kernel, dim3((unsigned)g), dim3(256), 0, a);
    } else {
        (void)use_dflag;
        ORX_LAUNCH(ctx, apply_rows_adagrad_kernel, dim3(grid_for_rows(0, a.n)), dim3(256), 0, a);
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// SGD with duplicate flags on two lists (same dim) in one launch; false: not applicable (the caller launches them one by one)
bool orx_launch_apply_rows_pair(orx_ctx* ctx, const RowsArgs& a, const RowsArgs& b, int* rc) {
    int lpr = 0;
    switch (a.D) { case 16: lpr = 4; break; case 32: lpr = 8; break; case 64: lpr = 16; break; case 128: lpr = 32; break; case 256: lpr = 64; break; }
    static const bool off = getenv("ORX_APPLY_NO_PAIR") != nullptr;
    if (off || lpr == 0 || a.D != b.D || a.n == 0 || b.n == 0 || a.g_stride % 4 != 0 || b.g_stride % 4 != 0) return false;
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    const unsigned ga = grid_for_rows(lpr, a.n), gb = grid_for_rows(lpr, b.n);
    const dim3 g(ga + gb);
    switch (lpr) {
        case 4: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged2_kernel<4>), g, dim3(256), 0, a, b, (int)ga); break;
        case 8: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged2_kernel<8>), g, dim3(256), 0, a, b, (int)ga); break;
        case 16: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged2_kernel<16>), g, dim3(256), 0, a, b, (int)ga); break;
        case 32: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged2_kernel<32>), g, dim3(256), 0, a, b, (int)ga); break;
        default: ORX_LAUNCH(ctx, (apply_rows_sgd_flagged2_kernel<64>), g, dim3(256), 0, a, b, (int)ga); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orx_set_error("apply_rows_sgd_flagged2_kernel launch failed: %s", hipGetErrorString(e)); *rc = ORX_ERR_HIP; } else *rc = ORX_OK;
    return true;
}

// ---------------------------------------------------------- loss accumulate ---
__global__ __launch_bounds__(256) void loss_accumulate_kernel(const float* partial, int64_t nwaves, double* accum) {
    __shared__ double sh[2][4];
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwaves; i += (int64_t)gridDim.x * blockDim.x) {
        const float2 v = *reinterpret_cast<const float2*>(partial + 2 * i);
        s0 += (double)v.x; s1 += (double)v.y;
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s0; sh[1][w] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {     // a few blocks, one fp64 atomic pair each (a single block over 16 k partials took 18 us)
        atomicAdd(accum + 0, sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(accum + 1, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

int orx_launch_loss_accumulate(orx_ctx* ctx, const float* partial, int64_t nwaves, double* accum) {
    ProfScope ps(ctx, ORX_K_REDUCE);
    const unsigned blocks = nwaves > (1 << 18) ? 128 : (nwaves > 2048 ? 16 : 1);       // (a chunk of steps at once: the library's sharded engine)
    ORX_LAUNCH(ctx, loss_accumulate_kernel, dim3(blocks), dim3(256), 0, partial, nwaves, accum);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// =====================================================================
// Device-side exchange plan (no host synchronization, fixed-capacity buckets)
// =====================================================================
// Bucket assignment: every element picks its destination rank and takes the next
// free slot of that destination's bucket.  Slots are claimed with one atomic per
// (wavefront, destination): ballot + popcount, so the counters see <= world
// atomics per wavefront.  The order inside a bucket is arbitrary (it only
// changes fp32 summation order downstream).
__device__ __forceinline__ int claim_slot(int dest, bool active, int world, int* counters) {
    // Two levels: the wavefronts of a workgroup claim their places inside the WORKGROUP's share with LDS atomics, one lane per
    // destination then claims the share with ONE global atomic.  Same-address global atomics serialize at ~100 ns: with one
    // per wavefront and destination, a step's 1024 wavefronts of a one-rank world queued 200 us on a single counter.
    __shared__ int blk_cnt[64], blk_base[64];          // (world <= 64)
    const int lane = threadIdx.x & 63;
    for (int k = threadIdx.x; k < world; k += blockDim.x) blk_cnt[k] = 0;
    __syncthreads();
    int slot = -1;
    for (int k = 0; k < world; ++k) {
        const unsigned long long m = __ballot(active && dest == k);
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&blk_cnt[k], __popcll(m));
        base = __shfl(base, leader);
        if (active && dest == k) slot = base + __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    for (int k = threadIdx.x; k < world; k += blockDim.x) blk_base[k] = blk_cnt[k] ? atomicAdd(counters + k, blk_cnt[k]) : 0;
    __syncthreads();
    if (active) slot += blk_base[dest];
    __syncthreads();                                     // (the arrays are reused by the next call)
    return slot;
}


// 1. route each triplet to the owner of its user row (rank = uid % world)
__global__ __launch_bounds__(1024) void shard_route_kernel(RouteArgs a) {
    const int64_t step = blockIdx.y;                    // K-step launch: one grid row per step
    a.uid += step * a.id_stride; a.pid += step * a.id_stride; a.nid += step * a.id_stride;
    a.send += step * (int64_t)a.world * a.cap * 3; a.counters += step * a.world;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool active = t < a.B;
    int u = 0, p = 0, n = 0;
    if (active) {
        u = a.uid[t]; p = a.pid[t]; n = a.nid[t];
        if (!(id_ok(u, a.NU) && id_ok(p, a.NI) && id_ok(n, a.NI))) { *a.err = 1; active = false; }
    }
    const int dest = active ? u % a.world : 0;
    const int slot = claim_slot(dest, active, a.world, a.counters);
    if (!active) return;
    if (slot >= a.cap) { *a.overflow = 1; return; }
    int32_t* o = a.send + ((int64_t)dest * a.cap + slot) * 3;
    o[0] = u; o[1] = p; o[2] = n;
}


// 2. request the two item rows of every live triplet from their owners (rank = id % world)
__global__ __launch_bounds__(1024) void shard_request_kernel(RequestArgs a) {
    const int64_t step = blockIdx.y;                    // K-step launch: one grid row per step
    const int nd = a.world + (a.hot ? 1 : 0);           // destinations: the ranks, and the local replica of the hot items
    a.trip += step * a.T * 3; a.send_ids += step * (int64_t)a.world * a.cap; a.slot += step * 2 * a.T;
    a.u_loc += step * a.T; a.counters += step * nd;
    if (a.hot) a.hot_ids += step * (int64_t)a.cap_hot;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = t < a.T;
    int u = -1, p = 0, n = 0;
    if (inb) { u = a.trip[3 * t]; p = a.trip[3 * t + 1]; n = a.trip[3 * t + 2]; }
    const bool live = inb && u >= 0;
    const bool hp = live && p < a.hot, hn = live && n < a.hot;
    const int sp = claim_slot(live ? (hp ? a.world : p % a.world) : 0, live, nd, a.counters);
    const int sn = claim_slot(live ? (hn ? a.world : n % a.world) : 0, live, nd, a.counters);
    if (!inb) return;
    int gp = -1, gn = -1;
    if (live) {
        if (hp) { if (sp < a.cap_hot) { gp = a.world * a.cap + sp; a.hot_ids[sp] = p; } else *a.overflow = 1; }
        else if (sp < a.cap) { gp = (p % a.world) * a.cap + sp; a.send_ids[gp] = p; } else *a.overflow = 1;
        if (hn) { if (sn < a.cap_hot) { gn = a.world * a.cap + sn; a.hot_ids[sn] = n; } else *a.overflow = 1; }
        else if (sn < a.cap) { gn = (n % a.world) * a.cap + sn; a.send_ids[gn] = n; } else *a.overflow = 1;
    }
    a.slot[t] = gp; a.slot[a.T + t] = gn;
    a.u_loc[t] = (live && gp >= 0 && gn >= 0) ? u / a.world : -1;
}

// 2'. the same request plan with PER-DESTINATION DEDUP: an item asked for by several references of this rank's step travels
// once (its row comes back once, the references' gradients are added into one slot before they leave).  Deterministic: the
// references are sorted by (owner, local row) -- orx_rows_sort, stable in the reference index -- and the distinct keys of an owner
// fill its bucket in ascending order.
__global__ __launch_bounds__(256) void shard_keys_kernel(DedupReqArgs a) {
    const int64_t k = blockIdx.y, r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * a.T) return;
    const int64_t t = r < a.T ? r : r - a.T;
    const int32_t* tr = a.trip + (k * a.T + t) * 3;
    const int u = tr[0], id = tr[r < a.T ? 1 : 2];
    a.keys[k * 2 * a.T + r] = u >= 0 ? (int32_t)((int64_t)(id % a.world) * a.Lr + id / a.world) : -1;
}

// Distinct keys of a sorted list, in parallel over chunks of 1024 entries (grid (chunks, lists)):
//   heads    : number of segment heads per chunk                       -> chunkcnt[k][c]
//   scan     : exclusive prefix of the chunk counts, one wave per list (in place)
//   uniq     : uq[i] = index of entry i's key among the list's distinct keys; ostart[k][o] = that index at owner o's first key
//   slots    : slot / dupref / send_ids of every reference; the segments of shared slots (slot, first entry) -> seglist
//   live     : a triplet lives if both of its requests found a slot
__device__ __forceinline__ bool dd_head(const uint2* sorted, int64_t i, int64_t n, uint32_t sentinel, uint32_t* key_out, uint32_t* prev_out) {
    const uint32_t key = i < n ? sorted[i].x : 0xffffffffu;
    const uint32_t prev = (i > 0 && i < n) ? sorted[i - 1].x : 0xffffffffu;
    *key_out = key; *prev_out = prev;
    return i < n && key < sentinel && (i == 0 || key != prev);
}

__global__ __launch_bounds__(1024) void shard_dd_heads_kernel(DedupReqArgs a) {
    __shared__ int wsum[16];
    const int64_t k = blockIdx.y, n = 2 * a.T, i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t key, prev;
    const bool head = dd_head(a.sorted + k * n, i, n, (uint32_t)((int64_t)a.world * a.Lr), &key, &prev);
    const int c = __popcll(__ballot(head));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; a.chunkcnt[k * a.nchunk + blockIdx.x] = t; }
}

__global__ __launch_bounds__(64) void shard_dd_scan_kernel(DedupReqArgs a) {
    int* p = a.chunkcnt + (int64_t)blockIdx.x * a.nchunk;
    const int lane = threadIdx.x;
    int carry = 0;
    for (int b0 = 0; b0 < a.nchunk; b0 += 64) {
        const int c = b0 + lane < a.nchunk ? p[b0 + lane] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (b0 + lane < a.nchunk) p[b0 + lane] = carry + incl - c;
        carry += __shfl(incl, 63);
    }
    if (lane == 0) a.segcount[blockIdx.x] = 0;
}

__global__ __launch_bounds__(1024) void shard_dd_uniq_kernel(DedupReqArgs a) {
    __shared__ int wsum[16];
    const int64_t k = blockIdx.y, n = 2 * a.T, i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t sentinel = (uint32_t)((int64_t)a.world * a.Lr);
    uint32_t key, prev;
    const bool head = dd_head(a.sorted + k * n, i, n, sentinel, &key, &prev);
    const unsigned long long m = __ballot(head);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int before = a.chunkcnt[k * a.nchunk + blockIdx.x];
    for (int w = 0; w < wave; ++w) before += wsum[w];
    const int u = before + __popcll(m & ((2ull << lane) - 1ull)) - 1;       // heads up to and including this entry, minus one
    if (i < n && key < sentinel) a.uq[k * n + i] = u;
    if (head && (i == 0 || prev / (uint32_t)a.Lr != key / (uint32_t)a.Lr)) a.ostart[k * 64 + key / (uint32_t)a.Lr] = u;
}

__global__ __launch_bounds__(256) void shard_dd_slots_kernel(DedupReqArgs a) {
    const int64_t k = blockIdx.y, n = 2 * a.T, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint2* sorted = a.sorted + k * n;
    int32_t* slot = a.slot + k * n;
    const uint32_t sentinel = (uint32_t)((int64_t)a.world * a.Lr);
    const uint2 e = i < n ? sorted[i] : make_uint2(0xffffffffu, 0u);
    const bool live = i < n && e.x < sentinel;
    if (i < n && !live) slot[e.y] = -1;
    const uint32_t prev = (live && i > 0) ? sorted[i - 1].x : 0xffffffffu, next = (live && i + 1 < n) ? sorted[i + 1].x : 0xffffffffu;
    const int o = live ? (int)(e.x / (uint32_t)a.Lr) : 0;
    const int j = live ? a.uq[k * n + i] - a.ostart[k * 64 + o] : 0;
    const bool shared = live && (e.x == prev || e.x == next);
    const int s_ = (live && j < a.cap) ? o * a.cap + j : -1;
    if (live) {
        // ONE scattered store per reference (the un-sorting is what this kernel costs): the shared flag rides in bit 30 of the
        // slot until shard_dd_live_kernel, reading by reference index, splits it off into dupref
        slot[e.y] = s_ < 0 ? (shared ? -2 : -1) : (s_ | (shared ? 0x40000000 : 0));
        if (s_ < 0) *a.overflow = 1;
        else if (e.x != prev) a.send_ids[k * (int64_t)a.world * a.cap + s_] = (int32_t)((e.x % (uint32_t)a.Lr) * (uint32_t)a.world + (uint32_t)o);
    }
    // the shared slots: (slot, first sorted entry) for the segment sums.  One atomic per wavefront (same-address atomics serialize);
    // the ORDER of this list does not reach any sum.
    const bool lead = s_ >= 0 && e.x != prev && shared && a.seglist != nullptr;
    const unsigned long long m = __ballot(lead);
    if (m) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&a.segcount[k], __popcll(m));
        base = __shfl(base, leader);
        if (lead) a.seglist[k * a.T + base + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(s_, (int)i);
    }
}

__global__ __launch_bounds__(256) void shard_dd_live_kernel(DedupReqArgs a) {
    const int64_t k = blockIdx.y, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.T) return;
    const int u = a.trip[(k * a.T + t) * 3];
    int32_t* slot = a.slot + k * 2 * a.T; unsigned char* dupref = a.dupref + k * 2 * a.T;
    const int rp = slot[t], rn = slot[a.T + t];
    const int sp = rp < 0 ? -1 : (rp & 0x3fffffff), sn = rn < 0 ? -1 : (rn & 0x3fffffff);
    slot[t] = sp; slot[a.T + t] = sn;
    dupref[t] = (rp == -2 || (rp >= 0 && (rp & 0x40000000))) ? 1 : 0;
    dupref[a.T + t] = (rn == -2 || (rn >= 0 && (rn & 0x40000000))) ? 1 : 0;
    a.u_loc[k * a.T + t] = (u >= 0 && sp >= 0 && sn >= 0) ? u / a.world : -1;
}

// The gradients of the references that share a slot, summed in the order of the reference index (the sorted list is stable): one
// wavefront per shared slot, rows of DS floats read from the side buffer the gradient kernel left them in (row = reference index).
__global__ __launch_bounds__(256) void shard_segsum_kernel(const int2* seglist, const int* segcount, const uint2* sorted, int64_t n,
                                                           const float* gdup, int DSg, float* send_g, int DS, int D, float* gb_out, int G) {
    // a group of G lanes per shared slot (G = the power of two >= DSg / 4, at most 64), float4 columns of the side-buffer rows;
    // the column that holds the bias gradient (D .. D + 3) goes to gb_out when the bias travels apart from the rows
    const int lane = threadIdx.x & 63, sub = lane % G, grp = lane / G, per_wave = 64 / G;
    const int nseg = *segcount, C4 = DSg / 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t sgi = wave * per_wave + grp; sgi < nseg; sgi += nwaves * per_wave) {
        const int2 sg = seglist[sgi];
        const uint32_t key = sorted[sg.y].x;
        for (int c = sub; c < C4; c += G) {
            f4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
            for (int64_t i = sg.y; i < n && sorted[i].x == key; ++i)
                acc = acc + *reinterpret_cast<const f4*>(gdup + (int64_t)sorted[i].y * DSg + 4 * c);
            if (4 * c < D) *reinterpret_cast<f4*>(send_g + (int64_t)sg.x * DS + 4 * c) = acc;
            else if (gb_out) gb_out[sg.x] = acc.x;
            else *reinterpret_cast<f4*>(send_g + (int64_t)sg.x * DS + 4 * c) = acc;
        }
    }
}

int orx_launch_shard_segsum(orx_ctx* ctx, const int2* seglist, const int* segcount, const uint2* sorted, int64_t n, const float* gdup, int DSg,
                            float* send_g, int DS, int D, float* gb_out) {
    int G = 1;
    while (G < DSg / 4 && G < 64) G *= 2;
    ORX_LAUNCH(ctx, shard_segsum_kernel, dim3(1024), dim3(256), 0, seglist, segcount, sorted, n, gdup, DSg, send_g, DS, D, gb_out, G);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_shard_keys(orx_ctx* ctx, const DedupReqArgs& a, int64_t K) {
    if (a.T == 0 || K == 0) return ORX_OK;
    ORX_LAUNCH(ctx, shard_keys_kernel, dim3((unsigned)((2 * a.T + 255) / 256), (unsigned)K), dim3(256), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_shard_dedup_slots(orx_ctx* ctx, const DedupReqArgs& a, int64_t K) {
    if (a.T == 0 || K == 0) return ORX_OK;
    const int64_t n = 2 * a.T;
    ORX_LAUNCH(ctx, shard_dd_heads_kernel, dim3((unsigned)a.nchunk, (unsigned)K), dim3(1024), 0, a);
    ORX_LAUNCH(ctx, shard_dd_scan_kernel, dim3((unsigned)K), dim3(64), 0, a);
    ORX_LAUNCH(ctx, shard_dd_uniq_kernel, dim3((unsigned)a.nchunk, (unsigned)K), dim3(1024), 0, a);
    ORX_LAUNCH(ctx, shard_dd_slots_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)K), dim3(256), 0, a);
    ORX_LAUNCH(ctx, shard_dd_live_kernel, dim3((unsigned)((a.T + 255) / 256), (unsigned)K), dim3(256), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// local row ids of a received id list: id / world (or -1)
__global__ void shard_localize_kernel(const int32_t* ids, int64_t n, int world, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = ids[i]; out[i] = v >= 0 ? v / world : -1; }
}


// 4. per live triplet: user row straight from the local shard, item rows from the receive
// buffer; gradients of the item rows go straight into the send buffer of the return trip.
template <int LPR, int MODEL, bool APPLY = false>
__global__ __launch_bounds__(256) void shard_grads_kernel(ShardGradArgs a) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4 * TPW;
    float loss_acc = 0.0f, sq_acc = 0.0f;
    for (int64_t t = wave_global * TPW + grp; t < a.T; t += stride) {
        const int sp = a.slot[t], sn = a.slot[a.T + t];
        const int ul = a.u_loc[t];
        if (ul < 0) {                    // empty slot, or a triplet dropped by a bucket overflow:
            f4 z; z.x = z.y = z.z = z.w = 0.0f;                     // its surviving request gets a zero gradient (in the side buffer if the slot is shared)
            if (sp >= 0) {
                const bool dp = a.dupref && a.dupref[t];
                float* o = dp ? a.gdup + t * a.DSg : a.send_g + (int64_t)sp * a.DS;
                *reinterpret_cast<f4*>(o + 4 * sub) = z;
                if (sub == 0) { if (!dp && a.gb_out) a.gb_out[sp] = 0.f; else o[D] = 0.f; }
            }
            if (sn >= 0) {
                const bool dn = a.dupref && a.dupref[a.T + t];
                float* o = dn ? a.gdup + (a.T + t) * a.DSg : a.send_g + (int64_t)sn * a.DS;
                *reinterpret_cast<f4*>(o + 4 * sub) = z;
                if (sub == 0) { if (!dn && a.gb_out) a.gb_out[sn] = 0.f; else o[D] = 0.f; }
            }
            if (APPLY && sub == 0) a.u_apply[t] = -1;
            continue;
        }
        const f4 ru = *reinterpret_cast<const f4*>(a.U + (size_t)ul * D + 4 * sub);
        const f4 rp = *reinterpret_cast<const f4*>(a.rows_in + (int64_t)sp * a.DS + 4 * sub);
        const f4 rn = *reinterpret_cast<const f4*>(a.rows_in + (int64_t)sn * a.DS + 4 * sub);
        const float bp = a.bias_in ? a.bias_in[sp] : a.rows_in[(int64_t)sp * a.DS + D];
        const float bn = a.bias_in ? a.bias_in[sn] : a.rows_in[(int64_t)sn * a.DS + D];
        const float red = group_allreduce<LPR>(score_partial<MODEL>(ru, rp, rn));
        float term, g;
        score<MODEL>(red, bp, bn, a.invB, a.margin, term, g);
        sq_acc += dot4(ru, ru) + dot4(rp, rp) + dot4(rn, rn);
        if (sub == 0) loss_acc += term;
        f4 gu, gp, gn; float gbp, gbn;
        row_grads<MODEL>(ru, rp, rn, g, a.l2w, gu, gp, gn, gbp, gbn);
        if (APPLY) {
            // SGD: a user row referenced once in this rank's step is updated right here (nobody else reads or writes it);
            // the references of a duplicated row leave their gradients for the flagged apply (atomics), as before
            const bool dup = a.fu[t] != 0;
            if (!dup) *reinterpret_cast<f4*>(a.Uw + (size_t)ul * D + 4 * sub) = ru - a.lr * gu;
            else *reinterpret_cast<f4*>(a.gu + t * D + 4 * sub) = gu;
            if (sub == 0) a.u_apply[t] = dup ? ul : -1;
        } else {
            *reinterpret_cast<f4*>(a.gu + t * D + 4 * sub) = gu;
        }
        // a reference that shares its slot leaves its gradient in the side buffer (row = reference index): shard_segsum_kernel adds
        // the rows of a slot in reference order
        const bool dp = a.dupref && a.dupref[t], dn = a.dupref && a.dupref[a.T + t];
        float* op = dp ? a.gdup + t * a.DSg : a.send_g + (int64_t)sp * a.DS;
        float* on = dn ? a.gdup + (a.T + t) * a.DSg : a.send_g + (int64_t)sn * a.DS;
        *reinterpret_cast<f4*>(op + 4 * sub) = gp;
        *reinterpret_cast<f4*>(on + 4 * sub) = gn;
        if (sub == 0) {
            if (!dp && a.gb_out) a.gb_out[sp] = gbp; else op[D] = gbp;
            if (!dn && a.gb_out) a.gb_out[sn] = gbn; else on[D] = gbn;
        }
    }
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        float2 v; v.x = ls; v.y = 0.5f * sq;
        *reinterpret_cast<float2*>(a.partial + 2 * wave_global) = v;
    }
}

// hot-item replication: the per-item sums of this rank's gradients of the replicated rows were accumulated into the replicas' scratch
// tables (orx_csr_accum on the plan-time sorted slot list: segmented sums in a fixed order, whatever the skew -- the head of a Zipf
// distribution sends thousands of references to ONE row); this packs them as the [H][D + 4] block that is summed over the ranks and
// applied (row, then the bias gradient at column D), and leaves the scratch tables all-zero again
__global__ __launch_bounds__(256) void shard_hot_pack_kernel(float* gV, float* gb, float* hg, int64_t H, int D, int DSh) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < H * DSh; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / DSh; const int c = (int)(i % DSh);
        float v = 0.0f;
        if (c < D) { v = gV[r * D + c]; gV[r * D + c] = 0.0f; }
        else if (c == D) { v = gb[r]; gb[r] = 0.0f; }
        hg[i] = v;
    }
}

int orx_launch_shard_hot_pack(orx_ctx* ctx, float* gV, float* gb, float* hg, int64_t H, int D, int DSh) {
    if (H == 0) return ORX_OK;
    ORX_LAUNCH(ctx, shard_hot_pack_kernel, dim3((unsigned)std::min<int64_t>((H * DSh + 255) / 256, 4096)), dim3(256), 0, gV, gb, hg, H, D, DSh);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_shard_route(orx_ctx* ctx, const RouteArgs& a, int64_t K) {
    if (a.B == 0 || K == 0) return ORX_OK;
    ORX_LAUNCH(ctx, shard_route_kernel, dim3((unsigned)((a.B + 1023) / 1024), (unsigned)K), dim3(1024), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_shard_request(orx_ctx* ctx, const RequestArgs& a, int64_t K) {
    if (a.T == 0 || K == 0) return ORX_OK;
    ORX_LAUNCH(ctx, shard_request_kernel, dim3((unsigned)((a.T + 1023) / 1024), (unsigned)K), dim3(1024), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// generic request plan: every live id (>= 0) claims a slot in the bucket of its owner (id % world)
__global__ __launch_bounds__(1024) void shard_bucket_kernel(const int32_t* ids, int64_t n, int world, int cap, int32_t* send_ids,
                                                           int32_t* slot, int* counters, int* overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = i < n;
    const int id = inb ? ids[i] : -1;
    const bool live = id >= 0;
    const int dest = live ? id % world : 0;
    const int s = claim_slot(dest, live, world, counters);
    if (!inb) return;
    int g = -1;
    if (live) {
        if (s < cap) { g = dest * cap + s; send_ids[g] = id; } else *overflow = 1;
    }
    slot[i] = g;
}

int orx_launch_shard_bucket(orx_ctx* ctx, const int32_t* ids, int64_t n, int world, int cap, int32_t* send_ids, int32_t* slot,
                            int* counters, int* overflow) {
    if (n == 0) return ORX_OK;
    ORX_LAUNCH(ctx, shard_bucket_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, ids, n, world, cap, send_ids, slot, counters, overflow);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_shard_localize(orx_ctx* ctx, const int32_t* ids, int64_t n, int world, int32_t* out) {
    if (n == 0) return ORX_OK;
    ORX_LAUNCH(ctx, shard_localize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ids, n, world, out);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// the loss partials (one pair per wavefront) a gradient launch over T triplets of dimension D writes; 0: no fast path for D
int orx_shard_grads_nwaves(int D, int64_t T) {
    int lpr = 0;
    switch (D) { case 16: lpr = 4; break; case 32: lpr = 8; break; case 64: lpr = 16; break; case 128: lpr = 32; break; case 256: lpr = 64; break; }
    return lpr == 0 ? 0 : (int)grid_for_rows(lpr, T) * 4;
}

int orx_launch_shard_grads(orx_ctx* ctx, int model, const ShardGradArgs& a, int* nwaves) {
    ProfScope ps(ctx, ORX_K_FUSED);
    int lpr = 0;
    switch (a.D) { case 16: lpr = 4; break; case 32: lpr = 8; break; case 64: lpr = 16; break; case 128: lpr = 32; break; case 256: lpr = 64; break; }
    ORX_ARG(lpr != 0 && a.DS % 4 == 0, "sharded fast path: dim must be 16/32/64/128/256 (got %d)", a.D);
    const dim3 g(grid_for_rows(lpr, a.T));
    if (nwaves) *nwaves = (int)g.x * 4;
#define SG(L) do { if (a.fu) { if (model == ORX_BPR) ORX_LAUNCH(ctx, (shard_grads_kernel<L, ORX_BPR, true>), g, dim3(256), 0, a); \
                                 else ORX_LAUNCH(ctx, (shard_grads_kernel<L, ORX_UCML, true>), g, dim3(256), 0, a); } \
                    else if (model == ORX_BPR) ORX_LAUNCH(ctx, (shard_grads_kernel<L, ORX_BPR>), g, dim3(256), 0, a); \
                    else ORX_LAUNCH(ctx, (shard_grads_kernel<L, ORX_UCML>), g, dim3(256), 0, a); } while (0)
    switch (lpr) { case 4: SG(4); break; case 8: SG(8); break; case 16: SG(16); break; case 32: SG(32); break; default: SG(64); break; }
#undef SG
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
