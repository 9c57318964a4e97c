// Ranking metrics of the evaluation step (openrec/tf2/metrics/ranking_metrics.py:8-69,
// driven by eval_step in tf2_examples/bpr_citeulike.py:41-46): per-user AUC, NDCG@k and
// Recall@k from a row of scores over ALL items plus a positive mask and an exclusion mask.
// The reference loops users with tf.map_fn; here one workgroup owns one user, the positives
// are compacted into LDS and the item row is streamed once per chunk of 16 positives.
#include "orx_device.h"

constexpr int EV_CAP = 8192;      // positives per user held in LDS
constexpr int EV_CH = 16;         // positives compared per sweep over the items


__global__ __launch_bounds__(256) void rank_metrics_kernel(EvalArgs a) {
    __shared__ int pos_idx[EV_CAP];
    __shared__ int n_pos_s, n_eval_s;
    __shared__ float red[4][2 * EV_CH];
    __shared__ float acc_ndcg[16], acc_rec[16];
    __shared__ float auc_sum_s;
    const int64_t q = blockIdx.x;
    const float* s = a.pred + q * a.NI;
    const unsigned char* pm = a.pos + q * a.NI;
    const unsigned char* em = a.excl + q * a.NI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { n_pos_s = 0; n_eval_s = 0; auc_sum_s = 0.0f; }
    if (tid < 16) { acc_ndcg[tid] = 0.0f; acc_rec[tid] = 0.0f; }
    __syncthreads();
    int my_eval = 0;
    for (int64_t j = tid; j < a.NI; j += 256) {
        const bool p = pm[j] != 0, e = em[j] != 0;
        if (p) { const int k = atomicAdd(&n_pos_s, 1); if (k < EV_CAP) pos_idx[k] = (int)j; }
        if (!p && !e) my_eval++;                                   // ranking_metrics.py:14
    }
    atomicAdd(&n_eval_s, my_eval);
    __syncthreads();
    const int n_pos = n_pos_s;
    if (n_pos > EV_CAP) { if (tid == 0) *a.err = 2; return; }
    for (int c0 = 0; c0 < n_pos; c0 += EV_CH) {
        float sp[EV_CH], vp[EV_CH], gt[EV_CH], le[EV_CH];
#pragma unroll
        for (int k = 0; k < EV_CH; ++k) {
            const int i = c0 + k < n_pos ? pos_idx[c0 + k] : -1;
            sp[k] = i >= 0 ? s[i] : 0.0f;
            vp[k] = i >= 0 ? (em[i] ? 0.0f : expf(sp[k])) : 0.0f;   // :33 / :56 exp(pred) * not(excl)
            gt[k] = 0.0f; le[k] = 0.0f;
        }
        for (int64_t j = tid; j < a.NI; j += 256) {
            const float sj = s[j];
            const bool e = em[j] != 0;
            const float vj = e ? 0.0f : expf(sj);
            const bool ev = !e && pm[j] == 0;
#pragma unroll
            for (int k = 0; k < EV_CH; ++k) {
                gt[k] += vj > vp[k] ? 1.0f : 0.0f;                 // :35 rank_above
                le[k] += (ev && sj <= sp[k]) ? 1.0f : 0.0f;        // :18 eval_pred <= pos_pred
            }
        }
#pragma unroll
        for (int k = 0; k < EV_CH; ++k) {
            for (int off = 32; off > 0; off >>= 1) { gt[k] += __shfl_xor(gt[k], off); le[k] += __shfl_xor(le[k], off); }
            if (lane == 0) { red[wave][k] = gt[k]; red[wave][EV_CH + k] = le[k]; }
        }
        __syncthreads();
        if (tid < EV_CH && c0 + tid < n_pos) {
            const float g = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            const float l = red[0][EV_CH + tid] + red[1][EV_CH + tid] + red[2][EV_CH + tid] + red[3][EV_CH + tid];
            atomicAdd(&auc_sum_s, l);
            const float lr = 1.0f / (logf(g + 2.0f) / logf(2.0f));   // :38 reciprocal(log2(rank_above + 2))
            for (int t = 0; t < a.nat; ++t) {
                if (g < a.at[t]) { atomicAdd(&acc_ndcg[t], lr); atomicAdd(&acc_rec[t], 1.0f); }
            }
        }
        __syncthreads();
    }
    if (tid == 0) a.auc[q] = auc_sum_s / ((float)n_pos * (float)n_eval_s);          // :18-19 (0/0 -> NaN like TF)
    if (tid < a.nat) {
        a.ndcg[q * a.nat + tid] = acc_ndcg[tid];
        a.recall[q * a.nat + tid] = acc_rec[tid] / (float)n_pos;                      // :62-63
    }
}

int orx_launch_rank_metrics(orx_ctx* ctx, const EvalArgs& a, int64_t n) {
    if (n == 0) return ORX_OK;
    ORX_LAUNCH(ctx, rank_metrics_kernel, dim3((unsigned)n), dim3(256), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}


// ---- the same metrics from CSR lists (no [n, items] masks cross PCIe or sit in HBM as bytes) -------------------------
// Each user's positives / exclusions arrive as item lists; they are scattered into two bitmaps ([n][W] words, W = ceil(NI / 32):
// 1/32 of the byte masks) and the positives are taken straight from the list, so the item row is swept once per CH positives
// with no compaction pass.  Same arithmetic as rank_metrics_kernel: exp() of the scores for the rank counts, raw scores for AUC.
__global__ __launch_bounds__(256) void mask_bits_kernel(EvalCsrArgs a, int clear) {
    const int64_t q = blockIdx.x;
    for (int which = 0; which < 2; ++which) {
        const int64_t* ptr = which ? a.excl_ptr : a.pos_ptr;
        const int32_t* it = which ? a.excl_items : a.pos_items;
        unsigned* bits = (which ? a.ebits : a.pbits) + q * a.W;
        const int64_t lo = ptr[q], hi = ptr[q + 1];
        for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
            const int32_t i = it[e];
            if (i < 0 || i >= a.NI) { if (!clear) *a.err = 1; continue; }
            if (clear) bits[i >> 5] = 0u;                             // the bitmaps go back to all-zero: no memset of 2 n W words per call
            else atomicOr(&bits[i >> 5], 1u << (i & 31));
        }
    }
    if (clear && q == 0 && threadIdx.x == 0) { *a.flag_out = *a.err; *a.err = 0; }   // (stream order: after every sweep / finish)
}

// One sweep over the user's score row serves NB - 1 positives (NB = 2^STEPS): the positives' scores are sorted into a threshold
// table in LDS, every evaluated item finds its bucket d = #{thresholds < s_j} by a branch-free binary search (STEPS LDS reads) and
// bumps a per-thread histogram column (no bank conflicts); prefix sums of the bucket totals give, for the k-th threshold,
//  #{s_j <= sp_k} = sum_{d <= k} L[d]   (ranking_metrics.py:18)   and   #{s_j > sp_k} = sum_{d > k} L[d].
// The reference ranks on exp(pred) (ranking_metrics.py:33-35), which can round two different scores to the same value: an item
// within 1e-6 of the threshold below it takes the exact test  expf(s_j) > expf(sp_k)  and is booked in corr[] when it fails.
// ~14 + 3 STEPS VALU operations per item instead of ~10 per (item, positive).  A row is cut into S segments (grid S x n) so that
// a batch of a few dozen users still fills the chip; the segments' bucket totals meet in rank_finish_kernel.
constexpr int RK_PS = 136;          // words per (user, segment) partial: tot[<= 64] | corr[<= 63] at 64 | nz 128 | covered 129 | positives 130

template <int STEPS>
struct RankLds {
    static constexpr int NB = 1 << STEPS, NT = NB - 1;
    float ts[NB + 1], tv[NB + 1];                                   // ts[0] = -inf, ts[1 + m] = m-th smallest positive score; tv = expf(ts)
    float raw_s[NT]; int raw_ex[NT], t_ex[NT];
};

// the chunk's thresholds, sorted by score (rank sort: at most 63 of them); every thread of the block calls this
template <int STEPS>
__device__ __forceinline__ void rank_thresholds(RankLds<STEPS>& L, const EvalCsrArgs& a, const float* s, const unsigned* eb,
                                                int64_t p0, int c0, int pc) {
    constexpr int NT = RankLds<STEPS>::NT;
    const int tid = threadIdx.x;
    if (tid < NT) {
        int i = tid < pc ? a.pos_items[p0 + c0 + tid] : -1;
        if ((int64_t)i >= a.NI) i = -1;                   // (an item id outside the table: mask_bits has flagged it, nothing is read for it)
        L.raw_s[tid] = i >= 0 ? s[i] : INFINITY;
        L.raw_ex[tid] = i >= 0 && ((eb[i >> 5] >> (i & 31)) & 1u);
    }
    if (tid == 0) { L.ts[0] = -INFINITY; L.tv[0] = 0.0f; }
    __syncthreads();
    if (tid < NT) {
        const float x = L.raw_s[tid];
        int r = 0;
        for (int m = 0; m < NT; ++m) r += (L.raw_s[m] < x) || (L.raw_s[m] == x && m < tid);
        L.ts[1 + r] = x; L.tv[1 + r] = tid < pc ? expf(x) : INFINITY; L.t_ex[r] = L.raw_ex[tid];
    }
    __syncthreads();
}

template <int STEPS>
__global__ __launch_bounds__(256) void rank_sweep_kernel(EvalCsrArgs a, int c0) {
    constexpr int NB = 1 << STEPS, NT = NB - 1;
    constexpr int COLS = STEPS <= 4 ? 256 : 64;                     // histogram columns: one per thread, or (64 buckets) one per lane shared by the 4 waves
    __shared__ unsigned hist[NB * COLS];                            // [bucket][column]
    __shared__ uint2 wbits[2][256];
    __shared__ RankLds<STEPS> L;
    __shared__ unsigned corr[NT];
    __shared__ unsigned nz_s, cover_s, npos_s;
    const int64_t q = a.q0 + blockIdx.y;
    const int seg = blockIdx.x;
    const float* s = a.pred + q * a.NI;
    const unsigned* pb = a.pbits + q * a.W;
    const unsigned* eb = a.ebits + q * a.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t p0 = a.pos_ptr[q];
    const int n_pos = (int)(a.pos_ptr[q + 1] - p0);
    if (c0 > 0 && c0 >= n_pos) return;                              // a chunk beyond this user's positives
    const int pc = max(0, min(NT, n_pos - c0));
    unsigned* out = a.part + ((size_t)q * a.S + seg) * RK_PS;
    rank_thresholds<STEPS>(L, a, s, eb, p0, c0, pc);
    for (int r = tid; r < NB * COLS; r += 256) hist[r] = 0;
    if (tid < NT) corr[tid] = 0;
    if (tid == 0) { nz_s = 0; cover_s = 0; npos_s = 0; }
    __syncthreads();
    // this segment: whole blocks of 8192 items (256 words of each bitmap)
    const unsigned NIu = (unsigned)a.NI;
    const unsigned nblk = (NIu + 8191u) / 8192u, bps = (nblk + a.S - 1) / a.S;
    const unsigned blk_lo = min(nblk, seg * bps), blk_hi = min(nblk, blk_lo + bps);
    if (c0 == 0) {                                                  // items that are positive or excluded; distinct positives
        int covered = 0, npos_bits = 0;
        for (int64_t wd = (int64_t)blk_lo * 256 + tid; wd < min(a.W, (int64_t)blk_hi * 256); wd += 256) {
            unsigned pw = pb[wd], ew = eb[wd];
            if (wd == a.W - 1 && (a.NI & 31)) { const unsigned m = (1u << (a.NI & 31)) - 1u; pw &= m; ew &= m; }
            covered += __popc(pw | ew); npos_bits += __popc(pw);
        }
        for (int off = 32; off > 0; off >>= 1) { covered += __shfl_xor(covered, off); npos_bits += __shfl_xor(npos_bits, off); }
        if (lane == 0) { atomicAdd(&cover_s, (unsigned)covered); atomicAdd(&npos_s, (unsigned)npos_bits); }
    }
    // ---- the sweep: groups of 8 x 256 items; the next group's scores are in flight while this one is searched
    unsigned nz = 0;
    const unsigned g_lo = blk_lo * 4, g_hi = min((NIu + 2047u) / 2048u, blk_hi * 4);
    if (pc > 0 && g_lo < g_hi) {
        float cur[8], nxt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = s[min(g_lo * 2048u + u * 256 + tid, NIu - 1u)];
        for (unsigned g = g_lo; g < g_hi; ++g) {
            const int buf = (g >> 2) & 1;
            if ((g & 3u) == 0) {                                    // bitmap words of the next 8192 items
                const int64_t wd = (int64_t)g * 64 + tid;
                wbits[buf][tid] = wd < a.W ? make_uint2(pb[wd], eb[wd]) : make_uint2(0u, 0u);
                __syncthreads();                                    // (two buffers: the next block's store cannot overtake these reads)
            }
            const unsigned base = g * 2048u + tid;
            if (g + 1 < g_hi) {
#pragma unroll
                for (int u = 0; u < 8; ++u) nxt[u] = s[min(base + 2048u + u * 256, NIu - 1u)];
            }
            uint2 w2[8]; int dd[8]; float tb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { w2[u] = wbits[buf][((g & 3u) * 8 + u) * 8 + (tid >> 5)]; dd[u] = 0; }
#pragma unroll
            for (int h = NB / 2; h >= 1; h >>= 1) {                 // the eight searches level by level: eight LDS reads in flight
#pragma unroll
                for (int u = 0; u < 8; ++u) dd[u] += (L.ts[dd[u] + h] < cur[u]) ? h : 0;        // ts[1 + (d + h - 1)]
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) tb[u] = L.ts[dd[u]];        // the threshold just below the item (ts[0] = -inf)
            const unsigned bit = 1u << (tid & 31);
            unsigned slow = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool valid = base + u * 256 < NIu;
                const bool e = w2[u].y & bit, p = w2[u].x & bit;
                const bool ev = valid && !p && !e;
                atomicAdd(&hist[dd[u] * COLS + (tid & (COLS - 1))], ev ? 1u : 0u);  // (each lane its own bank)
                nz += (valid && !e) ? 1u : 0u;                      // exp(pred) > 0 (what an excluded positive is ranked against); < -80: below
                // rare: exp may round the item onto the threshold below it (scores closer than 1e-6, or both where exp
                // overflows / leaves the normal range)
                const float sj = cur[u];
                const bool odd = (ev && (sj - tb[u] < 1e-6f || sj > 88.0f)) || (valid && !e && sj < -80.0f);
                slow |= odd ? 1u << u : 0u;
            }
            if (slow) {                                             // from scratch, one item at a time (no register arrays indexed by u)
#pragma unroll 1
                for (int u = 0; u < 8; ++u) {
                    if (!((slow >> u) & 1u)) continue;
                    const float sj = s[base + u * 256];
                    const uint2 w = wbits[buf][((g & 3u) * 8 + u) * 8 + (tid >> 5)];
                    const bool e = w.y & bit, p = w.x & bit;
                    const float vj = expf(sj);
                    if (!e && sj < -80.0f && !(vj > 0.0f)) --nz;
                    if (p || e) continue;
                    int d = 0;
                    for (int h = NB / 2; h >= 1; h >>= 1) d += (L.ts[d + h] < sj) ? h : 0;
                    int c = d;
                    while (c > 0 && (sj - L.ts[c] < 1e-6f || sj > 88.0f || sj < -87.0f)) {
                        if (L.tv[c] >= vj) --c; else break;
                    }
                    for (int m = c; m < d; ++m) atomicAdd(&corr[m], 1u);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        }
    }
    for (int off = 32; off > 0; off >>= 1) nz += __shfl_xor(nz, off);
    if (lane == 0) atomicAdd(&nz_s, nz);
    __syncthreads();
    for (int bkt = wave; bkt < NB; bkt += 4) {                      // bucket totals
        unsigned v = 0;
        for (int col = lane; col < COLS; col += 64) v += hist[bkt * COLS + col];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) out[bkt] = v;
    }
    if (tid < NT) out[64 + tid] = corr[tid];
    if (tid == 0) { out[128] = nz_s; out[129] = cover_s; out[130] = npos_s; }
}

// one wave per user: the segments' totals, the positives ranked against each other, the metric sums of this chunk
template <int STEPS>
__global__ __launch_bounds__(64) void rank_finish_kernel(EvalCsrArgs a, int c0, int last) {
    constexpr int NB = 1 << STEPS, NT = NB - 1;
    __shared__ RankLds<STEPS> L;
    __shared__ unsigned sum[RK_PS], gtp[NT];
    const int64_t q = a.q0 + blockIdx.x;
    const int lane = threadIdx.x;
    const float* s = a.pred + q * a.NI;
    const unsigned* eb = a.ebits + q * a.W;
    const int64_t p0 = a.pos_ptr[q];
    const int n_pos = (int)(a.pos_ptr[q + 1] - p0);
    if (c0 == 0 || c0 < n_pos) {
        const int pc = max(0, min(NT, n_pos - c0));
        for (int col = lane; col < RK_PS; col += 64) {
            unsigned v = 0;
            for (int seg = 0; seg < a.S; ++seg) v += a.part[((size_t)q * a.S + seg) * RK_PS + col];
            sum[col] = v;
        }
        if (lane < NT) gtp[lane] = 0;
        rank_thresholds<STEPS>(L, a, s, eb, p0, c0, pc);
        if (c0 == 0) {
            if (sum[130] != (unsigned)n_pos) { if (lane == 0) atomicCAS(a.err, 0, 3); }   // a repeated positive: the lists must be sets
            if (lane == 0) { a.neval[q] = (int)a.NI - (int)sum[129]; a.auc[q] = 0.0f; }   // ranking_metrics.py:14
            if (lane < a.nat) { a.ndcg[q * a.nat + lane] = 0.0f; a.recall[q * a.nat + lane] = 0.0f; }
        }
        // the positives themselves are ranked against too (they are not excluded from rank_above)
        for (int m = lane; m < n_pos; m += 64) {
            const int i = a.pos_items[p0 + m];
            if ((eb[i >> 5] >> (i & 31)) & 1u) continue;
            const float vm = expf(s[i]);
            for (int k = 0; k < pc; ++k) if (vm > L.tv[1 + k]) atomicAdd(&gtp[k], 1u);
        }
        __syncthreads();
        if (lane == 0 && pc > 0) {                                  // at most 63 positives: one thread, a fixed order of the float sums
            unsigned below = 0, all = 0;
            for (int d = 0; d < NB; ++d) all += sum[d];
            // non-excluded positives with exp(pred) > 0 belong to what an excluded positive is ranked against: sum[128] holds them
            float auc_sum = a.auc[q], nd[16], rc[16];
            for (int t = 0; t < a.nat; ++t) { nd[t] = a.ndcg[q * a.nat + t]; rc[t] = a.recall[q * a.nat + t]; }
            for (int k = 0; k < pc; ++k) {
                below += sum[k];                                    // sum_{d <= k} L[d]
                const unsigned above = L.t_ex[k] ? sum[128] : (all - below) - sum[64 + k] + gtp[k];
                auc_sum += (float)below;
                const float g = (float)above;
                const float lr = 1.0f / (logf(g + 2.0f) / logf(2.0f));   // :38 reciprocal(log2(rank_above + 2))
                for (int t = 0; t < a.nat; ++t)
                    if (g < a.at[t]) { nd[t] += lr; rc[t] += 1.0f; }
            }
            a.auc[q] = auc_sum;
            for (int t = 0; t < a.nat; ++t) { a.ndcg[q * a.nat + t] = nd[t]; a.recall[q * a.nat + t] = rc[t]; }
        }
    }
    if (last) {
        __syncthreads();
        if (lane == 0) a.auc[q] = a.auc[q] / ((float)n_pos * (float)a.neval[q]);     // :18-19 (0/0 -> NaN like TF)
        if (lane < a.nat) a.recall[q * a.nat + lane] = a.recall[q * a.nat + lane] / (float)n_pos;   // :62-63
    }
}

static int rank_steps(int64_t max_pos) { return max_pos <= 7 ? 3 : (max_pos <= 15 ? 4 : 6); }

// segments per user: enough workgroups to fill the chip several times over, none shorter than 4 blocks of 8192 items
int orx_rank_csr_segments(int64_t n, int64_t NI) {
    const int64_t nblk = (NI + 8191) / 8192;
    int64_t S = (8192 + n - 1) / (n > 0 ? n : 1);
    if (S > (nblk + 3) / 4) S = (nblk + 3) / 4;
    return (int)(S < 1 ? 1 : S);
}

int orx_launch_mask_bits(orx_ctx* ctx, const EvalCsrArgs& a, int64_t n, int clear) {
    if (n == 0) return ORX_OK;
    ORX_LAUNCH(ctx, mask_bits_kernel, dim3((unsigned)n), dim3(256), 0, a, clear);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// users [q0, q0 + nq) on ctx->stream: per chunk of positives one sweep over the score rows and one finish
int orx_launch_rank_sweeps(orx_ctx* ctx, const EvalCsrArgs& a_in, int64_t q0, int64_t nq, int64_t max_pos) {
    if (nq == 0) return ORX_OK;
    EvalCsrArgs a = a_in; a.q0 = q0;
    const int steps = rank_steps(max_pos), NT = (1 << steps) - 1;
    const int chunks = max_pos <= NT ? 1 : (int)((max_pos + NT - 1) / NT);
    const dim3 grid((unsigned)a.S, (unsigned)nq);
    for (int c = 0; c < chunks; ++c) {
        const int c0 = c * NT, last = c == chunks - 1;
        if (steps == 3) { ORX_LAUNCH(ctx, rank_sweep_kernel<3>, grid, dim3(256), 0, a, c0); ORX_LAUNCH(ctx, rank_finish_kernel<3>, dim3((unsigned)nq), dim3(64), 0, a, c0, last); }
        else if (steps == 4) { ORX_LAUNCH(ctx, rank_sweep_kernel<4>, grid, dim3(256), 0, a, c0); ORX_LAUNCH(ctx, rank_finish_kernel<4>, dim3((unsigned)nq), dim3(64), 0, a, c0, last); }
        else { ORX_LAUNCH(ctx, rank_sweep_kernel<6>, grid, dim3(256), 0, a, c0); ORX_LAUNCH(ctx, rank_finish_kernel<6>, dim3((unsigned)nq), dim3(64), 0, a, c0, last); }
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
