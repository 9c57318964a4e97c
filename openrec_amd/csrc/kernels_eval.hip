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
