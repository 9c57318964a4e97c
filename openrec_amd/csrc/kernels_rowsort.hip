// Deterministic application of per-occurrence gradient rows to an embedding table (the IndexedSlices of
// tf2_examples/dlrm_criteo.py:44-47 / bpr_citeulike.py:35-38 as Keras applies them: SGD scatter-add of every occurrence,
// Adagrad / Adam on the SUM of a row's occurrences -- SURVEY.md A.3-A.5).
//
//   1. stable LSD radix sort of (row id, position) pairs, K id lists at once (all steps of a K-step call are sorted before
//      the first of them runs: the ids are known up front);
//   2. one wavefront per 64 consecutive sorted entries sums the gradient rows of every run of equal ids IN POSITION ORDER and
//      applies the optimizer rule once per distinct row; runs that cross a 64-entry boundary leave partial sums that a second
//      launch adds in block order.
//
// No fp32 atomics and no arrival-order ranks anywhere: the result is a function of the inputs alone (bit-identical from run
// to run), whatever the duplicate structure -- tables of 3 rows with thousands of references per row and tables of 10 M rows
// take the same path.  HBM-bound: the gradient rows are read once, every distinct table row is read and written once.
#include <cstring>

#include "orx_csr_device.h"

namespace {

constexpr int SORT_WAVES = 4, SORT_ROUNDS = 8, SORT_TILE = 64 * SORT_WAVES * SORT_ROUNDS;     // 2048 keys per workgroup
constexpr int SORT_MAXBITS = 9;
constexpr uint32_t KEY_NONE = 0xffffffffu;

struct SortArgs {
    const int32_t* ids; int64_t id_stride;     // first pass: list k at ids + k * id_stride (NULL: read `in`)
    const uint2* in; uint2* out;               // [K][n] (key, position)
    int* hist;                                 // [K][nbins][nblk]
    int* binbase;                              // [K][nbins] first output position of a digit (after the scans)
    int64_t n; int nblk; uint32_t sentinel;    // sentinel = table rows: key of padding (id < 0) and out-of-range ids, sorts last
    int shift, bits;
    int* err;
};

__device__ __forceinline__ uint2 sort_load(const SortArgs& a, int64_t k, int64_t i) {
    if (a.ids != nullptr) {
        const int id = a.ids[k * a.id_stride + i];
        uint32_t key = (uint32_t)id;
        if (id < 0) key = a.sentinel;
        else if (key >= a.sentinel) { *a.err = 1; key = a.sentinel; }
        return make_uint2(key, (uint32_t)i);
    }
    return a.in[k * a.n + i];
}

__global__ __launch_bounds__(256) void sort_hist_kernel(SortArgs a) {
    __shared__ int h[1 << SORT_MAXBITS];
    const int nbins = 1 << a.bits;
    for (int d = threadIdx.x; d < nbins; d += 256) h[d] = 0;
    __syncthreads();
    const int64_t k = blockIdx.y, t0 = (int64_t)blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_TILE / 256; ++r) {
        const int64_t i = t0 + r * 256 + threadIdx.x;
        if (i < a.n) atomicAdd(&h[(sort_load(a, k, i).x >> a.shift) & (nbins - 1)], 1);      // (integer counts: order-free)
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nbins; d += 256) a.hist[((int64_t)k * nbins + d) * a.nblk + blockIdx.x] = h[d];
}

// Exclusive prefix over the (digit-major) counts of every list, in two small launches: one wavefront per (list, digit) scans
// that digit's per-workgroup counts in place and leaves the digit's total; one workgroup per list scans the totals.
__global__ __launch_bounds__(256) void sort_scan_blocks_kernel(int* hist, int* bintotal, int nblk, int64_t nrows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // (list, digit)
    if (row >= nrows) return;
    int* p = hist + row * nblk;
    int carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int c = b0 + lane < nblk ? p[b0 + lane] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (b0 + lane < nblk) p[b0 + lane] = carry + incl - c;
        carry += __shfl(incl, 63);
    }
    if (lane == 0) bintotal[row] = carry;
}

__global__ __launch_bounds__(512) void sort_scan_bins_kernel(int* bintotal, int nbins) {
    __shared__ int sh[1 << SORT_MAXBITS];
    int* p = bintotal + (int64_t)blockIdx.x * nbins;
    const int c = (int)threadIdx.x < nbins ? p[threadIdx.x] : 0;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 512; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    if ((int)threadIdx.x < nbins) p[threadIdx.x] = sh[threadIdx.x] - c;
}

// Stable scatter: entry i of the tile belongs to wavefront (i / 512), round (i / 64) % 8, lane i % 64.  Equal digits keep
// their order: inside a round by lane (ballot match + popcount of the lower lanes), across rounds and wavefronts by an
// exclusive prefix over the per-(wavefront, round) counts, across workgroups by the scanned histogram.
__global__ __launch_bounds__(256) void sort_scatter_kernel(SortArgs a) {
    extern __shared__ unsigned short cnt[];                  // [WAVES * ROUNDS][nbins] counts, then prefixes (<= 2048)
    __shared__ int goff[1 << SORT_MAXBITS];
    const int nbins = 1 << a.bits;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k = blockIdx.y, t0 = (int64_t)blockIdx.x * SORT_TILE;
    for (int e = threadIdx.x; e < SORT_WAVES * SORT_ROUNDS * nbins; e += 256) cnt[e] = 0;
    for (int d = threadIdx.x; d < nbins; d += 256) goff[d] = a.binbase[(int64_t)k * nbins + d] + a.hist[((int64_t)k * nbins + d) * a.nblk + blockIdx.x];
    __syncthreads();
    uint2 kv[SORT_ROUNDS]; int rank[SORT_ROUNDS];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const int64_t i = t0 + (wave * SORT_ROUNDS + r) * 64 + lane;
        const bool valid = i < a.n;
        kv[r] = valid ? sort_load(a, k, i) : make_uint2(0u, 0u);
        const int digit = (kv[r].x >> a.shift) & (nbins - 1);
        unsigned long long m = __ballot(valid);
        for (int b = 0; b < a.bits; ++b) {
            const bool bit = (digit >> b) & 1;
            const unsigned long long bal = __ballot(valid && bit);
            m &= bit ? bal : ~bal;
        }
        rank[r] = valid ? __popcll(m & below) : -1;
        if (valid && rank[r] == 0) cnt[(wave * SORT_ROUNDS + r) * nbins + digit] = (unsigned short)__popcll(m);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nbins; d += 256) {
        int run = 0;
        for (int wr = 0; wr < SORT_WAVES * SORT_ROUNDS; ++wr) { const int c = cnt[wr * nbins + d]; cnt[wr * nbins + d] = (unsigned short)run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        if (rank[r] < 0) continue;
        const int digit = (kv[r].x >> a.shift) & (nbins - 1);
        a.out[k * a.n + goff[digit] + cnt[(wave * SORT_ROUNDS + r) * nbins + digit] + rank[r]] = kv[r];
    }
}

template <int NE, int MODE>
__global__ __launch_bounds__(256) void csr_apply_kernel(CsrArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), i0 = b * 64;
    if (i0 >= a.n) return;
    const int cnt = (int)(a.n - i0 < 64 ? a.n - i0 : 64);
    const uint2 mine = lane < cnt ? a.sorted[i0 + lane] : make_uint2(KEY_NONE, 0u);
    const uint32_t prevk = b > 0 ? a.sorted[i0 - 1].x : KEY_NONE, nextk = i0 + 64 < a.n ? a.sorted[i0 + 64].x : KEY_NONE;
    // skip_single: an entry whose row differs from both neighbours is left out (its row was updated where its gradient was
    // formed).  The remaining entries -- whole runs, in order -- are numbered densely: lane l learns the block position of the
    // l-th of them (runs that cross a block boundary are never single, so the open ends below keep their meaning).
    int todo_n = cnt, orig = lane;
    if (a.skip_single) {
        const uint32_t lk = (uint32_t)__shfl_up((int)mine.x, 1), rk = (uint32_t)__shfl_down((int)mine.x, 1);
        const uint32_t left = lane == 0 ? prevk : lk, right = lane == cnt - 1 ? nextk : rk;
        const bool keep = lane < cnt && mine.x < a.rows && !(mine.x != left && mine.x != right);     // (padding ids: nothing to apply)
        const unsigned long long m = __ballot(keep);
        todo_n = __popcll(m);
        if (todo_n == 0) return;
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        orig = __builtin_amdgcn_ds_permute((keep ? rank : 63) * 4, lane);       // (lanes without an entry all write lane 63: only a block
        if (todo_n == 64) orig = lane;                                           //  of 64 kept entries reads it, and then the map is the identity)
    }
    float acc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[e] = 0.0f;
    uint32_t cur = KEY_NONE;
    int seg_first = 0, seg_last = 0;                         // block positions of the current run's first / latest entry
    RowState<NE, MODE> st_cur;
    auto flush = [&]() {
        if (cur >= a.rows) return;                           // padding / out-of-range ids (sorted last); nothing before the first entry
        const bool open_lo = seg_first == 0 && prevk == cur, open_hi = seg_last == 63 && nextk == cur;
        if (!open_lo && !open_hi) { csr_rule<NE, MODE>(a, cur, acc, st_cur, lane); return; }
        float* p = (open_lo ? a.part_lo : a.part_hi) + (size_t)b * a.Dp;
#pragma unroll
        for (int e = 0; e < NE; ++e) if (lane + 64 * e < a.D) p[lane + 64 * e] = acc[e];
    };
    constexpr int UN = 8;
    for (int j0 = 0; j0 < todo_n; j0 += UN) {
        uint32_t kj[UN]; int oj[UN]; float g[UN][NE]; RowState<NE, MODE> st[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {                       // UN gradient rows and the table rows they belong to in flight
            const int j = j0 + u < todo_n ? j0 + u : todo_n - 1;
            oj[u] = __builtin_amdgcn_readlane(orig, j);
            kj[u] = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, oj[u]);
            const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, oj[u]);
            const float* gp = a.grads + (size_t)pj * a.g_stride;
#pragma unroll
            for (int e = 0; e < NE; ++e) g[u][e] = (kj[u] < a.rows && lane + 64 * e < a.D) ? gp[lane + 64 * e] : 0.0f;
            // (only the entry that starts a run uses its table row; the others re-read a line the run's head just fetched)
            const uint32_t before = oj[u] > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)mine.x, oj[u] - 1) : KEY_NONE;
            if (j0 + u < todo_n && kj[u] != before) st[u].load(a, kj[u], lane);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (j0 + u >= todo_n) break;
            if (kj[u] != cur || j0 + u == 0) {
                if (j0 + u > 0) flush();
                cur = kj[u]; seg_first = oj[u]; st_cur = st[u];
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[e] = 0.0f;
            }
            seg_last = oj[u];
#pragma unroll
            for (int e = 0; e < NE; ++e) acc[e] += g[u][e];
        }
    }
    flush();
}

// runs that cross block boundaries (csr_finish_block, orx_csr_device.h) as a launch of their own
template <int NE, int MODE>
__global__ __launch_bounds__(256) void csr_finish_kernel(CsrArgs a) {
    csr_finish_block<NE, MODE>(a, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// lazy Adam: every distinct row of the list is replayed to step T before a forward pass reads it
__global__ __launch_bounds__(256) void csr_touch_kernel(CsrArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.n) return;
    const uint32_t row = a.sorted[i].x;
    if (row >= a.rows || (i > 0 && a.sorted[i - 1].x == row)) return;
    const int from = __builtin_amdgcn_readfirstlane(a.last[row]);
    if (from >= a.T) return;
    for (int col = lane; col < a.D; col += 64) {
        const size_t k = (size_t)row * a.D + col;
        float w = a.W[k], m = a.A[k], v = a.V[k];
        adam_replay1<true>(w, m, v, from, a.T, a.lrt, a.b1, a.b2, a.eps, a.newton != 0, a.cf);
        a.W[k] = w; a.A[k] = m; a.V[k] = v;
    }
    if (lane == 0) a.last[row] = a.T;
}

// finish == false: the first launch only (the caller has the finish pass carried by another launch: orx_csr_apply_split)
template <int MODE>
int launch_csr(orx_ctx* ctx, const CsrArgs& a, bool finish = true) {
    const int64_t nblk = (a.n + 63) / 64;
    const dim3 g((unsigned)((nblk + 3) / 4));
    const int ne = (a.D + 63) / 64;
#define ORX_CSR_GO(NE)                                                                  \
    do {                                                                                \
        ORX_LAUNCH(ctx, (csr_apply_kernel<NE, MODE>), g, dim3(256), 0, a);              \
        if (finish) ORX_LAUNCH(ctx, (csr_finish_kernel<NE, MODE>), g, dim3(256), 0, a); \
    } while (0)
    switch (ne) {
        case 1: ORX_CSR_GO(1); break;
        case 2: ORX_CSR_GO(2); break;
        case 3: ORX_CSR_GO(3); break;
        default: ORX_CSR_GO(4); break;
    }
#undef ORX_CSR_GO
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

}  // namespace

static inline int bit_width_u64(uint64_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; }

// flags[k][position] = 1 where the row at that position of list k is referenced by no other position of the list (its sorted
// neighbours differ), else 0 -- every position of a list appears exactly once in its sorted form, so every flag is written.
// The kernel that forms the gradient of such a reference may apply it in place: nobody else reads or writes the row in the step.
__global__ __launch_bounds__(256) void rows_single_flags_kernel(const uint2* sorted, int64_t n, uint32_t rows, unsigned char* flags) {
    const int64_t k = blockIdx.y;
    const uint2* s = sorted + k * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint2 e = s[i];
        const bool single = e.x < rows && (i == 0 || s[i - 1].x != e.x) && (i + 1 >= n || s[i + 1].x != e.x);
        flags[k * n + e.y] = single ? 1 : 0;
    }
}

int orx_rows_single_flags(orx_ctx* ctx, const uint2* sorted, int64_t K, int64_t n, int64_t rows, unsigned char* flags) {
    if (K == 0 || n == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_DEDUP);
    ORX_LAUNCH(ctx, rows_single_flags_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048), (unsigned)K), dim3(256), 0, sorted, n, (uint32_t)rows, flags);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// the sort's buffers for K lists of n ids, without sorting: a caller whose calls vary in length reserves for its longest chunk
// once (growing them inside a longer call costs hipFree + hipMalloc: ~0.5 ms of the DLRM step's first long call)
int orx_rows_sort_reserve(orx_ctx* ctx, int64_t K, int64_t n, int64_t rows) {
    if (K <= 0 || n <= 0) return ORX_OK;
    const int total_bits = bit_width_u64((uint64_t)rows);
    const int passes = (total_bits + SORT_MAXBITS - 1) / SORT_MAXBITS;
    const int bits = (total_bits + passes - 1) / passes;
    const int nblk = (int)((n + SORT_TILE - 1) / SORT_TILE), nbins = 1 << bits;
    if (orx_ensure((void**)&ctx->d_sort[0], &ctx->d_sort_cap[0], (size_t)K * n * sizeof(uint2)) != ORX_OK) return ORX_ERR_OOM;
    if (orx_ensure((void**)&ctx->d_sort[1], &ctx->d_sort_cap[1], (size_t)K * n * sizeof(uint2)) != ORX_OK) return ORX_ERR_OOM;
    if (orx_ensure((void**)&ctx->d_sort_hist, &ctx->d_sort_hist_cap, (size_t)K * nbins * (nblk + 1) * sizeof(int)) != ORX_OK) return ORX_ERR_OOM;
    return ORX_OK;
}

// sorted (row, position) pairs of K id lists of n ids each; `out` points into the context's sort buffers (valid until the
// next orx_rows_sort on this context)
int orx_rows_sort(orx_ctx* ctx, const int32_t* ids, int64_t K, int64_t n, int64_t id_stride, int64_t rows, const uint2** out) {
    ORX_ARG(rows > 0 && rows < (1LL << 31) && n < (1LL << 31), "rows_sort: tables of up to 2^31 - 1 rows, lists of up to 2^31 - 1 ids");
    *out = nullptr;
    if (K == 0 || n == 0) return ORX_OK;
    const int total_bits = bit_width_u64((uint64_t)rows);                 // the sentinel `rows` must be representable
    const int passes = (total_bits + SORT_MAXBITS - 1) / SORT_MAXBITS;
    const int bits = (total_bits + passes - 1) / passes;
    const int nblk = (int)((n + SORT_TILE - 1) / SORT_TILE), nbins = 1 << bits;
    if (orx_ensure((void**)&ctx->d_sort[0], &ctx->d_sort_cap[0], (size_t)K * n * sizeof(uint2)) != ORX_OK) return ORX_ERR_OOM;
    if (orx_ensure((void**)&ctx->d_sort[1], &ctx->d_sort_cap[1], (size_t)K * n * sizeof(uint2)) != ORX_OK) return ORX_ERR_OOM;
    if (orx_ensure((void**)&ctx->d_sort_hist, &ctx->d_sort_hist_cap, (size_t)K * nbins * (nblk + 1) * sizeof(int)) != ORX_OK) return ORX_ERR_OOM;
    ProfScope ps(ctx, ORX_K_DEDUP);
    SortArgs a;
    a.id_stride = id_stride; a.hist = ctx->d_sort_hist; a.binbase = ctx->d_sort_hist + (size_t)K * nbins * nblk; a.n = n; a.nblk = nblk; a.sentinel = (uint32_t)rows; a.bits = bits; a.err = ctx->d_err;
    int src = 1;                                                           // (first pass reads the ids)
    for (int p = 0; p < passes; ++p) {
        a.ids = p == 0 ? ids : nullptr;
        a.in = ctx->d_sort[src]; a.out = ctx->d_sort[src ^ 1]; a.shift = p * bits;
        const dim3 g((unsigned)nblk, (unsigned)K);
        ORX_LAUNCH(ctx, sort_hist_kernel, g, dim3(256), 0, a);
        ORX_LAUNCH(ctx, sort_scan_blocks_kernel, dim3((unsigned)((K * nbins + 3) / 4)), dim3(256), 0, a.hist, a.binbase, nblk, (int64_t)K * nbins);
        ORX_LAUNCH(ctx, sort_scan_bins_kernel, dim3((unsigned)K), dim3(512), 0, a.binbase, nbins);
        ORX_LAUNCH(ctx, sort_scatter_kernel, g, dim3(256), (size_t)SORT_WAVES * SORT_ROUNDS * nbins * sizeof(unsigned short), a);
        src ^= 1;
    }
    ORX_HIP(hipGetLastError());
    *out = ctx->d_sort[src];
    return ORX_OK;
}

static int csr_args(orx_ctx* ctx, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride, CsrArgs* a) {
    memset(a, 0, sizeof(*a));
    ORX_ARG(t->dim <= 256, "rows apply: dims up to 256 (got %d)", t->dim);
    a->sorted = sorted; a->n = n; a->rows = (uint32_t)t->rows; a->D = t->dim; a->grads = grads; a->g_stride = g_stride; a->W = t->w;
    a->Dp = (t->dim + 3) & ~3;
    const size_t bytes = (size_t)((n + 63) / 64) * a->Dp * sizeof(float);
    if (orx_ensure((void**)&ctx->d_csr_part[0], &ctx->d_csr_part_cap[0], bytes) != ORX_OK) return ORX_ERR_OOM;
    if (orx_ensure((void**)&ctx->d_csr_part[1], &ctx->d_csr_part_cap[1], bytes) != ORX_OK) return ORX_ERR_OOM;
    a->part_lo = ctx->d_csr_part[0]; a->part_hi = ctx->d_csr_part[1];
    return ORX_OK;
}

// SGD / Adagrad on the sorted list (one step)
int orx_csr_apply(orx_ctx* ctx, orx_opt* opt, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride, bool skip_single) {
    if (n == 0) return ORX_OK;
    ORX_ARG(opt->kind == ORX_SGD || opt->kind == ORX_ADAGRAD, "csr_apply: SGD / Adagrad (Adam: orx_csr_adam_apply)");
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    CsrArgs a;
    if (int rc = csr_args(ctx, t, sorted, n, grads, g_stride, &a)) return rc;
    a.lr = opt->lr; a.skip_single = skip_single ? 1 : 0;
    if (opt->kind == ORX_ADAGRAD) {
        OptSlots st;
        if (int rc = orx_opt_slots(opt, t, &st)) return rc;
        a.A = st.s0; a.eps = opt->p1;
        return launch_csr<CSR_ADAGRAD>(ctx, a);
    }
    return launch_csr<CSR_SGD>(ctx, a);
}

int orx_csr_apply_split(orx_ctx* ctx, orx_opt* opt, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride,
                        bool skip_single, CsrFinish* finish) {
    memset(finish, 0, sizeof(*finish));
    if (n == 0) return ORX_OK;
    ORX_ARG(opt->kind == ORX_SGD || opt->kind == ORX_ADAGRAD, "csr_apply: SGD / Adagrad (Adam: orx_csr_adam_apply)");
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    CsrArgs a;
    if (int rc = csr_args(ctx, t, sorted, n, grads, g_stride, &a)) return rc;
    a.lr = opt->lr; a.skip_single = skip_single ? 1 : 0;
    int rc;
    if (opt->kind == ORX_ADAGRAD) {
        OptSlots st;
        if (int rc2 = orx_opt_slots(opt, t, &st)) return rc2;
        a.A = st.s0; a.eps = opt->p1;
        rc = launch_csr<CSR_ADAGRAD>(ctx, a, false);
    } else rc = launch_csr<CSR_SGD>(ctx, a, false);
    if (rc != ORX_OK) return rc;
    finish->a = a; finish->mode = opt->kind == ORX_ADAGRAD ? CSR_ADAGRAD : CSR_SGD;
    finish->blocks = (int)(((n + 63) / 64 + 3) / 4);
    return ORX_OK;
}

// gsum[row] += sum of the row's gradient rows (the whole-table-sweep form of TF-2.0 Adam follows with its sweep)
int orx_csr_accum(orx_ctx* ctx, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride) {
    if (n == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    CsrArgs a;
    if (int rc = csr_args(ctx, t, sorted, n, grads, g_stride, &a)) return rc;
    if (int rc = orx_table_scratch(t)) return rc;
    a.G = t->gsum;
    return launch_csr<CSR_ACCUM>(ctx, a);
}

// lazy TF-2.0 Adam on the sorted list: touch (replay the distinct rows to step T) / step T
int orx_csr_adam(orx_ctx* ctx, bool step, const AdamRowsArgs& r, orx_table* t, const uint2* sorted, int64_t n) {
    if (n == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    CsrArgs a;
    if (int rc = csr_args(ctx, t, sorted, n, r.grads, r.g_stride, &a)) return rc;
    a.A = r.M; a.V = r.V; a.last = r.last; a.lrt = r.lrt; a.lr_T = r.lr_T; a.b1 = r.b1; a.b2 = r.b2; a.eps = r.eps; a.T = r.T; a.newton = r.newton; a.cf = r.cf;
    if (step) return launch_csr<CSR_ADAM>(ctx, a);
    ORX_LAUNCH(ctx, csr_touch_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
