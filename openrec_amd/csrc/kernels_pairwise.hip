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
// Design (DESIGN.md has the measurements behind each choice):
//   * The step is bound by the number of random memory transactions, not by
//     flops: 3 row reads + 3 row writes per triplet with no reuse.  A row of D
//     fp32 is owned by LPR = D/4 adjacent lanes (one float4 each): a D=64 row is
//     one coalesced 256-B segment, a wavefront carries 64/LPR triplets, and the
//     dot products reduce inside the lane group with DPP (quad_perm,
//     row_half_mirror, row_mirror) -- no LDS, no barriers in the hot kernel.
//   * TF takes every gradient of a step on the PRE-step tables.  A one-pass
//     in-place kernel breaks that only for rows referenced more than once in the
//     batch.  Per-row metadata in HBM would cost one extra random transaction
//     per reference (measured: about as expensive as a row access), so duplicates
//     are found WITHOUT touching HBM tables:
//       dedup_kernel    : row-range buckets of 425984 rows; each workgroup keeps
//                         three LDS bitmaps ("seen", "seen twice", "seen three
//                         times", 156 KiB) for its range while streaming the
//                         L2-resident id arrays: exact, no hashing, no overflow.
//                         It runs once for ALL K steps of a call, before the
//                         first step.  Output per step: the ids rewritten with
//                         the duplicate flag in bit 31 and the reference's role
//                         (first / second / any of >= 3) in bits 30:29, and the
//                         list of duplicated rows.
//       fused_kernel    : unique rows are read, scored and updated in place
//                         (exact: nobody else reads or writes them).  A
//                         duplicated row is left untouched; a reference to it
//                         deposits its gradient in scratch: rows referenced
//                         exactly twice (the vast majority) get ONE PLAIN STORE
//                         per reference into scratch row 1 / scratch row 2 (no
//                         atomics, bitwise reproducible), rows referenced >= 3
//                         times use fire-and-forget fp32 atomics into scratch row 1.
//       dup_apply_kernel: for every duplicated row of the step, apply the
//                         optimizer rule once with the summed gradient (TF's
//                         dedup-sum semantics for Adagrad; identical for SGD)
//                         and re-zero the scratch rows.
//     The scratch tables are all-zero between steps.
#include "orx_internal.h"

#include "orx_device.h"

// ------------------------------------------------------------ dedup kernel ---
// One workgroup per (step, table, 425984-row range).  Exact duplicate detection
// on the id arrays alone: three LDS bitmaps over the rows of the range; the ids of
// the step are streamed twice from L2 (mark, then emit).  No HBM table is touched.
constexpr int DD_WORDS = 13312;                 // 32-bit words per bitmap
constexpr int DD_ROWS = DD_WORDS * 32;          // rows covered by one workgroup (425984)
constexpr int DD_THREADS = 1024;
constexpr int DD_LDS_BYTES = 3 * DD_WORDS * 4;  // three bitmaps: seen / seen twice / seen three times (156 KiB of the 160 KiB LDS)

// visit every reference j in [0, n) of a table's id stream (segment A of length nA, then
// segment B); int4 loads when both segments are 16-byte aligned multiples of 4
// Column window (colF > 0): the id stream is a [n / colF][colF] matrix and only columns c0 .. c0+nc-1 can hold rows
// of this workgroup's range (DLRM: column f of the combined-table ids only holds rows of table f) -- the others
// are not read at all.
template <class F>
__device__ __forceinline__ void for_each_ref(const int32_t* idsA, int64_t nA, const int32_t* idsB, int64_t n, bool vec, F f,
                                             int colF = 0, int c0 = 0, int nc = 0) {
    if (colF > 0) {
        const int64_t m = (n / colF) * nc;
        for (int64_t t = threadIdx.x; t < m; t += DD_THREADS) {
            const int64_t r = t / nc;
            const int64_t j = r * colF + c0 + (t - r * nc);
            f(j, idsA[j]);
        }
        return;
    }
    if (vec) {
        // four independent 16-byte loads in flight per thread: the scan is latency-bound otherwise
        const int64_t n4 = n >> 2, nA4 = nA >> 2;
        for (int64_t q0 = threadIdx.x; q0 < n4; q0 += 4 * DD_THREADS) {
            int4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t q = q0 + (int64_t)k * DD_THREADS;
                if (q < n4) v[k] = q < nA4 ? reinterpret_cast<const int4*>(idsA)[q] : reinterpret_cast<const int4*>(idsB)[q - nA4];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t q = q0 + (int64_t)k * DD_THREADS;
                if (q < n4) { f(4 * q + 0, v[k].x); f(4 * q + 1, v[k].y); f(4 * q + 2, v[k].z); f(4 * q + 3, v[k].w); }
            }
        }
    } else {
        for (int64_t j = threadIdx.x; j < n; j += DD_THREADS) f(j, j < nA ? idsA[j] : idsB[j - nA]);
    }
}

// exclusive prefix sum of one int per thread over the DD_THREADS-thread workgroup; `total` = sum
__device__ __forceinline__ int block_scan_excl(int v, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < DD_THREADS / 64; ++k) {
        const int t = wave_tot[k];
        if (k < wave) before += t;
        all += t;
    }
    __syncthreads();            // wave_tot may be reused
    total = all;
    return before + incl - v;
}

__device__ __forceinline__ unsigned int lds_peek(const unsigned int* p) { return *reinterpret_cast<const volatile unsigned int*>(p); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- staging ------------------------------------------------------------------------------
// Rows referenced >= 3 times in a step ("tri" rows: hot items of a skewed id distribution) do not
// take atomics: every such reference gets a private slot in a staging buffer, the slots of a row
// are contiguous, and whoever applies the row sums its segment.  The plan is made here, on the
// id arrays alone:
//   dense numbering of the bucket's tri rows (popcount prefix over the tri bitmap, kept where the
//   "seen" bitmap lived), a per-step allocator hands the bucket a range of dense numbers;
//   in pass 2 every tri reference takes rank = atomicAdd(count[dense], 1) and records
//   (dense, rank); after the pass the counts are final: a second scan gives every row its
//   segment start.  A reference's slot = segstart[dense] + rank (computed by the fused kernel).
// The duplicate list carries (segment start, count) per row.  A segment longer than ORX_SEG_DIRECT
// is reduced by a tree of 64-to-1 partial sums before the apply (hot_reduce_kernel, up to three
// levels, plain stores only: same-address fp32 atomics cost ~100 ns each on this part, so even
// 200 of them on a hot row would dominate the step); its list entry then points at the last
// level's partial sums (count stored negative).
__global__ __launch_bounds__(DD_THREADS) void dedup_kernel(DedupArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int dd_lds[];
    unsigned int* seen = dd_lds;
    unsigned int* dup = dd_lds + DD_WORDS;
    unsigned int* tri = dd_lds + 2 * DD_WORDS;
    __shared__ int list_base;
    __shared__ int list_cnt;
    __shared__ int wave_tot[DD_THREADS / 64];
    __shared__ int sh_dense, sh_seg, sh_late;
    const int per_step = a.nbu + a.nbi;
    const int64_t s = blockIdx.x / per_step;
    int bk = blockIdx.x % per_step;
    const bool is_user = bk < a.nbu;
    if (!is_user) bk -= a.nbu;
    const int64_t r0 = (int64_t)bk * DD_ROWS;
    const int64_t rows = is_user ? a.NU : a.NI;
    const int64_t nA = is_user ? a.nU : a.nP;                // length of the first id segment
    const int32_t* idsA = (is_user ? a.uid : a.pid) + s * a.id_stride;
    const int32_t* idsB = a.nid + s * a.id_stride;
    const int64_t n = is_user ? a.nU : a.nP + a.nN;
    const int64_t ref0 = is_user ? 0 : a.nU;
    unsigned char* dflag = a.dflag ? a.dflag + s * a.flag_stride : nullptr;
    int32_t* ids_out = a.ids_out ? a.ids_out + s * a.flag_stride : nullptr;
    unsigned char* roles = a.roles ? a.roles + s * a.flag_stride : nullptr;
    // position of reference j in ids_out: [role][role_stride] (role 0 user, 1 pos item, 2 neg item)
    auto out_index = [&](int64_t j) -> int64_t {
        if (a.role_stride == 0) return ref0 + j;
        if (is_user) return j;
        return j < nA ? a.role_stride + j : 2 * a.role_stride + (j - nA);
    };
    const bool vec = ((nA & 3) == 0) && ((n & 3) == 0) && ((((uintptr_t)idsA) | ((uintptr_t)idsB)) & 15) == 0;

    for (int i = threadIdx.x; i < 3 * DD_WORDS; i += DD_THREADS) dd_lds[i] = 0u;
    if (threadIdx.x == 0) { list_cnt = 0; sh_late = 0; }
    __syncthreads();
    int late = 0;                                            // third-or-later references seen by this thread
    const int colF = (!is_user && a.col_win != nullptr) ? a.col_F : 0;
    const int2 cwin = colF ? a.col_win[bk] : make_int2(0, 0);   // (first column, number of columns) of this range
    for_each_ref(idsA, nA, idsB, n, vec, [&](int64_t j, int id) {
        const int64_t l = (int64_t)id - r0;
        if ((uint64_t)l < (uint64_t)DD_ROWS && id_ok(id, rows)) {
            const unsigned int bit = 1u << (l & 31);
            // The bitmaps only ever gain bits: a plain read that already shows the bit decides like the atomic would, and
            // only references that still see it clear go through the atomic (which arbitrates exactly one "first" and one
            // "second").  Thousands of references to ONE row (a 3-row DLRM table, the head of a Zipf distribution) would
            // otherwise serialize three LDS atomics each on the same word (Zipf(1.05) items at C2: 53.7 -> 50.7 us/step).
            const unsigned int cur = lds_peek(&seen[l >> 5]);
            const unsigned int old = (cur & bit) ? cur : atomicOr(&seen[l >> 5], bit);
            int role = 0;                                    // 0: first reference of the row, 1: second, 2: later
            if (old & bit) {
                const unsigned int cur2 = lds_peek(&dup[l >> 5]);
                const unsigned int old2 = (cur2 & bit) ? cur2 : atomicOr(&dup[l >> 5], bit);
                role = 1;
                if (old2 & bit) {
                    if (!(lds_peek(&tri[l >> 5]) & bit)) atomicOr(&tri[l >> 5], bit);
                    role = 2; ++late;
                }
            }
            if (roles) roles[out_index(j)] = (unsigned char)role;
            if (a.first_only) dflag[ref0 + j] = (old & bit) ? 1 : 0;
        }
    }, colF, cwin.x, cwin.y);
    if (a.refinfo != nullptr && late) atomicAdd(&sh_late, late);
    if (a.first_only) return;
    __syncthreads();
    // staging plan, part 1: dense numbers for the tri rows of this range
    int* prefix = reinterpret_cast<int*>(seen);             // "seen" is dead after pass 1
    // up to DD_WORDS/2 tri rows: 16-bit prefixes in the first half of that space, the rows' reference
    // counters in the second half (LDS atomics instead of global ones: a hot row takes thousands)
    unsigned short* prefix16 = reinterpret_cast<unsigned short*>(seen);
    int* lcnt = reinterpret_cast<int*>(seen) + DD_WORDS / 2;
    bool lds_cnt = false;
    int ntri = 0;                                            // tri rows of the range (workgroup-uniform)
    int dense0 = 0;
    int2* refinfo = nullptr; int* tricnt = nullptr; int* segstart = nullptr;
    // the plan pays for itself only where atomics would pile up: ranges with at least max(64, n/512)
    // third-or-later references (uniform ids over a large table stay below: ~110 per range at the
    // headline sizes); elsewhere role-2 references keep atomics, marked by refinfo = (-1, 0)
    const bool plan = a.refinfo != nullptr && sh_late >= (n / 512 > 64 ? n / 512 : 64);
    if (a.refinfo != nullptr) refinfo = a.refinfo + s * a.flag_stride;
    if (plan) {
        tricnt = a.tricnt + s * a.tri_stride; segstart = a.segstart + s * a.tri_stride;
        int mine3 = 0;
        for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) mine3 += __popc(tri[w]);
        int run = block_scan_excl(mine3, wave_tot, ntri);
        if (ntri) {
            if (threadIdx.x == 0) sh_dense = atomicAdd(a.alloc + 8 * s, ntri);
            lds_cnt = ntri <= DD_WORDS / 2;
            if (lds_cnt) {
                for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) { prefix16[w] = (unsigned short)run; run += __popc(tri[w]); }
                for (int i = threadIdx.x; i < ntri; i += DD_THREADS) lcnt[i] = 0;
            } else {
                for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) { prefix[w] = run; run += __popc(tri[w]); }
            }
            __syncthreads();
            dense0 = sh_dense;
        }
    }
    for_each_ref(idsA, nA, idsB, n, vec, [&](int64_t j, int id) {
        const int64_t l = (int64_t)id - r0;
        const bool ok = id_ok(id, rows);
        if ((uint64_t)l < (uint64_t)DD_ROWS && ok) {
            const unsigned int d = (dup[l >> 5] >> (l & 31)) & 1u;
            if (dflag) dflag[ref0 + j] = (unsigned char)d;
            if (ids_out) {
                uint32_t v = (uint32_t)id | (d << 31);
                if (roles && d) {
                    // rows with exactly two references: one plain store each into two scratch rows;
                    // three or more: every reference uses atomics (role 2)
                    const unsigned int tw = tri[l >> 5];
                    const unsigned int t3 = (tw >> (l & 31)) & 1u;
                    v |= (t3 ? 2u : (uint32_t)roles[out_index(j)]) << 29;
                    if (t3 && ntri) {
                        const int d = (lds_cnt ? (int)prefix16[l >> 5] : prefix[l >> 5]) + __popc(tw & ((1u << (l & 31)) - 1u));
                        refinfo[out_index(j)] = make_int2(dense0 + d, lds_cnt ? atomicAdd(lcnt + d, 1) : atomicAdd(tricnt + dense0 + d, 1));
                    } else if (t3 && refinfo != nullptr) {
                        refinfo[out_index(j)] = make_int2(-1, 0);
                    }
                }
                ids_out[out_index(j)] = (int32_t)v;
            }
        } else if (!ok && bk == 0 && ids_out) {
            ids_out[out_index(j)] = 0x7fffffff;              // out-of-range id: can never be a valid row
        }
    }, colF, cwin.x, cwin.y);
    if (a.dupbits != nullptr) {      // keep the "seen twice" bitmap of (step, range) for urgent_kernel
        unsigned int* out = a.dupbits + ((size_t)s * per_step + (is_user ? bk : a.nbu + bk)) * DD_WORDS;
        for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) out[w] = dup[w];
    }
    // staging plan, part 2: the counts are final -> segment start of every tri row.  A thread's tri rows
    // are numbered q = 0, 1, ... in (word, bit) order; count and segment start of the first TQ stay in
    // registers for the list emission below (one global round trip instead of three).
    constexpr int TQ = 6;
    int qcnt[TQ], qseg[TQ];
    // constant-index access only (a dynamically indexed local array would live in scratch memory)
    auto qput = [](int (&arr)[TQ], int q, int v) {
#pragma unroll
        for (int i = 0; i < TQ; ++i) if (q == i) arr[i] = v;
    };
    auto qget = [](const int (&arr)[TQ], int q) {
        int v = 0;
#pragma unroll
        for (int i = 0; i < TQ; ++i) if (q == i) v = arr[i];
        return v;
    };
    if (ntri) {
        // counts of this range are only ever touched by this workgroup (atomics and agent-scope loads at
        // the same L2): the barrier orders them, no fence (an agent-scope release writes back the whole L2)
        __syncthreads();
        int csum = 0, q = 0;
        for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) {
            const int c = __popc(tri[w]);
            const int d0 = lds_cnt ? (int)prefix16[w] : prefix[w];
            for (int k = 0; k < c; ++k, ++q) {
                const int v = lds_cnt ? lcnt[d0 + k] : ld_agent(tricnt + dense0 + d0 + k);
                qput(qcnt, q, v);
                csum += v;
            }
        }
        int total_refs;
        int run = block_scan_excl(csum, wave_tot, total_refs);
        if (threadIdx.x == 0) sh_seg = atomicAdd(a.alloc + 8 * s + 1, total_refs);
        __syncthreads();
        run += sh_seg;
        q = 0;
        for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) {
            const int c = __popc(tri[w]);
            const int d0 = lds_cnt ? (int)prefix16[w] : prefix[w];
            for (int k = 0; k < c; ++k, ++q) {
                segstart[dense0 + d0 + k] = run;
                qput(qseg, q, run);
                run += q < TQ ? qget(qcnt, q) : (lds_cnt ? lcnt[d0 + k] : ld_agent(tricnt + dense0 + d0 + k));
            }
        }
    }
    // append the duplicated rows of this range to the step's list
    int mine = 0;
    for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) mine += __popc(dup[w]);
    int off = 0;
    if (mine) off = atomicAdd(&list_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) list_base = list_cnt ? atomicAdd(a.dcount + s, list_cnt) : 0;
    __syncthreads();
    if (mine) {
        int64_t e = s * a.list_stride + list_base + off;
        const uint32_t tag = is_user ? 0u : 0x80000000u;
        int q = 0;
        for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) {
            unsigned int m = dup[w];
            const unsigned int tw = tri[w];
            while (m) {
                const int bpos = __ffs(m) - 1;
                m &= m - 1;
                const uint32_t ent = (uint32_t)(r0 + (int64_t)w * 32 + bpos) | tag;
                a.dlist[e] = ent;
                if (a.dcnt != nullptr) {
                    int c = 0, sg = 0;
                    if (ntri && ((tw >> bpos) & 1u)) {      // this thread wrote segstart[dense] itself (same word)
                        if (q < TQ) { c = qget(qcnt, q); sg = qget(qseg, q); }
                        else {
                            const int d = (lds_cnt ? (int)prefix16[w] : prefix[w]) + __popc(tw & ((1u << bpos) - 1u));
                            c = lds_cnt ? lcnt[d] : ld_agent(tricnt + dense0 + d);
                            sg = segstart[dense0 + d];
                        }
                        ++q;
                        if (c > ORX_SEG_DIRECT) {           // long segment: reduction tree
                            int4* items = a.items + s * a.item_stride;
                            int src = sg, len = c, level = 0;
                            do {
                                const int pieces = (len + ORX_PIECE - 1) / ORX_PIECE;
                                const int b0 = a.tree_off[level] + atomicAdd(a.alloc + 8 * s + 2 + level, pieces);
                                for (int k = 0; k < pieces; ++k) {
                                    const int rem = len - k * ORX_PIECE;
                                    items[b0 + k] = make_int4(src + k * ORX_PIECE, rem < ORX_PIECE ? rem : ORX_PIECE, b0 + k, 0);
                                }
                                src = b0; len = pieces; ++level;
                            } while (len > ORX_SEG_DIRECT && level < 3);
                            sg = src; c = -len;
                        }
                    }
                    a.dseg[e] = sg; a.dcnt[e] = c;
                }
                ++e;
            }
        }
    }
}

int orx_dedup_buckets(int64_t rows) { return (int)((rows + DD_ROWS - 1) / DD_ROWS); }
int64_t orx_dedup_range_rows() { return DD_ROWS; }
int orx_dedup_words(void) { return DD_WORDS; }

int orx_launch_dedup(orx_ctx* ctx, const DedupArgs& a, int64_t K) {
    ProfScope ps(ctx, ORX_K_DEDUP);
    const int64_t g = K * (a.nbu + a.nbi);
    if (g == 0) return ORX_OK;
    ORX_ARG(g < (1LL << 31), "dedup: grid too large (K=%lld, buckets=%d)", (long long)K, a.nbu + a.nbi);
    ORX_ONCE_PER_DEVICE(ctx, ORX_HIP(hipFuncSetAttribute((const void*)dedup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DD_LDS_BYTES)));
    ORX_LAUNCH(ctx, dedup_kernel, dim3((unsigned)g), dim3(DD_THREADS), DD_LDS_BYTES, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// A reference of step s whose row was duplicated in step s-1 is "urgent": the row's update of step
// s-1 is applied by the apply blocks of the SAME launch that processes step s (see fused_kernel),
// so this reference has to wait for the row's ready flag.  Marks such references with bit 28.
__global__ __launch_bounds__(DD_THREADS) void urgent_kernel(DedupArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int dd_lds[];
    const int per_step = a.nbu + a.nbi;
    const int64_t s = 1 + blockIdx.x / per_step;             // steps 1 .. K-1 of the chunk
    int bk = blockIdx.x % per_step;
    const unsigned int* prev = a.dupbits + ((size_t)(s - 1) * per_step + bk) * DD_WORDS;
    const bool is_user = bk < a.nbu;
    if (!is_user) bk -= a.nbu;
    const int64_t r0 = (int64_t)bk * DD_ROWS;
    const int64_t rows = is_user ? a.NU : a.NI;
    const int64_t nA = is_user ? a.nU : a.nP;
    const int32_t* idsA = (is_user ? a.uid : a.pid) + s * a.id_stride;
    const int32_t* idsB = a.nid + s * a.id_stride;
    const int64_t n = is_user ? a.nU : a.nP + a.nN;
    int32_t* ids_out = a.ids_out + s * a.flag_stride;
    const bool vec = ((nA & 3) == 0) && ((n & 3) == 0) && ((((uintptr_t)idsA) | ((uintptr_t)idsB)) & 15) == 0;
    for (int w = threadIdx.x; w < DD_WORDS; w += DD_THREADS) dd_lds[w] = prev[w];
    __syncthreads();
    for_each_ref(idsA, nA, idsB, n, vec, [&](int64_t j, int id) {
        const int64_t l = (int64_t)id - r0;
        if ((uint64_t)l < (uint64_t)DD_ROWS && id_ok(id, rows) && ((dd_lds[l >> 5] >> (l & 31)) & 1u)) {
            const int64_t idx = is_user ? j : (j < nA ? a.role_stride + j : 2 * a.role_stride + (j - nA));
            ids_out[idx] |= (1 << 28);
        }
    });
}

int orx_launch_urgent(orx_ctx* ctx, const DedupArgs& a, int64_t K) {
    ProfScope ps(ctx, ORX_K_DEDUP);
    const int64_t g = (K - 1) * (a.nbu + a.nbi);
    if (g <= 0) return ORX_OK;
    ORX_LAUNCH(ctx, urgent_kernel, dim3((unsigned)g), dim3(DD_THREADS), DD_WORDS * 4, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// -------------------------------------------------- loss partial reduction ---
// One block per step: sum nwaves x {loss, l2} fp32 partials in fp64.
__device__ __forceinline__ void loss_reduce_body(const ReduceArgs& a, int step) {
    __shared__ double sh[2][16];
    const float* part = a.partial + (size_t)step * a.nwaves * 2;
    double s0 = 0.0, s1 = 0.0;
    constexpr int UN = 8;                                    // 8 independent loads in flight per thread (the pass is latency-bound)
    for (int i0 = threadIdx.x; i0 < a.nwaves; i0 += UN * blockDim.x) {
        float2 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * blockDim.x;
            v[u] = i < a.nwaves ? *reinterpret_cast<const float2*>(part + 2 * i) : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        s1 += __shfl_xor(s1, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s0; sh[1][w] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { t0 += sh[0][k]; t1 += sh[1][k]; }
        a.out[2 * step] = t0;
        a.out[2 * step + 1] = t1;
    }
}

__global__ __launch_bounds__(1024) void loss_reduce_kernel(ReduceArgs a) { loss_reduce_body(a, (int)blockIdx.x); }

int orx_launch_loss_reduce(orx_ctx* ctx, const ReduceArgs& a, int64_t K) {
    ProfScope ps(ctx, ORX_K_REDUCE);
    ORX_LAUNCH(ctx, loss_reduce_kernel, dim3((unsigned)K), dim3(1024), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

#include "orx_apply_device.h"

// ------------------------------------------------------------ fused kernel ---
// LPR lanes own one row (D = 4*LPR).  MODE: see orx_internal.h.
// LONGGAP (lazy Adam): 0 = the merged replay loop, 1 = + the bounded per-row replay of rows far behind, 2 = the closed-form replay (orx_device.h AdamCF)
template <int LPR, int MODEL, int OPT, int MODE, bool CENSOR = false, bool STAGED = false, int LONGGAP = 0>
__global__ __launch_bounds__(256) void fused_kernel(PairArgs a) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    // (SGD only.  Adagrad: 111 VGPRs with the pairing tail against 84 without -- a wavefront of occupancy; bounded to 96 registers
    // (`__launch_bounds__(256, 5)`: no spills) the kernel with pairs still takes 51.5 us against 49.6 without and the step 62.1
    // against 57.7, K = 20, one box: profiles/r5_adagrad_pairing_ab.txt -- the pair tail reads and writes the accumulator row too)
    constexpr bool PAIRS = MODE == MODE_EXACT && OPT == ORX_SGD && TPW > 1;
    __shared__ f4 pair_xg[PAIRS ? 256 : 1];            // pairing: gradient exchange, one slot per lane
    __shared__ float pair_xb[PAIRS ? 256 / LPR : 1];   // ... and per lane group (item bias)
    __shared__ f4 pair_xw[PAIRS ? 256 : 1];            // the writer's copy of the shared row as read
    __shared__ float pair_xwb[PAIRS ? 256 / LPR : 1];
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int nab = MODE == MODE_EXACT ? a.n_apply_blocks : 0;
    if (MODE == MODE_EXACT && (int)blockIdx.x < nab) {          // apply role (block-uniform)
        inline_apply<LPR, OPT, CENSOR, STAGED>(a);
        return;
    }
    const int64_t wave_global = (int64_t)(blockIdx.x - nab) * 4 + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)(gridDim.x - nab) * 4 * TPW;
    float loss_acc = 0.0f, sq_acc = 0.0f;

    for (int64_t t = wave_global * TPW + grp; t < a.B; t += stride) {
        // ids as rewritten by the plan: bit 31 = "row is referenced more than once", bits 30:29 = role of this reference among the row's
        // references (0 / 1 = plain store into scratch row 1 / 2, 2 = atomics or staging slot, 3 = no store: pairing), bit 28 = urgent.
        // The flags stay IN the id words and are tested where they are needed (six flag registers fewer per lane).
        uint32_t uw, pw, nw;
        // pairing (kernels_plan.hip): position t processes the triplet the plan put there -- one 16-byte record: its three ids, the
        // pairing word (this triplet shares a row with another lane group of this wavefront) and where the triplet stood (t0: its
        // staging records are indexed by that)
        uint32_t pi = 0u;
        int64_t t0 = t;
        bool packed = false;
        if (PAIRS) { packed = a.ids4 != nullptr; }
        if (packed) {
            const int4 v = a.ids4[t];
            uw = (uint32_t)v.x; pw = (uint32_t)v.y; nw = (uint32_t)v.z;
            pi = (uint32_t)v.w & 0x3ffu; t0 = (int64_t)((uint32_t)v.w >> 10);
        } else {
            uw = (uint32_t)a.uid[t]; pw = (uint32_t)a.pid[t]; nw = (uint32_t)a.nid[t];
        }
        const uint32_t idmask = MODE == MODE_EXACT ? (a.role_bits ? 0x0fffffffu : 0x7fffffffu) : 0xffffffffu;
        const int u = (int)(uw & idmask), p = (int)(pw & idmask), n = (int)(nw & idmask);
#define du FLAG_DUP(uw)
#define dp FLAG_DUP(pw)
#define dn FLAG_DUP(nw)
#define ku FLAG_ROLE(uw)
#define kp FLAG_ROLE(pw)
#define kn FLAG_ROLE(nw)
#define FLAG_DUP(w) (MODE == MODE_ACCUM ? 1 : (MODE == MODE_EXACT ? (int)((w) >> 31) : 0))
#define FLAG_ROLE(w) ((MODE == MODE_EXACT && a.role_bits) ? (int)(((w) >> 29) & 3u) : 2)
        // bitwise &: all three id loads are issued together (a short-circuit && lets the compiler
        // sink the loads behind each other: three dependent round trips)
        if (!(id_ok(u, a.NU) & id_ok(p, a.NI) & id_ok(n, a.NI))) {
            if (sub == 0) *a.err = 1;       // the reference's CPU gather raises; the triplet is skipped
            if (PAIRS) {
                if (pi & ORX_PAIR_VALID) {  // (its partner must not add what an earlier iteration left in LDS)
                    f4 z; z.x = z.y = z.z = z.w = 0.0f;
                    pair_xg[threadIdx.x] = z;
                    if (sub == 0) pair_xb[threadIdx.x / LPR] = 0.0f;
                }
            }
            continue;
        }
        // staged references (role 2 with a staging plan): slot = segment start of the row + rank of the
        // reference, looked up only where such a reference deposits its gradient
        const int64_t Bp = a.pid - a.uid;           // the three id arrays of a step are Bp apart
        auto slot_of = [&](int64_t ref) -> int {
            if (!STAGED) return -1;
            const int2 ri = a.refinfo[ref];
            return ri.x < 0 ? -1 : a.segstart[ri.x] + ri.y;      // (-1, 0): the row's range made no plan -> atomics
        };
        if (MODE == MODE_EXACT && nab && a.role_bits && (((uw | pw | nw) >> 28) & 1u)) {      // (the marks are made before the host knows whether the launch applies)
            // a row of this triplet is being updated by an apply block of this launch
            if (sub == 0) {
                if ((uw >> 28) & 1u) wait_ready(a.readyU + u, a.epoch);
                if ((pw >> 28) & 1u) wait_ready(a.readyV + p, a.epoch);
                if ((nw >> 28) & 1u) wait_ready(a.readyV + n, a.epoch);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        float* Up = a.U + (size_t)u * D + 4 * sub;
        float* Pp = a.V + (size_t)p * D + 4 * sub;
        float* Np = a.V + (size_t)n * D + 4 * sub;
        f4 ru = *reinterpret_cast<const f4*>(Up);
        f4 rp = *reinterpret_cast<const f4*>(Pp);
        f4 rn = *reinterpret_cast<const f4*>(Np);
        float bp = a.b[p], bn = a.b[n];
        // lazy Adam: (w, m, v) of the three rows and two biases, replayed up to the step before this one -- the
        // forward then sees exactly what the whole-table sweeps of TF 2.0 would have left
        f4 mu, vu, mp, vp, mn, vn;
        float mbp = 0.f, vbp = 0.f, mbn = 0.f, vbn = 0.f;
        if (OPT == ORX_ADAM) {
            const int T1 = a.step_t - 1;
            mu = *reinterpret_cast<const f4*>(a.aU + (size_t)u * D + 4 * sub); vu = *reinterpret_cast<const f4*>(a.a2U + (size_t)u * D + 4 * sub);
            mp = *reinterpret_cast<const f4*>(a.aV + (size_t)p * D + 4 * sub); vp = *reinterpret_cast<const f4*>(a.a2V + (size_t)p * D + 4 * sub);
            mn = *reinterpret_cast<const f4*>(a.aV + (size_t)n * D + 4 * sub); vn = *reinterpret_cast<const f4*>(a.a2V + (size_t)n * D + 4 * sub);
            mbp = a.ab[p]; vbp = a.a2b[p]; mbn = a.ab[n]; vbn = a.a2b[n];
            // (the bias of an item shares the item row's stamp: the three tables are lazy together, api.hip)
            int lu = a.lastU[u], lp = a.lastV[p], ln = a.lastV[n];
            if (LONGGAP == 2) {
                const float4 Vt = a.lrv[T1];                 // (wave-uniform)
                AdamCF cf;
                if (lu < T1) { cf.setup(a.lrv, lu, T1, Vt, a.cf_lb1, a.cf_lb2); cf.row4(ru, mu, vu, a.eps, a.cf_delta); }
                if (lp < T1) { cf.setup(a.lrv, lp, T1, Vt, a.cf_lb1, a.cf_lb2); cf.row4(rp, mp, vp, a.eps, a.cf_delta); cf.elem(bp, mbp, vbp, a.eps, a.cf_delta); }
                if (ln < T1) { cf.setup(a.lrv, ln, T1, Vt, a.cf_lb1, a.cf_lb2); cf.row4(rn, mn, vn, a.eps, a.cf_delta); cf.elem(bn, mbn, vbn, a.eps, a.cf_delta); }
                lu = lp = ln = T1;
            }
            if (LONGGAP == 1) {      // large tables: rows that have waited very long take the bounded replay and leave the merged loop
                if (T1 - lu > ORX_ADAM_LONG_GAP) {
                    float z0 = 0.f, z1 = 0.f, z2 = 0.f;
                    adam_replay4_bounded(ru, mu, vu, z0, z1, z2, lu, T1, a.lrt, a.b1, a.b2, a.eps);
                    lu = T1;
                }
                if (T1 - lp > ORX_ADAM_LONG_GAP) { adam_replay4_bounded(rp, mp, vp, bp, mbp, vbp, lp, T1, a.lrt, a.b1, a.b2, a.eps); lp = T1; }
                if (T1 - ln > ORX_ADAM_LONG_GAP) { adam_replay4_bounded(rn, mn, vn, bn, mbn, vbn, ln, T1, a.lrt, a.b1, a.b2, a.eps); ln = T1; }
            }
            if (LONGGAP == 2) {}
            else if (a.newton) adam_catchup_triplet<true, LPR>(ru, mu, vu, lu, rp, mp, vp, lp, rn, mn, vn, ln, bp, mbp, vbp, bn, mbn, vbn, T1, a.lrt, a.b1, a.b2, a.eps);
            else adam_catchup_triplet<false, LPR>(ru, mu, vu, lu, rp, mp, vp, lp, rn, mn, vn, ln, bp, mbp, vbp, bn, mbn, vbn, T1, a.lrt, a.b1, a.b2, a.eps);
        }

        const float red = group_allreduce<LPR>(score_partial<MODEL>(ru, rp, rn));
        float term, g;
        score<MODEL>(red, bp, bn, a.invB, a.margin, term, g);
        sq_acc += dot4(ru, ru) + dot4(rp, rp) + dot4(rn, rn);
        if (sub == 0) loss_acc += term;
        if (MODE == MODE_LOSS) continue;

        f4 gu, gp, gn; float gbp, gbn;
        row_grads<MODEL>(ru, rp, rn, g, a.l2w, gu, gp, gn, gbp, gbn);

        // pairing: the two lane groups that share a row leave their gradient of it in LDS (one slot per lane; a wavefront's LDS
        // operations execute in order, so no barrier), the WRITER also the row and bias it read; for both the slot is then settled
        // (role 3: no store below).  After the other stores (pair_tail) the writer sums the two gradients -- TF sums the gradients
        // of duplicate indices before the sparse apply -- and updates the row as the unique row it has become.
        if (PAIRS) {
            if (pi & ORX_PAIR_VALID) {
                const int myslot = (pi >> 4) & 3;
                pair_xg[threadIdx.x] = myslot == 0 ? gu : (myslot == 1 ? gp : gn);
                if (pi & ORX_PAIR_WRITER) pair_xw[threadIdx.x] = myslot == 0 ? ru : (myslot == 1 ? rp : rn);
                if (sub == 0) {
                    pair_xb[threadIdx.x / LPR] = myslot == 1 ? gbp : gbn;
                    if (pi & ORX_PAIR_WRITER) pair_xwb[threadIdx.x / LPR] = myslot == 1 ? bp : bn;
                }
                if (myslot == 0) uw |= 0xe0000000u;         // duplicate flag + role 3
                else if (myslot == 1) pw |= 0xe0000000u;
                else nw |= 0xe0000000u;
            }
        }
        auto pair_tail = [&]() {
            if (!PAIRS) return;
            if (!(pi & ORX_PAIR_WRITER)) return;
            const int myslot = (pi >> 4) & 3, oslot = (pi >> 6) & 3;
            const int xsrc = (int)(threadIdx.x & ~63u) + (int)(pi & 15u) * LPR + sub;
            const int id = reinterpret_cast<const int*>(a.ids4 + t)[myslot] & 0x0fffffff;      // (the tail keeps nothing of the triplet alive but t and its pairing word)
            const size_t off = (size_t)id * D + 4 * sub;
            float* W = myslot == 0 ? a.U : a.V;
            float* A = myslot == 0 ? a.aU : a.aV;
            const f4 gs = pair_xg[threadIdx.x] + pair_xg[xsrc];
            const f4 w = pair_xw[threadIdx.x];
            if (CENSOR) {
                f4 wn = censor4<LPR>(opt_new4<OPT>(A + off, w, gs, a.lr, a.eps), a.min_norm);
                // a positive of one triplet and a negative of the other: censored once per id list (ucml.py:46-48)
                if (myslot != oslot) wn = censor4<LPR>(wn, a.min_norm);
                *reinterpret_cast<f4*>(W + off) = wn;
            } else {
                opt_apply4<OPT>(W + off, A + off, w, gs, a.lr, a.eps);
            }
            if (myslot != 0 && sub == 0)
                opt_apply1<OPT>(a.b + id, a.ab + id, pair_xwb[threadIdx.x / LPR], pair_xb[threadIdx.x / LPR] + pair_xb[xsrc / LPR], a.lr, a.eps);
        };

        if (OPT == ORX_ADAM) {
            // a row referenced once takes its step here (replayed state + gradient), a duplicated one deposits the gradient
            const float lrT = a.lrt[a.step_t];
            if (du == 0) {
                adam_elem4(ru, mu, vu, gu, lrT, a.b1, a.b2, a.eps);
                if (CENSOR) ru = censor4<LPR>(ru, a.min_norm);      // censor_vec fused into the write-back (see below)
                *reinterpret_cast<f4*>(Up) = ru; *reinterpret_cast<f4*>(a.aU + (size_t)u * D + 4 * sub) = mu;
                *reinterpret_cast<f4*>(a.a2U + (size_t)u * D + 4 * sub) = vu;
                if (sub == 0) a.lastU[u] = a.step_t;
            } else dup_store4s(a.gU, a.gU2, (size_t)u * D + 4 * sub, gu, ku, a.stage, ku == 2 ? slot_of(t0) : -1, D, sub);
            if (dp == 0) {
                adam_elem4(rp, mp, vp, gp, lrT, a.b1, a.b2, a.eps);
                if (CENSOR) rp = censor4<LPR>(rp, a.min_norm);
                *reinterpret_cast<f4*>(Pp) = rp; *reinterpret_cast<f4*>(a.aV + (size_t)p * D + 4 * sub) = mp;
                *reinterpret_cast<f4*>(a.a2V + (size_t)p * D + 4 * sub) = vp;
                if (sub == 0) {
                    adam_elem(bp, mbp, vbp, gbp, lrT, a.b1, a.b2, a.eps);
                    a.b[p] = bp; a.ab[p] = mbp; a.a2b[p] = vbp; a.lastV[p] = a.step_t; a.lastb[p] = a.step_t;
                }
            } else {
                const int sp = kp == 2 ? slot_of(Bp + t0) : -1;
                dup_store4s(a.gV, a.gV2, (size_t)p * D + 4 * sub, gp, kp, a.stage, sp, D, sub);
                if (sub == 0) { dup_store1s(a.gb, a.gb2, p, gbp, kp, a.stageb, sp); if (CENSOR) a.sideV[2 * (size_t)p] = a.epoch; }
            }
            if (dn == 0) {
                adam_elem4(rn, mn, vn, gn, lrT, a.b1, a.b2, a.eps);
                if (CENSOR) rn = censor4<LPR>(rn, a.min_norm);
                *reinterpret_cast<f4*>(Np) = rn; *reinterpret_cast<f4*>(a.aV + (size_t)n * D + 4 * sub) = mn;
                *reinterpret_cast<f4*>(a.a2V + (size_t)n * D + 4 * sub) = vn;
                if (sub == 0) {
                    adam_elem(bn, mbn, vbn, gbn, lrT, a.b1, a.b2, a.eps);
                    a.b[n] = bn; a.ab[n] = mbn; a.a2b[n] = vbn; a.lastV[n] = a.step_t; a.lastb[n] = a.step_t;
                }
            } else {
                const int sn = kn == 2 ? slot_of(2 * Bp + t0) : -1;
                dup_store4s(a.gV, a.gV2, (size_t)n * D + 4 * sub, gn, kn, a.stage, sn, D, sub);
                if (sub == 0) { dup_store1s(a.gb, a.gb2, n, gbn, kn, a.stageb, sn); if (CENSOR) a.sideV[2 * (size_t)n + 1] = a.epoch; }
            }
            continue;
        }
        // unique row: in place.  duplicated row: gradient into gsum, row untouched.
        if (CENSOR) {
            // censor_vec fused into the write-back: a row referenced once is censored once, here;
            // duplicated rows are censored by the kernel that applies their summed gradient
            f4 wu = ru, wp = rp, wn = rn;
            if (du == 0) wu = opt_new4<OPT>(a.aU + (size_t)u * D + 4 * sub, ru, gu, a.lr, a.eps);
            else dup_store4s(a.gU, a.gU2, (size_t)u * D + 4 * sub, gu, ku, a.stage, ku == 2 ? slot_of(t0) : -1, D, sub);
            if (dp == 0) {
                wp = opt_new4<OPT>(a.aV + (size_t)p * D + 4 * sub, rp, gp, a.lr, a.eps);
                if (sub == 0) opt_apply1<OPT>(a.b + p, a.ab + p, bp, gbp, a.lr, a.eps);
            } else {
                const int sp = kp == 2 ? slot_of(Bp + t0) : -1;
                dup_store4s(a.gV, a.gV2, (size_t)p * D + 4 * sub, gp, kp, a.stage, sp, D, sub);
                if (sub == 0) { dup_store1s(a.gb, a.gb2, p, gbp, kp, a.stageb, sp); if (kp != 3) a.sideV[2 * (size_t)p] = a.epoch; }
            }
            if (dn == 0) {
                wn = opt_new4<OPT>(a.aV + (size_t)n * D + 4 * sub, rn, gn, a.lr, a.eps);
                if (sub == 0) opt_apply1<OPT>(a.b + n, a.ab + n, bn, gbn, a.lr, a.eps);
            } else {
                const int sn = kn == 2 ? slot_of(2 * Bp + t0) : -1;
                dup_store4s(a.gV, a.gV2, (size_t)n * D + 4 * sub, gn, kn, a.stage, sn, D, sub);
                if (sub == 0) { dup_store1s(a.gb, a.gb2, n, gbn, kn, a.stageb, sn); if (kn != 3) a.sideV[2 * (size_t)n + 1] = a.epoch; }
            }
            wu = censor4<LPR>(wu, a.min_norm); wp = censor4<LPR>(wp, a.min_norm); wn = censor4<LPR>(wn, a.min_norm);
            if (du == 0) *reinterpret_cast<f4*>(Up) = wu;
            if (dp == 0) *reinterpret_cast<f4*>(Pp) = wp;
            if (dn == 0) *reinterpret_cast<f4*>(Np) = wn;
            pair_tail();
            continue;
        }
        if (du == 0) opt_apply4<OPT>(Up, a.aU + (size_t)u * D + 4 * sub, ru, gu, a.lr, a.eps);
        else dup_store4s(a.gU, a.gU2, (size_t)u * D + 4 * sub, gu, ku, a.stage, ku == 2 ? slot_of(t0) : -1, D, sub);
        if (dp == 0) {
            opt_apply4<OPT>(Pp, a.aV + (size_t)p * D + 4 * sub, rp, gp, a.lr, a.eps);
            if (sub == 0) opt_apply1<OPT>(a.b + p, a.ab + p, bp, gbp, a.lr, a.eps);
        } else {
            const int sp = kp == 2 ? slot_of(Bp + t0) : -1;
            dup_store4s(a.gV, a.gV2, (size_t)p * D + 4 * sub, gp, kp, a.stage, sp, D, sub);
            if (sub == 0) dup_store1s(a.gb, a.gb2, p, gbp, kp, a.stageb, sp);
        }
        if (dn == 0) {
            opt_apply4<OPT>(Np, a.aV + (size_t)n * D + 4 * sub, rn, gn, a.lr, a.eps);
            if (sub == 0) opt_apply1<OPT>(a.b + n, a.ab + n, bn, gbn, a.lr, a.eps);
        } else {
            const int sn = kn == 2 ? slot_of(2 * Bp + t0) : -1;
            dup_store4s(a.gV, a.gV2, (size_t)n * D + 4 * sub, gn, kn, a.stage, sn, D, sub);
            if (sub == 0) dup_store1s(a.gb, a.gb2, n, gbn, kn, a.stageb, sn);
        }
        pair_tail();
    }
#undef du
#undef dp
#undef dn
#undef ku
#undef kp
#undef kn
#undef FLAG_DUP
#undef FLAG_ROLE
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        float2 v; v.x = ls; v.y = 0.5f * sq;
        *reinterpret_cast<float2*>(a.partial + 2 * wave_global) = v;
    }
}

// dup_apply: one group of LPR lanes per duplicated row of the step.
template <int LPR, int OPT>
__device__ __forceinline__ void dup_apply_body(const PairArgs& a, int block, int nblocks) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int n = *a.dcount;
    const int wpb = (int)(blockDim.x >> 6);             // wavefronts per block (256 or 1024 threads)
    const int64_t stride = (int64_t)nblocks * wpb * TPW;
    for (int64_t e = ((int64_t)block * wpb + (threadIdx.x >> 6)) * TPW + grp; e < n; e += stride) {
        const uint32_t ent = a.dlist[e];
        if (ent == ORX_DLIST_DEAD) continue;            // (the row was paired after all: updated in place by the step's launch)
        const bool item = (ent >> 31) != 0;
        const size_t row = ent & 0x7fffffffu;
        float* W = item ? a.V : a.U;
        float* G = item ? a.gV : a.gU;
        float* A = item ? a.aV : a.aU;
        float* G2 = item ? a.gV2 : a.gU2;
        float* gp = G + row * D + 4 * sub;
        float* wp = W + row * D + 4 * sub;
        // staged row: scnt > 0: sum its staging segment here; scnt < 0: hot_reduce_kernel has reduced the
        // (long) segment to -scnt partial sums; scnt == 0: two-reference row, scratch rows
        const int scnt = a.dcnt != nullptr ? a.dcnt[e] : 0;
        const int sseg = scnt != 0 ? a.dseg[e] : 0;
        const f4 w = *reinterpret_cast<const f4*>(wp);
        f4 z; z.x = z.y = z.z = z.w = 0.0f;
        f4 g;
        if (scnt > 0) {
            g = segment_sum4<D>(a.stage, sseg, scnt, sub);
        } else if (scnt < 0) {
            g = segment_sum4<D>(a.part, sseg, -scnt, sub);
        } else {
            g = *reinterpret_cast<const f4*>(gp);
            *reinterpret_cast<f4*>(gp) = z;
            if (G2 != nullptr) {
                g = g + *reinterpret_cast<const f4*>(G2 + row * D + 4 * sub);
                *reinterpret_cast<f4*>(G2 + row * D + 4 * sub) = z;
            }
        }
        if (OPT == ORX_ADAM) {
            float* A2 = item ? a.a2V : a.a2U;
            int* L = item ? a.lastV : a.lastU;
            f4 wn = w;
            f4 mm = *reinterpret_cast<const f4*>(A + row * D + 4 * sub), vv = *reinterpret_cast<const f4*>(A2 + row * D + 4 * sub);
            adam_catchup4(wn, mm, vv, L[row], a.step_t - 1, a.lrt, a.b1, a.b2, a.eps);
            adam_elem4(wn, mm, vv, g, a.lrt[a.step_t], a.b1, a.b2, a.eps);
            if (a.censor) {
                wn = censor4<LPR>(wn, a.min_norm);
                if (item && censored_twice(a, row, a.epoch)) wn = censor4<LPR>(wn, a.min_norm);
            }
            *reinterpret_cast<f4*>(wp) = wn; *reinterpret_cast<f4*>(A + row * D + 4 * sub) = mm; *reinterpret_cast<f4*>(A2 + row * D + 4 * sub) = vv;
            if (sub == 0) L[row] = a.step_t;
        } else if (a.censor) {
            f4 wn = censor4<LPR>(opt_new4<OPT>(A + row * D + 4 * sub, w, g, a.lr, a.eps), a.min_norm);
            if (item && censored_twice(a, row, a.epoch)) wn = censor4<LPR>(wn, a.min_norm);
            *reinterpret_cast<f4*>(wp) = wn;
        } else {
            opt_apply4<OPT>(wp, A + row * D + 4 * sub, w, g, a.lr, a.eps);
        }
        float gbs = 0.0f;
        if (scnt != 0 && item && a.b != nullptr)
            gbs = scnt > 0 ? segment_sum1<LPR>(a.stageb, sseg, scnt, sub) : segment_sum1<LPR>(a.partb, sseg, -scnt, sub);
        if (item && a.b != nullptr && sub == 0) {
            float gb = gbs;
            if (scnt == 0) {
                gb = a.gb[row];
                a.gb[row] = 0.0f;
                if (a.gb2 != nullptr) { gb += a.gb2[row]; a.gb2[row] = 0.0f; }
            }
            if (OPT == ORX_ADAM) {
                float bw = a.b[row], bm = a.ab[row], bv = a.a2b[row];
                adam_catchup1(bw, bm, bv, a.lastb[row], a.step_t - 1, a.lrt, a.b1, a.b2, a.eps);
                adam_elem(bw, bm, bv, gb, a.lrt[a.step_t], a.b1, a.b2, a.eps);
                a.b[row] = bw; a.ab[row] = bm; a.a2b[row] = bv; a.lastb[row] = a.step_t;
            } else {
                opt_apply1<OPT>(a.b + row, a.ab + row, a.b[row], gb, a.lr, a.eps);
            }
        }
    }
}

template <int LPR, int OPT>
__global__ __launch_bounds__(256) void dup_apply_kernel(PairArgs a) { dup_apply_body<LPR, OPT>(a, (int)blockIdx.x, (int)gridDim.x); }

// The tail of a K-step call in ONE launch: the duplicated rows of the last step (what dup_apply_kernel does) and, in the first
// `nred` blocks, the per-step sums of the loss partials (what loss_reduce_kernel does) -- two launches and the gap between them
// fewer at the end of every call.
template <int LPR, int OPT>
__global__ __launch_bounds__(1024) void tail_kernel(PairArgs a, ReduceArgs r, int nred) {
    if ((int)blockIdx.x < nred) { loss_reduce_body(r, (int)blockIdx.x); return; }
    dup_apply_body<LPR, OPT>(a, (int)blockIdx.x - nred, (int)gridDim.x - nred);
}

// hot_reduce: one level of the reduction tree over long staging segments.  One wavefront per work
// item (src, len <= ORX_PIECE, dst): its 64/LPR lane groups each sum every (64/LPR)-th vector, the
// groups combine by cross-lane shuffles and group 0 stores the partial sum.  Level 1 reads the
// staged gradients, levels 2 and 3 the partial sums of the level below.  No atomics.
template <int LPR>
__global__ __launch_bounds__(256) void hot_reduce_kernel(PairArgs a, int level) {
    constexpr int TPW = 64 / LPR;
    constexpr int D = 4 * LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int n = a.nitems[level];
    const int4* items = a.items + a.tree_off[level];
    const float* src = level == 0 ? a.stage : a.part;
    const float* srcb = level == 0 ? a.stageb : a.partb;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < n; c += stride) {
        const int4 it = items[c];
        const float* p = src + (size_t)it.x * D + 4 * sub;
        f4 s0, s1; s0.x = s0.y = s0.z = s0.w = 0.0f; s1 = s0;
        int k = grp;
        for (; k + 3 * TPW < it.y; k += 4 * TPW) {
            const f4 a0 = *reinterpret_cast<const f4*>(p + (size_t)(k + 0 * TPW) * D);
            const f4 a1 = *reinterpret_cast<const f4*>(p + (size_t)(k + 1 * TPW) * D);
            const f4 a2 = *reinterpret_cast<const f4*>(p + (size_t)(k + 2 * TPW) * D);
            const f4 a3 = *reinterpret_cast<const f4*>(p + (size_t)(k + 3 * TPW) * D);
            s0 = s0 + (a0 + a1); s1 = s1 + (a2 + a3);
        }
        for (; k < it.y; k += TPW) s0 = s0 + *reinterpret_cast<const f4*>(p + (size_t)k * D);
        f4 g = s0 + s1;
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            g.x += __shfl_xor(g.x, off); g.y += __shfl_xor(g.y, off); g.z += __shfl_xor(g.z, off); g.w += __shfl_xor(g.w, off);
        }
        if (grp == 0) *reinterpret_cast<f4*>(a.part + (size_t)it.z * D + 4 * sub) = g;
        if (a.b != nullptr) {
            const float gb = wave_sum(lane < it.y ? srcb[it.x + lane] : 0.0f);
            if (lane == 0) a.partb[it.z] = gb;
        }
    }
}

template <int OPT>
__global__ __launch_bounds__(256) void dup_apply_generic_kernel(PairArgs a) {
    const int lane = threadIdx.x & 63;
    const int D = a.D;
    const int n = *a.dcount;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += stride) {
        const uint32_t ent = a.dlist[e];
        const bool item = (ent >> 31) != 0;
        const size_t row = ent & 0x7fffffffu;
        float* W = item ? a.V : a.U;
        float* G = item ? a.gV : a.gU;
        float* A = item ? a.aV : a.aU;
        float* G2 = item ? a.gV2 : a.gU2;
        for (int k = lane; k < D; k += 64) {
            const size_t i = row * D + k;
            float g = G[i];
            G[i] = 0.0f;
            if (G2 != nullptr) { g += G2[i]; G2[i] = 0.0f; }
            opt_apply1<OPT>(W + i, A + i, W[i], g, a.lr, a.eps);
        }
        if (item && a.b != nullptr && lane == 0) {
            float gb = a.gb[row];
            a.gb[row] = 0.0f;
            if (a.gb2 != nullptr) { gb += a.gb2[row]; a.gb2[row] = 0.0f; }
            opt_apply1<OPT>(a.b + row, a.ab + row, a.b[row], gb, a.lr, a.eps);
        }
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
        int u = a.uid[t], p = a.pid[t], n = a.nid[t];
        int du = 0, dp = 0, dn = 0;
        int ku = 2, kp = 2, kn = 2;     // duplicate role: 0 / 1 = plain store into scratch row 1 / 2, 2 = atomics
        if (MODE == MODE_EXACT) {       // ids rewritten by dedup_kernel: bit 31 = "row is referenced more than once"
            du = (uint32_t)u >> 31; dp = (uint32_t)p >> 31; dn = (uint32_t)n >> 31;
            if (a.role_bits) {          // bits 30:29 = role of this reference among the row's references
                ku = ((uint32_t)u >> 29) & 3; kp = ((uint32_t)p >> 29) & 3; kn = ((uint32_t)n >> 29) & 3;
                u &= 0x1fffffff; p &= 0x1fffffff; n &= 0x1fffffff;
            } else {
                u &= 0x7fffffff; p &= 0x7fffffff; n &= 0x7fffffff;
            }
        }
        if (MODE == MODE_ACCUM) { du = dp = dn = 1; }
        // bitwise &: all three id loads are issued together (a short-circuit && lets the compiler
        // sink the loads behind each other: three dependent round trips)
        if (!(id_ok(u, a.NU) & id_ok(p, a.NI) & id_ok(n, a.NI))) {
            if (lane == 0) *a.err = 1;
            continue;
        }
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
        for (int e = lane; e < D; e += 64) {
            const float x = Ur[e], y = Pr[e], z = Nr[e];
            float gu, gp, gn;
            if (MODEL == ORX_BPR) {
                gu = g * (y - z) + a.l2w * x; gp = g * x + a.l2w * y; gn = -g * x + a.l2w * z;
            } else {
                const float a2 = 2.0f * g;
                gu = -a2 * (y - z) + a.l2w * x; gp = -a2 * (x - y) + a.l2w * y; gn = a2 * (x - z) + a.l2w * z;
            }
            if (du == 0) opt_apply1<OPT>(Ur + e, a.aU + (size_t)u * D + e, x, gu, a.lr, a.eps);
            else dup_store1(a.gU, a.gU2, (size_t)u * D + e, gu, ku);
            if (dp == 0) opt_apply1<OPT>(Pr + e, a.aV + (size_t)p * D + e, y, gp, a.lr, a.eps);
            else dup_store1(a.gV, a.gV2, (size_t)p * D + e, gp, kp);
            if (dn == 0) opt_apply1<OPT>(Nr + e, a.aV + (size_t)n * D + e, z, gn, a.lr, a.eps);
            else dup_store1(a.gV, a.gV2, (size_t)n * D + e, gn, kn);
        }
        const float gbp = MODEL == ORX_BPR ? g : -g, gbn = -gbp;
        if (lane == 0) {
            if (dp == 0) opt_apply1<OPT>(a.b + p, a.ab + p, bp, gbp, a.lr, a.eps);
            else dup_store1(a.gb, a.gb2, p, gbp, kp);
            if (dn == 0) opt_apply1<OPT>(a.b + n, a.ab + n, bn, gbn, a.lr, a.eps);
            else dup_store1(a.gb, a.gb2, n, gbn, kn);
        }
    }
    const float ls = wave_sum(loss_acc);
    const float sq = wave_sum(sq_acc);
    if (lane == 0) {
        float2 v; v.x = ls; v.y = 0.5f * sq;
        *reinterpret_cast<float2*>(a.partial + 2 * wave_global) = v;
    }
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

template <int LPR, int MODEL, int OPT>
static void launch_fused_mode(int mode, dim3 g, orx_ctx* s, const PairArgs& a) {
    switch (mode) {
        case MODE_EXACT:
            if (a.censor && a.stage) ORX_LAUNCH(s, (fused_kernel<LPR, MODEL, OPT, MODE_EXACT, true, true>), g, dim3(256), 0, a);
            else if (a.censor) ORX_LAUNCH(s, (fused_kernel<LPR, MODEL, OPT, MODE_EXACT, true, false>), g, dim3(256), 0, a);
            else if (a.stage) ORX_LAUNCH(s, (fused_kernel<LPR, MODEL, OPT, MODE_EXACT, false, true>), g, dim3(256), 0, a);
            else ORX_LAUNCH(s, (fused_kernel<LPR, MODEL, OPT, MODE_EXACT>), g, dim3(256), 0, a);
            break;
        case MODE_HOGWILD: ORX_LAUNCH(s, (fused_kernel<LPR, MODEL, OPT, MODE_HOGWILD>), g, dim3(256), 0, a); break;
        case MODE_ACCUM: ORX_LAUNCH(s, (fused_kernel<LPR, MODEL, ORX_SGD, MODE_ACCUM>), g, dim3(256), 0, a); break;
        default: ORX_LAUNCH(s, (fused_kernel<LPR, MODEL, ORX_SGD, MODE_LOSS>), g, dim3(256), 0, a); break;
    }
}

template <int MODEL, int OPT>
static void launch_generic_mode(int mode, dim3 g, orx_ctx* s, const PairArgs& a) {
    switch (mode) {
        case MODE_EXACT: ORX_LAUNCH(s, (fused_generic_kernel<MODEL, OPT, MODE_EXACT>), g, dim3(256), 0, a); break;
        case MODE_HOGWILD: ORX_LAUNCH(s, (fused_generic_kernel<MODEL, OPT, MODE_HOGWILD>), g, dim3(256), 0, a); break;
        case MODE_ACCUM: ORX_LAUNCH(s, (fused_generic_kernel<MODEL, ORX_SGD, MODE_ACCUM>), g, dim3(256), 0, a); break;
        default: ORX_LAUNCH(s, (fused_generic_kernel<MODEL, ORX_SGD, MODE_LOSS>), g, dim3(256), 0, a); break;
    }
}

template <int MODEL, int OPT>
static void launch_fused_lpr(int lpr, int mode, dim3 g, orx_ctx* s, const PairArgs& a) {
    switch (lpr) {
        case 4: launch_fused_mode<4, MODEL, OPT>(mode, g, s, a); break;
        case 8: launch_fused_mode<8, MODEL, OPT>(mode, g, s, a); break;
        case 16: launch_fused_mode<16, MODEL, OPT>(mode, g, s, a); break;
        case 32: launch_fused_mode<32, MODEL, OPT>(mode, g, s, a); break;
        case 64: launch_fused_mode<64, MODEL, OPT>(mode, g, s, a); break;
        default: launch_generic_mode<MODEL, OPT>(mode, g, s, a); break;
    }
}

int orx_fused_can_inline_apply(int D) { return lpr_for_dim(D) != 0; }
int orx_fused_tpw(int D) { const int lpr = lpr_for_dim(D); return lpr ? 64 / lpr : 0; }

// lazy Adam (exact mode, float4 dims): its own small set of instantiations
template <int MODEL>
static void launch_fused_adam(int lpr, dim3 g, orx_ctx* s, const PairArgs& a) {
#define ORX_FC(L) do { if (a.stage) ORX_LAUNCH(s, (fused_kernel<L, MODEL, ORX_ADAM, MODE_EXACT, false, true, 2>), g, dim3(256), 0, a); \
                       else ORX_LAUNCH(s, (fused_kernel<L, MODEL, ORX_ADAM, MODE_EXACT, false, false, 2>), g, dim3(256), 0, a); } while (0)
    if (a.lrv != nullptr && !a.censor) {      // the closed-form replay (orx_device.h AdamCF): no loop over the skipped steps
        switch (lpr) {
            case 4: ORX_FC(4); break;
            case 8: ORX_FC(8); break;
            case 16: ORX_FC(16); break;
            case 32: ORX_FC(32); break;
            default: ORX_FC(64); break;
        }
        return;
    }
#undef ORX_FC
#define ORX_FL(L) do { if (a.stage) ORX_LAUNCH(s, (fused_kernel<L, MODEL, ORX_ADAM, MODE_EXACT, false, true, 1>), g, dim3(256), 0, a); \
                       else ORX_LAUNCH(s, (fused_kernel<L, MODEL, ORX_ADAM, MODE_EXACT, false, false, 1>), g, dim3(256), 0, a); } while (0)
    if (a.long_gap && !a.censor) {      // tables large relative to the batch: the variant with the bounded per-row replay
        switch (lpr) {
            case 4: ORX_FL(4); break;
            case 8: ORX_FL(8); break;
            case 16: ORX_FL(16); break;
            case 32: ORX_FL(32); break;
            default: ORX_FL(64); break;
        }
        return;
    }
#undef ORX_FL
#define ORX_FA(L) do { if (a.censor) { if (a.stage) ORX_LAUNCH(s, (fused_kernel<L, MODEL, ORX_ADAM, MODE_EXACT, true, true>), g, dim3(256), 0, a); \
                                         else ORX_LAUNCH(s, (fused_kernel<L, MODEL, ORX_ADAM, MODE_EXACT, true, false>), g, dim3(256), 0, a); } \
                       else if (a.stage) ORX_LAUNCH(s, (fused_kernel<L, MODEL, ORX_ADAM, MODE_EXACT, false, true>), g, dim3(256), 0, a); \
                       else ORX_LAUNCH(s, (fused_kernel<L, MODEL, ORX_ADAM, MODE_EXACT, false, false>), g, dim3(256), 0, a); } while (0)
    switch (lpr) {
        case 4: ORX_FA(4); break;
        case 8: ORX_FA(8); break;
        case 16: ORX_FA(16); break;
        case 32: ORX_FA(32); break;
        default: ORX_FA(64); break;
    }
#undef ORX_FA
}

int orx_launch_fused(orx_ctx* ctx, int model, int optkind, int mode, const PairArgs& a) {
    ProfScope ps(ctx, ORX_K_FUSED);
    const int lpr = lpr_for_dim(a.D);
    const dim3 g((unsigned)(fused_grid(a.D, a.B) + (mode == MODE_EXACT ? a.n_apply_blocks : 0)));
    if (optkind == ORX_ADAM && mode == MODE_EXACT) {
        ORX_ARG(lpr != 0 && a.lrt != nullptr, "fused: the lazy Adam path needs a float4 dim");
        if (model == ORX_BPR) launch_fused_adam<ORX_BPR>(lpr, g, ctx, a);
        else launch_fused_adam<ORX_UCML>(lpr, g, ctx, a);
        ORX_HIP(hipGetLastError());
        return ORX_OK;
    }
    const int ok = (optkind == ORX_ADAGRAD) ? ORX_ADAGRAD : ORX_SGD;
    if (model == ORX_BPR) {
        if (ok == ORX_ADAGRAD) launch_fused_lpr<ORX_BPR, ORX_ADAGRAD>(lpr, mode, g, ctx, a);
        else launch_fused_lpr<ORX_BPR, ORX_SGD>(lpr, mode, g, ctx, a);
    } else {
        if (ok == ORX_ADAGRAD) launch_fused_lpr<ORX_UCML, ORX_ADAGRAD>(lpr, mode, g, ctx, a);
        else launch_fused_lpr<ORX_UCML, ORX_SGD>(lpr, mode, g, ctx, a);
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

template <int OPT>
static void launch_dup_apply_lpr(int lpr, dim3 g, orx_ctx* s, const PairArgs& a) {
    switch (lpr) {
        case 4: ORX_LAUNCH(s, (dup_apply_kernel<4, OPT>), g, dim3(256), 0, a); break;
        case 8: ORX_LAUNCH(s, (dup_apply_kernel<8, OPT>), g, dim3(256), 0, a); break;
        case 16: ORX_LAUNCH(s, (dup_apply_kernel<16, OPT>), g, dim3(256), 0, a); break;
        case 32: ORX_LAUNCH(s, (dup_apply_kernel<32, OPT>), g, dim3(256), 0, a); break;
        case 64: ORX_LAUNCH(s, (dup_apply_kernel<64, OPT>), g, dim3(256), 0, a); break;
        default: ORX_LAUNCH(s, (dup_apply_generic_kernel<OPT>), g, dim3(256), 0, a); break;
    }
}

int orx_launch_hot_reduce(orx_ctx* ctx, const PairArgs& a, int level) {
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    const int lpr = lpr_for_dim(a.D);
    const dim3 g(level == 0 ? 2048 : 256);
    switch (lpr) {
        case 4: ORX_LAUNCH(ctx, (hot_reduce_kernel<4>), g, dim3(256), 0, a, level); break;
        case 8: ORX_LAUNCH(ctx, (hot_reduce_kernel<8>), g, dim3(256), 0, a, level); break;
        case 16: ORX_LAUNCH(ctx, (hot_reduce_kernel<16>), g, dim3(256), 0, a, level); break;
        case 32: ORX_LAUNCH(ctx, (hot_reduce_kernel<32>), g, dim3(256), 0, a, level); break;
        case 64: ORX_LAUNCH(ctx, (hot_reduce_kernel<64>), g, dim3(256), 0, a, level); break;
        default: orx_set_error("hot_reduce: no staging for dim %d", a.D); return ORX_ERR_ARG;
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// dup_apply of the call's last step + loss_reduce of its K steps in one launch (float4 dims; SGD / Adagrad); false: not applicable
bool orx_launch_tail(orx_ctx* ctx, int optkind, const PairArgs& a, const ReduceArgs& r, int64_t K, int* rc) {
    const int lpr = lpr_for_dim(a.D);
    if (lpr == 0 || optkind == ORX_ADAM || K > 4096) return false;
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    int64_t want = (a.B * 3 / 16) / (16 * (64 / lpr)) + 1;      // (1024-thread blocks: the loss sums want them)
    if (want > 512) want = 512;
    if (want < 16) want = 16;
    const dim3 g((unsigned)(want + K));
#define ORX_TL(L) do { if (optkind == ORX_ADAGRAD) ORX_LAUNCH(ctx, (tail_kernel<L, ORX_ADAGRAD>), g, dim3(1024), 0, a, r, (int)K); \
                       else ORX_LAUNCH(ctx, (tail_kernel<L, ORX_SGD>), g, dim3(1024), 0, a, r, (int)K); } while (0)
    switch (lpr) {
        case 4: ORX_TL(4); break;
        case 8: ORX_TL(8); break;
        case 16: ORX_TL(16); break;
        case 32: ORX_TL(32); break;
        default: ORX_TL(64); break;
    }
#undef ORX_TL
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orx_set_error("tail_kernel launch failed: %s", hipGetErrorString(e)); *rc = ORX_ERR_HIP; } else *rc = ORX_OK;
    return true;
}

// The number of duplicated rows lives in device memory: fixed grid, grid-stride loop.
int orx_launch_dup_apply(orx_ctx* ctx, int optkind, const PairArgs& a) {
    ProfScope ps(ctx, ORX_K_DUPAPPLY);
    const int lpr = lpr_for_dim(a.D);
    int64_t want = (a.B * 3 / 16) / (lpr ? 4 * (64 / lpr) : 4) + 1;     // ~ expected duplicates of a uniform batch
    if (want > 2048) want = 2048;
    if (want < 64) want = 64;
    const dim3 g((unsigned)want);
    if (optkind == ORX_ADAM && a.lrt != nullptr && lpr != 0) {       // lazy Adam (never the generic-dim kernel)
        switch (lpr) {
            case 4: ORX_LAUNCH(ctx, (dup_apply_kernel<4, ORX_ADAM>), g, dim3(256), 0, a); break;
            case 8: ORX_LAUNCH(ctx, (dup_apply_kernel<8, ORX_ADAM>), g, dim3(256), 0, a); break;
            case 16: ORX_LAUNCH(ctx, (dup_apply_kernel<16, ORX_ADAM>), g, dim3(256), 0, a); break;
            case 32: ORX_LAUNCH(ctx, (dup_apply_kernel<32, ORX_ADAM>), g, dim3(256), 0, a); break;
            default: ORX_LAUNCH(ctx, (dup_apply_kernel<64, ORX_ADAM>), g, dim3(256), 0, a); break;
        }
    } else if (optkind == ORX_ADAGRAD) launch_dup_apply_lpr<ORX_ADAGRAD>(lpr, g, ctx, a);
    else launch_dup_apply_lpr<ORX_SGD>(lpr, g, ctx, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
