// Bucketed duplicate plan of the exact train steps (gfx950): the same contract as dedup_kernel + urgent_kernel
// (kernels_pairwise.hip) -- ids rewritten with duplicate flag / role / urgent bit, the list of duplicated rows, the
// staging plan of rows referenced >= 3 times with its reduction tree -- with a latency that does not depend on how few
// steps a call carries.
//
// dedup_kernel gives one 1024-thread workgroup a 425 984-row range and lets it stream ALL ids of its table twice: six
// workgroups per step, each busy for 150-190 us.  With K = 200 steps in a call that is 3.4 us per step (the device is
// full); with the K = 20 of a short call the 120 workgroups leave half the CUs idle and the first fused launch starts
// 230 us late (dedup 193 us + urgent 39 us: 11.6 us per step on top of a 30-us kernel).
//
// Here the references are first PARTITIONED by row range (2^shift rows, 16 384 for tables up to 64 M rows), exactly
// (count -> scan -> scatter, no capacity guess, nothing depends on the id distribution), and a 256-thread workgroup
// per (step, range) then works on its own ~3-4 k references only:
//   plan_part_kernel<0>  histogram of the step's references over the ranges (LDS, then one atomic per touched bin)
//   plan_part_kernel<1>  exclusive prefix of the histogram (per workgroup), then (id, output position) pairs into the step's
//                        bucket lists and the coalesced copy of the ids; out-of-range ids are marked here
//   plan_range_kernel    three LDS bitmaps over the range (seen / twice / three times), roles by arrival, dense
//                        numbering of the rows with >= 3 references, their ranks and staging segments (per-row
//                        counters in LDS), duplicate list with (segment, count), reduction-tree work items
//   plan_urgent_kernel   bit 28 on the references of step s whose row was duplicated in step s-1 (in-launch apply)
// The reference's semantics being restated are TF's: every gradient of a step is taken on the pre-step tables and
// duplicate indices are summed before the sparse apply (tf2_examples/bpr_citeulike.py:35-38).
#include "orx_internal.h"

#include "orx_device.h"

constexpr int PL_THREADS = 256;
constexpr int PL_REFS = 8;                        // references per thread in the count / scatter kernels
constexpr int PL_CHUNK = PL_THREADS * PL_REFS;
constexpr int PL_LCNT = 2048;                     // per-row reference counters of the staging plan kept in LDS (more tri rows in a range: global counters)

struct PlanArgs {
    DedupArgs d;
    int shift;                 // rows per range = 1 << shift
    int nru, nri;              // ranges of the user / item table
    int* bcnt;                 // [K][3 nb + 1]: references per bucket [nb], scatter cursors [nb], exclusive offsets [nb + 1]
    int2* list;                // [K][nref] (id, output position | role << 30)
    int64_t nref;
    unsigned int* dupbits;     // [K][nb][words] "seen twice" bitmaps for plan_urgent_kernel, or NULL
    int min_late;              // staging plan in ranges with at least this many third-or-later references (< 0: max(64, n / 512))
};

// reference j of the step (users, then pos items, then neg items): id, table, position in ids_out / refinfo
__device__ __forceinline__ bool plan_ref(const PlanArgs& a, int64_t s, int64_t j, int& id, bool& is_user, int& pos) {
    const DedupArgs& d = a.d;
    if (j < d.nU) { id = d.uid[s * d.id_stride + j]; is_user = true; pos = (int)j; }
    else if (j < d.nU + d.nP) { id = d.pid[s * d.id_stride + (j - d.nU)]; is_user = false; pos = (int)(d.role_stride ? d.role_stride + (j - d.nU) : j); }
    else { id = d.nid[s * d.id_stride + (j - d.nU - d.nP)]; is_user = false; pos = (int)(d.role_stride ? 2 * d.role_stride + (j - d.nU - d.nP) : j); }
    return id_ok(id, is_user ? d.NU : d.NI);
}

// exclusive prefix sum of one int per thread over the PL_THREADS-thread workgroup; `total` = sum
__device__ __forceinline__ int plan_scan_excl(int v, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < PL_THREADS / 64; ++k) { const int t = wave_tot[k]; if (k < wave) before += t; all += t; }
    __syncthreads();
    total = all;
    return before + incl - v;
}

// SCATTER = false: count the references of a chunk per bucket.  SCATTER = true: write them into the bucket lists.
template <bool SCATTER>
__global__ __launch_bounds__(PL_THREADS) void plan_part_kernel(PlanArgs a) {
    extern __shared__ int pl_hist[];               // [nb] counts, (scatter) [nb] list positions, [nb + 1] bucket offsets
    const int nb = a.nru + a.nri;
    const int64_t s = blockIdx.y;
    int* hist = pl_hist;
    int* base = pl_hist + nb;
    for (int i = threadIdx.x; i < nb; i += PL_THREADS) hist[i] = 0;
    __syncthreads();
    const int64_t j0 = (int64_t)blockIdx.x * PL_CHUNK + threadIdx.x;
    int bk[PL_REFS], rk[PL_REFS], idv[PL_REFS], posv[PL_REFS];
#pragma unroll
    for (int k = 0; k < PL_REFS; ++k) {
        const int64_t j = j0 + (int64_t)k * PL_THREADS;
        bk[k] = -1;
        if (j < a.nref) {
            int id, pos; bool is_user;
            const bool ok = plan_ref(a, s, j, id, is_user, pos);
            if (ok) {
                bk[k] = (is_user ? 0 : a.nru) + (id >> a.shift);
                idv[k] = id; posv[k] = pos;
            }
            // the rewritten ids start as a (coalesced) copy; plan_range_kernel then touches only the duplicated references --
            // a scattered 4-byte store costs a memory transaction of its own.  0x7fffffff: out-of-range id, never a valid row
            if (SCATTER) a.d.ids_out[s * a.d.flag_stride + pos] = ok ? id : 0x7fffffff;
        }
    }
#pragma unroll
    for (int k = 0; k < PL_REFS; ++k) if (bk[k] >= 0) rk[k] = atomicAdd(&hist[bk[k]], 1);
    __syncthreads();
    int* cnt = a.bcnt + s * (3 * nb + 1);
    if (!SCATTER) {
        // the step's list and allocator counters start at zero (first used by plan_range_kernel): no memsets of their own
        if (blockIdx.x == 0 && threadIdx.x < 8) a.d.alloc[8 * s + threadIdx.x] = 0;
        if (blockIdx.x == 0 && threadIdx.x == 8) a.d.dcount[s] = 0;
        for (int i = threadIdx.x; i < nb; i += PL_THREADS) { const int c = hist[i]; if (c) atomicAdd(cnt + i, c); }
        return;
    }
    // exclusive prefix of the step's bucket counts, by every workgroup for itself (a scan kernel of its own cost a launch
    // and a gap: 8 us of a short call); workgroup 0 leaves it in memory for plan_range_kernel / plan_urgent_kernel
    __shared__ int wave_tot[PL_THREADS / 64];
    int* off = pl_hist + 2 * nb;
    for (int i = threadIdx.x; i < nb; i += PL_THREADS) off[i] = cnt[i];
    __syncthreads();
    {
        const int per = (nb + PL_THREADS - 1) / PL_THREADS;
        const int i0 = threadIdx.x * per;
        int mine = 0;
        for (int i = i0; i < i0 + per && i < nb; ++i) mine += off[i];
        int total;
        int run = plan_scan_excl(mine, wave_tot, total);
        for (int i = i0; i < i0 + per && i < nb; ++i) { const int c = off[i]; off[i] = run; run += c; }
        if (threadIdx.x == 0) off[nb] = total;
    }
    __syncthreads();
    if (blockIdx.x == 0) for (int i = threadIdx.x; i <= nb; i += PL_THREADS) cnt[2 * nb + i] = off[i];
    int* cur = cnt + nb;
    for (int i = threadIdx.x; i < nb; i += PL_THREADS) { const int c = hist[i]; if (c) base[i] = off[i] + atomicAdd(cur + i, c); }
    __syncthreads();
    int2* list = a.list + s * a.nref;
#pragma unroll
    for (int k = 0; k < PL_REFS; ++k) if (bk[k] >= 0) list[base[bk[k]] + rk[k]] = make_int2(idv[k], posv[k]);
}

// a reference counter after the ranks have been handed out (LDS, or global memory updated by L2 atomics)
__device__ __forceinline__ int pl_cnt(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int pl_peek(const unsigned int* p) { return *reinterpret_cast<const volatile unsigned int*>(p); }

// One workgroup per (range, step): the same plan dedup_kernel makes for its range, on the range's own references.
__global__ __launch_bounds__(PL_THREADS) void plan_range_kernel(PlanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int pl_lds[];
    const DedupArgs& d = a.d;
    const int W = (1 << a.shift) >> 5;                  // 32-bit words per bitmap
    unsigned int* seen = pl_lds;
    unsigned int* dup = pl_lds + W;
    unsigned int* tri = pl_lds + 2 * W;
    unsigned short* prefix16 = reinterpret_cast<unsigned short*>(pl_lds + 3 * W);     // dense number of a word's first tri row
    int* lcnt = reinterpret_cast<int*>(pl_lds + 3 * W + (W + 1) / 2);                  // references per dense tri row (LDS or global)
    __shared__ int wave_tot[PL_THREADS / 64];
    __shared__ int sh_late, sh_dense, sh_seg, list_cnt, list_base;
    const int nb = a.nru + a.nri;
    const int b = blockIdx.x;
    const int64_t s = blockIdx.y;
    const bool is_user = b < a.nru;
    const int64_t r0 = (int64_t)(is_user ? b : b - a.nru) << a.shift;
    const int* cnt = a.bcnt + s * (3 * nb + 1) + 2 * nb;
    const int lo = cnt[b], n = cnt[b + 1] - lo;
    int2* ent = a.list + s * a.nref + lo;
    unsigned int* dupout = a.dupbits ? a.dupbits + ((size_t)s * nb + b) * W : nullptr;
    if (n == 0) {                                       // nothing references this range in this step
        if (dupout) for (int w = threadIdx.x; w < W; w += PL_THREADS) dupout[w] = 0u;
        return;
    }
    int32_t* ids_out = d.ids_out + s * d.flag_stride;
    int2* refinfo = d.refinfo ? d.refinfo + s * d.flag_stride : nullptr;
    for (int i = threadIdx.x; i < 3 * W; i += PL_THREADS) pl_lds[i] = 0u;
    if (threadIdx.x == 0) { sh_late = 0; list_cnt = 0; }
    __syncthreads();
    // pass 1: bitmaps; the role of a reference among its row's references (first / second / later, by arrival)
    int late = 0;
    for (int i = threadIdx.x; i < n; i += PL_THREADS) {
        const int2 e = ent[i];
        const int l = (int)(e.x - r0);
        const unsigned int bit = 1u << (l & 31);
        // the bitmaps only gain bits: a plain read that already shows the bit decides like the atomic would (a hot row
        // would otherwise serialize thousands of LDS atomics on one word)
        const unsigned int cur = pl_peek(&seen[l >> 5]);
        const unsigned int old = (cur & bit) ? cur : atomicOr(&seen[l >> 5], bit);
        int role = 0;
        if (old & bit) {
            const unsigned int cur2 = pl_peek(&dup[l >> 5]);
            const unsigned int old2 = (cur2 & bit) ? cur2 : atomicOr(&dup[l >> 5], bit);
            role = 1;
            if (old2 & bit) {
                if (!(pl_peek(&tri[l >> 5]) & bit)) atomicOr(&tri[l >> 5], bit);
                role = 2; ++late;
            }
        }
        if (role) ent[i].y = (int)((uint32_t)e.y | ((uint32_t)role << 30));       // (read back by the same thread in pass 2)
    }
    if (refinfo != nullptr && late) atomicAdd(&sh_late, late);
    __syncthreads();
    // staging plan where atomics would pile up: ranges with at least max(64, n / 512) third-or-later references
    const bool plan = refinfo != nullptr && sh_late >= (a.min_late < 0 ? (n / 512 > 64 ? n / 512 : 64) : a.min_late);
    int ntri = 0, dense0 = 0;
    int* segstart = nullptr;
    if (plan) {
        segstart = d.segstart + s * d.tri_stride;
        // dense numbers of the tri rows in row order: a thread owns a contiguous run of words, so its running count is the prefix
        const int per = (W + PL_THREADS - 1) / PL_THREADS;
        const int w0 = threadIdx.x * per;
        int mine3 = 0;
        for (int w = w0; w < w0 + per && w < W; ++w) mine3 += __popc(tri[w]);
        int pre = plan_scan_excl(mine3, wave_tot, ntri);
        if (ntri > 65535) ntri = 0;                    // (ranges above 65 536 rows only: no plan, atomics)
        if (ntri) {
            if (threadIdx.x == 0) sh_dense = atomicAdd(d.alloc + 8 * s, ntri);
            for (int w = w0; w < w0 + per && w < W; ++w) { prefix16[w] = (unsigned short)pre; pre += __popc(tri[w]); }
            __syncthreads();
            dense0 = sh_dense;
            // the rows' reference counters: LDS for up to PL_LCNT tri rows, this range's own slice of tricnt otherwise
            if (ntri > PL_LCNT) lcnt = d.tricnt + s * d.tri_stride + dense0;
            for (int i = threadIdx.x; i < ntri; i += PL_THREADS) lcnt[i] = 0;
            __syncthreads();
        }
    }
    // pass 2: rewritten ids, (dense row, rank) of the references that stage
    for (int i = threadIdx.x; i < n; i += PL_THREADS) {
        const int2 e = ent[i];
        const int l = (int)(e.x - r0);
        const int pos = e.y & 0x3fffffff;
        const unsigned int dd = (dup[l >> 5] >> (l & 31)) & 1u;
        if (dd) {                                       // (references of unique rows keep the plain id plan_part_kernel wrote)
            uint32_t v = (uint32_t)e.x | (1u << 31);
            const unsigned int tw = tri[l >> 5];
            const unsigned int t3 = (tw >> (l & 31)) & 1u;
            v |= (t3 ? 2u : ((uint32_t)e.y >> 30)) << 29;
            if (t3 && ntri) {
                const int dn = (int)prefix16[l >> 5] + __popc(tw & ((1u << (l & 31)) - 1u));
                refinfo[pos] = make_int2(dense0 + dn, atomicAdd(lcnt + dn, 1));
            } else if (t3 && refinfo != nullptr) {
                refinfo[pos] = make_int2(-1, 0);
            }
            ids_out[pos] = (int32_t)v;
        }
    }
    if (dupout) for (int w = threadIdx.x; w < W; w += PL_THREADS) dupout[w] = dup[w];
    // segment start of every tri row (its references' slots are contiguous: segstart + rank)
    if (ntri) {
        __syncthreads();                               // the counts are final
        const int per = (ntri + PL_THREADS - 1) / PL_THREADS;
        const int d0 = threadIdx.x * per;
        int csum = 0;
        for (int k = d0; k < d0 + per && k < ntri; ++k) csum += pl_cnt(lcnt + k);
        int total_refs;
        int run = plan_scan_excl(csum, wave_tot, total_refs);
        if (threadIdx.x == 0) sh_seg = atomicAdd(d.alloc + 8 * s + 1, total_refs);
        __syncthreads();
        run += sh_seg;
        for (int k = d0; k < d0 + per && k < ntri; ++k) { segstart[dense0 + k] = run; run += pl_cnt(lcnt + k); }
        __syncthreads();                               // list emission below reads segstart of other threads' rows (agent-scope loads)
    }
    // append the duplicated rows of this range to the step's list
    int mine = 0;
    for (int w = threadIdx.x; w < W; w += PL_THREADS) mine += __popc(dup[w]);
    int off = 0;
    if (mine) off = atomicAdd(&list_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        list_base = list_cnt ? atomicAdd(d.dcount + s, list_cnt) : 0;
        if (list_cnt) atomicAdd(d.alloc + 8 * s + 5, list_cnt);      // (the host reads the allocators only: one copy)
    }
    __syncthreads();
    if (mine) {
        int64_t e = s * d.list_stride + list_base + off;
        const uint32_t tag = is_user ? 0u : 0x80000000u;
        for (int w = threadIdx.x; w < W; w += PL_THREADS) {
            unsigned int m = dup[w];
            const unsigned int tw = tri[w];
            while (m) {
                const int bpos = __ffs(m) - 1;
                m &= m - 1;
                d.dlist[e] = (uint32_t)(r0 + (int64_t)w * 32 + bpos) | tag;
                if (d.dcnt != nullptr) {
                    int c = 0, sg = 0;
                    if (ntri && ((tw >> bpos) & 1u)) {
                        const int dn = (int)prefix16[w] + __popc(tw & ((1u << bpos) - 1u));
                        c = pl_cnt(lcnt + dn);
                        sg = __hip_atomic_load(segstart + dense0 + dn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (c > ORX_SEG_DIRECT) {           // long segment: reduction tree (see dedup_kernel)
                            int4* items = d.items + s * d.item_stride;
                            int src = sg, len = c, level = 0;
                            do {
                                const int pieces = (len + ORX_PIECE - 1) / ORX_PIECE;
                                const int b0 = d.tree_off[level] + atomicAdd(d.alloc + 8 * s + 2 + level, pieces);
                                for (int k = 0; k < pieces; ++k) {
                                    const int rem = len - k * ORX_PIECE;
                                    items[b0 + k] = make_int4(src + k * ORX_PIECE, rem < ORX_PIECE ? rem : ORX_PIECE, b0 + k, 0);
                                }
                                src = b0; len = pieces; ++level;
                            } while (len > ORX_SEG_DIRECT && level < 3);
                            sg = src; c = -len;
                        }
                    }
                    d.dseg[e] = sg; d.dcnt[e] = c;
                }
                ++e;
            }
        }
    }
}

// bit 28 on the references of step s (= 1 + blockIdx.y) whose row was duplicated in step s-1
__global__ __launch_bounds__(PL_THREADS) void plan_urgent_kernel(PlanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int pl_lds[];
    const int W = (1 << a.shift) >> 5;
    const int nb = a.nru + a.nri;
    const int b = blockIdx.x;
    const int64_t s = 1 + blockIdx.y;
    const int* cnt = a.bcnt + s * (3 * nb + 1) + 2 * nb;
    const int lo = cnt[b], n = cnt[b + 1] - lo;
    if (n == 0) return;
    const int prev_n = cnt[b + 1 - (3 * nb + 1)] - cnt[b - (3 * nb + 1)];
    if (prev_n < 2) return;                             // no duplicated row without two references
    const unsigned int* prev = a.dupbits + ((size_t)(s - 1) * nb + b) * W;
    for (int w = threadIdx.x; w < W; w += PL_THREADS) pl_lds[w] = prev[w];
    __syncthreads();
    const int64_t r0 = (int64_t)(b < a.nru ? b : b - a.nru) << a.shift;
    const int2* ent = a.list + s * a.nref + lo;
    int32_t* ids_out = a.d.ids_out + s * a.d.flag_stride;
    for (int i = threadIdx.x; i < n; i += PL_THREADS) {
        const int2 e = ent[i];
        const int l = (int)(e.x - r0);
        if ((pl_lds[l >> 5] >> (l & 31)) & 1u) ids_out[e.y & 0x3fffffff] |= (1 << 28);
    }
}

// ------------------------------------------------------------------------------------------ host side ---
// rows per range = 1 << shift: 16 384 rows up to 64 ranges per table, then as many ranges as give a workgroup ~1 k
// references (a range with a handful of references costs a workgroup's fixed work: 10 M x 50 M tables at 16 384 rows per
// range are 3663 workgroups per step, 16 us; at 131 072 rows 459 and 5 us)
int orx_plan_shift(int64_t NU, int64_t NI, int64_t nref) {
    static const char* env = getenv("ORX_PLAN_SHIFT");      // experiments: rows per range = 1 << value
    const int64_t rows = NU > NI ? NU : NI;
    const int cap = rows > (1LL << 27) ? 18 : 17;
    if (env) { const int v = atoi(env); return v < 10 ? 10 : (v > 18 ? 18 : v); }
    int shift = 14;
    const int64_t want = std::max<int64_t>(64, nref / 1024);
    while (shift < cap && ((rows + (1LL << shift) - 1) >> shift) > want) ++shift;
    return shift;
}
int orx_plan_ranges(int64_t rows, int shift) { return (int)((rows + (1LL << shift) - 1) >> shift); }

// scratch of the bucketed plan for `chunk` steps of up to nref references over tables of NU / NI rows (grow-only; sized
// for the smallest range the plan may choose)
int orx_plan_buffers(orx_ctx* c, int64_t chunk, int64_t nref, int64_t NU, int64_t NI, bool want_dupbits) {
    const char* env = getenv("ORX_PLAN_SHIFT");
    const int smin = env ? std::min(14, std::max(10, atoi(env))) : 14;
    const int nb = orx_plan_ranges(NU, smin) + orx_plan_ranges(NI, smin);
    if (orx_ensure((void**)&c->d_pl_cnt, &c->d_pl_cnt_cap, (size_t)chunk * (3 * nb + 1) * sizeof(int))) return ORX_ERR_OOM;
    if (orx_ensure((void**)&c->d_pl_list, &c->d_pl_list_cap, (size_t)chunk * nref * sizeof(int2))) return ORX_ERR_OOM;
    // one bit per row, every range rounded up to whole words of its own
    const size_t words = (size_t)((NU + NI) / 32) + 2 * ((size_t)1 << 13) + (size_t)nb;
    if (want_dupbits && orx_ensure((void**)&c->d_dupbits, &c->d_dupbits_cap, (size_t)chunk * words * sizeof(unsigned int))) return ORX_ERR_OOM;
    return ORX_OK;
}

// The plan of kc steps.  `d` is filled as for orx_launch_dedup (dupbits non-NULL: the bitmaps for orx_plan_urgent are kept).
int orx_launch_plan(orx_ctx* ctx, const DedupArgs& d, int64_t kc, bool keep_dupbits) {
    ProfScope ps(ctx, ORX_K_DEDUP);
    PlanArgs a;
    a.d = d;
    a.nref = d.nU + d.nP + d.nN;
    a.shift = orx_plan_shift(d.nU ? d.NU : 0, (d.nP + d.nN) ? d.NI : 0, a.nref);
    a.nru = d.nU ? orx_plan_ranges(d.NU, a.shift) : 0;
    a.nri = (d.nP + d.nN) ? orx_plan_ranges(d.NI, a.shift) : 0;
    const int nb = a.nru + a.nri;
    if (nb == 0 || a.nref == 0 || kc == 0) return ORX_OK;
    ORX_ARG(a.nref < (1LL << 30) && kc < 65536, "plan: too many references per step (%lld) or steps (%lld)", (long long)a.nref, (long long)kc);
    a.bcnt = ctx->d_pl_cnt; a.list = ctx->d_pl_list;
    a.dupbits = keep_dupbits ? ctx->d_dupbits : nullptr;
    const char* ml = getenv("ORX_PLAN_MIN_LATE");      // experiments
    a.min_late = ml ? atoi(ml) : d.min_late;
    ORX_HIP(hipMemsetAsync(a.bcnt, 0, (size_t)kc * (3 * nb + 1) * sizeof(int), ctx->stream));
    const dim3 gp((unsigned)((a.nref + PL_CHUNK - 1) / PL_CHUNK), (unsigned)kc);
    const size_t hist_bytes = (size_t)(3 * nb + 1) * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        ORX_HIP(hipFuncSetAttribute((const void*)plan_part_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ORX_HIP(hipFuncSetAttribute((const void*)plan_part_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ORX_HIP(hipFuncSetAttribute((const void*)plan_range_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        attr_set = true;
    }
    ORX_LAUNCH(ctx, (plan_part_kernel<false>), gp, dim3(PL_THREADS), hist_bytes, a);
    ORX_LAUNCH(ctx, (plan_part_kernel<true>), gp, dim3(PL_THREADS), hist_bytes, a);
    const int W = (1 << a.shift) >> 5;
    const size_t lds = (size_t)(3 * W + (W + 1) / 2) * 4 + (size_t)PL_LCNT * 4;
    ORX_LAUNCH(ctx, plan_range_kernel, dim3((unsigned)nb, (unsigned)kc), dim3(PL_THREADS), lds, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_plan_urgent(orx_ctx* ctx, const DedupArgs& d, int64_t kc) {
    ProfScope ps(ctx, ORX_K_DEDUP);
    if (kc < 2) return ORX_OK;
    PlanArgs a;
    a.d = d;
    a.nref = d.nU + d.nP + d.nN;
    a.shift = orx_plan_shift(d.nU ? d.NU : 0, (d.nP + d.nN) ? d.NI : 0, a.nref);
    a.nru = d.nU ? orx_plan_ranges(d.NU, a.shift) : 0;
    a.nri = (d.nP + d.nN) ? orx_plan_ranges(d.NI, a.shift) : 0;
    a.bcnt = ctx->d_pl_cnt; a.list = ctx->d_pl_list; a.dupbits = ctx->d_dupbits; a.min_late = -1;
    const int W = (1 << a.shift) >> 5;
    ORX_LAUNCH(ctx, plan_urgent_kernel, dim3((unsigned)(a.nru + a.nri), (unsigned)(kc - 1)), dim3(PL_THREADS), (size_t)W * 4, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
