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
//   plan_pair_kernel     pairing (below): accepts / refuses the rows referenced exactly twice
//   plan_swap_kernel     pairing: the input records of the positions an accepted pair moves change places (the records themselves --
//                        (user, pos, neg, pairing word) per position -- are written by plan_part_kernel and flagged in place)
//
// PAIRING (round 4).  A row referenced exactly twice in a step costs the exact step ten row moves (two reads, two gradient
// deposits, and the apply's three reads + three writes) where a racy kernel pays four.  The order of the triplets inside a
// batch is free (the loss is a sum over the batch), so the plan brings the two triplets of such a row into the SAME WAVEFRONT of
// the fused kernel: there the two lane groups exchange their gradients of the shared row, one of them adds both and updates the
// row in place, the other does not write it -- two row moves, no deposit, no apply, no ready flag.  plan_range_kernel tells
// each of the two references where the other sits; plan_pair_kernel (one thread per entry of the step's list of duplicated rows)
// accepts a row when it is the first such row of both its triplets and a neighbouring position can be displaced -- a rule on data nothing writes meanwhile: no claims,
// deterministic -- and sends the refused rows down the deposit / apply path; plan_swap_kernel puts the fused kernel's input into the
// new order.  Rows with three or more references and rows whose two references sit in one triplet always keep the deposit path.
// The fused kernel sees a paired reference as a reference to a unique row.
// The reference's semantics being restated are TF's: every gradient of a step is taken on the pre-step tables and
// duplicate indices are summed before the sparse apply (tf2_examples/bpr_citeulike.py:35-38).
#include "orx_internal.h"

#include "orx_device.h"

#include <cstring>
#include <vector>
#include <algorithm>

constexpr int PL_THREADS = 256;
constexpr int PL_REFS = 8;                        // references per thread in the count / scatter kernels
constexpr int PL_CHUNK = PL_THREADS * PL_REFS;
constexpr int PL_LCNT = 1024;                     // per-row reference counters of the staging plan kept in LDS (more tri rows in a range: global counters)

struct PlanArgs {
    DedupArgs d;
    int shift;                 // rows per bucket <= 1 << shift
    int nru, nri;              // buckets of the user / item table: POWERS OF TWO.  Row r of a table belongs to bucket r & (n - 1) and is
    int lgu, lgi;              // row r >> lg(n) inside it: interleaved, so that a contiguous run of hot ids (a vocabulary sorted by
                               // frequency: the head of a Zipf distribution) spreads over all buckets instead of filling one
    int* bcnt;                 // [K][3 nb + 1]: references per bucket [nb], scatter cursors [nb], exclusive offsets [nb + 1]
    int2* list;                // [K][nref] (id, output position | role << 30)
    int64_t nref;
    unsigned int* dupbits;     // [K][nb][words] "seen twice" bitmaps for plan_urgent_kernel, or NULL
    int min_late;              // staging plan in ranges with at least this many third-or-later references (< 0: max(64, n / 512))
    int s_first;               // plan_urgent_kernel: first step to mark (1, or 0 when the plan's step 0 has a predecessor in the same arrays)
    unsigned long long* tstamp; // ORX_PLAN_TIMING: [workgroups][8] wall-clock stamps of plan_range_kernel's phases (NULL: off)
};
#define PL_STAMP(i) do { if (a.tstamp != nullptr && threadIdx.x == 0) a.tstamp[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)

// reference j of the step (users, then pos items, then neg items): id, table, position in ids_out / refinfo
__device__ __forceinline__ bool plan_ref(const PlanArgs& a, int64_t s, int64_t j, int& id, bool& is_user, int& pos) {
    const DedupArgs& d = a.d;
    if (j < d.nU) { id = d.uid[s * d.id_stride + j]; is_user = true; pos = (int)j; }
    else if (j < d.nU + d.nP) { id = d.pid[s * d.id_stride + (j - d.nU)]; is_user = false; pos = (int)(d.role_stride ? d.role_stride + (j - d.nU) : j); }
    else { id = d.nid[s * d.id_stride + (j - d.nU - d.nP)]; is_user = false; pos = (int)(d.role_stride ? 2 * d.role_stride + (j - d.nU - d.nP) : j); }
    return id_ok(id, is_user ? d.NU : d.NI);
}

// The rewritten id of reference `pos` of step s (pos = slot * role_stride + triplet).  With pairing it lives IN the fused kernel's
// input record (ids4[s][triplet], words x / y / z): plan_part_kernel writes the records themselves, the later plan kernels put their
// flags there, and what used to be a whole pass (a packed copy of ids_out in the paired order) is a swap of the few records that move.
__device__ __forceinline__ int32_t* plan_idword(const DedupArgs& d, int64_t s, int pos) {
    if (d.pair_tpw > 1) {
        const int Bp = (int)d.role_stride;
        const int slot = (pos >= Bp ? 1 : 0) + (pos >= 2 * Bp ? 1 : 0);
        return reinterpret_cast<int32_t*>(d.ids4 + s * d.pair_stride + (pos - slot * Bp)) + slot;
    }
    return d.ids_out + s * d.flag_stride + pos;
}

// Workgroup barrier that orders LDS traffic only: global loads and returning atomics already in flight stay in flight across it (the
// compiler waits for their registers where they are used).  __syncthreads() drains vmcnt too -- in kernels that are chains of
// dependent global round trips that serializes what could overlap.  NOT for handing GLOBAL data from one wavefront to another.
__device__ __forceinline__ void pl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive prefix sum of one int per thread over the PL_THREADS-thread workgroup; `total` = sum
template <int T = PL_THREADS>
__device__ __forceinline__ int plan_scan_excl(int v, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    pl_lds_barrier();
    int before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < T / 64; ++k) { const int t = wave_tot[k]; if (k < wave) before += t; all += t; }
    pl_lds_barrier();
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
                bk[k] = is_user ? (id & (a.nru - 1)) : a.nru + (id & (a.nri - 1));
                idv[k] = id; posv[k] = pos;
            }
            // the rewritten ids start as a (coalesced) copy; plan_range_kernel then touches only the duplicated references --
            // a scattered 4-byte store costs a memory transaction of its own.  0x7fffffff: out-of-range id, never a valid row
            if (SCATTER) {
                if (a.d.pair_tpw > 1) {
                    // pairing: the thread of a triplet's USER reference writes the fused kernel's input record -- the three ids (0x7fffffff:
                    // out of range), no pairing word, processed where it stands (w = position << 10)
                    if (j < a.d.pair_stride) {
                        const int p = a.d.pid[s * a.d.id_stride + j];
                        // (a pointwise step: word z is the sample's label, not an id -- DedupArgs::label)
                        const int n = a.d.label ? __float_as_int(a.d.label[s * a.d.id_stride + j]) : a.d.nid[s * a.d.id_stride + j];
                        const bool okp = id_ok(p, a.d.NI), okn = a.d.label ? true : id_ok(n, a.d.NI);
                        a.d.ids4[s * a.d.pair_stride + j] = make_int4(ok ? id : 0x7fffffff, okp ? p : 0x7fffffff, okn ? n : 0x7fffffff, (int)((uint32_t)j << 10));
                        // (the pairing record is NOT initialised: its words count only with this plan's generation, orx_internal.h -- except
                        // for a triplet with an out-of-range id, which the fused kernel skips altogether: it must never be the partner a valid
                        // triplet relies on, and says so in every slot; the slot of the invalid id is never overwritten)
                        if (!(ok & okp & okn)) {
                            const int poison = (int)(ORX_PARTNER_POISON | ((uint32_t)a.d.pair_gen << 24));
                            a.d.partner[s * a.d.pair_stride + j] = make_int4(poison, poison, poison, 0);
                        }
                    }
                } else {
                    a.d.ids_out[s * a.d.flag_stride + pos] = ok ? id : 0x7fffffff;
                }
            }
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

// visit the n entries of a bucket list, PL_UN independent 8-byte loads in flight per thread: a skewed id distribution puts
// most of a step's references into ONE bucket (Zipf(1.05) items: 80 k of 131 k), whose workgroup would otherwise walk its
// 300 entries per thread one dependent L2 round trip at a time (measured: 1-3 ms per such workgroup, 15 us per step)
constexpr int PL_UN = 8;
template <int T, class F>
__device__ __forceinline__ void pl_for_each(const int2* ent, int n, F f) {
    for (int i0 = threadIdx.x; i0 < n; i0 += PL_UN * T) {
        int2 e[PL_UN];
#pragma unroll
        for (int k = 0; k < PL_UN; ++k) { const int i = i0 + k * T; if (i < n) e[k] = ent[i]; }
#pragma unroll
        for (int k = 0; k < PL_UN; ++k) { const int i = i0 + k * T; if (i < n) f(i, e[k]); }
    }
}

// rank of a reference among its row's references = atomicAdd(count[row], 1) -- wave-aggregated for the rows that many lanes
// of the wavefront share: same-address LDS atomics with return serialize (a row with 12 k references of a Zipf step cost its
// workgroup 12 k of them; the ranges that hold the head of the distribution took 1-3 ms).  Up to four rounds peel off the
// groups of >= 4 lanes with one atomic each, the rest go one by one.
__device__ __forceinline__ int pl_rank(int* cnt, int dn) {
    const int lane = threadIdx.x & 63;
    int rank = -1;
    for (int round = 0; round < 4; ++round) {
        const unsigned long long todo = __ballot(rank < 0);
        if (!todo) break;
        const int ld = __shfl(dn, __ffsll((long long)todo) - 1);
        const bool mine = rank < 0 && dn == ld;
        const unsigned long long grp = __ballot(mine);
        const int gsz = __popcll(grp);
        if (gsz < 4) break;
        const int first = __ffsll((long long)grp) - 1;
        int base = 0;
        if (lane == first) base = atomicAdd(cnt + ld, gsz);
        base = __shfl(base, first);
        if (mine) rank = base + __popcll(grp & ((1ull << lane) - 1ull));
    }
    if (rank < 0) rank = atomicAdd(cnt + dn, 1);
    return rank;
}

constexpr int PL_PAIR_CAP = 512;                  // rows referenced exactly twice per range whose references learn of each other (positions in LDS)

// One workgroup per (range, step): the same plan dedup_kernel makes for its range, on the range's own references.
template <int T>
__global__ __launch_bounds__(T) void plan_range_kernel(PlanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int pl_lds[];
    const DedupArgs& d = a.d;
    const int W = (1 << a.shift) >> 5;                  // 32-bit words per bitmap
    unsigned int* seen = pl_lds;
    unsigned int* dup = pl_lds + W;
    unsigned int* tri = pl_lds + 2 * W;
    unsigned short* prefix16 = reinterpret_cast<unsigned short*>(pl_lds + 3 * W);     // dense number of a word's first tri row
    int* lcnt = reinterpret_cast<int*>(pl_lds + 3 * W + (W + 1) / 2);                  // references per dense tri row (LDS or global)
    // pairing: dense number of a word's first twice-referenced row, then per such row (local row, position of reference 0, of reference 1)
    unsigned short* dprefix16 = reinterpret_cast<unsigned short*>(seen);                // ("seen" is dead after pass 1)
    int* pairpos = reinterpret_cast<int*>(lcnt + PL_LCNT);
    __shared__ int wave_tot[T / 64];
    __shared__ int sh_late, sh_dense, sh_seg, list_cnt, list_base;
    const int nb = a.nru + a.nri;
    const int b = blockIdx.x;
    const int64_t s = blockIdx.y;
    const bool is_user = b < a.nru;
    const int lg = is_user ? a.lgu : a.lgi;             // local row of id: id >> lg; id of local row l: (l << lg) | bl
    const int bl = is_user ? b : b - a.nru;
    const int* cnt = a.bcnt + s * (3 * nb + 1) + 2 * nb;
    PL_STAMP(0);
    // (the step's counts and cursors have served plan_part_kernel: they go back to zero for the next plan, see orx_launch_plan)
    if (b == 0) for (int i = threadIdx.x; i < 2 * nb; i += T) a.bcnt[s * (3 * nb + 1) + i] = 0;
    const int lo = cnt[b], n = cnt[b + 1] - lo;
    int2* ent = a.list + s * a.nref + lo;
    unsigned int* dupout = a.dupbits ? a.dupbits + ((size_t)s * nb + b) * W : nullptr;
    if (threadIdx.x == 0 && n > 8192) atomicMax(d.alloc + 8 * s + 6, n);      // (the host sizes the next plan's workgroups by it)
    if (n == 0) {                                       // nothing references this range in this step
        if (dupout) for (int w = threadIdx.x; w < W; w += T) dupout[w] = 0u;
        return;
    }
    int2* refinfo = d.refinfo ? d.refinfo + s * d.flag_stride : nullptr;
    // The usual range (<= PL_UN entries per thread) keeps its entries in registers: they are requested before the LDS is zeroed and
    // serve both passes (one global round trip instead of three: load, role write-back, reload).
    const bool small = n <= PL_UN * T;
    int2 er[PL_UN];
    if (small) {
#pragma unroll
        for (int k = 0; k < PL_UN; ++k) { const int i = threadIdx.x + k * T; if (i < n) er[k] = ent[i]; }
    }
    for (int i = threadIdx.x; i < 3 * W; i += T) pl_lds[i] = 0u;
    if (threadIdx.x == 0) { sh_late = 0; list_cnt = 0; }
    pl_lds_barrier();
    PL_STAMP(1);
    // pass 1: bitmaps; the role of a reference among its row's references (first / second / later, by arrival)
    int late = 0;
    auto pass1 = [&](int2 e) -> int {
        const int l = e.x >> lg;
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
        return role;
    };
    if (small) {
#pragma unroll
        for (int k = 0; k < PL_UN; ++k) {
            const int i = threadIdx.x + k * T;
            if (i < n) er[k].y = (int)((uint32_t)er[k].y | ((uint32_t)pass1(er[k]) << 30));
        }
    } else {
        pl_for_each<T>(ent, n, [&](int i, int2 e) {
            const int role = pass1(e);
            if (role) ent[i].y = (int)((uint32_t)e.y | ((uint32_t)role << 30));       // (read back by the same thread in pass 2)
        });
    }
    if (late) atomicAdd(&sh_late, late);
    pl_lds_barrier();
    PL_STAMP(2);
    // The bitmaps are final: the range's share of the step's list of duplicated rows is allocated NOW -- the returning atomic is in
    // flight through pass 2 and is only looked at where the list is written.
    int mine = 0;
    for (int w = threadIdx.x; w < W; w += T) mine += __popc(dup[w]);
    int off = 0;
    if (mine) off = atomicAdd(&list_cnt, mine);
    pl_lds_barrier();
    int list_base_r = 0;
    if (threadIdx.x == 0 && list_cnt) {
        list_base_r = atomicAdd(d.dcount + s, list_cnt);
        atomicAdd(d.alloc + 8 * s + 5, list_cnt);      // (the host reads the allocators only: one copy)
    }
    // staging plan where atomics would pile up: ranges with at least max(64, n / 512) third-or-later references
    const bool want_plan = sh_late >= (a.min_late < 0 ? (n / 512 > 64 ? n / 512 : 64) : a.min_late);
    const bool plan = refinfo != nullptr && want_plan;
    // (a plan made with staging off: the allocator nobody uses then tells the host that a range would have staged -- the next
    // call's plan is made with staging on again, see orx_pairwise_step)
    if (want_plan && refinfo == nullptr && threadIdx.x == 0) atomicAdd(d.alloc + 8 * s + 1, sh_late);
    int ntri = 0, dense0 = 0;
    int* segstart = nullptr;
    if (plan) {
        segstart = d.segstart + s * d.tri_stride;
        // dense numbers of the tri rows in row order: a thread owns a contiguous run of words, so its running count is the prefix
        const int per = (W + T - 1) / T;
        const int w0 = threadIdx.x * per;
        int mine3 = 0;
        for (int w = w0; w < w0 + per && w < W; ++w) mine3 += __popc(tri[w]);
        int pre = plan_scan_excl<T>(mine3, wave_tot, ntri);
        if (ntri > 65535) ntri = 0;                    // (ranges above 65 536 rows only: no plan, atomics)
        if (ntri) {
            if (threadIdx.x == 0) sh_dense = atomicAdd(d.alloc + 8 * s, ntri);
            for (int w = w0; w < w0 + per && w < W; ++w) { prefix16[w] = (unsigned short)pre; pre += __popc(tri[w]); }
            __syncthreads();
            dense0 = sh_dense;
            // the rows' reference counters: LDS for up to PL_LCNT tri rows, this range's own slice of tricnt otherwise
            if (ntri > PL_LCNT) lcnt = d.tricnt + s * d.tri_stride + dense0;
            for (int i = threadIdx.x; i < ntri; i += T) lcnt[i] = 0;
            __syncthreads();
        }
    }
    // pairing: dense numbers of the rows referenced exactly twice (dup and not tri), in row order
    int ndup2 = 0;
    const bool pairing = d.pair_tpw > 1 && W <= 4096;      // (larger bitmaps leave no LDS for the pairing tables)
    if (pairing) {
        const int per = (W + T - 1) / T;
        const int w0 = threadIdx.x * per;
        int mine2 = 0;
        for (int w = w0; w < w0 + per && w < W; ++w) mine2 += __popc(dup[w] & ~tri[w]);
        int pre = plan_scan_excl<T>(mine2, wave_tot, ndup2);
        if (ndup2 > PL_PAIR_CAP) ndup2 = 0;             // (more such rows than the LDS tables hold: this range keeps the deposit path)
        if (ndup2) {
            for (int w = w0; w < w0 + per && w < W; ++w) { dprefix16[w] = (unsigned short)pre; pre += __popc(dup[w] & ~tri[w]); }
            for (int i = threadIdx.x; i < 2 * ndup2; i += T) pairpos[i] = -1;
            pl_lds_barrier();
        }
    }
    PL_STAMP(3);
    // pass 2: rewritten ids, (dense row, rank) of the references that stage
    auto pass2 = [&](int2 e) {
        const int l = e.x >> lg;
        const int pos = e.y & 0x3fffffff;
        const unsigned int dw = dup[l >> 5];
        const unsigned int dd = (dw >> (l & 31)) & 1u;
        if (dd) {                                       // (references of unique rows keep the plain id plan_part_kernel wrote)
            uint32_t v = (uint32_t)e.x | (1u << 31);
            const unsigned int tw = tri[l >> 5];
            const unsigned int t3 = (tw >> (l & 31)) & 1u;
            if (ndup2 && !t3) {
                // a row referenced exactly twice: the pairing phase below decides (and writes both rewritten ids)
                const int dn2 = (int)dprefix16[l >> 5] + __popc(dw & ~tw & ((1u << (l & 31)) - 1u));
                pairpos[2 * dn2 + (int)(((uint32_t)e.y >> 30) & 1u)] = pos;
                return;
            }
            v |= (t3 ? 2u : ((uint32_t)e.y >> 30)) << 29;
            if (t3 && ntri) {
                const int dn = (int)prefix16[l >> 5] + __popc(tw & ((1u << (l & 31)) - 1u));
                refinfo[pos] = make_int2(dense0 + dn, pl_rank(lcnt, dn));
            } else if (t3 && refinfo != nullptr) {
                refinfo[pos] = make_int2(-1, 0);
            }
            *plan_idword(d, s, pos) = (int32_t)v;
        }
    };
    if (small) {
#pragma unroll
        for (int k = 0; k < PL_UN; ++k) { const int i = threadIdx.x + k * T; if (i < n) pass2(er[k]); }
    } else {
        pl_for_each<T>(ent, n, [&](int, int2 e) { pass2(e); });
    }
    PL_STAMP(4);
    if (ndup2) {
        // pairing: the two references of a row referenced exactly twice learn where the other one sits (partner[], read by
        // plan_pair_kernel, which decides about the row: the rewritten ids of both, its list entry and its bit in the duplicate
        // bitmap are that kernel's)
        pl_lds_barrier();
        int* partner = reinterpret_cast<int*>(d.partner + s * d.pair_stride);      // record of triplet t, slot k: word 4 t + k
        const int Bp = (int)d.role_stride;
        for (int k = threadIdx.x; k < ndup2; k += T) {
            const int posA = pairpos[2 * k], posB = pairpos[2 * k + 1];
            const int sa = (posA >= Bp ? 1 : 0) + (posA >= 2 * Bp ? 1 : 0), sb = (posB >= Bp ? 1 : 0) + (posB >= 2 * Bp ? 1 : 0);
            const int tag = d.pair_gen << 24;
            partner[4 * (posA - sa * Bp) + sa] = posB | tag;                     // (bit 30 clear: the row's first reference, which owns the decision)
            partner[4 * (posB - sb * Bp) + sb] = posA | (1 << 30) | tag;
        }
    }
    PL_STAMP(5);
    if (dupout) for (int w = threadIdx.x; w < W; w += T) dupout[w] = ndup2 ? tri[w] : dup[w];      // (pairing: plan_pair_kernel adds the refused rows)
    // segment start of every tri row (its references' slots are contiguous: segstart + rank)
    if (ntri) {
        __syncthreads();                               // the counts are final
        const int per = (ntri + T - 1) / T;
        const int d0 = threadIdx.x * per;
        int csum = 0;
        for (int k = d0; k < d0 + per && k < ntri; ++k) csum += pl_cnt(lcnt + k);
        int total_refs;
        int run = plan_scan_excl<T>(csum, wave_tot, total_refs);
        if (threadIdx.x == 0) sh_seg = atomicAdd(d.alloc + 8 * s + 1, total_refs);
        __syncthreads();
        run += sh_seg;
        for (int k = d0; k < d0 + per && k < ntri; ++k) { segstart[dense0 + k] = run; run += pl_cnt(lcnt + k); }
        __syncthreads();                               // list emission below reads segstart of other threads' rows (agent-scope loads)
    }
    // append the duplicated rows of this range to the step's list (allocated after pass 1)
    PL_STAMP(6);
    if (threadIdx.x == 0) list_base = list_base_r;
    pl_lds_barrier();
    PL_STAMP(7);
    if (mine) {
        int64_t e = s * d.list_stride + list_base + off;
        const uint32_t tag = is_user ? 0u : 0x80000000u;
        for (int w = threadIdx.x; w < W; w += T) {
            unsigned int m = dup[w];
            const unsigned int m0 = m;
            const unsigned int tw = tri[w];
            while (m) {
                const int bpos = __ffs(m) - 1;
                m &= m - 1;
                if (ndup2 && !((tw >> bpos) & 1u)) {
                    // a row referenced exactly twice: the entry is reserved; plan_pair_kernel, which gets the position of the row's
                    // first reference here, fills it (the row, or ORX_DLIST_DEAD) and the pair record behind it
                    const int dn2 = (int)dprefix16[w] + __popc(m0 & ~tw & ((1u << bpos) - 1u));
                    reinterpret_cast<int2*>(d.pslot)[e] = make_int2(pairpos[2 * dn2], pairpos[2 * dn2 + 1]);
                    if (d.dcnt != nullptr) { d.dseg[e] = 0; d.dcnt[e] = 0; }
                    ++e;
                    continue;
                }
                if (d.pair_tpw > 1) reinterpret_cast<int2*>(d.pslot)[e] = make_int2(-1, -1);       // (not a row plan_pair_kernel decides about)
                d.dlist[e] = (uint32_t)((((int64_t)w * 32 + bpos) << lg) | bl) | tag;
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
template <int T>
__global__ __launch_bounds__(T) void plan_urgent_kernel(PlanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int pl_lds[];
    const int W = (1 << a.shift) >> 5;
    const int nb = a.nru + a.nri;
    const int b = blockIdx.x;
    const int64_t s = a.s_first + blockIdx.y;
    const int* cnt = a.bcnt + s * (3 * nb + 1) + 2 * nb;
    const int lo = cnt[b], n = cnt[b + 1] - lo;
    if (n == 0) return;
    const int prev_n = cnt[b + 1 - (3 * nb + 1)] - cnt[b - (3 * nb + 1)];
    if (prev_n < 2) return;                             // no duplicated row without two references
    const unsigned int* prev = a.dupbits + ((s - 1) * nb + b) * (int64_t)W;
    for (int w = threadIdx.x; w < W; w += T) pl_lds[w] = prev[w];
    __syncthreads();
    const int lg = b < a.nru ? a.lgu : a.lgi;
    const int2* ent = a.list + s * a.nref + lo;
    pl_for_each<T>(ent, n, [&](int, int2 e) {
        const int l = e.x >> lg;
        if ((pl_lds[l >> 5] >> (l & 31)) & 1u) *plan_idword(a.d, s, e.y & 0x3fffffff) |= (1 << 28);
    });
}

// pairing: one thread per entry of a step's list of duplicated rows decides about the row behind it, if that row is referenced
// exactly twice (plan_range_kernel left the position of its first reference in pslot), from partner[] alone -- which nothing
// writes here: no claims, no atomics on the way to a decision, the same outcome in every run.
//   A triplet's CHOICE is its first slot (user, pos item, neg item) that references such a row.  The row is ACCEPTED when it is the
//   choice of both its triplets (a triplet is then in at most one accepted pair) and the two can be brought together: they already
//   share a wavefront, or the position next to the lower one (its buddy, index ^ 1) may be displaced -- it is not itself in a mutual
//   pair -- or else the buddy of the upper one.  A position can only ever be displaced for its own buddy: accepted pairs never collide.
//   Accepted: the list entry is marked dead; perm[] sends the mover to the free position (and the displaced triplet to the mover's),
//   the two positions get their pairing words.  Both references stay what plan_part_kernel wrote: references to a unique row.
//   Refused: the two rewritten ids get the duplicate flag and roles 0 / 1, the entry its row, and the row its bit in the bitmap
//   plan_urgent_kernel reads.
__global__ __launch_bounds__(256) void plan_pair_kernel(PlanArgs a) {
    const DedupArgs& d = a.d;
    const int64_t s = blockIdx.y;
    const int B = (int)d.pair_stride, Bp = (int)d.role_stride, tpw = d.pair_tpw;
    const int4* part = d.partner + s * d.pair_stride;      // per triplet: where the partner of slot 0 / 1 / 2 sits (-1: none), w: its pairing word
    int* pword = reinterpret_cast<int*>(d.ids4 + s * d.pair_stride) + 3;          // the positions' pairing words: word w of the fused kernel's input records (4 t + 3)
    const int nb = a.nru + a.nri, W = (1 << a.shift) >> 5;
    const int n = d.dcount[s];
    auto slot_of = [&](int pos) { return (pos >= Bp ? 1 : 0) + (pos >= 2 * Bp ? 1 : 0); };
    // the choice of a triplet from its record: its first slot with a partner (-1: none)
    // a word of a record counts if it carries this plan's generation (the records are not initialised per plan) and is not the poison
    const uint32_t gen = (uint32_t)d.pair_gen;
    auto live = [gen](int w) { return (((uint32_t)w >> 24) & 63u) == gen && ((uint32_t)w & ORX_PARTNER_POS) != ORX_PARTNER_POISON; };
    auto poisoned = [gen](int4 r) {
        auto p = [gen](int w) { return (((uint32_t)w >> 24) & 63u) == gen && ((uint32_t)w & ORX_PARTNER_POS) == ORX_PARTNER_POISON; };
        return p(r.x) | p(r.y) | p(r.z);
    };
    auto choice = [live](int4 r) { return live(r.x) ? 0 : (live(r.y) ? 1 : (live(r.z) ? 2 : -1)); };
    auto word = [](int4 r, int k) { return k == 0 ? r.x : (k == 1 ? r.y : r.z); };
    const int4 none = make_int4(0, 0, 0, 0);               // (generation 0 is never a plan's)
    int npair = 0;
    // Two dependent levels of 16-byte record loads after the entry's two positions: the records of both triplets and of both buddies
    // (a buddy's record shares its triplet's 32 bytes); then the records of the buddies' partners.
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int64_t eg = s * d.list_stride + e;
        const int2 ab = reinterpret_cast<const int2*>(d.pslot)[eg];
        const int posA = ab.x, posB = ab.y;
        if (posA < 0) continue;                             // (a row with three or more references: plan_range_kernel's)
        const int sa = slot_of(posA), sb = slot_of(posB);
        const int ta = posA - sa * Bp, tb = posB - sb * Bp;
        const int lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
        const int bl = lo ^ 1, bh = hi ^ 1;                 // the buddies
        const bool same = (lo & ~(tpw - 1)) == (hi & ~(tpw - 1));
        const bool needq = ta != tb && !same;
        const int4 ra = part[ta], rb = part[tb];
        const int4 rl = (needq && bl < B) ? part[bl] : none, rh = (needq && bh < B) ? part[bh] : none;
        // the row is the choice of both its triplets
        const bool mutual = (ta != tb) & (choice(ra) == sa) & (choice(rb) == sb) & !(poisoned(ra) | poisoned(rb));       // (poison: a triplet with an invalid id)
        // is a buddy in a mutual pair of its own?  its choice's partner y must choose it back
        const int cl = choice(rl), ch = choice(rh);
        const int ppl = (int)((uint32_t)word(rl, cl) & ORX_PARTNER_POS), pph = (int)((uint32_t)word(rh, ch) & ORX_PARTNER_POS);
        const int syl = slot_of(ppl), syh = slot_of(pph);
        const int yl = ppl - syl * Bp, yh = pph - syh * Bp;
        const int4 ryl = (cl >= 0 && yl != bl) ? part[yl] : none, ryh = (ch >= 0 && yh != bh) ? part[yh] : none;
        const bool lo_taken = bl >= B || (cl >= 0 && yl != bl && choice(ryl) == syl);
        const bool hi_taken = bh >= B || (ch >= 0 && yh != bh && choice(ryh) == syh);
        bool acc = false;
        int stay = 0, mover = 0, q = 0;
        if (mutual) {
            if (same) { acc = true; stay = lo; mover = hi; q = hi; }
            else if (!lo_taken) { acc = true; stay = lo; mover = hi; q = bl; }
            else if (!hi_taken) { acc = true; stay = hi; mover = lo; q = bh; }
        }
        if (acc) {
            d.dlist[eg] = ORX_DLIST_DEAD;
            // position q processes the mover (and the mover's position the triplet that stood at q); the pairing words of the two positions
            const uint32_t s_stay = (uint32_t)(stay == ta ? sa : sb), s_mov = (uint32_t)(stay == ta ? sb : sa), m = (uint32_t)tpw - 1u;
            pword[4 * stay] = (int)(ORX_PAIR_VALID | ORX_PAIR_WRITER | ((uint32_t)q & m) | (s_stay << 4) | (s_mov << 6) | ((uint32_t)stay << 10));
            pword[4 * q] = (int)(ORX_PAIR_VALID | ((uint32_t)stay & m) | (s_mov << 4) | (s_stay << 6) | ((uint32_t)mover << 10));
            if (q != mover) pword[4 * mover] = (int)((uint32_t)q << 10);
            // (plan_swap_kernel exchanges the ids of the two positions once every flag sits on them)
            reinterpret_cast<int2*>(d.pslot)[eg] = q != mover ? make_int2(q, mover) : make_int2(-1, -1);
            npair += 1;
        } else {
            reinterpret_cast<int2*>(d.pslot)[eg] = make_int2(-1, -1);
            int32_t* wa = plan_idword(d, s, posA);
            const uint32_t id = (uint32_t)*wa;
            *wa = (int32_t)(id | (1u << 31));                                      // role 0
            *plan_idword(d, s, posB) = (int32_t)(id | (1u << 31) | (1u << 29));    // role 1
            d.dlist[eg] = id | (sa ? 0x80000000u : 0u);
            if (a.dupbits != nullptr) {
                const int bk = sa ? a.nru + (int)(id & (uint32_t)(a.nri - 1)) : (int)(id & (uint32_t)(a.nru - 1));
                const int l = (int)(id >> (sa ? a.lgi : a.lgu));
                atomicOr(a.dupbits + ((size_t)s * nb + bk) * W + (l >> 5), 1u << (l & 31));
            }
        }
    }
    // accepted pairs of the step (host: list entries - pairs = rows the apply really has): one atomic per workgroup
    __shared__ int sh_np;
    if (threadIdx.x == 0) sh_np = 0;
    __syncthreads();
    if (npair) atomicAdd(&sh_np, npair);
    __syncthreads();
    if (threadIdx.x == 0 && sh_np) atomicAdd(d.alloc + 8 * s + 7, sh_np);
}

// pairing: position q of an accepted pair processes the triplet that stood at `mover` (and the mover's position the triplet that stood
// at q).  The pairing words already say so (plan_pair_kernel); here the ids of the two records change places -- after every flag of the
// plan (duplicate flags and roles, urgent marks) has been put on them where the triplets STOOD.  A position is part of at most one such
// exchange (it can only ever be displaced for its own buddy, and a mover is never anybody's buddy-to-displace): no two threads touch
// the same record.  One thread per entry of the step's list of duplicated rows.
__global__ __launch_bounds__(256) void plan_swap_kernel(DedupArgs d) {
    const int64_t s = blockIdx.y;
    const int n = d.dcount[s];
    int4* rec = d.ids4 + s * d.pair_stride;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int2 qm = reinterpret_cast<const int2*>(d.pslot)[s * d.list_stride + e];
        if (qm.x < 0 || qm.y < 0) continue;
        const int4 a = rec[qm.x], b = rec[qm.y];
        rec[qm.x] = make_int4(b.x, b.y, b.z, a.w);
        rec[qm.y] = make_int4(a.x, a.y, a.z, b.w);
    }
}

// (round 6 measured the alternative -- no swap launch, the fused kernel follows the origin index of a moved position: -0.4 us per step at K = 20,
// below the bar: profiles/r6_plan_levers.txt)
int orx_launch_plan_swap(orx_ctx* ctx, const DedupArgs& d, int64_t kc) {
    if (d.pair_tpw < 2 || kc <= 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_DEDUP);
    ORX_LAUNCH(ctx, plan_swap_kernel, dim3((unsigned)std::max<int64_t>(4, (d.pair_stride / 4 + 255) / 256), (unsigned)kc), dim3(256), 0, d);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ------------------------------------------------------------------------------------------ host side ---
// Geometry of the plan: buckets per table (a power of two; row r -> bucket r & (n - 1)) such that a workgroup gets ~2-4 k
// references -- few references per bucket cost a workgroup's fixed work (10 M x 50 M tables in 16 384-row buckets were 3663
// workgroups per step, 16 us), many make one workgroup the whole plan's critical path (a 10 000-row table in ONE bucket:
// 131 k references, 23 us per step) -- within what the bitmaps allow (<= 2^17 rows per bucket, 2^18 above 2^27 rows).
static int plan_buckets(int64_t rows, int64_t nref_t) {
    if (rows <= 0 || nref_t <= 0) return 0;
    const int cap = rows > (1LL << 27) ? 18 : 17;
    int64_t want = std::max<int64_t>((rows + (1LL << cap) - 1) >> cap, std::min<int64_t>(nref_t / 2048, rows / 64));
    static const char* env = getenv("ORX_PLAN_SHIFT");      // experiments: at most 1 << value rows per bucket
    if (env) want = std::max<int64_t>(want, (rows + (1LL << atoi(env)) - 1) >> atoi(env));
    int n = 1;
    while (n < want && n < 8192) n *= 2;
    return n;
}
static int ilog2(int64_t n) { int l = 0; while ((1LL << l) < n) ++l; return l; }
static void plan_geometry(const DedupArgs& d, PlanArgs* a) {
    a->nref = d.nU + d.nP + d.nN;
    a->nru = plan_buckets(d.nU ? d.NU : 0, d.nU);
    a->nri = plan_buckets((d.nP + d.nN) ? d.NI : 0, d.nP + d.nN);
    a->lgu = ilog2(a->nru > 0 ? a->nru : 1); a->lgi = ilog2(a->nri > 0 ? a->nri : 1);
    const int64_t per_u = a->nru ? (d.NU + a->nru - 1) / a->nru : 1, per_i = a->nri ? (d.NI + a->nri - 1) / a->nri : 1;
    a->shift = std::max(5, ilog2(std::max(per_u, per_i)));          // bitmaps of 1 << shift bits
}

// scratch of the bucketed plan of `chunk` steps with this geometry (grow-only)
static int plan_ensure(orx_ctx* c, int64_t chunk, const PlanArgs& a, bool want_dupbits) {
    const int nb = a.nru + a.nri;
    if (orx_ensure((void**)&c->d_pl_cnt, &c->d_pl_cnt_cap, (size_t)chunk * (3 * nb + 1) * sizeof(int))) return ORX_ERR_OOM;
    if (orx_ensure((void**)&c->d_pl_list, &c->d_pl_list_cap, (size_t)chunk * a.nref * sizeof(int2))) return ORX_ERR_OOM;
    const size_t words = (size_t)nb * ((1u << a.shift) >> 5);
    if (want_dupbits && orx_ensure((void**)&c->d_dupbits, &c->d_dupbits_cap, (size_t)chunk * words * sizeof(unsigned int))) return ORX_ERR_OOM;
    return ORX_OK;
}

// pre-sizing (orx_pairwise_reserve and friends keep allocations out of a timed call): the geometries of a pairwise step
// (B user, 2 B item references), a pointwise step (B + B) and an id-list apply (B item references) over these tables
int orx_plan_buffers(orx_ctx* c, int64_t chunk, int64_t B, int64_t NU, int64_t NI, bool want_dupbits) {
    const int64_t shapes[3][3] = {{B, B, B}, {B, B, 0}, {0, B, 0}};
    for (const auto& sh : shapes) {
        DedupArgs d;
        memset(&d, 0, sizeof(d));
        d.nU = sh[0]; d.nP = sh[1]; d.nN = sh[2]; d.NU = NU; d.NI = NI;
        PlanArgs a;
        plan_geometry(d, &a);
        const int rc = plan_ensure(c, chunk, a, want_dupbits);
        if (rc != ORX_OK) return rc;
    }
    return ORX_OK;
}

// The plan of kc steps.  `d` is filled as for orx_launch_dedup (dupbits non-NULL: the bitmaps for orx_plan_urgent are kept).
// step0: the steps are number step0 .. step0 + kc - 1 of the chunk whose plan arrays the context holds (`d` already points at
// step step0 of every per-step array): a chunk is planned in pieces, each while the previous piece's steps run.
int orx_launch_plan(orx_ctx* ctx, const DedupArgs& d, int64_t kc, bool keep_dupbits, int64_t step0) {
    ProfScope ps(ctx, ORX_K_DEDUP);
    PlanArgs a;
    a.d = d;
    plan_geometry(d, &a);
    const int nb = a.nru + a.nri;
    if (nb == 0 || a.nref == 0 || kc == 0) return ORX_OK;
    ORX_ARG(a.nref < (1LL << 30) && kc < 65536, "plan: too many references per step (%lld) or steps (%lld)", (long long)a.nref, (long long)kc);
    { const int rc = plan_ensure(ctx, step0 + kc, a, keep_dupbits); if (rc != ORX_OK) return rc; }     // (a no-op after a reserve)
    const size_t words = (size_t)nb * ((1u << a.shift) >> 5);
    a.bcnt = ctx->d_pl_cnt + step0 * (3 * nb + 1); a.list = ctx->d_pl_list + step0 * a.nref;
    a.dupbits = keep_dupbits ? ctx->d_dupbits + step0 * words : nullptr;
    a.s_first = 1;
    a.tstamp = nullptr;
    static const bool timing = getenv("ORX_PLAN_TIMING") != nullptr;     // experiments: per-phase wall-clock stamps of plan_range_kernel
    if (timing) {
        ORX_HIP(hipMalloc((void**)&a.tstamp, (size_t)nb * kc * 8 * sizeof(unsigned long long)));
        ORX_HIP(hipMemsetAsync(a.tstamp, 0, (size_t)nb * kc * 8 * sizeof(unsigned long long), ctx->stream));
    }
    const char* ml = getenv("ORX_PLAN_MIN_LATE");      // experiments
    a.min_late = ml ? atoi(ml) : d.min_late;
    // The counts and cursors of a step are zeroed again by plan_range_kernel once the scatter has used them: no memset (a launch and
    // a gap at the head of every call) unless the buffer is new or was last used with another number of ranges per step.
    if (ctx->pl_cnt_clean != ctx->d_pl_cnt || ctx->pl_cnt_clean_cap != ctx->d_pl_cnt_cap || ctx->pl_cnt_nb != nb)
        ORX_HIP(hipMemsetAsync(ctx->d_pl_cnt, 0, ctx->d_pl_cnt_cap, ctx->stream));
    // The buffer counts as clean again only once plan_range_kernel (which re-zeroes what the scatter used) has been enqueued: any
    // error return between here and there leaves it marked dirty, and the next plan starts with the memset (self-healing).
    ctx->pl_cnt_clean = nullptr;
    struct CleanMark { orx_ctx* c; int nb; bool ok = false;
                       ~CleanMark() { if (ok) { c->pl_cnt_clean = c->d_pl_cnt; c->pl_cnt_clean_cap = c->d_pl_cnt_cap; c->pl_cnt_nb = nb; } } } clean_mark{ctx, nb};
    const dim3 gp((unsigned)((a.nref + PL_CHUNK - 1) / PL_CHUNK), (unsigned)kc);
    const size_t hist_bytes = (size_t)(3 * nb + 1) * sizeof(int);
    ORX_ONCE_PER_DEVICE(ctx, ORX_HIP(hipFuncSetAttribute((const void*)plan_part_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)));
    ORX_ONCE_PER_DEVICE(ctx, ORX_HIP(hipFuncSetAttribute((const void*)plan_part_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)));
    ORX_ONCE_PER_DEVICE(ctx, ORX_HIP(hipFuncSetAttribute((const void*)plan_range_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)));
    ORX_ONCE_PER_DEVICE(ctx, ORX_HIP(hipFuncSetAttribute((const void*)plan_range_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)));
    ORX_LAUNCH(ctx, (plan_part_kernel<false>), gp, dim3(PL_THREADS), hist_bytes, a);
    ORX_LAUNCH(ctx, (plan_part_kernel<true>), gp, dim3(PL_THREADS), hist_bytes, a);
    const int W = (1 << a.shift) >> 5;
    const size_t lds = (size_t)(3 * W + (W + 1) / 2) * 4 + (size_t)PL_LCNT * 4 +
                       (d.pair_tpw > 1 && W <= 4096 ? (size_t)2 * PL_PAIR_CAP * 4 : 0);      // (+ pairing: positions)
    // workgroups of 1024 threads where the previous plan met a bucket with more than 16 k references (skewed ids: the head of
    // a Zipf distribution puts 100 k of a step's 131 k item references into one range, 0.7 ms for 256 threads)
    if (ctx->plan_big) ORX_LAUNCH(ctx, plan_range_kernel<1024>, dim3((unsigned)nb, (unsigned)kc), dim3(1024), lds, a);
    else ORX_LAUNCH(ctx, plan_range_kernel<256>, dim3((unsigned)nb, (unsigned)kc), dim3(256), lds, a);
    // (a grid-stride loop over the step's list: ~0.16 B entries with uniform ids)
    if (d.pair_tpw > 1) ORX_LAUNCH(ctx, plan_pair_kernel, dim3((unsigned)std::max<int64_t>(4, (d.pair_stride / 4 + 255) / 256), (unsigned)kc), dim3(256), 0, a);
    ORX_HIP(hipGetLastError());
    clean_mark.ok = true;
    if (a.tstamp != nullptr) {
        // stamps are in ticks of the 100 MHz constant clock
        ORX_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<unsigned long long> h((size_t)nb * kc * 8);
        ORX_HIP(hipMemcpy(h.data(), a.tstamp, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        ORX_HIP(hipFree(a.tstamp));
        double sum[8] = {0}; unsigned long long t_min = ~0ull, t_max = 0; size_t live = 0;
        for (size_t w = 0; w < (size_t)nb * kc; ++w) {
            const unsigned long long* t = &h[w * 8];
            if (t[7] == 0) continue;                    // (an empty range returned early)
            ++live; t_min = std::min(t_min, t[0]); t_max = std::max(t_max, t[7]);
            for (int i = 1; i < 8; ++i) sum[i] += (double)(t[i] - t[i - 1]);
        }
        fprintf(stderr, "[orx plan timing] K=%lld, %zu workgroups, span %.1f us; mean us per phase: zero+cnt %.2f | pass1 %.2f | staging+prefix %.2f | pass2 %.2f | "
                "pairing %.2f | dupbits %.2f | list alloc %.2f\n", (long long)kc, live, (double)(t_max - t_min) * 0.01,
                sum[1] / live * 0.01, sum[2] / live * 0.01, sum[3] / live * 0.01, sum[4] / live * 0.01, sum[5] / live * 0.01, sum[6] / live * 0.01, sum[7] / live * 0.01);
    }
    return ORX_OK;
}

int orx_launch_plan_urgent(orx_ctx* ctx, const DedupArgs& d, int64_t kc, int64_t step0) {
    ProfScope ps(ctx, ORX_K_DEDUP);
    const int first = step0 > 0 ? 0 : 1;                // (step 0 of a chunk has no predecessor)
    if (kc - first < 1) return ORX_OK;
    PlanArgs a;
    a.d = d;
    plan_geometry(d, &a);
    const int nb = a.nru + a.nri;
    const size_t words = (size_t)nb * ((1u << a.shift) >> 5);
    a.bcnt = ctx->d_pl_cnt + step0 * (3 * nb + 1); a.list = ctx->d_pl_list + step0 * a.nref; a.dupbits = ctx->d_dupbits + step0 * words; a.min_late = -1;
    a.s_first = first;
    a.tstamp = nullptr;
    const int W = (1 << a.shift) >> 5;
    if (ctx->plan_big) ORX_LAUNCH(ctx, plan_urgent_kernel<1024>, dim3((unsigned)(a.nru + a.nri), (unsigned)(kc - first)), dim3(1024), (size_t)W * 4, a);
    else ORX_LAUNCH(ctx, plan_urgent_kernel<256>, dim3((unsigned)(a.nru + a.nri), (unsigned)(kc - first)), dim3(256), (size_t)W * 4, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
