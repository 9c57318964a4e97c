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
//   plan_swap_kernel     pairing (below): moves the two triplets of an accepted pair into one wavefront, writes their pairing words
//
// PAIRING (round 4).  A row referenced exactly twice in a step costs the exact step ten row moves (two reads, two gradient
// deposits, and the apply's three reads + three writes) where a racy kernel pays four.  The order of the triplets inside a
// batch is free (the loss is a sum over the batch), so plan_range_kernel proposes, for every such row, to bring its two
// triplets into the SAME WAVEFRONT of the fused kernel: there the two lane groups exchange their gradients of the shared row by
// a cross-lane permute, one of them adds both and updates the row in place, the other does not write it -- two row moves, no
// deposit, no apply, no ready flag.  A proposal claims the two triplets and, when they do not already share a wavefront, one
// more position next to one of them (atomic exchange on a per-step claim array); a triplet takes part in at most one pair and
// a displaced neighbour in none.  Proposals that lose a claim, rows with three or more references and rows whose two references
// sit in one triplet keep the deposit / apply path.  The fused kernel sees a paired reference as a reference to a unique row.
// The reference's semantics being restated are TF's: every gradient of a step is taken on the pre-step tables and
// duplicate indices are summed before the sparse apply (tf2_examples/bpr_citeulike.py:35-38).
#include "orx_internal.h"

#include "orx_device.h"

#include <cstring>

constexpr int PL_THREADS = 256;
constexpr int PL_REFS = 8;                        // references per thread in the count / scatter kernels
constexpr int PL_CHUNK = PL_THREADS * PL_REFS;
constexpr int PL_LCNT = 2048;                     // per-row reference counters of the staging plan kept in LDS (more tri rows in a range: global counters)

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
template <int T = PL_THREADS>
__device__ __forceinline__ int plan_scan_excl(int v, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < T / 64; ++k) { const int t = wave_tot[k]; if (k < wave) before += t; all += t; }
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
                bk[k] = is_user ? (id & (a.nru - 1)) : a.nru + (id & (a.nri - 1));
                idv[k] = id; posv[k] = pos;
            }
            // the rewritten ids start as a (coalesced) copy; plan_range_kernel then touches only the duplicated references --
            // a scattered 4-byte store costs a memory transaction of its own.  0x7fffffff: out-of-range id, never a valid row
            if (SCATTER) {
                a.d.ids_out[s * a.d.flag_stride + pos] = ok ? id : 0x7fffffff;
                if (a.d.pair_tpw > 1 && j < a.d.pair_stride) {          // pairing: the step's claims and pairing words start at zero
                    a.d.claim[s * a.d.pair_stride + j] = 0;
                    a.d.pinfo[s * a.d.pair_stride + j] = 0u;
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

// pairing: claim the triplets ta and tb (positions of the step's batch) and, unless they already sit in one wavefront of
// `tpw` triplets, a free position next to one of them.  On success: `mover` goes to position `q` (q == mover: nothing moves),
// `stay` stays.  Claims are exclusive (atomic exchange on the step's claim array, zero = free); a loser releases what it took.
constexpr int PL_PAIR_CAP = 1024;                 // rows referenced exactly twice that a range can pair (their positions live in LDS)
// (every atomic below is issued before any of its results is looked at: the chain is two or three memory round trips, not six)
__device__ __forceinline__ bool pair_claim(int* claim, int B, int tpw, int ta, int tb, int& stay, int& mover, int& q) {
    if (ta == tb) return false;                       // both references in ONE triplet (p == n, ...): nothing to exchange across lane groups
    const int ca = atomicExch(claim + ta, 1), cb = atomicExch(claim + tb, 1);
    if (ca != 0 || cb != 0) {
        if (ca == 0) atomicExch(claim + ta, 0);
        if (cb == 0) atomicExch(claim + tb, 0);
        return false;
    }
    const int qa = ta & ~(tpw - 1), qb = tb & ~(tpw - 1);
    if (qa == qb) { stay = ta; mover = tb; q = tb; return true; }
    // a free position beside ta, else beside tb: the (up to) three other positions of its aligned group of four at once -- inside
    // the wavefront for every tpw -- and the first free one is kept
    const int g = tpw < 4 ? tpw : 4;
    for (int side = 0; side < 2; ++side) {
        const int self = side ? tb : ta, q0 = self & ~(g - 1);
        int r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = q0 + k;
            r[k] = (k < g && c != self && c < B) ? atomicExch(claim + c, 1) : 1;
        }
        int keep = -1;
#pragma unroll
        for (int k = 3; k >= 0; --k) if (r[k] == 0) keep = k;
        if (keep >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (r[k] == 0 && k != keep) atomicExch(claim + q0 + k, 0);
            stay = self; mover = side ? ta : tb; q = q0 + keep;
            return true;
        }
    }
    atomicExch(claim + ta, 0); atomicExch(claim + tb, 0);
    return false;
}

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
    unsigned short* dprefix16 = reinterpret_cast<unsigned short*>(lcnt + PL_LCNT);
    int* pairrow = reinterpret_cast<int*>(lcnt + PL_LCNT + (W + 1) / 2);
    int* pairpos = pairrow + PL_PAIR_CAP;
    __shared__ int wave_tot[T / 64];
    __shared__ int sh_late, sh_dense, sh_seg, list_cnt, list_base;
    const int nb = a.nru + a.nri;
    const int b = blockIdx.x;
    const int64_t s = blockIdx.y;
    const bool is_user = b < a.nru;
    const int lg = is_user ? a.lgu : a.lgi;             // local row of id: id >> lg; id of local row l: (l << lg) | bl
    const int bl = is_user ? b : b - a.nru;
    const int* cnt = a.bcnt + s * (3 * nb + 1) + 2 * nb;
    const int lo = cnt[b], n = cnt[b + 1] - lo;
    int2* ent = a.list + s * a.nref + lo;
    unsigned int* dupout = a.dupbits ? a.dupbits + ((size_t)s * nb + b) * W : nullptr;
    if (threadIdx.x == 0 && n > 8192) atomicMax(d.alloc + 8 * s + 6, n);      // (the host sizes the next plan's workgroups by it)
    if (n == 0) {                                       // nothing references this range in this step
        if (dupout) for (int w = threadIdx.x; w < W; w += T) dupout[w] = 0u;
        return;
    }
    int32_t* ids_out = d.ids_out + s * d.flag_stride;
    int2* refinfo = d.refinfo ? d.refinfo + s * d.flag_stride : nullptr;
    for (int i = threadIdx.x; i < 3 * W; i += T) pl_lds[i] = 0u;
    if (threadIdx.x == 0) { sh_late = 0; list_cnt = 0; }
    __syncthreads();
    // pass 1: bitmaps; the role of a reference among its row's references (first / second / later, by arrival)
    int late = 0;
    pl_for_each<T>(ent, n, [&](int i, int2 e) {
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
        if (role) ent[i].y = (int)((uint32_t)e.y | ((uint32_t)role << 30));       // (read back by the same thread in pass 2)
    });
    if (refinfo != nullptr && late) atomicAdd(&sh_late, late);
    __syncthreads();
    // staging plan where atomics would pile up: ranges with at least max(64, n / 512) third-or-later references
    const bool plan = refinfo != nullptr && sh_late >= (a.min_late < 0 ? (n / 512 > 64 ? n / 512 : 64) : a.min_late);
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
        if (ndup2 > 65535) ndup2 = 0;                   // (16-bit prefixes; such a range keeps the deposit path)
        if (ndup2) {
            for (int w = w0; w < w0 + per && w < W; ++w) { dprefix16[w] = (unsigned short)pre; pre += __popc(dup[w] & ~tri[w]); }
            const int np = ndup2 < PL_PAIR_CAP ? ndup2 : PL_PAIR_CAP;
            for (int i = threadIdx.x; i < 2 * np; i += T) pairpos[i] = -1;
            __syncthreads();
        }
    }
    // pass 2: rewritten ids, (dense row, rank) of the references that stage
    pl_for_each<T>(ent, n, [&](int, int2 e) {
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
                if (dn2 < PL_PAIR_CAP) {
                    pairpos[2 * dn2 + (int)(((uint32_t)e.y >> 30) & 1u)] = pos;
                    pairrow[dn2] = l;
                    return;
                }
            }
            v |= (t3 ? 2u : ((uint32_t)e.y >> 30)) << 29;
            if (t3 && ntri) {
                const int dn = (int)prefix16[l >> 5] + __popc(tw & ((1u << (l & 31)) - 1u));
                refinfo[pos] = make_int2(dense0 + dn, pl_rank(lcnt, dn));
            } else if (t3 && refinfo != nullptr) {
                refinfo[pos] = make_int2(-1, 0);
            }
            ids_out[pos] = (int32_t)v;
        }
    });
    if (ndup2) {
        // pairing phase: one thread per row referenced exactly twice.  Accepted: both references become references to a
        // unique row (plain id), the row leaves the duplicate bitmap (no list entry, no urgent marks in the next step) and
        // the pair is recorded for plan_swap_kernel.  Refused: the roles the deposit path needs.
        __syncthreads();
        const int np = ndup2 < PL_PAIR_CAP ? ndup2 : PL_PAIR_CAP;
        const int Bp = (int)d.role_stride;
        int* claim = d.claim + s * d.pair_stride;
        for (int k0 = 0; k0 < np; k0 += T) {               // (wave-uniform trip count: the ballot below)
            const int k = k0 + threadIdx.x;
            bool ok = false;
            int posA = 0, posB = 0, l = 0, stay = 0, mover = 0, q = 0, sa = 0, sb = 0, ta = 0;
            if (k < np) {
                posA = pairpos[2 * k]; posB = pairpos[2 * k + 1];
                l = pairrow[k];
                if (posA >= 0 && posB >= 0) {               // (always: a row of this class has exactly one reference of each role)
                    sa = posA / Bp; sb = posB / Bp;
                    ta = posA - sa * Bp;
                    ok = pair_claim(claim, (int)d.pair_stride, d.pair_tpw, ta, posB - sb * Bp, stay, mover, q);
                    const uint32_t id = (uint32_t)((l << lg) | bl);
                    if (ok) {
                        ids_out[posA] = (int32_t)id; ids_out[posB] = (int32_t)id;
                        atomicAnd(&dup[l >> 5], ~(1u << (l & 31)));
                    } else {
                        ids_out[posA] = (int32_t)(id | (1u << 31));                   // role 0
                        ids_out[posB] = (int32_t)(id | (1u << 31) | (1u << 29));      // role 1
                    }
                }
            }
            // one allocation per wavefront for its accepted pairs (thousands of atomics on ONE counter serialize)
            const unsigned long long okm = __ballot(ok);
            if (okm) {
                const int lane = threadIdx.x & 63, first = __ffsll((long long)okm) - 1;
                int base = 0;
                if (lane == first) base = atomicAdd(d.alloc + 8 * s + 7, __popcll(okm));
                base = __shfl(base, first);
                if (ok) {
                    const int s_stay = stay == ta ? sa : sb, s_mov = stay == ta ? sb : sa;
                    d.swaps[s * d.swap_stride + base + __popcll(okm & ((1ull << lane) - 1ull))] = make_int4(stay, mover, q, s_stay | (s_mov << 2));
                }
            }
        }
        __syncthreads();
    }
    if (dupout) for (int w = threadIdx.x; w < W; w += T) dupout[w] = dup[w];
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
    // append the duplicated rows of this range to the step's list
    int mine = 0;
    for (int w = threadIdx.x; w < W; w += T) mine += __popc(dup[w]);
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
        for (int w = threadIdx.x; w < W; w += T) {
            unsigned int m = dup[w];
            const unsigned int tw = tri[w];
            while (m) {
                const int bpos = __ffs(m) - 1;
                m &= m - 1;
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
    int32_t* ids_out = a.d.ids_out + s * a.d.flag_stride;
    pl_for_each<T>(ent, n, [&](int, int2 e) {
        const int l = e.x >> lg;
        if ((pl_lds[l >> 5] >> (l & 31)) & 1u) ids_out[e.y & 0x3fffffff] |= (1 << 28);
    });
}

// pairing: one thread per accepted pair of a step -- the triplet `mover` changes places with the one at `q` (three rewritten ids
// each, and their staging records), then both partners get their pairing word.  Runs after plan_urgent_kernel: flags travel with the ids.
__global__ __launch_bounds__(256) void plan_swap_kernel(DedupArgs d) {
    const int64_t s = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d.alloc[8 * s + 7]) return;
    const int4 r = d.swaps[s * d.swap_stride + k];
    const int stay = r.x, mover = r.y, q = r.z;
    int32_t* ids = d.ids_out + s * d.flag_stride;
    int2* refinfo = d.refinfo ? d.refinfo + s * d.flag_stride : nullptr;
    const int64_t Bp = d.role_stride;
    if (q != mover) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int32_t x = ids[sl * Bp + q], y = ids[sl * Bp + mover];
            ids[sl * Bp + q] = y; ids[sl * Bp + mover] = x;
            if (refinfo) { const int2 u = refinfo[sl * Bp + q], v = refinfo[sl * Bp + mover]; refinfo[sl * Bp + q] = v; refinfo[sl * Bp + mover] = u; }
        }
    }
    const uint32_t s_stay = (uint32_t)r.w & 3u, s_mov = ((uint32_t)r.w >> 2) & 3u, m = (uint32_t)d.pair_tpw - 1u;
    uint32_t* pinfo = d.pinfo + s * d.pair_stride;
    pinfo[stay] = ORX_PAIR_VALID | ORX_PAIR_WRITER | ((uint32_t)q & m) | (s_stay << 4) | (s_mov << 6);
    pinfo[q] = ORX_PAIR_VALID | ((uint32_t)stay & m) | (s_mov << 4) | (s_stay << 6);
}

int orx_launch_plan_swap(orx_ctx* ctx, const DedupArgs& d, int64_t kc) {
    if (d.pair_tpw < 2 || kc <= 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_DEDUP);
    ORX_LAUNCH(ctx, plan_swap_kernel, dim3((unsigned)((d.swap_stride + 255) / 256), (unsigned)kc), dim3(256), 0, d);
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
    const char* ml = getenv("ORX_PLAN_MIN_LATE");      // experiments
    a.min_late = ml ? atoi(ml) : d.min_late;
    ORX_HIP(hipMemsetAsync(a.bcnt, 0, (size_t)kc * (3 * nb + 1) * sizeof(int), ctx->stream));
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
                       (d.pair_tpw > 1 && W <= 4096 ? (size_t)((W + 1) / 2) * 4 + (size_t)3 * PL_PAIR_CAP * 4 : 0);      // (+ pairing: prefixes, rows, positions)
    // workgroups of 1024 threads where the previous plan met a bucket with more than 16 k references (skewed ids: the head of
    // a Zipf distribution puts 100 k of a step's 131 k item references into one range, 0.7 ms for 256 threads)
    if (ctx->plan_big) ORX_LAUNCH(ctx, plan_range_kernel<1024>, dim3((unsigned)nb, (unsigned)kc), dim3(1024), lds, a);
    else ORX_LAUNCH(ctx, plan_range_kernel<256>, dim3((unsigned)nb, (unsigned)kc), dim3(256), lds, a);
    ORX_HIP(hipGetLastError());
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
    const int W = (1 << a.shift) >> 5;
    if (ctx->plan_big) ORX_LAUNCH(ctx, plan_urgent_kernel<1024>, dim3((unsigned)(a.nru + a.nri), (unsigned)(kc - first)), dim3(1024), (size_t)W * 4, a);
    else ORX_LAUNCH(ctx, plan_urgent_kernel<256>, dim3((unsigned)(a.nru + a.nri), (unsigned)(kc - first)), dim3(256), (size_t)W * 4, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
