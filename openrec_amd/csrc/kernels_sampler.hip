// On-device triplet sampler: the producer side of the train step
// (openrec/tf2/data/dataset.py:7-16 `_pairwise_generator`, utils.py:82-87 `next_random_record`,
// utils.py:102-116 `sample_negative_items`).  The reference draws sample by sample from CPython's
// Mersenne Twister (a sequential generator, ~3.5e5 triplets/s per process); the device sampler is
// counter-based, so sample g depends only on (seed, g):
//   positives : record perm_e(g mod R) of epoch e = g / R, where perm_e is a keyed Feistel
//               permutation of [0, R) (cycle walking) -> every record exactly once per epoch,
//               like the reference's shuffle-and-pop
//   negatives : uniform items re-drawn while they are positives of the user (membership by binary
//               search in the user's sorted CSR row), like the reference's rejection loop
#include "orx_device.h"

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// keyed permutation of [0, n): 4-round Feistel on 2*h bits (2^(2h) >= n) + cycle walking
__device__ __forceinline__ uint64_t feistel_perm(uint64_t x, uint64_t n, int h, uint64_t key) {
    const uint64_t mask = (1ull << h) - 1;
    do {
        uint64_t l = x >> h, r = x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint64_t f = mix64(r ^ (key + 0x632BE59BD9B4E019ull * (round + 1))) & mask;
            const uint64_t t = l ^ f;
            l = r; r = t;
        }
        x = (l << h) | r;
    } while (x >= n);
    return x;
}


__global__ __launch_bounds__(256) void sample_pairwise_kernel(SamplerArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const uint64_t g = (uint64_t)(a.first + i);
        const uint64_t epoch = g / (uint64_t)a.R, pos = g % (uint64_t)a.R;
        const uint64_t rec = feistel_perm(pos, (uint64_t)a.R, a.h, mix64(a.seed ^ (epoch * 0xD6E8FEB86659FD93ull)));
        const int u = a.rec_user[rec], p = a.rec_item[rec];
        const int64_t lo0 = a.ptr[u], hi0 = a.ptr[u + 1];
        int ng = 0;
        for (int attempt = 0; attempt < 256; ++attempt) {
            ng = (int)(mix64(a.seed ^ (g * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)attempt << 56) ^ 0xA5A5A5A5ull) % (uint64_t)a.total_items);
            int64_t lo = lo0, hi = hi0;                     // binary search: is ng a positive of u?
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.items[mid] < ng) lo = mid + 1; else hi = mid; }
            if (!(lo < hi0 && a.items[lo] == ng)) break;
        }
        a.uid[i] = u; a.pid[i] = p; a.nid[i] = ng;
    }
}

int orx_launch_sample_pairwise(orx_ctx* ctx, const SamplerArgs& a) {
    if (a.n == 0) return ORX_OK;
    int64_t g = (a.n + 255) / 256; if (g > 8192) g = 8192;
    ORX_LAUNCH(ctx, sample_pairwise_kernel, dim3((unsigned)g), dim3(256), 0, a);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}


// ------------------------------------------------------------------------------------------- pointwise samplers ---
// openrec/tf2/data/dataset.py:18-36 `_stratified_pointwise_generator` and :38-58 `_per_pos_stratified_pointwise_generator`,
// the producers of GMF / WRMF's train step.
//
// stratified(pos_ratio): every sample flips a coin; heads -> the next record of the shuffled epoch (label 1), tails -> a
// uniform (user, item) pair re-drawn while it is a positive (label 0).  The coin of sample g is a hash of (seed, g); the
// RANK of a positive among the positives (its place in the shuffle-and-pop order) is the number of heads before it, a
// prefix count over the whole stream: blocks count their heads, one block scans the counts on top of the sampler's running
// counter, blocks emit.  The stream is therefore sequential like the reference's generator (calls continue where the
// previous one stopped; first = 0 restarts it), and deterministic.
__device__ __forceinline__ bool strat_coin(uint64_t seed, uint64_t g, float pos_ratio) {
    // random.random() <= pos_ratio with 24 random bits
    return (float)(mix64(seed ^ (g * 0xC2B2AE3D27D4EB4Full) ^ 0x1234567ull) >> 40) * (1.0f / 16777216.0f) <= pos_ratio;
}

__device__ __forceinline__ bool is_positive(const SamplerArgs& a, int u, int item) {
    const int64_t lo0 = a.ptr[u], hi0 = a.ptr[u + 1];
    int64_t lo = lo0, hi = hi0;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.items[mid] < item) lo = mid + 1; else hi = mid; }
    return lo < hi0 && a.items[lo] == item;
}

__global__ __launch_bounds__(256) void strat_count_kernel(SamplerArgs a, float pos_ratio, int* blockcnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool head = i < a.n && strat_coin(a.seed, (uint64_t)(a.first + i), pos_ratio);
    const int c = __syncthreads_count(head);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = c;
}

// exclusive scan of the block counts on top of *counter; *counter advances by the total
__global__ __launch_bounds__(1024) void strat_scan_kernel(int* blockcnt, int nblocks, int64_t* counter, int64_t* blockbase) {
    __shared__ int64_t wave_tot[16];
    const int per = (nblocks + 1023) / 1024;
    const int i0 = threadIdx.x * per;
    int64_t mine = 0;
    for (int i = i0; i < i0 + per && i < nblocks; ++i) mine += blockcnt[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int64_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int64_t before = 0, all = 0;
    for (int k = 0; k < 16; ++k) { const int64_t t = wave_tot[k]; if (k < wave) before += t; all += t; }
    int64_t run = *counter + before + incl - mine;
    for (int i = i0; i < i0 + per && i < nblocks; ++i) { blockbase[i] = run; run += blockcnt[i]; }
    __syncthreads();
    if (threadIdx.x == 0) *counter += all;
}

__global__ __launch_bounds__(256) void strat_emit_kernel(SamplerArgs a, float pos_ratio, const int64_t* blockbase, float* label) {
    __shared__ int wave_cnt[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t g = (uint64_t)(a.first + i);
    const bool live = i < a.n;
    const bool head = live && strat_coin(a.seed, g, pos_ratio);
    const unsigned long long bal = __ballot(head);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int before = __popcll(bal & ((1ull << lane) - 1ull));
    for (int k = 0; k < wave; ++k) before += wave_cnt[k];
    if (!live) return;
    if (head) {
        const uint64_t position = (uint64_t)(blockbase[blockIdx.x] + before);
        const uint64_t epoch = position / (uint64_t)a.R, pos = position % (uint64_t)a.R;
        const uint64_t rec = feistel_perm(pos, (uint64_t)a.R, a.h, mix64(a.seed ^ (epoch * 0xD6E8FEB86659FD93ull)));
        a.uid[i] = a.rec_user[rec]; a.pid[i] = a.rec_item[rec]; label[i] = 1.0f;
    } else {
        int u = 0, item = 0;
        for (int attempt = 0; attempt < 256; ++attempt) {       // dataset.py:29-33: both ids are re-drawn while the pair is a positive
            const uint64_t r = mix64(a.seed ^ (g * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)attempt << 56) ^ 0x5A5A5A5Aull);
            u = (int)((r >> 32) % (uint64_t)a.total_users);
            item = (int)((r & 0xffffffffull) % (uint64_t)a.total_items);
            if (!is_positive(a, u, item)) break;
        }
        a.uid[i] = u; a.pid[i] = item; label[i] = 0.0f;
    }
}

// per_pos_stratified(pos_ratio): groups of 1 + nneg samples, nneg = int((1 - r) / r): a record (label 1), then the first nneg
// of `random.sample(range(total_items), nneg + 1)` that differ from the record's item (label 0; NOT checked against the user's
// other positives, dataset.py:49-57).  Fully counter-based: group q of the stream takes record perm(q), its candidates are
// the first nneg + 1 values of a keyed permutation of the items (distinct by construction).
__global__ __launch_bounds__(256) void perpos_kernel(SamplerArgs a, int nneg, int hi_items, float* label) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const uint64_t g = (uint64_t)(a.first + i);
        const uint64_t q = g / (uint64_t)(nneg + 1);
        const int slot = (int)(g % (uint64_t)(nneg + 1));
        const uint64_t epoch = q / (uint64_t)a.R, pos = q % (uint64_t)a.R;
        const uint64_t rec = feistel_perm(pos, (uint64_t)a.R, a.h, mix64(a.seed ^ (epoch * 0xD6E8FEB86659FD93ull)));
        const int u = a.rec_user[rec], p = a.rec_item[rec];
        int item = p; float lab = 1.0f;
        if (slot > 0) {
            const uint64_t key = mix64(a.seed ^ (q * 0x9FB21C651E98DF25ull) ^ 0x77ull);
            int taken = 0;
            for (int j = 0; j <= nneg; ++j) {
                const int c = (int)feistel_perm((uint64_t)j, (uint64_t)a.total_items, hi_items, key);
                if (c == p) continue;
                if (++taken == slot) { item = c; break; }
            }
            lab = 0.0f;
        }
        a.uid[i] = u; a.pid[i] = item; label[i] = lab;
    }
}

int orx_launch_sample_stratified(orx_ctx* ctx, const SamplerArgs& a, float pos_ratio, float* label, int* blockcnt, int64_t* blockbase,
                                 int64_t* counter) {
    if (a.n == 0) return ORX_OK;
    const int nblocks = (int)((a.n + 255) / 256);
    ORX_LAUNCH(ctx, strat_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, a, pos_ratio, blockcnt);
    ORX_LAUNCH(ctx, strat_scan_kernel, dim3(1), dim3(1024), 0, blockcnt, nblocks, counter, blockbase);
    ORX_LAUNCH(ctx, strat_emit_kernel, dim3((unsigned)nblocks), dim3(256), 0, a, pos_ratio, blockbase, label);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_sample_perpos(orx_ctx* ctx, const SamplerArgs& a, int nneg, float* label) {
    if (a.n == 0) return ORX_OK;
    int hi = 1;
    while ((1ll << (2 * hi)) < a.total_items) hi++;
    int64_t g = (a.n + 255) / 256; if (g > 8192) g = 8192;
    ORX_LAUNCH(ctx, perpos_kernel, dim3((unsigned)g), dim3(256), 0, a, nneg, hi, label);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}
