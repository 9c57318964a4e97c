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
