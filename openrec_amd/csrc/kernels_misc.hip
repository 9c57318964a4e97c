// Table utility kernels: initializers, Embedding gather (LatentFactor.__call__),
// LatentFactor.censor, and the Adam dense-decay sweep.  gfx950.
#include "orx_device.h"

typedef float f4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------- initializers ---
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Keras 'uniform' initializer (latent_factor.py:8-11): counter-based, so the
// value of element i depends only on (seed, i) -- identical for any launch shape.
__global__ void init_uniform_kernel(float* w, int64_t n, float lo, float hi, uint64_t seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = splitmix64(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);       // [0,1), 24 bits
        w[i] = lo + (hi - lo) * u;
    }
}

__global__ void fill_kernel(float* w, int64_t n, float v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w[i] = v;
}

static inline unsigned grid_for(int64_t n, int per_block) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int orx_launch_init_uniform(orx_ctx* ctx, float* w, int64_t n, float lo, float hi, uint64_t seed) {
    ORX_LAUNCH(ctx, init_uniform_kernel, dim3(grid_for(n, 256)), dim3(256), 0, w, n, lo, hi, seed);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_fill(orx_ctx* ctx, float* w, int64_t n, float v) {
    ORX_LAUNCH(ctx, fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, w, n, v);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ------------------------------------------------------------------ gather ---
// out[k, 0:dim] = W[ids[k], :]; out[k, dim] = bias[ids[k]] when bias != NULL.
template <bool VEC4>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                     int64_t rows, int dim, const int32_t* __restrict__ ids,
                                                     int64_t n, float* __restrict__ out, int64_t out_stride, int* err,
                                                     int skip_negative, float* __restrict__ bias_out) {
    const int per_row = VEC4 ? dim / 4 : dim;
    const int64_t total = n * per_row;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t k = i / per_row;
        const int e = (int)(i - k * per_row);
        const int r = ids[k];
        if (skip_negative && r < 0) continue;                  // padding slot of the sharded exchange
        if ((uint32_t)r >= (uint64_t)rows) { *err = 1; continue; }
        if (VEC4) {
            *reinterpret_cast<f4*>(out + k * out_stride + 4 * e) = *reinterpret_cast<const f4*>(w + (size_t)r * dim + 4 * e);
        } else {
            out[k * out_stride + e] = w[(size_t)r * dim + e];
        }
        if (bias != nullptr && e == 0) { if (bias_out) bias_out[k] = bias[r]; else out[k * out_stride + dim] = bias[r]; }
    }
}

int orx_launch_gather(orx_ctx* ctx, const float* w, const float* bias, int64_t rows, int dim,
                      const int32_t* ids, int64_t n, float* out, int64_t out_stride, int* err, int skip_negative, float* bias_out) {
    if (n == 0) return ORX_OK;
    const bool vec = (dim % 4 == 0) && (out_stride % 4 == 0) && (((uintptr_t)out) % 16 == 0);
    if (vec) {
        ORX_LAUNCH(ctx, (gather_kernel<true>), dim3(grid_for(n * (dim / 4), 256)), dim3(256), 0, w, bias, rows, dim, ids, n, out, out_stride, err, skip_negative, bias_out);
    } else {
        ORX_LAUNCH(ctx, (gather_kernel<false>), dim3(grid_for(n * dim, 256)), dim3(256), 0, w, bias, rows, dim, ids, n, out, out_stride, err, skip_negative, bias_out);
    }
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// ------------------------------------------------------------------ censor ---
// LatentFactor.censor (latent_factor.py:17-23).  tf.unique makes the update
// once per DISTINCT id; the dedup kernel (kernels_pairwise.hip, first_only mode)
// clears dflag for exactly one reference of every distinct row: that one applies.
__device__ __forceinline__ float wave_sum_f(float x) {
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

struct CensorSeg { float* w; const unsigned char* dflag; const int32_t* ids; int64_t n; int64_t rows; };

// One wavefront per id of segment A, then of segment B (two independent tables in one launch).
__global__ __launch_bounds__(256) void censor_apply_kernel(CensorSeg A, CensorSeg B, int dim, float min_norm, int* err) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    const int64_t total = A.n + B.n;
    for (int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < total; k += stride) {
        const bool inA = k < A.n;
        const CensorSeg& S = inA ? A : B;
        const int64_t i = inA ? k : k - A.n;
        const int r = S.ids[i] & 0x7fffffff;
        if ((uint32_t)S.ids[i] >= (uint64_t)S.rows) { if (lane == 0) *err = 1; continue; }
        if (S.dflag[i]) continue;                             // another reference of this row applies
        float* row = S.w + (size_t)r * dim;
        float s = 0.0f;
        for (int e = lane; e < dim; e += 64) s += row[e] * row[e];
        const float den = fmaxf(sqrtf(wave_sum_f(s)), min_norm);   // tf.norm, maximum(norm, 0.1)
        for (int e = lane; e < dim; e += 64) row[e] = row[e] / den;
    }
}

int orx_launch_censor(orx_ctx* ctx, float* w, const unsigned char* dflag, int64_t rows, int dim, const int32_t* ids,
                      int64_t n, float min_norm, int* err) {
    if (n == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_CENSOR);
    CensorSeg A{w, dflag, ids, n, rows}, B{nullptr, nullptr, nullptr, 0, 0};
    ORX_LAUNCH(ctx, censor_apply_kernel, dim3(grid_for(n, 4)), dim3(256), 0, A, B, dim, min_norm, err);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_censor2(orx_ctx* ctx, float* wA, const unsigned char* fA, int64_t rowsA, const int32_t* idsA, int64_t nA,
                       float* wB, const unsigned char* fB, int64_t rowsB, const int32_t* idsB, int64_t nB,
                       int dim, float min_norm) {
    if (nA + nB == 0) return ORX_OK;
    ProfScope ps(ctx, ORX_K_CENSOR);
    CensorSeg A{wA, fA, idsA, nA, rowsA}, B{wB, fB, idsB, nB, rowsB};
    ORX_LAUNCH(ctx, censor_apply_kernel, dim3(grid_for(nA + nB, 4)), dim3(256), 0, A, B, dim, min_norm, ctx->d_err);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// -------------------------------------------------------------- Adam sweep ---
// keras.optimizers.Adam._resource_apply_sparse (TF 2.0.x): m and v decay over
// the WHOLE table, the summed sparse gradient is added on its rows, and the
// update var -= lr_t * m / (sqrt(v) + eps) also covers the whole table.
// gsum holds the per-row summed gradient of this step (zero elsewhere) and is
// re-zeroed here.
__global__ __launch_bounds__(256) void adam_sweep_kernel(float* w, float* m, float* v, float* gsum, int64_t n,
                                                         float lr_t, float b1, float b2, float eps) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g = gsum[i];
        if (g != 0.0f) gsum[i] = 0.0f;
        float wi = w[i], mi = m[i], vi = v[i];
        adam_elem(wi, mi, vi, g, lr_t, b1, b2, eps);
        m[i] = mi;
        v[i] = vi;
        w[i] = wi;
    }
}

int orx_launch_adam_sweep(orx_ctx* ctx, float* w, float* m, float* v, float* gsum, int64_t n,
                          float lr_t, float b1, float b2, float eps) {
    ProfScope ps(ctx, ORX_K_SWEEP);
    ORX_LAUNCH(ctx, adam_sweep_kernel, dim3(grid_for(n, 256)), dim3(256), 0, w, m, v, gsum, n, lr_t, b1, b2, eps);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// lazy Adam: bring every row of a table up to optimizer step t_end (replaying its gradient-free steps), then
// mark the rows current.  Two kernels: all elements of a row read last[row] before it changes.
__global__ __launch_bounds__(256) void adam_flush_kernel(float* w, float* m, float* v, const int* last, int64_t n, int dim, int t_end,
                                                         const float* lrt, float b1, float b2, float eps, AdamCFParams cf) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int from = last[i / dim];
        if (from >= t_end) continue;
        float wi = w[i], mi = m[i], vi = v[i];
        adam_replay1<false>(wi, mi, vi, from, t_end, lrt, b1, b2, eps, (1.0f - sqrtf(b2)) <= 1e-3f, cf);     // closed form, or bounded loop: see orx_device.h
        w[i] = wi; m[i] = mi; v[i] = vi;
    }
}

__global__ void fill_int_kernel(int* p, int64_t n, int v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

int orx_launch_fill_int(orx_ctx* ctx, int* p, int64_t n, int v) {
    if (n == 0) return ORX_OK;
    ORX_LAUNCH(ctx, fill_int_kernel, dim3(grid_for(n, 256)), dim3(256), 0, p, n, v);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

int orx_launch_adam_flush(orx_ctx* ctx, float* w, float* m, float* v, int* last, int64_t rows, int dim, int t_end, const float* lrt,
                          float b1, float b2, float eps, AdamCFParams cf) {
    ProfScope ps(ctx, ORX_K_SWEEP);
    const int64_t n = rows * dim;
    if (n == 0) return ORX_OK;
    ORX_LAUNCH(ctx, adam_flush_kernel, dim3(grid_for(n, 256)), dim3(256), 0, w, m, v, last, n, dim, t_end, lrt, b1, b2, eps, cf);
    ORX_HIP(hipGetLastError());
    return orx_launch_fill_int(ctx, last, rows, t_end);
}

// ---- the streaming-copy yardstick (the fraction of the MEASURED streaming rate beside the fraction of the
// 8 TB/s peak).  One float4 per thread, whole buffer in one launch: the fastest form scratch/copy_bw.hip found on this part
// (6.1-6.3 TB/s read + write at 1 GiB; block-contiguous and grid-stride loops reach 5.0-5.8).
typedef float orx_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_f4_kernel(const orx_f4* __restrict__ src, orx_f4* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// bytes: size of EACH of the two buffers (allocated and freed here); *gbps_out = 2 * bytes / the mean launch time over `reps`
// launches (HIP events on the context's stream, three untimed launches first), in GB/s (1e9 bytes per second)
extern "C" int orx_copy_bandwidth(orx_ctx* c, int64_t bytes, int32_t reps, double* gbps_out) {
    ORX_ARG(c && gbps_out && bytes >= 4096 && bytes % 16 == 0 && bytes / 16 / 256 < (1LL << 31) && reps > 0 && reps <= 1000, "orx_copy_bandwidth: bad argument");
    ORX_HIP(hipSetDevice(c->device));
    orx_f4 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void**)&a, (size_t)bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&b, (size_t)bytes);
    if (e == hipSuccess) e = hipMemsetAsync(a, 1, (size_t)bytes, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, (size_t)bytes, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.f;
    if (e == hipSuccess) {
        const size_t n = (size_t)bytes / 16;
        const unsigned grid = (unsigned)((n + 255) / 256);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_f4_kernel, dim3(grid), dim3(256), 0, c->stream, a, b, n);
        e = hipEventRecord(e0, c->stream);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy_f4_kernel, dim3(grid), dim3(256), 0, c->stream, a, b, n);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    ORX_HIP(e);
    *gbps_out = ms > 0.f ? 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9 : 0.0;
    return ORX_OK;
}
