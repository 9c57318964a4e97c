// C-ABI entry points (include/openrec_hip.h): contexts, tables, optimizers and
// the host-side sequencing of the train-step kernels.  No torch, no CPU
// fallback: every entry point needs a usable HIP device.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "orx_internal.h"

#include <functional>

// ----------------------------------------------------------------- errors ---
static thread_local char g_err[1024] = "";

// ORX_HOST_TIMING=1 (experiments): wall-clock stamps of the host's way through a train-step call
#include <chrono>
static bool host_timing() { static const bool on = getenv("ORX_HOST_TIMING") != nullptr; return on; }
static thread_local double g_ht[16]; static thread_local const char* g_htn[16]; static thread_local int g_htc = 0;
static void ht_mark(const char* what) {
    if (!host_timing() || g_htc >= 16) return;
    g_ht[g_htc] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); g_htn[g_htc++] = what;
}
static void ht_dump() {
    if (!host_timing() || g_htc == 0) return;
    fprintf(stderr, "[orx host]");
    for (int i = 1; i < g_htc; ++i) fprintf(stderr, " %s +%.1f", g_htn[i], g_ht[i] - g_ht[i - 1]);
    fprintf(stderr, " | total %.1f us\n", g_ht[g_htc - 1] - g_ht[0]);
    g_htc = 0;
}

void orx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* orx_last_error(void) { return g_err; }
extern "C" int orx_version(void) { return 100; }

// ---------------------------------------------------------------- helpers ---
int orx_ensure(void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p != nullptr) return ORX_OK;
    if (*p) { ORX_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    size_t want = bytes < 4096 ? 4096 : bytes;
    ORX_HIP(hipMalloc(p, want));
    *cap = want;
    return ORX_OK;
}

#define ENSURE(ptr, cap, bytes)                                                        \
    do {                                                                               \
        int _rc = orx_ensure((void**)&(ptr), &(cap), (bytes));                         \
        if (_rc != ORX_OK) return _rc;                                                 \
    } while (0)

#define CHECK(call)                                                                    \
    do {                                                                               \
        int _rc = (call);                                                              \
        if (_rc != ORX_OK) return _rc;                                                 \
    } while (0)

void orx_prof_begin(orx_ctx* ctx, int kid) {
    hipEvent_t e0, e1;
    ctx->cur_e0 = ctx->cur_e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
    ctx->prof_slot[kid].ev.push_back(e0);
    ctx->prof_slot[kid].ev.push_back(e1);
    ctx->prof_order.push_back(kid);
    ctx->cur_e0 = e0; ctx->cur_e1 = e1;
}

void orx_prof_end(orx_ctx* ctx, int kid) {
    (void)kid;
    ctx->cur_e0 = ctx->cur_e1 = nullptr;
}

static int prof_collect(orx_ctx* ctx) {
    // ORX_PROF_TIMELINE=1 (experiments): begin / duration of every profiled launch since the last collection, in launch order, relative
    // to the first one's begin -- the gaps between launches without a tracing tool in the way
    static const bool timeline = getenv("ORX_PROF_TIMELINE") != nullptr;
    if (timeline && !ctx->prof_order.empty()) {
        size_t cur[ORX_K_NUM] = {0};
        hipEvent_t first = nullptr;
        float prev_end = 0.f;
        for (int kid : ctx->prof_order) {
            auto& s = ctx->prof_slot[kid];
            const size_t i = cur[kid]; cur[kid] += 2;
            if (i + 1 >= s.ev.size()) break;
            if (hipEventSynchronize(s.ev[i + 1]) != hipSuccess) break;
            if (!first) first = s.ev[i];
            float t0 = 0.f, dur = 0.f;
            if (hipEventElapsedTime(&t0, first, s.ev[i]) != hipSuccess || hipEventElapsedTime(&dur, s.ev[i], s.ev[i + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
            fprintf(stderr, "[orx timeline] %9.1f us  +%6.1f gap  %7.1f us  class %d\n", t0 * 1e3, (t0 - prev_end) * 1e3, dur * 1e3, kid);
            prev_end = t0 + dur;
        }
    }
    ctx->prof_order.clear();
    for (int k = 0; k < ORX_K_NUM; ++k) {
        auto& s = ctx->prof_slot[k];
        for (size_t i = 0; i + 1 < s.ev.size(); i += 2) {
            float ms = 0.f;
            ORX_HIP(hipEventSynchronize(s.ev[i + 1]));
            ORX_HIP(hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
            s.total_ms += ms;
            s.launches += 1;
            hipEventDestroy(s.ev[i]);
            hipEventDestroy(s.ev[i + 1]);
        }
        s.ev.clear();
    }
    return ORX_OK;
}

// ---------------------------------------------------------------- context ---
extern "C" int orx_ctx_create(int device, void* stream, orx_ctx** out) {
    ORX_ARG(out != nullptr, "orx_ctx_create: out is NULL");
    int ndev = 0;
    ORX_HIP(hipGetDeviceCount(&ndev));
    ORX_ARG(device >= 0 && device < ndev, "orx_ctx_create: device %d out of range (%d devices)", device, ndev);
    ORX_HIP(hipSetDevice(device));
    orx_ctx* c = new orx_ctx();
    c->device = device;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { ORX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    ORX_HIP(hipMalloc((void**)&c->d_err, sizeof(int)));
    ORX_HIP(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
    hipDeviceProp_t prop;
    ORX_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount;
    *out = c;
    return ORX_OK;
}

extern "C" int orx_ctx_destroy(orx_ctx* c) {
    if (!c) return ORX_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    prof_collect(c);
    hipFree(c->d_err); hipFree(c->d_ids); hipFree(c->d_lab); hipFree(c->d_dflag); hipFree(c->d_evalbits); hipFree(c->d_ids2); hipFree(c->d_roles); hipFree(c->d_cflag); hipFree(c->d_dupbits); hipFree(c->d_dlist);
    hipFree(c->d_dcount); hipFree(c->d_refinfo); hipFree(c->d_tricnt); hipFree(c->d_segstart); hipFree(c->d_alloc); hipFree(c->d_dseg); hipFree(c->d_dcnt); hipFree(c->d_chunks); hipFree(c->d_part); hipFree(c->d_partb); hipFree(c->d_stage); hipFree(c->d_stageb); hipFree(c->d_partial); hipFree(c->d_loss); hipFree(c->d_tmp);
    hipFree(c->d_wpart); hipFree(c->d_pl_cnt); hipFree(c->d_pl_list);
    hipFree(c->d_partner); hipFree(c->d_pslot); hipFree(c->d_ids4);
    hipFree(c->d_sort[0]); hipFree(c->d_sort[1]); hipFree(c->d_sort_hist); hipFree(c->d_csr_part[0]); hipFree(c->d_csr_part[1]); hipFree(c->d_splitk);
    if (c->h_plan) hipHostFree(c->h_plan);
    if (c->plan_ev) hipEventDestroy(c->plan_ev);
    if (c->stats_ev) hipEventDestroy(c->stats_ev);
    for (int k = 0; k < 2; ++k) { if (c->pipe_cnt[k]) hipEventDestroy(c->pipe_cnt[k]); if (c->pipe_done[k]) hipEventDestroy(c->pipe_done[k]); }
    if (c->plan_stream) hipStreamDestroy(c->plan_stream);
    if (c->wait_ev) hipEventDestroy(c->wait_ev);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return ORX_OK;
}

extern "C" int orx_synchronize(orx_ctx* c) {
    ORX_ARG(c, "orx_synchronize: NULL context");
    ORX_HIP(hipStreamSynchronize(c->stream));
    return ORX_OK;
}

extern "C" int orx_ctx_wait_stream(orx_ctx* c, void* producer_stream) {
    ORX_ARG(c, "orx_ctx_wait_stream: NULL context");
    if ((hipStream_t)producer_stream == c->stream) return ORX_OK;
    ORX_HIP(hipSetDevice(c->device));
    // nothing pending on the producer's stream: whatever it produced is there already, and the context's stream need not carry a
    // cross-stream wait (a barrier packet and its latency at the head of the next call: a few us of a K = 20 call)
    const hipError_t q = hipStreamQuery((hipStream_t)producer_stream);
    if (q == hipSuccess) return ORX_OK;
    if (q != hipErrorNotReady) ORX_HIP(q);
    (void)hipGetLastError();
    if (!c->wait_ev) ORX_HIP(hipEventCreateWithFlags(&c->wait_ev, hipEventDisableTiming));
    ORX_HIP(hipEventRecord(c->wait_ev, (hipStream_t)producer_stream));
    ORX_HIP(hipStreamWaitEvent(c->stream, c->wait_ev, 0));
    return ORX_OK;
}

extern "C" int orx_check_index_error(orx_ctx* c) {
    ORX_ARG(c, "orx_check_index_error: NULL context");
    int flag = 0;
    ORX_HIP(hipMemcpyAsync(&flag, c->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    ORX_HIP(hipStreamSynchronize(c->stream));
    if (flag) {
        ORX_HIP(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream));
        orx_set_error("id out of range: an index in the batch is < 0 or >= the table's row count");
        return ORX_ERR_INDEX;
    }
    return ORX_OK;
}

// ----------------------------------------------------------------- tables ---
static int table_new(orx_ctx* ctx, void* ptr, int64_t rows, int32_t dim, orx_table** out) {
    ORX_ARG(ctx && out, "table: NULL context/out");
    ORX_ARG(rows > 0 && dim > 0, "table: rows (%lld) and dim (%d) must be positive", (long long)rows, dim);
    ORX_ARG(rows <= 0x7fffffffLL, "table: rows must fit int32 ids");
    ORX_HIP(hipSetDevice(ctx->device));
    orx_table* t = new orx_table();
    t->ctx = ctx; t->rows = rows; t->dim = dim;
    if (ptr) { t->w = (float*)ptr; t->owned = false; }
    else {
        hipError_t e = hipMalloc((void**)&t->w, (size_t)rows * dim * sizeof(float));
        if (e != hipSuccess) {
            delete t;
            orx_set_error("table: hipMalloc of %lld x %d fp32 failed: %s", (long long)rows, dim, hipGetErrorString(e));
            return ORX_ERR_OOM;
        }
    }
    *out = t;
    return ORX_OK;
}

extern "C" int orx_table_create(orx_ctx* ctx, int64_t rows, int32_t dim, orx_table** out) {
    return table_new(ctx, nullptr, rows, dim, out);
}

extern "C" int orx_table_wrap(orx_ctx* ctx, void* device_ptr, int64_t rows, int32_t dim, orx_table** out) {
    ORX_ARG(device_ptr, "orx_table_wrap: NULL device pointer");
    return table_new(ctx, device_ptr, rows, dim, out);
}

extern "C" int orx_table_destroy(orx_table* t) {
    if (!t) return ORX_OK;
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
    for (orx_opt* o : t->ctx->opts) {       // optimizer state of this table goes with it (and never meets a new table at this address)
        auto it = o->slots.find(t);
        if (it == o->slots.end()) continue;
        hipFree(it->second.s0); hipFree(it->second.s1); hipFree(it->second.last);
        o->slots.erase(it);
    }
    if (t->owned) hipFree(t->w);
    hipFree(t->gsum);
    hipFree(t->gsum2);
    hipFree(t->ready);
    hipFree(t->side);
    delete t;
    return ORX_OK;
}

extern "C" int64_t orx_table_rows(const orx_table* t) { return t ? t->rows : -1; }
extern "C" int32_t orx_table_dim(const orx_table* t) { return t ? t->dim : -1; }
extern "C" void* orx_table_device_ptr(const orx_table* t) {
    if (t == nullptr) return nullptr;
    orx_table_sync(const_cast<orx_table*>(t));        // the raw pointer escapes: rows must be current
    return (void*)t->w;
}

int orx_table_scratch(orx_table* t, bool second) {
    ORX_HIP(hipSetDevice(t->ctx->device));
    const size_t bytes = (size_t)t->rows * t->dim * sizeof(float);
    if (!t->gsum) {
        ORX_HIP(hipMalloc((void**)&t->gsum, bytes));
        ORX_HIP(hipMemsetAsync(t->gsum, 0, bytes, t->ctx->stream));
    }
    if (second && !t->gsum2) {
        ORX_HIP(hipMalloc((void**)&t->gsum2, bytes));
        ORX_HIP(hipMemsetAsync(t->gsum2, 0, bytes, t->ctx->stream));
    }
    if (second && !t->ready) {
        ORX_HIP(hipMalloc((void**)&t->ready, (size_t)t->rows * sizeof(int)));
        ORX_HIP(hipMemsetAsync(t->ready, 0, (size_t)t->rows * sizeof(int), t->ctx->stream));
    }
    return ORX_OK;
}

int orx_table_side(orx_table* t) {
    if (t->side) return ORX_OK;
    ORX_HIP(hipSetDevice(t->ctx->device));
    ORX_HIP(hipMalloc((void**)&t->side, (size_t)t->rows * 2 * sizeof(int)));
    ORX_HIP(hipMemsetAsync(t->side, 0, (size_t)t->rows * 2 * sizeof(int), t->ctx->stream));
    return ORX_OK;
}

extern "C" int orx_table_init_uniform(orx_table* t, float lo, float hi, uint64_t seed) {
    if (t) CHECK(orx_table_sync(t));
    ORX_ARG(t, "orx_table_init_uniform: NULL table");
    t->version += 1;
    ORX_HIP(hipSetDevice(t->ctx->device));
    return orx_launch_init_uniform(t->ctx, t->w, t->rows * t->dim, lo, hi, seed);
}

extern "C" int orx_table_fill(orx_table* t, float value) {
    if (t) CHECK(orx_table_sync(t));
    ORX_ARG(t, "orx_table_fill: NULL table");
    t->version += 1;
    ORX_HIP(hipSetDevice(t->ctx->device));
    return orx_launch_fill(t->ctx, t->w, t->rows * t->dim, value);
}

static int rows_copy(orx_ctx* ctx, float* dev, int64_t rows, int32_t dim, int64_t row0, int64_t nrows,
                     float* host, bool to_host) {
    ORX_ARG(host || nrows == 0, "table copy: NULL host buffer");
    ORX_ARG(row0 >= 0 && nrows >= 0 && row0 + nrows <= rows, "table copy: rows [%lld, %lld) outside [0, %lld)",
            (long long)row0, (long long)(row0 + nrows), (long long)rows);
    if (nrows == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)nrows * dim * sizeof(float);
    float* d = dev + (size_t)row0 * dim;
    if (to_host) ORX_HIP(hipMemcpyAsync(host, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    else ORX_HIP(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    ORX_HIP(hipStreamSynchronize(ctx->stream));
    return ORX_OK;
}

extern "C" int orx_table_read(orx_table* t, int64_t row0, int64_t nrows, float* host_dst) {
    if (t) CHECK(orx_table_sync(t));
    ORX_ARG(t, "orx_table_read: NULL table");
    return rows_copy(t->ctx, t->w, t->rows, t->dim, row0, nrows, host_dst, true);
}

extern "C" int orx_table_write(orx_table* t, int64_t row0, int64_t nrows, const float* host_src) {
    if (t) CHECK(orx_table_sync(t));
    ORX_ARG(t, "orx_table_write: NULL table");
    t->version += 1;
    return rows_copy(t->ctx, t->w, t->rows, t->dim, row0, nrows, (float*)host_src, false);
}

// upload host ids into the context's staging buffer at element offset `off`
int stage_ids(orx_ctx* c, const int32_t* host, int64_t n, int64_t off) {
    ORX_HIP(hipMemcpyAsync(c->d_ids + off, host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    return ORX_OK;
}

extern "C" int orx_table_gather(orx_table* t, const int32_t* ids, int64_t n, float* out, int flags) {
    if (t) CHECK(orx_table_sync(t));
    ORX_ARG(t && (n == 0 || (ids && out)), "orx_table_gather: NULL argument");
    if (n == 0) return ORX_OK;
    orx_ctx* c = t->ctx;
    ORX_HIP(hipSetDevice(c->device));
    if (flags & ORX_IDS_DEVICE) {
        CHECK(orx_launch_gather(c, t->w, nullptr, t->rows, t->dim, ids, n, out, t->dim, c->d_err));
        return ORX_OK;
    }
    ENSURE(c->d_ids, c->d_ids_cap, (size_t)n * sizeof(int32_t));
    ENSURE(c->d_tmp, c->d_tmp_cap, (size_t)n * t->dim * sizeof(float));
    CHECK(stage_ids(c, ids, n, 0));
    CHECK(orx_launch_gather(c, t->w, nullptr, t->rows, t->dim, c->d_ids, n, c->d_tmp, t->dim, c->d_err));
    ORX_HIP(hipMemcpyAsync(out, c->d_tmp, (size_t)n * t->dim * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return orx_check_index_error(c);
}

// censor: dflag[i] = 0 for exactly one reference of every distinct row, 1 for the others
static int dedup_single(orx_ctx* c, const int32_t* d_ids, int64_t n, int64_t rows) {
    ENSURE(c->d_dflag, c->d_dflag_cap, (size_t)n);
    DedupArgs d;
    memset(&d, 0, sizeof(d));
    d.uid = d_ids; d.pid = d_ids; d.nid = d_ids; d.id_stride = n;
    d.dflag = c->d_dflag; d.flag_stride = n;
    d.nU = n; d.nP = 0; d.nN = 0; d.NU = rows; d.NI = 0; d.nbu = orx_dedup_buckets(rows); d.nbi = 0; d.first_only = 1;
    return orx_launch_dedup(c, d, 1);
}

extern "C" int orx_table_censor(orx_table* t, const int32_t* ids, int64_t n, float min_norm, int flags) {
    if (t) CHECK(orx_table_sync(t));
    ORX_ARG(t && (n == 0 || ids), "orx_table_censor: NULL argument");
    if (n == 0) return ORX_OK;
    orx_ctx* c = t->ctx;
    ORX_HIP(hipSetDevice(c->device));
    const int32_t* d = ids;
    if (!(flags & ORX_IDS_DEVICE)) {
        ENSURE(c->d_ids, c->d_ids_cap, (size_t)n * sizeof(int32_t));
        CHECK(stage_ids(c, ids, n, 0));
        d = c->d_ids;
    }
    CHECK(dedup_single(c, d, n, t->rows));
    CHECK(orx_launch_censor(c, t->w, c->d_dflag, t->rows, t->dim, d, n, min_norm, c->d_err));
    if (!(flags & ORX_IDS_DEVICE)) return orx_check_index_error(c);
    return ORX_OK;
}

// ------------------------------------------------------------- optimizers ---
extern "C" int orx_opt_create(orx_ctx* ctx, int kind, float lr, float p0, float p1, float p2, orx_opt** out) {
    ORX_ARG(ctx && out, "orx_opt_create: NULL context/out");
    ORX_ARG(kind == ORX_SGD || kind == ORX_ADAGRAD || kind == ORX_ADAM, "orx_opt_create: unknown optimizer kind %d", kind);
    orx_opt* o = new orx_opt();
    o->ctx = ctx; o->kind = kind; o->lr = lr; o->p0 = p0; o->p1 = p1; o->p2 = p2;
    ctx->opts.push_back(o);
    *out = o;
    return ORX_OK;
}

extern "C" int orx_opt_destroy(orx_opt* o) {
    if (!o) return ORX_OK;
    hipSetDevice(o->ctx->device);
    hipStreamSynchronize(o->ctx->stream);
    for (auto& kv : o->slots) {             // (keys are live tables: a destroyed table erases its entry)
        if (kv.first->lazy == o) orx_table_sync(kv.first);       // the table outlives the optimizer: finish its rows
        hipFree(kv.second.s0); hipFree(kv.second.s1); hipFree(kv.second.last);
    }
    hipFree(o->d_lrt); hipFree(o->d_lrv);
    auto& live = o->ctx->opts;
    live.erase(std::remove(live.begin(), live.end(), o), live.end());
    delete o;
    return ORX_OK;
}

extern "C" int orx_opt_set_lr(orx_opt* o, float lr) {
    ORX_ARG(o, "orx_opt_set_lr: NULL optimizer");
    if (lr != o->lr && (int64_t)o->h_lrt.size() > o->t + 1) {      // steps already taken keep the rate they were taken with
        o->h_lrt.resize((size_t)o->t + 1);
        o->lrt_uploaded = std::min<int64_t>(o->lrt_uploaded, o->t + 1);
    }
    // (closed-form replay: the moments of the last J steps looked ahead at rates that now change)
    if (lr != o->lr) o->lrv_done = std::min<int64_t>(o->lrv_done, std::max<int64_t>(0, o->t + 1 - ORX_ADAM_CF_TERMS));
    o->lr = lr;
    return ORX_OK;
}

extern "C" int orx_opt_get_step(orx_opt* o, int64_t* step_out) {
    ORX_ARG(o && step_out, "orx_opt_get_step: NULL argument");
    *step_out = o->t;
    return ORX_OK;
}

// Keras' Adam updates only the variables handed to apply_gradients.  The lazily-applied form replays every increment of
// the step counter as a gradient-free step of a lazy table's rows, so before the counter advances for a step over `keep`,
// every OTHER table that is lazy under `o` (a second model sharing the optimizer) is brought up to date and leaves the
// lazy set: it takes no decay for steps it is not part of (and re-enters, stamped with the then-current step, with its
// own next step).
int orx_opt_isolate(orx_opt* o, orx_table* const* keep, int n_keep) {
    if (o->kind != ORX_ADAM) return ORX_OK;
    for (auto& kv : o->slots) {
        orx_table* t = kv.first;
        if (t->lazy != o) continue;
        bool kept = false;
        for (int i = 0; i < n_keep; ++i) kept = kept || keep[i] == t;
        if (!kept) CHECK(orx_table_sync(t));
    }
    return ORX_OK;
}

extern "C" int orx_opt_set_step(orx_opt* o, int64_t step) {
    ORX_ARG(o && step >= 0 && step < 0x7fffffff, "orx_opt_set_step: NULL optimizer or step out of range");
    // rows pending under the old counter are finished under it; nothing is replayed across the jump, so the cache of
    // per-step rates (indexed by absolute step, filled with the rate in force when a step was taken) starts afresh
    for (auto& kv : o->slots)
        if (kv.first->lazy == o) CHECK(orx_table_sync(kv.first));
    if (step != o->t) { o->h_lrt.clear(); o->lrt_uploaded = 0; }
    o->t = step;
    return ORX_OK;
}

extern "C" int orx_opt_advance(orx_opt* o, orx_table* const* tables, int32_t n_tables) {
    ORX_ARG(o && (n_tables <= 0 || tables), "orx_opt_advance: NULL optimizer or table list");
    ORX_ARG(o->t + 1 < 0x7fffffff, "orx_opt_advance: step counter overflow");
    if (n_tables >= 0) CHECK(orx_opt_isolate(o, tables, n_tables));      // (< 0: the step is over every table the optimizer holds)
    o->t += 1;
    return ORX_OK;
}

int orx_opt_slots(orx_opt* o, orx_table* t, OptSlots* out) {
    auto it = o->slots.find(t);
    if (it != o->slots.end()) { *out = it->second; return ORX_OK; }
    OptSlots s;
    const size_t n = (size_t)t->rows * t->dim;
    ORX_HIP(hipSetDevice(o->ctx->device));
    if (o->kind == ORX_ADAGRAD) {
        ORX_HIP(hipMalloc((void**)&s.s0, n * sizeof(float)));
        CHECK(orx_launch_fill(o->ctx, s.s0, (int64_t)n, o->p0));      // initial_accumulator_value
    } else if (o->kind == ORX_ADAM) {
        ORX_HIP(hipMalloc((void**)&s.s0, n * sizeof(float)));
        ORX_HIP(hipMalloc((void**)&s.s1, n * sizeof(float)));
        ORX_HIP(hipMemsetAsync(s.s0, 0, n * sizeof(float), o->ctx->stream));
        ORX_HIP(hipMemsetAsync(s.s1, 0, n * sizeof(float), o->ctx->stream));
    }
    o->slots[t] = s;
    *out = s;
    return ORX_OK;
}

bool orx_adam_cf_ok(const orx_opt* o) {
    static const bool off = getenv("ORX_ADAM_NO_CF") != nullptr;
    return !off && o->kind == ORX_ADAM && o->p0 > 0.f && o->p0 <= 0.95f && o->p1 < 1.0f && (1.0f - sqrtf(o->p1)) <= 1e-3f;
}

AdamCFParams orx_adam_cf_params(const orx_opt* o) {
    AdamCFParams p; p.lrv = nullptr; p.delta = p.lb1 = p.lb2 = 0.f;
    if (orx_adam_cf_ok(o) && o->d_lrv != nullptr) {
        p.lrv = reinterpret_cast<const float4*>(o->d_lrv);
        p.delta = (float)(-0.5 * std::log((double)o->p1)); p.lb1 = (float)std::log2((double)o->p0); p.lb2 = (float)std::log2((double)o->p1);
    }
    return p;
}

// lr_t of steps 1..upto on the device (host mirror keeps every value ever used: a row may replay old steps)
int orx_adam_lrt(orx_opt* o, int64_t upto) {
    if ((int64_t)o->h_lrt.size() < upto + 1) {
        const double b1 = o->p0, b2 = o->p1;
        const size_t old = o->h_lrt.size();
        o->h_lrt.resize((size_t)upto + 1);
        for (size_t k = old; k <= (size_t)upto; ++k)
            o->h_lrt[k] = k == 0 ? 0.f : (float)(o->lr * std::sqrt(1.0 - std::pow(b2, (double)k)) / (1.0 - std::pow(b1, (double)k)));
    }
    if (orx_adam_cf_ok(o) && o->lrv_done < upto + 1) {
        // V_q[k] = sum_{j=1..J} b1^j j^q lr_{k+j} in double; lr of steps beyond the table: the rate they WILL have if the learning rate stays
        // (orx_opt_set_lr takes the entries that looked ahead of a change back)
        const double b1 = o->p0, b2 = o->p1;
        auto L = [&](int64_t k) -> double {
            return k < (int64_t)o->h_lrt.size() ? (double)o->h_lrt[(size_t)k] : o->lr * std::sqrt(1.0 - std::pow(b2, (double)k)) / (1.0 - std::pow(b1, (double)k));
        };
        o->h_lrv.resize(4 * ((size_t)upto + 1));
        std::vector<double> Lw((size_t)ORX_ADAM_CF_TERMS + 1), pw((size_t)ORX_ADAM_CF_TERMS + 1);
        for (int j = 1; j <= ORX_ADAM_CF_TERMS; ++j) pw[(size_t)j] = std::pow(b1, (double)j);
        const int64_t k0 = o->lrv_done;
        // (a sliding window over lr: entry k needs lr_{k+1 .. k+J})
        std::vector<double> lr_all((size_t)(upto - k0 + 1 + ORX_ADAM_CF_TERMS) + 1);
        for (size_t i = 0; i < lr_all.size(); ++i) lr_all[i] = L(k0 + (int64_t)i);
        for (int64_t k = k0; k <= upto; ++k) {
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            const double* lp = lr_all.data() + (k - k0);
            for (int j = ORX_ADAM_CF_TERMS; j >= 1; --j) { const double t = pw[(size_t)j] * lp[j]; v0 += t; v1 += t * j; v2 += t * (double)j * j; }
            float* e = o->h_lrv.data() + 4 * (size_t)k;
            e[0] = (float)v0; e[1] = (float)v1; e[2] = (float)v2; e[3] = 0.f;
        }
        ORX_HIP(hipSetDevice(o->ctx->device));
        if (o->lrv_cap < (size_t)upto + 1) {
            const size_t cap = std::max<size_t>(4096, 2 * ((size_t)upto + 1));
            float* p = nullptr;
            ORX_HIP(hipMalloc((void**)&p, cap * 4 * sizeof(float)));
            ORX_HIP(hipStreamSynchronize(o->ctx->stream));
            if (o->d_lrv) { ORX_HIP(hipMemcpy(p, o->d_lrv, (size_t)k0 * 4 * sizeof(float), hipMemcpyDeviceToDevice)); ORX_HIP(hipFree(o->d_lrv)); }
            o->d_lrv = p; o->lrv_cap = cap;
        }
        ORX_HIP(hipMemcpyAsync(o->d_lrv + 4 * (size_t)k0, o->h_lrv.data() + 4 * (size_t)k0, (size_t)(upto + 1 - k0) * 4 * sizeof(float), hipMemcpyHostToDevice, o->ctx->stream));
        // entries that looked beyond `upto` used predicted rates: final only while the learning rate stays (set_lr moves lrv_done back)
        o->lrv_done = upto + 1;
    }
    if (o->lrt_uploaded >= upto + 1) return ORX_OK;
    ORX_HIP(hipSetDevice(o->ctx->device));
    if (o->lrt_cap < (size_t)upto + 1) {
        const size_t cap = std::max<size_t>(4096, 2 * ((size_t)upto + 1));
        float* p = nullptr;
        ORX_HIP(hipMalloc((void**)&p, cap * sizeof(float)));
        ORX_HIP(hipStreamSynchronize(o->ctx->stream));
        if (o->d_lrt) { ORX_HIP(hipMemcpy(p, o->d_lrt, (size_t)o->lrt_uploaded * sizeof(float), hipMemcpyDeviceToDevice)); ORX_HIP(hipFree(o->d_lrt)); }
        o->d_lrt = p; o->lrt_cap = cap;
    }
    ORX_HIP(hipMemcpyAsync(o->d_lrt + o->lrt_uploaded, o->h_lrt.data() + o->lrt_uploaded, (size_t)(upto + 1 - o->lrt_uploaded) * sizeof(float),
                           hipMemcpyHostToDevice, o->ctx->stream));
    o->lrt_uploaded = upto + 1;
    return ORX_OK;
}

// bring every row of `t` up to its lazy optimizer's current step
int orx_table_sync(orx_table* t) {
    if (t == nullptr || t->lazy == nullptr) return ORX_OK;
    orx_opt* o = t->lazy;
    t->lazy = nullptr;
    auto it = o->slots.find(t);
    if (it == o->slots.end() || it->second.last == nullptr) return ORX_OK;
    ORX_HIP(hipSetDevice(o->ctx->device));
    CHECK(orx_adam_lrt(o, o->t));
    return orx_launch_adam_flush(o->ctx, t->w, it->second.s0, it->second.s1, it->second.last, t->rows, t->dim, (int)o->t, o->d_lrt,
                                 o->p0, o->p1, o->p2, orx_adam_cf_params(o));
}

// per-row step stamps of the lazy Adam (all rows current at the optimizer's present step)
int orx_opt_last(orx_opt* o, orx_table* t, bool restamp, int** out, int64_t stamp) {
    OptSlots s;
    CHECK(orx_opt_slots(o, t, &s));
    auto& ref = o->slots[t];
    if (ref.last == nullptr) {
        ORX_HIP(hipMalloc((void**)&ref.last, (size_t)t->rows * sizeof(int)));
        restamp = true;
    }
    if (restamp) CHECK(orx_launch_fill_int(o->ctx, ref.last, t->rows, (int)(stamp >= 0 ? stamp : o->t)));
    *out = ref.last;
    return ORX_OK;
}

static int slot_ptr(orx_opt* o, orx_table* t, int slot, float** p) {
    ORX_ARG(o && t, "optimizer slot: NULL argument");
    OptSlots s;
    CHECK(orx_opt_slots(o, t, &s));
    *p = slot == 0 ? s.s0 : (slot == 1 ? s.s1 : nullptr);
    ORX_ARG(*p, "optimizer slot %d does not exist for optimizer kind %d", slot, o->kind);
    return ORX_OK;
}

extern "C" int orx_opt_slot_read(orx_opt* o, orx_table* t, int slot, int64_t row0, int64_t nrows, float* host_dst) {
    if (t) CHECK(orx_table_sync(t));
    float* p = nullptr;
    CHECK(slot_ptr(o, t, slot, &p));
    return rows_copy(t->ctx, p, t->rows, t->dim, row0, nrows, host_dst, true);
}

extern "C" int orx_opt_slot_write(orx_opt* o, orx_table* t, int slot, int64_t row0, int64_t nrows, const float* host_src) {
    if (t) CHECK(orx_table_sync(t));
    float* p = nullptr;
    CHECK(slot_ptr(o, t, slot, &p));
    return rows_copy(t->ctx, p, t->rows, t->dim, row0, nrows, (float*)host_src, false);
}

// ------------------------------------------------------------ the hot path ---
static int check_pair_tables(orx_table* U, orx_table* V, orx_table* b) {
    ORX_ARG(U && V && b, "pairwise: NULL table");
    ORX_ARG(U->ctx == V->ctx && V->ctx == b->ctx, "pairwise: tables belong to different contexts");
    ORX_ARG(U->dim == V->dim, "pairwise: user dim %d != item dim %d (every model multiplies them element-wise)", U->dim, V->dim);
    ORX_ARG(b->dim == 1 && b->rows == V->rows, "pairwise: item_bias must be [%lld, 1]", (long long)V->rows);
    return ORX_OK;
}

// stage ids of K steps; returns device pointers + stride
static int stage_triplets(orx_ctx* c, const int32_t* uid, const int32_t* pid, const int32_t* nid,
                          int64_t K, int64_t B, int64_t id_stride, int flags,
                          const int32_t** du, const int32_t** dp, const int32_t** dn, int64_t* dstride) {
    if (flags & ORX_IDS_DEVICE) { *du = uid; *dp = pid; *dn = nid; *dstride = id_stride; return ORX_OK; }
    const int64_t n = K * B;
    ENSURE(c->d_ids, c->d_ids_cap, (size_t)3 * n * sizeof(int32_t));
    if (id_stride == B || K == 1) {
        CHECK(stage_ids(c, uid, n, 0));
        CHECK(stage_ids(c, pid, n, n));
        CHECK(stage_ids(c, nid, n, 2 * n));
    } else {
        for (int64_t s = 0; s < K; ++s) {
            CHECK(stage_ids(c, uid + s * id_stride, B, s * B));
            CHECK(stage_ids(c, pid + s * id_stride, B, n + s * B));
            CHECK(stage_ids(c, nid + s * id_stride, B, 2 * n + s * B));
        }
    }
    *du = c->d_ids; *dp = c->d_ids + n; *dn = c->d_ids + 2 * n; *dstride = B;
    return ORX_OK;
}

int fetch_losses(orx_ctx* c, int64_t K, float* loss_out, float* l2_out) {
    if (!loss_out && !l2_out) return ORX_OK;
    std::vector<double> h((size_t)2 * K);
    ORX_HIP(hipMemcpyAsync(h.data(), c->d_loss, sizeof(double) * 2 * K, hipMemcpyDeviceToHost, c->stream));
    ORX_HIP(hipStreamSynchronize(c->stream));
    for (int64_t s = 0; s < K; ++s) {
        if (loss_out) loss_out[s] = (float)h[2 * s];
        if (l2_out) l2_out[s] = (float)h[2 * s + 1];
    }
    return ORX_OK;
}


// the bucketed plan (kernels_plan.hip) needs the role bits; ORX_PLAN_V1=1 keeps dedup_kernel + urgent_kernel
bool orx_plan_v2(bool role_bits) {
    static const bool v1 = getenv("ORX_PLAN_V1") != nullptr;
    return role_bits && !v1;
}

// sizes every per-call buffer of the exact pairwise step (grow-only)
int orx_exact_buffers(orx_ctx* c, orx_table* U, orx_table* V, int64_t K, int64_t B, int mode, bool role_bits,
                      bool inline_apply, bool staging, int nb_total, int nw, PairPlan* plan) {
    // steps are processed in chunks so that the per-step scratch stays bounded (~80 B per triplet and
    // step with a staging plan, ~25 B without)
    int64_t chunk = (int64_t)((256ull << 20) / ((size_t)3 * B * sizeof(int32_t)));
    if (staging && chunk > 256) chunk = 256;
    if (chunk < 1) chunk = 1;
    // The scratch is sized for at least 32 steps (6 MB per step at B = 65536) whatever K is: a training loop's calls vary
    // in length (the driver's protocol: 5 warm-up steps, then 20 timed ones; the Python step queue: whatever was queued
    // when something observed the model), and growing a dozen buffers -- hipFree + hipMalloc each -- inside the longer
    // call cost it ~60 us (3 us per step of a K = 20 call; DESIGN.md 4.0 had this as "the first K = 20 call after a shorter
    // one runs longer").
    const int64_t chunk_cap = std::max<int64_t>(std::min<int64_t>(chunk, 32), std::min<int64_t>(chunk, K));
    if (chunk > K) chunk = K;
    const int64_t used_chunk = chunk;
    chunk = chunk_cap;                                   // (everything below sizes buffers)
    ENSURE(c->d_partial, c->d_partial_cap, (size_t)chunk * nw * 2 * sizeof(float));
    ENSURE(c->d_loss, c->d_loss_cap, (size_t)K * 2 * sizeof(double));
    const int64_t list_stride = 2 * B;          // distinct duplicated rows <= B/2 (users) + B (items)
    // the three rewritten id arrays of a step are padded apart: with B a power of two their
    // addresses would otherwise differ by exact multiples of 256 KiB (same cache set / HBM channel)
    const int64_t Bp = ((B + 3) / 4) * 4 + 96;
    const bool v2 = orx_plan_v2(role_bits);
    if (mode == MODE_EXACT) {
        ENSURE(c->d_ids2, c->d_ids2_cap, (size_t)chunk * 3 * Bp * sizeof(int32_t));
        if (v2) {
            // bucketed plan (kernels_plan.hip): bucket counters, lists of up to 3 B references per step, per-range bitmaps
            CHECK(orx_plan_buffers(c, chunk, B, U->rows, V->rows, inline_apply));
            if (c->h_plan_cap < (size_t)chunk * 9 * sizeof(int)) {
                if (c->h_plan) ORX_HIP(hipHostFree(c->h_plan));
                c->h_plan = nullptr; c->h_plan_cap = 0; c->stats_pending = false;      // (counters a call left there are gone)
                ORX_HIP(hipHostMalloc((void**)&c->h_plan, (size_t)chunk * 9 * sizeof(int), hipHostMallocDefault));
                c->h_plan_cap = (size_t)chunk * 9 * sizeof(int);
            }
            if (!c->plan_ev) ORX_HIP(hipEventCreateWithFlags(&c->plan_ev, hipEventDisableTiming));
            if (!c->plan_stream) {                          // the plan pipeline's stream and events (orx_pairwise_step)
                int lo_p = 0, hi_p = 0;
                ORX_HIP(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
                ORX_HIP(hipStreamCreateWithPriority(&c->plan_stream, hipStreamNonBlocking, hi_p));
                for (int k = 0; k < 2; ++k) {
                    ORX_HIP(hipEventCreateWithFlags(&c->pipe_cnt[k], hipEventDisableTiming));
                    ORX_HIP(hipEventCreateWithFlags(&c->pipe_done[k], hipEventDisableTiming));
                }
            }
        } else {
            if (role_bits) ENSURE(c->d_roles, c->d_roles_cap, (size_t)chunk * 3 * Bp);
            if (inline_apply) ENSURE(c->d_dupbits, c->d_dupbits_cap, (size_t)chunk * nb_total * orx_dedup_words() * sizeof(unsigned int));
        }
        ENSURE(c->d_dlist, c->d_dlist_cap, (size_t)chunk * list_stride * sizeof(uint32_t));
        ENSURE(c->d_dcount, c->d_dcount_cap, (size_t)chunk * sizeof(int));
    }
    // reduction tree over segments longer than ORX_SEG_DIRECT: level 1 has < 3B/64 + 3B/17 work items (one per
    // started 64 references of a long row), level 2 < level 1 / 64 + 3B/1024, level 3 the rest
    const int64_t cap1 = 3 * B / 64 + 3 * B / 17 + 64, cap2 = cap1 / 64 + 3 * B / 1024 + 64, cap3 = cap2 / 64 + 64;
    const int64_t item_stride = cap1 + cap2 + cap3;
    if (v2) ENSURE(c->d_alloc, c->d_alloc_cap, (size_t)chunk * 8 * sizeof(int));
    if (staging) {
        ENSURE(c->d_refinfo, c->d_refinfo_cap, (size_t)chunk * 3 * Bp * sizeof(int2));
        ENSURE(c->d_tricnt, c->d_tricnt_cap, (size_t)chunk * B * sizeof(int));      // (v2: only ranges with > 2048 tri rows use it)
        ENSURE(c->d_segstart, c->d_segstart_cap, (size_t)chunk * B * sizeof(int));
        ENSURE(c->d_alloc, c->d_alloc_cap, (size_t)chunk * 8 * sizeof(int));
        ENSURE(c->d_dseg, c->d_dseg_cap, (size_t)chunk * list_stride * sizeof(int));
        ENSURE(c->d_dcnt, c->d_dcnt_cap, (size_t)chunk * list_stride * sizeof(int));
        ENSURE(c->d_chunks, c->d_chunks_cap, (size_t)chunk * item_stride * sizeof(int4));
        ENSURE(c->d_part, c->d_part_cap, (size_t)item_stride * U->dim * sizeof(float));
        ENSURE(c->d_partb, c->d_partb_cap, (size_t)item_stride * sizeof(float));
        ENSURE(c->d_stage, c->d_stage_cap, (size_t)2 * 3 * B * U->dim * sizeof(float));
        ENSURE(c->d_stageb, c->d_stageb_cap, (size_t)2 * 3 * B * sizeof(float));
    }
    plan->item_stride = item_stride; plan->tree_off[0] = 0; plan->tree_off[1] = (int)cap1; plan->tree_off[2] = (int)(cap1 + cap2);

    chunk = used_chunk;
    plan->nw = nw; plan->chunk = chunk; plan->list_stride = list_stride; plan->Bp = Bp; plan->cap = chunk_cap;
    plan->min_late = -1;
    return ORX_OK;
}

// DedupArgs of steps i0 .. of the chunk's plan arrays
static void plan_dedup_args(orx_ctx* c, orx_table* U, orx_table* V, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t ds,
                            int64_t nU, int64_t nP, int64_t nN, int64_t B, bool role_bits, bool inline_apply, bool staging,
                            const PairPlan& plan, int64_t i0, DedupArgs* out) {
    DedupArgs& d = *out;
    memset(&d, 0, sizeof(d));
    d.uid = uid; d.pid = pid; d.nid = nid; d.id_stride = ds;
    d.dflag = nullptr; d.ids_out = c->d_ids2 + (size_t)i0 * 3 * plan.Bp; d.dlist = c->d_dlist + (size_t)i0 * plan.list_stride; d.dcount = c->d_dcount + i0;
    d.roles = role_bits ? c->d_roles : nullptr;
    d.dupbits = inline_apply ? c->d_dupbits : nullptr;
    d.flag_stride = 3 * plan.Bp; d.role_stride = plan.Bp; d.list_stride = plan.list_stride;
    d.nU = nU; d.nP = nP; d.nN = nN; d.NU = U->rows; d.NI = V->rows;
    d.nbu = orx_dedup_buckets(U->rows); d.nbi = orx_dedup_buckets(V->rows);
    d.min_late = plan.min_late;
    d.alloc = c->d_alloc ? c->d_alloc + 8 * i0 : nullptr;
    if (plan.pair_tpw > 1) {        // pairing (kernels_plan.hip): per-step claims, pairing words, accepted pairs
        d.pair_tpw = plan.pair_tpw; d.pair_stride = B; d.pair_gen = c->pair_gen;
        d.partner = c->d_partner + (size_t)i0 * B; d.pslot = c->d_pslot + (size_t)i0 * 2 * plan.list_stride; d.ids4 = c->d_ids4 + (size_t)i0 * B;
        d.label = nN == 0 ? c->plan_label : nullptr;
    }
    if (staging) {
        d.refinfo = c->d_refinfo + (size_t)i0 * 3 * plan.Bp; d.tricnt = c->d_tricnt + (size_t)i0 * B; d.segstart = c->d_segstart + (size_t)i0 * B;
        d.dseg = c->d_dseg + (size_t)i0 * plan.list_stride; d.dcnt = c->d_dcnt + (size_t)i0 * plan.list_stride; d.items = c->d_chunks + (size_t)i0 * plan.item_stride;
        d.tri_stride = B; d.item_stride = plan.item_stride;
        for (int l = 0; l < 3; ++l) d.tree_off[l] = plan.tree_off[l];
    }
}

// host-side decisions from the per-step counters of steps i0 .. i0 + kc - 1: dc = duplicated rows per step, al = allocators [8] per step
static void plan_decide(int64_t kc, int64_t B, bool inline_apply, bool staging, const int* dc, const int* al, ExactChunk* out) {
    if (inline_apply) {
        // The in-launch apply hides the duplicate apply behind the next step while few references have
        // to wait for it (the headline: ~0.16 B duplicated rows per step, 34 vs 39 us).  With many
        // duplicated rows most references of the next step wait and the launch serializes (100k x 100k
        // tables: 148 vs 56 us per step; break-even measured at ~0.19 B, 800k x 800k tables), so from
        // B/5 duplicated rows on the apply is its own launch.
        int max_dup = 0;
        for (int64_t i = 0; i < kc; ++i) max_dup = std::max(max_dup, dc[i]);
        const char* thr = getenv("ORX_INLINE_DUP_DIV");        // debug: threshold = B / value
        out->dense_dups = (int64_t)max_dup * (thr ? atoi(thr) : 5) > B;
    }
    if (staging) {
        // long segments (a row referenced > 16 times in one step) need the hot_reduce_kernel levels between the
        // fused launch and the apply: the per-step allocators decide
        out->hot = false; out->tree_levels = 0;
        int max_staged = 0;
        for (int64_t i = 0; i < kc; ++i) {
            for (int l = 0; l < 3; ++l) if (al[8 * i + 2 + l] > 0) out->tree_levels = std::max(out->tree_levels, l + 1);
            max_staged = std::max(max_staged, al[8 * i + 1]);
        }
        out->hot = out->tree_levels > 0;
        // the plan covers only row ranges where atomics would pile up; without any plan in the chunk the kernels
        // without the segment bookkeeping are launched
        out->use_stage = max_staged > 0;
    }
}

int orx_exact_plan_issue(orx_ctx* c, orx_table* U, orx_table* V, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t ds,
                         int64_t nU, int64_t nP, int64_t nN, int64_t kc, int64_t B, bool inline_apply, bool staging,
                         const PairPlan& plan, int64_t i0, hipEvent_t counters, const std::function<int()>* after_readback) {
    if (plan.pair_tpw > 1) {
        // the pairing records are not initialised per plan: their words carry the plan's generation (orx_internal.h, ORX_PARTNER_*).
        // The buffer starts all-zero (generation 0 is nobody's) and is zeroed again when the generations wrap.
        c->pair_gen = c->pair_gen % 63 + 1;
        if (c->pair_gen == 1 || c->partner_zeroed != c->d_partner || c->partner_zeroed_cap != c->d_partner_cap) {
            ORX_HIP(hipMemsetAsync(c->d_partner, 0, c->d_partner_cap, c->stream));
            c->partner_zeroed = c->d_partner; c->partner_zeroed_cap = c->d_partner_cap;
        }
    }
    DedupArgs d;
    plan_dedup_args(c, U, V, uid, pid, nid, ds, nU, nP, nN, B, true, inline_apply, staging, plan, i0, &d);
    // bucketed plan; ONE read-back of the per-step counters into pinned memory, and the urgent marks are made while
    // the host waits for it (the fused kernel ignores them in a launch without apply blocks)
    d.roles = nullptr; d.dupbits = nullptr;
    CHECK(orx_launch_plan(c, d, kc, inline_apply, i0));
    if (counters != nullptr) {      // (NULL: nobody waits for this plan's counters, see orx_plan_stats_*)
        // counters a no-read-back pairwise call parked in h_plan are overwritten here (pointwise / row plans share the buffer):
        // they must not be attributed to that call later -- the next pairwise call plans with the read-back again
        if (c->stats_pending) { c->stats_pending = false; c->plan_stats.valid = false; }
        ORX_HIP(hipMemcpyAsync(c->h_plan + 8 * i0, c->d_alloc + 8 * i0, (size_t)kc * 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));   // ([5] = duplicated rows)
        ORX_HIP(hipEventRecord(counters, c->stream));
    }
    if (d.pair_tpw > 1) {
        // pairing: the ids of the positions an accepted pair moves change places once every flag sits on them (urgent marks included).
        // (round 4 measured the marks + records of steps 1 .. on a side stream beside step 0: 37.8 against 37.1 us per step --
        // profiles/r4_plan_side_stream.txt; the switch is gone with the packed copy it scheduled)
        if (inline_apply) CHECK(orx_launch_plan_urgent(c, d, kc, i0));
        CHECK(orx_launch_plan_swap(c, d, kc));
        if (after_readback && *after_readback) CHECK((*after_readback)());
        return ORX_OK;
    }
    // step 0 of a chunk carries no apply blocks and needs no urgent marks: it may go out before them
    if (after_readback && *after_readback) CHECK((*after_readback)());
    if (inline_apply) CHECK(orx_launch_plan_urgent(c, d, kc, i0));
    return ORX_OK;
}

// What the host takes from the per-step counters of a plan (hp: [kc][8]) whether it waited for them or finds them later: the
// geometry of the next plan's workgroups, the pairing pause, and whether the plan was QUIET -- no range wanted a staging plan, the
// duplicated rows stay well below the in-launch apply's break-even, no oversized bucket.
static bool plan_counters_seen(orx_ctx* c, int64_t kc, int64_t B, int64_t i0, const int* hp, bool pairing_on, std::vector<int>* dcv) {
    int big = 0, max_dup = 0;
    bool quiet = true;
    if (dcv) dcv->resize((size_t)kc);
    for (int64_t i = 0; i < kc; ++i) {
        const int dc = hp[8 * i + 5] - hp[8 * i + 7];       // ([5] list entries, [7] of them paired after all)
        if (dcv) (*dcv)[i] = dc;
        max_dup = std::max(max_dup, dc); big = std::max(big, hp[8 * i + 6]);
        if (hp[8 * i + 1] > 0 || hp[8 * i + 2] > 0 || hp[8 * i + 3] > 0 || hp[8 * i + 4] > 0) quiet = false;      // staged references / tree items
    }
    c->plan_big = big > 16384;          // ([6] = the step's largest bucket, if above 8 k references)
    c->stat_max_dup = max_dup; c->stat_pairs = 0;
    for (int64_t i = 0; i < kc; ++i) c->stat_pairs += hp[8 * i + 7];
    if ((int64_t)max_dup * 6 > B || c->plan_big) quiet = false;      // (the in-launch apply stops paying at B / 5 duplicated rows: plan_decide)
    if (pairing_on && kc > 0 && getenv("ORX_PAIR_ALWAYS") == nullptr) {
        int64_t pairs = 0;
        for (int64_t i = 0; i < kc; ++i) pairs += hp[8 * i + 7];
        if (pairs * 16 < kc * B) c->pair_pause = 32;        // fewer than B / 16 accepted pairs per step: not worth its plan
    }
    if (getenv("ORX_PLAN_DEBUG") != nullptr && kc > 0)
        fprintf(stderr, "[orx plan] steps %lld..%lld: step %lld has %d duplicated rows left for the apply, %d accepted pairs, %d staged references; %s\n",
                (long long)i0, (long long)(i0 + kc - 1), (long long)i0, hp[5] - hp[7], hp[7], hp[1], quiet ? "quiet" : "not quiet");
    return quiet;
}

int orx_exact_plan_finish(orx_ctx* c, int64_t kc, int64_t B, bool inline_apply, bool staging, int64_t i0, hipEvent_t counters, ExactChunk* out, bool pairing_on) {
    *out = ExactChunk();
    ORX_HIP(hipEventSynchronize(counters));
    std::vector<int> dcv;
    const int* hp = c->h_plan + 8 * i0;
    const bool quiet = plan_counters_seen(c, kc, B, i0, hp, pairing_on, &dcv);
    plan_decide(kc, B, inline_apply, staging, dcv.data(), hp, out);
    out->quiet = quiet;
    return ORX_OK;
}

// ---- no read-back (orx_pairwise_step).  A K-step call used to stop in its middle: the plan's counters were copied to the host, which
// waited for them before it enqueued steps 1 .. K-1 (is the in-launch apply worth it, did any range stage, are there tree levels).
// Those answers rarely change from one call of a training loop to the next, and none of them is needed for CORRECTNESS once the
// plan is made with staging off (rows referenced >= 3 times then use fp32 atomics; the in-launch apply is exact at any density):
// a call whose predecessor's counters were quiet takes that form, enqueues everything without a host wait, and leaves the copy of its
// own counters behind its last launch for the next call to look at (not quiet any more: that call plans with the read-back again).
static int plan_stats_poll(orx_ctx* c) {
    if (!c->stats_pending) return ORX_OK;
    hipError_t e = hipEventQuery(c->stats_ev);
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        if (++c->stats_age < 8) return ORX_OK;              // (a host far ahead of the device: keep the old answer a little longer)
        e = hipEventSynchronize(c->stats_ev);
    }
    c->stats_pending = false; c->stats_age = 0;
    if (e != hipSuccess) c->plan_stats.valid = false;     // (the counters never arrived: later calls read back again instead of failing here)
    ORX_HIP(e);
    const bool quiet = plan_counters_seen(c, c->stats_kc, c->stats_B, 0, c->h_plan, c->stats_pairing, nullptr);
    c->plan_stats.valid = true; c->plan_stats.quiet = quiet;
    for (int k = 0; k < 5; ++k) c->plan_stats.key[k] = c->stats_key[k];
    return ORX_OK;
}
extern "C" int orx_ctx_stat(orx_ctx* c, int what, int64_t* out) {
    ORX_ARG(c && out && what >= 0 && what <= 3, "orx_ctx_stat: bad argument");
    ORX_HIP(hipSetDevice(c->device));
    if (c->stats_pending) { c->stats_age = 1 << 20; CHECK(plan_stats_poll(c)); }      // (wait for the counters the last call left behind)
    *out = what == 0 ? c->stat_pairs : what == 1 ? c->stat_max_dup : what == 2 ? c->stat_nowait_calls : (int64_t)(c->plan_stats.valid && c->plan_stats.quiet);
    return ORX_OK;
}
static int plan_stats_leave(orx_ctx* c, int64_t kc, int64_t B, bool pairing_on, const int64_t key[5]) {
    c->stat_nowait_calls += 1;
    if (!c->stats_ev) ORX_HIP(hipEventCreateWithFlags(&c->stats_ev, hipEventDisableTiming));
    ORX_HIP(hipMemcpyAsync(c->h_plan, c->d_alloc, (size_t)kc * 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    ORX_HIP(hipEventRecord(c->stats_ev, c->stream));
    c->stats_pending = true; c->stats_kc = kc; c->stats_B = B; c->stats_pairing = pairing_on; c->stats_age = 0;
    for (int k = 0; k < 5; ++k) c->stats_key[k] = key[k];
    return ORX_OK;
}

// The plan of a chunk of kc steps of an exact step (pairwise: three id lists per step; pointwise: two, nN = 0):
// duplicate detection, roles, staging plan, and the host-side decisions read back from it.
int orx_exact_plan_chunk(orx_ctx* c, orx_table* U, orx_table* V, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t ds,
                 int64_t nU, int64_t nP, int64_t nN, int64_t kc, int64_t B, bool role_bits, bool inline_apply, bool staging,
                 const PairPlan& plan, ExactChunk* out, const std::function<int()>* while_waiting) {
    *out = ExactChunk();
    if (orx_plan_v2(role_bits)) {
        // work that does not depend on the counters goes to the device before the host blocks on them (the first fused launch
        // of the chunk: the device would otherwise idle through the host's wake-up and the first launch's latency)
        CHECK(orx_exact_plan_issue(c, U, V, uid, pid, nid, ds, nU, nP, nN, kc, B, inline_apply, staging, plan, 0, c->plan_ev, while_waiting));
        ht_mark("plan_issued");
        const int rc_fin = orx_exact_plan_finish(c, kc, B, inline_apply, staging, 0, c->plan_ev, out, plan.pair_tpw > 1);
        ht_mark("counters");
        return rc_fin;
    }
    // duplicate detection for every step of the chunk, on the id arrays alone
    DedupArgs d;
    plan_dedup_args(c, U, V, uid, pid, nid, ds, nU, nP, nN, B, role_bits, inline_apply, staging, plan, 0, &d);
    ORX_HIP(hipMemsetAsync(c->d_dcount, 0, (size_t)kc * sizeof(int), c->stream));     // (the bucketed plan zeroes its counters itself)
    if (staging) {
        ORX_HIP(hipMemsetAsync(c->d_tricnt, 0, (size_t)kc * B * sizeof(int), c->stream));
        ORX_HIP(hipMemsetAsync(c->d_alloc, 0, (size_t)kc * 8 * sizeof(int), c->stream));
    }
    std::vector<int> dc_v1, al_v1;
    CHECK(orx_launch_dedup(c, d, kc));
    if (inline_apply) {
        dc_v1.resize((size_t)kc);
        ORX_HIP(hipMemcpyAsync(dc_v1.data(), c->d_dcount, dc_v1.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        ORX_HIP(hipStreamSynchronize(c->stream));
    }
    if (staging) {
        al_v1.resize((size_t)kc * 8);
        ORX_HIP(hipMemcpyAsync(al_v1.data(), c->d_alloc, al_v1.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        ORX_HIP(hipStreamSynchronize(c->stream));
    }
    plan_decide(kc, B, inline_apply, staging, dc_v1.data(), al_v1.data(), out);
    if (inline_apply && !out->hot && !out->dense_dups) CHECK(orx_launch_urgent(c, d, kc));
    return ORX_OK;
}

// per-step views into the plan arrays of the current chunk (step i of the chunk)
void orx_exact_step_views(orx_ctx* c, const PairPlan& plan, int64_t i, int64_t B, int D, bool use_stage, PairArgs* a) {
    const int64_t Bp = plan.Bp, list_stride = plan.list_stride;
    a->dlist = c->d_dlist + (size_t)i * list_stride; a->dcount = c->d_dcount + i;
    if (use_stage) {
        const size_t par = (size_t)(i & 1), ppar = par ^ 1;
        a->refinfo = c->d_refinfo + (size_t)i * 3 * Bp; a->segstart = c->d_segstart + (size_t)i * B;
        a->stage = c->d_stage + par * 3 * B * D; a->stageb = c->d_stageb + par * 3 * B;
        a->dseg = c->d_dseg + (size_t)i * list_stride; a->dcnt = c->d_dcnt + (size_t)i * list_stride;
        a->items = c->d_chunks + (size_t)i * plan.item_stride; a->nitems = c->d_alloc + 8 * i + 2;
        a->part = c->d_part; a->partb = c->d_partb;
        for (int l = 0; l < 3; ++l) a->tree_off[l] = plan.tree_off[l];
        a->prev_stage = c->d_stage + ppar * 3 * B * D; a->prev_stageb = c->d_stageb + ppar * 3 * B;
        a->prev_dseg = i > 0 ? c->d_dseg + (size_t)(i - 1) * list_stride : nullptr;
        a->prev_dcnt = i > 0 ? c->d_dcnt + (size_t)(i - 1) * list_stride : nullptr;
    } else {
        a->stage = nullptr; a->stageb = nullptr; a->dcnt = nullptr; a->dseg = nullptr; a->prev_dcnt = nullptr; a->prev_dseg = nullptr;
    }
}

// pairing (kernels_plan.hip): SGD on the float4 dims with >= 2 triplets per wavefront, bucketed plan
// (ORX_FORCE_FALLBACK bit 4 / ORX_NO_PAIR=1: off)
bool orx_pairing_wanted(int mode, bool role_bits, int optkind, int dim, int64_t B, int fb);
int orx_pairing_buffers(orx_ctx* c, int64_t B, int dim, PairPlan* plan);
static bool pairing_wanted(int mode, bool role_bits, int optkind, int dim, int64_t B, int fb) {
    return mode == MODE_EXACT && orx_plan_v2(role_bits) && optkind == ORX_SGD && orx_fused_tpw(dim) > 1 && B >= 2 && B <= (1 << 22) && !(fb & 16) &&
           getenv("ORX_NO_PAIR") == nullptr;
}
static int pairing_buffers(orx_ctx* c, int64_t B, int dim, PairPlan* plan) {
    ENSURE(c->d_partner, c->d_partner_cap, (size_t)plan->cap * B * sizeof(int4));
    ENSURE(c->d_pslot, c->d_pslot_cap, (size_t)plan->cap * 2 * plan->list_stride * sizeof(int));
    ENSURE(c->d_ids4, c->d_ids4_cap, (size_t)plan->cap * B * sizeof(int4));
    plan->pair_tpw = orx_fused_tpw(dim);
    return ORX_OK;
}

bool orx_pairing_wanted(int mode, bool role_bits, int optkind, int dim, int64_t B, int fb) { return pairing_wanted(mode, role_bits, optkind, dim, B, fb); }
int orx_pairing_buffers(orx_ctx* c, int64_t B, int dim, PairPlan* plan) { return pairing_buffers(c, B, dim, plan); }

// TF-2.0 Adam applied lazily (see orx_pairwise_step): float4 dims, role bits available, not hogwild
static bool lazy_adam_ok(const orx_opt* opt, const orx_table* U, const orx_table* V, int flags) {
    return opt->kind == ORX_ADAM && !(flags & ORX_HOGWILD) && orx_fused_can_inline_apply(U->dim) &&
           U->rows < (1LL << 28) && V->rows < (1LL << 28) && U->owned && V->owned &&      // (wrapped memory is read behind our back)
           getenv("ORX_ADAM_DENSE") == nullptr;
}

extern "C" int orx_pairwise_step(orx_ctx* c, int model, orx_opt* opt,
                                 orx_table* U, orx_table* V, orx_table* b,
                                 const int32_t* uid, const int32_t* pid, const int32_t* nid,
                                 int64_t K, int64_t B, int64_t id_stride, float margin, int flags,
                                 float* loss_out, float* l2_out) {
    ORX_ARG(c && opt, "orx_pairwise_step: NULL context/optimizer");
    ORX_ARG(model == ORX_BPR || model == ORX_UCML, "orx_pairwise_step: unknown model %d", model);
    CHECK(check_pair_tables(U, V, b));
    ORX_ARG(U->ctx == c && opt->ctx == c, "orx_pairwise_step: objects belong to a different context");
    ORX_ARG(K >= 0 && B >= 0, "orx_pairwise_step: negative K or B");
    ORX_ARG(K == 0 || B == 0 || (uid && pid && nid), "orx_pairwise_step: NULL id pointer");
    if (K == 0) return ORX_OK;
    g_htc = 0; ht_mark("enter");
    ORX_HIP(hipSetDevice(c->device));
    if (B == 0) {           // empty batch: reduce_mean of nothing is NaN in TF; tables untouched
        for (int64_t s = 0; s < K; ++s) { if (loss_out) loss_out[s] = model == ORX_BPR ? NAN : 0.f; if (l2_out) l2_out[s] = 0.f; }
        return ORX_OK;
    }
    const int32_t *du, *dp, *dn; int64_t ds;
    CHECK(stage_triplets(c, uid, pid, nid, K, B, id_stride, flags, &du, &dp, &dn, &ds));

    const bool hogwild = (flags & ORX_HOGWILD) != 0;
    // TF-2.0 Adam decays m, v and moves var on EVERY row every step.  On the float4 dims that is applied lazily and
    // exactly: a row's gradient-free steps are replayed when the row is next touched (or observed: orx_table_sync);
    // the dense form (every reference accumulates, then three whole-table sweeps per step) remains for the other
    // cases and behind ORX_ADAM_DENSE=1.
    const bool lazy_adam = lazy_adam_ok(opt, U, V, flags) && b->owned;
    const int mode = (opt->kind == ORX_ADAM && !lazy_adam) ? MODE_ACCUM : (hogwild ? MODE_HOGWILD : MODE_EXACT);
    // the three tables are lazy together under one optimizer (the item rows and their biases then share step stamps),
    // or not at all: anything else first brings every row up to date
    const bool lazy_resume = lazy_adam && U->lazy == opt && V->lazy == opt && b->lazy == opt;
    if (!lazy_resume) for (orx_table* t : {U, V, b}) CHECK(orx_table_sync(t));
    { orx_table* mine[3] = {U, V, b}; CHECK(orx_opt_isolate(opt, mine, 3)); }      // (a shared optimizer: see orx_opt_isolate)
    // rows referenced exactly twice get plain stores into two scratch rows; the role of a reference
    // travels in bits 30:29 of its id, which needs tables below 2^29 rows
    // ORX_FORCE_FALLBACK (debug / tests): bit 0 = behave as if the tables had >= 2^28 rows (no role bits: every
    // duplicate uses atomics, separate dup_apply launches), bit 1 = no in-launch apply
    const char* fb_env = getenv("ORX_FORCE_FALLBACK");
    const int fb = fb_env ? atoi(fb_env) : 0;
    const bool role_bits = mode == MODE_EXACT && U->rows < (1LL << 28) && V->rows < (1LL << 28) && !(fb & 1);
    // censor_vec after every step (ucml.py:44-48): fused into the row write-back for the float4 dims in
    // exact mode (fb bit 2 forces the separate passes); separate passes otherwise
    const bool want_censor = (flags & ORX_CENSOR) != 0;
    const bool fused_censor = want_censor && mode == MODE_EXACT && orx_fused_can_inline_apply(U->dim) && !(fb & 4);
    const bool censor = want_censor && !fused_censor;
    // the previous step's duplicated rows are applied by extra blocks of the next step's launch
    // (no dup_apply launch, no kernel boundary) -- not with a separate censor pass between the steps
    const bool inline_apply = role_bits && !censor && K > 1 && orx_fused_can_inline_apply(U->dim) && !(fb & 2);
    // rows referenced >= 3 times in a step: private staging slots instead of atomics (fb bit 3: atomics)
    bool staging = role_bits && orx_fused_can_inline_apply(U->dim) && !(fb & 8);
    // no read-back (see plan_stats_poll): the previous call of this shape was quiet
    const int64_t stats_key[5] = {B, U->rows, V->rows, (int64_t)model * 16 + opt->kind, (int64_t)U->dim * 4 + (want_censor ? 1 : 0) + (inline_apply ? 2 : 0)};
    CHECK(plan_stats_poll(c));
    const bool plan_wait = getenv("ORX_PLAN_WAIT") != nullptr || getenv("ORX_PLAN_PIPE") != nullptr;      // (experiments / tests: always read back)
    bool nowait = mode == MODE_EXACT && orx_plan_v2(role_bits) && staging && !censor && opt->kind != ORX_ADAM && !plan_wait &&
                  c->plan_stats.valid && c->plan_stats.quiet;
    for (int k = 0; k < 5 && nowait; ++k) nowait = c->plan_stats.key[k] == stats_key[k];
    if (nowait) staging = false;
    const int nb_total = orx_dedup_buckets(U->rows) + orx_dedup_buckets(V->rows);
    if (mode != MODE_HOGWILD) {
        CHECK(orx_table_scratch(U, role_bits)); CHECK(orx_table_scratch(V, role_bits)); CHECK(orx_table_scratch(b, role_bits));
    }
    OptSlots sU, sV, sb;
    CHECK(orx_opt_slots(opt, U, &sU)); CHECK(orx_opt_slots(opt, V, &sV)); CHECK(orx_opt_slots(opt, b, &sb));

    PairPlan plan;
    CHECK(orx_exact_buffers(c, U, V, K, B, mode, role_bits, inline_apply, staging, nb_total, orx_fused_nwaves(U->dim, B), &plan));
    const int nw = plan.nw;
    const int64_t chunk = plan.chunk, list_stride = plan.list_stride, Bp = plan.Bp;
    // Adam's normalised update amplifies summation-order noise where an element's summed gradient nearly cancels: its
    // rows referenced >= 3 times always take staging slots (fixed summation order), never fp32 atomics
    if (opt->kind == ORX_ADAM) plan.min_late = 1;
    // pairing (kernels_plan.hip): the two triplets of a row referenced exactly twice share a wavefront and exchange gradients there
    // (SGD only -- Adagrad with pairing measured slower, profiles/r5_adagrad_pairing_ab.txt -- on the float4 dims with >= 2 triplets per wavefront; fb bit 4 / ORX_NO_PAIR=1: off)
    // The pairing plan costs ~1.5 us per step (records, decisions, the swaps); it pays where a good share of the batch pairs
    // (uniform ids over tables ~ 10 x the batch: 11 %).  Tables so small that most duplicated rows have three or more references, or
    // ids so skewed that the hot rows take them, pair little: the plan of a call tells (accepted pairs per step), and pairing then
    // pauses for 32 calls before it is tried again.  ORX_PAIR_ALWAYS=1: no pause.
    const bool pair_ok = pairing_wanted(mode, role_bits, opt->kind, U->dim, B, fb);
    if (pair_ok && c->pair_pause > 0) c->pair_pause -= 1;
    else if (pair_ok) CHECK(pairing_buffers(c, B, U->dim, &plan));

    PairArgs a;
    memset(&a, 0, sizeof(a));
    a.U = U->w; a.V = V->w; a.b = b->w;
    a.gU = U->gsum; a.gV = V->gsum; a.gb = b->gsum;
    if (role_bits) { a.gU2 = U->gsum2; a.gV2 = V->gsum2; a.gb2 = b->gsum2; a.role_bits = 1; a.readyU = U->ready; a.readyV = V->ready; }
    a.aU = sU.s0; a.aV = sV.s0; a.ab = sb.s0;
    a.B = B; a.NU = U->rows; a.NI = V->rows; a.D = U->dim;
    a.lr = opt->lr;
    a.eps = opt->kind == ORX_ADAGRAD ? opt->p1 : 0.f;
    a.margin = margin;
    a.invB = 1.0f / (float)B;
    a.l2w = (flags & ORX_NO_L2) ? 0.f : 1.f;
    a.err = c->d_err;
    if (lazy_adam) {
        a.a2U = sU.s1; a.a2V = sV.s1; a.a2b = sb.s1;
        CHECK(orx_opt_last(opt, U, !lazy_resume, &a.lastU)); CHECK(orx_opt_last(opt, V, !lazy_resume, &a.lastV));
        CHECK(orx_opt_last(opt, b, !lazy_resume, &a.lastb));
        CHECK(orx_adam_lrt(opt, opt->t + K));
        a.lrt = opt->d_lrt; a.b1 = opt->p0; a.b2 = opt->p1; a.eps = opt->p2;
        if (orx_adam_cf_ok(opt) && opt->d_lrv != nullptr) {      // closed-form replay (fused_kernel LONGGAP = 2)
            a.lrv = reinterpret_cast<const float4*>(opt->d_lrv);
            a.cf_delta = (float)(-0.5 * std::log((double)opt->p1)); a.cf_lb1 = (float)std::log2((double)opt->p0); a.cf_lb2 = (float)std::log2((double)opt->p1);
        }
        a.newton = (1.0f - sqrtf(opt->p1)) <= 1e-3f && getenv("ORX_ADAM_NO_NEWTON") == nullptr;
        // expected steps between two references of a row = rows / references per step
        a.long_gap = (U->rows / B > 64 || V->rows / (2 * B) > 64) && getenv("ORX_ADAM_NO_LONGGAP") == nullptr;
        U->lazy = opt; V->lazy = opt; b->lazy = opt;
    }
    // epochs are consumed one per step; on wrap-around every table clears its epoch-tagged arrays
    if ((int64_t)c->epoch + K + 16 > 0x7fffffff) { c->epoch = 0; c->epoch_gen += 1; }
    for (orx_table* t : {U, V}) {
        if (t->tag_gen != c->epoch_gen) {
            if (t->ready) ORX_HIP(hipMemsetAsync(t->ready, 0, (size_t)t->rows * sizeof(int), c->stream));
            if (t->side) ORX_HIP(hipMemsetAsync(t->side, 0, (size_t)t->rows * 2 * sizeof(int), c->stream));
            t->tag_gen = c->epoch_gen;
        }
    }
    if (fused_censor) { CHECK(orx_table_side(V)); a.censor = 1; a.min_norm = 0.1f; a.sideV = V->side; }

    ht_mark("setup");
    for (int64_t s0 = 0; s0 < K; s0 += chunk) {
        const int64_t kc = (K - s0 < chunk) ? (K - s0) : chunk;
        // arguments of step i of the chunk, and its fused launch
        auto launch_step = [&](int64_t i, bool stage_views, bool with_apply) -> int {
            const int64_t s = s0 + i;
            if (mode == MODE_EXACT) {       // ids rewritten by the plan (duplicate flag in bit 31)
                a.uid = c->d_ids2 + (size_t)i * 3 * Bp; a.pid = a.uid + Bp; a.nid = a.uid + 2 * Bp;
            } else {
                a.uid = du + s * ds; a.pid = dp + s * ds; a.nid = dn + s * ds;
            }
            orx_exact_step_views(c, plan, i, B, U->dim, stage_views, &a);
            a.ids4 = plan.pair_tpw > 1 ? c->d_ids4 + (size_t)i * B : nullptr;
            a.partial = c->d_partial + (size_t)i * nw * 2;
            a.epoch = ++c->epoch;           // one epoch per step: ready flags and censor side marks are tagged with it
            if (lazy_adam) { opt->t += 1; a.step_t = (int)opt->t; }
            if (with_apply && i > 0) {    // this launch also applies the duplicated rows of step i-1
                // one lane group per duplicated row in a single pass for the usual ~0.15*B duplicated rows
                // (the count lives in device memory; surplus blocks exit, a larger count grid-strides)
                a.n_apply_blocks = (int)std::min<int64_t>(2048, std::max<int64_t>(16, (B / 4) / (1024 / U->dim) + 1));
                a.prev_dlist = c->d_dlist + (size_t)(i - 1) * list_stride; a.prev_dcount = c->d_dcount + (i - 1);
            } else {
                a.n_apply_blocks = 0; a.prev_dlist = nullptr; a.prev_dcount = nullptr;
            }
            return orx_launch_fused(c, model, opt->kind, mode, a);
        };
        bool first_launched = false;
        bool tail_done = false;             // the chunk's loss sums left with the last step's duplicate apply
        // ---- the plan of the chunk, optionally in PIECES (bucketed plan, in-launch apply): piece 0 (6 steps) is planned on the step
        // stream, every later piece (4x the previous one) on a second stream while the steps of the piece before it run -- the fused
        // launches of a K-step call then start ~45 us after the call instead of after the plan of all K steps (110 us at K = 20,
        // 620 us at K = 200).  The pieces share the chunk's plan arrays, so the in-launch apply runs across their boundaries.
        std::vector<int64_t> pc_lo, pc_hi;                   // piece j = steps [pc_lo[j], pc_hi[j]) of the chunk
        std::vector<ExactChunk> pck;
        const bool v2 = mode == MODE_EXACT && orx_plan_v2(role_bits);
        // MEASURED AND LEFT OFF (profiles/r3_plan_pipeline.txt; ORX_PLAN_PIPE=1 turns it on): what runs beside the fused kernels takes
        // from them what it gets.  K = 20: 41.5 / 40.5 us per step with the pipeline against 39.2 without (fused launch 32.5 against
        // 30.8 us); K = 40: 37.9 / 36.9; K = 64: 37.2 / 36.3; K = 200: 34.3 / 33.7 -- the same verdict as round 2's chunk-level overlap.
        const bool pipe = v2 && inline_apply && !censor && kc > 8 && getenv("ORX_PLAN_PIPE") != nullptr;
        if (mode == MODE_EXACT) {
            if (pipe) {
                for (int64_t lo = 0, sz = 6; lo < kc; lo += sz, sz *= 4) {
                    int64_t hi = std::min<int64_t>(kc, lo + sz);
                    if (kc - hi < 3) hi = kc;                // (no 1-2 step tail piece)
                    pc_lo.push_back(lo); pc_hi.push_back(hi);
                    if (hi == kc) break;
                }
            } else {
                pc_lo.push_back(0); pc_hi.push_back(kc);
            }
            pck.resize(pc_lo.size());
            // With the bucketed plan the chunk's FIRST fused launch goes out before the host waits for the plan's counters:
            // step 0 never carries apply blocks, and the kernel built with the staging bookkeeping is right whether or not
            // any range made a staging plan (references without one carry (-1, 0) and use atomics).
            const std::function<int()> early = [&]() -> int { first_launched = true; return launch_step(0, staging, false); };
            const bool can_early = v2 && !censor && getenv("ORX_PLAN_NO_EARLY") == nullptr;
            if (nowait) {
                CHECK(orx_exact_plan_issue(c, U, V, du + s0 * ds, dp + s0 * ds, dn + s0 * ds, ds, B, B, B, kc, B, inline_apply, staging, plan, 0, nullptr, nullptr));
                pck[0] = ExactChunk();                       // (in-launch apply where the call allows it, no staging, no tree)
                ht_mark("plan_issued");
            } else if (pipe) {
                CHECK(orx_exact_plan_issue(c, U, V, du + s0 * ds, dp + s0 * ds, dn + s0 * ds, ds, B, B, B, pc_hi[0], B, inline_apply, staging, plan, 0,
                                           c->plan_ev, can_early ? &early : nullptr));
                CHECK(orx_exact_plan_finish(c, pc_hi[0], B, inline_apply, staging, 0, c->plan_ev, &pck[0], plan.pair_tpw > 1));
            } else {
                CHECK(orx_exact_plan_chunk(c, U, V, du + s0 * ds, dp + s0 * ds, dn + s0 * ds, ds, B, B, B, kc, B, role_bits, inline_apply, staging,
                                           plan, &pck[0], can_early ? &early : nullptr));
            }
        }
        // enqueue the plan of piece j (>= 1) on the plan stream; it starts once the device has passed `after`
        auto issue_piece = [&](size_t j, hipEvent_t after) -> int {
            hipStream_t main_stream = c->stream;
            ORX_HIP(hipStreamWaitEvent(c->plan_stream, after, 0));
            c->stream = c->plan_stream;
            const int64_t lo = pc_lo[j], n = pc_hi[j] - lo;
            const int rc = orx_exact_plan_issue(c, U, V, du + (s0 + lo) * ds, dp + (s0 + lo) * ds, dn + (s0 + lo) * ds, ds, B, B, B, n, B,
                                                inline_apply, staging, plan, lo, c->pipe_cnt[j & 1], nullptr);
            if (rc == ORX_OK) { const hipError_t e = hipEventRecord(c->pipe_done[j & 1], c->plan_stream); if (e != hipSuccess) { c->stream = main_stream; ORX_HIP(e); } }
            c->stream = main_stream;
            return rc;
        };
        if (censor) {
            // one elected reference per distinct row of each of the three id lists, for every step of the chunk
            ENSURE(c->d_cflag, c->d_cflag_cap, (size_t)3 * kc * B);
            const int32_t* lists[3] = {du + s0 * ds, dp + s0 * ds, dn + s0 * ds};
            const int64_t rows[3] = {U->rows, V->rows, V->rows};
            for (int l = 0; l < 3; ++l) {
                DedupArgs d;
                memset(&d, 0, sizeof(d));
                d.uid = lists[l]; d.pid = lists[l]; d.nid = lists[l]; d.id_stride = ds;
                d.dflag = c->d_cflag + (size_t)l * kc * B; d.flag_stride = B;
                d.nU = B; d.NU = rows[l]; d.nbu = orx_dedup_buckets(rows[l]); d.first_only = 1;
                CHECK(orx_launch_dedup(c, d, kc));
            }
        }
        // per piece: may the duplicated rows of its steps be applied by the NEXT step's launch?
        auto piece_inl = [&](size_t j) { return inline_apply && !pck[j].hot && !pck[j].dense_dups; };
        const size_t npc = mode == MODE_EXACT ? pc_lo.size() : 1;
        for (size_t j = 0; j < npc; ++j) {
            const int64_t lo = mode == MODE_EXACT ? pc_lo[j] : 0, hi = mode == MODE_EXACT ? pc_hi[j] : kc;
            // (piece 1 may start behind piece 0's counters: plan_ev sits behind its plan kernels, ahead of step 0 and the urgent marks)
            if (j + 1 < npc) CHECK(issue_piece(j + 1, j == 0 ? c->plan_ev : c->pipe_done[j & 1]));
            if (j > 0) ORX_HIP(hipStreamWaitEvent(c->stream, c->pipe_done[j & 1], 0));      // the step stream takes up this piece's steps behind its plan
            const bool inl_j = mode == MODE_EXACT && piece_inl(j);
            const int tree_levels = mode == MODE_EXACT ? pck[j].tree_levels : 0;
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t s = s0 + i;
                bool inl_next = inl_j;                         // is step i + 1 launched with apply blocks for step i's rows?
                if (i == hi - 1 && j + 1 < npc) {
                    // the last step of a piece: the next piece's counters decide (its plan has had the whole piece to finish)
                    CHECK(orx_exact_plan_finish(c, pc_hi[j + 1] - pc_lo[j + 1], B, inline_apply, staging, pc_lo[j + 1], c->pipe_cnt[(j + 1) & 1], &pck[j + 1], plan.pair_tpw > 1));
                    inl_next = inl_j && piece_inl(j + 1);
                }
                const bool defer = mode == MODE_EXACT && inl_next && i < kc - 1;      // step i's duplicated rows wait for launch i + 1
                if (!(i == 0 && first_launched)) {
                    const bool prev_deferred = i > 0 && (i > lo ? inl_j : (j > 0 && piece_inl(j - 1) && inl_j));
                    // (a step reads the staging segments of the step before it: the views cover both pieces' plans)
                    const bool sv = mode == MODE_EXACT && (pck[j].use_stage || (i == lo && j > 0 && pck[j - 1].use_stage));
                    CHECK(launch_step(i, sv, prev_deferred));
                }
                for (int l = 0; l < tree_levels; ++l) CHECK(orx_launch_hot_reduce(c, a, l));
                if (mode == MODE_EXACT && !defer) {
                    // the chunk's last step: its duplicated rows and the chunk's loss sums leave in ONE launch where that is possible
                    if (i == kc - 1 && !censor && !lazy_adam) {
                        ReduceArgs r;
                        r.partial = c->d_partial; r.out = c->d_loss + 2 * s0; r.nwaves = nw;
                        int rc = ORX_OK;
                        if (orx_launch_tail(c, opt->kind, a, r, kc, &rc)) { CHECK(rc); tail_done = true; }
                    }
                    if (!tail_done) CHECK(orx_launch_dup_apply(c, opt->kind, a));
                }
                if (mode == MODE_ACCUM) {   // Adam: dense-decay sweep of TF 2.0 over the whole tables
                    opt->t += 1;
                    const double b1 = opt->p0, b2 = opt->p1;
                    const float lr_t = (float)(opt->lr * std::sqrt(1.0 - std::pow(b2, (double)opt->t)) / (1.0 - std::pow(b1, (double)opt->t)));
                    CHECK(orx_launch_adam_sweep(c, U->w, sU.s0, sU.s1, U->gsum, U->rows * U->dim, lr_t, opt->p0, opt->p1, opt->p2));
                    CHECK(orx_launch_adam_sweep(c, V->w, sV.s0, sV.s1, V->gsum, V->rows * V->dim, lr_t, opt->p0, opt->p1, opt->p2));
                    CHECK(orx_launch_adam_sweep(c, b->w, sb.s0, sb.s1, b->gsum, b->rows, lr_t, opt->p0, opt->p1, opt->p2));
                }
                if (censor) {                   // ucml.py:44-48: users and pos items (two tables), then neg items
                    const unsigned char* fu = c->d_cflag + (size_t)i * B;
                    const unsigned char* fp = c->d_cflag + (size_t)(kc + i) * B;
                    const unsigned char* fn = c->d_cflag + (size_t)(2 * kc + i) * B;
                    CHECK(orx_launch_censor2(c, U->w, fu, U->rows, du + s * ds, B, V->w, fp, V->rows, dp + s * ds, B, U->dim, 0.1f));
                    CHECK(orx_launch_censor2(c, V->w, fn, V->rows, dn + s * ds, B, nullptr, nullptr, 0, nullptr, 0, U->dim, 0.1f));
                }
            }
        }
        if (!tail_done) {
            ReduceArgs r;
            r.partial = c->d_partial; r.out = c->d_loss + 2 * s0; r.nwaves = nw;
            CHECK(orx_launch_loss_reduce(c, r, kc));
        }
        if (mode == MODE_EXACT && orx_plan_v2(role_bits)) {
            // what this chunk's plan counted: left behind the last launch for the next call (no read-back), or seen already
            if (nowait) CHECK(plan_stats_leave(c, kc, B, plan.pair_tpw > 1, stats_key));
            else {
                bool quiet = true;
                for (const ExactChunk& ck : pck) quiet = quiet && ck.quiet;
                c->plan_stats.valid = true; c->plan_stats.quiet = quiet; c->stats_pending = false;
                for (int k = 0; k < 5; ++k) c->plan_stats.key[k] = stats_key[k];
            }
        }
    }
    ht_mark("launched");
    CHECK(fetch_losses(c, K, loss_out, l2_out));
    ht_mark("losses"); ht_dump();
    if (!(flags & ORX_IDS_DEVICE)) return orx_check_index_error(c);
    return ORX_OK;
}

extern "C" int orx_pairwise_reserve(orx_ctx* c, orx_opt* opt, orx_table* U, orx_table* V, orx_table* b, int64_t K, int64_t B) {
    ORX_ARG(c && opt, "orx_pairwise_reserve: NULL context/optimizer");
    CHECK(check_pair_tables(U, V, b));
    ORX_ARG(K > 0 && B > 0, "orx_pairwise_reserve: K and B must be positive");
    ORX_HIP(hipSetDevice(c->device));
    const int mode = (opt->kind == ORX_ADAM && !lazy_adam_ok(opt, U, V, 0)) ? MODE_ACCUM : MODE_EXACT;
    const bool role_bits = mode == MODE_EXACT && U->rows < (1LL << 28) && V->rows < (1LL << 28);
    const bool inline_apply = role_bits && K > 1 && orx_fused_can_inline_apply(U->dim);
    const bool staging = role_bits && orx_fused_can_inline_apply(U->dim);
    const int nb_total = orx_dedup_buckets(U->rows) + orx_dedup_buckets(V->rows);
    CHECK(orx_table_scratch(U, role_bits)); CHECK(orx_table_scratch(V, role_bits)); CHECK(orx_table_scratch(b, role_bits));
    OptSlots s;
    CHECK(orx_opt_slots(opt, U, &s)); CHECK(orx_opt_slots(opt, V, &s)); CHECK(orx_opt_slots(opt, b, &s));
    PairPlan plan;
    CHECK(orx_exact_buffers(c, U, V, K, B, mode, role_bits, inline_apply, staging, nb_total, orx_fused_nwaves(U->dim, B), &plan));
    if (pairing_wanted(mode, role_bits, opt->kind, U->dim, B, 0)) CHECK(pairing_buffers(c, B, U->dim, &plan));
    ORX_HIP(hipStreamSynchronize(c->stream));
    return ORX_OK;
}

extern "C" int orx_pairwise_loss(orx_ctx* c, int model, orx_table* U, orx_table* V, orx_table* b,
                                 const int32_t* uid, const int32_t* pid, const int32_t* nid,
                                 int64_t B, float margin, int flags, float* loss_out, float* l2_out) {
    if (U) CHECK(orx_table_sync(U));
    if (V) CHECK(orx_table_sync(V));
    if (b) CHECK(orx_table_sync(b));
    ORX_ARG(c, "orx_pairwise_loss: NULL context");
    ORX_ARG(model == ORX_BPR || model == ORX_UCML, "orx_pairwise_loss: unknown model %d", model);
    CHECK(check_pair_tables(U, V, b));
    ORX_ARG(B > 0 && uid && pid && nid, "orx_pairwise_loss: empty batch or NULL ids");
    ORX_HIP(hipSetDevice(c->device));
    const int32_t *du, *dp, *dn; int64_t ds;
    CHECK(stage_triplets(c, uid, pid, nid, 1, B, B, flags, &du, &dp, &dn, &ds));
    const int nw = orx_fused_nwaves(U->dim, B);
    ENSURE(c->d_partial, c->d_partial_cap, (size_t)nw * 2 * sizeof(float));
    ENSURE(c->d_loss, c->d_loss_cap, 2 * sizeof(double));
    PairArgs a;
    memset(&a, 0, sizeof(a));
    a.U = U->w; a.V = V->w; a.b = b->w;
    a.B = B; a.NU = U->rows; a.NI = V->rows; a.D = U->dim;
    a.margin = margin; a.invB = 1.0f / (float)B; a.l2w = 1.f;
    a.partial = c->d_partial; a.err = c->d_err;
    a.uid = du; a.pid = dp; a.nid = dn;
    CHECK(orx_launch_fused(c, model, ORX_SGD, MODE_LOSS, a));
    ReduceArgs r;
    r.partial = c->d_partial; r.out = c->d_loss; r.nwaves = nw;
    CHECK(orx_launch_loss_reduce(c, r, 1));
    CHECK(fetch_losses(c, 1, loss_out, l2_out));
    return orx_check_index_error(c);
}

// ---------------------------------------------------------------- profiling ---
extern "C" int orx_prof_enable(orx_ctx* c, int on) {
    ORX_ARG(c, "orx_prof_enable: NULL context");
    c->prof = on != 0;
    return ORX_OK;
}

extern "C" int orx_prof_reset(orx_ctx* c) {
    ORX_ARG(c, "orx_prof_reset: NULL context");
    ORX_HIP(hipStreamSynchronize(c->stream));
    CHECK(prof_collect(c));
    for (int k = 0; k < ORX_K_NUM; ++k) { c->prof_slot[k].total_ms = 0.0; c->prof_slot[k].launches = 0; }
    return ORX_OK;
}

extern "C" int orx_prof_get(orx_ctx* c, int kid, double* total_ms, int64_t* launches) {
    ORX_ARG(c && kid >= 0 && kid < ORX_K_NUM, "orx_prof_get: bad argument");
    ORX_HIP(hipStreamSynchronize(c->stream));
    CHECK(prof_collect(c));
    if (total_ms) *total_ms = c->prof_slot[kid].total_ms;
    if (launches) *launches = c->prof_slot[kid].launches;
    return ORX_OK;
}
