// The row-sharded pairwise step as ONE host call per K steps: exchange plan, gathers, gradients,
// applies and the exchanges themselves (RCCL point-to-point groups on the context's stream) run from here, no interpreter between
// the phases.  The per-phase entry points (orx_shard_*, orx_gather_rows, orx_apply_rows*) stay exported: openrec_amd/sharded.py
// drives the same sequence over torch.distributed for the gloo tests and the in-process virtual clusters.
//
// RCCL is loaded at run time (dlopen of librccl.so in orx_comm_create): the library itself does not link against it, a
// one-rank communicator made without an id never touches it.
#include "orx_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <vector>
#include <utility>

#define CHECK(call) do { const int rc_ = (call); if (rc_ != ORX_OK) return rc_; } while (0)

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl_api() {
    // (a function-local static: initialised once, thread-safely, whichever thread asks first)
    static RcclApi* loaded = []() -> RcclApi* {
        static RcclApi api;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void* h = nullptr;
        for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return nullptr;
#define ORX_SYM(field, name) do { *(void**)(&api.field) = dlsym(h, name); if (!api.field) { dlclose(h); return nullptr; } } while (0)
        ORX_SYM(GetUniqueId, "ncclGetUniqueId"); ORX_SYM(CommInitRank, "ncclCommInitRank"); ORX_SYM(CommDestroy, "ncclCommDestroy");
        ORX_SYM(GroupStart, "ncclGroupStart"); ORX_SYM(GroupEnd, "ncclGroupEnd"); ORX_SYM(Send, "ncclSend"); ORX_SYM(Recv, "ncclRecv");
        ORX_SYM(GetErrorString, "ncclGetErrorString"); ORX_SYM(AllReduce, "ncclAllReduce");
#undef ORX_SYM
        api.handle = h;
        return &api;
    }();
    return loaded;
}

#define ORX_NCCL(api, expr) do { const ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
        orx_set_error("RCCL: %s failed: %s", #expr, (api)->GetErrorString(r_)); return ORX_ERR_HIP; } } while (0)

// [Kc][N][cw] words <-> [N][Kc][cw] words: the K-step plan's buckets regrouped so that every peer's share is one block
__global__ __launch_bounds__(256) void shard_regroup_kernel(const uint32_t* src, uint32_t* dst, int Kc, int N, int64_t cw, int back) {
    const int64_t total = (int64_t)Kc * N * cw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t w = i % cw, blk = i / cw;
        // forward: dst index i = (p, k, w) reads src (k, p, w); back: dst index i = (k, p, w) reads src (p, k, w)
        const int64_t a = blk / (back ? N : Kc), b2 = blk % (back ? N : Kc);
        const int64_t s = back ? ((int64_t)b2 * Kc + a) : ((int64_t)b2 * N + a);
        dst[i] = src[s * cw + w];
    }
}

template <typename W>
__global__ __launch_bounds__(256) void shard_copy_kernel(const W* src, W* dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

struct Buf { void* p = nullptr; size_t cap = 0; };

}  // namespace

// An in-process group of ranks on ONE device (one thread and one context per rank): the engine's whole exchange schedule with
// world > 1 and no second GPU.  Exchanges meet at a host barrier; blocks are copied by kernels between the ranks' buffers.
struct orx_vgroup {
    int world = 1;
    std::mutex mu; std::condition_variable cv; int waiting = 0; long generation = 0; bool broken = false;
    const void* send[64] = {}; const void* send2[64] = {};
    const void* red[64] = {};                              // all-reduce of the virtual group: every rank's vector
    bool wait() {                                           // false: the group was aborted (a rank failed)
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const long gen = generation;
        if (++waiting == world) { waiting = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != gen || broken; });
        return !broken;
    }
};

struct orx_comm {
    orx_ctx* ctx = nullptr;
    orx_vgroup* vg = nullptr;
    ncclComm_t comm = nullptr;                           // NULL: a one-rank communicator without RCCL (every exchange is the identity)
    int rank = 0, world = 1;
    hipStream_t xstream = nullptr;                       // the exchanges of the overlapped path run here, beside the kernels
    hipEvent_t ev[8] = {};
    // orx_comm_stats: counters of the exchanges, (start, stop) event pairs awaiting collection
    bool stats_on = false; double st_exchanges = 0, st_wire = 0, st_self = 0, st_ms = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> st_ev;
    // the engine's exchange buffers (grown on demand, kept between calls)
    Buf dl_send, dl_slot, dl_req, dl_reqloc, dl_rows_out, dl_rows_in, dl_send_g, dl_g_in, dl_idx, dl_ids, dl_flat, dl_sum, dl_cnt, dl_ptrs;     // hybrid-parallel DLRM engine
    Buf send1, mine, tmp, cnt, send2, req, req_loc, slot, u_loc, fu, fv, rows_out, rows_in, gu, u_apply, send_g, g_in, dupref, dsorted, seglist, segcount, gdup, bias_x, partials;
    Buf hot_ids, hot_g, hot_arange, hot_sorted; int64_t hot_arange_n = -1;        // hot-item replication: slot ids [L][2T], the [hot, D + 4] gradient block, 0 .. hot-1
};

static int ensure(orx_comm* c, Buf& b, size_t bytes) {
    if (bytes <= b.cap) return ORX_OK;
    if (b.p) { hipStreamSynchronize(c->ctx->stream); hipFree(b.p); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) { orx_set_error("sharded engine: out of device memory (%zu bytes)", want); return ORX_ERR_OOM; }
    b.cap = want;
    return ORX_OK;
}

static int exchange(orx_comm* c, const void* send, void* recv, size_t bytes, const void** result, hipStream_t stream = nullptr,
                    const void* send2 = nullptr, void* recv2 = nullptr, size_t bytes2 = 0, const void** result2 = nullptr);

extern "C" int orx_comm_stats(orx_comm* c, int start, double* out4) {
    ORX_ARG(c, "orx_comm_stats: NULL communicator");
    ORX_HIP(hipSetDevice(c->ctx->device));
    if (start) {
        for (auto& p : c->st_ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
        c->st_ev.clear();
        c->st_exchanges = c->st_wire = c->st_self = c->st_ms = 0;
        c->stats_on = true;
        return ORX_OK;
    }
    c->stats_on = false;
    ORX_HIP(hipStreamSynchronize(c->ctx->stream));
    if (c->xstream) ORX_HIP(hipStreamSynchronize(c->xstream));
    for (auto& p : c->st_ev) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) c->st_ms += ms;
        hipEventDestroy(p.first); hipEventDestroy(p.second);
    }
    c->st_ev.clear();
    if (out4) { out4[0] = c->st_exchanges; out4[1] = c->st_wire; out4[2] = c->st_ms; out4[3] = c->st_self; }
    return ORX_OK;
}

extern "C" int orx_comm_ping(orx_comm* c, int64_t bytes, int32_t reps, double* out3) {
    ORX_ARG(c && out3 && bytes > 0 && reps > 0, "orx_comm_ping: bad arguments");
    ORX_ARG(!c->vg, "orx_comm_ping: a virtual group has no wire");
    if (!c->comm) {         // a one-rank communicator without RCCL: every exchange is the identity (nothing is enqueued, nothing to time)
        out3[0] = out3[1] = out3[2] = 0.0;
        return ORX_OK;
    }
    ORX_HIP(hipSetDevice(c->ctx->device));
    const int N = c->world;
    void *sbuf = nullptr, *rbuf = nullptr;
    ORX_HIP(hipMalloc(&sbuf, (size_t)bytes * N)); 
    if (hipMalloc(&rbuf, (size_t)bytes * N) != hipSuccess) { hipFree(sbuf); orx_set_error("orx_comm_ping: out of device memory"); return ORX_ERR_OOM; }
    hipMemsetAsync(sbuf, 1, (size_t)bytes * N, c->ctx->stream);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const void* res = nullptr;
    const bool was_on = c->stats_on; c->stats_on = false;
    int rc = exchange(c, sbuf, rbuf, (size_t)bytes, &res, nullptr, nullptr, nullptr, 0, nullptr);              // warm-up (connections are set up on first use)
    if (rc == ORX_OK) {
        hipEventRecord(e0, c->ctx->stream);
        for (int r = 0; r < reps && rc == ORX_OK; ++r) rc = exchange(c, sbuf, rbuf, (size_t)bytes, &res, nullptr, nullptr, nullptr, 0, nullptr);
        hipEventRecord(e1, c->ctx->stream);
        hipStreamSynchronize(c->ctx->stream);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double sec = (double)ms * 1e-3, links = N > 1 ? N - 1 : 1;
        out3[0] = sec > 0 ? (double)bytes * links * reps / sec / 1e9 : 0.0;
        out3[1] = out3[0] / links;
        out3[2] = (double)ms * 1e3 / reps;
    }
    c->stats_on = was_on;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(sbuf); hipFree(rbuf);
    return rc;
}

extern "C" int orx_comm_unique_id(void* id_out) {
    ORX_ARG(id_out, "orx_comm_unique_id: NULL argument");
    RcclApi* api = rccl_api();
    ORX_ARG(api, "orx_comm_unique_id: librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == ORX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ORX_NCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return ORX_OK;
}

extern "C" int orx_comm_create(orx_ctx* ctx, const void* unique_id, int32_t rank, int32_t world, orx_comm** out) {
    ORX_ARG(ctx && out, "orx_comm_create: NULL argument");
    ORX_ARG(world >= 1 && world <= 64 && rank >= 0 && rank < world, "orx_comm_create: rank %d of world %d (at most 64 ranks)", rank, world);
    ORX_ARG(unique_id || world == 1, "orx_comm_create: %d ranks need the id that rank 0 made with orx_comm_unique_id", world);
    ORX_HIP(hipSetDevice(ctx->device));
    orx_comm* c = new orx_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    if (unique_id) {
        RcclApi* api = rccl_api();
        if (!api) { delete c; orx_set_error("orx_comm_create: librccl.so could not be loaded"); return ORX_ERR_ARG; }
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof(id));
        const ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) { delete c; orx_set_error("RCCL: ncclCommInitRank failed: %s", api->GetErrorString(r)); return ORX_ERR_HIP; }
    }
    *out = c;
    return ORX_OK;
}

extern "C" int orx_vgroup_create(int32_t world, orx_vgroup** out) {
    ORX_ARG(out && world >= 1 && world <= 64, "orx_vgroup_create: world in [1, 64]");
    orx_vgroup* g = new orx_vgroup();
    g->world = world;
    *out = g;
    return ORX_OK;
}

extern "C" int orx_vgroup_abort(orx_vgroup* g) {
    if (!g) return ORX_OK;
    { std::lock_guard<std::mutex> lk(g->mu); g->broken = true; }
    g->cv.notify_all();
    return ORX_OK;
}

extern "C" int orx_vgroup_destroy(orx_vgroup* g) { delete g; return ORX_OK; }

extern "C" int orx_comm_create_virtual(orx_ctx* ctx, orx_vgroup* group, int32_t rank, orx_comm** out) {
    ORX_ARG(ctx && group && out && rank >= 0 && rank < group->world, "orx_comm_create_virtual: bad argument");
    orx_comm* c = new orx_comm();
    c->ctx = ctx; c->vg = group; c->rank = rank; c->world = group->world;
    *out = c;
    return ORX_OK;
}

extern "C" int orx_comm_destroy(orx_comm* c) {
    if (!c) return ORX_OK;
    hipSetDevice(c->ctx->device);
    hipStreamSynchronize(c->ctx->stream);
    if (c->comm) rccl_api()->CommDestroy(c->comm);
    if (c->xstream) hipStreamDestroy(c->xstream);
    for (hipEvent_t e : c->ev) if (e) hipEventDestroy(e);
    for (Buf* b : {&c->send1, &c->mine, &c->tmp, &c->cnt, &c->send2, &c->req, &c->req_loc, &c->slot, &c->u_loc, &c->fu, &c->fv,
                   &c->rows_out, &c->rows_in, &c->gu, &c->u_apply, &c->send_g, &c->g_in, &c->dupref, &c->dsorted, &c->seglist, &c->segcount, &c->gdup, &c->bias_x,
                   &c->partials, &c->hot_ids, &c->hot_g, &c->hot_arange, &c->hot_sorted, &c->dl_send, &c->dl_slot, &c->dl_req, &c->dl_reqloc, &c->dl_rows_out, &c->dl_rows_in,
                   &c->dl_send_g, &c->dl_g_in, &c->dl_idx, &c->dl_ids, &c->dl_flat, &c->dl_sum, &c->dl_cnt, &c->dl_ptrs})
        if (b->p) hipFree(b->p);
    delete c;
    return ORX_OK;
}

extern "C" int orx_comm_rank(orx_comm* c) { return c ? c->rank : -1; }
extern "C" int orx_comm_world(orx_comm* c) { return c ? c->world : -1; }

// all-to-all of equal blocks: peer p gets send[p * bytes .. ), its block lands in recv[p * bytes .. ).  Returns where the result
// is: `recv`, or `send` itself for a one-rank communicator without RCCL.
// device copy of one block (the rank's own share of an exchange; the virtual group's transfers)
static int copy_block(const void* src, void* dst, size_t bytes, hipStream_t stream) {
    ORX_ARG(bytes % 4 == 0, "sharded engine: exchange blocks are 4-byte words");
    const bool wide = bytes % 16 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0;
    const size_t nw = wide ? bytes / 16 : bytes / 4;
    if (!nw) return ORX_OK;
    const dim3 grid((unsigned)std::min<size_t>((nw + 255) / 256, 2048));
    if (wide) hipLaunchKernelGGL(shard_copy_kernel<uint4>, grid, dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, (int64_t)nw);
    else hipLaunchKernelGGL(shard_copy_kernel<uint32_t>, grid, dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, (int64_t)nw);
    return ORX_OK;
}

// all-to-all of equal blocks: peer p gets send[p * bytes .. ), its block lands in recv[p * bytes .. ); optionally a second,
// smaller block per peer in the same group (the biases beside the rows).  `result` / `result2`: where the data is afterwards --
// `recv`, or `send` itself for a one-rank communicator without RCCL.
static int exchange(orx_comm* c, const void* send, void* recv, size_t bytes, const void** result, hipStream_t stream,
                    const void* send2, void* recv2, size_t bytes2, const void** result2) {
    if (!stream) stream = c->ctx->stream;
    *result = recv;
    if (result2) *result2 = recv2;
    const int nblk = send2 ? 2 : 1;
    const void* sv[2] = {send, send2}; void* rv[2] = {recv, recv2}; const size_t bv[2] = {bytes, bytes2};
    if (c->vg) {
        ORX_HIP(hipStreamSynchronize(stream));                      // my blocks are written
        c->vg->send[c->rank] = send; c->vg->send2[c->rank] = send2;
        ORX_ARG(c->vg->wait(), "virtual group: another rank failed");
        for (int p = 0; p < c->world; ++p) {
            CHECK(copy_block((const char*)c->vg->send[p] + (size_t)c->rank * bytes, (char*)recv + (size_t)p * bytes, bytes, stream));
            if (send2) CHECK(copy_block((const char*)c->vg->send2[p] + (size_t)c->rank * bytes2, (char*)recv2 + (size_t)p * bytes2, bytes2, stream));
        }
        ORX_HIP(hipStreamSynchronize(stream));                      // ... and read before their owners write them again
        ORX_ARG(c->vg->wait(), "virtual group: another rank failed");
        return ORX_OK;
    }
    if (!c->comm) { *result = send; if (result2) *result2 = send2; return ORX_OK; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->stats_on) {
        ORX_HIP(hipEventCreate(&e0)); ORX_HIP(hipEventCreate(&e1));
        ORX_HIP(hipEventRecord(e0, stream));
        c->st_exchanges += 1;
        static const bool self_on_wire = getenv("ORX_SHARD_RCCL_SELF") && atoi(getenv("ORX_SHARD_RCCL_SELF")) != 0;
        for (int q = 0; q < nblk; ++q) {                // (the rank's own block is a device copy unless ORX_SHARD_RCCL_SELF sends it through RCCL too)
            c->st_wire += (double)bv[q] * (c->world - 1 + (self_on_wire ? 1 : 0));
            c->st_self += self_on_wire ? 0.0 : (double)bv[q];
        }
    }
    struct Stop {                       // the stop event goes behind whatever this call put on the stream, on every way out
        orx_comm* c; hipEvent_t e0, e1; hipStream_t s;
        ~Stop() { if (e0) { hipEventRecord(e1, s); c->st_ev.emplace_back(e0, e1); } }
    } stop{c, e0, e1, stream};
    // this rank's own block is a copy kernel (RCCL's send-to-self kernel moved 18 MB in 25 us; hipMemcpyAsync cost ~100 us of runtime
    // bookkeeping per call); ORX_SHARD_RCCL_SELF=1 sends it through RCCL like any other (what the one-rank test uses to exercise
    // ncclSend / ncclRecv)
    static const bool rccl_self = getenv("ORX_SHARD_RCCL_SELF") && atoi(getenv("ORX_SHARD_RCCL_SELF")) != 0;
    if (!rccl_self)
        for (int q = 0; q < nblk; ++q) CHECK(copy_block((const char*)sv[q] + (size_t)c->rank * bv[q], (char*)rv[q] + (size_t)c->rank * bv[q], bv[q], stream));
    if (c->world == 1 && !rccl_self) return ORX_OK;
    RcclApi* api = rccl_api();
    ORX_NCCL(api, api->GroupStart());
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank && !rccl_self) continue;
        for (int q = 0; q < nblk; ++q) {
            ORX_NCCL(api, api->Send((const char*)sv[q] + (size_t)p * bv[q], bv[q], ncclInt8, p, c->comm, stream));
            ORX_NCCL(api, api->Recv((char*)rv[q] + (size_t)p * bv[q], bv[q], ncclInt8, p, c->comm, stream));
        }
    }
    ORX_NCCL(api, api->GroupEnd());
    return ORX_OK;
}

// the plan's per-step buckets x [Kc][N * cw words] through ONE all-to-all: regroup by peer, exchange, regroup by step
static int exchange_steps(orx_comm* c, const void* x, void* out, Buf& tmp, int Kc, int64_t cw, const void** result) {
    if (!c->comm && !c->vg) { *result = x; return ORX_OK; }
    const int N = c->world;
    const size_t bytes = (size_t)Kc * N * cw * 4;
    CHECK(ensure(c, tmp, 2 * bytes));
    uint32_t* a = (uint32_t*)tmp.p; uint32_t* b = a + (size_t)Kc * N * cw;
    const unsigned grid = (unsigned)std::min<int64_t>(((int64_t)Kc * N * cw + 255) / 256, 4096);
    ORX_LAUNCH(c->ctx, shard_regroup_kernel, dim3(grid), dim3(256), 0, (const uint32_t*)x, a, Kc, N, cw, 0);
    const void* r = nullptr;
    CHECK(exchange(c, a, b, (size_t)Kc * cw * 4, &r));
    ORX_LAUNCH(c->ctx, shard_regroup_kernel, dim3(grid), dim3(256), 0, (const uint32_t*)r, (uint32_t*)out, Kc, N, cw, 1);
    ORX_HIP(hipGetLastError());
    *result = out;
    return ORX_OK;
}

// (exported for the tests: the regrouping of a K-step plan's buckets, [K][world][words] <-> [world][K][words] 4-byte words)
extern "C" int orx_shard_regroup(orx_ctx* ctx, const void* src, void* dst, int64_t K, int32_t world, int64_t words, int back) {
    ORX_ARG(ctx && src && dst && K >= 0 && world >= 1 && words >= 0, "orx_shard_regroup: bad argument");
    if (K * world * words == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    const unsigned grid = (unsigned)std::min<int64_t>((K * world * words + 255) / 256, 4096);
    ORX_LAUNCH(ctx, shard_regroup_kernel, dim3(grid), dim3(256), 0, (const uint32_t*)src, (uint32_t*)dst, (int)K, (int)world, words, back);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

// capacity of one (source, destination) bucket for n elements spread over `world` ranks (openrec_amd/sharded.py::_cap)
static int64_t bucket_cap(int64_t n, int world, double slack) {
    const double mean = (double)n / world;
    return (int64_t)std::ceil(mean * slack + 6.0 * std::sqrt(mean) + 16.0);
}

extern "C" int orx_sharded_caps(int64_t B, int32_t world, float slack, int64_t* cap1, int64_t* cap2) {
    ORX_ARG(B > 0 && world >= 1 && cap1 && cap2, "orx_sharded_caps: bad argument");
    *cap1 = bucket_cap(B, world, slack);
    *cap2 = bucket_cap(2 * (int64_t)world * *cap1, world, slack);
    return ORX_OK;
}

static int all_reduce(orx_comm* c, float* x, int64_t n, Buf& tmp, Buf& ptrs);

// hot > 0: HOT-ITEM REPLICATION.  Items 0 .. hot-1 (a vocabulary sorted by popularity) live in the replica tables
// Vh [hot, D] / bh [hot, 1], identical on every rank.  Their references ask nobody: the request plan gives them slots in a region
// BEHIND the exchanged buckets of the row / gradient buffers, the rows come from the local replica, the gradient kernel reads and
// writes that region like any other slot; the step's gradients of the replicated rows are summed per item (segmented sums over the plan-time sorted slots: orx_csr_accum),
// then over the ranks by ONE all-reduce of the [hot, D + 4] block, and every rank applies the same sums to its replica.  Exact in
// TF's sense: duplicate ids are summed before the sparse apply, and a sum over ranks of per-rank sums is such a sum.
static int sharded_pairwise_impl(orx_comm* c, orx_opt* opt, int model, orx_table* U, orx_table* V, orx_table* b, orx_table* Vh, orx_table* bh,
                                 int64_t hot, float cold_fraction, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t K, int64_t B,
                                 int64_t id_stride, int64_t users_global, int64_t items_global, float margin, float slack,
                                 int32_t plan_chunk, int flags, double* loss_l2_accum, int32_t* overflow) {
    ORX_ARG(c && opt && U && V && b && uid && pid && nid && loss_l2_accum && overflow, "orx_sharded_pairwise_steps: NULL argument");
    ORX_ARG(hot >= 0 && hot <= items_global && hot < (1LL << 30), "orx_sharded_pairwise_steps: hot_items out of range");
    ORX_ARG(hot == 0 || (Vh && bh && Vh->ctx == c->ctx && bh->ctx == c->ctx && Vh->rows >= hot && bh->rows >= hot && Vh->dim == V->dim && bh->dim == 1),
            "orx_sharded_pairwise_steps: the replicas must be tables [hot_items, D] and [hot_items, 1] on the communicator's context");
    ORX_ARG(hot == 0 || c->world < 64, "orx_sharded_pairwise_steps: hot-item replication takes at most 63 ranks");
    ORX_ARG(hot == 0 || (cold_fraction > 0.f && cold_fraction <= 1.f), "orx_sharded_pairwise_steps: cold_fraction must be in (0, 1]");
    ORX_ARG(model == ORX_BPR || model == ORX_UCML, "orx_sharded_pairwise_steps: unknown model %d", model);
    ORX_ARG(K >= 0 && B > 0 && id_stride >= B && plan_chunk >= 1 && slack >= 1.0f, "orx_sharded_pairwise_steps: bad sizes");
    ORX_ARG(U->dim == V->dim && b->dim == 1 && b->rows == V->rows, "orx_sharded_pairwise_steps: table shapes do not match");
    const int D = U->dim;
    ORX_ARG(D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "orx_sharded_pairwise_steps: dim must be 16/32/64/128/256 (got %d)", D);
    orx_ctx* ctx = c->ctx;
    ORX_ARG(U->ctx == ctx && V->ctx == ctx && b->ctx == ctx, "orx_sharded_pairwise_steps: the tables belong to another context");
    const int N = c->world;
    ORX_ARG(U->rows >= (users_global - c->rank + N - 1) / N && V->rows >= (items_global - c->rank + N - 1) / N,
            "orx_sharded_pairwise_steps: the local shards are smaller than rows r = rank (mod world) of the global tables");
    ORX_ARG(!(hot && (flags & ORX_SHARD_DEDUP)), "orx_sharded_pairwise_steps_hot: ORX_SHARD_DEDUP cannot be combined with hot_items > 0 "
            "(the request plan that treats the replica as a destination is the one without per-destination dedup)");
    if (K == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    // Overlap: a step is cut into two half-batches, planned as separate lists, whose exchanges run on a second stream beside the
    // other half's kernels:   gather A | rows A || gather B | rows B || grads A | g A || grads B | g B || apply U | apply V(A) | apply V(B)
    // All gathers and gradient kernels of a step still precede its applies (TF's snapshot semantics); the duplicate flags of the apply
    // lists are taken over both halves together; Adagrad / Adam apply each table once per step from the halves' buffers side by side.
    const int H = ((flags & ORX_SHARD_OVERLAP) && (c->comm || c->vg) && c->world > 1 && B % 2 == 0 && id_stride == B) ? 2 : 1;
    // Per-destination dedup (an item several references of a list ask for travels once) costs a sort and an un-sort of the
    // references at plan time (~25 us per step at 131 k references): on by default where the item references a rank handles per list (2 B) are at least
    // half as many as the items (then most slots are shared), off for sparse lists (1 M items: 6 % of the references repeat)
    // (replication: the request plan with the replica as an extra destination is the one without dedup -- what is left for the wire
    // after the head of the distribution stayed at home repeats little)
    const bool dedup = hot ? false : (flags & ORX_SHARD_DEDUP) ? true : (flags & ORX_SHARD_NO_DEDUP) ? false : (4 * (B / H) >= items_global);
    const int64_t Bh = B / H;
    // (replication: the exchanged buckets are sized for the COLD share of a list's item references -- that is what takes the hot rows
    // off the wire, the buckets travel whole; a list with more cold references than that overflows and is reported as usual)
    const int64_t cap1 = bucket_cap(Bh, N, slack), T = N * cap1;
    const int64_t cap2 = bucket_cap(hot ? (int64_t)std::ceil(2.0 * (double)T * cold_fraction) : 2 * T, N, slack), M = N * cap2;
    const int64_t capH = hot ? 2 * T : 0;                 // every item reference of a list may be a hot one
    const int64_t Mx = M + capH;                          // rows of a list's row / gradient buffers: the buckets, then the replica's region
    const int DSh = U->dim + 4;
    // SGD: the biases travel apart from the rows (D + 1 floats per requested row on the wire; the rows stay 16-byte aligned);
    // Adagrad / Adam (whose applies take row + bias gradient as one row): row + bias column, D + 4 floats
    const bool split = opt->kind == ORX_SGD;
    const int DS = split ? D : D + 4, DSg = D + 4;
    const int64_t B_global = B * N;
    const bool sgd = opt->kind == ORX_SGD;
    const int gflags = flags & ORX_NO_L2;
    orx_table* tabs[5] = {U, V, b, Vh, bh};
    const int ntabs = hot ? 5 : 3;
    if (H == 2 && !c->xstream) {
        ORX_HIP(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
        for (hipEvent_t& e : c->ev) ORX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipStream_t S = ctx->stream, X = H == 2 ? c->xstream : ctx->stream;
    for (int64_t k0 = 0; k0 < K; k0 += plan_chunk) {
        const int Kc = (int)std::min<int64_t>(plan_chunk, K - k0);
        const int L = Kc * H;                             // lists of this chunk: half h of step k is list H k + h
        // (the buffers are sized for a whole chunk whatever K is: a loop's calls vary in length, and growing them -- a stream
        // synchronisation, hipFree and hipMalloc each -- inside a longer call is paid in that call)
        const size_t Lr = (size_t)plan_chunk * H;
        CHECK(ensure(c, c->send1, Lr * T * 3 * 4)); CHECK(ensure(c, c->mine, Lr * T * 3 * 4));
        CHECK(ensure(c, c->cnt, Lr * (N + 1) * 4));
        CHECK(ensure(c, c->send2, Lr * M * 4)); CHECK(ensure(c, c->req, Lr * M * 4)); CHECK(ensure(c, c->req_loc, Lr * M * 4));
        CHECK(ensure(c, c->slot, Lr * 2 * T * 4)); CHECK(ensure(c, c->u_loc, Lr * T * 4));
        CHECK(ensure(c, c->rows_out, (size_t)H * Mx * DS * 4)); CHECK(ensure(c, c->rows_in, (size_t)H * Mx * DS * 4));
        CHECK(ensure(c, c->send_g, (size_t)H * Mx * DS * 4)); CHECK(ensure(c, c->g_in, (size_t)H * M * DS * 4));
        if (hot) {
            CHECK(ensure(c, c->hot_ids, Lr * (size_t)capH * 4)); CHECK(ensure(c, c->hot_g, (size_t)hot * DSh * 4));
            CHECK(ensure(c, c->hot_sorted, Lr * (size_t)capH * 8));
            CHECK(orx_rows_sort_reserve(ctx, (int64_t)Lr, capH, hot));
            CHECK(orx_table_scratch(Vh)); CHECK(orx_table_scratch(bh));
            if (c->hot_arange_n != hot) {                 // 0 .. hot-1: the id list of the replica's apply
                CHECK(ensure(c, c->hot_arange, (size_t)hot * 4));
                std::vector<int32_t> ar((size_t)hot);
                for (int64_t i = 0; i < hot; ++i) ar[(size_t)i] = (int32_t)i;
                ORX_HIP(hipMemcpyAsync(c->hot_arange.p, ar.data(), (size_t)hot * 4, hipMemcpyHostToDevice, ctx->stream));
                ORX_HIP(hipStreamSynchronize(ctx->stream));
                c->hot_arange_n = hot;
            }
        }
        CHECK(ensure(c, c->gu, (size_t)H * T * D * 4)); CHECK(ensure(c, c->u_apply, (size_t)H * T * 4));
        const int nw_list = orx_shard_grads_nwaves(D, T);             // the loss partials of every list of a chunk: ONE accumulate launch per chunk
        ORX_ARG(nw_list > 0, "sharded engine: dim must be 16/32/64/128/256 (got %d)", D);
        CHECK(ensure(c, c->partials, Lr * (size_t)nw_list * 2 * 4));
        int nw_now = 0;
        if (split) CHECK(ensure(c, c->bias_x, (size_t)4 * H * Mx * 4));         // biases out | in | bias gradients out | in, [H][Mx] each
        if (sgd) {
            CHECK(ensure(c, c->fu, Lr * T)); CHECK(ensure(c, c->fv, Lr * M));
            // (the context's own scratch of orx_rows_dupflags, for a whole chunk as well)
            const size_t lists = (size_t)plan_chunk, n_ids = (size_t)H * std::max<int64_t>(T, M);
            CHECK(orx_ensure((void**)&ctx->d_dlist, &ctx->d_dlist_cap, lists * (n_ids / 2 + 1) * sizeof(uint32_t)));
            CHECK(orx_ensure((void**)&ctx->d_dcount, &ctx->d_dcount_cap, lists * sizeof(int)));
        }
        if (dedup) {
            CHECK(ensure(c, c->dupref, Lr * 2 * T)); CHECK(ensure(c, c->dsorted, Lr * 2 * T * 8)); CHECK(ensure(c, c->seglist, Lr * T * 8));
            CHECK(ensure(c, c->segcount, Lr * 4)); CHECK(ensure(c, c->gdup, (size_t)H * 2 * T * DSg * 4));
            CHECK(orx_rows_sort_reserve(ctx, (int64_t)Lr, 2 * T, (int64_t)N * ((items_global + N - 1) / N)));
            CHECK(orx_ensure((void**)&ctx->d_tmp, &ctx->d_tmp_cap, (Lr * 2 * T * 2 + Lr * ((2 * T + 1023) / 1024 + 64)) * sizeof(int32_t)));
        }
        int32_t* send1 = (int32_t*)c->send1.p; int32_t* cnt = (int32_t*)c->cnt.p; int32_t* send2 = (int32_t*)c->send2.p;
        int32_t* slot = (int32_t*)c->slot.p; int32_t* u_loc = (int32_t*)c->u_loc.p; int32_t* req_loc = (int32_t*)c->req_loc.p;
        unsigned char* fu = (unsigned char*)c->fu.p; unsigned char* fv = (unsigned char*)c->fv.p;
        // ---- the plan of the chunk's lists: routes 1 and 2 depend on the ids alone
        CHECK(orx_shard_route_steps(ctx, uid + k0 * id_stride, pid + k0 * id_stride, nid + k0 * id_stride, L, Bh, H == 2 ? Bh : id_stride,
                                    users_global, items_global, N, (int32_t)cap1, send1, cnt, overflow));
        const void* mine = nullptr;
        CHECK(exchange_steps(c, send1, c->mine.p, c->tmp, L, cap1 * 3, &mine));                   // 1. triplets -> user owner
        unsigned char* dupref = dedup ? (unsigned char*)c->dupref.p : nullptr;
        if (dedup) CHECK(orx_shard_request_dedup_steps(ctx, (const int32_t*)mine, L, T, N, (int32_t)cap2, items_global, send2, slot, u_loc, dupref,
                                                       c->dsorted.p, c->seglist.p, (int32_t*)c->segcount.p, overflow));
        else if (hot) {
            int32_t* hot_ids = (int32_t*)c->hot_ids.p;
            ORX_HIP(hipMemsetAsync(send2, 0xFF, (size_t)L * M * 4, S));
            ORX_HIP(hipMemsetAsync(hot_ids, 0xFF, (size_t)L * capH * 4, S));
            ORX_HIP(hipMemsetAsync(cnt, 0, (size_t)L * (N + 1) * 4, S));
            RequestArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.trip = (const int32_t*)mine; ra.T = T; ra.world = N; ra.cap = (int)cap2; ra.send_ids = send2; ra.slot = slot; ra.u_loc = u_loc;
            ra.counters = cnt; ra.overflow = overflow; ra.hot = (int)hot; ra.cap_hot = (int)capH; ra.hot_ids = hot_ids;
            CHECK(orx_launch_shard_request(ctx, ra, L));
            // the slots of every list sorted by item (stable in the slot index), once per plan: the per-item gradient sums of the steps
            // are then segmented sums over this order -- no atomics, the same sums in every run
            const uint2* hs = nullptr;
            CHECK(orx_rows_sort(ctx, hot_ids, L, capH, capH, hot, &hs));
            ORX_HIP(hipMemcpyAsync(c->hot_sorted.p, hs, (size_t)L * capH * sizeof(uint2), hipMemcpyDeviceToDevice, S));
        }
        else CHECK(orx_shard_request_steps(ctx, (const int32_t*)mine, L, T, N, (int32_t)cap2, send2, slot, u_loc, cnt, overflow));
        const void* req = nullptr;
        CHECK(exchange_steps(c, send2, c->req.p, c->tmp, L, cap2, &req));                          // 2. item ids -> item owner
        CHECK(orx_shard_localize(ctx, (const int32_t*)req, (int64_t)L * M, N, req_loc));
        if (sgd) {                                         // duplicate flags of every step's two apply lists, one launch each
            CHECK(orx_rows_dupflags(ctx, U->rows, u_loc, Kc, H * T, H * T, fu));
            CHECK(orx_rows_dupflags(ctx, V->rows, req_loc, Kc, H * M, H * M, fv));
        }
        for (int k = 0; k < Kc; ++k) {
            const void* rows_in[2] = {nullptr, nullptr};
            const void* g_in[2] = {nullptr, nullptr};
            const void* b_in[2] = {nullptr, nullptr};
            const void* gb_in[2] = {nullptr, nullptr};
            float* bx = (float*)c->bias_x.p;                 // [4][H][M]
            for (int h = 0; h < H; ++h) {                  // 3. owners gather row + bias, the rows travel back
                const int l = k * H + h;
                float* ro = (float*)c->rows_out.p + (size_t)h * Mx * DS;
                const int32_t* rl = req_loc + (size_t)l * M;
                if (split) {
                    CHECK(orx_table_touch(V, rl, M)); CHECK(orx_table_touch(b, rl, M));
                    CHECK(orx_launch_gather(ctx, V->w, b->w, V->rows, D, rl, M, ro, DS, ctx->d_err, 1, bx + (size_t)h * Mx));
                } else CHECK(orx_gather_rows(ctx, V, b, rl, M, ro, DS));
                if (H == 2) { ORX_HIP(hipEventRecord(c->ev[h], S)); ORX_HIP(hipStreamWaitEvent(X, c->ev[h], 0)); }
                if (split) CHECK(exchange(c, ro, (float*)c->rows_in.p + (size_t)h * Mx * DS, (size_t)cap2 * DS * 4, &rows_in[h], X,
                                          bx + (size_t)h * Mx, bx + (size_t)(H + h) * Mx, (size_t)cap2 * 4, &b_in[h]));
                else CHECK(exchange(c, ro, (float*)c->rows_in.p + (size_t)h * Mx * DS, (size_t)cap2 * DS * 4, &rows_in[h], X));
                if (H == 2) ORX_HIP(hipEventRecord(c->ev[2 + h], X));
                if (hot) {
                    // the references to replicated items read the local replica: its rows (and biases) go into the region behind the
                    // buckets of the buffer the gradient kernel will read (the exchange's receive buffer, or -- one rank without RCCL:
                    // the exchange is the identity -- the gather's own output); on the step stream, beside the exchange
                    const int32_t* hi = (const int32_t*)c->hot_ids.p + (size_t)l * capH;
                    float* dst = const_cast<float*>((const float*)rows_in[h]) + (size_t)M * DS;
                    CHECK(orx_table_touch(Vh, hi, capH)); CHECK(orx_table_touch(bh, hi, capH));
                    if (split) CHECK(orx_launch_gather(ctx, Vh->w, bh->w, Vh->rows, D, hi, capH, dst, DS, ctx->d_err, 1, const_cast<float*>((const float*)b_in[h]) + M));
                    else CHECK(orx_gather_rows(ctx, Vh, bh, hi, capH, dst, DS));
                }
            }
            for (int h = 0; h < H; ++h) {                  // 4. gradients (+ SGD's apply of the user rows referenced once); 6. item gradients leave
                const int l = k * H + h;
                const int32_t* ul = u_loc + (size_t)l * T; const int32_t* sl = slot + (size_t)l * 2 * T;
                float* gu = (float*)c->gu.p + (size_t)h * T * D; float* sg = (float*)c->send_g.p + (size_t)h * Mx * DS;
                float* gbo = split ? bx + (size_t)(2 * H + h) * Mx : nullptr;
                const unsigned char* dr = dedup ? dupref + (size_t)l * 2 * T : nullptr;
                const void* so = dedup ? (const char*)c->dsorted.p + (size_t)l * 2 * T * 8 : nullptr;
                const void* sgl = dedup ? (const char*)c->seglist.p + (size_t)l * T * 8 : nullptr;
                const int32_t* sgc = dedup ? (const int32_t*)c->segcount.p + l : nullptr;
                float* gd = dedup ? (float*)c->gdup.p + (size_t)h * 2 * T * DSg : nullptr;
                if (H == 2) ORX_HIP(hipStreamWaitEvent(S, c->ev[2 + h], 0));
                CHECK(orx_shard_grads_impl(ctx, model, sgd ? opt : nullptr, U, (const float*)rows_in[h], (const float*)b_in[h], ul, sl, dr, so, sgl, sgc, gd,
                                           sgd ? fu + (size_t)l * T : nullptr, T, DS, B_global, margin, gflags, gu,
                                           sgd ? (int32_t*)c->u_apply.p + (size_t)h * T : nullptr, sg, gbo, loss_l2_accum,
                                           (float*)c->partials.p + (size_t)l * 2 * (size_t)nw_list, &nw_now));
                ORX_ARG(nw_now == nw_list, "sharded engine: the gradient launch wrote %d loss partials, %d were expected", nw_now, nw_list);
                if (H == 2) { ORX_HIP(hipEventRecord(c->ev[4 + h], S)); ORX_HIP(hipStreamWaitEvent(X, c->ev[4 + h], 0)); }
                if (split) CHECK(exchange(c, sg, (float*)c->g_in.p + (size_t)h * M * DS, (size_t)cap2 * DS * 4, &g_in[h], X,
                                          gbo, bx + (size_t)(3 * H + h) * Mx, (size_t)cap2 * 4, &gb_in[h]));
                else CHECK(exchange(c, sg, (float*)c->g_in.p + (size_t)h * M * DS, (size_t)cap2 * DS * 4, &g_in[h], X));
                if (H == 2) ORX_HIP(hipEventRecord(c->ev[6 + h], X));
            }
            if (opt->kind == ORX_ADAM) CHECK(orx_opt_advance(opt, tabs, ntabs));   // Keras `iterations` += 1: after the step's gathers, before its applies
            if (sgd && H == 1) {                           // 5. the duplicated user rows (local) and the item-row gradients at their owners: one launch
                CHECK(orx_apply_rows_flagged_pair(ctx, opt, U, nullptr, (int32_t*)c->u_apply.p, T, (float*)c->gu.p, D, nullptr, fu + (size_t)k * T,
                                                  V, b, req_loc + (size_t)k * M, M, (const float*)g_in[0], DS, (const float*)gb_in[0], fv + (size_t)k * M));
            } else if (sgd) {                              // ... half by half: the user rows, then each half's item rows as its gradients arrive
                // (both halves' user lists in one launch: their duplicate flags were made over the two lists together)
                CHECK(orx_apply_rows_flagged_pair(ctx, opt, U, nullptr, (int32_t*)c->u_apply.p, T, (float*)c->gu.p, D, nullptr, fu + (size_t)(k * H) * T,
                                                  U, nullptr, (int32_t*)c->u_apply.p + (size_t)T, T, (float*)c->gu.p + (size_t)T * D, D, nullptr, fu + (size_t)(k * H + 1) * T, true));
                for (int h = 0; h < H; ++h) {              // item-row gradients at their owners
                    if (H == 2) ORX_HIP(hipStreamWaitEvent(S, c->ev[6 + h], 0));
                    CHECK(orx_apply_rows_flagged_impl(ctx, opt, V, b, req_loc + (size_t)(k * H + h) * M, M, (const float*)g_in[h], DS, (const float*)gb_in[h],
                                                      fv + (size_t)(k * H + h) * M));
                }
            } else {                                       // Adagrad / Adam sum a row's duplicates FIRST: one list per table and step
                CHECK(orx_apply_rows(ctx, opt, U, nullptr, u_loc + (size_t)k * H * T, H * T, (float*)c->gu.p, D));
                for (int h = 0; h < H; ++h) if (H == 2) ORX_HIP(hipStreamWaitEvent(S, c->ev[6 + h], 0));
                const float* g = H == 2 ? (const float*)c->g_in.p : (const float*)g_in[0];
                CHECK(orx_apply_rows(ctx, opt, V, b, req_loc + (size_t)k * H * M, H * M, g, DS));
            }
            if (hot) {
                // the replicated rows: this rank's gradients summed per item (both halves), ONE all-reduce of the block, the same apply on
                // every replica (a row nobody referenced carries a zero gradient: SGD / Adagrad leave it alone, TF-2.0 Adam decays it
                // like every other row)
                for (int h = 0; h < H; ++h) {
                    const uint2* hs = (const uint2*)c->hot_sorted.p + (size_t)(k * H + h) * capH;
                    const float* gr = (const float*)c->send_g.p + ((size_t)h * Mx + M) * DS;
                    CHECK(orx_csr_accum(ctx, Vh, hs, capH, gr, DS));
                    if (split) CHECK(orx_csr_accum(ctx, bh, hs, capH, bx + (size_t)(2 * H + h) * Mx + M, 1));
                    else CHECK(orx_csr_accum(ctx, bh, hs, capH, gr + D, DS));
                }
                CHECK(orx_launch_shard_hot_pack(ctx, Vh->gsum, bh->gsum, (float*)c->hot_g.p, hot, D, DSh));
                CHECK(all_reduce(c, (float*)c->hot_g.p, hot * DSh, c->dl_sum, c->dl_ptrs));
                CHECK(orx_apply_rows(ctx, opt, Vh, bh, (const int32_t*)c->hot_arange.p, hot, (const float*)c->hot_g.p, DSh));
            }
        }
        CHECK(orx_launch_loss_accumulate(ctx, (const float*)c->partials.p, (int64_t)L * nw_list, loss_l2_accum));
    }
    return ORX_OK;
}


extern "C" int orx_sharded_pairwise_steps(orx_comm* c, orx_opt* opt, int model, orx_table* U, orx_table* V, orx_table* b,
                                          const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t K, int64_t B,
                                          int64_t id_stride, int64_t users_global, int64_t items_global, float margin, float slack,
                                          int32_t plan_chunk, int flags, double* loss_l2_accum, int32_t* overflow) {
    return sharded_pairwise_impl(c, opt, model, U, V, b, nullptr, nullptr, 0, 1.0f, uid, pid, nid, K, B, id_stride, users_global, items_global, margin, slack,
                                 plan_chunk, flags, loss_l2_accum, overflow);
}

extern "C" int orx_sharded_pairwise_steps_hot(orx_comm* c, orx_opt* opt, int model, orx_table* U, orx_table* V, orx_table* b,
                                              orx_table* Vh, orx_table* bh, int64_t hot_items, float cold_fraction,
                                              const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t K, int64_t B,
                                              int64_t id_stride, int64_t users_global, int64_t items_global, float margin, float slack,
                                              int32_t plan_chunk, int flags, double* loss_l2_accum, int32_t* overflow) {
    return sharded_pairwise_impl(c, opt, model, U, V, b, Vh, bh, hot_items, cold_fraction, uid, pid, nid, K, B, id_stride, users_global, items_global, margin, slack,
                                 plan_chunk, flags, loss_l2_accum, overflow);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The hybrid-parallel DLRM step inside the library (the single-process step it shards: recommenders/dlrm.py:63-100
// under tf2_examples/dlrm_criteo.py:42-48).  Embedding rows are sharded by combined row id (row r on rank r % world), the MLPs are
// replicated; per step, all on the context's stream:
//   combined ids of this rank's B x n_emb lookups -> buckets by owner -> exchange (ids) -> owners gather -> exchange (rows) ->
//   forward + backward with the rows read where they arrived (orx_dlrm_grads_indirect) -> exchange (row gradients) ->
//   ncclAllReduce of the packed dense gradients (one buffer) -> owners apply the row gradients -> every replica applies the dense rule.
// Same arithmetic as the per-phase entry points driven by openrec_amd/sharded_dlrm.py (which the gloo tests keep driving).
int orx_dlrm_geometry(orx_dlrm* m, int* m_spa, int* n_emb, int* dense_dim, const int64_t** d_offset, const int64_t** d_rows);
orx_ctx* orx_dlrm_ctx(orx_dlrm* m);
int orx_launch_dlrm_ids(orx_ctx* ctx, const int32_t* sparse, const int64_t* offset, const int64_t* rows, int nf, int64_t B, int32_t* idx);

namespace {
// lookup (b, f) reads row slot[b, f] of the receive buffer; a lookup dropped by a full bucket (or an invalid id) reads the zero row
// `trash`; the pad column of the interaction's id matrix stays -1
__global__ __launch_bounds__(256) void dlrm_slot_idx_kernel(const int32_t* slot, int64_t n, int F, int32_t trash, int32_t* idx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int f = (int)(i % F);
        const int32_t s = slot[i];
        idx[i] = f == F - 1 ? -1 : (s >= 0 ? s : trash);
    }
}
__global__ __launch_bounds__(256) void vec_sum_kernel(const float* const* src, int N, float* dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float a = src[0][i];
        for (int p = 1; p < N; ++p) a += src[p][i];          // rank order: the same sum on every rank
        dst[i] = a;
    }
}
}  // namespace

// in-place sum of `x` over the ranks
static int all_reduce(orx_comm* c, float* x, int64_t n, Buf& tmp, Buf& ptrs) {
    hipStream_t S = c->ctx->stream;
    if (c->vg) {
        CHECK(ensure(c, tmp, (size_t)n * 4)); CHECK(ensure(c, ptrs, 64 * sizeof(void*)));
        ORX_HIP(hipStreamSynchronize(S));
        c->vg->red[c->rank] = x;
        ORX_ARG(c->vg->wait(), "virtual group: another rank failed");
        const void* h[64];
        for (int p = 0; p < c->world; ++p) h[p] = c->vg->red[p];
        ORX_HIP(hipMemcpyAsync(ptrs.p, h, c->world * sizeof(void*), hipMemcpyHostToDevice, S));
        ORX_LAUNCH(c->ctx, vec_sum_kernel, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0,
                   (const float* const*)ptrs.p, c->world, (float*)tmp.p, n);
        ORX_HIP(hipStreamSynchronize(S));                   // everybody has read everybody's vector ...
        ORX_ARG(c->vg->wait(), "virtual group: another rank failed");
        ORX_HIP(hipMemcpyAsync(x, tmp.p, (size_t)n * 4, hipMemcpyDeviceToDevice, S));      // ... before any of them is overwritten
        return ORX_OK;
    }
    if (!c->comm) return ORX_OK;                            // one rank without RCCL: the sum of one
    RcclApi* api = rccl_api();
    ORX_NCCL(api, api->AllReduce(x, x, (size_t)n, ncclFloat, ncclSum, c->comm, S));
    return ORX_OK;
}

extern "C" int orx_sharded_dlrm_steps(orx_comm* c, orx_dlrm* m, orx_opt* opt, orx_table* emb, const float* dense, const int32_t* sparse,
                                      const float* label, int64_t K, int64_t B, float slack, double* loss_accum, int32_t* overflow) {
    ORX_ARG(c && m && opt && emb && dense && sparse && label && loss_accum && overflow, "orx_sharded_dlrm_steps: NULL argument");
    ORX_ARG(K >= 0 && B > 0 && slack >= 1.0f, "orx_sharded_dlrm_steps: bad sizes");
    orx_ctx* ctx = c->ctx;
    ORX_ARG(orx_dlrm_ctx(m) == ctx && emb->ctx == ctx && opt->ctx == ctx, "orx_sharded_dlrm_steps: model, table and optimizer must live on the communicator's context");
    ORX_ARG(orx_dlrm_direct_ok(m), "orx_sharded_dlrm_steps: this model's shapes need the copying form of the local step (drive the per-phase entry points)");
    if (K == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    int d = 0, nf = 0, dd = 0; const int64_t *d_off = nullptr, *d_rows = nullptr;
    CHECK(orx_dlrm_geometry(m, &d, &nf, &dd, &d_off, &d_rows));
    ORX_ARG(emb->dim == d, "orx_sharded_dlrm_steps: the embedding shard has dim %d, the model %d", emb->dim, d);
    const int N = c->world, F = nf + 1;
    const int64_t n = B * nf, nF = B * F;
    const int64_t cap = (int64_t)std::ceil((double)n / N * slack) + 8, trash = (int64_t)N * cap;
    ORX_ARG(trash + 1 < (1LL << 31), "orx_sharded_dlrm_steps: batch too large");
    int64_t n_dense = 0;
    CHECK(orx_dlrm_dense_count(m, &n_dense));
    CHECK(ensure(c, c->dl_ids, (size_t)nF * 4)); CHECK(ensure(c, c->dl_send, (size_t)trash * 4)); CHECK(ensure(c, c->dl_slot, (size_t)nF * 4));
    CHECK(ensure(c, c->dl_req, (size_t)trash * 4)); CHECK(ensure(c, c->dl_reqloc, (size_t)trash * 4)); CHECK(ensure(c, c->dl_idx, (size_t)nF * 4));
    const size_t rows_out_before = c->dl_rows_out.cap;
    CHECK(ensure(c, c->dl_rows_out, (size_t)(trash + 1) * d * 4)); CHECK(ensure(c, c->dl_g_in, (size_t)trash * d * 4));
    const size_t rows_in_before = c->dl_rows_in.cap;
    CHECK(ensure(c, c->dl_rows_in, (size_t)(trash + 1) * d * 4)); CHECK(ensure(c, c->dl_send_g, (size_t)(trash + 1) * d * 4));
    CHECK(ensure(c, c->dl_flat, (size_t)n_dense * 4)); CHECK(ensure(c, c->dl_cnt, 64 * 4));
    hipStream_t S = ctx->stream;
    // (row `trash` of the receive buffer stands in for dropped lookups: it must read as zeros; the exchanges never write it)
    if (c->dl_rows_in.cap != rows_in_before) ORX_HIP(hipMemsetAsync(c->dl_rows_in.p, 0, c->dl_rows_in.cap, S));
    else ORX_HIP(hipMemsetAsync((float*)c->dl_rows_in.p + (size_t)trash * d, 0, (size_t)d * 4, S));
    // (a one-rank communicator without RCCL hands the send buffer back as "received": the zero row then lives behind the gathered rows)
    if (c->dl_rows_out.cap != rows_out_before) ORX_HIP(hipMemsetAsync(c->dl_rows_out.p, 0, c->dl_rows_out.cap, S));
    else ORX_HIP(hipMemsetAsync((float*)c->dl_rows_out.p + (size_t)trash * d, 0, (size_t)d * 4, S));
    int32_t* ids = (int32_t*)c->dl_ids.p; int32_t* send = (int32_t*)c->dl_send.p; int32_t* slot = (int32_t*)c->dl_slot.p;
    int32_t* req_loc = (int32_t*)c->dl_reqloc.p; int32_t* idx = (int32_t*)c->dl_idx.p;
    float* rows_out = (float*)c->dl_rows_out.p; float* rows_in = (float*)c->dl_rows_in.p; float* send_g = (float*)c->dl_send_g.p;
    float* flat = (float*)c->dl_flat.p;
    for (int64_t k = 0; k < K; ++k) {
        const float* de = dense + k * B * dd; const int32_t* sp = sparse + k * B * nf; const float* la = label + k * B;
        // 1. requests to the owners
        CHECK(orx_launch_dlrm_ids(ctx, sp, d_off, d_rows, nf, B, ids));                    // combined row ids [B][n_emb + 1] (pad column -1)
        CHECK(orx_shard_bucket(ctx, ids, nF, N, (int32_t)cap, send, slot, (int32_t*)c->dl_cnt.p, overflow));
        ORX_LAUNCH(ctx, dlrm_slot_idx_kernel, dim3((unsigned)std::min<int64_t>(1024, (nF + 255) / 256)), dim3(256), 0, (const int32_t*)slot, nF, F, (int32_t)trash, idx);
        const void* req = nullptr;
        CHECK(exchange(c, send, c->dl_req.p, (size_t)cap * 4, &req));
        CHECK(orx_shard_localize(ctx, (const int32_t*)req, trash, N, req_loc));
        // 2. owners gather, the rows travel back
        CHECK(orx_gather_rows(ctx, emb, nullptr, req_loc, trash, rows_out, d));
        const void* rin = nullptr;
        CHECK(exchange(c, rows_out, rows_in, (size_t)cap * d * 4, &rin));
        // 3. local forward + backward on the rows where they arrived; the gradient of lookup (b, f) goes to row idx[b, f] of send_g
        CHECK(orx_dlrm_grads_indirect(m, de, (const float*)rin, trash + 1, idx, la, B, B * N, send_g, loss_accum));
        // 4. row gradients back to the owners; dense gradients: one all-reduce
        const void* gin = nullptr;
        CHECK(exchange(c, send_g, c->dl_g_in.p, (size_t)cap * d * 4, &gin));
        CHECK(orx_dlrm_dense_pack(m, flat));
        CHECK(all_reduce(c, flat, n_dense, c->dl_sum, c->dl_ptrs));
        // 5. replicas apply the dense rule (Keras `iterations` += 1 happens there: before the row applies, as in the per-phase path),
        // owners apply the row gradients (padding slots carry garbage and are skipped: id -1)
        CHECK(orx_dlrm_dense_apply(m, opt, flat));
        CHECK(orx_apply_rows(ctx, opt, emb, nullptr, req_loc, trash, (const float*)gin, d));
    }
    return ORX_OK;
}
