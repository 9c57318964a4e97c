// DLRM model object and train step (C ABI: orx_dlrm_*).
// Host-side sequencing of: combined-table embedding gather, bottom MLP, feature
// interaction, top MLP, loss, full backward, sparse + dense optimizer apply.
// Restates openrec/tf2/recommenders/dlrm.py:8-100 and tf2_examples/dlrm_criteo.py:42-48.
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#include "orx_internal.h"
#include "orx_csr_device.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int _rc = (call);                                                              \
        if (_rc != ORX_OK) return _rc;                                                 \
    } while (0)

struct DenseLayer {
    int in = 0, out = 0, act = 0;       // act: 1 relu, 2 sigmoid
    orx_table* W = nullptr;             // [in, out]
    orx_table* b = nullptr;             // [1, out]
    // fp16 copies of W for ORX_DLRM_FP16_MLP: w16 [in][ld16] (operand of dY*W^T), w16t [out][ld16t] (operand of X*W)
    void* w16 = nullptr; void* w16t = nullptr;
    int ld16 = 0, ld16t = 0;
    // second-generation fp16 path (kernels_gemm16.hip)
    bool lean = false;                  // the layer's output lives only as its fp16 copy (no fp32 store): its consumers -- the next
                                        // layer's three products and the fused activation backward -- all read fp16
    bool dw16 = false;                  // the weight gradient comes from the batch-major fp16 copies (gemm16_tn)
    bool head = false;                  // the 1-unit last layer of the top MLP on the fused head kernels (kernels_gemm16.hip)
    float* slab = nullptr;              // split-K slices of this layer's weight gradient, [tiles][S][128 * 128]
    int slab_S = 1, slab_tiles = 0;
    uint64_t shadow_version = ~0ull;    // W->version the fp16 copies were made from
    // relu mask of the layer's fp16 output, one 64-bit word per lane and tile of the forward product (kernels_gemm16.hip Nt16Args::mask_out):
    // the input-gradient launch of the layer above applies relu' from it instead of fetching the output again
    unsigned long long* relu_mask = nullptr; int64_t mask_words = 0; int mask_cfg = 0; int64_t mask_B = 0;
    void* dz16 = nullptr;               // top MLP, deferred weight gradients (round 6): this layer's dZ16 keeps a buffer of its own until its
                                        // weight-gradient product has run (g16 / g16b are reused two layers further down)
    float* gbpart = nullptr;            // bias-gradient partial rows [row blocks][out] (ColPart), carved from orx_dlrm::colpart
    float* gwpart = nullptr;            // the head layer: weight-gradient partial rows [row blocks][in]
};

static inline int up8(int x) { return (x + 7) & ~7; }
// leading dimension of an fp16 operand whose rows are read in 64-deep K tiles (kernels_gemm16.hip: the LDS-DMA kernels take whole
// tiles; a row that ends inside one costs the slower zero-chunk form) -- rows of 64 halves and more are padded to whole tiles
static inline int up_k(int x) { return x >= 64 ? (x + 63) & ~63 : up8(x); }

struct orx_dlrm {
    orx_ctx* ctx = nullptr;
    int m_spa = 0, n_emb = 0, dense_dim = 0, flags = 0;
    float thr = 0.f;
    std::vector<int64_t> ln_emb, offset;
    int64_t* d_offset = nullptr;
    int2* d_colwin = nullptr;          // per dedup range of the combined table: (first id column, number of columns) that can hold its rows
    int64_t* d_rows = nullptr;
    orx_table* emb = nullptr;           // combined [sum(ln_emb), m_spa]
    std::vector<DenseLayer> bot, top;
    int F = 0, P = 0;
    int ldR = 0;                        // row stride of R / dR: d + P rounded up to 4 floats (16-byte rows)
    // activations / gradients, sized for `cap` samples
    int64_t cap = 0;
    float *d_dense = nullptr, *d_label = nullptr;
    int32_t *d_sparse = nullptr, *d_idx = nullptr;
    float *Z = nullptr, *dZ = nullptr, *R = nullptr;
    std::vector<float*> bot_y, top_y;   // outputs of every layer (bot last layer lives in Z)
    float *gA = nullptr, *gB = nullptr; // ping-pong gradient buffers [cap, maxwidth]
    // tables of <= TINY_ROWS rows: SGD gradient sums through LDS (dlrm_tiny_apply_kernel)
    std::vector<int> tiny_f; int tiny_max_rows = 0;
    int* d_tiny_f = nullptr; unsigned char* d_is_tiny = nullptr; int32_t* d_idx_big = nullptr;
    // fp16 copies of the activations that feed MLP products (ORX_DLRM_FP16_MLP)
    void* R16 = nullptr; int ldR16 = 0;
    std::vector<void*> top_y16;         // output of top layer l (l < last), [cap][up8(out)]
    void* g16 = nullptr, *g16b = nullptr;   // dY after the activation backward, [cap][up8(maxw)], ping-pong
    ShadowParam* d_shadow = nullptr; int n_shadow = 0; int64_t shadow_max = 0;
    const int32_t* direct_idx = nullptr;    // set by forward(): the interaction reads embedding rows through these ids (no gathered copy in Z)
    const float* direct_base = nullptr; int64_t direct_rows = 0;     // ... from this array (the combined table, or rows handed in)
    // hybrid-parallel step with the exchanged rows read in place (orx_dlrm_grads_indirect): rows, their count, where the gradients go
    const float* ext_rows = nullptr; int64_t ext_n = 0; float* ext_gdst = nullptr;
    bool gen2 = false;                  // ORX_DLRM_FP16_MLP with the kernels of kernels_gemm16.hip (ORX_DLRM_GEMM_V1 = the round-1 kernels)
    SlabReduce* d_slabjobs[2] = {nullptr, nullptr}; int n_slabjobs[2] = {0, 0}, slab_max_tiles[2] = {0, 0};   // [0] bottom, [1] top MLP
    void* dense16 = nullptr; int ld_dense16 = 0;   // fp16 copy of the dense features (operand of the first bottom layer)
    std::vector<void*> bot_y16;         // output of bottom layer l (l < last), [cap][up8(out)]
    float* colpart = nullptr;           // pool of the layers' partial rows
    int32_t* d_idx_all = nullptr; int64_t idx_all_cap = 0;          // combined row ids of a chunk of steps (planned sparse apply)
    unsigned char* d_single = nullptr; int64_t single_cap = 0;       // [chunk][B F]: lookups whose row nobody else references in their step
    // set by orx_dlrm_step for the backward of the current step: those rows are updated by the interaction backward itself
    const unsigned char* fuse_single = nullptr; orx_opt* fuse_opt = nullptr;
    // the loss folded into the head's backward (orx_dlrm_step, fp16 mode with the 1-unit head kernels): set for the backward of the
    // current step; `hl_used` says the head branch took it
    double* d_loss_part = nullptr; int64_t loss_part_cap = 0; const HeadLoss* cur_hl = nullptr; bool hl_used = false; bool head_fwd_in_bwd = false;
    // fp16 copies of the dense features of ALL steps of a call (ids on the device: one cast launch per call instead of one per step)
    void* dense16_all = nullptr; int64_t dense16_all_cap = 0; const void* dense16_cur = nullptr;
    int32_t* d_sparse_all = nullptr; int64_t sparse_all_cap = 0;
    DenseParam* d_params = nullptr;     // descriptors of the dense parameters for the multi-tensor optimizer launch
    orx_opt* params_opt = nullptr;
    DenseFused* d_fused = nullptr;      // ... and for the fused launch of the fp16 mode (slab reduce + rule + fp16 copies)
    orx_opt* fused_opt = nullptr; int fused_tiles = 0; DenseFusedTiles fused_tt;
    bool grads_pending = false;         // orx_dlrm_grads ran, orx_dlrm_dense_apply not yet
    // (round 6) the weight-gradient products of the top MLP only feed the optimizer: they are collected during the input-gradient chain and
    // run on a side stream BESIDE the interaction backward (HBM-bound, the matrix pipes idle) -- ORX_DLRM_DEFER_DW=1 | 2; measured SLOWER
    // than leaving them where they stand (ensure_buffers has the numbers): off by default, kept as the experiment's switch
    int defer_dw = 0; hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; bool side_pending = false;
    std::vector<std::function<int()>> deferred;
    std::vector<ColJob> pending_coljobs;   // fp16 mode of orx_dlrm_step: the step's partial-row sums, added by the fused optimizer launch
    std::vector<const float*> fused_out;   // descriptor index -> the gradient array (gsum) that descriptor applies
    std::vector<const float*> fused_cparts; std::vector<int> fused_cn;      // ... and its partial-row workspace / row length (NULL / 0: none)
    void* d_flatseg = nullptr; const void* flatseg_key = nullptr; int flatseg_n = 0;     // descriptors of dlrm_flat_kernel
    double* d_loss = nullptr;           // [Kcap]
    int64_t loss_cap = 0;
    int maxw = 0;
};

static int make_layers(orx_ctx* ctx, std::vector<DenseLayer>& L, int in, int n, const int32_t* units, int last_act, uint64_t seed) {
    for (int l = 0; l < n; ++l) {
        DenseLayer d;
        d.in = in; d.out = units[l]; d.act = (l == n - 1) ? last_act : 1;
        ORX_ARG(d.out > 0, "dlrm: layer width must be positive");
        CHECK(orx_table_create(ctx, d.in, d.out, &d.W));
        CHECK(orx_table_create(ctx, 1, d.out, &d.b));
        const float lim = std::sqrt(6.0f / (float)(d.in + d.out));          // Keras glorot_uniform
        CHECK(orx_table_init_uniform(d.W, -lim, lim, seed + 101 * (uint64_t)(l + 1)));
        CHECK(orx_table_fill(d.b, 0.f));
        L.push_back(d);
        in = d.out;
    }
    return ORX_OK;
}

extern "C" int orx_dlrm_create(orx_ctx* ctx, int32_t m_spa, int32_t n_emb, const int64_t* ln_emb,
                               int32_t n_bot, const int32_t* ln_bot, int32_t n_top, const int32_t* ln_top,
                               int32_t dense_dim, int flags, float loss_threshold, uint64_t seed, orx_dlrm** out) {
    ORX_ARG(ctx && out && ln_emb && ln_bot && ln_top, "orx_dlrm_create: NULL argument");
    ORX_ARG(m_spa > 0 && n_emb > 0 && n_bot > 0 && n_top > 0 && dense_dim > 0, "orx_dlrm_create: sizes must be positive");
    ORX_ARG(ln_bot[n_bot - 1] == m_spa, "orx_dlrm_create: the bottom MLP must end at m_spa=%d (got %d)", m_spa, ln_bot[n_bot - 1]);
    ORX_ARG(ln_top[n_top - 1] == 1, "orx_dlrm_create: the top MLP must end at 1 unit");
    ORX_HIP(hipSetDevice(ctx->device));
    orx_dlrm* m = new orx_dlrm();
    m->ctx = ctx; m->m_spa = m_spa; m->n_emb = n_emb; m->dense_dim = dense_dim; m->flags = flags; m->thr = loss_threshold;
    int64_t total = 0;
    for (int f = 0; f < n_emb; ++f) {
        ORX_ARG(ln_emb[f] > 0, "orx_dlrm_create: ln_emb[%d] must be positive", f);
        m->ln_emb.push_back(ln_emb[f]); m->offset.push_back(total); total += ln_emb[f];
    }
    if (!(flags & ORX_DLRM_NO_EMB)) {
        CHECK(orx_table_create(ctx, total, m_spa, &m->emb));
        CHECK(orx_table_init_uniform(m->emb, -0.05f, 0.05f, seed));        // LatentFactor 'uniform' (dlrm.py:32-33)
    }
    ORX_HIP(hipMalloc((void**)&m->d_offset, sizeof(int64_t) * n_emb));
    ORX_HIP(hipMalloc((void**)&m->d_rows, sizeof(int64_t) * n_emb));
    ORX_HIP(hipMemcpy(m->d_offset, m->offset.data(), sizeof(int64_t) * n_emb, hipMemcpyHostToDevice));
    ORX_HIP(hipMemcpy(m->d_rows, m->ln_emb.data(), sizeof(int64_t) * n_emb, hipMemcpyHostToDevice));
    if (m->emb) {   // column f of the id matrix only holds rows of table f: the duplicate analysis of a row range reads those columns only
        const int64_t R = orx_dedup_range_rows();
        std::vector<int2> win((size_t)orx_dedup_buckets(total));
        for (size_t bk = 0; bk < win.size(); ++bk) {
            const int64_t r0 = (int64_t)bk * R, r1 = r0 + R;
            int lo = n_emb, hi = -1;
            for (int f = 0; f < n_emb; ++f)
                if (m->offset[f] < r1 && m->offset[f] + m->ln_emb[f] > r0) { lo = std::min(lo, f); hi = std::max(hi, f); }
            win[bk] = make_int2(hi >= lo ? lo : 0, hi >= lo ? hi - lo + 1 : 0);
        }
        ORX_HIP(hipMalloc((void**)&m->d_colwin, sizeof(int2) * win.size()));
        ORX_HIP(hipMemcpy(m->d_colwin, win.data(), sizeof(int2) * win.size(), hipMemcpyHostToDevice));
    }
    {   // tiny tables (LDS must hold rows * m_spa floats)
        const int TINY_ROWS = 64;
        std::vector<unsigned char> is_tiny((size_t)n_emb + 1, 0);
        for (int f = 0; f < n_emb; ++f)
            if (ln_emb[f] <= TINY_ROWS && ln_emb[f] * (int64_t)m_spa * 4 <= 48 * 1024 && m_spa <= 256) {
                m->tiny_f.push_back(f); is_tiny[f] = 1; m->tiny_max_rows = std::max(m->tiny_max_rows, (int)ln_emb[f]);
            }
        if (!m->tiny_f.empty()) {
            ORX_HIP(hipMalloc((void**)&m->d_tiny_f, m->tiny_f.size() * sizeof(int)));
            ORX_HIP(hipMemcpy(m->d_tiny_f, m->tiny_f.data(), m->tiny_f.size() * sizeof(int), hipMemcpyHostToDevice));
            ORX_HIP(hipMalloc((void**)&m->d_is_tiny, is_tiny.size()));
            ORX_HIP(hipMemcpy(m->d_is_tiny, is_tiny.data(), is_tiny.size(), hipMemcpyHostToDevice));
        }
    }
    m->F = n_emb + 1;
    const bool itself = flags & ORX_DLRM_INTERACT_ITSELF;
    m->P = itself ? m->F * (m->F + 1) / 2 : m->F * (m->F - 1) / 2;
    CHECK(make_layers(ctx, m->bot, dense_dim, n_bot, ln_bot, (flags & ORX_DLRM_SIGMOID_BOT) ? 2 : 1, seed + 7));
    CHECK(make_layers(ctx, m->top, m_spa + m->P, n_top, ln_top, (flags & ORX_DLRM_SIGMOID_TOP) ? 2 : 1, seed + 13));
    if (flags & ORX_DLRM_FP16_MLP) {
        std::vector<ShadowParam> sp;
        auto add = [&](DenseLayer& D) -> int {
            D.ld16 = up8(D.out); D.ld16t = up_k(D.in);
            // (rows up to a multiple of 8, the extra ones zero: the input-gradient product of a 479-wide layer runs as 480 columns -- 16-byte
            // rows for its epilogue -- and leaves a zero in the padding column of its output)
            ORX_HIP(hipMalloc(&D.w16, (size_t)up8(D.in) * D.ld16 * 2)); ORX_HIP(hipMemset(D.w16, 0, (size_t)up8(D.in) * D.ld16 * 2));
            ORX_HIP(hipMalloc(&D.w16t, (size_t)D.out * D.ld16t * 2)); ORX_HIP(hipMemset(D.w16t, 0, (size_t)D.out * D.ld16t * 2));
            ShadowParam p; p.w = D.W->w; p.w16 = D.w16; p.w16t = D.w16t; p.in = D.in; p.out = D.out; p.ld16 = D.ld16; p.ld16t = D.ld16t;
            sp.push_back(p); m->shadow_max = std::max<int64_t>(m->shadow_max, (int64_t)D.in * D.out);
            return ORX_OK;
        };
        for (auto& D : m->top) CHECK(add(D));
        if (getenv("ORX_DLRM_GEMM_V1") == nullptr && getenv("ORX_DLRM_BOT_V1") == nullptr)
            for (auto& D : m->bot) CHECK(add(D));       // (round 1: the bottom MLP read fp32 operands)
        m->n_shadow = (int)sp.size();
        ORX_HIP(hipMalloc((void**)&m->d_shadow, sp.size() * sizeof(ShadowParam)));
        ORX_HIP(hipMemcpy(m->d_shadow, sp.data(), sp.size() * sizeof(ShadowParam), hipMemcpyHostToDevice));
    }
    m->gen2 = (flags & ORX_DLRM_FP16_MLP) && getenv("ORX_DLRM_GEMM_V1") == nullptr;
    if (m->gen2) {
        // layer l's relu output is fp16-only when layer l + 1 runs all three of its products on fp16 copies
        auto flags_of = [&](std::vector<DenseLayer>& L, int in0) {
            for (size_t l = 0; l < L.size(); ++l) {
                const int64_t ldx16 = up8(l == 0 ? in0 : L[l - 1].out);
                L[l].dw16 = L[l].w16 != nullptr && L[l].out >= 32 && orx_gemm16_tn_ok(ldx16, L[l].out, L[l].out);
            }
            for (size_t l = 0; l + 1 < L.size(); ++l)
                L[l].lean = L[l].act != 2 && L[l].out % 8 == 0 && L[l].out >= 64 && L[l + 1].dw16 &&
                            orx_gemm16_nt_ok(L[l + 1].out, L[l + 1].ld16, L[l + 1].in, L[l + 1].out) && getenv("ORX_DLRM_NO_LEAN") == nullptr;
        };
        flags_of(m->bot, dense_dim);
        flags_of(m->top, m_spa + m->P);
        const size_t nt_ = m->top.size();
        if (nt_ >= 2 && getenv("ORX_DLRM_NO_HEAD") == nullptr) {
            DenseLayer& H = m->top[nt_ - 1]; DenseLayer& Bl = m->top[nt_ - 2];
            H.head = H.out == 1 && H.w16t != nullptr && Bl.w16 != nullptr && Bl.act != 2 && Bl.out % 8 == 0 && orx_head16_ok(H.in, up8(Bl.out));
            if (H.head) Bl.lean = getenv("ORX_DLRM_NO_LEAN") == nullptr;          // nothing reads the fp32 output of the layer below the head
        }
    }
    m->ldR = (m_spa + m->P + 3) & ~3;
    m->maxw = m->ldR;
    for (auto& d : m->bot) m->maxw = std::max(m->maxw, std::max(d.in, d.out));
    for (auto& d : m->top) m->maxw = std::max(m->maxw, std::max(d.in, d.out));
    *out = m;
    return ORX_OK;
}

static void free_buffers(orx_dlrm* m) {
    hipFree(m->R16); hipFree(m->g16); hipFree(m->g16b); m->R16 = m->g16 = m->g16b = nullptr;
    for (auto& D : m->top) { hipFree(D.dz16); D.dz16 = nullptr; }
    for (int k = 0; k < 2; ++k) for (auto& D : (k == 0 ? m->bot : m->top)) { hipFree(D.relu_mask); D.relu_mask = nullptr; D.mask_words = 0; D.mask_cfg = 0; }
    for (auto& D : m->top) { hipFree(D.slab); D.slab = nullptr; }
    for (auto& D : m->bot) { hipFree(D.slab); D.slab = nullptr; }
    for (int k = 0; k < 2; ++k) { hipFree(m->d_slabjobs[k]); m->d_slabjobs[k] = nullptr; m->n_slabjobs[k] = 0; m->slab_max_tiles[k] = 0; }
    hipFree(m->dense16); m->dense16 = nullptr;
    hipFree(m->colpart); m->colpart = nullptr;
    m->fused_opt = nullptr;              // (the fused optimizer's descriptors point at the split-K workspaces)
    for (void* p : m->bot_y16) hipFree(p);
    m->bot_y16.clear();
    for (void* p : m->top_y16) hipFree(p);
    m->top_y16.clear();
    hipFree(m->d_dense); hipFree(m->d_label); hipFree(m->d_sparse); hipFree(m->d_idx); hipFree(m->d_idx_big); m->d_idx_big = nullptr;
    hipFree(m->Z); hipFree(m->dZ); hipFree(m->R); hipFree(m->gA); hipFree(m->gB);
    for (float* p : m->bot_y) hipFree(p);
    for (float* p : m->top_y) hipFree(p);
    m->bot_y.clear(); m->top_y.clear();
    m->d_dense = m->d_label = m->Z = m->dZ = m->R = m->gA = m->gB = nullptr;
    m->d_sparse = m->d_idx = nullptr;
    m->cap = 0;
}

extern "C" int orx_dlrm_destroy(orx_dlrm* m) {
    if (!m) return ORX_OK;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    free_buffers(m);
    hipFree(m->d_single); hipFree(m->d_loss_part); hipFree(m->dense16_all);
    if (m->side) { hipStreamSynchronize(m->side); hipStreamDestroy(m->side); }
    if (m->ev_fork) hipEventDestroy(m->ev_fork);
    if (m->ev_join) hipEventDestroy(m->ev_join);
    hipFree(m->d_offset); hipFree(m->d_colwin); hipFree(m->d_rows); hipFree(m->d_loss); hipFree(m->d_params); hipFree(m->d_fused); hipFree(m->d_idx_all); hipFree(m->d_sparse_all); hipFree(m->d_tiny_f); hipFree(m->d_is_tiny); hipFree(m->d_flatseg);
    orx_table_destroy(m->emb);
    for (auto& d : m->bot) { orx_table_destroy(d.W); orx_table_destroy(d.b); hipFree(d.w16); hipFree(d.w16t); }
    for (auto& d : m->top) { orx_table_destroy(d.W); orx_table_destroy(d.b); hipFree(d.w16); hipFree(d.w16t); }
    hipFree(m->d_shadow);
    delete m;
    return ORX_OK;
}

extern "C" int orx_dlrm_param(orx_dlrm* m, int kind, int layer, orx_table** out) {
    ORX_ARG(m && out, "orx_dlrm_param: NULL argument");
    if (kind == ORX_DLRM_EMB) {
        ORX_ARG(m->emb, "orx_dlrm_param: the model was created with ORX_DLRM_NO_EMB");
        *out = m->emb; return ORX_OK;
    }
    std::vector<DenseLayer>& L = (kind == ORX_DLRM_BOT_W || kind == ORX_DLRM_BOT_B) ? m->bot : m->top;
    ORX_ARG(kind >= ORX_DLRM_BOT_W && kind <= ORX_DLRM_TOP_B, "orx_dlrm_param: unknown kind %d", kind);
    ORX_ARG(layer >= 0 && layer < (int)L.size(), "orx_dlrm_param: layer %d out of range", layer);
    *out = (kind == ORX_DLRM_BOT_W || kind == ORX_DLRM_TOP_W) ? L[layer].W : L[layer].b;
    return ORX_OK;
}

static int ensure_buffers(orx_dlrm* m, int64_t B) {
    if (B <= m->cap) return ORX_OK;
    hipStreamSynchronize(m->ctx->stream);
    free_buffers(m);
    const size_t F = m->F, d = m->m_spa;
    ORX_HIP(hipMalloc((void**)&m->d_dense, sizeof(float) * B * m->dense_dim));
    ORX_HIP(hipMalloc((void**)&m->d_label, sizeof(float) * B));
    ORX_HIP(hipMalloc((void**)&m->d_sparse, sizeof(int32_t) * B * m->n_emb));
    ORX_HIP(hipMalloc((void**)&m->d_idx, sizeof(int32_t) * B * F));
    ORX_HIP(hipMalloc((void**)&m->d_idx_big, sizeof(int32_t) * B * F));
    ORX_HIP(hipMalloc((void**)&m->Z, sizeof(float) * B * F * d));
    ORX_HIP(hipMalloc((void**)&m->dZ, sizeof(float) * B * F * d));
    ORX_HIP(hipMalloc((void**)&m->R, sizeof(float) * B * m->ldR));
    ORX_HIP(hipMemsetAsync(m->R, 0, sizeof(float) * B * m->ldR, m->ctx->stream));
    ORX_HIP(hipMalloc((void**)&m->gA, sizeof(float) * B * m->maxw));
    ORX_HIP(hipMalloc((void**)&m->gB, sizeof(float) * B * m->maxw));
    for (size_t l = 0; l + 1 < m->bot.size(); ++l) {          // the last bottom layer writes into Z[:, F-1, :]
        float* p; ORX_HIP(hipMalloc((void**)&p, sizeof(float) * B * m->bot[l].out)); m->bot_y.push_back(p);
    }
    for (size_t l = 0; l < m->top.size(); ++l) {
        float* p; ORX_HIP(hipMalloc((void**)&p, sizeof(float) * B * m->top[l].out)); m->top_y.push_back(p);
    }
    {   // partial rows of the column sums: at most one per 32 samples (act_bwd_colsum's short slabs)
        const size_t maxP = (size_t)B / 32 + 2;
        size_t total = 0;
        for (int k = 0; k < 2; ++k) for (auto& D : (k == 0 ? m->bot : m->top)) total += maxP * (size_t)(D.out + (D.head ? D.in : 0));
        ORX_HIP(hipMalloc((void**)&m->colpart, total * sizeof(float)));
        float* p = m->colpart;
        for (int k = 0; k < 2; ++k)
            for (auto& D : (k == 0 ? m->bot : m->top)) {
                D.gbpart = p; p += maxP * D.out;
                D.gwpart = nullptr;
                if (D.head) { D.gwpart = p; p += maxP * D.in; }
            }
    }
    if (m->flags & ORX_DLRM_FP16_MLP) {
        m->ldR16 = (up_k(m->m_spa + m->P) + 127) / 128 * 128 <= up_k(m->m_spa + m->P) + 64 ? (up_k(m->m_spa + m->P) + 127) / 128 * 128 : up_k(m->m_spa + m->P);   // (whole 128-column tiles of the weight gradient where that costs <= 64 columns)
        ORX_HIP(hipMalloc(&m->R16, (size_t)B * m->ldR16 * 2)); ORX_HIP(hipMemsetAsync(m->R16, 0, (size_t)B * m->ldR16 * 2, m->ctx->stream));
        ORX_HIP(hipMalloc(&m->g16, (size_t)B * up8(m->maxw) * 2));
        ORX_HIP(hipMalloc(&m->g16b, (size_t)B * up8(m->maxw) * 2));
        for (size_t l = 0; l + 1 < m->top.size(); ++l) {
            void* p; const size_t bytes = (size_t)B * up8(m->top[l].out) * 2;
            ORX_HIP(hipMalloc(&p, bytes)); ORX_HIP(hipMemsetAsync(p, 0, bytes, m->ctx->stream));
            m->top_y16.push_back(p);
        }
    }
    if (m->gen2) {
        // relu masks for the layers whose activation backward is fused into the input-gradient product of the layer above (not the last layer of
        // an MLP, not the layer below the 1-unit head: their dZ comes from other kernels)
        for (int k = 0; k < 2; ++k) {
            std::vector<DenseLayer>& L = k == 0 ? m->bot : m->top;
            for (size_t l = 0; l + 1 < L.size(); ++l) {
                if (L[l].act != 1 || L[l].w16t == nullptr || L[l + 1].head || L[l].out % 8 != 0) continue;
                L[l].mask_words = (int64_t)((B + 127) / 128) * ((L[l].out + 63) / 64) * 256 + 1024;      // (the 128 x 64 tiles need the most words)
                ORX_HIP(hipMalloc((void**)&L[l].relu_mask, (size_t)L[l].mask_words * 8));
            }
        }
    }
    if (m->gen2) {
        // measured (profiles/r6_defer_dw.txt, one box): 0 = 476-477 us per step, 1 = 488-491, 2 = 494-496 -- beside each other the interaction
        // backward takes 88 us instead of 69, the 1024 x 512 weight gradient 69 instead of 17 (its 256 workgroups get a CU only as the 2048
        // queued workgroups of the interaction drain), the bottom MLP's first grouped launch 29 instead of 13.  Off.
        static const int defer_env = getenv("ORX_DLRM_DEFER_DW") ? atoi(getenv("ORX_DLRM_DEFER_DW")) : 0;
        m->defer_dw = defer_env;
        if (m->defer_dw > 0) {
            for (auto& D : m->top) if (D.dw16) ORX_HIP(hipMalloc(&D.dz16, (size_t)B * up8(D.out) * 2));
            if (!m->side) {
                ORX_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
                ORX_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
                ORX_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
            }
        }
    }
    if (m->gen2) {          // fp16 copies on the bottom side, weight-gradient split-K workspaces, descriptors of the reduce launches
        if (m->bot[0].w16) {
            m->ld_dense16 = up8(m->dense_dim);
            ORX_HIP(hipMalloc(&m->dense16, (size_t)B * m->ld_dense16 * 2));
            for (size_t l = 0; l + 1 < m->bot.size(); ++l) {
                void* p; const size_t bytes = (size_t)B * up8(m->bot[l].out) * 2;
                ORX_HIP(hipMalloc(&p, bytes)); ORX_HIP(hipMemsetAsync(p, 0, bytes, m->ctx->stream));
                m->bot_y16.push_back(p);
            }
        }
        for (int k = 0; k < 2; ++k) {
            std::vector<DenseLayer>& L = k == 0 ? m->bot : m->top;
            std::vector<SlabReduce> jobs;
            for (size_t l = 0; l < L.size(); ++l) {
                DenseLayer& D = L[l];
                if (!D.dw16) continue;
                int S, tiles, kchunk;
                orx_gemm16_tn_plan(m->ctx, D.in, D.out, (int)B, &S, &tiles, &kchunk);
                D.slab_S = S; D.slab_tiles = tiles;
                if (S > 1) {
                    ORX_HIP(hipMalloc((void**)&D.slab, (size_t)tiles * S * ORX_SLAB_STRIDE * sizeof(float)));
                    CHECK(orx_table_scratch(D.W));
                    SlabReduce j; j.slab = D.slab; j.C = D.W->gsum; j.ldc = D.out; j.M = D.in; j.N = D.out; j.S = S; j.ntn = (D.out + 127) / 128; j.tiles = tiles;
                    jobs.push_back(j); m->slab_max_tiles[k] = std::max(m->slab_max_tiles[k], tiles);
                }
            }
            m->n_slabjobs[k] = (int)jobs.size();
            if (!jobs.empty()) {
                ORX_HIP(hipMalloc((void**)&m->d_slabjobs[k], jobs.size() * sizeof(SlabReduce)));
                ORX_HIP(hipMemcpy(m->d_slabjobs[k], jobs.data(), jobs.size() * sizeof(SlabReduce), hipMemcpyHostToDevice));
            }
        }
    }
    m->cap = B;
    return ORX_OK;
}

struct Batch { const float* dense; const int32_t* sparse; const float* label; };

// MLP product: exact fp32 MFMA, or fp16 MFMA in the performance mode
static int mlp_gemm(orx_dlrm* m, const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1,
                    float* C, int64_t ldc, const float* bias, int M, int N, int K, int act, bool c_zero = false, float out_scale = 1.0f) {
    if (m->flags & ORX_DLRM_FP16_MLP) return orx_launch_gemm_f16(m->ctx, A, sa0, sa1, B, sb0, sb1, C, ldc, bias, M, N, K, act, c_zero, out_scale);
    return orx_launch_gemm(m->ctx, A, sa0, sa1, B, sb0, sb1, C, ldc, bias, M, N, K, act, c_zero, out_scale);
}

// fp16 mode: the backward pass carries S * gradient, S a power of two of the order of the batch the loss mean runs over (the
// loss kernel multiplies dLoss/dPred by it), so that the fp16 copies of the gradients sit in fp16's NORMAL range: dLoss/dPred
// is O(1 / B), 1e-5 at the C5 batch, where fp16 is subnormal and holds 2 to 7 bits.  Every fp32 sink divides by S again --
// exact both ways: weight gradients (slab reduce / product epilogue), bias gradients (partial-row reduce), the interaction's
// embedding gradients.  The exact fp32 mode runs with S = 1.
static float loss_scale(const orx_dlrm* m, int64_t n_mean) {
    if (!(m->flags & ORX_DLRM_FP16_MLP) || getenv("ORX_DLRM_NO_LOSS_SCALE") != nullptr) return 1.0f;
    float s = 1.0f;
    while (s < (float)n_mean && s < 32768.0f) s *= 2.0f;
    return s;
}

// forward of one batch; leaves every activation in the model's buffers.  emb_rows != NULL: the
// embedding rows [B, n_emb, d] are handed in (hybrid-parallel step) instead of gathered here.
static int forward(orx_dlrm* m, const Batch& bt, int64_t B, const float* emb_rows = nullptr, const int32_t* idx = nullptr,
                   bool touched = false) {
    orx_ctx* c = m->ctx;
    const int F = m->F, d = m->m_spa;
    const int compat = (m->flags & ORX_DLRM_REFERENCE_COMPAT) ? 1 : 0, itself = (m->flags & ORX_DLRM_INTERACT_ITSELF) ? 1 : 0;
    m->direct_idx = nullptr; m->direct_base = nullptr; m->direct_rows = 0;
    if (m->ext_rows != nullptr) {                     // rows of an exchange buffer, read in place through idx
        ORX_ARG(idx != nullptr && orx_interact_direct_ok(F, d, compat), "dlrm: rows in place need the MFMA interaction kernels");
        m->direct_idx = idx; m->direct_base = m->ext_rows; m->direct_rows = m->ext_n;
    } else if (emb_rows != nullptr) {
        CHECK(orx_launch_copy2d(c, m->Z, (int64_t)F * d, emb_rows, (int64_t)m->n_emb * d, (int)B, m->n_emb * d));
    } else {
        ORX_ARG(m->emb, "dlrm: the model was created with ORX_DLRM_NO_EMB (use orx_dlrm_grads)");
        if (idx == nullptr) {           // combined-table row ids of this batch (a K-step call computes them up front)
            CHECK(orx_launch_dlrm_ids(c, bt.sparse, m->d_offset, m->d_rows, m->n_emb, B, m->d_idx));
            idx = m->d_idx;
        }
        // a table under the lazy Adam: the rows about to be read are replayed to the optimizer's step first
        if (!touched) CHECK(orx_table_touch(m->emb, idx, B * F));
        // dlrm.py:83-85: the n_emb gathers = one gather on the combined table (dense slot skipped) -- or none at all: the
        // MFMA interaction kernels (forward and backward) read the rows from the table through idx
        m->direct_idx = orx_interact_direct_ok(F, d, compat) ? idx : nullptr;
        if (m->direct_idx) { m->direct_base = m->emb->w; m->direct_rows = m->emb->rows; }
        if (!m->direct_idx) CHECK(orx_launch_gather(c, m->emb->w, nullptr, m->emb->rows, d, idx, B * F, m->Z, d, c->d_err, 1));
    }
    const bool f16 = (m->flags & ORX_DLRM_FP16_MLP) != 0;
    // the relu-mask buffer of a layer for this batch size, or NULL (no buffer, or the product takes a tile form without masks); remembers the form
    auto mask_for = [&](DenseLayer& D, int64_t Bn) -> unsigned long long* {
        D.mask_cfg = 0; D.mask_B = Bn;
        if (D.relu_mask == nullptr) return nullptr;
        int64_t words = 0;
        const int cfg = orx_gemm16_nt_config(c, (int)Bn, D.out, &words);
        if (cfg == 0 || words > D.mask_words) return nullptr;
        D.mask_cfg = cfg;
        return D.relu_mask;
    };
    if (f16) {       // fp16 copies of the dense kernels: written by the fused optimizer launch; refreshed here when something else wrote a kernel
        bool stale = false;
        for (int k = 0; k < 2; ++k) for (auto& D : (k == 0 ? m->bot : m->top)) if (D.w16 != nullptr && D.shadow_version != D.W->version) stale = true;
        if (stale) {
            CHECK(orx_launch_dense_shadow(c, m->d_shadow, m->n_shadow, m->shadow_max));
            for (int k = 0; k < 2; ++k) for (auto& D : (k == 0 ? m->bot : m->top)) D.shadow_version = D.W->version;
        }
    }
    // dlrm.py:87: bottom MLP; its last layer writes straight into slot F-1 of Z
    const float* x = bt.dense; int64_t ldx = m->dense_dim;
    const bool bot16 = m->gen2 && m->dense16 != nullptr;
    const void* bx16 = m->dense16; int64_t ldbx16 = m->ld_dense16;
    if (bot16 && m->dense16_cur != nullptr) bx16 = m->dense16_cur;           // (cast for all steps of the call at once: orx_dlrm_step)
    else if (bot16) CHECK(orx_launch_cast16(c, bt.dense, m->dense_dim, m->dense16, m->ld_dense16, (int)B, m->dense_dim));
    for (size_t l = 0; l < m->bot.size(); ++l) {
        const DenseLayer& L = m->bot[l];
        const bool last = l + 1 == m->bot.size();
        float* y = last ? m->Z + (size_t)(F - 1) * d : m->bot_y[l];
        const int64_t ldy = last ? (int64_t)F * d : L.out;
        if (bot16 && bx16 != nullptr && orx_gemm16_nt_ok(ldbx16, L.ld16t, L.out, L.in)) {
            void* y16 = last ? nullptr : m->bot_y16[l];
            CHECK(orx_launch_gemm16_nt(c, bx16, ldbx16, L.w16t, L.ld16t, L.lean ? nullptr : y, ldy, y16, up8(L.out), L.b->w, (int)B, L.out, L.in, L.act,
                                       nullptr, nullptr, 0, 0, nullptr, mask_for(m->bot[l], B)));
            bx16 = y16; ldbx16 = up8(L.out);
        } else {
            ORX_ARG(l == 0 || !m->bot[l - 1].lean, "dlrm forward: bottom layer %d has no fp32 input", (int)l);
            CHECK(mlp_gemm(m, x, ldx, 1, L.W->w, L.out, 1, y, ldy, L.b->w, (int)B, L.out, L.in, L.act));
            bx16 = nullptr;
            if (bot16 && !last) {       // a layer off the fp16 kernels (width < 32, or fewer than 8 inputs): the layers above and this
                                        // layer's consumers in the backward pass (dw16) still read the fp16 copy of its output
                CHECK(orx_launch_cast16(c, y, ldy, m->bot_y16[l], up8(L.out), (int)B, L.out));
                bx16 = m->bot_y16[l]; ldbx16 = up8(L.out);
            }
        }
        x = y; ldx = ldy;
    }
    // dlrm.py:89-92: R = concat(dense_emb, interaction)
    bool have16 = false;                                         // does the current activation have an fp16 copy?
    CHECK(orx_launch_interact(c, true, m->Z, nullptr, F, d, compat, itself, m->R, m->P, B, m->ldR, f16 ? m->R16 : nullptr, m->ldR16, &have16,
                              m->direct_idx ? m->direct_base : nullptr, m->direct_idx, m->direct_idx ? m->direct_rows : 0));
    if (m->gen2 && !have16) {            // (the LDS interaction kernels -- reference_compat, odd shapes -- write fp32 only)
        CHECK(orx_launch_cast16(c, m->R, m->ldR, m->R16, m->ldR16, (int)B, m->m_spa + m->P));
        have16 = true;
    }
    x = m->R; ldx = m->ldR;
    const void* x16 = m->R16; int64_t ldx16 = m->ldR16;
    for (size_t l = 0; l < m->top.size(); ++l) {
        const DenseLayer& L = m->top[l];
        const bool last = l + 1 == m->top.size();
        if (f16 && have16) {            // X16 * W16T: fp16-resident operands, fp16 copy of the output for the next layer
            void* y16 = last ? nullptr : m->top_y16[l];
            if (L.head) {
                // (round 6) inside orx_dlrm_step with the loss folded into the head's backward, that kernel runs the head's forward too: no launch here
                if (!m->head_fwd_in_bwd) CHECK(orx_launch_head_fwd(c, x16, ldx16, L.w16t, L.b->w, L.act, m->top_y[l], (int)B, L.in));
            }
            else if (m->gen2 && orx_gemm16_nt_ok(ldx16, L.ld16t, L.out, L.in))
                CHECK(orx_launch_gemm16_nt(c, x16, ldx16, L.w16t, L.ld16t, L.lean ? nullptr : m->top_y[l], L.out, y16, up8(L.out), L.b->w, (int)B, L.out, L.in, L.act,
                                           nullptr, nullptr, 0, 0, nullptr, mask_for(m->top[l], B)));
            else
            CHECK(orx_launch_gemm_f16s(c, x16, ldx16, L.w16t, L.ld16t, m->top_y[l], L.out, y16, up8(L.out), L.b->w, (int)B, L.out, L.in, L.act));
            x16 = y16; ldx16 = up8(L.out); have16 = y16 != nullptr;
        } else {
            CHECK(mlp_gemm(m, x, ldx, 1, L.W->w, L.out, 1, m->top_y[l], L.out, L.b->w, (int)B, L.out, L.in, L.act));
            have16 = false;
        }
        x = m->top_y[l]; ldx = L.out;
    }
    return ORX_OK;
}

// SGD / Adagrad / Adam (lr_t of the step) on every dense parameter in one launch (gradients in the tables' gsum).
// fused (fp16 mode, orx_dlrm_step): the launch also adds the split-K slices of the weight gradients that mlp_backward left in
// the slab workspaces (no slab_reduce launches) and writes the fp16 copies of the new kernels (no dense_shadow launch).
static int dense_apply_all(orx_dlrm* m, orx_opt* opt, float lr_t = 0.f, bool fused = false, float slab_scale = 1.0f, const CsrFinish* finish = nullptr) {
    orx_ctx* c = m->ctx;
    if (fused) {
        if (m->fused_opt != opt) {
            std::vector<DenseFused> h;
            int tiles = 0;
            m->fused_out.clear(); m->fused_cparts.clear(); m->fused_cn.clear();
            DenseLayer* bias_of = nullptr;
            auto add = [&](orx_table* t, DenseLayer* D) -> int {
                OptSlots s;
                CHECK(orx_opt_slots(opt, t, &s));
                CHECK(orx_table_scratch(t));
                DenseFused p;
                memset(&p, 0, sizeof(p));
                p.w = t->w; p.acc = s.s0; p.acc2 = s.s1; p.g = t->gsum; p.rows = (int)t->rows; p.cols = t->dim;
                if (D != nullptr) {
                    if (D->dw16 && D->slab_S > 1) { p.slab = D->slab; p.S = D->slab_S; p.ntn = (D->out + 127) / 128; }
                    p.w16 = D->w16; p.w16t = D->w16t; p.ld16 = D->ld16; p.ld16t = D->ld16t;
                }
                if (p.rows == 1 || p.cols == 1) {             // (where its gradient's partial rows live, if it gets any: see below)
                    p.cN = p.rows * p.cols;
                    p.cparts = D != nullptr ? D->gwpart : bias_of->gbpart;
                }
                m->fused_out.push_back(t->gsum); m->fused_cparts.push_back(p.cparts); m->fused_cn.push_back(p.cN);
                p.tile0 = tiles; p.tiles_x = (p.cols + 63) / 64;
                tiles += p.tiles_x * ((p.rows + 15) / 16);        // (16 x 64 tiles: kernels_dense.hip DF_ROWS)
                h.push_back(p);
                return ORX_OK;
            };
            for (int k = 0; k < 2; ++k) for (auto& D : (k == 0 ? m->bot : m->top)) { CHECK(add(D.W, &D)); bias_of = &D; CHECK(add(D.b, nullptr)); }
            if (!m->d_fused) ORX_HIP(hipMalloc((void**)&m->d_fused, h.size() * sizeof(DenseFused)));
            ORX_HIP(hipMemcpyAsync(m->d_fused, h.data(), h.size() * sizeof(DenseFused), hipMemcpyHostToDevice, c->stream));
            ORX_HIP(hipStreamSynchronize(c->stream));
            ORX_ARG(h.size() <= 48, "dlrm: more than 24 dense layers");
            m->fused_tt.count = (int)h.size();
            for (size_t i = 0; i < h.size(); ++i) m->fused_tt.tile0[i] = h[i].tile0;
            m->fused_opt = opt; m->fused_tiles = tiles;
        }
        // the step's partial-row sums (bias gradients, the head's weight gradient): added inside the launch where the descriptor knows the
        // workspace, by a reduce launch otherwise
        for (int i = 0; i < m->fused_tt.count; ++i) m->fused_tt.cP[i] = 0;
        std::vector<ColJob> rest;
        for (const ColJob& j : m->pending_coljobs) {
            int idx = -1;
            for (size_t i = 0; i < m->fused_out.size(); ++i) if (m->fused_out[i] == j.out) { idx = (int)i; break; }
            if (idx >= 0 && m->fused_cparts[idx] == j.parts && m->fused_cn[idx] == j.N && j.scale == slab_scale && m->fused_tt.cP[idx] == 0 && j.P > 0) m->fused_tt.cP[idx] = j.P;
            else rest.push_back(j);
        }
        m->pending_coljobs.clear();
        if (!rest.empty()) CHECK(orx_launch_colparts_reduce(c, rest.data(), (int)rest.size()));
        if (opt->kind == ORX_ADAM) CHECK(orx_launch_dense_apply_fused(c, m->d_fused, m->fused_tt, m->fused_tiles, ORX_ADAM, lr_t, opt->p2, opt->p0, opt->p1, slab_scale, finish));
        else CHECK(orx_launch_dense_apply_fused(c, m->d_fused, m->fused_tt, m->fused_tiles, opt->kind, opt->lr, opt->p1, 0.f, 0.f, slab_scale, finish));
        for (int k = 0; k < 2; ++k) for (auto& D : (k == 0 ? m->bot : m->top)) { D.W->version += 1; D.b->version += 1; D.shadow_version = D.W->version; }
        return ORX_OK;
    }
    std::vector<DenseParam> h;
    int64_t max_n = 0;
    auto add = [&](orx_table* t) -> int {
        OptSlots s;
        CHECK(orx_opt_slots(opt, t, &s));
        CHECK(orx_table_scratch(t));
        DenseParam p; p.w = t->w; p.acc = s.s0; p.acc2 = s.s1; p.g = t->gsum; p.n = t->rows * t->dim;
        h.push_back(p); max_n = std::max(max_n, p.n);
        t->version += 1;
        return ORX_OK;
    };
    for (auto& D : m->bot) { CHECK(add(D.W)); CHECK(add(D.b)); }
    for (auto& D : m->top) { CHECK(add(D.W)); CHECK(add(D.b)); }
    if (m->d_params == nullptr || m->params_opt != opt) {       // pointers are stable: upload once per optimizer
        if (!m->d_params) ORX_HIP(hipMalloc((void**)&m->d_params, h.size() * sizeof(DenseParam)));
        ORX_HIP(hipMemcpyAsync(m->d_params, h.data(), h.size() * sizeof(DenseParam), hipMemcpyHostToDevice, c->stream));
        ORX_HIP(hipStreamSynchronize(c->stream));
        m->params_opt = opt;
    }
    if (opt->kind == ORX_ADAM) return orx_launch_dense_apply_multi(c, m->d_params, (int)h.size(), max_n, ORX_ADAM, lr_t, opt->p2, opt->p0, opt->p1);
    return orx_launch_dense_apply_multi(c, m->d_params, (int)h.size(), max_n, opt->kind, opt->lr, opt->p1);
}

// backward through one MLP; dy [B, last.out] is consumed (in place), returns d(input) in *dx_out.
// ins16 / outs16 (top MLP in fp16 mode, else NULL): the fp16 copies of every layer's input and output.
static int mlp_backward(orx_dlrm* m, std::vector<DenseLayer>& L, const std::vector<const float*>& ins, const std::vector<int64_t>& ld_in,
                        const std::vector<const float*>& outs, const std::vector<int64_t>& ld_out,
                        float* dy, float* other, int64_t B, bool need_dx0, float** dx_out, float gscale, bool defer_slabs,
                        const std::vector<const void*>* ins16 = nullptr, const std::vector<int64_t>* ld_in16 = nullptr,
                        const std::vector<const void*>* outs16 = nullptr, std::vector<ColJob>* jobs_out = nullptr,
                        const float* dy_src = nullptr, int64_t dy_src_ld = 0, float dy_src_scale = 1.0f) {
    orx_ctx* c = m->ctx;
    const int which = &L == &m->top ? 1 : 0;
    bool act_done = false;              // the activation backward of layer l was fused into the product above it
    void* dy16 = nullptr;               // fp16 copy of the current dy (g16 / g16b ping-pong)
    bool dy32 = true;                   // does `dy` hold the fp32 gradient (false: only dy16 is valid)
    int slabs = 0;
    std::vector<ColJob> coljobs;        // bias gradients (and the head's weight gradient) leave as partial rows per row block
    auto colpart_of = [&](float* parts) { ColPart cp; cp.parts = parts; return cp; };
    const float inv_scale = 1.0f / gscale;                  // dy and everything derived from it carries the loss scale
    // deferred weight gradients (top MLP of orx_dlrm_step's backward, see orx_dlrm::defer_dw): every layer's dZ16 in its own buffer
    const bool own_dz = which == 1 && m->defer_dw > 0 && m->side != nullptr && defer_slabs;
    auto add_job = [&](const ColPart& cp, float* out, int N) { ColJob j; j.parts = cp.parts; j.out = out; j.N = N; j.P = cp.P; j.scale = inv_scale; coljobs.push_back(j); };
    for (int l = (int)L.size() - 1; l >= 0; --l) {
        DenseLayer& D = L[l];
        CHECK(orx_table_scratch(D.W)); CHECK(orx_table_scratch(D.b));
        const bool want_dx = l > 0 || need_dx0;
        if (D.head && l > 0 && !act_done && ins16 && (*ins16)[l] && m->g16) {
            // the 1-unit head: activation backward, weight / bias gradient, input gradient and the activation backward of
            // the layer below in one pass over that layer's fp16 output
            DenseLayer& Bl = L[l - 1];
            CHECK(orx_table_scratch(Bl.b));
            const bool below16 = Bl.dw16 && orx_gemm16_nt_ok(Bl.out, Bl.ld16, Bl.in, Bl.out);
            ColPart pW = colpart_of(D.gwpart), pb = colpart_of(D.gbpart), pbb = colpart_of(Bl.gbpart);
            const HeadLoss* hl = (which == 1 && l == (int)L.size() - 1) ? m->cur_hl : nullptr;
            void* dz_below = (own_dz && Bl.dz16 && Bl.out % 8 == 0) ? Bl.dz16 : m->g16;
            CHECK(orx_launch_head_bwd(c, (*ins16)[l], (*ld_in16)[l], D.w16t, dy, outs[l], D.act, Bl.act, &pW, &pb,
                                      dz_below, Bl.out, below16 ? nullptr : other, ld_in[l], &pbb, (int)B, D.in, hl));
            if (hl) m->hl_used = true;
            add_job(pW, D.W->gsum, D.in); add_job(pb, D.b->gsum, 1); add_job(pbb, Bl.b->gsum, Bl.out);
            act_done = true; dy16 = dz_below; dy32 = !below16;
            float* t = dy; dy = other; other = t;
            continue;
        }
        const bool s16 = D.w16 != nullptr && m->g16 != nullptr && D.out % 8 == 0 && want_dx;
        // dZ = dY * act'(Y) and gb = colsum(dZ) in one pass (unless the product above already did it).  The gradient
        // buffers are zero here (the optimizer kernels zero them behind themselves), so slab sums / split-K just add.
        if (!act_done) {
            ORX_ARG(dy32, "dlrm backward: fp32 gradient missing for layer %d", l);
            dy16 = (s16 || (D.dw16 && m->g16 != nullptr)) ? m->g16 : nullptr;
            ColPart pb = colpart_of(D.gbpart);
            // (dy_src: the MLP's incoming gradient is still a strided, unscaled slice of another array -- read in place, once)
            CHECK(orx_launch_act_bwd_colsum(c, dy, outs[l], ld_out[l], (int)B, D.out, D.act, &pb, dy16, D.out, dy_src, dy_src_ld, dy_src_scale));
            dy_src = nullptr;
            add_job(pb, D.b->gsum, D.out);
        }
        act_done = false;
        // (round 6) layers whose two backward products cannot fill the chip alone run them in ONE launch (gemm16_group_kernel): both read dZ16
        const bool nt_here = want_dx && s16 && dy16 != nullptr && m->gen2 && orx_gemm16_nt_ok(D.out, D.ld16, D.in, D.out);
        const bool can_defer = own_dz && D.dw16 && D.dz16 != nullptr && dy16 == D.dz16 && ins16 && (*ins16)[l];
        const bool grouped = D.dw16 && nt_here && ins16 && (*ins16)[l] && orx_gemm16_group_ok(c, (int)B, D.in, D.out, (*ld_in16)[l], D.ld16) &&
                             !(can_defer && m->defer_dw >= 2);          // (ORX_DLRM_DEFER_DW=2: the grouped layers' weight gradients are deferred too)
        // gW [in, out] = X^T * dZ
        if (grouped) {
            if (D.slab_S > 1) ++slabs;
        } else if (D.dw16) {
            ORX_ARG(dy16 != nullptr && ins16 && (*ins16)[l], "dlrm backward: fp16 operands missing for layer %d", l);
            // (tried in round 3: these products on a second stream beside the input-gradient chain -- they only feed the optimizer.  The step
            // went from 0.6125 to 0.628 ms: side by side the products slow each other down by more than the overlap gains.  Round 6: beside
            // the INTERACTION backward instead -- HBM-bound, the matrix pipes idle -- see backward())
            const void* x16 = (*ins16)[l]; const int64_t ldx16 = (*ld_in16)[l]; const void* dz = dy16;
            DenseLayer* Dp = &D; const int Bi = (int)B;
            auto go = [c, x16, ldx16, dz, Dp, Bi, inv_scale]() -> int {
                return orx_launch_gemm16_tn(c, x16, ldx16, dz, Dp->out, Dp->W->gsum, Dp->out, Dp->slab, Dp->in, Dp->out, Bi, inv_scale);
            };
            if (can_defer) m->deferred.push_back(go); else CHECK(go());
            if (D.slab_S > 1) ++slabs;
        } else {
            ORX_ARG(dy32, "dlrm backward: fp32 gradient missing for layer %d", l);
            CHECK(mlp_gemm(m, ins[l], 1, ld_in[l], dy, D.out, 1, D.W->gsum, D.out, nullptr, D.in, D.out, (int)B, 0, true, inv_scale));
        }
        if (want_dx) {
            // dX [B, in] = dZ * W^T  (fp16-resident operands where they exist: dZ16, W16).  With a layer below, the
            // epilogue also applies that layer's activation backward, sums its bias gradient and writes its dZ16.
            if (s16 && dy16 != nullptr) {
                const bool fuse = l > 0 && L[l - 1].w16 != nullptr && L[l - 1].out % 8 == 0 && ld_in[l] == L[l - 1].out;
                void* next16 = (own_dz && l > 0 && L[l - 1].dz16 && L[l - 1].out % 8 == 0) ? L[l - 1].dz16 : ((dy16 == m->g16) ? m->g16b : m->g16);
                const bool nt = m->gen2 && orx_gemm16_nt_ok(D.out, D.ld16, D.in, D.out);
                if (fuse) {
                    CHECK(orx_table_scratch(L[l - 1].b));
                    if (nt) {
                        // the layer below runs both of its products on the fp16 copy: no fp32 store of its dZ
                        const bool below16 = L[l - 1].dw16 && (l - 1 == 0 ? !need_dx0 || orx_gemm16_nt_ok(L[0].out, L[0].ld16, L[0].in, L[0].out)
                                                                          : orx_gemm16_nt_ok(L[l - 1].out, L[l - 1].ld16, L[l - 1].in, L[l - 1].out));
                        const bool y16 = L[l - 1].lean;
                        ORX_ARG(!y16 || (outs16 && (*outs16)[l - 1]), "dlrm backward: fp16 activation missing for layer %d", l - 1);
                        ColPart pbb = colpart_of(L[l - 1].gbpart);
                        // the layer below's relu mask, if its forward launch wrote one in the tile form this launch takes
                        const unsigned long long* mask_in = nullptr;
                        if (y16 && L[l - 1].act == 1 && L[l - 1].relu_mask != nullptr && L[l - 1].mask_cfg != 0 && L[l - 1].mask_B == B &&
                            L[l - 1].mask_cfg == (grouped ? 3 : orx_gemm16_nt_config(c, (int)B, D.in, nullptr)))
                            mask_in = L[l - 1].relu_mask;
                        if (grouped)
                            CHECK(orx_launch_gemm16_group(c, (*ins16)[l], (*ld_in16)[l], dy16, D.out, D.W->gsum, D.out, D.slab, D.in, D.out, (int)B, inv_scale,
                                                          D.w16, D.ld16, below16 ? nullptr : other, ld_in[l], next16, L[l - 1].out,
                                                          y16 ? nullptr : outs[l - 1], y16 ? (*outs16)[l - 1] : nullptr,
                                                          y16 ? (int64_t)up8(L[l - 1].out) : ld_out[l - 1], L[l - 1].act, &pbb, mask_in));
                        else
                        CHECK(orx_launch_gemm16_nt(c, dy16, D.out, D.w16, D.ld16, below16 ? nullptr : other, ld_in[l], next16, L[l - 1].out,
                                                   nullptr, (int)B, D.in, D.out, 0, y16 ? nullptr : outs[l - 1], y16 ? (*outs16)[l - 1] : nullptr,
                                                   y16 ? (int64_t)up8(L[l - 1].out) : ld_out[l - 1], L[l - 1].act, &pbb, nullptr, mask_in));
                        add_job(pbb, L[l - 1].b->gsum, L[l - 1].out);
                        dy16 = next16; dy32 = !below16;
                    } else {
                        ORX_ARG(!L[l - 1].lean, "dlrm backward: layer %d has no fp32 activation", l - 1);
                        ColPart pbb = colpart_of(L[l - 1].gbpart);
                        CHECK(orx_launch_gemm_f16s(c, dy16, D.out, D.w16, D.ld16, other, ld_in[l], (l - 1 > 0 || need_dx0) ? next16 : nullptr, L[l - 1].out,
                                                   nullptr, (int)B, D.in, D.out, 0, outs[l - 1], ld_out[l - 1], L[l - 1].act, &pbb));
                        add_job(pbb, L[l - 1].b->gsum, L[l - 1].out);
                        dy16 = (l - 1 > 0 || need_dx0) ? next16 : nullptr; dy32 = true;
                    }
                    act_done = true;
                } else {
                    // (the output's padding columns are computed too -- zeros, the operand's extra rows are zero -- where that gives the product whole
                    // 16-byte rows: 479 -> 480 columns of dR)
                    const int in_cols = (ld_in[l] >= up8(D.in) && getenv("ORX_DLRM_NO_PAD_DX") == nullptr) ? up8(D.in) : D.in;
                    if (nt && grouped)
                        CHECK(orx_launch_gemm16_group(c, (*ins16)[l], (*ld_in16)[l], dy16, D.out, D.W->gsum, D.out, D.slab, D.in, D.out, (int)B, inv_scale,
                                                      D.w16, D.ld16, other, ld_in[l], nullptr, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, in_cols));
                    else if (nt) CHECK(orx_launch_gemm16_nt(c, dy16, D.out, D.w16, D.ld16, other, ld_in[l], nullptr, 0, nullptr, (int)B, in_cols, D.out, 0));
                    else CHECK(orx_launch_gemm_f16s(c, dy16, D.out, D.w16, D.ld16, other, ld_in[l], nullptr, 0, nullptr, (int)B, D.in, D.out, 0));
                    dy32 = true;
                }
            } else {
                ORX_ARG(dy32, "dlrm backward: fp32 gradient missing for layer %d", l);
                CHECK(mlp_gemm(m, dy, D.out, 1, D.W->w, 1, D.out, other, ld_in[l], nullptr, (int)B, D.in, D.out, 0));
            }
            float* t = dy; dy = other; other = t;
        }
    }
    // (jobs_out: the caller reduces the partial rows of both MLPs with one launch)
    if (jobs_out) jobs_out->insert(jobs_out->end(), coljobs.begin(), coljobs.end());
    else CHECK(orx_launch_colparts_reduce(c, coljobs.data(), (int)coljobs.size()));
    if (slabs > 0) {
        ORX_ARG(slabs == m->n_slabjobs[which], "dlrm backward: %d of %d split-K weight gradients were produced", slabs, m->n_slabjobs[which]);
        // (deferred: the fused optimizer launch adds the slices itself -- orx_dlrm_step in fp16 mode)
        if (!defer_slabs) CHECK(orx_launch_slab_reduce(c, m->d_slabjobs[which], m->n_slabjobs[which], m->slab_max_tiles[which], inv_scale));
    }
    *dx_out = dy;
    return ORX_OK;
}

// backward of the batch whose activations `forward` left behind; m->gA holds dLoss/dPred.
// Leaves dZ [B, F, d] (slot F-1 = d dense_emb) and every dense gradient in its table's gsum.
static int backward(orx_dlrm* m, const Batch& bt, int64_t B, float gscale, bool defer_slabs = false) {
    orx_ctx* c = m->ctx;
    m->deferred.clear();
    const int F = m->F, d = m->m_spa;
    const int compat = (m->flags & ORX_DLRM_REFERENCE_COMPAT) ? 1 : 0, itself = (m->flags & ORX_DLRM_INTERACT_ITSELF) ? 1 : 0;
    {
    // ---- top MLP backward
    std::vector<const float*> ins, outs; std::vector<int64_t> ldi, ldo;
    for (size_t l = 0; l < m->top.size(); ++l) {
        ins.push_back(l == 0 ? m->R : m->top_y[l - 1]); ldi.push_back(l == 0 ? m->ldR : m->top[l - 1].out);
        outs.push_back(m->top_y[l]); ldo.push_back(m->top[l].out);
    }
    float* dR = nullptr;
    std::vector<const void*> ins16, outs16; std::vector<int64_t> ldi16;
    if (m->gen2)
        for (size_t l = 0; l < m->top.size(); ++l) {
            ins16.push_back(l == 0 ? m->R16 : m->top_y16[l - 1]); ldi16.push_back(l == 0 ? m->ldR16 : up8(m->top[l - 1].out));
            outs16.push_back(l + 1 < m->top.size() ? m->top_y16[l] : nullptr);
        }
    std::vector<ColJob> coljobs;                         // bias gradients (and the head's weight gradient) of both MLPs: one reduce launch at the end
    CHECK(mlp_backward(m, m->top, ins, ldi, outs, ldo, m->gA, m->gB, B, true, &dR, gscale, defer_slabs, m->gen2 ? &ins16 : nullptr, &ldi16, &outs16, &coljobs));
    // ---- interaction backward: dZ for every slot (slot F-1 = d dense_emb)
    // (dR carries the loss scale; dZ -- the embedding rows' gradients -- leaves unscaled)
    if (!m->deferred.empty()) {
        // the top MLP's deferred weight gradients start with the interaction backward, on the side stream
        ORX_HIP(hipEventRecord(m->ev_fork, c->stream));
        ORX_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
        hipStream_t main_stream = c->stream;
        c->stream = m->side;
        int rc = ORX_OK;
        for (auto& f : m->deferred) { rc = f(); if (rc != ORX_OK) break; }
        c->stream = main_stream;
        m->deferred.clear();
        CHECK(rc);
        ORX_HIP(hipEventRecord(m->ev_join, m->side));
        m->side_pending = true;
    }
    {
        // (round 6) the rows referenced once in the step take their SGD / Adagrad update inside this launch (kernels_dense.hip FusedRows)
        const bool fuse = m->fuse_single != nullptr && m->direct_idx != nullptr && m->ext_rows == nullptr && m->emb != nullptr && m->direct_base == m->emb->w;
        float* acc_rows = nullptr;
        if (fuse && m->fuse_opt->kind == ORX_ADAGRAD) { OptSlots st; CHECK(orx_opt_slots(m->fuse_opt, m->emb, &st)); acc_rows = st.s0; }
        CHECK(orx_launch_interact(c, false, m->Z, dR, F, d, compat, itself, m->dZ, m->P, B, m->ldR, nullptr, 0, nullptr,
                                  m->direct_idx ? m->direct_base : nullptr, m->direct_idx, m->direct_idx ? m->direct_rows : 0, 1.0f / gscale,
                                  m->ext_rows ? m->ext_gdst : nullptr, fuse ? m->fuse_single : nullptr, fuse ? m->fuse_opt->kind : 0,
                                  fuse ? m->fuse_opt->lr : 0.f, fuse ? m->fuse_opt->p1 : 0.f, acc_rows));
    }
    // ---- bottom MLP backward from dZ[:, F-1, :]
    float* dy = (dR == m->gA) ? m->gB : m->gA;
    float* other = (dy == m->gA) ? m->gB : m->gA;
    ins.clear(); outs.clear(); ldi.clear(); ldo.clear();
    for (size_t l = 0; l < m->bot.size(); ++l) {
        const bool last = l + 1 == m->bot.size();
        ins.push_back(l == 0 ? bt.dense : m->bot_y[l - 1]); ldi.push_back(l == 0 ? m->dense_dim : m->bot[l - 1].out);
        outs.push_back(last ? m->Z + (size_t)(F - 1) * d : m->bot_y[l]); ldo.push_back(last ? (int64_t)F * d : m->bot[l].out);
    }
    float* dx0 = nullptr;
    ins16.clear(); outs16.clear(); ldi16.clear();
    const bool bot16 = m->gen2 && m->dense16 != nullptr;
    if (bot16)
        for (size_t l = 0; l < m->bot.size(); ++l) {
            ins16.push_back(l == 0 ? (m->dense16_cur ? m->dense16_cur : m->dense16) : m->bot_y16[l - 1]); ldi16.push_back(l == 0 ? m->ld_dense16 : up8(m->bot[l - 1].out));
            outs16.push_back(l + 1 < m->bot.size() ? m->bot_y16[l] : nullptr);
        }
    // (the bottom MLP's dY = gscale * dZ[:, F-1, :]: its first activation backward reads the slice in place)
    CHECK(mlp_backward(m, m->bot, ins, ldi, outs, ldo, dy, other, B, false, &dx0, gscale, defer_slabs, bot16 ? &ins16 : nullptr, &ldi16, &outs16, &coljobs,
                       m->dZ + (size_t)(F - 1) * d, (int64_t)F * d, gscale));
    // (defer_slabs = the fused optimizer launch follows: it adds the partial rows itself, see dense_apply_all)
    if (defer_slabs && getenv("ORX_DLRM_COLPARTS_LAUNCH") == nullptr) m->pending_coljobs = coljobs;
    else CHECK(orx_launch_colparts_reduce(c, coljobs.data(), (int)coljobs.size()));
    }
    if (m->side_pending) {                                   // the optimizer launch reads the deferred products' slabs
        ORX_HIP(hipStreamWaitEvent(c->stream, m->ev_join, 0));
        m->side_pending = false;
    }
    return ORX_OK;
}

static int stage(orx_dlrm* m, const float* dense, const int32_t* sparse, const float* label, int64_t B, int flags, Batch* out) {
    if (flags & ORX_IDS_DEVICE) { out->dense = dense; out->sparse = sparse; out->label = label; return ORX_OK; }
    hipStream_t s = m->ctx->stream;
    ORX_HIP(hipMemcpyAsync(m->d_dense, dense, sizeof(float) * B * m->dense_dim, hipMemcpyHostToDevice, s));
    ORX_HIP(hipMemcpyAsync(m->d_sparse, sparse, sizeof(int32_t) * B * m->n_emb, hipMemcpyHostToDevice, s));
    if (label) ORX_HIP(hipMemcpyAsync(m->d_label, label, sizeof(float) * B, hipMemcpyHostToDevice, s));
    out->dense = m->d_dense; out->sparse = m->d_sparse; out->label = m->d_label;
    return ORX_OK;
}

extern "C" int orx_dlrm_step(orx_dlrm* m, orx_opt* opt, const float* dense, const int32_t* sparse, const float* label,
                             int64_t K, int64_t B, int flags, float* loss_out) {
    ORX_ARG(m && opt && (K == 0 || (dense && sparse && label)), "orx_dlrm_step: NULL argument");
    ORX_ARG(K >= 0 && B > 0, "orx_dlrm_step: K must be >= 0 and B > 0");
    if (K == 0) return ORX_OK;
    orx_ctx* c = m->ctx;
    ORX_HIP(hipSetDevice(c->device));
    CHECK(ensure_buffers(m, B));
    if (m->loss_cap < K) {
        if (m->d_loss) ORX_HIP(hipFree(m->d_loss));
        ORX_HIP(hipMalloc((void**)&m->d_loss, sizeof(double) * K)); m->loss_cap = K;
    }
    const int F = m->F, d = m->m_spa;
    // Sparse optimizer: the row ids of all K steps are known up front, so the (row, position) pairs of every step's id list
    // are SORTED once per chunk of steps (kernels_rowsort.hip) and each step applies its gradient rows as segmented sums in
    // position order -- no fp32 atomics, no arrival-order ranks: a step is bit-reproducible, whatever the duplicate
    // structure (Criteo has tables of 3 rows next to tables of 10 M).  ORX_ROWS_ATOMICS=1 keeps the round-2 paths (plan with
    // arrival-order staging / LDS sums for the tiny tables / scatter-add atomics) for A/B measurements.
    const bool atomics = getenv("ORX_ROWS_ATOMICS") != nullptr || d > 256;
    const bool sorted_apply = m->emb != nullptr && !atomics;
    const bool planned = atomics && m->emb != nullptr && (opt->kind == ORX_SGD || opt->kind == ORX_ADAGRAD) && orx_fused_can_inline_apply(d) &&
                         orx_dedup_buckets(m->emb->rows) <= 8 && getenv("ORX_DLRM_NO_PLAN") == nullptr;
    // steps per plan / sort: what orx_exact_buffers accepts as one chunk for B*F ids per step; <= 8 M sorted entries
    const int64_t PC = sorted_apply ? std::max<int64_t>(1, std::min<int64_t>(64, (8LL << 20) / (B * F)))
                                    : std::max<int64_t>(1, std::min<int64_t>(64, (int64_t)((256ull << 20) / ((size_t)3 * B * F * sizeof(int32_t)))));
    RowsPlan rp;
    const uint2* sorted = nullptr;
    const int32_t* sparse_dev = sparse;
    const bool lazy_adam = m->emb != nullptr && orx_adam_rows_lazy(opt, m->emb);
    { orx_table* mine[1] = {m->emb}; CHECK(orx_opt_isolate(opt, mine, m->emb ? 1 : 0)); }   // (a shared optimizer: api.hip orx_opt_isolate)
    ColWindows cw;
    if (getenv("ORX_DLRM_NO_COLWIN") == nullptr) { cw.F = F; cw.win = m->d_colwin; }
    if (planned || sorted_apply) {
        // sized for a whole chunk whatever K is: a loop's calls vary in length (warm-up calls of 5 steps, then one of 20), and growing
        // these inside the longer call put ~0.5 ms of hipFree / hipMalloc into it
        const int64_t kp = sorted_apply ? PC : std::min<int64_t>(K, PC);
        if (sorted_apply) CHECK(orx_rows_sort_reserve(c, PC, B * F, m->emb->rows));
        if (m->idx_all_cap < kp * B * F) {
            if (m->d_idx_all) ORX_HIP(hipFree(m->d_idx_all));
            ORX_HIP(hipMalloc((void**)&m->d_idx_all, sizeof(int32_t) * kp * B * F)); m->idx_all_cap = kp * B * F;
        }
        if (!(flags & ORX_IDS_DEVICE)) {                      // all K steps' sparse ids to the device at once
            if (m->sparse_all_cap < K * B * m->n_emb) {
                if (m->d_sparse_all) ORX_HIP(hipFree(m->d_sparse_all));
                ORX_HIP(hipMalloc((void**)&m->d_sparse_all, sizeof(int32_t) * K * B * m->n_emb)); m->sparse_all_cap = K * B * m->n_emb;
            }
            ORX_HIP(hipMemcpyAsync(m->d_sparse_all, sparse, sizeof(int32_t) * K * B * m->n_emb, hipMemcpyHostToDevice, c->stream));
            sparse_dev = m->d_sparse_all;
        }
    }
    if (sorted_apply && !lazy_adam) CHECK(orx_table_sync(m->emb));
    // (round 6) SGD / Adagrad: a row that a single lookup of the step references is updated by the interaction backward itself
    // (kernels_dense.hip FusedRows); the sorted list names those lookups for free.  ORX_DLRM_NO_FUSED_SPARSE=1: every row through the sorted apply
    const bool fuse_rows = sorted_apply && !lazy_adam && (opt->kind == ORX_SGD || opt->kind == ORX_ADAGRAD) &&
                           orx_interact_fuse_ok(F, d, (m->flags & ORX_DLRM_REFERENCE_COMPAT) ? 1 : 0);
    if (fuse_rows && m->single_cap < PC * B * F) {
        if (m->d_single) ORX_HIP(hipFree(m->d_single));
        ORX_HIP(hipMalloc((void**)&m->d_single, (size_t)(PC * B * F))); m->single_cap = PC * B * F;
    }
    // (round 6) fewer launches per step: the loss is folded into the head's backward where the 1-unit head kernels run (its value leaves
    // as per-workgroup partials, summed once per call), and the fp16 copy of the dense features is made for all K steps at once
    const bool fold_loss = m->gen2 && m->top.size() >= 2 && m->top.back().head && m->g16 != nullptr && getenv("ORX_DLRM_NO_FOLDED_LOSS") == nullptr;
    const int nb_loss = orx_head_bwd_blocks(c, (int)B);
    if (fold_loss && m->loss_part_cap < K * nb_loss) {
        if (m->d_loss_part) ORX_HIP(hipFree(m->d_loss_part));
        const int64_t cap = std::max<int64_t>(K, 64) * nb_loss;
        ORX_HIP(hipMalloc((void**)&m->d_loss_part, sizeof(double) * cap)); m->loss_part_cap = cap;
    }
    const bool cast_all = m->gen2 && m->dense16 != nullptr && (flags & ORX_IDS_DEVICE) && getenv("ORX_DLRM_CAST_PER_STEP") == nullptr;
    if (cast_all) {
        if (m->dense16_all_cap < K * B) {
            if (m->dense16_all) ORX_HIP(hipFree(m->dense16_all));
            const int64_t cap = std::max<int64_t>(K, 64) * B;
            ORX_HIP(hipMalloc(&m->dense16_all, (size_t)cap * m->ld_dense16 * 2)); m->dense16_all_cap = cap;
        }
        ORX_ARG(K * B < (1LL << 31), "orx_dlrm_step: K * B must stay below 2^31");
        CHECK(orx_launch_cast16(c, dense, m->dense_dim, m->dense16_all, m->ld_dense16, (int)(K * B), m->dense_dim));
    }
    struct Restore { orx_dlrm* m; ~Restore() { m->dense16_cur = nullptr; m->cur_hl = nullptr; m->fuse_single = nullptr; m->fuse_opt = nullptr; } } restore{m};
    static const bool no_carry = getenv("ORX_DLRM_FINISH_LAUNCH") != nullptr;
    static const bool head_fwd_launch = getenv("ORX_DLRM_HEAD_FWD_LAUNCH") != nullptr;
    for (int64_t s = 0; s < K; ++s) {
        m->dense16_cur = cast_all ? (const char*)m->dense16_all + (size_t)s * B * m->ld_dense16 * 2 : nullptr;
        if ((planned || sorted_apply) && s % PC == 0) {
            const int64_t kp = std::min<int64_t>(K - s, PC);
            CHECK(orx_launch_dlrm_ids(c, sparse_dev + s * B * m->n_emb, m->d_offset, m->d_rows, m->n_emb, kp * B, m->d_idx_all));
            if (planned) CHECK(orx_apply_rows_plan(c, m->emb, m->d_idx_all, kp, B * F, B * F, &rp));
            else CHECK(orx_rows_sort(c, m->d_idx_all, kp, B * F, B * F, m->emb->rows, &sorted));
            if (fuse_rows) CHECK(orx_rows_single_flags(c, sorted, kp, B * F, m->emb->rows, m->d_single));
        }
        const int32_t* idx_s = (planned || sorted_apply) ? m->d_idx_all + (s % PC) * B * F : nullptr;
        const uint2* sorted_s = sorted_apply ? sorted + (s % PC) * B * F : nullptr;
        Batch bt;
        CHECK(stage(m, dense + s * B * m->dense_dim, sparse + s * B * m->n_emb, label + s * B, B, flags, &bt));
        // lazy Adam: the rows of this batch are replayed to the current step before the forward reads
        // them; the duplicate analysis of the id list is shared with the apply below
        bool deduped = false;
        if (lazy_adam) {
            if (!sorted_apply) {
                CHECK(orx_launch_dlrm_ids(c, bt.sparse, m->d_offset, m->d_rows, m->n_emb, B, m->d_idx));
                idx_s = m->d_idx;
            }
            if (m->emb->lazy == opt && opt->t > 0) {
                if (sorted_apply) CHECK(orx_adam_rows_sorted(c, opt, m->emb, sorted_s, B * F, nullptr, 0, false));
                else CHECK(orx_adam_rows_touch(c, opt, m->emb, idx_s, B * F, false, cw));
                deduped = getenv("ORX_DLRM_NO_SHARED_DEDUP") == nullptr;
            } else {
                CHECK(orx_table_sync(m->emb));
            }
        }
        // (round 6) with the loss folded into the head's backward, that launch also runs the head's FORWARD (one pass over the layer below's
        // output instead of two): the forward pass below skips its head launch.  ORX_DLRM_HEAD_FWD_LAUNCH=1: head_fwd_kernel as before
        const bool head_in_bwd = fold_loss && !head_fwd_launch;
        m->head_fwd_in_bwd = head_in_bwd;
        const int rc_fwd = forward(m, bt, B, nullptr, idx_s, lazy_adam);
        m->head_fwd_in_bwd = false;
        CHECK(rc_fwd);
        float* pred = m->top_y.back();
        // loss + dLoss/dP  (dlrm.py:72-73, :97-98)
        const float gscale = loss_scale(m, B);
        HeadLoss hl{bt.label, (m->flags & ORX_DLRM_LOSS_BCE) ? 1 : 0, m->thr, B, gscale, fold_loss ? m->d_loss_part + s * nb_loss : nullptr,
                    head_in_bwd ? m->top.back().b->w : nullptr, head_in_bwd ? pred : nullptr};
        if (!fold_loss) CHECK(orx_launch_dlrm_loss(c, pred, bt.label, B, (m->flags & ORX_DLRM_LOSS_BCE) ? 1 : 0, m->thr, m->gA, m->d_loss + s, 0, 0, gscale));
        m->cur_hl = fold_loss ? &hl : nullptr; m->hl_used = false;
        const bool fuse_dense = m->gen2 && getenv("ORX_DLRM_NO_FUSED_DENSE") == nullptr;
        // (round 6) SGD / Adagrad with the fused dense optimizer launch behind it: the sorted apply's finish pass rides in THAT launch (kernels_dense.hip
        // dense_apply_fused_kernel) instead of being the step's 24th.  ORX_DLRM_FINISH_LAUNCH=1: its own launch
        CsrFinish fin;
        memset(&fin, 0, sizeof(fin));
        const bool carry_finish = fuse_dense && sorted_apply && !lazy_adam && (opt->kind == ORX_SGD || opt->kind == ORX_ADAGRAD) && !no_carry;
        m->fuse_single = fuse_rows ? m->d_single + (s % PC) * B * F : nullptr; m->fuse_opt = opt;
        const int rc_bwd = backward(m, bt, B, gscale, fuse_dense);
        const bool rows_fused = m->fuse_single != nullptr && m->direct_idx != nullptr;      // (backward applied them: see there)
        m->fuse_single = nullptr; m->fuse_opt = nullptr; m->cur_hl = nullptr;
        CHECK(rc_bwd);
        ORX_ARG(!fold_loss || m->hl_used, "dlrm step: the loss was to be folded into the head's backward, which did not run");
        // (tried, round 4: the sorted sparse apply on a second stream beside the bottom MLP's backward and the dense apply -- 0.578 against
        // 0.580 ms per step: the launches fill the chip one after the other either way.)
        // ---- optimizer: one step counter for all variables (Keras `iterations`)
        opt->t += 1;
        float lr_t = 0.f;
        if (opt->kind == ORX_ADAM) {
            const double b1 = opt->p0, b2 = opt->p1;
            lr_t = (float)(opt->lr * std::sqrt(1.0 - std::pow(b2, (double)opt->t)) / (1.0 - std::pow(b1, (double)opt->t)));
        }
        // sparse: per-occurrence rows dZ[b, f, :] onto the combined table (dense slot has id -1)
        if (sorted_apply) {
            if (lazy_adam) CHECK(orx_adam_rows_sorted(c, opt, m->emb, sorted_s, B * F, m->dZ, d, true));
            else if (opt->kind == ORX_ADAM) CHECK(orx_adam_dense_sorted(c, opt, m->emb, sorted_s, B * F, m->dZ, d));
            else if (carry_finish) CHECK(orx_csr_apply_split(c, opt, m->emb, sorted_s, B * F, m->dZ, d, rows_fused, &fin));
            else CHECK(orx_csr_apply(c, opt, m->emb, sorted_s, B * F, m->dZ, d, rows_fused));
        } else if (planned) {
            CHECK(orx_apply_rows_planned_step(c, opt, m->emb, nullptr, rp, s % PC, idx_s, m->dZ, d));
        } else if (lazy_adam) {
            CHECK(orx_adam_rows_apply(c, opt, m->emb, idx_s, B * F, m->dZ, d, deduped, cw));
        } else if (opt->kind == ORX_ADAGRAD) {
            CHECK(orx_table_sync(m->emb));
            CHECK(orx_adagrad_rows_apply(c, opt, m->emb, nullptr, m->d_idx, B * F, m->dZ, d, cw));
        } else if (opt->kind == ORX_SGD && !m->tiny_f.empty()) {
            // tiny tables: per-slab LDS sums; the generic scatter then skips their slots
            CHECK(orx_launch_dlrm_tiny_apply(c, m->d_idx, m->dZ, m->d_tiny_f, (int)m->tiny_f.size(), m->tiny_max_rows, m->d_offset,
                                             m->d_rows, F, d, B, opt->lr, m->emb->w));
            CHECK(orx_launch_dlrm_mask_tiny(c, m->d_idx, m->d_is_tiny, F, B * F, m->d_idx_big));
            CHECK(orx_apply_rows(c, opt, m->emb, nullptr, m->d_idx_big, B * F, m->dZ, d));
        } else {
            CHECK(orx_apply_rows(c, opt, m->emb, nullptr, m->d_idx, B * F, m->dZ, d));
        }
        CHECK(dense_apply_all(m, opt, lr_t, fuse_dense, 1.0f / gscale, fin.blocks > 0 ? &fin : nullptr));
    }
    if (fold_loss) CHECK(orx_launch_head_loss_finish(c, m->d_loss_part, nb_loss, nb_loss, K, B, m->d_loss));
    if (loss_out) {
        std::vector<double> h((size_t)K);
        ORX_HIP(hipMemcpyAsync(h.data(), m->d_loss, sizeof(double) * K, hipMemcpyDeviceToHost, c->stream));
        ORX_HIP(hipStreamSynchronize(c->stream));
        for (int64_t s = 0; s < K; ++s) loss_out[s] = (float)h[s];
    }
    if (!(flags & ORX_IDS_DEVICE)) return orx_check_index_error(c);
    return ORX_OK;
}

extern "C" int orx_dlrm_inference(orx_dlrm* m, const float* dense, const int32_t* sparse, int64_t B, int flags, float* pred_out) {
    ORX_ARG(m && dense && sparse && pred_out && B > 0, "orx_dlrm_inference: bad argument");
    orx_ctx* c = m->ctx;
    ORX_HIP(hipSetDevice(c->device));
    CHECK(ensure_buffers(m, B));
    Batch bt;
    CHECK(stage(m, dense, sparse, nullptr, B, flags, &bt));
    CHECK(forward(m, bt, B));
    float* pred = m->top_y.back();
    if (m->thr > 0.f && m->thr < 1.f) {
        if (m->loss_cap < 1) { ORX_HIP(hipMalloc((void**)&m->d_loss, sizeof(double))); m->loss_cap = 1; }
        // clip only (label-free): reuse the loss kernel with y = pred, no gradient output
        CHECK(orx_launch_dlrm_loss(c, pred, pred, B, 0, m->thr, nullptr, m->d_loss));
    }
    if (flags & ORX_IDS_DEVICE) {
        ORX_HIP(hipMemcpyAsync(pred_out, pred, sizeof(float) * B, hipMemcpyDeviceToDevice, c->stream));
        return ORX_OK;
    }
    ORX_HIP(hipMemcpyAsync(pred_out, pred, sizeof(float) * B, hipMemcpyDeviceToHost, c->stream));
    return orx_check_index_error(c);
}

// ------------------------------------------------- hybrid-parallel building blocks ---
static void dense_params(orx_dlrm* m, std::vector<orx_table*>& out) {
    for (auto& D : m->bot) { out.push_back(D.W); out.push_back(D.b); }
    for (auto& D : m->top) { out.push_back(D.W); out.push_back(D.b); }
}

extern "C" int orx_dlrm_grads(orx_dlrm* m, const float* dense, const float* emb_rows, const float* label, int64_t B,
                              int64_t global_B, float* emb_grads, double* loss_accum) {
    ORX_ARG(m && dense && emb_rows && label && emb_grads && loss_accum, "orx_dlrm_grads: NULL argument");
    ORX_ARG(B > 0 && global_B >= B, "orx_dlrm_grads: need 0 < B <= global_B");
    ORX_ARG(!m->grads_pending, "orx_dlrm_grads: the dense gradients of the previous call were not applied (orx_dlrm_dense_apply)");
    orx_ctx* c = m->ctx;
    ORX_HIP(hipSetDevice(c->device));
    CHECK(ensure_buffers(m, B));
    Batch bt; bt.dense = dense; bt.sparse = nullptr; bt.label = label;
    CHECK(forward(m, bt, B, emb_rows));
    const float gscale = loss_scale(m, global_B);
    CHECK(orx_launch_dlrm_loss(c, m->top_y.back(), label, B, (m->flags & ORX_DLRM_LOSS_BCE) ? 1 : 0, m->thr, m->gA, loss_accum,
                               global_B, 1, gscale));
    CHECK(backward(m, bt, B, gscale));
    m->grads_pending = true;
    const int d = m->m_spa;
    return orx_launch_copy2d(c, emb_grads, (int64_t)m->n_emb * d, m->dZ, (int64_t)m->F * d, (int)B, m->n_emb * d);
}

// can the hybrid-parallel step read the exchanged rows in place (MFMA interaction kernels for this model's shapes)?
extern "C" int orx_dlrm_direct_ok(orx_dlrm* m) {
    return m && orx_interact_direct_ok(m->F, m->m_spa, (m->flags & ORX_DLRM_REFERENCE_COMPAT) ? 1 : 0) ? 1 : 0;
}

// orx_dlrm_grads with the embedding rows read IN PLACE: rows [n_rows][m_spa] is the buffer the exchange filled, idx[b (n_emb + 1) + f]
// the row of lookup f of sample b in it (slot n_emb of every sample is not read), and the gradient of that lookup is written to
// row idx[...] of grads_dst -- the buffer that travels back.  Four passes over B n_emb m_spa floats fewer than the copying form.
extern "C" int orx_dlrm_grads_indirect(orx_dlrm* m, const float* dense, const float* rows, int64_t n_rows, const int32_t* idx,
                                       const float* label, int64_t B, int64_t global_B, float* grads_dst, double* loss_accum) {
    ORX_ARG(m && dense && rows && idx && label && grads_dst && loss_accum && n_rows > 0, "orx_dlrm_grads_indirect: NULL argument");
    ORX_ARG(B > 0 && global_B >= B, "orx_dlrm_grads_indirect: need 0 < B <= global_B");
    ORX_ARG(!m->grads_pending, "orx_dlrm_grads_indirect: the dense gradients of the previous call were not applied (orx_dlrm_dense_apply)");
    ORX_ARG(orx_dlrm_direct_ok(m), "orx_dlrm_grads_indirect: this model's shapes need the copying form (orx_dlrm_grads)");
    orx_ctx* c = m->ctx;
    ORX_HIP(hipSetDevice(c->device));
    CHECK(ensure_buffers(m, B));
    Batch bt; bt.dense = dense; bt.sparse = nullptr; bt.label = label;
    m->ext_rows = rows; m->ext_n = n_rows; m->ext_gdst = grads_dst;
    int rc = forward(m, bt, B, nullptr, idx);
    if (rc == ORX_OK) {
        const float gscale = loss_scale(m, global_B);
        rc = orx_launch_dlrm_loss(c, m->top_y.back(), label, B, (m->flags & ORX_DLRM_LOSS_BCE) ? 1 : 0, m->thr, m->gA, loss_accum, global_B, 1, gscale);
        if (rc == ORX_OK) rc = backward(m, bt, B, gscale);
    }
    m->ext_rows = nullptr; m->ext_n = 0; m->ext_gdst = nullptr;
    if (rc == ORX_OK) m->grads_pending = true;
    return rc;
}

extern "C" int orx_dlrm_dense_count(orx_dlrm* m, int64_t* count) {
    ORX_ARG(m && count, "orx_dlrm_dense_count: NULL argument");
    std::vector<orx_table*> ps; dense_params(m, ps);
    int64_t n = 0;
    for (orx_table* t : ps) n += t->rows * t->dim;
    *count = n;
    return ORX_OK;
}

// every dense parameter's gradient (gsum) <-> one flat vector (the all-reduce buffer of the hybrid-parallel step), ONE launch:
// a hipMemcpyAsync per tensor is ~16 runtime calls of 5-10 us each, twice per step
struct FlatSeg { float* p; int64_t off; int64_t n; };
__global__ __launch_bounds__(256) void dlrm_flat_kernel(const FlatSeg* seg, float* flat, int to_flat) {
    const FlatSeg sg = seg[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sg.n; i += (int64_t)gridDim.x * 256) {
        if (to_flat) flat[sg.off + i] = sg.p[i];
        else sg.p[i] = flat[sg.off + i];
    }
}

static int dlrm_flat(orx_dlrm* m, float* flat, int to_flat) {
    orx_ctx* c = m->ctx;
    std::vector<orx_table*> ps; dense_params(m, ps);
    std::vector<FlatSeg> h;
    int64_t off = 0, max_n = 0;
    for (orx_table* t : ps) {
        CHECK(orx_table_scratch(t));
        const int64_t n = t->rows * t->dim;
        h.push_back(FlatSeg{t->gsum, off, n});
        off += n; max_n = std::max(max_n, n);
    }
    if (!m->d_flatseg || m->flatseg_key != (const void*)ps[0]->gsum || m->flatseg_n != (int)h.size()) {
        if (!m->d_flatseg) ORX_HIP(hipMalloc((void**)&m->d_flatseg, 64 * sizeof(FlatSeg)));
        ORX_ARG(h.size() <= 64, "dlrm: too many dense parameters (%zu)", h.size());
        ORX_HIP(hipMemcpyAsync(m->d_flatseg, h.data(), h.size() * sizeof(FlatSeg), hipMemcpyHostToDevice, c->stream));
        ORX_HIP(hipStreamSynchronize(c->stream));          // (h is a local: the copy must have read it)
        m->flatseg_key = (const void*)ps[0]->gsum; m->flatseg_n = (int)h.size();
    }
    const int gx = (int)std::min<int64_t>(256, (max_n + 1023) / 1024 + 1);
    ORX_LAUNCH(c, dlrm_flat_kernel, dim3((unsigned)gx, (unsigned)h.size()), dim3(256), 0, (const FlatSeg*)m->d_flatseg, flat, to_flat);
    ORX_HIP(hipGetLastError());
    return ORX_OK;
}

extern "C" int orx_dlrm_dense_pack(orx_dlrm* m, float* flat) {
    ORX_ARG(m && flat, "orx_dlrm_dense_pack: NULL argument");
    ORX_HIP(hipSetDevice(m->ctx->device));
    std::vector<orx_table*> ps; dense_params(m, ps);
    for (orx_table* t : ps) ORX_ARG(t->gsum, "orx_dlrm_dense_pack: no gradients yet (call orx_dlrm_grads first)");
    return dlrm_flat(m, flat, 1);
}

// shapes of the model for the library's hybrid-parallel engine (sharded_engine.hip)
int orx_dlrm_geometry(orx_dlrm* m, int* m_spa, int* n_emb, int* dense_dim, const int64_t** d_offset, const int64_t** d_rows) {
    *m_spa = m->m_spa; *n_emb = m->n_emb; *dense_dim = m->dense_dim; *d_offset = m->d_offset; *d_rows = m->d_rows;
    return ORX_OK;
}
orx_ctx* orx_dlrm_ctx(orx_dlrm* m) { return m->ctx; }

extern "C" int orx_dlrm_dense_apply(orx_dlrm* m, orx_opt* opt, const float* flat) {
    ORX_ARG(m && opt && flat, "orx_dlrm_dense_apply: NULL argument");
    orx_ctx* c = m->ctx;
    ORX_HIP(hipSetDevice(c->device));
    std::vector<orx_table*> ps; dense_params(m, ps);
    opt->t += 1;
    float lr_t = 0.f;
    if (opt->kind == ORX_ADAM) {
        const double b1 = opt->p0, b2 = opt->p1;
        lr_t = (float)(opt->lr * std::sqrt(1.0 - std::pow(b2, (double)opt->t)) / (1.0 - std::pow(b1, (double)opt->t)));
    }
    CHECK(dlrm_flat(m, const_cast<float*>(flat), 0));
    CHECK(dense_apply_all(m, opt, lr_t));
    m->grads_pending = false;
    return ORX_OK;
}

// ------------------------------------------------ the DLRM modules on plain arrays ---
// multi_layer_perceptron.py:5-18 called on an array (not inside the composition dlrm.py:76-100, which is orx_dlrm_step /
// orx_dlrm_inference): y = act_L(... act_1(x W_1 + b_1) ...) through the same exact-fp32 MFMA product kernel the DLRM step uses.
// kernels[l]: table [in_l, out_l] (the Keras Dense layout); biases[l]: table [1, out_l] or NULL; acts[l]: 0 none, 1 relu, 2 sigmoid.
extern "C" int orx_mlp_forward(orx_ctx* c, int32_t n_layers, orx_table* const* kernels, orx_table* const* biases, const int32_t* acts,
                               const float* x, int64_t B, int32_t in_dim, int flags, float* y_out) {
    ORX_ARG(c && kernels && acts && x && y_out && n_layers > 0 && B > 0 && in_dim > 0, "orx_mlp_forward: bad argument");
    ORX_ARG(B < (1LL << 31), "orx_mlp_forward: batch too large");
    ORX_HIP(hipSetDevice(c->device));
    int64_t widest = in_dim, d = in_dim;
    for (int l = 0; l < n_layers; ++l) {
        ORX_ARG(kernels[l] && kernels[l]->ctx == c && kernels[l]->rows == d, "orx_mlp_forward: layer %d expects a [%lld, out] kernel", l, (long long)d);
        ORX_ARG(!biases || !biases[l] || (biases[l]->rows == 1 && biases[l]->dim == kernels[l]->dim), "orx_mlp_forward: layer %d: bias must be [1, %d]", l, kernels[l]->dim);
        ORX_ARG(acts[l] >= 0 && acts[l] <= 2, "orx_mlp_forward: unknown activation %d", acts[l]);
        CHECK(orx_table_sync(kernels[l]));
        d = kernels[l]->dim;
        widest = std::max<int64_t>(widest, d);
    }
    // two ping-pong activations + (host input) the staged x
    const size_t act_elems = (size_t)B * widest;
    const bool dev = (flags & ORX_IDS_DEVICE) != 0;
    if (orx_ensure((void**)&c->d_tmp, &c->d_tmp_cap, (2 * act_elems + (dev ? 0 : (size_t)B * in_dim)) * sizeof(float)) != ORX_OK) return ORX_ERR_OOM;
    float* buf[2] = {c->d_tmp, c->d_tmp + act_elems};
    const float* cur = x; int64_t ld = in_dim;
    if (!dev) {
        float* xs = c->d_tmp + 2 * act_elems;
        ORX_HIP(hipMemcpyAsync(xs, x, (size_t)B * in_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
        cur = xs;
    }
    d = in_dim;
    for (int l = 0; l < n_layers; ++l) {
        const int out = kernels[l]->dim;
        const bool last = l + 1 == n_layers;
        float* y = (last && dev) ? y_out : buf[l & 1];
        CHECK(orx_launch_gemm(c, cur, ld, 1, kernels[l]->w, out, 1, y, out, (biases && biases[l]) ? biases[l]->w : nullptr, (int)B, out, (int)d, acts[l]));
        cur = y; ld = out; d = out;
    }
    if (!dev) {
        ORX_HIP(hipMemcpyAsync(y_out, cur, (size_t)B * d * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        ORX_HIP(hipStreamSynchronize(c->stream));
    }
    return ORX_OK;
}

// second_order_feature_interaction.py:12-34 on plain arrays: z [B, F, d] (the F inputs stacked, :19), out [B, P] with
// P = F (F - 1) / 2, or F (F + 1) / 2 with self_interaction -- the selected elements of Z Z^T in boolean_mask (row-major) order.
// reference_compat != 0: the reference's text as written (lower triangle kept, strictly upper triangle selected: zeros).
extern "C" int orx_interact_forward(orx_ctx* c, const float* z, int64_t B, int32_t F, int32_t d, int self_interaction, int reference_compat,
                                    int flags, float* out) {
    ORX_ARG(c && z && out && B > 0 && F > 0 && d > 0, "orx_interact_forward: bad argument");
    ORX_HIP(hipSetDevice(c->device));
    const int P = self_interaction ? F * (F + 1) / 2 : F * (F - 1) / 2;
    const int ldR = ((d + P + 3) / 4) * 4;
    const bool dev = (flags & ORX_IDS_DEVICE) != 0;
    const size_t zel = (size_t)B * F * d, rel = (size_t)B * ldR;
    if (orx_ensure((void**)&c->d_tmp, &c->d_tmp_cap, (rel + (dev ? 0 : zel)) * sizeof(float)) != ORX_OK) return ORX_ERR_OOM;
    float* R = c->d_tmp;
    const float* Z = z;
    if (!dev) {
        float* zs = c->d_tmp + rel;
        ORX_HIP(hipMemcpyAsync(zs, z, zel * sizeof(float), hipMemcpyHostToDevice, c->stream));
        Z = zs;
    }
    CHECK(orx_launch_interact(c, true, Z, nullptr, F, d, reference_compat ? 1 : 0, self_interaction ? 1 : 0, R, P, B, ldR));
    if (P > 0)
        ORX_HIP(hipMemcpy2DAsync(out, (size_t)P * sizeof(float), R + d, (size_t)ldR * sizeof(float), (size_t)P * sizeof(float), (size_t)B,
                                 dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (!dev) ORX_HIP(hipStreamSynchronize(c->stream));
    return ORX_OK;
}
