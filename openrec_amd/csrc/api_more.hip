// C-ABI entry points of the sharded building blocks (and, for now, the
// not-yet-built pointwise step).
#include <cstring>

#include "orx_internal.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int _rc = (call);                                                              \
        if (_rc != ORX_OK) return _rc;                                                 \
    } while (0)
#define ENSURE(ptr, cap, bytes)                                                        \
    do {                                                                               \
        int _rc = orx_ensure((void**)&(ptr), &(cap), (bytes));                         \
        if (_rc != ORX_OK) return _rc;                                                 \
    } while (0)

extern "C" int orx_pointwise_step(orx_ctx*, int, orx_opt*, orx_table*, orx_table*, orx_table*, orx_table*,
                                  const int32_t*, const int32_t*, const float*, int64_t, int64_t, int64_t,
                                  float, float, int, float*, float*) {
    orx_set_error("orx_pointwise_step: not implemented yet");
    return ORX_ERR_STATE;
}

extern "C" int orx_gather_rows(orx_ctx* ctx, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                               float* out, int64_t out_stride) {
    ORX_ARG(ctx && t && (n == 0 || (ids && out)), "orx_gather_rows: NULL argument");
    ORX_ARG(out_stride >= t->dim + (bias ? 1 : 0), "orx_gather_rows: out_stride %lld too small", (long long)out_stride);
    ORX_HIP(hipSetDevice(ctx->device));
    return orx_launch_gather(ctx, t->w, bias ? bias->w : nullptr, t->rows, t->dim, ids, n, out, out_stride, ctx->d_err, 1);
}

extern "C" int orx_pair_grads(orx_ctx* ctx, int model, int32_t D,
                              const float* u_rows, const float* p_rows, const float* n_rows, int64_t row_stride,
                              const int32_t* valid, int64_t T, int64_t B_global, float margin, int flags,
                              float* gu, float* gp, float* gn, int64_t g_stride, double* loss_l2_accum) {
    ORX_ARG(ctx && u_rows && p_rows && n_rows && gu && gp && gn, "orx_pair_grads: NULL argument");
    ORX_ARG(model == ORX_BPR || model == ORX_UCML, "orx_pair_grads: unknown model %d", model);
    ORX_ARG(D > 0 && row_stride > D && g_stride > D, "orx_pair_grads: strides must leave room for the bias column");
    ORX_ARG(B_global > 0, "orx_pair_grads: B_global must be positive");
    if (T == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    GradArgs a;
    memset(&a, 0, sizeof(a));
    a.u = u_rows; a.p = p_rows; a.n = n_rows; a.row_stride = row_stride; a.valid = valid;
    a.gu = gu; a.gp = gp; a.gn = gn; a.g_stride = g_stride;
    a.T = T; a.D = D; a.invB = 1.0f / (float)B_global; a.margin = margin;
    a.l2w = (flags & ORX_NO_L2) ? 0.f : 1.f;
    // upper bound of the wave count (one lane group per triplet, 4 waves per block)
    ENSURE(ctx->d_partial, ctx->d_partial_cap, (size_t)(T + 4) * 2 * sizeof(float));
    a.partial = ctx->d_partial;
    int nw = 0;
    CHECK(orx_launch_pair_grads(ctx, model, a, &nw));
    if (loss_l2_accum) CHECK(orx_launch_loss_accumulate(ctx, ctx->d_partial, nw, loss_l2_accum));
    return ORX_OK;
}

extern "C" int orx_apply_rows(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias,
                              const int32_t* ids, int64_t n, const float* grads, int64_t g_stride) {
    ORX_ARG(ctx && opt && t && (n == 0 || (ids && grads)), "orx_apply_rows: NULL argument");
    ORX_ARG(opt->kind == ORX_SGD || opt->kind == ORX_ADAGRAD, "orx_apply_rows: only SGD and Adagrad are supported on sharded tables");
    ORX_ARG(g_stride >= t->dim + (bias ? 1 : 0), "orx_apply_rows: g_stride too small");
    ORX_ARG(!bias || (bias->dim == 1 && bias->rows == t->rows), "orx_apply_rows: bias must be [%lld, 1]", (long long)t->rows);
    if (n == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    RowsArgs a;
    memset(&a, 0, sizeof(a));
    a.W = t->w; a.bias = bias ? bias->w : nullptr;
    a.ids = ids; a.grads = grads; a.g_stride = g_stride; a.n = n; a.rows = t->rows; a.D = t->dim;
    a.lr = opt->lr; a.err = ctx->d_err;
    if (opt->kind == ORX_SGD) return orx_launch_apply_rows(ctx, ORX_SGD, false, a);

    // Adagrad: dedup-sum semantics
    CHECK(orx_table_scratch(t));
    if (bias) CHECK(orx_table_scratch(bias));
    OptSlots st, sb;
    CHECK(orx_opt_slots(opt, t, &st));
    if (bias) CHECK(orx_opt_slots(opt, bias, &sb));
    ENSURE(ctx->d_dflag, ctx->d_dflag_cap, (size_t)n);
    ENSURE(ctx->d_dlist, ctx->d_dlist_cap, (size_t)(n / 2 + 1) * sizeof(uint32_t));
    ENSURE(ctx->d_dcount, ctx->d_dcount_cap, sizeof(int));
    ORX_HIP(hipMemsetAsync(ctx->d_dcount, 0, sizeof(int), ctx->stream));
    DedupArgs d;
    memset(&d, 0, sizeof(d));
    // scan the id list in the ITEM role so that dup_apply also finishes the bias
    d.uid = ids; d.pid = ids; d.nid = ids; d.id_stride = n;
    d.dflag = ctx->d_dflag; d.dlist = ctx->d_dlist; d.dcount = ctx->d_dcount;
    d.flag_stride = n; d.list_stride = n / 2 + 1;
    d.nU = 0; d.nP = n; d.nN = 0; d.NU = 0; d.NI = t->rows; d.nbu = 0; d.nbi = orx_dedup_buckets(t->rows);
    CHECK(orx_launch_dedup(ctx, d, 1));
    a.G = t->gsum; a.gb = bias ? bias->gsum : nullptr;
    a.A = st.s0; a.ab = bias ? sb.s0 : nullptr;
    a.dflag = ctx->d_dflag; a.eps = opt->p1;
    CHECK(orx_launch_apply_rows(ctx, ORX_ADAGRAD, true, a));
    PairArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.V = t->w; pa.gV = t->gsum; pa.aV = st.s0;
    pa.b = bias ? bias->w : nullptr; pa.gb = bias ? bias->gsum : nullptr; pa.ab = bias ? sb.s0 : nullptr;
    pa.dlist = ctx->d_dlist; pa.dcount = ctx->d_dcount;
    pa.B = n; pa.D = t->dim; pa.lr = opt->lr; pa.eps = opt->p1;
    return orx_launch_dup_apply(ctx, ORX_ADAGRAD, pa);
}
