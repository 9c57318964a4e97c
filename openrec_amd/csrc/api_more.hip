// C-ABI entry points of the sharded building blocks (and, for now, the
// not-yet-built pointwise step).
#include <vector>
#include <cmath>
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

// ------------------------------------------------------------ pointwise step ---
static int check_point_tables(int model, orx_table* U, orx_table* V, orx_table* b, orx_table* w) {
    ORX_ARG(model == ORX_GMF || model == ORX_WRMF, "pointwise: unknown model %d", model);
    ORX_ARG(U && V && b, "pointwise: NULL table");
    ORX_ARG(U->dim == V->dim, "pointwise: user dim %d != item dim %d", U->dim, V->dim);
    ORX_ARG(b->dim == 1 && b->rows == V->rows, "pointwise: item_bias must be [%lld, 1]", (long long)V->rows);
    ORX_ARG(model != ORX_GMF || (w && w->rows == U->dim && w->dim == 1), "pointwise: GMF needs the Dense kernel w as a [D, 1] table");
    return ORX_OK;
}

// stage (uid, iid, label) of K steps: ids into d_ids [2][K*B], labels into d_lab [K*B]
static int stage_point(orx_ctx* c, const int32_t* uid, const int32_t* iid, const float* label,
                       int64_t K, int64_t B, int64_t id_stride, int flags,
                       const int32_t** du, const int32_t** di, const float** dl, int64_t* ds) {
    if (flags & ORX_IDS_DEVICE) { *du = uid; *di = iid; *dl = label; *ds = id_stride; return ORX_OK; }
    const int64_t n = K * B;
    ENSURE(c->d_ids, c->d_ids_cap, (size_t)2 * n * sizeof(int32_t));
    ENSURE(c->d_lab, c->d_lab_cap, (size_t)n * sizeof(float));
    for (int64_t s = 0; s < K; ++s) {
        CHECK(stage_ids(c, uid + s * id_stride, B, s * B));
        CHECK(stage_ids(c, iid + s * id_stride, B, n + s * B));
        ORX_HIP(hipMemcpyAsync(c->d_lab + s * B, label + s * id_stride, (size_t)B * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    *du = c->d_ids; *di = c->d_ids + n; *dl = c->d_lab; *ds = B;
    return ORX_OK;
}

extern "C" int orx_pointwise_step(orx_ctx* c, int model, orx_opt* opt,
                                  orx_table* U, orx_table* V, orx_table* b, orx_table* w,
                                  const int32_t* uid, const int32_t* iid, const float* label,
                                  int64_t K, int64_t B, int64_t id_stride, float a_w, float b_w, int flags,
                                  float* loss_out, float* l2_out) {
    ORX_ARG(c && opt, "orx_pointwise_step: NULL context/optimizer");
    CHECK(check_point_tables(model, U, V, b, w));
    ORX_ARG(K >= 0 && B > 0, "orx_pointwise_step: K must be >= 0 and B > 0");
    ORX_ARG(K == 0 || (uid && iid && label), "orx_pointwise_step: NULL id/label pointer");
    if (K == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(c->device));
    const int32_t *du, *di; const float* dl; int64_t ds;
    CHECK(stage_point(c, uid, iid, label, K, B, id_stride, flags, &du, &di, &dl, &ds));
    const bool hogwild = (flags & ORX_HOGWILD) != 0;
    const char* fb_env = getenv("ORX_FORCE_FALLBACK");
    const int fb = fb_env ? atoi(fb_env) : 0;
    // TF-2.0 Adam applied lazily, as in orx_pairwise_step: float4 dims with the exact-step plan; the
    // Dense(1) kernel of GMF is a dense parameter and takes the plain rule.  Otherwise every reference accumulates
    // and the tables are swept whole.
    const bool lazy_adam = opt->kind == ORX_ADAM && !hogwild && orx_fused_can_inline_apply(U->dim) && U->rows < (1LL << 28) &&
                           V->rows < (1LL << 28) && !(fb & 1) && U->owned && V->owned && b->owned && getenv("ORX_ADAM_DENSE") == nullptr;
    const int mode = (opt->kind == ORX_ADAM && !lazy_adam) ? MODE_ACCUM : (hogwild ? MODE_HOGWILD : MODE_EXACT);
    const bool lazy_resume = lazy_adam && U->lazy == opt && V->lazy == opt && b->lazy == opt;
    if (!lazy_resume) for (orx_table* t : {U, V, b}) CHECK(orx_table_sync(t));
    { orx_table* mine[3] = {U, V, b}; CHECK(orx_opt_isolate(opt, mine, 3)); }      // (a shared optimizer: api.hip orx_opt_isolate)
    if (w) CHECK(orx_table_sync(w));
    CHECK(orx_table_scratch(U)); CHECK(orx_table_scratch(V)); CHECK(orx_table_scratch(b));
    if (w) CHECK(orx_table_scratch(w));
    OptSlots sU, sV, sb, sw;
    CHECK(orx_opt_slots(opt, U, &sU)); CHECK(orx_opt_slots(opt, V, &sV)); CHECK(orx_opt_slots(opt, b, &sb));
    if (w) CHECK(orx_opt_slots(opt, w, &sw));
    const int D = U->dim;
    const int nw = orx_point_nwaves(D, B);
    const int nslot = nw + (model == ORX_GMF ? 1 : 0);      // + one slot for 0.5*||w||^2
    ENSURE(c->d_partial, c->d_partial_cap, (size_t)K * nslot * 2 * sizeof(float));
    ENSURE(c->d_loss, c->d_loss_cap, (size_t)K * 2 * sizeof(double));
    if (model == ORX_GMF) ENSURE(c->d_wpart, c->d_wpart_cap, (size_t)(nw + 256) * D * sizeof(float));   // + stage-1 rows of dense_reduce
    const int64_t list_stride = 2 * B;
    // exact mode on the float4 dims below 2^28 rows: the plan of the pairwise step (roles: rows referenced twice get
    // plain stores; staging plan + reduction tree for hot rows), made once for all K steps.  Otherwise byte flags +
    // atomics for every duplicate.
    const bool role_bits = mode == MODE_EXACT && orx_fused_can_inline_apply(D) && U->rows < (1LL << 28) && V->rows < (1LL << 28) && !(fb & 1);
    const bool staging = role_bits && !(fb & 8);
    // the previous step's duplicated rows are applied by extra blocks of the next step's launch, as in the pairwise step
    // (WRMF 27.1 -> see DESIGN 4.4; fb bit 2: separate dup_apply launches)
    const bool inline_apply = role_bits && K > 1 && orx_plan_v2(true) && !(fb & 2);
    PairPlan plan;
    memset(&plan, 0, sizeof(plan));
    if (role_bits) {
        CHECK(orx_table_scratch(U, true)); CHECK(orx_table_scratch(V, true)); CHECK(orx_table_scratch(b, true));
        const int nb_total = orx_dedup_buckets(U->rows) + orx_dedup_buckets(V->rows);
        const size_t part_cap = c->d_partial_cap;       // orx_exact_buffers sizes d_partial for its own use: keep ours
        CHECK(orx_exact_buffers(c, U, V, K, B, mode, true, inline_apply, staging, nb_total, nslot, &plan));
        if (opt->kind == ORX_ADAM) plan.min_late = 1;      // Adam: fixed summation order for every row referenced >= 3 times (api.hip)
        // pairing (round 6): as in orx_pairwise_step -- SGD, float4 dims with >= 2 samples per wavefront, paused for 32 calls when a plan
        // accepts few pairs.  ORX_NO_PAIR=1 / ORX_POINT_NO_PAIR=1 / ORX_FORCE_FALLBACK bit 4: off
        if (orx_pairing_wanted(mode, true, opt->kind, D, B, fb) && getenv("ORX_POINT_NO_PAIR") == nullptr) {
            if (c->pair_pause > 0) c->pair_pause -= 1;
            else CHECK(orx_pairing_buffers(c, B, D, &plan));
        }
        (void)part_cap;
        ENSURE(c->d_partial, c->d_partial_cap, (size_t)K * nslot * 2 * sizeof(float));
    } else if (mode == MODE_EXACT) {
        ENSURE(c->d_dflag, c->d_dflag_cap, (size_t)K * 2 * B);
        ENSURE(c->d_dlist, c->d_dlist_cap, (size_t)K * list_stride * sizeof(uint32_t));
        ENSURE(c->d_dcount, c->d_dcount_cap, (size_t)K * sizeof(int));
        DedupArgs d;
        memset(&d, 0, sizeof(d));
        d.uid = du; d.pid = di; d.nid = di; d.id_stride = ds;
        d.dflag = c->d_dflag; d.dlist = c->d_dlist; d.dcount = c->d_dcount;
        d.flag_stride = 2 * B; d.list_stride = list_stride;
        d.nU = B; d.nP = B; d.nN = 0; d.NU = U->rows; d.NI = V->rows;
        d.nbu = orx_dedup_buckets(U->rows); d.nbi = orx_dedup_buckets(V->rows);
        ORX_HIP(hipMemsetAsync(c->d_dcount, 0, (size_t)K * sizeof(int), c->stream));
        CHECK(orx_launch_dedup(c, d, K));
    }
    PointArgs a;
    memset(&a, 0, sizeof(a));
    a.U = U->w; a.V = V->w; a.b = b->w; a.w = w ? w->w : nullptr;
    a.gU = U->gsum; a.gV = V->gsum; a.gb = b->gsum;
    a.aU = sU.s0; a.aV = sV.s0; a.ab = sb.s0;
    a.B = B; a.NU = U->rows; a.NI = V->rows; a.D = D;
    a.lr = opt->lr; a.eps = opt->kind == ORX_ADAGRAD ? opt->p1 : 0.f;
    a.invB = 1.0f / (float)B; a.l2w = (flags & ORX_NO_L2) ? 0.f : 1.f; a.a_w = a_w; a.b_w = b_w;
    a.sigmoid = (model == ORX_WRMF && (flags & ORX_POINT_SIGMOID)) ? 1 : 0;
    a.wpartial = c->d_wpart; a.err = c->d_err;
    // GMF on the float4 kernels: the Dense(1) gradient is reduced and applied by the last workgroups of the step's own launch
    const bool dense_tail = model == ORX_GMF && orx_point_dense_tail_ok(D);
    if (dense_tail) {
        // the partial rows and the group rows behind them start out EMPTY (all ones); every launch hands them back that way
        const size_t nrows = (size_t)orx_point_wparts(D, B) + (size_t)orx_point_reducers(D, B) - 1;
        ORX_HIP(hipMemsetAsync(c->d_wpart, 0xff, nrows * D * sizeof(float), c->stream));
        a.wt_nred = orx_point_reducers(D, B); a.wt_rows = c->d_wpart + (size_t)orx_point_wparts(D, B) * D;
        const bool to_gout = mode == MODE_ACCUM || lazy_adam;                        // (Adam: adam_sweep_kernel applies the gradient)
        a.wt_gout = to_gout ? w->gsum : nullptr; a.wt_acc = to_gout ? nullptr : sw.s0; a.wt_optkind = to_gout ? -1 : opt->kind;
    }
    if (role_bits) { a.role_bits = 1; a.gU2 = U->gsum2; a.gV2 = V->gsum2; a.gb2 = b->gsum2; a.readyU = U->ready; a.readyV = V->ready; }
    PairArgs pa;                                 // view of the same tables for dup_apply_kernel / hot_reduce_kernel / the in-launch apply
    memset(&pa, 0, sizeof(pa));
    pa.U = U->w; pa.V = V->w; pa.b = b->w; pa.gU = U->gsum; pa.gV = V->gsum; pa.gb = b->gsum;
    if (role_bits) { pa.gU2 = U->gsum2; pa.gV2 = V->gsum2; pa.gb2 = b->gsum2; pa.role_bits = 1; pa.readyU = U->ready; pa.readyV = V->ready; }
    // epochs tag the ready flags: one per step; on wrap-around the tables clear theirs (as orx_pairwise_step does)
    if ((int64_t)c->epoch + K + 16 > 0x7fffffff) { c->epoch = 0; c->epoch_gen += 1; }
    for (orx_table* t : {U, V}) {
        if (t->tag_gen != c->epoch_gen) {
            if (t->ready) ORX_HIP(hipMemsetAsync(t->ready, 0, (size_t)t->rows * sizeof(int), c->stream));
            if (t->side) ORX_HIP(hipMemsetAsync(t->side, 0, (size_t)t->rows * 2 * sizeof(int), c->stream));
            t->tag_gen = c->epoch_gen;
        }
    }
    pa.aU = sU.s0; pa.aV = sV.s0; pa.ab = sb.s0; pa.B = B; pa.D = D; pa.lr = a.lr; pa.eps = a.eps;
    if (lazy_adam) {
        a.a2U = pa.a2U = sU.s1; a.a2V = pa.a2V = sV.s1; a.a2b = pa.a2b = sb.s1;
        CHECK(orx_opt_last(opt, U, !lazy_resume, &a.lastU)); CHECK(orx_opt_last(opt, V, !lazy_resume, &a.lastV));
        CHECK(orx_opt_last(opt, b, !lazy_resume, &a.lastb));
        pa.lastU = a.lastU; pa.lastV = a.lastV; pa.lastb = a.lastb;
        CHECK(orx_adam_lrt(opt, opt->t + K));
        a.lrt = pa.lrt = opt->d_lrt; a.b1 = pa.b1 = opt->p0; a.b2 = pa.b2 = opt->p1; a.eps = pa.eps = opt->p2;
        if (orx_adam_cf_ok(opt) && opt->d_lrv != nullptr) {
            a.lrv = reinterpret_cast<const float4*>(opt->d_lrv);
            a.cf_delta = (float)(-0.5 * std::log((double)opt->p1)); a.cf_lb1 = (float)std::log2((double)opt->p0); a.cf_lb2 = (float)std::log2((double)opt->p1);
        }
        a.newton = pa.newton = (1.0f - sqrtf(opt->p1)) <= 1e-3f && getenv("ORX_ADAM_NO_NEWTON") == nullptr;
        U->lazy = opt; V->lazy = opt; b->lazy = opt;
    }
    const int64_t chunk = role_bits ? plan.chunk : K;
    for (int64_t s0 = 0; s0 < K; s0 += chunk) {
    const int64_t kc = (K - s0 < chunk) ? (K - s0) : chunk;
    ExactChunk ck;
    if (role_bits) {
        c->plan_label = plan.pair_tpw > 1 ? dl + s0 * ds : nullptr;
        const int rc_plan = orx_exact_plan_chunk(c, U, V, du + s0 * ds, di + s0 * ds, di + s0 * ds, ds, B, B, 0, kc, B, true, inline_apply, staging, plan, &ck);
        c->plan_label = nullptr;
        CHECK(rc_plan);
    }
    const bool inl = inline_apply && !ck.hot && !ck.dense_dups;      // (many duplicated rows / reduction-tree levels: separate launches, api.hip)
    for (int64_t i = 0; i < kc; ++i) {
        const int64_t s = s0 + i;
        a.label = dl + s * ds;
        a.partial = c->d_partial + (size_t)s * nslot * 2;
        a.wt_l2slot = a.partial + 2 * nw;
        if (role_bits) {
            a.uid = c->d_ids2 + (size_t)i * 3 * plan.Bp; a.iid = a.uid + plan.Bp;
            a.ids4 = plan.pair_tpw > 1 ? c->d_ids4 + (size_t)i * B : nullptr;      // (with pairing the SoA copy is not written: the records are the input)
            orx_exact_step_views(c, plan, i, B, D, ck.use_stage, &pa);
            a.refinfo = pa.refinfo; a.segstart = pa.segstart; a.stage = pa.stage; a.stageb = pa.stageb;
        } else {
            a.uid = du + s * ds; a.iid = di + s * ds;
            a.dflag = c->d_dflag + (size_t)s * 2 * B;
            pa.dlist = c->d_dlist + (size_t)s * list_stride; pa.dcount = c->d_dcount + s;
        }
        if (lazy_adam) { opt->t += 1; a.step_t = pa.step_t = (int)opt->t; }
        a.n_apply_blocks = 0;
        if (role_bits) {
            a.epoch = pa.epoch = ++c->epoch;
            if (inl && i > 0) {                  // this launch also applies the duplicated rows of step i-1
                pa.n_apply_blocks = (int)std::min<int64_t>(2048, std::max<int64_t>(16, (B / 4) / (1024 / D) + 1));
                pa.prev_dlist = c->d_dlist + (size_t)(i - 1) * plan.list_stride; pa.prev_dcount = c->d_dcount + (i - 1);
                a.n_apply_blocks = pa.n_apply_blocks;
                a.ap = pa;
            }
        }
        CHECK(orx_launch_point_fused(c, model, opt->kind, mode, a));
        if (mode == MODE_EXACT) {
            for (int l = 0; l < ck.tree_levels; ++l) CHECK(orx_launch_hot_reduce(c, pa, l));
            if (!inl || i == kc - 1) CHECK(orx_launch_dup_apply(c, opt->kind, pa));
        }
        float lr_t = lazy_adam ? opt->h_lrt[(size_t)opt->t] : 0.f;
        if (mode == MODE_ACCUM) {
            opt->t += 1;
            const double b1 = opt->p0, b2 = opt->p1;
            lr_t = (float)(opt->lr * std::sqrt(1.0 - std::pow(b2, (double)opt->t)) / (1.0 - std::pow(b1, (double)opt->t)));
            CHECK(orx_launch_adam_sweep(c, U->w, sU.s0, sU.s1, U->gsum, U->rows * D, lr_t, opt->p0, opt->p1, opt->p2));
            CHECK(orx_launch_adam_sweep(c, V->w, sV.s0, sV.s1, V->gsum, V->rows * D, lr_t, opt->p0, opt->p1, opt->p2));
            CHECK(orx_launch_adam_sweep(c, b->w, sb.s0, sb.s1, b->gsum, b->rows, lr_t, opt->p0, opt->p1, opt->p2));
        }
        if (model == ORX_GMF) {                  // dense Dense(1) kernel: reduce partials, apply the dense rule
            if (mode == MODE_ACCUM || lazy_adam) {
                if (!dense_tail) CHECK(orx_launch_dense_reduce(c, c->d_wpart, orx_point_wparts(D, B), D, w->w, a.l2w, w->gsum, a.partial + 2 * nw, nullptr, -1, 0.f, 0.f));
                CHECK(orx_launch_adam_sweep(c, w->w, sw.s0, sw.s1, w->gsum, D, lr_t, opt->p0, opt->p1, opt->p2));
            } else if (!dense_tail) {        // reduce + dense SGD / Adagrad rule in one launch
                CHECK(orx_launch_dense_reduce(c, c->d_wpart, orx_point_wparts(D, B), D, w->w, a.l2w, nullptr, a.partial + 2 * nw, sw.s0, opt->kind, opt->lr, a.eps));
            }
        }
    }
    }
    ReduceArgs r;
    r.partial = c->d_partial; r.out = c->d_loss; r.nwaves = nslot;
    CHECK(orx_launch_loss_reduce(c, r, K));
    CHECK(fetch_losses(c, K, loss_out, l2_out));
    if (!(flags & ORX_IDS_DEVICE)) return orx_check_index_error(c);
    return ORX_OK;
}

extern "C" int orx_pointwise_loss(orx_ctx* c, int model, orx_table* U, orx_table* V, orx_table* b, orx_table* w,
                                  const int32_t* uid, const int32_t* iid, const float* label,
                                  int64_t B, float a_w, float b_w, int flags, float* loss_out, float* l2_out) {
    if (U) CHECK(orx_table_sync(U));
    if (V) CHECK(orx_table_sync(V));
    if (b) CHECK(orx_table_sync(b));
    if (w) CHECK(orx_table_sync(w));
    ORX_ARG(c, "orx_pointwise_loss: NULL context");
    CHECK(check_point_tables(model, U, V, b, w));
    ORX_ARG(B > 0 && uid && iid && label, "orx_pointwise_loss: empty batch or NULL pointer");
    ORX_HIP(hipSetDevice(c->device));
    const int32_t *du, *di; const float* dl; int64_t ds;
    CHECK(stage_point(c, uid, iid, label, 1, B, B, flags, &du, &di, &dl, &ds));
    const int D = U->dim;
    const int nw = orx_point_nwaves(D, B);
    const int nslot = nw + (model == ORX_GMF ? 1 : 0);
    ENSURE(c->d_partial, c->d_partial_cap, (size_t)nslot * 2 * sizeof(float));
    ENSURE(c->d_loss, c->d_loss_cap, 2 * sizeof(double));
    PointArgs a;
    memset(&a, 0, sizeof(a));
    a.U = U->w; a.V = V->w; a.b = b->w; a.w = w ? w->w : nullptr;
    a.B = B; a.NU = U->rows; a.NI = V->rows; a.D = D;
    a.invB = 1.0f / (float)B; a.l2w = 1.f; a.a_w = a_w; a.b_w = b_w;
    a.sigmoid = (model == ORX_WRMF && (flags & ORX_POINT_SIGMOID)) ? 1 : 0;
    a.partial = c->d_partial; a.err = c->d_err;
    a.uid = du; a.iid = di; a.label = dl;
    CHECK(orx_launch_point_fused(c, model, ORX_SGD, MODE_LOSS, a));
    if (model == ORX_GMF) CHECK(orx_launch_dense_reduce(c, nullptr, 0, D, w->w, 0.f, nullptr, c->d_partial + 2 * nw, nullptr, -1, 0.f, 0.f));
    ReduceArgs r;
    r.partial = c->d_partial; r.out = c->d_loss; r.nwaves = nslot;
    CHECK(orx_launch_loss_reduce(c, r, 1));
    CHECK(fetch_losses(c, 1, loss_out, l2_out));
    return orx_check_index_error(c);
}

extern "C" int orx_score_all_items(orx_ctx* c, int kind, orx_table* U, orx_table* V, orx_table* b, orx_table* w,
                                   const int32_t* uid, int64_t n, float* out) {
    if (U) CHECK(orx_table_sync(U));
    if (V) CHECK(orx_table_sync(V));
    if (b) CHECK(orx_table_sync(b));
    if (w) CHECK(orx_table_sync(w));
    ORX_ARG(c && U && V && b && (n == 0 || (uid && out)), "orx_score_all_items: NULL argument");
    ORX_ARG(kind >= 0 && kind <= 2, "orx_score_all_items: unknown kind %d", kind);
    ORX_ARG(U->dim == V->dim && b->rows == V->rows && b->dim == 1, "orx_score_all_items: table shapes do not match");
    ORX_ARG(kind != 2 || (w && w->rows == U->dim && w->dim == 1), "orx_score_all_items: GMF needs w [D, 1]");
    ORX_ARG(U->dim <= 1024, "orx_score_all_items: dim too large for the LDS user tile");
    if (n == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(c->device));
    ENSURE(c->d_ids, c->d_ids_cap, (size_t)n * sizeof(int32_t));
    ENSURE(c->d_tmp, c->d_tmp_cap, (size_t)n * V->rows * sizeof(float));
    CHECK(stage_ids(c, uid, n, 0));
    CHECK(orx_launch_score_all(c, U->w, V->w, b->w, w ? w->w : nullptr, c->d_ids, n, U->rows, V->rows, U->dim, kind, c->d_tmp));
    ORX_HIP(hipMemcpyAsync(out, c->d_tmp, (size_t)n * V->rows * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return orx_check_index_error(c);
}

// the same scores left in device memory (out_dev [n, item_rows] fp32): what the metrics of the evaluation step read
extern "C" int orx_score_all_items_device(orx_ctx* c, int kind, orx_table* U, orx_table* V, orx_table* b, orx_table* w,
                                          const int32_t* uid, int64_t n, float* out_dev) {
    if (U) CHECK(orx_table_sync(U));
    if (V) CHECK(orx_table_sync(V));
    if (b) CHECK(orx_table_sync(b));
    if (w) CHECK(orx_table_sync(w));
    ORX_ARG(c && U && V && b && (n == 0 || (uid && out_dev)), "orx_score_all_items_device: NULL argument");
    ORX_ARG(kind >= 0 && kind <= 2, "orx_score_all_items_device: unknown kind %d", kind);
    ORX_ARG(U->dim == V->dim && b->rows == V->rows && b->dim == 1, "orx_score_all_items_device: table shapes do not match");
    ORX_ARG(kind != 2 || (w && w->rows == U->dim && w->dim == 1), "orx_score_all_items_device: GMF needs w [D, 1]");
    ORX_ARG(U->dim <= 1024, "orx_score_all_items_device: dim too large for the LDS user tile");
    if (n == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(c->device));
    ENSURE(c->d_ids, c->d_ids_cap, (size_t)n * sizeof(int32_t));
    CHECK(stage_ids(c, uid, n, 0));
    CHECK(orx_launch_score_all(c, U->w, V->w, b->w, w ? w->w : nullptr, c->d_ids, n, U->rows, V->rows, U->dim, kind, out_dev));
    return orx_check_index_error(c);
}

extern "C" int orx_gather_rows(orx_ctx* ctx, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                               float* out, int64_t out_stride) {
    if (t) CHECK(orx_table_touch(t, ids, n));            // a lazy table: only the rows read are brought up to date
    if (bias) CHECK(orx_table_touch(bias, ids, n));
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

// ---- lazy TF-2.0 Adam on gradient rows -------------------------------------------------------------
// The rule moves every row every step; a row's gradient-free steps are replayed when it is next referenced
// (gathered for a forward pass: touch; given a gradient: apply) or when the table is observed (orx_table_sync).
bool orx_adam_rows_lazy(const orx_opt* opt, const orx_table* t) {
    return opt->kind == ORX_ADAM && t->rows < (1LL << 31) && t->owned && getenv("ORX_ADAM_DENSE") == nullptr;
}

// duplicate flags + list of the duplicated rows of `ids` (item role) into the context's dedup buffers
int orx_adam_rows_dedup(orx_ctx* ctx, orx_table* t, const int32_t* ids, int64_t n, ColWindows cw) {
    ENSURE(ctx->d_dflag, ctx->d_dflag_cap, (size_t)n);
    ENSURE(ctx->d_dlist, ctx->d_dlist_cap, (size_t)(n / 2 + 1) * sizeof(uint32_t));
    ENSURE(ctx->d_dcount, ctx->d_dcount_cap, sizeof(int));
    ORX_HIP(hipMemsetAsync(ctx->d_dcount, 0, sizeof(int), ctx->stream));
    DedupArgs d;
    memset(&d, 0, sizeof(d));
    d.uid = ids; d.pid = ids; d.nid = ids; d.id_stride = n;
    d.dflag = ctx->d_dflag; d.dlist = ctx->d_dlist; d.dcount = ctx->d_dcount;
    d.flag_stride = n; d.list_stride = n / 2 + 1;
    d.nU = 0; d.nP = n; d.nN = 0; d.NU = 0; d.NI = t->rows; d.nbu = 0; d.nbi = orx_dedup_buckets(t->rows);
    if (cw.win != nullptr && cw.F > 0 && n % cw.F == 0) { d.col_F = cw.F; d.col_win = cw.win; }
    return orx_launch_dedup(ctx, d, 1);
}

// the table becomes (or stays) lazy under `opt`; every row is current at step `now` when it was not lazy before
static int adam_rows_args(orx_ctx* ctx, orx_opt* opt, orx_table* t, int64_t now, AdamRowsArgs* a) {
    if (t->lazy != nullptr && t->lazy != opt) CHECK(orx_table_sync(t));
    const bool resume = t->lazy == opt;
    OptSlots st;
    CHECK(orx_opt_slots(opt, t, &st));
    memset(a, 0, sizeof(*a));
    CHECK(orx_opt_last(opt, t, !resume, &a->last, now));
    // lr_t entries are uploaded well ahead of use (one small copy per 1024 steps instead of one per step; the kernels
    // fetch them through the scalar cache, so an entry is also in place before any kernel can have cached its line)
    if ((int64_t)opt->lrt_uploaded < opt->t + 2) CHECK(orx_adam_lrt(opt, opt->t + 1024));
    t->lazy = opt;
    a->W = t->w; a->M = st.s0; a->V = st.s1; a->rows = t->rows; a->D = t->dim;
    a->lrt = opt->d_lrt; a->b1 = opt->p0; a->b2 = opt->p1; a->eps = opt->p2;
    a->newton = (1.0f - sqrtf(opt->p1)) <= 1e-3f;
    a->cf = orx_adam_cf_params(opt);
    a->err = ctx->d_err;
    return ORX_OK;
}

int orx_adam_rows_touch(orx_ctx* ctx, orx_opt* opt, orx_table* t, const int32_t* ids, int64_t n, bool have_dedup, ColWindows cw) {
    if (n == 0 || opt->t == 0) return ORX_OK;                // nothing has moved yet
    if (t->lazy != opt) return orx_table_sync(t);           // not lazy under this optimizer: every row is (made) current
    AdamRowsArgs a;
    CHECK(adam_rows_args(ctx, opt, t, opt->t, &a));
    if (!have_dedup) CHECK(orx_adam_rows_dedup(ctx, t, ids, n, cw));
    a.dflag = ctx->d_dflag; a.dlist = (const uint32_t*)ctx->d_dlist; a.dcount = ctx->d_dcount;     // (allocated by the dedup)
    a.ids = ids; a.n = n; a.T = (int)opt->t;
    return orx_launch_adam_rows(ctx, false, a, n / 2 + 1);
}

// step opt->t (the caller has advanced the counter) with per-occurrence gradient rows
int orx_adam_rows_apply(orx_ctx* ctx, orx_opt* opt, orx_table* t, const int32_t* ids, int64_t n, const float* grads, int64_t g_stride,
                        bool have_dedup, ColWindows cw) {
    ORX_ARG(opt->t >= 1, "adam_rows_apply: the step counter has not been advanced");
    CHECK(orx_table_scratch(t));
    AdamRowsArgs a;
    CHECK(adam_rows_args(ctx, opt, t, opt->t - 1, &a));
    if (!have_dedup) CHECK(orx_adam_rows_dedup(ctx, t, ids, n, cw));
    a.dflag = ctx->d_dflag; a.dlist = (const uint32_t*)ctx->d_dlist; a.dcount = ctx->d_dcount;
    a.G = t->gsum; a.ids = ids; a.n = n; a.grads = grads; a.g_stride = g_stride;
    a.T = (int)opt->t; a.lr_T = opt->h_lrt[(size_t)opt->t];
    return orx_launch_adam_rows(ctx, true, a, n / 2 + 1);
}

// the same two operations on a SORTED id list (kernels_rowsort.hip): distinct rows are the heads of the runs, a row's
// gradient rows are summed in position order -- no duplicate flags, no gsum, no atomics.  step: the caller has advanced opt->t.
int orx_adam_rows_sorted(orx_ctx* ctx, orx_opt* opt, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride, bool step) {
    if (n == 0) return ORX_OK;
    AdamRowsArgs a;
    if (!step) {
        if (opt->t == 0) return ORX_OK;
        if (t->lazy != opt) return orx_table_sync(t);
        CHECK(adam_rows_args(ctx, opt, t, opt->t, &a));
        a.T = (int)opt->t;
        return orx_csr_adam(ctx, false, a, t, sorted, n);
    }
    ORX_ARG(opt->t >= 1, "adam_rows_sorted: the step counter has not been advanced");
    CHECK(adam_rows_args(ctx, opt, t, opt->t - 1, &a));
    a.grads = grads; a.g_stride = g_stride; a.T = (int)opt->t; a.lr_T = opt->h_lrt[(size_t)opt->t];
    return orx_csr_adam(ctx, true, a, t, sorted, n);
}

// TF-2.0 sparse Adam in its literal form (ORX_ADAM_DENSE): summed gradient rows into gsum + a dense-decay sweep of the whole table
int orx_adam_dense_sorted(orx_ctx* ctx, orx_opt* opt, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride) {
    CHECK(orx_table_scratch(t));
    OptSlots st;
    CHECK(orx_opt_slots(opt, t, &st));
    CHECK(orx_csr_accum(ctx, t, sorted, n, grads, g_stride));
    const double b1 = opt->p0, b2 = opt->p1;
    const double tt = (double)(opt->t > 0 ? opt->t : 1);
    const float lr_t = (float)(opt->lr * std::sqrt(1.0 - std::pow(b2, tt)) / (1.0 - std::pow(b1, tt)));
    return orx_launch_adam_sweep(ctx, t->w, st.s0, st.s1, t->gsum, t->rows * t->dim, lr_t, opt->p0, opt->p1, opt->p2);
}

// rows about to be read by a forward pass: replayed to the lazy optimizer's step (no-op for a table that is current)
int orx_table_touch(orx_table* t, const int32_t* ids, int64_t n) {
    if (t == nullptr || t->lazy == nullptr) return ORX_OK;
    return orx_adam_rows_touch(t->ctx, t->lazy, t, ids, n, false);
}

// Adagrad on per-occurrence gradient rows: dedup-sum semantics (duplicate flags + gsum + dup_apply)
int orx_adagrad_rows_apply(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n, const float* grads,
                           int64_t g_stride, ColWindows cw) {
    RowsArgs a;
    memset(&a, 0, sizeof(a));
    a.W = t->w; a.bias = bias ? bias->w : nullptr;
    a.ids = ids; a.grads = grads; a.g_stride = g_stride; a.n = n; a.rows = t->rows; a.D = t->dim;
    a.lr = opt->lr; a.err = ctx->d_err;
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
    if (cw.win != nullptr && cw.F > 0 && n % cw.F == 0) { d.col_F = cw.F; d.col_win = cw.win; }
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

extern "C" int orx_apply_rows(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias,
                              const int32_t* ids, int64_t n, const float* grads, int64_t g_stride) {
    const bool lazy = ctx && opt && t && orx_adam_rows_lazy(opt, t);
    if (t && !lazy) CHECK(orx_table_sync(t));
    if (bias && !lazy) CHECK(orx_table_sync(bias));
    ORX_ARG(ctx && opt && t && (n == 0 || (ids && grads)), "orx_apply_rows: NULL argument");
    ORX_ARG(opt->kind == ORX_SGD || opt->kind == ORX_ADAGRAD || (opt->kind == ORX_ADAM && (!bias || lazy)),
            "orx_apply_rows: the whole-table-sweep form of Adam (ORX_ADAM_DENSE) takes no bias column");
    ORX_ARG(g_stride >= t->dim + (bias ? 1 : 0), "orx_apply_rows: g_stride too small");
    ORX_ARG(!bias || (bias->dim == 1 && bias->rows == t->rows), "orx_apply_rows: bias must be [%lld, 1]", (long long)t->rows);
    if (n == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    if (bias == nullptr && t->dim <= 256 && getenv("ORX_ROWS_ATOMICS") == nullptr) {
        // the deterministic path (kernels_rowsort.hip): sort the (row, position) pairs, sum every row's gradients in position
        // order, apply the rule once per distinct row
        const uint2* sorted = nullptr;
        CHECK(orx_rows_sort(ctx, ids, 1, n, n, t->rows, &sorted));
        if (lazy) return orx_adam_rows_sorted(ctx, opt, t, sorted, n, grads, g_stride, true);
        if (opt->kind == ORX_ADAM) return orx_adam_dense_sorted(ctx, opt, t, sorted, n, grads, g_stride);
        return orx_csr_apply(ctx, opt, t, sorted, n, grads, g_stride);
    }
    RowsArgs a;
    memset(&a, 0, sizeof(a));
    a.W = t->w; a.bias = bias ? bias->w : nullptr;
    a.ids = ids; a.grads = grads; a.g_stride = g_stride; a.n = n; a.rows = t->rows; a.D = t->dim;
    a.lr = opt->lr; a.err = ctx->d_err;
    if (opt->kind == ORX_SGD) return orx_launch_apply_rows(ctx, ORX_SGD, false, a);
    if (lazy) {             // the bias column is a [rows, 1] table of its own with the same id list (one duplicate analysis)
        CHECK(orx_adam_rows_apply(ctx, opt, t, ids, n, grads, g_stride, false));
        if (bias) CHECK(orx_adam_rows_apply(ctx, opt, bias, ids, n, grads + t->dim, g_stride, true));
        return ORX_OK;
    }
    if (opt->kind == ORX_ADAM) {
        // TF-2.0 sparse Adam: summed gradient rows + a dense-decay sweep of the whole table.
        // The caller advances opt->t once per step (see dlrm.hip).
        CHECK(orx_table_scratch(t));
        OptSlots st;
        CHECK(orx_opt_slots(opt, t, &st));
        CHECK(orx_launch_rows_accum(ctx, t->gsum, ids, grads, g_stride, n, t->dim, t->rows));
        const double b1 = opt->p0, b2 = opt->p1;
        const double tt = (double)(opt->t > 0 ? opt->t : 1);
        const float lr_t = (float)(opt->lr * std::sqrt(1.0 - std::pow(b2, tt)) / (1.0 - std::pow(b1, tt)));
        return orx_launch_adam_sweep(ctx, t->w, st.s0, st.s1, t->gsum, t->rows * t->dim, lr_t, opt->p0, opt->p1, opt->p2);
    }

    return orx_adagrad_rows_apply(ctx, opt, t, bias, ids, n, grads, g_stride);
}

// ---- planned apply of K id lists against one table (see orx_internal.h) ------------------------------------------
int orx_apply_rows_plan(orx_ctx* c, orx_table* t, const int32_t* ids, int64_t K, int64_t n, int64_t id_stride, RowsPlan* out) {
    out->ready = false;
    if (K == 0 || n == 0) return ORX_OK;
    ORX_ARG(orx_fused_can_inline_apply(t->dim) && t->rows < (1LL << 28), "apply_rows_plan: needs a float4 dim and fewer than 2^28 rows");
    ORX_HIP(hipSetDevice(c->device));
    CHECK(orx_table_scratch(t, true));
    const int nb_total = orx_dedup_buckets(t->rows);
    CHECK(orx_exact_buffers(c, t, t, K, n, MODE_EXACT, true, false, true, nb_total, 1, &out->plan));
    ORX_ARG(out->plan.chunk >= K, "apply_rows_plan: %lld lists of %lld ids exceed one plan chunk (%lld)", (long long)K, (long long)n,
            (long long)out->plan.chunk);
    // the id lists play the ITEM role (so that dup_apply also finishes a bias column); no user list
    CHECK(orx_exact_plan_chunk(c, t, t, ids, ids, ids, id_stride, 0, n, 0, K, n, true, false, true, out->plan, &out->ck));
    out->n = n; out->ready = true;
    return ORX_OK;
}

int orx_apply_rows_planned_step(orx_ctx* c, orx_opt* opt, orx_table* t, orx_table* bias, const RowsPlan& rp, int64_t i,
                                const int32_t* ids, const float* grads, int64_t g_stride) {
    ORX_ARG(rp.ready && (opt->kind == ORX_SGD || opt->kind == ORX_ADAGRAD), "apply_rows_planned_step: no plan, or not SGD / Adagrad");
    if (bias) CHECK(orx_table_scratch(bias, true));
    OptSlots st, sb;
    CHECK(orx_opt_slots(opt, t, &st));
    if (bias) CHECK(orx_opt_slots(opt, bias, &sb));
    const int64_t n = rp.n;
    PairArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.V = t->w; pa.gV = t->gsum; pa.gV2 = t->gsum2; pa.aV = st.s0; pa.role_bits = 1;
    pa.b = bias ? bias->w : nullptr; pa.gb = bias ? bias->gsum : nullptr; pa.gb2 = bias ? bias->gsum2 : nullptr; pa.ab = bias ? sb.s0 : nullptr;
    pa.B = n; pa.D = t->dim; pa.lr = opt->lr; pa.eps = opt->kind == ORX_ADAGRAD ? opt->p1 : 0.f;
    orx_exact_step_views(c, rp.plan, i, n, t->dim, rp.ck.use_stage, &pa);
    RowsArgs a;
    memset(&a, 0, sizeof(a));
    a.W = t->w; a.bias = pa.b; a.G = t->gsum; a.G2 = t->gsum2; a.gb = pa.gb; a.gb2 = pa.gb2; a.A = st.s0; a.ab = pa.ab;
    a.ids = ids; a.ids2 = c->d_ids2 + (size_t)i * 3 * rp.plan.Bp + rp.plan.Bp;      // the item list of step i
    a.grads = grads; a.g_stride = g_stride; a.n = n; a.rows = t->rows; a.D = t->dim; a.lr = pa.lr; a.eps = pa.eps; a.err = c->d_err;
    a.refinfo = pa.refinfo ? pa.refinfo + rp.plan.Bp : nullptr; a.segstart = pa.segstart; a.stage = pa.stage; a.stageb = pa.stageb;
    CHECK(orx_launch_rows_planned(c, opt->kind, a));
    for (int l = 0; l < rp.ck.tree_levels; ++l) CHECK(orx_launch_hot_reduce(c, pa, l));
    return orx_launch_dup_apply(c, opt->kind, pa);
}

// duplicate flags of K id lists against a table of `rows` rows: dflag[k*n + i] = 1 iff the row ids[k*id_stride + i]
// occurs more than once in list k (ids < 0 are skipped, their flag is not written)
extern "C" int orx_rows_dupflags(orx_ctx* ctx, int64_t rows, const int32_t* ids, int64_t K, int64_t n, int64_t id_stride,
                                 unsigned char* dflag) {
    ORX_ARG(ctx && rows > 0 && (K == 0 || n == 0 || (ids && dflag)), "orx_rows_dupflags: bad argument");
    if (K == 0 || n == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    const int64_t list_stride = n / 2 + 1;
    ENSURE(ctx->d_dlist, ctx->d_dlist_cap, (size_t)K * list_stride * sizeof(uint32_t));
    ENSURE(ctx->d_dcount, ctx->d_dcount_cap, (size_t)K * sizeof(int));
    ORX_HIP(hipMemsetAsync(ctx->d_dcount, 0, (size_t)K * sizeof(int), ctx->stream));
    DedupArgs d;
    memset(&d, 0, sizeof(d));
    d.uid = ids; d.pid = ids; d.nid = ids; d.id_stride = id_stride;
    d.dflag = dflag; d.dlist = ctx->d_dlist; d.dcount = ctx->d_dcount;
    d.flag_stride = n; d.list_stride = list_stride;
    d.nU = 0; d.nP = n; d.nN = 0; d.NU = 0; d.NI = rows; d.nbu = 0; d.nbi = orx_dedup_buckets(rows);
    return orx_launch_dedup(ctx, d, K);
}

// orx_apply_rows for SGD with the duplicate flags of the id list already known (orx_rows_dupflags)
int orx_apply_rows_flagged_impl(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                                const float* grads, int64_t g_stride, const float* gbias, const unsigned char* dflag);
extern "C" int orx_apply_rows_flagged(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                                      const float* grads, int64_t g_stride, const unsigned char* dflag) {
    return orx_apply_rows_flagged_impl(ctx, opt, t, bias, ids, n, grads, g_stride, nullptr, dflag);
}

static int flagged_args(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                        const float* grads, int64_t g_stride, const float* gbias, const unsigned char* dflag, RowsArgs* out) {
    if (t) CHECK(orx_table_sync(t));
    if (bias) CHECK(orx_table_sync(bias));
    ORX_ARG(ctx && opt && t && (n == 0 || (ids && grads && dflag)), "orx_apply_rows_flagged: NULL argument");
    ORX_ARG(opt->kind == ORX_SGD, "orx_apply_rows_flagged: SGD only (Adagrad / Adam: orx_apply_rows)");
    ORX_ARG(g_stride >= t->dim + ((bias && !gbias) ? 1 : 0), "orx_apply_rows_flagged: g_stride too small");
    ORX_ARG(!bias || (bias->dim == 1 && bias->rows == t->rows), "orx_apply_rows_flagged: bias must be [%lld, 1]", (long long)t->rows);
    RowsArgs a;
    memset(&a, 0, sizeof(a));
    a.W = t->w; a.bias = bias ? bias->w : nullptr;
    a.ids = ids; a.grads = grads; a.g_stride = g_stride; a.n = n; a.rows = t->rows; a.D = t->dim;
    a.lr = opt->lr; a.err = ctx->d_err; a.dflag = dflag; a.gbias = gbias;
    *out = a;
    return ORX_OK;
}

// gbias != NULL: the bias gradients come in an array of their own ([n]) instead of column dim of the gradient rows
int orx_apply_rows_flagged_impl(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                                const float* grads, int64_t g_stride, const float* gbias, const unsigned char* dflag) {
    RowsArgs a;
    CHECK(flagged_args(ctx, opt, t, bias, ids, n, grads, g_stride, gbias, dflag, &a));
    if (n == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    return orx_launch_apply_rows(ctx, ORX_SGD, true, a);
}

int orx_apply_rows_flagged_pair(orx_ctx* ctx, orx_opt* opt, orx_table* tA, orx_table* biasA, const int32_t* idsA, int64_t nA, const float* gA, int64_t strideA,
                                const float* gbiasA, const unsigned char* flagA, orx_table* tB, orx_table* biasB, const int32_t* idsB, int64_t nB, const float* gB,
                                int64_t strideB, const float* gbiasB, const unsigned char* flagB, bool flags_over_both) {
    RowsArgs a, b;
    CHECK(flagged_args(ctx, opt, tA, biasA, idsA, nA, gA, strideA, gbiasA, flagA, &a));
    CHECK(flagged_args(ctx, opt, tB, biasB, idsB, nB, gB, strideB, gbiasB, flagB, &b));
    ORX_HIP(hipSetDevice(ctx->device));
    int rc = ORX_OK;
    // (two lists of ONE table: only where the caller's flags were made over both lists together -- a row in both is then an atomic add in both)
    if ((tA != tB || flags_over_both) && orx_launch_apply_rows_pair(ctx, a, b, &rc)) return rc;
    if (nA) CHECK(orx_launch_apply_rows(ctx, ORX_SGD, true, a));
    if (nB) CHECK(orx_launch_apply_rows(ctx, ORX_SGD, true, b));
    return ORX_OK;
}

// ------------------------------------------------- device-side exchange plan ---
extern "C" int orx_shard_route(orx_ctx* ctx, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t B,
                               int64_t users_global, int64_t items_global, int32_t world, int32_t cap,
                               int32_t* send, int32_t* counters, int32_t* overflow) {
    ORX_ARG(ctx && uid && pid && nid && send && counters && overflow, "orx_shard_route: NULL argument");
    ORX_ARG(world >= 1 && world <= 64 && cap >= 1, "orx_shard_route: world must be in [1, 64] and cap positive");
    ORX_HIP(hipSetDevice(ctx->device));
    ORX_HIP(hipMemsetAsync(send, 0xFF, (size_t)world * cap * 3 * sizeof(int32_t), ctx->stream));
    ORX_HIP(hipMemsetAsync(counters, 0, (size_t)world * sizeof(int32_t), ctx->stream));
    RouteArgs a;
    memset(&a, 0, sizeof(a));
    a.uid = uid; a.pid = pid; a.nid = nid; a.B = B; a.world = world; a.cap = cap;
    a.send = send; a.counters = counters; a.overflow = overflow; a.err = ctx->d_err;
    a.NU = users_global; a.NI = items_global;
    return orx_launch_shard_route(ctx, a);
}

// K steps at once (the plan of a K-step call depends on the ids alone): per-step arrays are contiguous,
// uid/pid/nid [K][id_stride], send [K][world*cap][3], counters [K][world]
extern "C" int orx_shard_route_steps(orx_ctx* ctx, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t K, int64_t B,
                                     int64_t id_stride, int64_t users_global, int64_t items_global, int32_t world, int32_t cap,
                                     int32_t* send, int32_t* counters, int32_t* overflow) {
    ORX_ARG(ctx && uid && pid && nid && send && counters && overflow, "orx_shard_route_steps: NULL argument");
    ORX_ARG(world >= 1 && world <= 64 && cap >= 1 && K >= 0 && K < 65536, "orx_shard_route_steps: world in [1, 64], cap positive, K < 65536");
    ORX_HIP(hipSetDevice(ctx->device));
    ORX_HIP(hipMemsetAsync(send, 0xFF, (size_t)K * world * cap * 3 * sizeof(int32_t), ctx->stream));
    ORX_HIP(hipMemsetAsync(counters, 0, (size_t)K * world * sizeof(int32_t), ctx->stream));
    RouteArgs a;
    memset(&a, 0, sizeof(a));
    a.uid = uid; a.pid = pid; a.nid = nid; a.B = B; a.world = world; a.cap = cap; a.id_stride = id_stride;
    a.send = send; a.counters = counters; a.overflow = overflow; a.err = ctx->d_err;
    a.NU = users_global; a.NI = items_global;
    return orx_launch_shard_route(ctx, a, K);
}

extern "C" int orx_shard_request_steps(orx_ctx* ctx, const int32_t* trip, int64_t K, int64_t T, int32_t world, int32_t cap,
                                       int32_t* send_ids, int32_t* slot, int32_t* u_loc, int32_t* counters, int32_t* overflow) {
    ORX_ARG(ctx && trip && send_ids && slot && u_loc && counters && overflow, "orx_shard_request_steps: NULL argument");
    ORX_ARG(world >= 1 && world <= 64 && cap >= 1 && K >= 0 && K < 65536, "orx_shard_request_steps: world in [1, 64], cap positive, K < 65536");
    ORX_HIP(hipSetDevice(ctx->device));
    ORX_HIP(hipMemsetAsync(send_ids, 0xFF, (size_t)K * world * cap * sizeof(int32_t), ctx->stream));
    ORX_HIP(hipMemsetAsync(counters, 0, (size_t)K * world * sizeof(int32_t), ctx->stream));
    RequestArgs a;
    memset(&a, 0, sizeof(a));
    a.trip = trip; a.T = T; a.world = world; a.cap = cap;
    a.send_ids = send_ids; a.slot = slot; a.u_loc = u_loc; a.counters = counters; a.overflow = overflow;
    return orx_launch_shard_request(ctx, a, K);
}

// orx_shard_request_steps with per-destination dedup: an item several references of a list ask for claims ONE slot.
// slot[r] of those references is the same, dupref[r] = 1 marks them (their gradients are added into the slot: orx_shard_grads);
// the distinct items of an owner fill its bucket in ascending row order.
extern "C" int orx_shard_request_dedup_steps(orx_ctx* ctx, const int32_t* trip, int64_t K, int64_t T, int32_t world, int32_t cap,
                                             int64_t items_global, int32_t* send_ids, int32_t* slot, int32_t* u_loc,
                                             uint8_t* dupref, void* sorted_out, void* seglist, int32_t* segcount, int32_t* overflow) {
    ORX_ARG(ctx && trip && send_ids && slot && u_loc && dupref && sorted_out && seglist && segcount && overflow, "orx_shard_request_dedup_steps: NULL argument");
    ORX_ARG(world >= 1 && world <= 64 && cap >= 1 && K >= 0 && K < 65536 && items_global > 0, "orx_shard_request_dedup_steps: world in [1, 64], cap positive, K < 65536");
    if (K == 0 || T == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    ORX_HIP(hipMemsetAsync(send_ids, 0xFF, (size_t)K * world * cap * sizeof(int32_t), ctx->stream));
    DedupReqArgs a;
    memset(&a, 0, sizeof(a));
    a.trip = trip; a.T = T; a.world = world; a.cap = cap; a.Lr = (items_global + world - 1) / world;
    a.nchunk = (int)((2 * T + 1023) / 1024);
    // scratch: keys [K][2T] | uq [K][2T] | chunkcnt [K][nchunk] | ostart [K][64]
    ENSURE(ctx->d_tmp, ctx->d_tmp_cap, ((size_t)K * 2 * T * 2 + (size_t)K * (a.nchunk + 64)) * sizeof(int32_t));
    a.keys = (int32_t*)ctx->d_tmp; a.uq = a.keys + (size_t)K * 2 * T; a.chunkcnt = a.uq + (size_t)K * 2 * T; a.ostart = a.chunkcnt + (size_t)K * a.nchunk;
    a.send_ids = send_ids; a.slot = slot; a.u_loc = u_loc; a.dupref = dupref; a.overflow = overflow;
    a.seglist = (int2*)seglist; a.segcount = segcount;
    CHECK(orx_launch_shard_keys(ctx, a, K));
    const uint2* sorted = nullptr;
    CHECK(orx_rows_sort(ctx, a.keys, K, 2 * T, 2 * T, (int64_t)world * a.Lr, &sorted));
    // (the context's sort buffers are reused by the next sort -- the applies of the steps sort too: the plan keeps a copy)
    ORX_HIP(hipMemcpyAsync(sorted_out, sorted, (size_t)K * 2 * T * sizeof(uint2), hipMemcpyDeviceToDevice, ctx->stream));
    a.sorted = (const uint2*)sorted_out;
    return orx_launch_shard_dedup_slots(ctx, a, K);
}

extern "C" int orx_shard_request(orx_ctx* ctx, const int32_t* trip, int64_t T, int32_t world, int32_t cap,
                                 int32_t* send_ids, int32_t* slot, int32_t* u_loc, int32_t* counters, int32_t* overflow) {
    ORX_ARG(ctx && trip && send_ids && slot && u_loc && counters && overflow, "orx_shard_request: NULL argument");
    ORX_ARG(world >= 1 && world <= 64 && cap >= 1, "orx_shard_request: world must be in [1, 64] and cap positive");
    ORX_HIP(hipSetDevice(ctx->device));
    ORX_HIP(hipMemsetAsync(send_ids, 0xFF, (size_t)world * cap * sizeof(int32_t), ctx->stream));
    ORX_HIP(hipMemsetAsync(counters, 0, (size_t)world * sizeof(int32_t), ctx->stream));
    RequestArgs a;
    memset(&a, 0, sizeof(a));
    a.trip = trip; a.T = T; a.world = world; a.cap = cap;
    a.send_ids = send_ids; a.slot = slot; a.u_loc = u_loc; a.counters = counters; a.overflow = overflow;
    return orx_launch_shard_request(ctx, a);
}

extern "C" int orx_shard_bucket(orx_ctx* ctx, const int32_t* ids, int64_t n, int32_t world, int32_t cap,
                                int32_t* send_ids, int32_t* slot, int32_t* counters, int32_t* overflow) {
    ORX_ARG(ctx && (n == 0 || ids) && send_ids && slot && counters && overflow, "orx_shard_bucket: NULL argument");
    ORX_ARG(world >= 1 && world <= 64 && cap >= 1, "orx_shard_bucket: world must be in [1, 64] and cap positive");
    ORX_HIP(hipSetDevice(ctx->device));
    ORX_HIP(hipMemsetAsync(send_ids, 0xFF, (size_t)world * cap * sizeof(int32_t), ctx->stream));
    ORX_HIP(hipMemsetAsync(counters, 0, (size_t)world * sizeof(int32_t), ctx->stream));
    return orx_launch_shard_bucket(ctx, ids, n, world, cap, send_ids, slot, counters, overflow);
}

extern "C" int orx_shard_localize(orx_ctx* ctx, const int32_t* ids, int64_t n, int32_t world, int32_t* out) {
    ORX_ARG(ctx && (n == 0 || (ids && out)) && world >= 1, "orx_shard_localize: bad argument");
    ORX_HIP(hipSetDevice(ctx->device));
    return orx_launch_shard_localize(ctx, ids, n, world, out);
}

// Both gradient entry points and the library's own sharded engine.  opt / dup_u / u_apply != NULL: SGD's apply of the user rows
// referenced once folded in.  bias_in / gb_out != NULL: the biases travel apart from the rows (row_stride = D; the side buffer of
// shared slots keeps rows of D + 4 floats with the bias gradient at column D), else at column D of the rows (row_stride > D).
int orx_shard_grads_impl(orx_ctx* ctx, int model, orx_opt* opt, orx_table* user, const float* rows_in, const float* bias_in,
                         const int32_t* u_loc, const int32_t* slot, const uint8_t* dupref, const void* sorted, const void* seglist,
                         const int32_t* segcount, float* gdup, const uint8_t* dup_u, int64_t T, int64_t row_stride, int64_t B_global,
                         float margin, int flags, float* gu, int32_t* u_apply, float* send_g, float* gb_out, double* loss_l2_accum,
                         float* partial_ext, int* nwaves_out) {
    const bool fold = dup_u != nullptr;
    if (fold) CHECK(orx_table_sync(user));
    else if (user && u_loc) CHECK(orx_table_touch(user, u_loc, T));   // a lazy table: the user rows read here are brought up to date
    ORX_ARG(ctx && user && rows_in && u_loc && slot && gu && send_g && (!fold || (opt && u_apply)), "orx_shard_grads: NULL argument");
    ORX_ARG(!fold || opt->kind == ORX_SGD, "orx_shard_grads_sgd: the folded apply is SGD's (Adagrad / Adam sum duplicates first: orx_shard_grads + orx_apply_rows)");
    ORX_ARG(model == ORX_BPR || model == ORX_UCML, "orx_shard_grads: unknown model %d", model);
    ORX_ARG((bias_in != nullptr) == (gb_out != nullptr), "orx_shard_grads: biases in and bias gradients out travel apart together");
    ORX_ARG((bias_in ? row_stride >= user->dim : row_stride > user->dim) && B_global > 0, "orx_shard_grads: row_stride must leave room for the bias column");
    ORX_ARG(!dupref || (sorted && seglist && segcount && gdup), "orx_shard_grads: dedup needs the plan's sorted list, segment list and a side buffer");
    if (T == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(ctx->device));
    ShardGradArgs a;
    memset(&a, 0, sizeof(a));
    a.U = user->w; a.rows_in = rows_in; a.u_loc = u_loc; a.slot = slot; a.gu = gu; a.send_g = send_g; a.dupref = dupref; a.gdup = gdup;
    a.bias_in = bias_in; a.gb_out = gb_out;
    a.T = T; a.D = user->dim; a.DS = (int)row_stride; a.DSg = bias_in ? user->dim + 4 : (int)row_stride;
    a.invB = 1.0f / (float)B_global; a.margin = margin; a.l2w = (flags & ORX_NO_L2) ? 0.f : 1.f;
    if (fold) { a.fu = dup_u; a.Uw = user->w; a.lr = opt->lr; a.u_apply = u_apply; }
    // partial_ext (the library's engine): the launch's loss partials stay in the caller's buffer, which adds a whole chunk of steps
    // to the sums with ONE launch (a launch per step and half cost 4 us + a kernel boundary of every 84 us step)
    if (partial_ext == nullptr) ENSURE(ctx->d_partial, ctx->d_partial_cap, (size_t)(T + 4) * 2 * sizeof(float));
    a.partial = partial_ext ? partial_ext : ctx->d_partial;
    int nw = 0;
    CHECK(orx_launch_shard_grads(ctx, model, a, &nw));
    if (nwaves_out) *nwaves_out = nw;
    if (dupref) CHECK(orx_launch_shard_segsum(ctx, (const int2*)seglist, segcount, (const uint2*)sorted, 2 * T, gdup, a.DSg, send_g, a.DS, a.D, gb_out));
    if (loss_l2_accum && partial_ext == nullptr) CHECK(orx_launch_loss_accumulate(ctx, ctx->d_partial, nw, loss_l2_accum));
    return ORX_OK;
}

extern "C" int orx_shard_grads(orx_ctx* ctx, int model, orx_table* user, const float* rows_in, const int32_t* u_loc,
                               const int32_t* slot, const uint8_t* dupref, const void* sorted, const void* seglist, const int32_t* segcount,
                               float* gdup, int64_t T, int64_t row_stride, int64_t B_global, float margin, int flags,
                               float* gu, float* send_g, double* loss_l2_accum) {
    return orx_shard_grads_impl(ctx, model, nullptr, user, rows_in, nullptr, u_loc, slot, dupref, sorted, seglist, segcount, gdup, nullptr, T, row_stride,
                                B_global, margin, flags, gu, nullptr, send_g, nullptr, loss_l2_accum);
}

// orx_shard_grads with the SGD apply of the (local) user rows folded in: rows referenced once in the step are updated in
// place by the gradient kernel, the references of duplicated rows leave (u_apply[t] = local row, gu[t] = gradient) for
// orx_apply_rows_flagged; u_apply[t] = -1 everywhere else.  dup_u: the flags orx_rows_dupflags made for u_loc.
extern "C" int orx_shard_grads_sgd(orx_ctx* ctx, int model, orx_opt* opt, orx_table* user, const float* rows_in, const int32_t* u_loc,
                                   const int32_t* slot, const uint8_t* dupref, const void* sorted, const void* seglist, const int32_t* segcount,
                                   float* gdup, const uint8_t* dup_u, int64_t T, int64_t row_stride, int64_t B_global,
                                   float margin, int flags, float* gu, int32_t* u_apply, float* send_g, double* loss_l2_accum) {
    ORX_ARG(opt && dup_u && u_apply, "orx_shard_grads_sgd: NULL argument");
    return orx_shard_grads_impl(ctx, model, opt, user, rows_in, nullptr, u_loc, slot, dupref, sorted, seglist, segcount, gdup, dup_u, T, row_stride,
                                B_global, margin, flags, gu, u_apply, send_g, nullptr, loss_l2_accum);
}

// ------------------------------------------------------------- ranking metrics ---
extern "C" int orx_rank_metrics(orx_ctx* c, int kind, orx_table* U, orx_table* V, orx_table* b, orx_table* w,
                                const int32_t* uid, const float* pred, const uint8_t* pos_mask, const uint8_t* excl_mask,
                                int64_t n, int64_t items, const float* at, int32_t nat,
                                float* auc, float* ndcg, float* recall) {
    if (U) CHECK(orx_table_sync(U));
    if (V) CHECK(orx_table_sync(V));
    if (b) CHECK(orx_table_sync(b));
    if (w) CHECK(orx_table_sync(w));
    ORX_ARG(c && pos_mask && excl_mask && at, "orx_rank_metrics: NULL argument");
    ORX_ARG(nat >= 1 && nat <= 16, "orx_rank_metrics: nat must be in [1, 16]");
    ORX_ARG(pred || (U && V && b && uid), "orx_rank_metrics: need either pred or tables + user ids");
    ORX_ARG(pred || V->rows == items, "orx_rank_metrics: items must equal the item table's rows");
    if (n == 0) return ORX_OK;
    ORX_HIP(hipSetDevice(c->device));
    const size_t cells = (size_t)n * items;
    // layout of d_tmp: pred [cells] | auc [n] | ndcg [n*nat] | recall [n*nat] | at [nat]
    const size_t nf = cells + (size_t)n * (1 + 2 * nat) + nat;
    ENSURE(c->d_tmp, c->d_tmp_cap, nf * sizeof(float));
    ENSURE(c->d_dflag, c->d_dflag_cap, 2 * cells);
    float* d_pred = c->d_tmp; float* d_auc = d_pred + cells; float* d_ndcg = d_auc + n; float* d_rec = d_ndcg + (size_t)n * nat;
    float* d_at = d_rec + (size_t)n * nat;
    unsigned char* d_pos = c->d_dflag; unsigned char* d_excl = d_pos + cells;
    ORX_HIP(hipMemcpyAsync(d_pos, pos_mask, cells, hipMemcpyHostToDevice, c->stream));
    ORX_HIP(hipMemcpyAsync(d_excl, excl_mask, cells, hipMemcpyHostToDevice, c->stream));
    ORX_HIP(hipMemcpyAsync(d_at, at, sizeof(float) * nat, hipMemcpyHostToDevice, c->stream));
    if (pred) {
        ORX_HIP(hipMemcpyAsync(d_pred, pred, cells * sizeof(float), hipMemcpyHostToDevice, c->stream));
    } else {
        ORX_ARG(kind >= 0 && kind <= 2 && U->dim == V->dim && U->dim <= 1024, "orx_rank_metrics: bad scorer arguments");
        ENSURE(c->d_ids, c->d_ids_cap, (size_t)n * sizeof(int32_t));
        CHECK(stage_ids(c, uid, n, 0));
        CHECK(orx_launch_score_all(c, U->w, V->w, b->w, w ? w->w : nullptr, c->d_ids, n, U->rows, V->rows, U->dim, kind, d_pred));
    }
    EvalArgs a;
    a.pred = d_pred; a.pos = d_pos; a.excl = d_excl; a.NI = items; a.at = d_at; a.nat = nat;
    a.auc = d_auc; a.ndcg = d_ndcg; a.recall = d_rec; a.err = c->d_err;
    CHECK(orx_launch_rank_metrics(c, a, n));
    if (auc) ORX_HIP(hipMemcpyAsync(auc, d_auc, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    if (ndcg) ORX_HIP(hipMemcpyAsync(ndcg, d_ndcg, sizeof(float) * n * nat, hipMemcpyDeviceToHost, c->stream));
    if (recall) ORX_HIP(hipMemcpyAsync(recall, d_rec, sizeof(float) * n * nat, hipMemcpyDeviceToHost, c->stream));
    ORX_HIP(hipStreamSynchronize(c->stream));
    int flag = 0;
    ORX_HIP(hipMemcpy(&flag, c->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (flag == 2) {
        ORX_HIP(hipMemset(c->d_err, 0, sizeof(int)));
        orx_set_error("orx_rank_metrics: a user has more than 8192 positive items");
        return ORX_ERR_ARG;
    }
    return orx_check_index_error(c);
}

// The same metrics with the masks as CSR lists over the call's users (pos_ptr / excl_ptr: host int64[n + 1] starting at 0).
extern "C" int orx_rank_metrics_csr(orx_ctx* c, int kind, orx_table* U, orx_table* V, orx_table* b, orx_table* w,
                                    const int32_t* uid, const float* pred, int32_t pred_on_device, int64_t n, int64_t items,
                                    const int64_t* pos_ptr, const int32_t* pos_items, const int64_t* excl_ptr, const int32_t* excl_items,
                                    const float* at, int32_t nat, float* auc, float* ndcg, float* recall) {
    if (U) CHECK(orx_table_sync(U));
    if (V) CHECK(orx_table_sync(V));
    if (b) CHECK(orx_table_sync(b));
    if (w) CHECK(orx_table_sync(w));
    ORX_ARG(c && pos_ptr && excl_ptr && at && n >= 0 && items > 0, "orx_rank_metrics_csr: NULL argument");
    ORX_ARG(nat >= 1 && nat <= 16, "orx_rank_metrics_csr: nat must be in [1, 16]");
    ORX_ARG(pred || (U && V && b && uid), "orx_rank_metrics_csr: need either pred or tables + user ids");
    ORX_ARG(pred || V->rows == items, "orx_rank_metrics_csr: items must equal the item table's rows");
    if (n == 0) return ORX_OK;
    ORX_ARG(pos_ptr[0] == 0 && excl_ptr[0] == 0, "orx_rank_metrics_csr: the lists start at offset 0");
    int64_t max_pos = 0;
    for (int64_t q = 0; q < n; ++q) {
        const int64_t lp = pos_ptr[q + 1] - pos_ptr[q], le = excl_ptr[q + 1] - excl_ptr[q];
        ORX_ARG(lp >= 0 && le >= 0, "orx_rank_metrics_csr: offsets must not decrease");
        if (lp > max_pos) max_pos = lp;
    }
    const int64_t npos = pos_ptr[n], nexcl = excl_ptr[n];
    ORX_ARG((npos == 0 || pos_items) && (nexcl == 0 || excl_items), "orx_rank_metrics_csr: NULL item list");
    ORX_HIP(hipSetDevice(c->device));
    const size_t cells = (size_t)n * items;
    const int64_t W = (items + 31) / 32;
    const bool dev_pred = pred && pred_on_device;
    const int S = orx_rank_csr_segments(n, items);
    // one upload: pos_ptr [n+1] | excl_ptr [n+1] (int64) | pos_items | excl_items (int32) | at [16] (float)
    const size_t up_bytes = 2 * (size_t)(n + 1) * 8 + ((size_t)npos + nexcl + 16) * 4;
    // one download: auc [n] | ndcg [n nat] | recall [n nat] | error flag
    const size_t nout = (size_t)n * (1 + 2 * nat) + 1;
    static thread_local std::vector<char> pack;
    pack.resize(up_bytes > nout * 4 ? up_bytes : nout * 4);
    {
        char* h = pack.data();
        memcpy(h, pos_ptr, 8 * (size_t)(n + 1)); h += 8 * (size_t)(n + 1);
        memcpy(h, excl_ptr, 8 * (size_t)(n + 1)); h += 8 * (size_t)(n + 1);
        if (npos) memcpy(h, pos_items, 4 * (size_t)npos);
        h += 4 * (size_t)npos;
        if (nexcl) memcpy(h, excl_items, 4 * (size_t)nexcl);
        h += 4 * (size_t)nexcl;
        memset(h, 0, 64); memcpy(h, at, sizeof(float) * nat);
    }
    // d_dflag: the upload | partials [n S 136] | n_eval [n] | results [nout];  d_evalbits: the two bitmaps [2 n W], all zero between calls
    const size_t bytes = up_bytes + ((size_t)n * S * 136 + n + nout) * 4;
    ENSURE(c->d_dflag, c->d_dflag_cap, bytes);
    const size_t bit_bytes = 2 * (size_t)n * W * 4;
    if (bit_bytes > c->d_evalbits_cap) {
        if (c->d_evalbits) ORX_HIP(hipFree(c->d_evalbits));
        c->d_evalbits = nullptr; c->d_evalbits_cap = 0;
        ORX_HIP(hipMalloc((void**)&c->d_evalbits, bit_bytes));
        c->d_evalbits_cap = bit_bytes;
        ORX_HIP(hipMemsetAsync(c->d_evalbits, 0, bit_bytes, c->stream));
    }
    if (!dev_pred) ENSURE(c->d_tmp, c->d_tmp_cap, cells * sizeof(float));
    float* d_pred = dev_pred ? const_cast<float*>(pred) : c->d_tmp;
    int64_t* d_pp = (int64_t*)c->d_dflag; int64_t* d_ep = d_pp + (n + 1);
    int32_t* d_pi = (int32_t*)(d_ep + (n + 1)); int32_t* d_ei = d_pi + npos;
    float* d_at = (float*)(d_ei + nexcl);
    unsigned* d_part = (unsigned*)(d_at + 16); int* d_neval = (int*)(d_part + (size_t)n * S * 136);
    float* d_auc = (float*)(d_neval + n); float* d_ndcg = d_auc + n; float* d_rec = d_ndcg + (size_t)n * nat;
    int* d_flag = (int*)(d_rec + (size_t)n * nat);
    ORX_HIP(hipMemcpyAsync(c->d_dflag, pack.data(), up_bytes, hipMemcpyHostToDevice, c->stream));
    EvalCsrArgs a;
    a.pred = d_pred; a.pbits = c->d_evalbits; a.ebits = c->d_evalbits + (size_t)n * W; a.pos_ptr = d_pp; a.pos_items = d_pi;
    a.excl_ptr = d_ep; a.excl_items = d_ei; a.NI = items; a.W = W; a.at = d_at; a.nat = nat;
    a.auc = d_auc; a.ndcg = d_ndcg; a.recall = d_rec; a.err = c->d_err; a.part = d_part; a.neval = d_neval; a.S = S;
    a.flag_out = d_flag; a.q0 = 0;
    // Everything that can fail without touching the bitmaps comes FIRST: the bitmaps are all zero between calls, the first
    // orx_launch_mask_bits sets bits and only the last one clears them again -- an early return in between would leave every later
    // evaluation on this context with stale masks.
    const bool scorer = !(dev_pred || pred);
    if (scorer) {
        ORX_ARG(kind >= 0 && kind <= 2 && U->dim == V->dim && U->dim <= 1024, "orx_rank_metrics_csr: bad scorer arguments");
        ENSURE(c->d_ids, c->d_ids_cap, (size_t)n * sizeof(int32_t));
        CHECK(stage_ids(c, uid, n, 0));
    }
    CHECK(orx_launch_mask_bits(c, a, n, 0));
    // ... and whatever fails from here on clears them before it leaves
    auto body = [&]() -> int {
        if (!scorer) {
            if (!dev_pred) ORX_HIP(hipMemcpyAsync(d_pred, pred, cells * sizeof(float), hipMemcpyHostToDevice, c->stream));
            CHECK(orx_launch_rank_sweeps(c, a, 0, n, max_pos));
        } else {
            // (scoring and sweeping slices of the batch on two streams -- a write stream beside a read stream -- was measured and is
            // slower at every slice count: profiles/r3_eval_notes.txt)
            CHECK(orx_launch_score_all(c, U->w, V->w, b->w, w ? w->w : nullptr, c->d_ids, n, U->rows, V->rows, U->dim, kind, d_pred));
            CHECK(orx_launch_rank_sweeps(c, a, 0, n, max_pos));
        }
        return ORX_OK;
    };
    const int body_rc = body();
    if (body_rc != ORX_OK) {
        hipMemsetAsync(c->d_evalbits, 0, bit_bytes, c->stream);
        return body_rc;
    }
    CHECK(orx_launch_mask_bits(c, a, n, 1));
    ORX_HIP(hipMemcpyAsync(pack.data(), d_auc, nout * 4, hipMemcpyDeviceToHost, c->stream));
    ORX_HIP(hipStreamSynchronize(c->stream));
    const float* r = (const float*)pack.data();
    if (auc) memcpy(auc, r, sizeof(float) * n);
    if (ndcg) memcpy(ndcg, r + n, sizeof(float) * n * nat);
    if (recall) memcpy(recall, r + n + (size_t)n * nat, sizeof(float) * n * nat);
    int flag; memcpy(&flag, r + nout - 1, sizeof(int));
    if (flag == 3) {
        orx_set_error("orx_rank_metrics_csr: a user's positive list repeats an item (the lists are sets)");
        return ORX_ERR_ARG;
    }
    if (flag) {
        orx_set_error("id out of range: an index in the batch is < 0 or >= the table's row count");
        return ORX_ERR_INDEX;
    }
    return ORX_OK;
}

// ------------------------------------------------------------- device sampler ---
struct orx_sampler {
    orx_ctx* ctx = nullptr;
    int32_t *rec_user = nullptr, *rec_item = nullptr, *items = nullptr;
    int64_t* ptr = nullptr;
    int64_t R = 0, total_users = 0, total_items = 0;
    int h = 1;
    // stratified pointwise stream: positives consumed so far (device), next sample index and seed the counter belongs to
    int64_t* d_counter = nullptr; int64_t strat_next = -1; uint64_t strat_seed = 0;
    int* d_blockcnt = nullptr; int64_t* d_blockbase = nullptr; size_t block_cap = 0;
};

extern "C" int orx_sampler_create(orx_ctx* ctx, const int32_t* rec_user, const int32_t* rec_item, int64_t n_records,
                                  const int64_t* csr_ptr, const int32_t* csr_items, int64_t total_users, int64_t total_items,
                                  orx_sampler** out) {
    ORX_ARG(ctx && rec_user && rec_item && csr_ptr && csr_items && out, "orx_sampler_create: NULL argument");
    ORX_ARG(n_records > 0 && total_users > 0 && total_items > 0, "orx_sampler_create: sizes must be positive");
    ORX_HIP(hipSetDevice(ctx->device));
    orx_sampler* s = new orx_sampler();
    s->ctx = ctx; s->R = n_records; s->total_users = total_users; s->total_items = total_items;
    while ((1ll << (2 * s->h)) < n_records) s->h++;
    const int64_t nnz = csr_ptr[total_users];
    ORX_HIP(hipMalloc((void**)&s->rec_user, sizeof(int32_t) * n_records));
    ORX_HIP(hipMalloc((void**)&s->rec_item, sizeof(int32_t) * n_records));
    ORX_HIP(hipMalloc((void**)&s->ptr, sizeof(int64_t) * (total_users + 1)));
    ORX_HIP(hipMalloc((void**)&s->items, sizeof(int32_t) * (nnz > 0 ? nnz : 1)));
    ORX_HIP(hipMemcpy(s->rec_user, rec_user, sizeof(int32_t) * n_records, hipMemcpyHostToDevice));
    ORX_HIP(hipMemcpy(s->rec_item, rec_item, sizeof(int32_t) * n_records, hipMemcpyHostToDevice));
    ORX_HIP(hipMemcpy(s->ptr, csr_ptr, sizeof(int64_t) * (total_users + 1), hipMemcpyHostToDevice));
    if (nnz > 0) ORX_HIP(hipMemcpy(s->items, csr_items, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
    *out = s;
    return ORX_OK;
}

extern "C" int orx_sampler_destroy(orx_sampler* s) {
    if (!s) return ORX_OK;
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    hipFree(s->rec_user); hipFree(s->rec_item); hipFree(s->ptr); hipFree(s->items);
    hipFree(s->d_counter); hipFree(s->d_blockcnt); hipFree(s->d_blockbase);
    delete s;
    return ORX_OK;
}

extern "C" int orx_sampler_pairwise(orx_sampler* s, uint64_t seed, int64_t first, int64_t n,
                                    int32_t* uid_dev, int32_t* pid_dev, int32_t* nid_dev) {
    ORX_ARG(s && uid_dev && pid_dev && nid_dev && first >= 0 && n >= 0, "orx_sampler_pairwise: bad argument");
    ORX_HIP(hipSetDevice(s->ctx->device));
    SamplerArgs a;
    a.rec_user = s->rec_user; a.rec_item = s->rec_item; a.R = s->R; a.ptr = s->ptr; a.items = s->items;
    a.total_items = s->total_items; a.total_users = s->total_users; a.seed = seed; a.first = first; a.n = n; a.h = s->h;
    a.uid = uid_dev; a.pid = pid_dev; a.nid = nid_dev;
    return orx_launch_sample_pairwise(s->ctx, a);
}

static void sampler_args(orx_sampler* s, uint64_t seed, int64_t first, int64_t n, int32_t* uid, int32_t* iid, SamplerArgs* a) {
    a->rec_user = s->rec_user; a->rec_item = s->rec_item; a->R = s->R; a->ptr = s->ptr; a->items = s->items;
    a->total_items = s->total_items; a->total_users = s->total_users; a->seed = seed; a->first = first; a->n = n; a->h = s->h;
    a->uid = uid; a->pid = iid; a->nid = nullptr;
}

extern "C" int orx_sampler_stratified(orx_sampler* s, uint64_t seed, int64_t first, int64_t n, float pos_ratio,
                                      int32_t* uid_dev, int32_t* iid_dev, float* label_dev) {
    ORX_ARG(s && uid_dev && iid_dev && label_dev && first >= 0 && n >= 0, "orx_sampler_stratified: bad argument");
    ORX_ARG(pos_ratio >= 0.f && pos_ratio <= 1.f, "orx_sampler_stratified: pos_ratio must lie in [0, 1]");
    ORX_ARG(first == 0 || (first == s->strat_next && seed == s->strat_seed),
            "orx_sampler_stratified: the stream is sequential (a positive's place in the epoch counts the positives before it): "
            "continue at sample %lld of the same seed, or restart at 0", (long long)s->strat_next);
    ORX_HIP(hipSetDevice(s->ctx->device));
    if (!s->d_counter) ORX_HIP(hipMalloc((void**)&s->d_counter, sizeof(int64_t)));
    if (first == 0) ORX_HIP(hipMemsetAsync(s->d_counter, 0, sizeof(int64_t), s->ctx->stream));
    const size_t nblocks = (size_t)((n + 255) / 256);
    if (s->block_cap < nblocks) {
        ORX_HIP(hipStreamSynchronize(s->ctx->stream));
        hipFree(s->d_blockcnt); hipFree(s->d_blockbase); s->d_blockcnt = nullptr; s->d_blockbase = nullptr; s->block_cap = 0;
        ORX_HIP(hipMalloc((void**)&s->d_blockcnt, sizeof(int) * nblocks));
        ORX_HIP(hipMalloc((void**)&s->d_blockbase, sizeof(int64_t) * nblocks));
        s->block_cap = nblocks;
    }
    SamplerArgs a;
    sampler_args(s, seed, first, n, uid_dev, iid_dev, &a);
    s->strat_next = first + n; s->strat_seed = seed;
    return orx_launch_sample_stratified(s->ctx, a, pos_ratio, label_dev, s->d_blockcnt, s->d_blockbase, s->d_counter);
}

extern "C" int orx_sampler_per_pos_stratified(orx_sampler* s, uint64_t seed, int64_t first, int64_t n, double pos_ratio,
                                              int32_t* uid_dev, int32_t* iid_dev, float* label_dev) {
    ORX_ARG(s && uid_dev && iid_dev && label_dev && first >= 0 && n >= 0, "orx_sampler_per_pos_stratified: bad argument");
    ORX_ARG(pos_ratio > 0.0 && pos_ratio <= 1.0, "orx_sampler_per_pos_stratified: pos_ratio must lie in (0, 1]");
    // dataset.py:40 computes int((1 - pos_ratio) / pos_ratio) in Python doubles; in fp32 the quotient lands on the other side of
    // an integer for common ratios (0.05: 19 instead of 18; 1/3: 1 instead of 2), i.e. another group size than the reference's
    const int nneg = (int)((1.0 - pos_ratio) / pos_ratio);
    ORX_ARG(nneg + 1 <= s->total_items, "orx_sampler_per_pos_stratified: %d negatives per positive need more than %lld items "
            "(random.sample would raise ValueError)", nneg, (long long)s->total_items);
    ORX_HIP(hipSetDevice(s->ctx->device));
    SamplerArgs a;
    sampler_args(s, seed, first, n, uid_dev, iid_dev, &a);
    return orx_launch_sample_perpos(s->ctx, a, nneg, label_dev);
}
