/* openrec_hip.h -- C ABI of the MI355X-native OpenRec training hot path.
 *
 * The reference (ylongqi/openrec, /root/reference) has NO FFI boundary: the
 * path is reached through Python classes on top of TensorFlow.  This header is
 * the boundary a maintainer would bind instead (ctypes stub in INTEGRATION.md);
 * every entry point cites the reference interface it replaces (paths relative
 * to /root/reference).
 *
 * Conventions
 *   - every function returns int: 0 = ORX_OK, < 0 = error; the message is
 *     available from orx_last_error() (thread local).  Nothing aborts, no C++
 *     exception crosses the boundary.
 *   - handles (orx_ctx / orx_table / orx_opt) are opaque and owned by the
 *     library; pointer arguments are caller-owned and only need to stay valid
 *     for the duration of the call.
 *   - tables are fp32, row-major [rows, dim], resident in HBM for their whole
 *     life (Keras Embedding weight layout, latent_factor.py:12-15).
 *   - ids are int32 (openrec/tf2/data/dataset.py:113-115).  With
 *     ORX_IDS_DEVICE the id pointers are device pointers, otherwise host.
 *   - calls on one context are serialized on that context's HIP stream; a
 *     context is not thread-safe, distinct contexts are independent.
 *   - there is no CPU fallback: every call fails with ORX_ERR_HIP if no
 *     gfx950 device is usable.
 */
#ifndef OPENREC_HIP_H
#define OPENREC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orx_ctx orx_ctx;
typedef struct orx_table orx_table;
typedef struct orx_opt orx_opt;

enum orx_status {
    ORX_OK = 0,
    ORX_ERR_ARG = -1,    /* bad argument / handle / shape                      */
    ORX_ERR_HIP = -2,    /* HIP runtime error (message has hipGetErrorString)  */
    ORX_ERR_OOM = -3,    /* device allocation failed                           */
    ORX_ERR_INDEX = -4,  /* id out of range (TF CPU gather raises, bpr.py:23)  */
    ORX_ERR_STATE = -5   /* call sequence error                                */
};

/* tensorflow.keras.optimizers.* used by tf2_examples/bpr_citeulike.py:31.
 * ORX_ADAM is the TF-2.0 sparse rule: EVERY row of a table decays (m, v) and moves (var) at every step.
 * orx_pairwise_step (float4 dims, no censor), orx_dlrm_step and orx_apply_rows apply it lazily: a row takes its
 * gradient-free steps when it is next referenced, and every entry point that observes a table or its slots
 * (read / gather / device_ptr / slot_read / inference / another optimizer / orx_opt_destroy) first brings
 * the rows it exposes up to date, so the result is the dense rule's at every observation.
 * Tables over caller-owned memory (orx_table_wrap) always take the literal whole-table sweeps, as does
 * everything with ORX_ADAM_DENSE=1 in the environment.  A pointer obtained from orx_table_device_ptr is
 * current when returned; after further Adam steps, ask again. */
enum orx_opt_kind { ORX_SGD = 0, ORX_ADAGRAD = 1, ORX_ADAM = 2 };

/* pairwise recommenders: recommenders/bpr.py:5, recommenders/ucml.py:5 */
enum orx_pair_model { ORX_BPR = 0, ORX_UCML = 1 };
/* pointwise recommenders: recommenders/gmf.py:5, recommenders/wrmf.py:5 */
enum orx_point_model { ORX_GMF = 0, ORX_WRMF = 1 };

enum orx_flags {
    ORX_IDS_DEVICE = 1,   /* id / label pointers are device pointers            */
    ORX_HOGWILD = 2,      /* skip duplicate handling: one in-place racy pass
                             (NOT reference semantics; speed comparisons only)  */
    ORX_NO_L2 = 4,        /* objective = loss only (tape over `loss` alone)
                             instead of the example's (loss, l2_loss) tuple     */
    ORX_CENSOR = 8,       /* orx_pairwise_step: after every step, UCML.censor_vec
                             (ucml.py:44-48) on that step's ids: users, then pos
                             items, then neg items, min_norm 0.1                */
    ORX_POINT_SIGMOID = 16 /* orx_pointwise_step / _loss with ORX_WRMF: PointwiseMSELoss(sigmoid=True),
                             the prediction goes through a sigmoid before the weighted squared
                             error (modules/pointwise_mse_loss.py:24-25)        */
};

/* kernels whose device time can be sampled with orx_prof_* */
enum orx_kernel_id {
    ORX_K_DEDUP = 0,      /* duplicate-reference detection on the id arrays     */
    ORX_K_FUSED = 1,      /* fused gather-score-loss-grad-update (dominant)     */
    ORX_K_REDUCE = 2,     /* per-step loss reduction (once per call)            */
    ORX_K_SWEEP = 3,      /* Adam dense-decay sweep                             */
    ORX_K_CENSOR = 4,
    ORX_K_POINT = 5,      /* fused pointwise (GMF/WRMF) step                    */
    ORX_K_DUPAPPLY = 6,   /* optimizer apply of the duplicate rows of a step    */
    ORX_K_GEMM = 7,       /* DLRM MLP products (fp32 or fp16 MFMA)              */
    ORX_K_NUM = 8
};

int orx_version(void);
const char* orx_last_error(void);

/* ---- context ----------------------------------------------------------
 * device: HIP ordinal.  stream: an existing hipStream_t to enqueue on (e.g.
 * torch's current stream) or NULL to create a private one. */
int orx_ctx_create(int device, void* stream, orx_ctx** out);
int orx_ctx_destroy(orx_ctx* ctx);
int orx_synchronize(orx_ctx* ctx);
/* Ordering contract for device buffers handed over with ORX_IDS_DEVICE (ids, labels, gradient rows ...): the library
 * reads them on the context's stream.  A buffer produced on ANOTHER stream (a framework's current stream) must be
 * complete there first: orx_ctx_wait_stream makes every later call on `ctx` wait for the work `producer_stream`
 * (a hipStream_t; NULL = the legacy default stream) holds now -- an event, no host synchronisation.  The reference
 * has no such notion: TensorFlow orders its own ops (tf2_examples/bpr_citeulike.py:33-39 runs inside tf.function). */
int orx_ctx_wait_stream(orx_ctx* ctx, void* producer_stream);
/* raises the sticky "id out of range" condition recorded by the kernels
 * (returns ORX_ERR_INDEX once, then clears it).  Synchronizes. */
int orx_check_index_error(orx_ctx* ctx);

/* ---- tables: replace LatentFactor.__init__/variables (latent_factor.py:6-15)
 * orx_table_wrap adopts caller-owned device memory (e.g. a torch tensor). */
int orx_table_create(orx_ctx* ctx, int64_t rows, int32_t dim, orx_table** out);
int orx_table_wrap(orx_ctx* ctx, void* device_ptr, int64_t rows, int32_t dim, orx_table** out);
int orx_table_destroy(orx_table* t);
int64_t orx_table_rows(const orx_table* t);
int32_t orx_table_dim(const orx_table* t);
void* orx_table_device_ptr(const orx_table* t);
/* Keras 'uniform' initializer = U(-0.05, 0.05) / 'zeros' (latent_factor.py:8-11) */
int orx_table_init_uniform(orx_table* t, float lo, float hi, uint64_t seed);
int orx_table_fill(orx_table* t, float value);
/* checkpoint / parity access: rows [row0, row0+nrows) <-> host fp32 */
int orx_table_read(orx_table* t, int64_t row0, int64_t nrows, float* host_dst);
int orx_table_write(orx_table* t, int64_t row0, int64_t nrows, const float* host_src);
/* LatentFactor.__call__ = Embedding gather (bpr.py:23-27): out[k,:] = W[ids[k],:].
 * ids/out are host unless ORX_IDS_DEVICE (then both are device pointers). */
int orx_table_gather(orx_table* t, const int32_t* ids, int64_t n, float* out, int flags);
/* LatentFactor.censor (latent_factor.py:17-23): for each DISTINCT id,
 * W[i] <- W[i] / max(||W[i]||_2, min_norm). */
int orx_table_censor(orx_table* t, const int32_t* ids, int64_t n, float min_norm, int flags);

/* ---- optimizers: replace optimizers.{SGD,Adagrad,Adam}.apply_gradients
 * (tf2_examples/bpr_citeulike.py:31,38) with TF-2.0 sparse-apply semantics.
 *   SGD:     p0..p2 unused
 *   ADAGRAD: p0 = initial_accumulator_value, p1 = epsilon
 *   ADAM:    p0 = beta_1, p1 = beta_2, p2 = epsilon   (dense-decay sparse apply) */
int orx_opt_create(orx_ctx* ctx, int kind, float lr, float p0, float p1, float p2, orx_opt** out);
int orx_opt_destroy(orx_opt* opt);
int orx_opt_set_lr(orx_opt* opt, float lr);
/* the optimizer's step counter (Keras `optimizer.iterations`: Adam's bias correction depends on it);
 * a checkpoint saves it next to the slots, a resume sets it before the next step.  The step entry points
 * advance it themselves; a host that drives Adam through orx_apply_rows advances it by one per step with
 * orx_opt_advance after the step's gathers and before its applies.
 *   set_step : a jump of the counter (resume, Keras `iterations.assign`).  Every table that is lazily applied
 *              under `opt` is first finished under the old counter; no row is replayed across the jump.
 *   advance  : "the next apply_gradients of `opt` updates exactly these tables" (tf2_examples/bpr_citeulike.py:38 passes
 *              the model's variables): counter += 1.  Keras' Adam touches only the variables it is handed, so any
 *              OTHER table lazily applied under `opt` (a second model sharing the optimizer) is finished first and
 *              takes no decay step for this one.  The step entry points (orx_pairwise_step, orx_pointwise_step,
 *              orx_dlrm_step) do the same for their own tables.  n_tables < 0: the step is over every table the
 *              optimizer holds (an engine with an optimizer of its own): the counter advances, nothing is finished. */
int orx_opt_get_step(orx_opt* opt, int64_t* step_out);
int orx_opt_set_step(orx_opt* opt, int64_t step);
int orx_opt_advance(orx_opt* opt, orx_table* const* tables, int32_t n_tables);
/* read/write an optimizer slot of a table (checkpointing, parity):
 * slot 0 = Adagrad accumulator / Adam m, slot 1 = Adam v. */
int orx_opt_slot_read(orx_opt* opt, orx_table* t, int slot, int64_t row0, int64_t nrows, float* host_dst);
int orx_opt_slot_write(orx_opt* opt, orx_table* t, int slot, int64_t row0, int64_t nrows, const float* host_src);

/* ---- the hot path ------------------------------------------------------
 * K consecutive train steps of BPR.call / UCML.call + tape.gradient((loss,
 * l2_loss)) + optimizer.apply_gradients  (bpr.py:21-37, ucml.py:21-42,
 * pairwise_log_loss.py:15-34, tf2_examples/bpr_citeulike.py:33-39).
 *   user/item/bias : tables [NU,D], [NI,D], [NI,1]
 *   uid/pid/nid    : int32, step s uses elements [s*id_stride, s*id_stride+B)
 *   margin         : UCML margin (ucml.py:7); ignored for BPR
 *   loss_out/l2_out: host float[K] or NULL (NULL = fully asynchronous call)
 * Gradients of every step are taken on the pre-step tables (snapshot
 * semantics), duplicates handled as TF does (SGD accumulates every
 * occurrence; Adagrad/Adam sum duplicates first). */
int orx_pairwise_step(orx_ctx* ctx, int model, orx_opt* opt,
                      orx_table* user, orx_table* item, orx_table* bias,
                      const int32_t* uid, const int32_t* pid, const int32_t* nid,
                      int64_t K, int64_t B, int64_t id_stride, float margin, int flags,
                      float* loss_out, float* l2_out);

/* Pre-size every per-call scratch buffer for calls of up to K steps of B triplets on these tables
 * (duplicate-detection outputs, rewritten ids, loss partials, scratch tables), so that a later
 * orx_pairwise_step performs no device allocation.  Optional: buffers also grow on demand. */
int orx_pairwise_reserve(orx_ctx* ctx, orx_opt* opt, orx_table* user, orx_table* item, orx_table* bias,
                         int64_t K, int64_t B);

/* Forward only: (loss, l2_loss) of one batch without touching the tables
 * (BPR.call / UCML.call outside a tape). */
int orx_pairwise_loss(orx_ctx* ctx, int model,
                      orx_table* user, orx_table* item, orx_table* bias,
                      const int32_t* uid, const int32_t* pid, const int32_t* nid,
                      int64_t B, float margin, int flags, float* loss_out, float* l2_out);

/* K train steps of GMF.call / WRMF.call (gmf.py:22-34, wrmf.py:21-34,
 * pointwise_mse_loss.py:18-31).  w: GMF's Dense(1, use_bias=False) kernel as a
 * [D,1] table (NULL for WRMF).  a, b: WRMF confidence weights (wrmf.py:7). */
int orx_pointwise_step(orx_ctx* ctx, int model, orx_opt* opt,
                       orx_table* user, orx_table* item, orx_table* bias, orx_table* w,
                       const int32_t* uid, const int32_t* iid, const float* label,
                       int64_t K, int64_t B, int64_t id_stride, float a, float b, int flags,
                       float* loss_out, float* l2_out);

/* Forward only for the pointwise models (GMF.call / WRMF.call outside a tape). */
int orx_pointwise_loss(orx_ctx* ctx, int model,
                       orx_table* user, orx_table* item, orx_table* bias, orx_table* w,
                       const int32_t* uid, const int32_t* iid, const float* label,
                       int64_t B, float a, float b, int flags, float* loss_out, float* l2_out);

/* Recommender.inference (bpr.py:39-43, wrmf.py:36-40: kind 0 = U[uid] . V^T + b;
 * ucml.py:50-53: kind 1 = -||U[uid] - V||^2 + b; gmf.py:36-41: kind 2 = sum_d w_d u_d v_d + b).
 * uid: host int32[n]; out: host float[n * item_rows], row-major [n, item_rows]. */
int orx_score_all_items(orx_ctx* ctx, int kind, orx_table* user, orx_table* item, orx_table* bias, orx_table* w,
                        const int32_t* uid, int64_t n, float* out);

/* orx_score_all_items with the scores left on the device: out_dev device float[n * item_rows]. */
int orx_score_all_items_device(orx_ctx* ctx, int kind, orx_table* user, orx_table* item, orx_table* bias, orx_table* w,
                               const int32_t* uid, int64_t n, float* out_dev);

/* Ranking metrics of the evaluation step (openrec/tf2/metrics/ranking_metrics.py:8-69; eval_step in
 * tf2_examples/bpr_citeulike.py:41-46).  For each of n users: scores over ALL items (computed on the
 * device like orx_score_all_items when pred == NULL, else taken from the host array pred[n*items]),
 * pos_mask / excl_mask host uint8 [n*items]; at: host float[nat] cut-offs (nat <= 16).
 * Outputs (host): auc[n], ndcg[n*nat], recall[n*nat]; any of them may be NULL. */
int orx_rank_metrics(orx_ctx* ctx, int kind, orx_table* user, orx_table* item, orx_table* bias, orx_table* w,
                     const int32_t* uid, const float* pred, const uint8_t* pos_mask, const uint8_t* excl_mask,
                     int64_t n, int64_t items, const float* at, int32_t nat,
                     float* auc, float* ndcg, float* recall);

/* The same metrics with the two masks as CSR item lists over the call's n users (the form the reference's Dataset holds them
 * in before openrec/tf2/data/dataset.py:60-82 _evaluation_generator densifies them): pos_ptr / excl_ptr host int64[n + 1]
 * starting at 0, pos_items / excl_items host int32, each user's list a set (a repeated positive is an error, an id outside
 * [0, items) an index error).  2 x n x items mask bytes become the lists; results equal orx_rank_metrics on the dense masks.
 * pred_on_device != 0: pred is device memory (what orx_score_all_items_device wrote). */
int orx_rank_metrics_csr(orx_ctx* ctx, int kind, orx_table* user, orx_table* item, orx_table* bias, orx_table* w,
                         const int32_t* uid, const float* pred, int32_t pred_on_device, int64_t n, int64_t items,
                         const int64_t* pos_ptr, const int32_t* pos_items, const int64_t* excl_ptr, const int32_t* excl_items,
                         const float* at, int32_t nat, float* auc, float* ndcg, float* recall);

/* ---- on-device triplet sampler: the producer of the train step
 * (openrec/tf2/data/dataset.py:7-16 _pairwise_generator; utils.py:82-87, 102-116).
 * rec_user / rec_item: the n_records interaction records; csr_ptr[total_users + 1] / csr_items: the
 * positives of every user, item ids sorted ascending inside a row (host arrays, copied to the device).
 * orx_sampler_pairwise writes samples [first, first + n) of the stream `seed` to DEVICE arrays:
 * every record is the positive exactly once per epoch of n_records samples (keyed permutation);
 * negatives are uniform over the items that are not positives of the user.  Sample g depends only
 * on (seed, g), never on the launch shape. */
typedef struct orx_sampler orx_sampler;
int orx_sampler_create(orx_ctx* ctx, const int32_t* rec_user, const int32_t* rec_item, int64_t n_records,
                       const int64_t* csr_ptr, const int32_t* csr_items, int64_t total_users, int64_t total_items,
                       orx_sampler** out);
int orx_sampler_destroy(orx_sampler* s);
int orx_sampler_pairwise(orx_sampler* s, uint64_t seed, int64_t first, int64_t n,
                         int32_t* uid_dev, int32_t* pid_dev, int32_t* nid_dev);
/* The pointwise producers of GMF / WRMF (dataset.py:18-36 _stratified_pointwise_generator, :38-58
 * _per_pos_stratified_pointwise_generator): samples [first, first + n) as (user, item, label) DEVICE arrays.
 *   stratified         : with probability pos_ratio the next record of the shuffled epoch (label 1), otherwise a uniform
 *                        (user, item) pair that is not a positive (label 0).  Sequential stream, like the reference's
 *                        generator: a call continues at the sample the previous one (same seed) stopped at; first = 0 restarts.
 *   per_pos_stratified : groups of 1 + int((1 - pos_ratio) / pos_ratio): a record (label 1), then that many DISTINCT items
 *                        other than the record's (label 0; not checked against the user's other positives, as in the
 *                        reference).  Counter-based: any window can be regenerated. */
int orx_sampler_stratified(orx_sampler* s, uint64_t seed, int64_t first, int64_t n, float pos_ratio,
                           int32_t* uid_dev, int32_t* iid_dev, float* label_dev);
int orx_sampler_per_pos_stratified(orx_sampler* s, uint64_t seed, int64_t first, int64_t n, double pos_ratio,
                                   int32_t* uid_dev, int32_t* iid_dev, float* label_dev);

/* ---- DLRM (recommenders/dlrm.py:6-100, modules/multi_layer_perceptron.py:5-18,
 * modules/second_order_feature_interaction.py:4-34; train step as in
 * tf2_examples/dlrm_criteo.py:42-48).  The n_emb embedding tables (all of dim
 * m_spa) live in ONE combined table, table f starting at row sum(ln_emb[0:f]).
 * The bottom MLP must end at m_spa (its output is stacked with the embeddings). */
typedef struct orx_dlrm orx_dlrm;
enum orx_dlrm_flags {
    ORX_DLRM_INTERACT_ITSELF = 1,   /* arch_interaction_itself                           */
    ORX_DLRM_SIGMOID_BOT = 2,       /* sigmoid_bot (else relu), dlrm.py:34-35            */
    ORX_DLRM_SIGMOID_TOP = 4,       /* sigmoid_top (else relu), dlrm.py:36-37            */
    ORX_DLRM_LOSS_BCE = 8,          /* loss_func='bce' (else 'mse'), dlrm.py:52-55       */
    ORX_DLRM_FP16_MLP = 32,         /* performance mode: the MLP products run on fp16 MFMA
                                       (operands rounded to fp16, fp32 accumulate / storage);
                                       NOT within the 1e-5 parity tolerance                 */
    ORX_DLRM_REFERENCE_COMPAT = 16, /* reproduce second_order_feature_interaction.py:21-32
                                       literally: lower triangle kept, upper selected -> the
                                       interaction output is 0; without this
                                       flag the strictly lower triangle is used             */
    ORX_DLRM_NO_EMB = 64            /* the model owns no embedding table: the rows come from
                                       outside (row-sharded tables, orx_dlrm_grads)         */
};
enum orx_dlrm_param_kind { ORX_DLRM_EMB = 0, ORX_DLRM_BOT_W = 1, ORX_DLRM_BOT_B = 2, ORX_DLRM_TOP_W = 3, ORX_DLRM_TOP_B = 4 };

int orx_dlrm_create(orx_ctx* ctx, int32_t m_spa, int32_t n_emb, const int64_t* ln_emb,
                    int32_t n_bot, const int32_t* ln_bot, int32_t n_top, const int32_t* ln_top,
                    int32_t dense_dim, int flags, float loss_threshold, uint64_t seed, orx_dlrm** out);
int orx_dlrm_destroy(orx_dlrm* m);
/* borrowed handle of a parameter: the combined embedding table, or Dense layer `layer`'s
 * kernel [in, out] / bias [1, out] (Keras layout).  Owned by the model. */
int orx_dlrm_param(orx_dlrm* m, int kind, int layer, orx_table** out);
/* K train steps (DLRM.call + tape.gradient + apply_gradients).  dense [K*B, dense_dim] fp32,
 * sparse [K*B, n_emb] int32 (id within its own table), label [K*B] fp32; host pointers unless
 * ORX_IDS_DEVICE.  loss_out: host float[K] or NULL. */
int orx_dlrm_step(orx_dlrm* m, orx_opt* opt, const float* dense, const int32_t* sparse, const float* label,
                  int64_t K, int64_t B, int flags, float* loss_out);
/* DLRM.inference (dlrm.py:76-100): pred_out host/device float[B] */
int orx_dlrm_inference(orx_dlrm* m, const float* dense, const int32_t* sparse, int64_t B, int flags, float* pred_out);
/* Hybrid-parallel train step (embedding tables row-sharded across ranks, MLPs data-parallel;
 * no reference equivalent).  All pointers are DEVICE pointers.
 *   orx_dlrm_grads       : forward + backward of this rank's B samples with the embedding rows
 *                          handed in (emb_rows [B, n_emb, m_spa], already exchanged); the loss
 *                          mean runs over global_B samples.  Writes d loss / d emb_rows to
 *                          emb_grads [B, n_emb, m_spa], leaves the dense gradients inside the
 *                          model and adds this rank's part of the loss to loss_accum[0] (double).
 *   orx_dlrm_dense_count : number of fp32 dense parameters (all kernels and biases)
 *   orx_dlrm_dense_pack  : copy the dense gradients into flat[count] (for ONE all-reduce)
 *   orx_dlrm_dense_apply : optimizer step of every dense parameter with the gradients in
 *                          flat[count] (after the all-reduce); advances the step counter. */
int orx_dlrm_grads(orx_dlrm* m, const float* dense, const float* emb_rows, const float* label, int64_t B,
                   int64_t global_B, float* emb_grads, double* loss_accum);
/*   orx_dlrm_grads_indirect : the same with the embedding rows read IN PLACE from the buffer the exchange filled (rows
 *                          [n_rows][m_spa], idx[b (n_emb + 1) + f] = row of lookup f of sample b, slot n_emb unused) and the
 *                          gradient of each lookup written to row idx[...] of grads_dst, the buffer that travels back: no
 *                          reordering passes.  orx_dlrm_direct_ok: 1 if this model's shapes allow it. */
int orx_dlrm_direct_ok(orx_dlrm* m);
/* The DLRM modules called on PLAIN ARRAYS (outside the composition dlrm.py:76-100, which is orx_dlrm_step / orx_dlrm_inference):
 *   orx_mlp_forward      openrec/tf2/modules/multi_layer_perceptron.py:5-18: y = act_L(... act_1(x W_1 + b_1) ...); kernels[l] is a
 *                        table [in_l, out_l] (Keras Dense layout), biases[l] a table [1, out_l] or NULL (biases may be NULL altogether),
 *                        acts[l] 0 none / 1 relu / 2 sigmoid; x [B, in_dim] -> y_out [B, out_L]; exact fp32 products.
 *   orx_interact_forward openrec/tf2/modules/second_order_feature_interaction.py:12-34: z [B, F, d] = the F inputs stacked on axis 1,
 *                        out [B, P], P = F (F - 1) / 2 (self_interaction: F (F + 1) / 2), the selected elements of Z Z^T in
 *                        boolean_mask order; reference_compat != 0 reproduces the reference's text (zeros: SURVEY.md E.1).
 * flags & ORX_IDS_DEVICE: x / z and the output are device pointers (asynchronous on the context's stream); otherwise host pointers. */
int orx_mlp_forward(orx_ctx* ctx, int32_t n_layers, orx_table* const* kernels, orx_table* const* biases, const int32_t* acts,
                    const float* x, int64_t B, int32_t in_dim, int flags, float* y_out);
int orx_interact_forward(orx_ctx* ctx, const float* z, int64_t B, int32_t F, int32_t d, int self_interaction, int reference_compat,
                         int flags, float* out);
int orx_dlrm_grads_indirect(orx_dlrm* m, const float* dense, const float* rows, int64_t n_rows, const int32_t* idx,
                            const float* label, int64_t B, int64_t global_B, float* grads_dst, double* loss_accum);
int orx_dlrm_dense_count(orx_dlrm* m, int64_t* count);
int orx_dlrm_dense_pack(orx_dlrm* m, float* flat);
int orx_dlrm_dense_apply(orx_dlrm* m, orx_opt* opt, const float* flat);

/* ---- sharded building blocks (row-wise sharded tables, one rank per GPU;
 * the exchange itself is RCCL all-to-all driven by the host, see
 * openrec_amd/sharded.py).  No reference equivalent (the reference is single
 * process).  All pointers are DEVICE pointers; ids are LOCAL row ids, an id < 0
 * marks a padding slot and is skipped.
 *   gather_rows : out[k, 0:D] = W[ids[k]], out[k, D] = bias[ids[k]] (bias may be NULL)
 *   pair_grads  : from gathered rows [T, row_stride] (bias at column D) compute the
 *                 per-occurrence gradients of J (bias gradient at column D of gp/gn)
 *                 and ADD this rank's (loss, l2_loss) contribution to loss_l2_accum[2];
 *                 the loss mean is taken over B_global; triplet k is live iff
 *                 valid == NULL or valid[k] >= 0
 *   apply_rows  : optimizer sparse apply of per-occurrence gradient rows
 *                 (SGD: every occurrence accumulated; Adagrad / Adam: duplicates summed first;
 *                 Adam takes the step the optimizer's counter stands at: orx_dlrm_dense_apply
 *                 advances it once per step, otherwise the host does with orx_opt_advance)
 */
int orx_gather_rows(orx_ctx* ctx, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                    float* out, int64_t out_stride);
int orx_pair_grads(orx_ctx* ctx, int model, int32_t D,
                   const float* u_rows, const float* p_rows, const float* n_rows, int64_t row_stride,
                   const int32_t* valid, int64_t T, int64_t B_global, float margin, int flags,
                   float* gu, float* gp, float* gn, int64_t g_stride, double* loss_l2_accum);
int orx_apply_rows(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias,
                   const int32_t* ids, int64_t n, const float* grads, int64_t g_stride);
/* K-step plans know every id list up front: duplicate flags of K lists in one launch
 * (dflag[k*n + i] = 1 iff row ids[k*id_stride + i] occurs more than once in list k), and the SGD
 * apply that uses them: rows referenced once are plain read-modify-writes, only duplicated rows
 * take atomics.  Same result as orx_apply_rows. */
int orx_rows_dupflags(orx_ctx* ctx, int64_t rows, const int32_t* ids, int64_t K, int64_t n, int64_t id_stride,
                      unsigned char* dflag);
int orx_apply_rows_flagged(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                           const float* grads, int64_t g_stride, const unsigned char* dflag);

/* Device-side exchange plan of the sharded step (openrec_amd/sharded.py): fixed-capacity
 * buckets, no host synchronization.  Row r of a table lives on rank r % world at local
 * index r / world.  `overflow` is a device int set to 1 when a bucket was full.
 *   shard_route   : triplets -> send[world*cap][3] (bucket = uid % world); fills unused slots with -1
 *   shard_request : received triplets [T][3] (u < 0 = empty) -> item-id requests
 *                   send_ids[world*cap] (bucket = id % world, -1 unused), slot[2T] (where the p / n
 *                   request of triplet t went, -1 if none) and u_loc[T] (local user row or -1)
 *   shard_localize: out[i] = ids[i] / world (or -1)
 *   shard_grads   : user rows from the local shard + received item rows [world*cap][DS] ->
 *                   gu[T][D] and the item-row gradients written in place into send_g at the slots
 *                   of their requests; adds the (loss, l2_loss) contribution to loss_l2_accum[2] */
int orx_shard_route(orx_ctx* ctx, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t B,
                    int64_t users_global, int64_t items_global, int32_t world, int32_t cap,
                    int32_t* send, int32_t* counters, int32_t* overflow);
int orx_shard_request(orx_ctx* ctx, const int32_t* trip, int64_t T, int32_t world, int32_t cap,
                      int32_t* send_ids, int32_t* slot, int32_t* u_loc, int32_t* counters, int32_t* overflow);
/* the same two plans for K steps in one launch each (the plan depends on the ids alone): uid/pid/nid
 * [K][id_stride], send [K][world*cap][3], trip [K][T][3], send_ids [K][world*cap], slot [K][2T],
 * u_loc [K][T], counters [K][world] */
int orx_shard_route_steps(orx_ctx* ctx, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t K, int64_t B,
                          int64_t id_stride, int64_t users_global, int64_t items_global, int32_t world, int32_t cap,
                          int32_t* send, int32_t* counters, int32_t* overflow);
int orx_shard_request_steps(orx_ctx* ctx, const int32_t* trip, int64_t K, int64_t T, int32_t world, int32_t cap,
                            int32_t* send_ids, int32_t* slot, int32_t* u_loc, int32_t* counters, int32_t* overflow);
int orx_shard_localize(orx_ctx* ctx, const int32_t* ids, int64_t n, int32_t world, int32_t* out);
/* generic request plan (hybrid-parallel DLRM lookups): every id >= 0 claims a slot in the bucket of its
 * owner rank (id % world): send_ids[world*cap] (-1 = empty), slot[n] = position in send_ids or -1;
 * counters[world] scratch, overflow[1] sticky flag (a bucket was full: that id is dropped) */
int orx_shard_bucket(orx_ctx* ctx, const int32_t* ids, int64_t n, int32_t world, int32_t cap,
                     int32_t* send_ids, int32_t* slot, int32_t* counters, int32_t* overflow);
/* orx_shard_request_steps with PER-DESTINATION DEDUP: an item that several references of a list ask for claims one slot of its
 * owner's bucket (its row travels once, the sum of the references' gradients travels back).  The distinct items of an owner fill
 * its bucket in ascending row order (a stable sort of the references by (owner, row): deterministic).  Outputs besides
 * send_ids / slot / u_loc: dupref[K][2T] = 1 on the references that share their slot; sorted_out [K][2T] x 8 bytes (the sorted
 * (key, reference) list), seglist [K][T] x 8 bytes and segcount [K] (the shared slots): opaque, handed back to orx_shard_grads.
 * items_global: rows of the global item table. */
int orx_shard_request_dedup_steps(orx_ctx* ctx, const int32_t* trip, int64_t K, int64_t T, int32_t world, int32_t cap,
                                  int64_t items_global, int32_t* send_ids, int32_t* slot, int32_t* u_loc, uint8_t* dupref,
                                  void* sorted_out, void* seglist, int32_t* segcount, int32_t* overflow);
/* dupref == NULL: every reference has a slot of its own.  Otherwise the list's outputs of orx_shard_request_dedup_steps and a side
 * buffer gdup [2T][row_stride]: references that share a slot leave their gradients there and a second launch writes the slot's
 * SUM, taken in reference order (no atomics: the result does not depend on timing). */
int orx_shard_grads(orx_ctx* ctx, int model, orx_table* user, const float* rows_in, const int32_t* u_loc,
                    const int32_t* slot, const uint8_t* dupref, const void* sorted, const void* seglist, const int32_t* segcount,
                    float* gdup, int64_t T, int64_t row_stride, int64_t B_global, float margin, int flags,
                    float* gu, float* send_g, double* loss_l2_accum);
/* orx_shard_grads with SGD's apply of the local user rows folded in (row-sharded BPR / UCML step, phases 4 + 5 of
 * openrec_amd/sharded.py; the reference's single-process apply is tf2_examples/bpr_citeulike.py:38): a user row referenced
 * once in the step is updated in place; the references of a duplicated row leave gu[t] and u_apply[t] = local row for
 * orx_apply_rows_flagged (u_apply[t] = -1 elsewhere).  dup_u[T]: orx_rows_dupflags of u_loc. */
int orx_shard_grads_sgd(orx_ctx* ctx, int model, orx_opt* opt, orx_table* user, const float* rows_in, const int32_t* u_loc,
                        const int32_t* slot, const uint8_t* dupref, const void* sorted, const void* seglist, const int32_t* segcount,
                        float* gdup, const uint8_t* dup_u, int64_t T, int64_t row_stride, int64_t B_global,
                        float margin, int flags, float* gu, int32_t* u_apply, float* send_g, double* loss_l2_accum);

/* ---- the row-sharded pairwise step as one host call (the single-process step it shards is
 * tf2_examples/bpr_citeulike.py:27-39).  Row r of every table lives on rank r % world at local index r / world; the ranks'
 * batches together are ONE global batch (loss mean over world * B).  An orx_comm wraps an RCCL communicator (librccl.so is
 * loaded at run time by orx_comm_create; the exchanges are ncclSend / ncclRecv groups on the context's stream) and owns the
 * exchange buffers of the engine.
 *   orx_comm_unique_id : rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *   orx_comm_create    : collective over the ranks (ncclCommInitRank).  unique_id == NULL with world 1: a one-rank
 *                        communicator without RCCL (every exchange is the identity)
 *   orx_sharded_pairwise_steps : K steps of this rank's uid / pid / nid [K][id_stride] (DEVICE int32, B ids per step) on its
 *                        shards U / V / b.  The exchange plan (triplets -> owner of the user row, item ids -> owners) is made
 *                        for plan_chunk steps at a time with one exchange per phase; a step is then gather -> exchange (rows)
 *                        -> gradients -> apply users -> exchange (gradients) -> apply items.  Buckets have a fixed capacity
 *                        (mean * slack + 6 sigma + 16, orx_sharded_caps): `overflow` (device int, sticky) reports a full one.
 *                        loss_l2_accum: device double[2], this rank's (sum of loss, sum of l2_loss) contributions are added.
 *                        flags: ORX_NO_L2; ORX_SHARD_OVERLAP (needs id_stride == B, B even, a communicator with RCCL): every
 *                        step is cut into two half-batches whose exchanges run on a second stream beside the other half's
 *                        kernels.  Same results as the per-phase entry points driven by openrec_amd/sharded.py. */
#define ORX_COMM_ID_BYTES 128
#define ORX_SHARD_OVERLAP 0x100
#define ORX_SHARD_NO_DEDUP 0x200   /* every item reference claims a slot of its own (orx_shard_request_steps) */
#define ORX_SHARD_DEDUP 0x400      /* per-destination dedup of the item requests (orx_shard_request_dedup_steps); neither flag: on
                                    * when a list's item references number at least half the items (most slots are then shared) */
typedef struct orx_comm orx_comm;
int orx_comm_unique_id(void* id_out);
int orx_comm_create(orx_ctx* ctx, const void* unique_id, int32_t rank, int32_t world, orx_comm** out);
int orx_comm_destroy(orx_comm* comm);
/* An in-process group of `world` ranks that share ONE device -- one host thread and one context per rank, exchanges meeting at
 * a host barrier, blocks copied by kernels: the engine's complete exchange schedule with world > 1 where there is no second GPU
 * (tests/test_gpu_shard_engine.py).  orx_vgroup_abort releases ranks waiting for one that failed. */
typedef struct orx_vgroup orx_vgroup;
int orx_vgroup_create(int32_t world, orx_vgroup** out);
int orx_vgroup_abort(orx_vgroup* group);
int orx_vgroup_destroy(orx_vgroup* group);
int orx_comm_create_virtual(orx_ctx* ctx, orx_vgroup* group, int32_t rank, orx_comm** out);
int orx_comm_rank(orx_comm* comm);
int orx_comm_world(orx_comm* comm);
/* Measurement of the exchanges (bench.py --gpus N: phases_us / link_GBps).
 *   orx_comm_stats(comm, 1, NULL): start counting, counters at zero (HIP events around every exchange from now on);
 *   orx_comm_stats(comm, 0, out) : stop; out[0] = exchanges, out[1] = bytes this rank SENT to other ranks, out[2] = their device
 *                                  time in ms (sum over the exchanges, each timed on the stream it ran on), out[3] = bytes of the
 *                                  rank's own blocks (device copies, not on the wire).
 *   orx_comm_ping(comm, bytes, reps, out): `reps` all-to-alls of `bytes` per peer (ncclSend / ncclRecv to every other rank at once,
 *                                  as the engine's exchanges do) on fresh buffers; out[0] = GB/s this rank sent (all links together),
 *                                  out[1] = GB/s per link, out[2] = microseconds per all-to-all.  Collective: every rank calls it.
 *                                  A one-rank RCCL communicator measures the device copy of its own block; one made WITHOUT RCCL
 *                                  (every exchange is the identity) has nothing to time: out = {0, 0, 0}. */
int orx_comm_stats(orx_comm* comm, int start, double* out4);
int orx_comm_ping(orx_comm* comm, int64_t bytes, int32_t reps, double* out3);
int orx_sharded_caps(int64_t B, int32_t world, float slack, int64_t* cap1, int64_t* cap2);
/* the regrouping the engine applies around the plan's exchanges: src [K][world][words] -> dst [world][K][words] 4-byte words
 * (back != 0: the inverse); device pointers */
int orx_shard_regroup(orx_ctx* ctx, const void* src, void* dst, int64_t K, int32_t world, int64_t words, int back);
int orx_sharded_pairwise_steps(orx_comm* comm, orx_opt* opt, int model, orx_table* user, orx_table* item, orx_table* bias,
                               const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t K, int64_t B, int64_t id_stride,
                               int64_t users_global, int64_t items_global, float margin, float slack, int32_t plan_chunk,
                               int flags, double* loss_l2_accum, int32_t* overflow);
/* The same with HOT-ITEM REPLICATION (skewed item popularity): items 0 .. hot_items-1 of a vocabulary sorted by
 * popularity live in the replica tables item_hot [hot_items, D] / bias_hot [hot_items, 1], identical on every rank (the caller fills
 * them from the owners' shards before the first call and writes them back when it wants the shards current: openrec_amd/sharded.py
 * load_hot / sync_hot).  References to those items read the local replica and put nothing on the wire; their gradients are summed
 * per item on the rank, then over the ranks by ONE all-reduce of the [hot_items, D + 4] block per step, and every rank applies the
 * same sums to its replica (the replicas stay identical).  Exact: TF sums the gradients of duplicate ids before the sparse apply.
 * cold_fraction in (0, 1]: the share of a list's item references expected NOT to be hot -- the exchanged buckets (which travel
 * whole: fixed capacities, no size exchange) are sized for that share, which is what takes the hot rows off the wire; a list with
 * more cold references than its buckets hold sets *overflow like any other overflow.  1.0: buckets as without replication.
 * hot_items = 0: orx_sharded_pairwise_steps.  At most 63 ranks.  ORX_SHARD_DEDUP is refused together with hot_items > 0
 * (ORX_ERR_ARG): the plan with the replica as a destination is the one without per-destination dedup.  As with every overflow,
 * a call that sets *overflow dropped references: the tables AND the replicas (whose summed block the dropped triplets' slots
 * still entered) are not the result of the steps any more -- restore from a checkpoint, raise `slack` / `cold_fraction`. */
int orx_sharded_pairwise_steps_hot(orx_comm* comm, orx_opt* opt, int model, orx_table* user, orx_table* item, orx_table* bias,
                                   orx_table* item_hot, orx_table* bias_hot, int64_t hot_items, float cold_fraction,
                                   const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t K, int64_t B, int64_t id_stride,
                                   int64_t users_global, int64_t items_global, float margin, float slack, int32_t plan_chunk, int flags,
                                   double* loss_l2_accum, int32_t* overflow);

/* The hybrid-parallel DLRM step as one host call (BASELINE.json configs[4]; the single-process step: recommenders/dlrm.py:63-100 under
 * tf2_examples/dlrm_criteo.py:42-48).  `emb` is this rank's shard of the COMBINED embedding table (row r of the concatenated tables
 * on rank r % world at local index r / world), `m` a model made with ORX_DLRM_NO_EMB (its MLPs are replicas), dense / sparse / label
 * this rank's K x B slice of the global batches (DEVICE pointers; sparse ids are ids within their own table).  Per step: lookups ->
 * owners (ncclSend / ncclRecv groups), rows back, forward + backward on the rows where they arrived, row gradients to the owners,
 * ONE ncclAllReduce of the packed dense gradients, applies.  loss_accum: device double[1], this rank's share of the global-batch
 * loss is added per step; overflow: device int, sticky (a request bucket -- mean * slack + 8 -- was full).  Same results as the
 * per-phase entry points (orx_shard_bucket, orx_gather_rows, orx_dlrm_grads_indirect, orx_dlrm_dense_pack / _apply, orx_apply_rows)
 * driven by openrec_amd/sharded_dlrm.py. */
int orx_sharded_dlrm_steps(orx_comm* comm, orx_dlrm* m, orx_opt* opt, orx_table* emb, const float* dense, const int32_t* sparse,
                           const float* label, int64_t K, int64_t B, float slack, double* loss_accum, int32_t* overflow);

/* ---- device-time sampling of the kernels (HIP events on the ctx stream) --- */
int orx_prof_enable(orx_ctx* ctx, int on);
int orx_prof_reset(orx_ctx* ctx);
/* total device milliseconds and launch count recorded for kernel `kid` */
int orx_prof_get(orx_ctx* ctx, int kid, double* total_ms, int64_t* launches);
/* What the plan of the most recent exact pairwise call counted (tests, diagnosis; waits for that call's counters if nobody has
 * looked at them yet -- see DESIGN.md 4.0 "no read-back"):
 *   what = 0  rows referenced exactly twice that were PAIRED (accepted pairs, summed over the call's steps)
 *        = 1  largest number of duplicated rows one step left for the apply
 *        = 2  pairwise calls so far that enqueued every launch without waiting for their plan's counters
 *        = 3  1 if those counters were "quiet" (no range wanted a staging plan, few duplicated rows), else 0 */
int orx_ctx_stat(orx_ctx* ctx, int what, int64_t* out);
/* The streaming-copy yardstick of this device (bench.py's `roofline.frac_of_copy_peak`; no reference equivalent): a float4 copy
 * kernel over two buffers of `bytes` each (allocated and released inside the call; bytes % 16 == 0), three untimed launches, then
 * `reps` timed ones on the context's stream; *gbps_out = 2 * bytes / mean launch time, in 1e9 bytes per second. */
int orx_copy_bandwidth(orx_ctx* ctx, int64_t bytes, int32_t reps, double* gbps_out);

#ifdef __cplusplus
}
#endif
#endif /* OPENREC_HIP_H */
