// Internal declarations shared by the translation units of libopenrec_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/openrec_hip.h"

// ---------------------------------------------------------------- errors ---
void orx_set_error(const char* fmt, ...);

#define ORX_HIP(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            orx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                          __FILE__, __LINE__);                                         \
            return _e == hipErrorOutOfMemory ? ORX_ERR_OOM : ORX_ERR_HIP;              \
        }                                                                              \
    } while (0)

#define ORX_ARG(cond, ...)                                                             \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            orx_set_error(__VA_ARGS__);                                                \
            return ORX_ERR_ARG;                                                        \
        }                                                                              \
    } while (0)

// --------------------------------------------------------------- objects ---
struct ProfSlot {
    std::vector<hipEvent_t> ev;   // pairs (start, stop) awaiting collection
    double total_ms = 0.0;
    int64_t launches = 0;
};

struct orx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int* d_err = nullptr;             // sticky device-side "id out of range" flag
    std::vector<struct orx_opt*> opts;  // live optimizers: a table that is destroyed takes its slots out of them
    // staging buffers (grown on demand)
    int32_t* d_ids = nullptr;  size_t d_ids_cap = 0;       // host-id upload
    float* d_lab = nullptr;    size_t d_lab_cap = 0;
    unsigned* d_evalbits = nullptr; size_t d_evalbits_cap = 0;  // evaluation bitmaps (orx_rank_metrics_csr): all zero between calls
    unsigned char* d_dflag = nullptr; size_t d_dflag_cap = 0;   // [K][2B] duplicate flags (pointwise, censor)
    int32_t* d_ids2 = nullptr; size_t d_ids2_cap = 0;            // [K][3B] ids with the duplicate flag in bit 31
    unsigned char* d_roles = nullptr; size_t d_roles_cap = 0;    // [K][3B] dedup scratch
    unsigned char* d_cflag = nullptr; size_t d_cflag_cap = 0;    // [3][K][B] censor election flags
    unsigned int* d_dupbits = nullptr; size_t d_dupbits_cap = 0; // [K][buckets][words] duplicate bitmaps
    // bucketed plan (kernels_plan.hip): references per (step, row range), scatter cursors, bucket lists
    int* d_pl_cnt = nullptr;   size_t d_pl_cnt_cap = 0;          // [K][3 ranges + 1]: counts, cursors, offsets
    const void* pl_cnt_clean = nullptr; size_t pl_cnt_clean_cap = 0; int pl_cnt_nb = -1;      // the buffer (and ranges per step) whose counts and cursors are all zero between plans
    int2* d_pl_list = nullptr; size_t d_pl_list_cap = 0;         // [K][references per step] (id, output position | role << 30)
    bool plan_big = false;                                       // the last plan met a bucket of > 16 k references: 1024-thread workgroups
    int pair_gen = 0; const void* partner_zeroed = nullptr; size_t partner_zeroed_cap = 0;      // generation of the last pairing plan; the partner buffer (pointer, size) that has been zeroed
    int pair_pause = 0;                                          // pairing: calls left before it is tried again (the last plan that paired accepted too few rows to pay for itself)
    // pairing (kernels_plan.hip, "pairing"): the two references of a row referenced exactly twice are brought into one wavefront
    int4* d_partner = nullptr; size_t d_partner_cap = 0;         // [K][B] per triplet: where the partner of its slot 0 / 1 / 2 sits if that row is referenced exactly twice (generation-tagged words, ORX_PARTNER_*)
    int* d_pslot = nullptr;    size_t d_pslot_cap = 0;           // [K][2B] parallel to dlist: position of the FIRST reference of a row referenced exactly twice (-1: another kind of row)
    int4* d_ids4 = nullptr;    size_t d_ids4_cap = 0;            // [K][B] the fused kernel's input with pairing, written by the plan itself: (user, pos, neg) rewritten ids of position j, pairing word | origin << 10
    const float* plan_label = nullptr;                           // pointwise step with pairing: the labels of the chunk being planned (DedupArgs::label)
    int* h_plan = nullptr;                                       // pinned host mirror of the per-step plan counters
    size_t h_plan_cap = 0;
    hipEvent_t plan_ev = nullptr;                                // "plan counters have arrived on the host"
    // What the counters of the LAST plan that was looked at said (api.hip, "no read-back"): a pairwise call whose predecessor was quiet
    // -- no range wanted a staging plan, few duplicated rows, no oversized bucket -- makes its plan with staging off, launches every step
    // without waiting for its own counters and leaves them for the next call to look at (stats_ev: their copy has arrived).
    struct { bool valid = false, quiet = false; int64_t key[5] = {0, 0, 0, 0, 0}; } plan_stats;
    int64_t stat_pairs = 0, stat_max_dup = 0, stat_nowait_calls = 0;      // orx_ctx_stat
    hipEvent_t stats_ev = nullptr;
    bool stats_pending = false; int64_t stats_kc = 0, stats_B = 0; bool stats_pairing = false; int stats_age = 0; int64_t stats_key[5] = {0, 0, 0, 0, 0};
    // plan pipeline (api.hip): the pieces of a chunk after the first are planned on a second stream while the previous piece's
    // steps run; per piece parity: counters on the host / plan complete on the device
    hipStream_t plan_stream = nullptr;
    hipEvent_t pipe_cnt[2] = {nullptr, nullptr}, pipe_done[2] = {nullptr, nullptr};
    hipEvent_t wait_ev = nullptr;                                // orx_ctx_wait_stream
    int epoch = 0;                                               // step epoch: tags ready flags and censor side marks
    int epoch_gen = 0;                                           // bumped when `epoch` wraps (tables then clear their tags)
    // staging plan of rows referenced >= 3 times in a step (see kernels_pairwise.hip, "staging")
    int2* d_refinfo = nullptr; size_t d_refinfo_cap = 0;         // [K][3Bp] (dense row, rank) of role-2 references
    int* d_tricnt = nullptr;   size_t d_tricnt_cap = 0;          // [K][B] references per dense row
    int* d_segstart = nullptr; size_t d_segstart_cap = 0;        // [K][B] first staging slot per dense row
    int* d_alloc = nullptr;    size_t d_alloc_cap = 0;           // [K][8] allocators: dense rows, staging slots, tree items per level
    int* d_dseg = nullptr;     size_t d_dseg_cap = 0;            // [K][2B] parallel to dlist: staging segment start
    int* d_dcnt = nullptr;     size_t d_dcnt_cap = 0;            // [K][2B] parallel to dlist: segment length (0: two-reference row, <0: long)
    int4* d_chunks = nullptr;  size_t d_chunks_cap = 0;          // [K][item_stride] work items of the reduction tree over long segments
    float* d_part = nullptr;   size_t d_part_cap = 0;            // [item_stride][D] partial sums of the tree (one step at a time)
    float* d_partb = nullptr;  size_t d_partb_cap = 0;           // [item_stride] the same for the bias gradients
    float* d_stage = nullptr;  size_t d_stage_cap = 0;           // [2][3B][D] staged gradients (double-buffered by step parity)
    float* d_stageb = nullptr; size_t d_stageb_cap = 0;          // [2][3B] staged bias gradients
    uint32_t* d_dlist = nullptr; size_t d_dlist_cap = 0;         // [K][2B] duplicated rows
    int* d_dcount = nullptr;   size_t d_dcount_cap = 0;          // [K]
    float* d_partial = nullptr; size_t d_partial_cap = 0;   // [K][nwaves][2] loss partials
    double* d_loss = nullptr;  size_t d_loss_cap = 0;       // [K][2] step results
    float* d_tmp = nullptr;    size_t d_tmp_cap = 0;        // misc fp32 scratch
    float* d_wpart = nullptr;  size_t d_wpart_cap = 0;      // [nwaves][D] dense-kernel gradient partials
    // deterministic row apply (kernels_rowsort.hip): ping-pong buffers of the radix sort, its histograms, partial sums of runs
    // that cross a 64-entry block
    uint2* d_sort[2] = {nullptr, nullptr}; size_t d_sort_cap[2] = {0, 0};
    int* d_sort_hist = nullptr; size_t d_sort_hist_cap = 0;
    float* d_csr_part[2] = {nullptr, nullptr}; size_t d_csr_part_cap[2] = {0, 0};
    float* d_splitk = nullptr; size_t d_splitk_cap = 0;     // [splits][M][N] partial products of a split-K fp32 product
    bool prof = false;
    ProfSlot prof_slot[ORX_K_NUM];
    std::vector<int> prof_order;                     // kernel class of every profiled launch, in launch order (ORX_PROF_TIMELINE)
    hipEvent_t cur_e0 = nullptr, cur_e1 = nullptr;   // events of the launch being profiled (or null)
    int num_cu = 256;
};

struct orx_table {
    orx_ctx* ctx = nullptr;
    float* w = nullptr;               // [rows, dim]
    int64_t rows = 0;
    int32_t dim = 0;
    bool owned = true;
    // per-step scratch, allocated on first use by a train step
    float* gsum = nullptr;            // [rows, dim] duplicate-row gradient sums (all-zero between steps)
    float* gsum2 = nullptr;           // second scratch rows of the pairwise step (rows referenced exactly twice)
    int* ready = nullptr;             // [rows] ready flags of the in-launch duplicate apply
    int tag_gen = 0;                  // ctx->epoch_gen the ready / side tags belong to
    int* side = nullptr;              // [rows][2] fused censor: last epoch with a duplicated pos / neg reference
    struct orx_opt* lazy = nullptr;   // lazily-applied Adam: rows are caught up on demand (orx_table_sync flushes)
    uint64_t version = 0;             // bumped by every host-side write / fill / init and by the dense optimizer kernels: derived copies
                                      // (the fp16 copies of the DLRM kernels) know when they are stale
};

struct OptSlots {
    float* s0 = nullptr;              // Adagrad acc / Adam m
    float* s1 = nullptr;              // Adam v
    int* last = nullptr;              // lazy Adam: optimizer step up to which (w, m, v) of the row are current
};

struct orx_opt {
    orx_ctx* ctx = nullptr;
    int kind = ORX_SGD;
    float lr = 0.01f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
    int64_t t = 0;                    // Adam step counter
    std::map<orx_table*, OptSlots> slots;
    // lazy Adam: lr_t of every step taken so far (index = step), host mirror + device copy
    std::vector<float> h_lrt; float* d_lrt = nullptr; size_t lrt_cap = 0; int64_t lrt_uploaded = 0;
    // closed-form replay (round 6, orx_device.h adam_cf_*): per step k the three moments V_q[k] = sum_{j=1..J} b1^j j^q lr_{k+j}, q = 0, 1, 2
    // (float4 per step, w unused); entries [0, lrv_done) are final for the present learning rate
    std::vector<float> h_lrv; float* d_lrv = nullptr; size_t lrv_cap = 0; int64_t lrv_done = 0;
};
constexpr int ORX_ADAM_CF_TERMS = 512;        // J: b1^J is negligible for b1 <= 0.95
// what a kernel needs for the closed-form replay: the moments table and the constants (lrv NULL: the replay loops)
struct AdamCFParams { const float4* lrv; float delta, lb1, lb2; };
AdamCFParams orx_adam_cf_params(const orx_opt* o);
bool orx_adam_cf_ok(const orx_opt* o);        // closed-form replay applicable (b1 <= 0.95, 1 - sqrt(b2) <= 1e-3, ORX_ADAM_NO_CF unset)
// Lazily-applied TF-2.0 Adam: the dense decay of a row that no triplet touches (m *= b1, v *= b2, w -= lr_t m /
// (sqrt(v)+eps), every step) is replayed exactly when the row is next needed.  orx_table_sync brings every row of a
// table up to the optimizer's current step; every entry point that reads or writes a table calls it first.
int orx_table_sync(orx_table* t);
int orx_opt_isolate(orx_opt* o, orx_table* const* keep, int n_keep);   // finish every OTHER table that is lazy under `o`
int orx_opt_last(orx_opt* o, orx_table* t, bool restamp, int** out, int64_t stamp = -1);            // allocate / fetch the per-row step stamps
int orx_launch_fill_int(orx_ctx* ctx, int* p, int64_t n, int v);
int orx_adam_lrt(orx_opt* o, int64_t upto);                      // make lr_t of steps 1..upto available on the device
int orx_launch_adam_flush(orx_ctx* ctx, float* w, float* m, float* v, int* last, int64_t rows, int dim, int t_end, const float* lrt,
                          float b1, float b2, float eps, struct AdamCFParams cf);

// ------------------------------------------------------- helpers (api.hip) ---
int orx_ensure(void** p, size_t* cap, size_t bytes);           // grow a device buffer
int orx_table_scratch(orx_table* t, bool second = false);       // allocate gsum (and gsum2)
int orx_table_side(orx_table* t);                               // allocate the fused-censor side marks
int orx_opt_slots(orx_opt* opt, orx_table* t, OptSlots* out);   // allocate optimizer slots
int stage_ids(orx_ctx* c, const int32_t* host, int64_t n, int64_t off);   // H2D into ctx->d_ids
int fetch_losses(orx_ctx* c, int64_t K, float* loss_out, float* l2_out);
void orx_prof_begin(orx_ctx* ctx, int kid);
void orx_prof_end(orx_ctx* ctx, int kid);

// While a ProfScope is alive, ORX_LAUNCH attaches a (start, stop) event pair to
// the kernel dispatch itself (hipExtLaunchKernelGGL), i.e. the events carry the
// kernel's own begin/end timestamps -- the same quantity rocprofv3 reports.
struct ProfScope {
    orx_ctx* c; int k;
    ProfScope(orx_ctx* c_, int k_) : c(c_), k(k_) { if (c->prof) orx_prof_begin(c, k); }
    ~ProfScope() { if (c->prof) orx_prof_end(c, k); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: one bit per device ordinal, per call site
#define ORX_ONCE_PER_DEVICE(ctx, stmt)                                                  \
    do {                                                                               \
        static unsigned long long _done = 0ull;                                        \
        const unsigned long long _bit = 1ull << ((ctx)->device & 63);                  \
        if (!(_done & _bit)) { stmt; _done |= _bit; }                                  \
    } while (0)

#define ORX_LAUNCH(ctx, kernel, grid, block, shm, ...)                                 \
    hipExtLaunchKernelGGL(kernel, grid, block, shm, (ctx)->stream, (ctx)->cur_e0, (ctx)->cur_e1, 0, __VA_ARGS__)

// ------------------------------------------------ kernel launch parameters ---
// Duplicate detection output, per step (reference index r: user lookup k -> k,
// pos-item lookup k -> B + k, neg-item lookup k -> 2B + k):
//   ids_out[r] = id | (dup << 31): the id with "row referenced more than once in this batch" in bit 31
//                (the pairwise fused kernel reads these instead of the caller's ids); dflag[r] = the same
//                flag as a byte (pointwise / censor / apply_rows)
//   dlist[...] = the distinct duplicated rows of the step: row | (is_item << 31)
//   dcount     = number of entries in dlist
struct PairArgs {
    float* U; float* V; float* b;
    float* gU; float* gV; float* gb;          // duplicate-row gradient sums (zero between steps)
    float* gU2; float* gV2; float* gb2;       // second scratch rows (rows referenced exactly twice), may be NULL
    float* aU; float* aV; float* ab;          // Adagrad accumulators / Adam m
    float* a2U; float* a2V; float* a2b;       // Adam v
    int* lastU; int* lastV; int* lastb;       // lazy Adam: per-row step stamps (NULL: not the lazy Adam path)
    const float* lrt; float b1; float b2; int step_t;   // lr_t per step, betas, index of THIS step (its lr_t = lrt[step_t])
    int newton;                 // lazy Adam: carry 1/(sqrt(v)+eps) by Newton steps (1 - sqrt(beta_2) <= 1e-3)
    int long_gap;               // lazy Adam: tables large relative to the batch (rows wait hundreds of steps): LONGGAP kernel
    const float4* lrv;          // lazy Adam, closed-form replay: V_q per step (orx_opt::d_lrv), NULL: the replay loops
    float cf_delta, cf_lb1, cf_lb2;     // -ln sqrt(b2), log2 b1, log2 b2
    const int32_t* uid; const int32_t* pid; const int32_t* nid;
    const int4* ids4;                         // pairing: [B] (user, pos item, neg item) rewritten ids of the triplet processed at position j and, in w, its pairing
                                              // word (ORX_PAIR_*, bits 9:0) and the triplet's original position (bits 31:10); NULL: uid / pid / nid as usual
    int role_bits;                            // ids carry role (bits 30:29) and urgent (bit 28): tables < 2^28 rows
    // in-launch application of the PREVIOUS step's duplicated rows (n_apply_blocks == 0: off)
    int n_apply_blocks; int epoch;
    const uint32_t* prev_dlist; const int* prev_dcount;
    int censor; float min_norm;               // censor_vec fused into the row write-back (fast dims, exact mode)
    int* sideV;                               // [NI][2] epoch of the last step that referenced the item as a duplicated pos / neg
    int* readyU; int* readyV;                 // per-row ready flags (value = epoch of the launch that applied the row)
    const uint32_t* dlist;                    // duplicated rows of this step
    const int* dcount;
    // staging of rows referenced >= 3 times (NULL stage: such references use atomics into gsum)
    const int2* refinfo; const int* segstart;         // this step
    float* stage; float* stageb;                      // this step's staging buffers
    const int* dseg; const int* dcnt;                 // parallel to dlist
    const float* prev_stage; const float* prev_stageb; const int* prev_dseg; const int* prev_dcnt;   // step s-1 (in-launch apply)
    const int4* items; const int* nitems;             // reduction tree over this step's long segments (hot_reduce_kernel):
    float* part; float* partb;                        //   items of level l at items + tree_off[l], count nitems[l]; partial sums
    int tree_off[3];
    int64_t B; int64_t NU; int64_t NI;
    int D;
    float lr; float eps; float margin; float invB; float l2w;
    float* partial;                           // [nwaves][2] loss / l2 partials of this step
    int* err;
};

struct DedupArgs {
    const int32_t* uid; const int32_t* pid; const int32_t* nid;   // step 0 of the chunk
    int64_t id_stride;                        // elements between consecutive steps
    unsigned char* dflag;                     // [K][flag_stride] or NULL
    int32_t* ids_out;                         // [K][flag_stride] or NULL: id | (dup << 31); 0x7fffffff = invalid id
    unsigned int* dupbits;                    // [K][nbu+nbi][DD_WORDS] "seen twice" bitmaps kept for urgent_kernel, or NULL
    unsigned char* roles;                     // [K][flag_stride] scratch (pass 1 -> pass 2) or NULL: no role bits
    int64_t role_stride;                      // ids_out layout [3][role_stride] per step (0: compact reference order)
    uint32_t* dlist;                          // [K][list_stride]
    int* dcount;                              // [K], zeroed before the launch
    int64_t flag_stride; int64_t list_stride;
    // reference numbering: user refs [0, nU), then pid refs [nU, nU+nP), then nid refs [nU+nP, nU+nP+nN)
    int64_t nU; int64_t nP; int64_t nN;
    int64_t NU; int64_t NI;                   // table rows
    int nbu; int nbi;                         // row-range buckets per table (0 = table not scanned)
    int first_only;                           // censor: dflag = 1 only on non-first references
    // column windows (DLRM): the item id stream is an [nP / col_F][col_F] matrix whose column c only holds rows of one
    // table; item range bk scans columns col_win[bk].x .. +col_win[bk].y-1 only (NULL: the whole stream)
    int col_F; const int2* col_win;
    // staging plan (all NULL: rows referenced >= 3 times keep role 2 = atomics)
    int2* refinfo;                            // [K][flag_stride]
    int* tricnt; int* segstart;               // [K][tri_stride], tricnt zeroed before the launch
    int* alloc;                               // [K][8], zeroed before the launch: 0 dense rows, 1 staging slots, 2.. tree items of level 1, 2, 3
    int* dseg; int* dcnt;                     // [K][list_stride]
    int4* items;                              // [K][item_stride] tree work items (src, len, dst, -), level l at offset tree_off[l]
    int64_t tri_stride; int64_t item_stride;
    int tree_off[3];
    int min_late;                             // bucketed plan: see PairPlan
    // pairing (bucketed plan only; pair_tpw < 2: off): triplets per wavefront of the fused kernel that will run the steps,
    // per-step arrays [K][pair_stride] partner (16-byte record per triplet) and ids4, [K][2 list_stride] pslot; the number of accepted pairs
    // of step s is alloc[8 s + 7]
    int pair_tpw; int4* partner; int* pslot; int4* ids4; int64_t pair_stride;
    int pair_gen;                             // generation tag (1 .. 63) of this plan's partner words: see ORX_PARTNER_* below
    // pointwise steps (two id lists, nN = 0) with pairing: word z of a sample's input record carries its LABEL (the bits of the float)
    // instead of a third id -- it travels with the ids when plan_swap_kernel moves the sample.  [K][id_stride], NULL: pairwise
    const float* label;
};
// pairing word of a triplet (PairArgs::pinfo): the triplet shares one row with the triplet of lane group PARTNER of the same wavefront;
// the WRITER adds the partner's gradient of that row to its own and updates the row in place, the other one does not write it
// A word of a triplet's pairing record partner[t].{x,y,z}: bits 23:0 the position of the other reference of that slot's row, bit 30
// "this is the row's second reference", bits 29:24 the GENERATION of the plan that wrote it.  The records are never initialised per
// plan (16 bytes per triplet and step, a fifth of what plan_part_kernel<true> wrote): a word counts only if it carries the current
// plan's generation; the buffer is zeroed when it is allocated and whenever the generations wrap (every 63 plans).  Position
// ORX_PARTNER_POISON marks a triplet with an out-of-range id (written for such triplets only): never paired.
constexpr uint32_t ORX_PARTNER_POS = 0x00ffffffu, ORX_PARTNER_POISON = 0x00ffffffu;
constexpr uint32_t ORX_DLIST_DEAD = 0xffffffffu;     // entry of a step's list of duplicated rows whose row was paired after all: nothing to apply
constexpr uint32_t ORX_PAIR_VALID = 0x200u, ORX_PAIR_WRITER = 0x100u;      // bits 3:0 partner lane group, 5:4 my slot (0 user, 1 pos, 2 neg), 7:6 partner's slot
constexpr int ORX_SEG_DIRECT = 16;            // the apply sums up to this many staged gradients / partial sums of a row itself
constexpr int ORX_PIECE = 64;                 // longer segments: a tree of 64-to-1 partial sums (hot_reduce_kernel, one wavefront per piece)

struct ReduceArgs {
    const float* partial;                     // [K][nwaves][2]
    double* out;                              // [K][2]
    int nwaves;
};

struct RowsArgs {                              // sharded building blocks (kernels_sharded.hip)
    float* W; float* bias;                    // table [rows, D], bias [rows] or NULL
    float* G; float* gb;                      // gsum scratch
    float* A; float* ab;                      // Adagrad accumulators
    const int32_t* ids;                       // local row ids, < 0 = skip
    const unsigned char* dflag;
    const float* grads; int64_t g_stride;     // [n, g_stride]; bias gradient at column D ...
    const float* gbias;                       // ... or, if not NULL, in this array [n] (apply_rows_sgd_flagged_kernel)
    int64_t n; int64_t rows; int D;
    float lr; float eps;
    int* err;
    // planned apply (orx_apply_rows_planned): ids rewritten by dedup_kernel (item role), scratch rows, staging
    const int32_t* ids2; float* G2; float* gb2;
    const int2* refinfo; const int* segstart; float* stage; float* stageb;
};

struct AdamRowsArgs {                          // lazy TF-2.0 Adam on gradient rows (kernels_sharded.hip)
    float* W; float* M; float* V; int* last;  // table, slots, per-row step stamps
    float* G;                                 // gsum scratch (duplicated rows)
    const int32_t* ids;                       // local row ids, < 0 = skip
    const unsigned char* dflag; const uint32_t* dlist; const int* dcount;    // dedup of ids (item role)
    const float* grads; int64_t g_stride;
    int64_t n; int64_t rows; int D;
    const float* lrt; float lr_T; float b1; float b2; float eps;
    int T; int newton;
    int* err;
    AdamCFParams cf;                          // closed-form replay (orx_device.h AdamCF)
};
int orx_launch_adam_rows(orx_ctx* ctx, bool step, const AdamRowsArgs& a, int64_t max_dups);
// lazy Adam on a table's gradient rows: replay the rows of `ids` to the optimizer's step (touch) / take step opt->t
struct ColWindows { int F = 0; const int2* win = nullptr; };    // device array, one entry per 425 984-row range of the table
int orx_adam_rows_dedup(orx_ctx* ctx, orx_table* t, const int32_t* ids, int64_t n, ColWindows cw = ColWindows());
int orx_adam_rows_touch(orx_ctx* ctx, orx_opt* opt, orx_table* t, const int32_t* ids, int64_t n, bool have_dedup, ColWindows cw = ColWindows());
int orx_adam_rows_apply(orx_ctx* ctx, orx_opt* opt, orx_table* t, const int32_t* ids, int64_t n, const float* grads, int64_t g_stride,
                        bool have_dedup, ColWindows cw = ColWindows());
int orx_adagrad_rows_apply(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n, const float* grads,
                           int64_t g_stride, ColWindows cw = ColWindows());
int orx_table_touch(orx_table* t, const int32_t* ids, int64_t n);       // no-op unless the table is lazy
bool orx_adam_rows_lazy(const orx_opt* opt, const orx_table* t);

struct GradArgs {
    const float* u; const float* p; const float* n; int64_t row_stride;   // gathered rows, bias at column D
    const int32_t* valid;                     // triplet k is live iff valid == NULL or valid[k] >= 0
    float* gu; float* gp; float* gn; int64_t g_stride;
    int64_t T; int D;
    float invB; float margin; float l2w;
    float* partial;
};

int orx_launch_pair_grads(orx_ctx* ctx, int model, const GradArgs& a, int* nwaves);
int orx_launch_apply_rows(orx_ctx* ctx, int optkind, bool use_dflag, const RowsArgs& a);
bool orx_launch_apply_rows_pair(orx_ctx* ctx, const RowsArgs& a, const RowsArgs& b, int* rc);
int orx_launch_loss_accumulate(orx_ctx* ctx, const float* partial, int64_t nwaves, double* accum);
int orx_shard_grads_nwaves(int D, int64_t T);

// pointwise (GMF / WRMF) step, kernels_pointwise.hip
struct PointArgs {
    float* U; float* V; float* b; const float* w;
    float* gU; float* gV; float* gb;
    float* aU; float* aV; float* ab;
    const int32_t* uid; const int32_t* iid; const float* label;
    const unsigned char* dflag;               // [2B]: user refs then item refs (tables >= 2^28 rows / generic dims)
    // exact mode with role bits: uid / iid are the ids REWRITTEN by dedup_kernel (flag 31, role 30:29), iid = uid + Bp
    int role_bits;
    float* gU2; float* gV2; float* gb2;       // second scratch rows (rows referenced exactly twice)
    const int2* refinfo; const int* segstart; float* stage; float* stageb;   // staging plan (stage NULL: atomics)
    int64_t B; int64_t NU; int64_t NI;
    int D;
    float lr; float eps; float invB; float l2w; float a_w; float b_w;
    int sigmoid;                              // WRMF: PointwiseMSELoss(sigmoid=True), pointwise_mse_loss.py:24-25
    float* partial;                           // [nwaves][2]
    float* wpartial;                          // [nwaves][D] (GMF)
    int* err;
    // lazy TF-2.0 Adam: second slots, per-row step stamps (the bias shares its item row's), lr_t table
    float* a2U; float* a2V; float* a2b; int* lastU; int* lastV; int* lastb;
    const float* lrt; float b1; float b2; int step_t; int newton;
    const float4* lrv; float cf_delta, cf_lb1, cf_lb2;      // closed-form replay (orx_device.h AdamCF; NULL: the replay loops)
    // in-launch application of the PREVIOUS step's duplicated rows, as in the pairwise step (n_apply_blocks == 0: off): the first
    // blocks of the launch run inline_apply on `ap` (the tables seen as a pairwise step sees them, prev_* = step s-1's lists);
    // references marked urgent (bit 28 of the rewritten ids) wait for their row's ready flag
    int n_apply_blocks; int epoch; const int* readyU; const int* readyV;
    PairArgs ap;
    // pairing (round 6; kernels_plan.hip): the step's input records (user id, item id, label bits, pairing word | origin << 10), NULL: off
    const int4* ids4;
    // GMF, float4 kernels (round 6): the Dense(1) gradient is summed and its rule applied by wt_nred REDUCER workgroups at the end of the
    // launch itself (kernels_pointwise.hip dense_tail_reducer) instead of two more launches between this step and the next.  0: off
    int wt_nred;                              // orx_point_reducers(D, B)
    float* wt_rows;                           // [wt_nred - 1][D] the sums of 64 consecutive workgroups' partial rows (slots start out WT_EMPTY)
    float* wt_gout; float* wt_l2slot; float* wt_acc; int wt_optkind;      // as dense_reduce_kernel's arguments
};

int orx_launch_point_fused(orx_ctx* ctx, int model, int optkind, int mode, const PointArgs& a);
int orx_launch_dense_reduce(orx_ctx* ctx, const float* wpartial, int nwaves, int D, float* w, float l2w, float* gout,
                            float* l2slot, float* acc, int optkind /* -1: gradient to gout only */, float lr, float eps);
int orx_launch_dense_apply(orx_ctx* ctx, float* w, float* acc, float* g, int n, int optkind, float lr, float eps);
int orx_launch_score_all(orx_ctx* ctx, const float* U, const float* V, const float* b, const float* w,
                         const int32_t* uid, int64_t nq, int64_t NU, int64_t NI, int D, int kind, float* out);
int orx_point_nwaves(int D, int64_t B);
int orx_point_wparts(int D, int64_t B);
bool orx_point_dense_tail_ok(int D);
int orx_point_reducers(int D, int64_t B);
int orx_launch_score_mfma(orx_ctx* ctx, const float* U, const float* V, const float* b, const float* w, const int32_t* uid,
                          int64_t nq, int64_t NU, int64_t NI, int D, int kind, float* out, bool* launched);

// launchers implemented in kernels_pairwise.hip
int orx_launch_dedup(orx_ctx* ctx, const DedupArgs& a, int64_t K);
int orx_launch_fused(orx_ctx* ctx, int model, int optkind, int mode, const PairArgs& a);
int orx_launch_hot_reduce(orx_ctx* ctx, const PairArgs& a, int level);

// host-side plan of the exact steps (api.hip), shared by the pairwise and the pointwise step
struct PairPlan { int nw; int64_t chunk; int64_t list_stride; int64_t Bp; int64_t item_stride; int tree_off[3];
                  int min_late = -1;      // staging plan from this many third-or-later references per range on (< 0: max(64, n / 512))
                  int pair_tpw = 0;       // pairing: triplets per wavefront of the fused kernel (0: off)
                  int64_t cap = 0; };     // steps the per-step scratch is sized for (>= chunk)
struct ExactChunk { bool hot = false, use_stage = false, dense_dups = false; int tree_levels = 0;
                    bool quiet = false; };      // the plan's counters were quiet (api.hip, "no read-back")
// pairing (kernels_plan.hip): is it on for a step of this form, and its per-call buffers (api.hip)
bool orx_pairing_wanted(int mode, bool role_bits, int optkind, int dim, int64_t B, int fb);
int orx_pairing_buffers(orx_ctx* c, int64_t B, int dim, PairPlan* plan);
int orx_exact_buffers(orx_ctx* c, orx_table* U, orx_table* V, int64_t K, int64_t B, int mode, bool role_bits,
                      bool inline_apply, bool staging, int nb_total, int nw, PairPlan* plan);
int orx_exact_plan_chunk(orx_ctx* c, orx_table* U, orx_table* V, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t ds,
                         int64_t nU, int64_t nP, int64_t nN, int64_t kc, int64_t B, bool role_bits, bool inline_apply, bool staging,
                         const PairPlan& plan, ExactChunk* out, const std::function<int()>* while_waiting = nullptr);
// the same in two halves, for steps i0 .. i0 + kc - 1 of the chunk's plan arrays (bucketed plan only): issue enqueues the plan on the
// context's CURRENT stream and records `counters` after the read-back of the per-step counters; finish waits for it on the host
int orx_exact_plan_issue(orx_ctx* c, orx_table* U, orx_table* V, const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t ds,
                         int64_t nU, int64_t nP, int64_t nN, int64_t kc, int64_t B, bool inline_apply, bool staging,
                         const PairPlan& plan, int64_t i0, hipEvent_t counters, const std::function<int()>* after_readback = nullptr);
int orx_exact_plan_finish(orx_ctx* c, int64_t kc, int64_t B, bool inline_apply, bool staging, int64_t i0, hipEvent_t counters, ExactChunk* out, bool pairing_on = false);
void orx_exact_step_views(orx_ctx* c, const PairPlan& plan, int64_t i, int64_t B, int D, bool use_stage, PairArgs* a);
int orx_launch_rows_planned(orx_ctx* ctx, int optkind, const RowsArgs& a);
// K id lists of n local rows each (ids [K][n], < 0 = padding) against ONE table: plan once (duplicate roles, staging
// plan, reduction tree), then per list i: orx_apply_rows_planned_step(i, grads).  SGD / Adagrad.
struct RowsPlan { PairPlan plan; ExactChunk ck; int64_t n = 0; bool ready = false; };
int orx_apply_rows_plan(orx_ctx* ctx, orx_table* t, const int32_t* ids, int64_t K, int64_t n, int64_t id_stride, RowsPlan* out);
int orx_apply_rows_planned_step(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const RowsPlan& rp, int64_t i,
                                const int32_t* ids, const float* grads, int64_t g_stride);
int orx_launch_dup_apply(orx_ctx* ctx, int optkind, const PairArgs& a);
bool orx_launch_tail(orx_ctx* ctx, int optkind, const PairArgs& a, const ReduceArgs& r, int64_t K, int* rc);     // dup_apply of the last step + loss_reduce in one launch
int orx_launch_urgent(orx_ctx* ctx, const DedupArgs& a, int64_t K);
// bucketed plan (kernels_plan.hip): same outputs as orx_launch_dedup (+ orx_launch_urgent) with `d` filled the same way
bool orx_plan_v2(bool role_bits);
int orx_plan_buffers(orx_ctx* c, int64_t chunk, int64_t B, int64_t NU, int64_t NI, bool want_dupbits);
int orx_launch_plan(orx_ctx* ctx, const DedupArgs& d, int64_t kc, bool keep_dupbits, int64_t step0 = 0);
int orx_launch_plan_urgent(orx_ctx* ctx, const DedupArgs& d, int64_t kc, int64_t step0 = 0);
int orx_launch_plan_swap(orx_ctx* ctx, const DedupArgs& d, int64_t kc);      // pairing: the records of the positions an accepted pair moves change places
int orx_fused_tpw(int D);                        // triplets per wavefront of the float4 fused kernel (0: generic dim)
int orx_fused_can_inline_apply(int D);
int orx_dedup_words(void);
int orx_launch_loss_reduce(orx_ctx* ctx, const ReduceArgs& a, int64_t K);
int orx_fused_nwaves(int D, int64_t B);
int orx_dedup_buckets(int64_t rows);
int64_t orx_dedup_range_rows();                  // rows per dedup range (one workgroup)

// fused-kernel modes
enum { MODE_EXACT = 0,     // unique rows in place; duplicate rows -> gsum, applied by dup_apply
       MODE_HOGWILD = 1,   // everything in place, racy, no duplicate handling
       MODE_ACCUM = 2,     // everything -> gsum (Adam: dense sweep follows)
       MODE_LOSS = 3 };    // forward only

// kernels_misc.hip
int orx_launch_init_uniform(orx_ctx* ctx, float* w, int64_t n, float lo, float hi, uint64_t seed);
int orx_launch_fill(orx_ctx* ctx, float* w, int64_t n, float v);
int orx_launch_gather(orx_ctx* ctx, const float* w, const float* bias, int64_t rows, int dim,
                      const int32_t* ids, int64_t n, float* out, int64_t out_stride, int* err,
                      int skip_negative = 0, float* bias_out = nullptr);   // bias_out: the biases go there ([n]) instead of column dim of out
int orx_launch_censor(orx_ctx* ctx, float* w, const unsigned char* dflag, int64_t rows, int dim, const int32_t* ids,
                      int64_t n, float min_norm, int* err);
int orx_launch_censor2(orx_ctx* ctx, float* wA, const unsigned char* fA, int64_t rowsA, const int32_t* idsA, int64_t nA,
                       float* wB, const unsigned char* fB, int64_t rowsB, const int32_t* idsB, int64_t nB,
                       int dim, float min_norm);
int orx_launch_adam_sweep(orx_ctx* ctx, float* w, float* m, float* v, float* gsum, int64_t n,
                          float lr_t, float b1, float b2, float eps);

// kernels_dense.hip (DLRM dense side)
int orx_launch_gemm(orx_ctx* ctx, const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1,
                    float* C, int64_t ldc, const float* bias, int M, int N, int K, int act, bool c_zero = false, float out_scale = 1.0f);
int orx_launch_gemm_f16(orx_ctx* ctx, const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1,
                        float* C, int64_t ldc, const float* bias, int M, int N, int K, int act, bool c_zero = false, float out_scale = 1.0f);
struct DenseParam { float* w; float* acc; float* acc2; float* g; int64_t n; };
int orx_launch_dense_apply_multi(orx_ctx* ctx, const DenseParam* ps_dev, int count, int64_t max_n, int optkind, float lr, float eps,
                                 float b1 = 0.f, float b2 = 0.f);
// the same rule in 64 x 64 tiles, with the rest of a dense parameter's per-step work folded in: the split-K slices of its gradient
// (kernels_gemm16.hip) are added on the way in (slice order; times slab_scale = 1 / loss scale), the fp16 copies of the new weights
// (w16 [rows][ld16], w16t [cols][ld16t], transposed through LDS) are written on the way out
struct DenseFused {
    float* w; float* acc; float* acc2; float* g; int rows, cols;
    const float* slab; int S, ntn;                 // NULL: the gradient is g alone
    void* w16; void* w16t; int ld16, ld16t;        // NULL: no fp16 copies
    int tile0, tiles_x;                            // first tile of this parameter in the launch's flat tile list; tiles per tile row
    const float* cparts; int cN;                   // a vector parameter (rows == 1 or cols == 1) whose gradient arrives as partial rows [P][cN]
                                                   // (ColPart): added here in colparts_reduce_kernel's order -- no reduce launch (round 6)
};
// first tile of every parameter, and how many partial rows its gradient has THIS step (0: none) -- kernel arguments
struct DenseFusedTiles { int count; int tile0[48]; int cP[48]; };
struct CsrFinish;       // (orx_csr_device.h) the sorted sparse apply's finish pass, carried by the dense optimizer's launch
int orx_launch_dense_apply_fused(orx_ctx* ctx, const DenseFused* ps_dev, const DenseFusedTiles& tt, int total_tiles, int optkind, float lr, float eps,
                                 float b1, float b2, float slab_scale, const CsrFinish* finish = nullptr);
// orx_csr_apply without its second launch: the blocks' sums + rule are launched, the finish pass is DESCRIBED (*finish) for a launch that takes it along
int orx_csr_apply_split(orx_ctx* ctx, orx_opt* opt, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride,
                        bool skip_single, CsrFinish* finish);
// Column sums (bias gradients) leave their producers as one partial row per row block -- parts[p * N + c], plain stores --
// and one colparts_reduce launch per MLP backward adds the blocks in order: no fp32 atomics, reproducible sums.
struct ColPart { float* parts = nullptr; int P = 0; };        // in: the workspace; out: row blocks written
struct ColJob { const float* parts; float* out; int N, P; float scale; };     // out = scale * sum of the partial rows
constexpr int ORX_COLJOBS_MAX = 24;
struct ColJobs { ColJob j[ORX_COLJOBS_MAX]; };
int orx_launch_colparts_reduce(orx_ctx* ctx, const ColJob* jobs, int n);
int orx_launch_act_bwd_colsum(orx_ctx* ctx, float* dY, const float* Y, int64_t ldy, int M, int N, int act, ColPart* gb,
                              void* d16 = nullptr, int64_t ld16 = 0, const float* src = nullptr, int64_t ld_src = 0, float src_scale = 1.0f);
int orx_launch_gemm_f16s(orx_ctx* ctx, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                         void* C16, int64_t ldc16, const float* bias, int M, int N, int K, int act,
                         const float* actY = nullptr, int64_t ldy = 0, int act_y = 0, ColPart* gb = nullptr);
// second-generation fp16 products (kernels_gemm16.hip): C = A16 * B16^T on K-contiguous fp16 operands with the fused epilogues,
// and the weight gradient C += A16^T * B16 from batch-major fp16 copies (split-K through fp32 slabs + one reduce launch)
bool orx_gemm16_nt_ok(int64_t lda, int64_t ldb, int N, int K);
int orx_launch_gemm16_nt(orx_ctx* ctx, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                         void* C16, int64_t ldc16, const float* bias, int M, int N, int K, int act,
                         const float* actY = nullptr, const void* actY16 = nullptr, int64_t ldy = 0, int act_y = 0, ColPart* gb = nullptr,
                         // relu masks (kernels_gemm16.hip Nt16Args): written by a relu layer's forward launch, read by the input-gradient launch above it
                         unsigned long long* mask_out = nullptr, const unsigned long long* mask_in = nullptr);
int orx_gemm16_nt_config(orx_ctx* ctx, int M, int N, int64_t* words_out);
#define ORX_SLAB_STRIDE (128 * 128 + 64)          // floats per (tile, slice) of a split-K workspace (kernels_gemm16.hip)
struct SlabReduce { const float* slab; float* C; int64_t ldc; int M, N, S, ntn, tiles; };
bool orx_gemm16_tn_ok(int64_t lda, int64_t ldb, int N);
void orx_gemm16_tn_plan(orx_ctx* ctx, int M, int N, int K, int* S_out, int* tiles_out, int* kchunk_out);
int orx_launch_gemm16_tn(orx_ctx* ctx, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C, int64_t ldc,
                         float* slab, int M, int N, int K, float out_scale = 1.0f);
int orx_launch_slab_reduce(orx_ctx* ctx, const void* jobs_dev, int n_jobs, int max_tiles, float out_scale = 1.0f);
// the weight gradient (arguments of orx_launch_gemm16_tn) and the input gradient (of orx_launch_gemm16_nt) of one layer in ONE launch
bool orx_gemm16_group_ok(orx_ctx* ctx, int B, int in, int out, int64_t ldx16, int64_t ldw16);
int orx_launch_gemm16_group(orx_ctx* ctx, const void* X16, int64_t ldx, const void* dZ16, int64_t lddz, float* gW, int64_t ldgw, float* slab,
                            int in, int out, int B, float out_scale,
                            const void* W16, int64_t ldw, float* C, int64_t ldc, void* C16, int64_t ldc16,
                            const float* actY, const void* actY16, int64_t ldy, int act_y, ColPart* gbp, const unsigned long long* mask_in = nullptr,
                            int nt_cols = 0);     // > 0: the input-gradient product's column count (the weight operand padded with zero rows)
int orx_launch_cast16(orx_ctx* ctx, const float* src, int64_t lds_, void* dst16, int64_t ld16, int M, int N);
bool orx_head16_ok(int K, int64_t ldx);
int orx_launch_head_fwd(orx_ctx* ctx, const void* X16, int64_t ldx, const void* w16, const float* bias, int act, float* pred, int B, int K);
// hl != NULL: the loss is folded into the head's backward (no dlrm_loss_kernel launch): dy is formed from pred and the label, the loss
// terms leave as one fp64 partial per workgroup in loss_part[0 .. orx_head_bwd_blocks(B))
struct HeadLoss { const float* label; int bce; float thr; int64_t n_mean; float gscale; double* loss_part;
                  const float* fwd_bias; float* pred_out; };       // fwd_bias != NULL: head_bwd_kernel runs the head's forward as well (pred -> pred_out)
int orx_launch_head_bwd(orx_ctx* ctx, const void* X16, int64_t ldx, const void* w16, const float* dy, const float* pred, int act, int act_below,
                        ColPart* gW, ColPart* gb, void* dZ16, int64_t ld16, float* dZ32, int64_t ld32, ColPart* gb_below, int B, int K,
                        const HeadLoss* hl = nullptr);
int orx_head_bwd_blocks(orx_ctx* ctx, int B);
int orx_launch_head_loss_finish(orx_ctx* ctx, const double* parts, int64_t stride, int n, int64_t K, int64_t n_mean, double* loss_out);
struct ShadowParam { const float* w; void* w16; void* w16t; int in, out, ld16, ld16t; };
int orx_launch_dense_shadow(orx_ctx* ctx, const ShadowParam* ps_dev, int count, int64_t max_n);
int orx_launch_act_bwd(orx_ctx* ctx, float* dY, const float* Y, int64_t ldy, int M, int N, int act);
int orx_launch_copy2d(orx_ctx* ctx, float* dst, int64_t ldd, const float* src, int64_t lds_, int M, int N, float scale = 1.0f);
int orx_launch_interact(orx_ctx* ctx, bool fwd, const float* Z, const float* dR, int F, int d, int compat, int itself,
                        float* out, int P, int64_t B, int ldR, void* R16 = nullptr, int ldR16 = 0, bool* wrote16 = nullptr,
                        const float* emb = nullptr, const int32_t* idx = nullptr, int64_t emb_rows = 0, float scale = 1.0f,
                        float* gdst = nullptr,
                        // backward with direct rows: flags [B][F] of the lookups whose row nobody else references in the step -- their
                        // SGD / Adagrad update is applied in place by the kernel (kernels_dense.hip FusedRows), no gradient row is written
                        const unsigned char* single = nullptr, int opt_kind = 0, float lr = 0.f, float eps = 0.f, float* acc_rows = nullptr);
bool orx_interact_direct_ok(int F, int d, int compat);
bool orx_interact_fuse_ok(int F, int d, int compat);
int orx_launch_dlrm_loss(orx_ctx* ctx, float* P, const float* y, int64_t B, int bce, float thr, float* dP, double* loss_out,
                         int64_t n_mean = 0, int accumulate = 0, float gscale = 1.0f);
int orx_launch_dlrm_tiny_apply(orx_ctx* ctx, const int32_t* idx, const float* dZ, const int* tiny_f_dev, int n_tiny, int max_rows,
                               const int64_t* offset, const int64_t* rows, int F, int d, int64_t B, float lr, float* W);
int orx_launch_dlrm_mask_tiny(orx_ctx* ctx, const int32_t* idx, const unsigned char* is_tiny_dev, int F, int64_t total, int32_t* out);
int orx_launch_dlrm_ids(orx_ctx* ctx, const int32_t* sparse, const int64_t* offset, const int64_t* rows, int nf, int64_t B,
                        int32_t* idx);
int orx_launch_rows_accum(orx_ctx* ctx, float* G, const int32_t* ids, const float* grads, int64_t g_stride, int64_t n, int D, int64_t rows);

// kernels_rowsort.hip: deterministic apply of per-occurrence gradient rows (stable sort by row + segmented sums in position order)
int orx_rows_sort(orx_ctx* ctx, const int32_t* ids, int64_t K, int64_t n, int64_t id_stride, int64_t rows, const uint2** sorted);
int orx_rows_sort_reserve(orx_ctx* ctx, int64_t K, int64_t n, int64_t rows);
int orx_csr_apply(orx_ctx* ctx, orx_opt* opt, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride,
                  bool skip_single = false);       // skip_single: rows referenced once were updated by the kernel that formed their gradient
int orx_rows_single_flags(orx_ctx* ctx, const uint2* sorted, int64_t K, int64_t n, int64_t rows, unsigned char* flags);
int orx_csr_accum(orx_ctx* ctx, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride);
int orx_csr_adam(orx_ctx* ctx, bool step, const AdamRowsArgs& r, orx_table* t, const uint2* sorted, int64_t n);
int orx_adam_rows_sorted(orx_ctx* ctx, orx_opt* opt, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride, bool step);
int orx_adam_dense_sorted(orx_ctx* ctx, orx_opt* opt, orx_table* t, const uint2* sorted, int64_t n, const float* grads, int64_t g_stride);

// device-side exchange plan of the sharded step (kernels_sharded.hip)
struct RouteArgs {
    const int32_t* uid; const int32_t* pid; const int32_t* nid; int64_t B;
    int world; int cap;
    int32_t* send;           // [world*cap][3], pre-filled with -1
    int* counters;           // [world], zeroed
    int* overflow; int* err;
    int64_t NU, NI;          // GLOBAL table rows (id validation)
    int64_t id_stride;       // K-step launch (grid.y = step): ids of step k at + k*id_stride, send / counters of step k
                             // at + k*world*cap*3 / + k*world
};

struct RequestArgs {
    const int32_t* trip;     // [T][3] received triplets, u < 0 = empty slot
    int64_t T; int world; int cap;
    int32_t* send_ids;       // [world*cap], pre-filled with -1
    int32_t* slot;           // [2T]: slot of the p / n request of triplet t, -1 if none
    int32_t* u_loc;          // [T]: local user row or -1
    int* counters;           // [world], zeroed
    int* overflow;
    // K-step launch (grid.y = step): every array of step k at + k * (its per-step size)
    // hot-item replication (0: off): items 0 .. hot-1 live in a replica on every rank -- their references ask nobody.
    // They take slots world * cap .. world * cap + cap_hot - 1 (the region BEHIND the exchanged buckets in the row / gradient buffers)
    // and leave their ids in hot_ids [cap_hot] (pre-filled with -1); counters then has world + 1 entries per step
    int hot; int cap_hot; int32_t* hot_ids;
};

struct ShardGradArgs {
    const float* U;          // local user shard [rows, D]
    const float* rows_in;    // [world*cap2][DS] received item rows, bias at column D
    const int32_t* u_loc;    // [T]
    const int32_t* slot;     // [2T]
    float* gu;               // [T][D]
    float* send_g;           // [world*cap2][DS]
    int64_t T; int D; int DS;
    float invB; float margin; float l2w;
    float* partial;
    // SGD with the user apply folded in (orx_shard_grads_sgd): duplicate flags of the u_loc list, writable table, rate,
    // and the id list left for the flagged apply of the duplicated rows (-1: already applied)
    const unsigned char* fu; float* Uw; float lr; int32_t* u_apply;
    // per-destination dedup of the requests (orx_shard_request_dedup_steps): references that share a slot add their gradients
    // into it (send_g zeroed by the caller); NULL: every reference has a slot of its own
    const unsigned char* dupref;   // [2T]
    float* gdup;                   // [2T][DSg] side buffer: gradients of the references that share a slot (row = reference index)
    // the bias apart from the rows (the sharded engine's exchange carries it as a message of its own: D + 1 floats per row on the
    // wire instead of D + 4): received biases [world*cap2], bias gradients out [world*cap2]; rows_in / send_g rows are then DS = D floats
    const float* bias_in; float* gb_out; int DSg;   // DSg: row stride of gdup (D + 4; its column D is the bias gradient)
};

struct DedupReqArgs {
    const int32_t* trip;     // [K][T][3]
    int64_t T; int world; int cap; int64_t Lr;     // Lr = rows of the item table per rank (ceil(items_global / world))
    int32_t* keys;           // [K][2T] scratch: owner * Lr + local row of every item reference (p refs, then n refs), -1 dead
    const uint2* sorted;     // [K][2T] (key, reference) ascending
    int32_t* uq;             // [K][2T] scratch
    int* chunkcnt; int nchunk;  // [K][nchunk] heads per chunk of 1024 sorted entries, then their exclusive prefix
    int* ostart;             // [K][64] index (among the list's distinct keys) of every owner's first key
    int2* seglist; int* segcount;   // [K][T] (slot, first sorted entry) of the shared slots, [K] their number (NULL: not wanted)
    int32_t* send_ids; int32_t* slot; int32_t* u_loc; unsigned char* dupref; int* overflow;
};
int orx_launch_shard_keys(orx_ctx* ctx, const DedupReqArgs& a, int64_t K);
int orx_launch_shard_dedup_slots(orx_ctx* ctx, const DedupReqArgs& a, int64_t K);
int orx_shard_grads_impl(orx_ctx* ctx, int model, orx_opt* opt, orx_table* user, const float* rows_in, const float* bias_in,
                         const int32_t* u_loc, const int32_t* slot, const uint8_t* dupref, const void* sorted, const void* seglist,
                         const int32_t* segcount, float* gdup, const uint8_t* dup_u, int64_t T, int64_t row_stride, int64_t B_global,
                         float margin, int flags, float* gu, int32_t* u_apply, float* send_g, float* gb_out, double* loss_l2_accum,
                         float* partial_ext = nullptr, int* nwaves_out = nullptr);
int orx_apply_rows_flagged_impl(orx_ctx* ctx, orx_opt* opt, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                                const float* grads, int64_t g_stride, const float* gbias, const unsigned char* dflag);
// two flagged SGD lists (tables of one dim) in one launch where that applies, else one after the other
int orx_apply_rows_flagged_pair(orx_ctx* ctx, orx_opt* opt, orx_table* tA, orx_table* biasA, const int32_t* idsA, int64_t nA, const float* gA, int64_t strideA,
                                const float* gbiasA, const unsigned char* flagA, orx_table* tB, orx_table* biasB, const int32_t* idsB, int64_t nB, const float* gB,
                                int64_t strideB, const float* gbiasB, const unsigned char* flagB, bool flags_over_both = false);
int orx_launch_shard_segsum(orx_ctx* ctx, const int2* seglist, const int* segcount, const uint2* sorted, int64_t n, const float* gdup, int DSg,
                            float* send_g, int DS, int D, float* gb_out);
int orx_launch_shard_route(orx_ctx* ctx, const RouteArgs& a, int64_t K = 1);
int orx_launch_shard_request(orx_ctx* ctx, const RequestArgs& a, int64_t K = 1);
int orx_launch_shard_hot_pack(orx_ctx* ctx, float* gV, float* gb, float* hg, int64_t H, int D, int DSh);      // hot-item replication: scratch sums -> [H][D + 4] block
int orx_launch_shard_bucket(orx_ctx* ctx, const int32_t* ids, int64_t n, int world, int cap, int32_t* send_ids, int32_t* slot,
                            int* counters, int* overflow);
int orx_launch_shard_localize(orx_ctx* ctx, const int32_t* ids, int64_t n, int world, int32_t* out);
int orx_launch_shard_grads(orx_ctx* ctx, int model, const ShardGradArgs& a, int* nwaves);

// kernels_eval.hip (ranking metrics of the evaluation step)
struct EvalArgs {
    const float* pred;            // [n, NI]
    const unsigned char* pos;     // [n, NI]
    const unsigned char* excl;    // [n, NI]
    int64_t NI;
    const float* at; int nat;     // cut-offs
    float* auc; float* ndcg; float* recall;   // [n], [n, nat], [n, nat]
    int* err;
};
int orx_launch_rank_metrics(orx_ctx* ctx, const EvalArgs& a, int64_t n);
struct EvalCsrArgs {
    const float* pred;            // [n, NI]
    unsigned* pbits; unsigned* ebits;                 // [n, W] bitmaps built from the lists (ebits = pbits + n W)
    const int64_t* pos_ptr; const int32_t* pos_items; // CSR over the n users of the call
    const int64_t* excl_ptr; const int32_t* excl_items;
    int64_t NI, W;
    const float* at; int nat;
    float* auc; float* ndcg; float* recall;
    unsigned* part; int* neval; int S; int* flag_out;
    int64_t q0;                                       // first user of this launch (slices of the batch)                // per-(user, segment) partial counts [n][S][136]; evaluated items per user
    int* err;
};
int orx_rank_csr_segments(int64_t n, int64_t NI);
int orx_launch_mask_bits(orx_ctx* ctx, const EvalCsrArgs& a, int64_t n, int clear);
int orx_launch_rank_sweeps(orx_ctx* ctx, const EvalCsrArgs& a, int64_t q0, int64_t nq, int64_t max_pos);

// kernels_sampler.hip (on-device triplet sampler)
struct SamplerArgs {
    const int32_t* rec_user; const int32_t* rec_item; int64_t R;      // interaction records
    const int64_t* ptr; const int32_t* items;                          // CSR of positives (sorted per user)
    int64_t total_items; int64_t total_users;
    uint64_t seed; int64_t first; int64_t n; int h;
    int32_t* uid; int32_t* pid; int32_t* nid;
};
int orx_launch_sample_pairwise(orx_ctx* ctx, const SamplerArgs& a);
int orx_launch_sample_stratified(orx_ctx* ctx, const SamplerArgs& a, float pos_ratio, float* label, int* blockcnt, int64_t* blockbase,
                                 int64_t* counter);
int orx_launch_sample_perpos(orx_ctx* ctx, const SamplerArgs& a, int nneg, float* label);
