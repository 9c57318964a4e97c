// Entry points still to be filled in (pointwise models, sharded building blocks).
#include "orx_internal.h"

extern "C" int orx_pointwise_step(orx_ctx*, int, orx_opt*, orx_table*, orx_table*, orx_table*, orx_table*,
                                  const int32_t*, const int32_t*, const float*, int64_t, int64_t, int64_t,
                                  float, float, int, float*, float*) {
    orx_set_error("orx_pointwise_step: not implemented yet");
    return ORX_ERR_STATE;
}
extern "C" int orx_gather_rows(orx_ctx* ctx, orx_table* t, orx_table* bias, const int32_t* ids, int64_t n,
                               float* out, int64_t out_stride) {
    ORX_ARG(ctx && t && (n == 0 || (ids && out)), "orx_gather_rows: NULL argument");
    ORX_HIP(hipSetDevice(ctx->device));
    return orx_launch_gather(ctx, t->w, bias ? bias->w : nullptr, t->rows, t->dim, ids, n, out, out_stride, ctx->d_err);
}
extern "C" int orx_pair_grads(orx_ctx*, int, int32_t, const float*, const float*, const float*, int64_t,
                              int64_t, int64_t, float, int, float*, float*, float*, int64_t, double*) {
    orx_set_error("orx_pair_grads: not implemented yet");
    return ORX_ERR_STATE;
}
extern "C" int orx_apply_rows(orx_ctx*, orx_opt*, orx_table*, orx_table*, const int32_t*, int64_t, const float*, int64_t) {
    orx_set_error("orx_apply_rows: not implemented yet");
    return ORX_ERR_STATE;
}
