"""ctypes binding of include/openrec_hip.h (libopenrec_hip.so, built in-tree by
openrec_amd.build).  There is no CPU fallback: if the library cannot be loaded
the import fails loudly, and every call fails if no HIP device is usable."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORX_LIB_PATH") or os.path.join(HERE, "_lib", "libopenrec_hip.so")      # (ORX_LIB_PATH: A/B runs of two builds on one box)

ORX_OK, ORX_ERR_ARG, ORX_ERR_HIP, ORX_ERR_OOM, ORX_ERR_INDEX, ORX_ERR_STATE = 0, -1, -2, -3, -4, -5
ORX_SGD, ORX_ADAGRAD, ORX_ADAM = 0, 1, 2
ORX_BPR, ORX_UCML = 0, 1
ORX_GMF, ORX_WRMF = 0, 1
ORX_IDS_DEVICE, ORX_HOGWILD, ORX_NO_L2, ORX_CENSOR, ORX_POINT_SIGMOID = 1, 2, 4, 8, 16
ORX_SHARD_OVERLAP, ORX_SHARD_NO_DEDUP, ORX_SHARD_DEDUP, ORX_COMM_ID_BYTES = 0x100, 0x200, 0x400, 128
ORX_DLRM_INTERACT_ITSELF, ORX_DLRM_SIGMOID_BOT, ORX_DLRM_SIGMOID_TOP, ORX_DLRM_LOSS_BCE, ORX_DLRM_REFERENCE_COMPAT = 1, 2, 4, 8, 16
ORX_DLRM_FP16_MLP = 32
ORX_DLRM_NO_EMB = 64
ORX_K_DEDUP, ORX_K_FUSED, ORX_K_REDUCE, ORX_K_SWEEP, ORX_K_CENSOR, ORX_K_POINT, ORX_K_DUPAPPLY, ORX_K_GEMM, ORX_K_NUM = 0, 1, 2, 3, 4, 5, 6, 7, 8
KERNEL_NAMES = {ORX_K_DEDUP: "dedup", ORX_K_FUSED: "fused", ORX_K_REDUCE: "loss_reduce", ORX_K_SWEEP: "adam_sweep",
                ORX_K_CENSOR: "censor", ORX_K_POINT: "pointwise", ORX_K_DUPAPPLY: "dup_apply", ORX_K_GEMM: "gemm"}

_p = c_void_p
_pp = POINTER(c_void_p)
_ip = c_void_p      # int32* (host numpy .ctypes.data or device pointer)
_fp = c_void_p      # float*

# name -> (restype, argtypes); must list EVERY symbol declared in include/openrec_hip.h
SIGNATURES = {
    "orx_version": (c_int, []),
    "orx_last_error": (c_char_p, []),
    "orx_ctx_create": (c_int, [c_int, _p, _pp]),
    "orx_ctx_destroy": (c_int, [_p]),
    "orx_synchronize": (c_int, [_p]),
    "orx_ctx_wait_stream": (c_int, [_p, _p]),
    "orx_check_index_error": (c_int, [_p]),
    "orx_table_create": (c_int, [_p, c_int64, c_int32, _pp]),
    "orx_table_wrap": (c_int, [_p, _p, c_int64, c_int32, _pp]),
    "orx_table_destroy": (c_int, [_p]),
    "orx_table_rows": (c_int64, [_p]),
    "orx_table_dim": (c_int32, [_p]),
    "orx_table_device_ptr": (c_void_p, [_p]),
    "orx_table_init_uniform": (c_int, [_p, c_float, c_float, c_uint64]),
    "orx_table_fill": (c_int, [_p, c_float]),
    "orx_table_read": (c_int, [_p, c_int64, c_int64, _fp]),
    "orx_table_write": (c_int, [_p, c_int64, c_int64, _fp]),
    "orx_table_gather": (c_int, [_p, _ip, c_int64, _fp, c_int]),
    "orx_table_censor": (c_int, [_p, _ip, c_int64, c_float, c_int]),
    "orx_opt_create": (c_int, [_p, c_int, c_float, c_float, c_float, c_float, _pp]),
    "orx_opt_destroy": (c_int, [_p]),
    "orx_opt_set_lr": (c_int, [_p, c_float]),
    "orx_opt_get_step": (c_int, [_p, POINTER(c_int64)]),
    "orx_opt_set_step": (c_int, [_p, c_int64]),
    "orx_opt_advance": (c_int, [_p, _p, c_int32]),
    "orx_opt_slot_read": (c_int, [_p, _p, c_int, c_int64, c_int64, _fp]),
    "orx_opt_slot_write": (c_int, [_p, _p, c_int, c_int64, c_int64, _fp]),
    "orx_pairwise_step": (c_int, [_p, c_int, _p, _p, _p, _p, _ip, _ip, _ip, c_int64, c_int64, c_int64,
                                  c_float, c_int, _fp, _fp]),
    "orx_pairwise_reserve": (c_int, [_p, _p, _p, _p, _p, c_int64, c_int64]),
    "orx_pairwise_loss": (c_int, [_p, c_int, _p, _p, _p, _ip, _ip, _ip, c_int64, c_float, c_int, _fp, _fp]),
    "orx_pointwise_step": (c_int, [_p, c_int, _p, _p, _p, _p, _p, _ip, _ip, _fp, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_int, _fp, _fp]),
    "orx_pointwise_loss": (c_int, [_p, c_int, _p, _p, _p, _p, _ip, _ip, _fp, c_int64, c_float, c_float, c_int, _fp, _fp]),
    "orx_score_all_items": (c_int, [_p, c_int, _p, _p, _p, _p, _ip, c_int64, _fp]),
    "orx_score_all_items_device": (c_int, [_p, c_int, _p, _p, _p, _p, _ip, c_int64, _fp]),
    "orx_rank_metrics": (c_int, [_p, c_int, _p, _p, _p, _p, _ip, _fp, _p, _p, c_int64, c_int64, _fp, c_int32, _fp, _fp, _fp]),
    "orx_rank_metrics_csr": (c_int, [_p, c_int, _p, _p, _p, _p, _ip, _fp, c_int32, c_int64, c_int64, _p, _p, _p, _p, _fp, c_int32, _fp, _fp, _fp]),
    "orx_sampler_create": (c_int, [_p, _ip, _ip, c_int64, _p, _ip, c_int64, c_int64, _pp]),
    "orx_sampler_destroy": (c_int, [_p]),
    "orx_sampler_pairwise": (c_int, [_p, c_uint64, c_int64, c_int64, _ip, _ip, _ip]),
    "orx_sampler_stratified": (c_int, [_p, c_uint64, c_int64, c_int64, c_float, _ip, _ip, _p]),
    "orx_sampler_per_pos_stratified": (c_int, [_p, c_uint64, c_int64, c_int64, c_double, _ip, _ip, _p]),
    "orx_dlrm_create": (c_int, [_p, c_int32, c_int32, _p, c_int32, _p, c_int32, _p, c_int32, c_int, c_float, c_uint64, _pp]),
    "orx_dlrm_destroy": (c_int, [_p]),
    "orx_dlrm_param": (c_int, [_p, c_int, c_int, _pp]),
    "orx_dlrm_step": (c_int, [_p, _p, _fp, _ip, _fp, c_int64, c_int64, c_int, _fp]),
    "orx_dlrm_inference": (c_int, [_p, _fp, _ip, c_int64, c_int, _fp]),
    "orx_dlrm_grads": (c_int, [_p, _p, _p, _p, c_int64, c_int64, _p, _p]),
    "orx_dlrm_direct_ok": (c_int, [_p]),
    "orx_dlrm_grads_indirect": (c_int, [_p, _fp, _fp, c_int64, _ip, _fp, c_int64, c_int64, _fp, _p]),
    "orx_dlrm_dense_count": (c_int, [_p, POINTER(c_int64)]),
    "orx_dlrm_dense_pack": (c_int, [_p, _p]),
    "orx_dlrm_dense_apply": (c_int, [_p, _p, _p]),
    "orx_gather_rows": (c_int, [_p, _p, _p, _ip, c_int64, _fp, c_int64]),
    "orx_pair_grads": (c_int, [_p, c_int, c_int32, _fp, _fp, _fp, c_int64, _ip, c_int64, c_int64, c_float, c_int,
                               _fp, _fp, _fp, c_int64, _p]),
    "orx_apply_rows": (c_int, [_p, _p, _p, _p, _ip, c_int64, _fp, c_int64]),
    "orx_rows_dupflags": (c_int, [_p, c_int64, _ip, c_int64, c_int64, c_int64, _p]),
    "orx_apply_rows_flagged": (c_int, [_p, _p, _p, _p, _ip, c_int64, _fp, c_int64, _p]),
    "orx_shard_route": (c_int, [_p, _ip, _ip, _ip, c_int64, c_int64, c_int64, c_int32, c_int32, _ip, _ip, _ip]),
    "orx_shard_request": (c_int, [_p, _ip, c_int64, c_int32, c_int32, _ip, _ip, _ip, _ip, _ip]),
    "orx_shard_localize": (c_int, [_p, _ip, c_int64, c_int32, _ip]),
    "orx_comm_unique_id": (c_int, [_p]),
    "orx_comm_create": (c_int, [_p, _p, c_int32, c_int32, _pp]),
    "orx_comm_destroy": (c_int, [_p]),
    "orx_vgroup_create": (c_int, [c_int32, _pp]),
    "orx_vgroup_abort": (c_int, [_p]),
    "orx_vgroup_destroy": (c_int, [_p]),
    "orx_comm_create_virtual": (c_int, [_p, _p, c_int32, _pp]),
    "orx_comm_rank": (c_int, [_p]),
    "orx_comm_world": (c_int, [_p]),
    "orx_comm_stats": (c_int, [_p, c_int, _p]),
    "orx_comm_ping": (c_int, [_p, c_int64, c_int32, _p]),
    "orx_sharded_caps": (c_int, [c_int64, c_int32, c_float, _p, _p]),
    "orx_shard_regroup": (c_int, [_p, _p, _p, c_int64, c_int32, c_int64, c_int]),
    "orx_sharded_pairwise_steps": (c_int, [_p, _p, c_int, _p, _p, _p, _ip, _ip, _ip, c_int64, c_int64, c_int64, c_int64, c_int64,
                                           c_float, c_float, c_int32, c_int, _p, _ip]),
    "orx_sharded_pairwise_steps_hot": (c_int, [_p, _p, c_int, _p, _p, _p, _p, _p, c_int64, c_float, _ip, _ip, _ip, c_int64, c_int64, c_int64, c_int64,
                                               c_int64, c_float, c_float, c_int32, c_int, _p, _ip]),
    "orx_sharded_dlrm_steps": (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int64, c_int64, c_float, _p, _p]),
    "orx_shard_route_steps": (c_int, [_p, _ip, _ip, _ip, c_int64, c_int64, c_int64, c_int64, c_int64, c_int32, c_int32, _ip, _ip, _ip]),
    "orx_shard_request_steps": (c_int, [_p, _ip, c_int64, c_int64, c_int32, c_int32, _ip, _ip, _ip, _ip, _ip]),
    "orx_shard_bucket": (c_int, [_p, _ip, c_int64, c_int32, c_int32, _ip, _ip, _ip, _ip]),
    "orx_shard_grads": (c_int, [_p, c_int, _p, _fp, _ip, _ip, _p, _p, _p, _ip, _fp, c_int64, c_int64, c_int64, c_float, c_int, _fp, _fp, _p]),
    "orx_shard_grads_sgd": (c_int, [_p, c_int, _p, _p, _fp, _ip, _ip, _p, _p, _p, _ip, _fp, _p, c_int64, c_int64, c_int64, c_float, c_int, _fp, _ip, _fp, _p]),
    "orx_shard_request_dedup_steps": (c_int, [_p, _ip, c_int64, c_int64, c_int32, c_int32, c_int64, _ip, _ip, _ip, _p, _p, _p, _ip, _ip]),
    "orx_prof_enable": (c_int, [_p, c_int]),
    "orx_prof_reset": (c_int, [_p]),
    "orx_prof_get": (c_int, [_p, c_int, POINTER(c_double), POINTER(c_int64)]),
    "orx_ctx_stat": (c_int, [_p, c_int, POINTER(c_int64)]),
    "orx_copy_bandwidth": (c_int, [_p, c_int64, c_int32, POINTER(c_double)]),
    "orx_mlp_forward": (c_int, [_p, c_int32, POINTER(_p), POINTER(_p), POINTER(c_int32), _p, c_int64, c_int32, c_int, _p]),
    "orx_interact_forward": (c_int, [_p, _p, c_int64, c_int32, c_int32, c_int, c_int, c_int, _p]),
}

_lib = None


class OrxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[orx {code}] {msg}")
        self.code = code


def load():
    """Load the shared library and attach signatures (no HIP call is made)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m openrec_amd.build` "
            "(hipcc, gfx950).  openrec_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 and must be the one that gets
    # loaded (the library resolves the same soname).  With the system runtime loaded first, a later
    # `import torch` finds no usable device ("No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc == ORX_OK:
        return
    msg = load().orx_last_error().decode("utf-8", "replace")
    if rc == ORX_ERR_INDEX:
        raise IndexError(msg)          # the reference's CPU gather raises on an out-of-range id
    if rc == ORX_ERR_OOM:
        raise MemoryError(msg)
    if rc == ORX_ERR_ARG:
        raise ValueError(msg)
    raise OrxError(rc, msg)
