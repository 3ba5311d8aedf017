"""ctypes binding of libhiertcn_gfx950.so (include/hiertcn.h).  There is NO fallback: if the
library is missing or does not export the ABI, importing the compute path raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhiertcn_gfx950.so")

MAX_BLOCKS = 8
MAX_LAYERS = 4
MAX_TOPK = 16
ABI_VERSION = 12
N_STRUCTS = 6
PRECISION_STORAGE16, PRECISION_FP32, PRECISION_SPLIT16 = 0, 1, 2

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_u16p = C.POINTER(C.c_uint16)
c_f64p = C.POINTER(C.c_double)


class Config(C.Structure):
    _fields_ = [("item_num", C.c_int32), ("emb_dim", C.c_int32), ("hidden_dim", C.c_int32),
                ("num_layer", C.c_int32), ("kernel_size", C.c_int32), ("n_blocks", C.c_int32),
                ("channels", C.c_int32 * MAX_BLOCKS), ("hier", C.c_int32), ("low_type", C.c_int32)]


class Weights(C.Structure):
    _fields_ = [("emb_kernel", C.c_void_p), ("emb_bias", C.c_void_p), ("proj_kernel", C.c_void_p),
                ("conv_kernel", C.c_void_p * MAX_BLOCKS), ("conv_bias", C.c_void_p * MAX_BLOCKS),
                ("res_kernel", C.c_void_p * MAX_BLOCKS), ("res_bias", C.c_void_p * MAX_BLOCKS),
                ("out_kernel", C.c_void_p), ("out_bias", C.c_void_p),
                ("gru_gate_kernel", C.c_void_p * MAX_LAYERS), ("gru_gate_bias", C.c_void_p * MAX_LAYERS),
                ("gru_cand_kernel", C.c_void_p * MAX_LAYERS), ("gru_cand_bias", C.c_void_p * MAX_LAYERS),
                ("low_gate_kernel", C.c_void_p * MAX_LAYERS), ("low_gate_bias", C.c_void_p * MAX_LAYERS),
                ("low_cand_kernel", C.c_void_p * MAX_LAYERS), ("low_cand_bias", C.c_void_p * MAX_LAYERS)]


class GmmConfig(C.Structure):
    _fields_ = [("item_num", C.c_int32), ("output_dim", C.c_int32), ("num_mixture", C.c_int32), ("kernel_size", C.c_int32),
                ("n_blocks", C.c_int32), ("channels", C.c_int32 * MAX_BLOCKS), ("sigma", C.c_float),
                ("trunk", C.c_int32), ("hidden_dim", C.c_int32), ("num_layer", C.c_int32), ("input_dim", C.c_int32), ("table_pass", C.c_int32)]


class GmmWeights(C.Structure):
    _fields_ = [("conv_kernel", C.c_void_p * MAX_BLOCKS), ("conv_bias", C.c_void_p * MAX_BLOCKS),
                ("res_kernel", C.c_void_p * MAX_BLOCKS), ("res_bias", C.c_void_p * MAX_BLOCKS),
                ("head_weights", C.c_void_p), ("head_biases", C.c_void_p),
                ("gru_gate_kernel", C.c_void_p * MAX_LAYERS), ("gru_gate_bias", C.c_void_p * MAX_LAYERS),
                ("gru_cand_kernel", C.c_void_p * MAX_LAYERS), ("gru_cand_bias", C.c_void_p * MAX_LAYERS),
                # training of trunk 2 as model_hier's low level only: the high level's variables (hiertcn_gmm_train_step_hier)
                ("hl_emb_kernel", C.c_void_p), ("hl_emb_bias", C.c_void_p),
                ("hl_gate_kernel", C.c_void_p * MAX_LAYERS), ("hl_gate_bias", C.c_void_p * MAX_LAYERS),
                ("hl_cand_kernel", C.c_void_p * MAX_LAYERS), ("hl_cand_bias", C.c_void_p * MAX_LAYERS)]


TUNING_FIELDS = ("gru_scan_persistent", "gru_scan_persistent_train", "gru_bptt_persistent", "gru_bptt_fused", "lgru_persistent", "lgru_pipe", "lgru_tile",
                 "lgru_tile_bwd", "lgru_wide", "lgru_wide_bwd", "lgru_fp32_cells", "tcn_layered", "gru_persist_max_lanes", "gemm_split_max_tiles",
                 "skinny_max_rows", "lgru_tile_max_sessions", "lgru_wide_max_launches", "sgrad_dyn_max_blocks", "sgrad0_min_stages", "tcn_gs2_max_sessions",
                 "lgru_defer_max_gib", "train_fp32_chunk_mib")


class Tuning(C.Structure):
    """hiertcn_tuning_t (include/hiertcn.h, ABI 10): launch selection (-1 automatic, 0 never, 1 wherever the shape allows) and thresholds of a handle."""
    _fields_ = [(f, C.c_int32) for f in TUNING_FIELDS] + [("reserved", C.c_int32 * 10)]


# The environment variables the LIBRARY used to read on its launch path (rounds 1-4), mapped onto hiertcn_tuning_t fields.  Parsed HERE, once per
# Engine (tuning_from_env), never in C: name -> (field, value string -> field value or None to leave the default).
ENV_TUNING = {
    "HIERTCN_GRU_CHAIN": ("gru_scan_persistent", lambda v: 0 if v == "1" else None),            # 1: the chain of GEMM launches (a shared GPU: distributed.py)
    "HIERTCN_GRU_TRAIN_CHAIN": ("gru_scan_persistent_train", lambda v: 0),
    "HIERTCN_GRU_BPTT_CHAIN": ("gru_bptt_persistent", lambda v: 0),
    "HIERTCN_GRU_BPTT_UNFUSED": ("gru_bptt_fused", lambda v: 0),
    "HIERTCN_LGRU_CHAIN": ("lgru_persistent", lambda v: 0 if v == "1" else 1 if v == "0" else None),   # 0: the persistent launch wherever it fits
    "HIERTCN_LGRU_PIPE": ("lgru_pipe", lambda v: 0 if v == "0" else None),
    "HIERTCN_LGRU_TILE": ("lgru_tile", lambda v: 0 if v == "0" else None),
    "HIERTCN_LGRU_TILE_BWD_OFF": ("lgru_tile_bwd", lambda v: 0),
    "HIERTCN_LGRU_WIDE": ("lgru_wide", lambda v: 0 if v == "0" else None),
    "HIERTCN_LGRU_WIDE_BWD": ("lgru_wide_bwd", lambda v: 0 if v == "0" else None),
    "HIERTCN_LGRU_FP32": ("lgru_fp32_cells", lambda v: 1),
    "HIERTCN_FORCE_LAYERED": ("tcn_layered", lambda v: 1),
    "HIERTCN_GRU_PERSIST_MAX_LANES": ("gru_persist_max_lanes", int),
    "HIERTCN_GEMM_SPLIT_MAX_TILES": ("gemm_split_max_tiles", int),
    "HIERTCN_SKINNY_MAX_ROWS": ("skinny_max_rows", int),
    "HIERTCN_LGRU_TILE_MAX": ("lgru_tile_max_sessions", int),
    "HIERTCN_LGRU_WIDE_MAX_LAUNCHES": ("lgru_wide_max_launches", int),
    "HIERTCN_SGRAD_DYN_MAX_BLOCKS": ("sgrad_dyn_max_blocks", int),
    "HIERTCN_SGRAD0_MIN_STAGES": ("sgrad0_min_stages", int),
    "HIERTCN_TCN_GS2_MAX": ("tcn_gs2_max_sessions", lambda v: -1 if int(v) <= 0 else int(v)),
    "HIERTCN_LGRU_DEFER_MAX_GIB": ("lgru_defer_max_gib", lambda v: -1 if int(v) <= 0 else int(v)),   # <= 0: never defer (0 in the struct = the default)
    "HIERTCN_TRAIN_FP32_CHUNK_MIB": ("train_fp32_chunk_mib", int),
}


def tuning_from_env(environ=None) -> dict:
    """{field: value} for every variable of ENV_TUNING that is set -- read once, when an Engine is created."""
    environ = os.environ if environ is None else environ
    out = {}
    for name, (field, conv) in ENV_TUNING.items():
        if name in environ:
            v = conv(environ[name])
            if v is not None:
                out[field] = int(v)
    return out


class ScoreOut(C.Structure):
    _fields_ = [("rank", C.c_void_p), ("ce", C.c_void_p), ("target", C.c_void_p),
                ("topk_idx", C.c_void_p), ("topk_val", C.c_void_p), ("topk", C.c_int32)]


# every symbol include/hiertcn.h declares: (restype, argtypes)
SYMBOLS = {
    "hiertcn_abi_version": (C.c_int, []),
    "hiertcn_storage_dtype": (C.c_int, []),
    "hiertcn_last_error": (C.c_char_p, []),
    "hiertcn_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Config)]),
    "hiertcn_destroy": (C.c_int, [C.c_void_p]),
    "hiertcn_get_config": (C.c_int, [C.c_void_p, C.POINTER(Config)]),
    "hiertcn_set_weights": (C.c_int, [C.c_void_p, C.POINTER(Weights), C.c_void_p]),
    "hiertcn_set_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "hiertcn_get_precision": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "hiertcn_get_tuning": (C.c_int, [C.c_void_p, C.POINTER(Tuning)]),
    "hiertcn_set_tuning": (C.c_int, [C.c_void_p, C.POINTER(Tuning)]),
    "hiertcn_struct_sizes": (C.c_int, [C.POINTER(C.c_size_t)]),
    "hiertcn_session_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "hiertcn_embed_concat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "hiertcn_gru_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "hiertcn_gru_scan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hiertcn_tcn_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32]),
    "hiertcn_tcn_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "hiertcn_train_enable": (C.c_int, [C.c_void_p]),
    "hiertcn_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "hiertcn_train_forward_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                  C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hiertcn_train_step_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hiertcn_train_step_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_size_t, C.c_void_p]),
    "hiertcn_count_users": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hiertcn_set_score_ids": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hiertcn_train_segment": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "hiertcn_grad_rows_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "hiertcn_grad_rows_list": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    "hiertcn_grad_rows_op": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "hiertcn_grad_arena": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "hiertcn_train_arena": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "hiertcn_adam_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hiertcn_adam_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "hiertcn_export_weights": (C.c_int, [C.c_void_p, C.POINTER(Weights), C.c_int32, C.c_void_p]),
    "hiertcn_score_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(ScoreOut), C.c_void_p]),
    "hiertcn_score_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "hiertcn_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hiertcn_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "hiertcn_forward_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(ScoreOut), C.c_void_p]),
    "hiertcn_gmm_last_error": (C.c_char_p, []),
    "hiertcn_gmm_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(GmmConfig)]),
    "hiertcn_gmm_destroy": (C.c_int, [C.c_void_p]),
    "hiertcn_gmm_set_weights": (C.c_int, [C.c_void_p, C.POINTER(GmmWeights), C.c_void_p]),
    "hiertcn_gmm_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64]),
    "hiertcn_gmm_forward_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32]),
    "hiertcn_gmm_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hiertcn_gmm_forward_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hiertcn_gmm_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hiertcn_gmm_head_means": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "hiertcn_gmm_lane_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hiertcn_gmm_train_enable": (C.c_int, [C.c_void_p, C.POINTER(GmmWeights), C.c_void_p]),
    "hiertcn_gmm_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32]),
    "hiertcn_gmm_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "hiertcn_gmm_train_hier_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32]),
    "hiertcn_gmm_train_step_hier": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hiertcn_gmm_adam_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "hiertcn_gmm_adam_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hiertcn_gmm_train_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hiertcn_gmm_train_copy": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(GmmWeights), C.c_void_p]),
    "hiertcn_gmm_train_arena": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "hiertcn_timing_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "hiertcn_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
    "hiertcn_persistent_health": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "hiertcn_persistent_health_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
_lib_f16 = None
_lib_variant = {}
LIB_PATH_F16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhiertcn_gfx950_f16.so")
# other builds of the bfloat16 library (hiertcn_amd/build.py): 'ablation' = + every superseded kernel and its A/B switch; 'testhooks' = + fault injection
LIB_VARIANTS = {"ablation": os.path.join(HERE, "libhiertcn_gfx950_ablation.so"), "testhooks": os.path.join(HERE, "libhiertcn_gfx950_testhooks.so")}


class HierTCNError(RuntimeError):
    pass


def load(f16: bool = False, variant: str = None):
    """Load the shared library (once per storage variant).  Raises if it is absent: the product path never falls
    back to a CPU implementation.  f16=True loads the IEEE-half build (inference only); variant: a LIB_VARIANTS key (A/B runs, tests)."""
    global _lib, _lib_f16
    if variant:
        if variant not in _lib_variant:
            _lib_variant[variant] = _open(LIB_VARIANTS[variant], 0)
        return _lib_variant[variant]
    if f16:
        if _lib_f16 is None:
            _lib_f16 = _open(LIB_PATH_F16, 1)
        return _lib_f16
    if _lib is not None:
        return _lib
    _lib = _open(os.environ.get("HIERTCN_LIB", LIB_PATH), 0)   # (HIERTCN_LIB: another build of the same ABI, for A/B runs on one box)
    return _lib


def _open(path, want_dtype):
    # torch first: its wheel bundles the HIP runtime (libamdhip64.so.7); loading it before our
    # library makes both resolve to ONE runtime, which sharing streams and device pointers needs.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise HierTCNError(
            f"{path} not found: build it with `python -m hiertcn_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the compute path.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.hiertcn_abi_version() != ABI_VERSION:
        raise HierTCNError(f"ABI mismatch: library {lib.hiertcn_abi_version()} vs binding {ABI_VERSION}")
    if lib.hiertcn_storage_dtype() != want_dtype:
        raise HierTCNError(f"{path}: storage dtype {lib.hiertcn_storage_dtype()}, expected {want_dtype}")
    return lib


def check(rc: int, lib=None):
    if rc != 0:
        msg = (lib or load()).hiertcn_last_error()
        raise HierTCNError(f"hiertcn call failed ({rc}): {msg.decode() if msg else '?'}")
