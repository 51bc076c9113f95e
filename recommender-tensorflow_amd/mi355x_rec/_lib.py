"""ctypes binding of libmi355x_rec.so (include/mi355x_rec.h).

The library is the product: there is no Python / PyTorch fallback for any entry point.  If the
shared object is missing or an entry fails, an exception is raised.

torch must be imported before the library is loaded: PyTorch-ROCm bundles its own
``libamdhip64.so`` (soname ``libamdhip64.so.7``) and the dynamic loader then resolves this
library's DT_NEEDED ``libamdhip64.so.7`` to that already-loaded runtime, so device pointers and
streams handed over by torch belong to the same HIP context.
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime this library must share)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355x_rec.so")

ABI_VERSION = 21


class MiError(RuntimeError):
    pass


class OptHparams(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("lr_t", C.c_float), ("decay", C.c_float),
                ("momentum", C.c_float), ("lr_power", C.c_float), ("l1", C.c_float),
                ("l2", C.c_float)]


AMAX_SLOTS = 64      # MI_AMAX_SLOTS: floats per abs-max vector
MAX_WEIGHT_JOBS = 8  # MI_MAX_WEIGHT_JOBS


class GemmAmax(C.Structure):
    """mi_gemm_amax_t: device pointers to the abs-max vectors of a GEMM's operands / result."""
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p)]


class Planes(C.Structure):
    """mi_planes_t: a matrix as fp16 high + low planes with one power-of-two exponent per row."""
    _fields_ = [("data", C.c_void_p), ("row_exp", C.c_void_p), ("blk_stride", C.c_int64)]


class WgradJob(C.Structure):
    """mi_wgrad_job_t: one layer of mi_dense_bwd_weight_planes_batch"""
    _fields_ = [("X", Planes), ("dY", Planes), ("dW", C.c_void_p), ("db", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32),
                ("amax", GemmAmax)]


class WeightJob(C.Structure):
    """mi_weight_job_t"""
    _fields_ = [("offset", C.c_int64), ("K", C.c_int32), ("N", C.c_int32), ("w", Planes), ("wt", Planes)]


_p = C.c_void_p
_i32, _i64, _u32, _u64, _f32, _sz = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t
_amax = C.POINTER(GemmAmax)
_pl = C.POINTER(Planes)

# name -> (restype, argtypes); mirrors include/mi355x_rec.h declaration by declaration
SIGNATURES = {
    "mi_abi_version": (_i32, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_build_info": (C.c_char_p, []),
    "mi_set_step_state": (_i32, [_p]),
    "mi_step_advance": (_i32, [_p, _p, _p]),
    "mi_fingerprint64": (_u64, [_p, _sz]),
    "mi_crc32c": (_u32, [_p, _sz, _u32]),
    "mi_hash_bucket_i64": (_i32, [_p, _i64, _i64, _p]),
    "mi_hash_bucket_bytes": (_i32, [_p, _p, _i64, _i64, _p]),
    "mi_bucketize_f32": (_i32, [_p, _i64, _p, _i32, _p]),
    "mi_embed_fm_linear_fwd": (_i32, [_p, _p, _p, _p, _i64, _i32, _i32, _p, _i64, _p, _p, _p, _p, _i32, _i64, _p]),
    "mi_gather_rows": (_i32, [_p, _p, _p, _i64, _i32, _p, _p, _i32, _i64, _i64, _p]),
    "mi_numeric_embed_fwd": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _i64, _i64, _p, _p, _p, _p]),
    "mi_numeric_raw_fwd": (_i32, [_p, _p, _i64, _i32, _p, _i64, _i64, _i32, _p, _p]),
    "mi_numeric_raw_bwd_workspace_bytes": (_sz, [_i64, _i32]),
    "mi_numeric_raw_bwd": (_i32, [_p, _p, _i64, _i32, _p, _p, _sz, _p]),
    "mi_embed_fm_linear_bwd": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "mi_entry_grads_segsum": (_i32, [_p, _p, _p, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _i32, _i32, _p, _p, _i64, _i64, _i64, _p]),
    "mi_numeric_embed_bwd_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "mi_numeric_embed_bwd": (_i32, [_p, _p, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _i32, _i32, _p,
                                    _p, _p, _sz, _p]),
    "mi_sort_unique_workspace_bytes": (_sz, [_i64]),
    "mi_sort_unique_fields_workspace_bytes": (_sz, [_i64, _i32]),
    "mi_sort_unique_fields": (_i32, [_p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _p, _sz, _i32, _p]),
    "mi_sort_unique_rows": (_i32, [_p, _i64, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "mi_sort_unique_rows_slots": (_i32, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "mi_global_rows": (_i32, [_p, _p, _i64, _i32, _p, _p]),
    "mi_shard_keys": (_i32, [_p, _i64, _i32, _i64, _i64, _i32, _p, _p]),
    "mi_route_requests": (_i32, [_p, _p, _i64, _i64, _i32, _p, _p, _p]),
    "mi_segment_slots": (_i32, [_p, _p, _p, _i64, _p, _p]),
    "mi_axpy": (_i32, [_p, _p, _i64, _f32, _p]),
    "mi_gather_u32": (_i32, [_p, _p, _i64, _p, _p]),
    "mi_dense_apply": (_i32, [_p, _p, _p, _p, _i64, C.POINTER(OptHparams), _p]),
    "mi_sparse_apply": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _i32, _i32,
                               C.POINTER(OptHparams), _i32, _i64, _i64, _p]),
    "mi_sparse_apply_fused": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _i32,
                                     _i32, _i32, C.POINTER(OptHparams), _i32, _i64, _p]),
    "mi_dense_fwd_gathered": (_i32, [_p, _p, _p, _i32, _i32, _p, _p, _p, _i64, _i64, _i32, _i32, _f32, _u64, _amax, _i64, _p]),
    "mi_dense_bwd_weight_gathered": (_i32, [_p, _p, _p, _i32, _i32, _p, _i64, _p, _p, _i64, _i32, _p, _sz, _amax, _i64, _p]),
    "mi_selftest_sqrt": (_i32, [_u32, _i64, _p, _p]),
    "mi_selftest_div": (_i32, [_f32, _u32, _i64, _p, _p]),
    "mi_catchup_gap_keys": (_i32, [_p, _p, _p, _i64, _i32, _p, _i32, _p]),
    "mi_catchup_rows_by_gap": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _sz, _p]),
    "mi_sparse_catchup": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _f32, _f32,
                                 _f32, _i32, _i32, _i64, _p]),
    "mi_set_gemm_mode": (_i32, [_i32]),
    "mi_absmax": (_i32, [_p, _i64, _p, _p]),
    "mi_get_gemm_mode": (_i32, []),
    "mi_dense_fwd": (_i32, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _f32, _u64, _amax, _p]),
    "mi_dense_bwd_data": (_i32, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i32, _i32, _f32, _i32, _amax, _p]),
    "mi_dense_bwd_weight_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "mi_dense_bwd_data_vec_planes": (_i32, [_p, _i64, _p, _p, _i64, _f32, _p, _i64, _pl, _i64, _i32, _p, _p, _i64, _p]),
    "mi_dense_bwd_weight_planes_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "mi_dense_bwd_weight_planes": (_i32, [_p, _p, _p, _p, _i64, _i32, _i32, _p, _sz, _p, _p]),
    "mi_dense_bwd_weight": (_i32, [_p, _i64, _p, _i64, _p, _p, _i64, _i32, _i32, _p, _sz, _amax, _p]),
    "mi_dense_bwd_weight_planes_batch_workspace_bytes": (_sz, [_p, _i32, _i64]),
    "mi_dense_bwd_weight_planes_batch": (_i32, [_p, _i32, _i64, _p, _sz, _p]),
    "mi_planes_bytes": (_sz, [_i64, _i32]),
    "mi_split_rows": (_i32, [_p, _i64, _i64, _i32, _i32, _pl, _p, _p]),
    "mi_split_weights": (_i32, [_p, _p, _i32, _p, _p]),
    "mi_merge_rows": (_i32, [_pl, _i64, _i32, _p, _i64, _p]),
    "mi_dense_fwd_planes": (_i32, [_pl, _pl, _p, _p, _i64, _pl, _i64, _i32, _i32, _i32, _f32, _u64, _p, _p, _i64, _p]),
    "mi_dense_bwd_data_planes": (_i32, [_pl, _pl, _pl, _p, _i64, _pl, _i64, _i32, _i32, _f32, _p, _p, _i64, _p]),
    "mi_embed_fm_planes_fwd": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _pl, _p, _p, _i32, _i32, _i64, _p]),
    "mi_logits_head_fused_workspace_bytes": (_sz, [_i64, _i32]),
    "mi_logits_head_fused": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i32, _f32, _p, _i64, _f32, _p, _p, _p, _p, _p, _p, _p, _pl,
                                    _p, _i64, _p, _p, _sz, _p]),
    "mi_hidden_logits_head_fused_workspace_bytes": (_sz, [_i64, _i32]),
    "mi_hidden_logits_head_fused": (_i32, [_pl, _pl, _p, _i64, _i32, _i32, _i32, _f32, _u64, _p, _p, _p, _p, _p, _p, _f32, _p, _p, _p, _p,
                                           _p, _p, _p, _pl, _p, _p, _sz, _p]),
    "mi_head_workspace_bytes": (_sz, [_i64]),
    "mi_sigmoid_ce_head": (_i32, [_p, _p, _p, _p, _p, _i64, _f32, _p, _p, _p, _p, _p, _sz, _p]),
    "mi_colsum_workspace_bytes": (_sz, [_i64, _i32]),
    "mi_colsum": (_i32, [_p, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "mi_layer_stats_workspace_bytes": (_sz, [_i64]),
    "mi_layer_stats": (_i32, [_p, _i64, _p, _p, _sz, _p]),
    "mi_binary_predictions": (_i32, [_p, _p, _i64, _p, _p, _p, _p, _p]),
    "mi_layer_histogram": (_i32, [_p, _i64, _p, _i32, _p, _p, _p]),
    "mi_eval_accumulate": (_i32, [_p, _p, _i64, _p, _p, _p, _p]),
}

_lib = None


def load():
    """Load the shared library once; raises MiError if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MiError("libmi355x_rec.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.mi_abi_version() != ABI_VERSION:
        raise MiError("ABI version %d != expected %d" % (lib.mi_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise MiError("%s failed (%d): %s" % (what, rc, load().mi_last_error().decode()))


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return t.data_ptr()
    return t.ctypes.data


def cur_stream():
    return torch.cuda.current_stream().cuda_stream


def hip_runtime_paths():
    """Every libamdhip64 mapped into this process — must be exactly one (see module docstring)."""
    out = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                out.add(line.split()[-1])
    return sorted(out)
