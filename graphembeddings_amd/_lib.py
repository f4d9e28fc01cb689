"""ctypes binding of libge_hip.so (include/ge_hip.h).

Loaded the way the reference loads its own native library (transE.py:9-10:
``ctypes.cdll.LoadLibrary("./init.so")``), with raw buffer addresses passed as integers
(transE.py:95-112).  There is NO CPU fallback: if the library is missing or no MI355X is visible
the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libge_hip.so")

GE_EINVAL, GE_ENOTSUP, GE_ENOMEM = -22, -95, -12

# every symbol include/ge_hip.h declares: (restype, argtypes)
_i32, _i64, _u64, _f, _p, _sz = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p, C.c_size_t
SYMBOLS = {
    "ge_version": (C.c_int, []),
    "ge_max_dim": (C.c_int, []),
    "ge_complex_score": (C.c_int, [_p, _i64, _i32, _p, _i64, _f, C.c_int, _p, _p]),
    "ge_complex_score_strided": (C.c_int, [_p, _i64, _i32, _i64, _p, _i64, _f, C.c_int, _p, _p]),
    "ge_complex_logloss": (C.c_int, [_p, _i64, _i32, _p, _i64, _f, _f, _f, _p, _p, _sz, _p]),
    "ge_hole_score": (C.c_int, [_p, _i64, _i32, _p, _i64, _f, C.c_int, _p, _p]),
    "ge_hole_to_spectral": (C.c_int, [_p, _i64, _i32, _p]),
    "ge_hole_from_spectral": (C.c_int, [_p, _i64, _i32, _p]),
    "ge_hole_spectral_score": (C.c_int, [_p, _i64, _i32, _p, _i64, _f, C.c_int, _p, _p]),
    "ge_hinge_loss": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _f, _f, C.c_int, _p, _p, _p]),
    "ge_hinge_step_workspace_bytes": (_sz, [_i64, _i32]),
    "ge_complex_hinge_step": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _f, _f, _f, _p, _p, _sz, _p]),
    "ge_hole_hinge_step": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _f, _f, _f, _p, _p, _sz, _p]),
    "ge_logloss_step_workspace_bytes": (_sz, [_i64, _i32]),
    "ge_complex_logloss_step": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _f, _f, _f, _p, _p, _sz, _p]),
    "ge_hinge_grad": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _f, _f, _f, C.c_int, _p, _p, _p, _p]),
    "ge_scatter_add_rows": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _p]),
    "ge_gather_rows": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p]),
    "ge_corrupt_batch": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _p, _u64, _u64, _i32, _i32, _p, _p]),
    "ge_bernoulli_corrupt_batch": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _p, _i32, _i32, _i32, _u64, _u64, _p, _p]),
    "ge_complex_score_1vK": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _i64, _f, C.c_int, C.c_int, _p, _p]),
    "ge_validation_workspace_bytes": (C.c_size_t, [_i64]),
    "ge_validation_tick": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _p, _p, _i32, _p, C.c_uint64, C.c_uint64, _i32, _i32,
                                      _f, _f, C.c_int, _p, C.c_size_t, _p, _p, _p, _p]),
    "ge_validation_logloss_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "ge_validation_tick_logloss": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _p, _p, _i32, _p, C.c_uint64, C.c_uint64, _i32, _i32,
                                             _i32, _f, _f, _p, C.c_size_t, _p, _p, _p, _p]),
    "ge_rank_max_dim": (C.c_int, []),
    "ge_complex_rank_1vK": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p, _i64, _f, C.c_int, _p, _p, _p, _p, _p, _p, _p]),
    "ge_rank_1vK": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p, _i64, _f, C.c_int, C.c_int, _p, _p, _p, _p, _p, _p, _p]),
    "ge_known_cells": (C.c_int, [C.c_int, _p, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _p, _p, _p, _p]),
    "ge_rank_planes_bytes": (_i64, [_i64, _i32, _i64]),
    "ge_rank_planes": (C.c_int, [_p, _i64, _i32, _p, _i64, _f, C.c_int, _p, _p]),
    "ge_rank_1vK_planes": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p, _i64, _f, C.c_int, C.c_int, _p, _p, _p, _p, _p, _p, _p, _p]),
    "ge_rank_1vK_vs_loss": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p, _p, _i64, _f, C.c_int, C.c_int, _p, _p, _p, _p, _p, _p]),
    "ge_train_workspace_bytes": (_sz, [_i64, _i32]),
    "ge_train_pipeline_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ge_train_pipeline_reset": (C.c_int, [_p]),
    "ge_train_pipeline_destroy": (C.c_int, [_p]),
    "ge_train_steps": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _i64, _i64, _p, _p, _i32, _p, _u64, _u64, _i32,
                                 _i32, _f, _f, _f, _f, _f, C.c_int, _p, C.c_int, _p, _p, _sz, _p, C.c_int, _p, _p]),
    "ge_train_logloss_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "ge_train_steps_logloss": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _i64, _i64, _p, _p, _i32, _p, _u64, _u64, _i32,
                                         _i32, _i32, _f, _f, _f, _f, _f, _p, C.c_int, _p, _p, _sz, _p, _p]),
    "ge_train_prepared_layout": (C.c_int, [_i64, C.POINTER(C.c_int64)]),
    "ge_train_prepare_bytes": (_sz, [_i64, _i64]),
    "ge_train_prepare_steps": (C.c_int, [_p, _i64, _i64, _i64, _i64, _p, _i64, _p, _i32, _p, _u64, _u64, _i32, _i32,
                                         C.c_int, _p, _sz, _p]),
    "ge_shard_plan_workspace_bytes": (_sz, [_i64, _i64]),
    "ge_shard_plan": (C.c_int, [_p, _p, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _sz, _i32, _p]),
    "ge_shard_grad": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p, _p, _i64, _i64, _i32, _f, _f, _f, C.c_int, _p, _p, _p, _p, _p, _p]),
    "ge_shard_apply": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _i32, _p, _p, _p, _p]),
    "ge_shard_owner_record_words": (_i64, [_i64]),
    "ge_shard_owner_workspace_bytes": (_sz, [_i64, _i64]),
    "ge_shard_owner_plan": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _p, _sz, _p]),
    "ge_shard_owner_apply": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p]),
    "ge_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ge_event_destroy": (C.c_int, [_p]),
    "ge_event_record": (C.c_int, [_p, _p]),
    "ge_event_synchronize": (C.c_int, [_p]),
    "ge_event_elapsed_ms": (C.c_int, [_p, _p, C.POINTER(C.c_float)]),
}


class GeError(RuntimeError):
    def __init__(self, fn: str, code: int):
        self.code = code
        if code == GE_EINVAL:
            what = "invalid argument (dimension, null pointer or misaligned buffer)"
        elif code == GE_ENOTSUP:
            what = "embedding_dim outside the compiled kernel range"
        elif code == GE_ENOMEM:
            what = "workspace too small"
        else:
            what = f"hipError_t {code}"
        super().__init__(f"{fn} failed: {what}")


_lib = None


def load():
    """Load libge_hip.so; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and must be
        # loaded FIRST so that this library binds to the same runtime instance (same soname); the
        # stream handles and device pointers we are handed belong to it.  Loaded the other way round
        # the process ends up with two runtimes and the second one finds no device.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found. Build it with `python -m graphembeddings_amd.build` "
                "(hipcc --offload-arch=gfx950). graphembeddings_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the header diverge
            fn.restype = res
            fn.argtypes = args
        if lib.ge_version() < 320:
            raise RuntimeError("libge_hip.so is older than the Python host expects")
        _lib = lib
    return _lib


def call(name: str, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise GeError(name, rc)
