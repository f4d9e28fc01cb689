"""ctypes front-end of oracle/libge_oracle.so (the C fp32 restatement).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.  See ge_oracle.c for the reference citations.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libge_oracle.so")
    src = os.path.join(_HERE, "ge_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libge_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _tri(t):
    t = np.ascontiguousarray(t, dtype=np.int32)
    assert t.ndim == 2 and t.shape[1] == 3
    return t


def complex_score(table, triples, max_norm=1.0, apply_sigmoid=True, threads=1, hole=False):
    table = np.ascontiguousarray(table, dtype=np.float32)
    triples = _tri(triples)
    out = np.empty(len(triples), dtype=np.float32)
    fn = lib().oracle_hole_score if hole else lib().oracle_complex_score
    rc = fn(_p(table, C.c_float), C.c_int64(table.shape[0]), C.c_int32(table.shape[1]),
            _p(triples, C.c_int32), C.c_int64(len(triples)), C.c_float(max_norm),
            C.c_int(int(apply_sigmoid)), _p(out, C.c_float), C.c_int(threads))
    assert rc == 0, rc
    return out


def hinge_step(table, pos, neg, margin, lr, max_norm=1.0, hole=False, threads=1):
    """In-place on `table` (must be a contiguous float32 array). Returns loss[B]."""
    assert table.dtype == np.float32 and table.flags.c_contiguous
    pos, neg = _tri(pos), _tri(neg)
    loss = np.empty(len(pos), dtype=np.float32)
    rc = lib().oracle_hinge_step(_p(table, C.c_float), C.c_int64(table.shape[0]), C.c_int32(table.shape[1]),
                                 _p(pos, C.c_int32), _p(neg, C.c_int32), C.c_int64(len(pos)),
                                 C.c_float(margin), C.c_float(lr), C.c_float(max_norm), C.c_int(int(hole)),
                                 _p(loss, C.c_float), C.c_int(threads))
    assert rc == 0, rc
    return loss


def corrupt_batch(pos, id_to_type, type_offsets, type_ids, seed, step, padded_size=1024, mode=0):
    pos = _tri(pos)
    id_to_type = np.ascontiguousarray(id_to_type, dtype=np.int32)
    type_offsets = np.ascontiguousarray(type_offsets, dtype=np.int64)
    type_ids = np.ascontiguousarray(type_ids, dtype=np.int32)
    neg = np.empty_like(pos)
    rc = lib().oracle_corrupt_batch(_p(pos, C.c_int32), C.c_int64(len(pos)), _p(id_to_type, C.c_int32),
                                    C.c_int64(len(id_to_type)), _p(type_offsets, C.c_int64),
                                    C.c_int32(len(type_offsets) - 1), _p(type_ids, C.c_int32),
                                    C.c_uint64(seed), C.c_uint64(step), C.c_int32(padded_size),
                                    C.c_int32(mode), _p(neg, C.c_int32))
    assert rc == 0, rc
    return neg


def complex_score_1vK(table, hr, cand, max_norm=1.0, apply_sigmoid=True, cand_is_head=False, threads=1):
    table = np.ascontiguousarray(table, dtype=np.float32)
    hr = np.ascontiguousarray(hr, dtype=np.int32)
    cand = np.ascontiguousarray(cand, dtype=np.int32)
    out = np.empty((len(hr), len(cand)), dtype=np.float32)
    rc = lib().oracle_complex_score_1vK(_p(table, C.c_float), C.c_int64(table.shape[0]), C.c_int32(table.shape[1]),
                                        _p(hr, C.c_int32), C.c_int64(len(hr)), _p(cand, C.c_int32),
                                        C.c_int64(len(cand)), C.c_float(max_norm), C.c_int(int(apply_sigmoid)),
                                        C.c_int(int(cand_is_head)), _p(out, C.c_float), C.c_int(threads))
    assert rc == 0, rc
    return out


def num_threads() -> int:
    return int(lib().oracle_num_threads())
