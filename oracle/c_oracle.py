"""ctypes binding of oracle/liboracle_spmm.so.  TEST INFRASTRUCTURE ONLY
(tests/, __graft_entry__.smoke(), bench.py cpu_baseline) -- see spmm_ref.c."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_spmm.so")
_lib = None

REDUCE = {"sum": 0, "add": 0, "mean": 1, "max": 2}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "spmm_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_spmm.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(C.c_int(n))


def spmm_csr_f32(rowptr, col, val, x, reduce="sum"):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, F = len(rowptr) - 1, x.shape[1]
    y = np.empty((n, F), dtype=np.float32)
    lib().oracle_spmm_csr_f32(_p(rowptr), _p(col), _p(val), C.c_int64(n), _p(x), C.c_int64(F),
                              _p(y), C.c_int64(F), C.c_int64(F), C.c_int(REDUCE[reduce]))
    return y


def spmm_csr_f64acc(rowptr, col, val, x, want_bound=False):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, F = len(rowptr) - 1, x.shape[1]
    y = np.empty((n, F), dtype=np.float64)
    ab = np.empty((n, F), dtype=np.float64) if want_bound else None
    lib().oracle_spmm_csr_f64acc(_p(rowptr), _p(col), _p(val), C.c_int64(n), _p(x), C.c_int64(F),
                                 _p(y), C.c_int64(F), C.c_int64(F), _p(ab))
    return (y, ab) if want_bound else y


def khop_csr_f32(rowptr, col, val, x, K):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, F = x.shape
    y = np.empty_like(x)
    tmp = np.empty_like(x)
    lib().oracle_khop_csr_f32(_p(rowptr), _p(col), _p(val), C.c_int64(n), _p(x), _p(y), _p(tmp),
                              C.c_int64(F), C.c_int(K))
    return y


def gcn_norm_coo_f32(row, col, w, N, flow="source_to_target"):
    row = np.ascontiguousarray(row, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    E = len(row)
    deg = np.empty(N, dtype=np.float32)
    out = np.empty(E, dtype=np.float32)
    lib().oracle_gcn_norm_coo_f32(_p(row), _p(col), _p(w), C.c_int64(E), C.c_int64(N),
                                  C.c_int(0 if flow == "source_to_target" else 1), _p(deg), _p(out))
    return out
