"""ctypes binding of libdisco_hip.so (the C ABI declared in include/disco_hip.h).

There is NO fallback: if the library is missing or a call fails this module raises.  The
library must have been built beforehand (``python -m disco_amd.build`` or
``__graft_entry__.build()``); it is never compiled at import time.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the process's HIP runtime before our library binds to it)

from .build import lib_path

_lib = None

c_void_p, c_int, c_int64, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_size_t


class ScheduleInfo(C.Structure):
    """Mirror of ``disco_schedule_info`` (include/disco_hip.h)."""
    _fields_ = [(n, C.c_int64) for n in (
        "n_rows", "nnz", "hub_threshold", "hub_piece", "n_hub_rows", "n_hub_pieces", "n_hub_partials",
        "n_multi_rows", "max_row_nnz", "off_perm", "off_piece_row", "off_piece_begin", "off_piece_end",
        "off_piece_slot", "off_multi_row", "off_multi_first", "ws_bytes",
        "sweep_windows", "sweep_window_rows", "sweep_count_padded", "sweep_n_cols")] + [("sweep_ws", C.c_void_p), ("n_cols", C.c_int64), ("xcd_run", C.c_int64)]


_SIGNATURES = {
    "disco_abi_version": (c_int, []),
    "disco_status_string": (C.c_char_p, [c_int]),
    "disco_last_hip_error": (c_int, []),
    "disco_set_tuning": (c_int, [C.c_char_p, c_int64]),
    "disco_schedule_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "disco_schedule_build": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_size_t,
                                     C.POINTER(ScheduleInfo), c_void_p]),
    "disco_schedule_grouped_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "disco_schedule_build_grouped": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_size_t,
                                             C.POINTER(ScheduleInfo), c_void_p]),
    "disco_locality_workspace_bytes": (c_size_t, [c_int64]),
    "disco_locality_labels": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "disco_sweep_workspace_bytes": (c_size_t, [C.POINTER(ScheduleInfo), c_int64, c_int64]),
    "disco_sweep_build": (c_int, [c_void_p, c_void_p, c_void_p, C.POINTER(ScheduleInfo), c_int64, c_int64, c_void_p, c_size_t,
                                  c_void_p]),
    "disco_spmm_workspace_bytes": (c_size_t, [C.POINTER(ScheduleInfo), c_int64]),
    "disco_spmm_csr_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_int64, c_int, c_void_p, C.POINTER(ScheduleInfo), c_void_p, c_size_t, c_void_p]),
    "disco_spmm_csr_f32_noalloc": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                           c_int64, c_int, c_void_p, C.POINTER(ScheduleInfo), c_void_p]),
    "disco_khop_csr_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p,
                                   c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p,
                                   C.POINTER(ScheduleInfo), c_void_p, c_size_t, c_void_p]),
    "disco_csr_rowsum_f64": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "disco_gcn_norm_csr_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "disco_degree_count_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "disco_degree_count_sorted_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "disco_degree_count_sorted_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "disco_degree_ordered_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "disco_inv_sqrt_f32": (c_int, [c_void_p, c_int64, c_void_p]),
    "disco_gcn_norm_coo_apply_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "disco_pairscore_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "disco_pairscore_prepare_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64] + [c_void_p] * 14 +
                                    [C.c_float, c_void_p, c_size_t, c_void_p]),
    "disco_pairscore_block_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                          c_int, c_void_p]),
    "disco_pairscore_edges_f32": (c_int, [c_void_p, c_int64, c_int64, C.c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                          c_void_p]),
    "disco_sym_threshold_count": (c_int, [c_void_p, c_int64, c_int64, C.c_float, c_int, c_void_p, c_void_p]),
    "disco_sym_threshold_fill": (c_int, [c_void_p, c_int64, c_int64, C.c_float, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                         c_void_p]),
    "disco_coo_to_csr_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "disco_coo_to_csr": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                 C.POINTER(c_int64), c_void_p, c_size_t, c_void_p]),
    "disco_csr_insert_diagonal_workspace_bytes": (c_size_t, [c_int64]),
    "disco_csr_insert_diagonal": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                          C.POINTER(c_int64), c_void_p, c_size_t, c_void_p]),
    "disco_gather_rows_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "disco_anchor_max_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                     c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "disco_pair_gather_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                      c_void_p, c_int64, c_void_p, c_void_p]),
}
NOALLOC_SCRATCH_BYTES = 16 << 20
COO_SYMMETRISE, COO_DUP_KEEP, COO_DUP_SUM, COO_DUP_FIRST = 1, 0, 2, 4

# symbols that may legitimately be absent from an older build (none at the moment)
_OPTIONAL = set()


def exported_symbols():
    """Names this binding expects libdisco_hip.so to export."""
    return sorted(_SIGNATURES)


def _check_single_hip_runtime() -> None:
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    if len(paths) > 1:
        raise RuntimeError(
            "two HIP runtimes are mapped into this process (%s): libdisco_hip.so must bind to the "
            "runtime torch loaded, otherwise device pointers and streams do not interoperate" % sorted(paths))


def load(path: str | None = None):
    """Load (once) and return the ctypes handle of libdisco_hip.so."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("DISCO_HIP_LIB") or lib_path()
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built. Run `python -m disco_amd.build` "
            "(or __graft_entry__.build()). disco_amd has no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if name in _OPTIONAL:
                continue
            raise RuntimeError(f"{p} does not export {name}; rebuild the extension")
        fn.restype = res
        fn.argtypes = args
    if lib.disco_abi_version() != 2:
        raise RuntimeError("libdisco_hip.so ABI version mismatch")
    _check_single_hip_runtime()
    if path is None:
        _lib = lib
    return lib


class DiscoError(RuntimeError):
    pass


def check(status: int, what: str) -> None:
    if status != 0:
        lib = load()
        msg = lib.disco_status_string(status).decode()
        extra = ""
        if status == -4:
            extra = f" (hipError {lib.disco_last_hip_error()})"
        raise DiscoError(f"{what}: {msg}{extra}")


def ptr(t) -> int | None:
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
