"""The C-ABI library loads and exports every symbol include/disco_hip.h declares (no compute)."""
import ctypes
import os
import re

from helpers import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "disco_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(disco_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = declared_symbols()
    for must in ("disco_spmm_csr_f32", "disco_khop_csr_f32", "disco_schedule_build", "disco_gcn_norm_csr_f32",
                 "disco_gcn_norm_coo_apply_f32", "disco_degree_count_f32"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from disco_amd.build import lib_path
    so = lib_path()
    assert os.path.exists(so), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(so)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/disco_hip.h but not exported"
    assert lib.disco_abi_version() == 2
    lib.disco_status_string.restype = ctypes.c_char_p
    assert lib.disco_status_string(0) == b"ok"
    assert lib.disco_status_string(-3) == b"workspace too small"


def test_python_binding_covers_header():
    from disco_amd import _lib
    bound = set(_lib.exported_symbols())
    for name in declared_symbols():
        assert name in bound, f"{name} has no ctypes signature in disco_amd/_lib.py"


def test_argument_validation_without_gpu():
    # entry points reject bad arguments before touching the device
    from disco_amd import _lib
    lib = _lib.load()
    assert lib.disco_spmm_csr_f32(None, None, None, -1, None, 0, None, 0, 4, 0, None, None, None, 0, None) == -1
    assert lib.disco_spmm_csr_f32(None, None, None, 0, None, 0, None, 0, 4, 0, None, None, None, 0, None) == 0
    assert lib.disco_spmm_csr_f32(None, None, None, 5, None, 4, None, 4, 4, 0, None, None, None, 0, None) == -1
    assert lib.disco_khop_csr_f32(None, None, None, 5, 0, None, 4, None, 4, None, None, 4, 4, 1, None, None, None, 0, None) == -1
    assert lib.disco_set_tuning(b"nope", 1) == -1


def test_argument_validation_of_the_set_up_entry_points():
    import ctypes as C
    from disco_amd import _lib
    lib = _lib.load()
    nnz = C.c_int64(7)
    # negative sizes, missing outputs, unknown duplicate mode, symmetrise on a rectangular shape
    assert lib.disco_coo_to_csr(None, None, None, -1, 4, 4, 0, None, None, None, C.byref(nnz), None, 0, None) == -1
    assert lib.disco_coo_to_csr(None, None, None, 3, 4, 4, 0, None, None, None, None, None, 0, None) == -1
    assert lib.disco_coo_to_csr(None, None, None, 3, 4, 4, 6, 1, None, None, C.byref(nnz), None, 0, None) == -1
    assert lib.disco_coo_to_csr(None, None, None, 3, 4, 5, 1, 1, None, None, C.byref(nnz), None, 0, None) == -1
    assert lib.disco_coo_to_csr(None, None, None, 3, 4, 2 ** 31, 0, 1, None, None, C.byref(nnz), None, 0, None) == -2
    assert lib.disco_coo_to_csr_workspace_bytes(-1, 0, 0) == 0
    # pair gather: row pitch smaller than the width, output pitch smaller than two halves, empty work
    assert lib.disco_pair_gather_f32(None, 10, 4, 8, None, None, None, 5, 5, None, 16, None, None) == -1
    assert lib.disco_pair_gather_f32(None, 10, 8, 8, None, None, None, 5, 5, None, 15, None, None) == -1
    assert lib.disco_pair_gather_f32(None, 10, 8, 8, None, None, None, 5, 0, None, 16, None, None) == 0
    assert lib.disco_pair_gather_f32(None, 10, 8, 8, None, None, None, 5, 5, None, 16, None, None) == -1
    # scorer: unsupported hidden width
    assert lib.disco_pairscore_workspace_bytes(-1, 256) == 0


def test_tuning_defaults_of_the_test_fixture_cover_every_key():
    """tests/conftest.py restores every tuning key after each GPU test: its list must name exactly the keys
    disco_set_tuning accepts, and every one of them must be accepted by the built library."""
    import re
    from conftest import _TUNING_DEFAULTS
    src = open(os.path.join(ROOT, "disco_amd", "csrc", "disco_norm.hip")).read()
    keys = set(re.findall(r'strcmp\(key, "(\w+)"\)', src))
    assert keys == set(_TUNING_DEFAULTS), keys ^ set(_TUNING_DEFAULTS)
    from disco_amd.build import lib_path
    lib = ctypes.CDLL(lib_path())
    lib.disco_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    for k, v in _TUNING_DEFAULTS.items():
        assert lib.disco_set_tuning(k.encode(), v) == 0
