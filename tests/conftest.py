import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# library-wide tuning keys and their defaults (disco_amd/csrc/disco_internal.h, struct Tuning): a test that changes one
# and fails before its own `finally`, or forgets a key, must not change what every later test in the process runs
_TUNING_DEFAULTS = {"force_g": 0, "force_nv": 0, "max_vec": 0, "sweep_mode": 2, "win_cols": 0, "win_rpg": 0, "sweep_blocks_per_cu": 0,
                    "psync_pct": 0, "psync_polls": 0, "addr24": 1, "row_cols": 0, "narrow_groups": 1, "stream_u": 0, "stream_lead": 0,
                    "xcd_map": -1}


@pytest.fixture(autouse=True)
def _restore_tuning(request):
    yield
    if "gpu" not in request.keywords:
        return
    from disco_amd import _lib
    lib = _lib.load()
    for k, v in _TUNING_DEFAULTS.items():
        assert lib.disco_set_tuning(k.encode(), v) == 0, k
