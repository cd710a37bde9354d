"""Shared test helpers: seeded test graphs and the fp32 tolerance used for the SpMM parity checks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import disco_oracle as orc  # noqa: E402

# north_star: "fp32 relative error <= 1e-5 per element".  Where outputs cancel towards 0 an
# element-relative bound is ill posed, so the check is the dot-product forward
# error form  |y - y64| <= RTOL * sum_e |w_e * x_e|  against a float64 oracle.
RTOL = 1e-5


def assert_fwd_close(y, y64, bound, rtol=RTOL, what=""):
    y = np.asarray(y, dtype=np.float64)
    err = np.abs(y - y64)
    lim = rtol * bound + 1e-37
    bad = err > lim
    assert not bad.any(), f"{what}: {bad.sum()} elements exceed {rtol} * sum|w x| (worst ratio {np.max(err / np.maximum(bound, 1e-300)):.3e})"


def random_csr(n_rows, n_cols, avg_deg, seed, hubs=(), empty_frac=0.1, weighted=True, sort_cols=True, dup=False):
    """Seeded CSR with chosen hub rows (list of (row, nnz)), a share of empty rows and optional duplicates."""
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, n_rows).astype(np.int64)
    deg[rng.random(n_rows) < empty_frac] = 0
    for r, d in hubs:
        deg[r] = d
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    nnz = int(rowptr[-1])
    col = rng.integers(0, n_cols, nnz, dtype=np.int64)
    if dup and nnz > 4:
        col[1::7] = col[0:-1:7][: len(col[1::7])]
    if sort_cols:
        for r in range(n_rows):
            col[rowptr[r]:rowptr[r + 1]].sort()
    val = rng.standard_normal(nnz).astype(np.float32) if weighted else None
    return rowptr, col, val


def csr_to_coo(rowptr, col):
    row = np.repeat(np.arange(len(rowptr) - 1, dtype=np.int64), np.diff(rowptr))
    return row, col
