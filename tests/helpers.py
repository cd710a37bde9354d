"""Shared test helpers: seeded test graphs and the fp32 tolerance used for the SpMM parity checks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import disco_oracle as orc  # noqa: E402

# north_star: "fp32 relative error <= 1e-5 per element".  Two checks against a float64 oracle:
#   (1) everywhere, the dot-product forward-error form  |y - y64| <= RTOL * sum_e |w_e * x_e|  (an element-relative
#       bound is ill posed where an output cancels towards 0);
#   (2) the plain element-relative form  |y - y64| <= RTOL * |y64|  on every element that is NOT tiny against its own
#       terms, |y64| >= REL_FLOOR * sum_e |w_e * x_e|  (no cancellation worth the name: north_star's wording applies as is).
RTOL = 1e-5
REL_FLOOR = 0.05


def assert_fwd_close(y, y64, bound, rtol=RTOL, what="", rel_floor=REL_FLOOR):
    y = np.asarray(y, dtype=np.float64)
    y64 = np.asarray(y64, dtype=np.float64)
    err = np.abs(y - y64)
    lim = rtol * bound + 1e-37
    bad = ~(err <= lim)                    # NaN / inf in y compare false with everything: they must count as failures
    assert not bad.any(), f"{what}: {bad.sum()} elements exceed {rtol} * sum|w x| (worst ratio {np.max(err / np.maximum(bound, 1e-300)):.3e})"
    big = (np.abs(y64) >= rel_floor * bound) & (bound > 0)
    bad_rel = big & ~(err <= rtol * np.abs(y64) + 1e-37)
    assert not bad_rel.any(), (f"{what}: {bad_rel.sum()} of {big.sum()} non-cancelling elements exceed the element-relative "
                               f"{rtol} (worst {np.max(err[big] / np.abs(y64[big])):.3e})")


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
