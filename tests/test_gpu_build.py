"""CSR construction from raw COO edges on the device against the oracle's scipy / numpy
restatements of the reference's set-up code (utils.py:99-106, LargeScaleCondensing.py:486).
Index work is bit exact; summed duplicate weights are exact for the reference's unit weights and within
one float32 rounding of the float64 sum for arbitrary weights."""
import numpy as np
import pytest
import torch

from helpers import orc

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _edges(n, e, seed, dup_share=0.2):
    rng = np.random.default_rng(seed)
    r = rng.integers(0, n, e)
    c = rng.integers(0, n, e)
    k = int(e * dup_share)
    if k:
        src = rng.integers(0, e, k)
        r = np.concatenate([r, r[src]])
        c = np.concatenate([c, c[src]])
    p = rng.permutation(r.size)
    return np.stack([r[p], c[p]]).astype(np.int64)


@pytest.mark.parametrize("n,e", [(1, 1), (7, 30), (1000, 20000), (70000, 300000), (5, 0)])
def test_keep_mode_is_a_stable_row_major_sort(n, e):
    from disco_amd.sparse import coo_to_csr
    ei = _edges(n, e, seed=n + e)
    val = np.random.default_rng(1).standard_normal(ei.shape[1]).astype(np.float32)
    rp, c, v = coo_to_csr(torch.from_numpy(ei[0]).to(dev()), torch.from_numpy(ei[1]).to(dev()),
                          torch.from_numpy(val).to(dev()), (n, n), dup="keep")
    order = np.lexsort((np.arange(ei.shape[1]), ei[1], ei[0]))      # stable in the input position
    ref_rp, ref_c, _ = orc.coo_to_csr(ei[0], ei[1], None, n)
    assert np.array_equal(rp.cpu().numpy(), ref_rp)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), ref_c)
    assert np.array_equal(v.cpu().numpy(), val[order])


@pytest.mark.parametrize("undirected", [False, True])
@pytest.mark.parametrize("n,e", [(9, 40), (3000, 50000), (120000, 400000)])
def test_pyg2dpr_adjacency_matches_scipy(n, e, undirected):
    from disco_amd import utils
    ei = _edges(n, e, seed=3 * n + e)
    adj = utils.adjacency_from_edge_index(torch.from_numpy(ei).to(dev()), n, undirected=undirected)
    rp, col, data = orc.pyg2dpr_adjacency(ei, n, undirected)
    grp, gcol, gval = adj.csr()
    assert np.array_equal(grp.cpu().numpy(), rp)
    assert np.array_equal(gcol.cpu().numpy(), col)
    assert np.array_equal(gval.cpu().numpy().astype(np.float64), data)      # small integer counts: exact
    if not undirected:
        assert data.max() >= 2, "the generator must produce duplicate edges"


def test_to_undirected_matches_oracle_and_rejects_bad_indices():
    from disco_amd import _lib, utils
    n = 5000
    ei = _edges(n, 60000, seed=9)
    got = utils.to_undirected(torch.from_numpy(ei).to(dev()), None, n).cpu().numpy()
    assert np.array_equal(got, orc.to_undirected(ei, n))
    bad = ei.copy()
    bad[1, 17] = n
    with pytest.raises(_lib.DiscoError):
        utils.to_undirected(torch.from_numpy(bad).to(dev()), None, n)


def test_weighted_duplicates_fold_in_float64():
    from disco_amd.sparse import coo_to_csr
    n = 400
    ei = _edges(n, 30000, seed=4, dup_share=1.0)
    val = (np.random.default_rng(2).standard_normal(ei.shape[1]) * 10).astype(np.float32)
    rp, c, v = coo_to_csr(torch.from_numpy(ei[0]).to(dev()), torch.from_numpy(ei[1]).to(dev()),
                          torch.from_numpy(val).to(dev()), (n, n), dup="sum")
    import scipy.sparse as sp
    m = sp.csr_matrix((val.astype(np.float64), (ei[0], ei[1])), shape=(n, n))
    m.sum_duplicates()
    m.sort_indices()
    assert np.array_equal(rp.cpu().numpy(), m.indptr) and np.array_equal(c.cpu().numpy(), m.indices)
    got = v.cpu().numpy().astype(np.float64)
    absum = sp.csr_matrix((np.abs(val).astype(np.float64), (ei[0], ei[1])), shape=(n, n))
    absum.sum_duplicates()
    absum.sort_indices()
    # one rounding of the float64 sum (whose own order-dependent error is ~1e-16 of the absolute sum)
    assert np.all(np.abs(got - m.data) <= 2.0 ** -24 * np.abs(m.data) + 1e-13 * absum.data)
    # first-wins mode keeps the first value in input order
    rp2, c2, v2 = coo_to_csr(torch.from_numpy(ei[0]).to(dev()), torch.from_numpy(ei[1]).to(dev()),
                             torch.from_numpy(val).to(dev()), (n, n), dup="first")
    key = ei[0] * n + ei[1]
    _, first = np.unique(key, return_index=True)
    assert np.array_equal(v2.cpu().numpy(), val[first]) and np.array_equal(c2.cpu().numpy(), c.cpu().numpy())


def test_end_to_end_raw_edges_to_normalised_adj_t():
    """raw edge list -> Pyg2Dpr adjacency -> normalize_adj -> .t(): the whole set-up of
    LargeScaleCondensing.py:478-488 on the device against scipy."""
    from disco_amd import utils
    n = 20000
    ei = _edges(n, 150000, seed=12)
    adj = utils.adjacency_from_edge_index(torch.from_numpy(ei).to(dev()), n, undirected=True)
    adj_t = utils.normalize_adj_csr(adj).t()
    und = orc.to_undirected(ei, n)
    r, c, v = orc.normalize_adj_scipy(und[0], und[1], np.ones(und.shape[1], np.float32), n)
    rp, col, val = orc.adj_t_from_coo(r, c, v, n)
    grp, gcol, gval = adj_t.csr()
    assert np.array_equal(grp.cpu().numpy(), rp) and np.array_equal(gcol.cpu().numpy(), col)
    g, w = gval.cpu().numpy(), val.astype(np.float32)
    ulp = np.abs(g.view(np.int32).astype(np.int64) - w.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1


@pytest.mark.parametrize("n,avg", [(1, 0), (50, 3), (4000, 12), (90000, 6)])
def test_diagonal_insertion_equals_the_sort_based_merge(n, avg):
    """disco_csr_insert_diagonal (merge on sorted columns) against the same structure obtained by appending a
    zero diagonal and sorting again: identical rowptr / col / val, incl. empty rows and rows that already hold
    their diagonal entry."""
    from disco_amd import SparseTensor
    from disco_amd.sparse import coo_to_csr
    from disco_amd.utils import _with_diagonal
    rng = np.random.default_rng(n + avg)
    e = n * avg
    r = rng.integers(0, n, e)
    c = rng.integers(0, n, e)
    keep = rng.random(n) < 0.3                       # 30 % of the nodes get an explicit self-loop
    r = np.concatenate([r, np.flatnonzero(keep)])
    c = np.concatenate([c, np.flatnonzero(keep)])
    key = np.unique(r * n + c)
    r, c = key // n, key % n
    v = rng.standard_normal(r.size).astype(np.float32)
    adj = SparseTensor(row=torch.from_numpy(r).to(dev()), col=torch.from_numpy(c).to(dev()), value=torch.from_numpy(v).to(dev()),
                       sparse_sizes=(n, n), is_sorted=True)
    row, col, val = adj.coo()
    got = _with_diagonal(adj, row, col, val)
    d = torch.arange(n, device=dev())
    rp, c32, vv = coo_to_csr(torch.cat([row, d]), torch.cat([col, d]), torch.cat([val, torch.zeros(n, device=dev())]), (n, n), dup="sum")
    assert torch.equal(got._rowptr, rp) and torch.equal(got._col32, c32) and torch.equal(got._value, vv)
    assert got.nnz() == r.size + n - int(np.unique(r[r == c]).size)
