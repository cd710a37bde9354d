"""The CPU oracle against (a) closed forms on hand-checkable graphs, (b) independent formulations
built from stock packages, (c) the committed golden vectors, and (d) for the pair scorer the vectors
produced by the reference's own PGE_Edge (tests/golden/make_scorer_golden.py).  No GPU."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from helpers import ROOT, assert_fwd_close, csr_to_coo, orc, random_csr
from oracle import c_oracle

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "graph_golden.npz"))
SCORER = np.load(os.path.join(ROOT, "tests", "golden", "scorer_golden.npz"))
GRAPHS = ["path", "star", "loop0", "dup", "hub"]


def test_torch_pow_is_inv_sqrt_bitwise():
    # the oracle restates torch's CPU pow_(-0.5) (utils.py:245) as 1/sqrt
    x = torch.cat([torch.rand(100000) * 1000, torch.tensor([0.0, 1.0, 2.0, 3.0, 1e-30, 1e30])])
    a = x.clone().pow_(-0.5).numpy()
    with np.errstate(divide="ignore"):
        b = np.float32(1) / np.sqrt(x.numpy())
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("weighted", [False, True])
def test_gcn_norm_coo_matches_literal_torch(weighted):
    # literal restatement of utils.py:242-247 with stock torch CPU ops (index_add_ == serial scatter_add)
    rng = np.random.default_rng(1)
    n, E = 300, 5000
    ei = rng.integers(0, n, (2, E))
    ei[1, ei[1] == 7] = 8            # node 7: zero in-degree -> inf -> 0
    w = rng.random(E).astype(np.float32) if weighted else None
    _, got = orc.gcn_norm_coo(ei, w, n, add_self_loops=False)
    tw = torch.ones(E) if w is None else torch.from_numpy(w)
    row, col = torch.from_numpy(ei[0]), torch.from_numpy(ei[1])
    deg = torch.zeros(n).index_add_(0, col, tw)
    dis = deg.pow_(-0.5)
    dis.masked_fill_(dis == float("inf"), 0)
    ref = dis[row] * tw * dis[col]
    assert np.array_equal(got.view(np.uint32), ref.numpy().view(np.uint32))
    c = c_oracle.gcn_norm_coo_f32(ei[0], ei[1], w, n)
    assert np.array_equal(got.view(np.uint32), c.view(np.uint32))
    assert np.all(got[ei[0] == 7] == 0)


def test_gcn_norm_add_remaining_self_loops():
    ei = np.array([[0, 1, 2, 2], [1, 0, 2, 0]])
    w = np.array([1.0, 2.0, 5.0, 3.0], dtype=np.float32)
    ei2, w2 = orc.add_remaining_self_loops(ei, w, np.float32(1.0), 3)
    assert ei2.tolist() == [[0, 1, 2, 0, 1, 2], [1, 0, 0, 0, 1, 2]]
    assert w2.tolist() == [1.0, 2.0, 3.0, 1.0, 1.0, 5.0]   # existing loop weight (5) is kept


def test_normalize_adj_closed_form_path():
    # path 0-1-2-3 (+ isolated 4), node 0 has no loop -> +I everywhere; deg = [2,3,3,2,1]
    n, ei = int(GOLD["path_n"]), GOLD["path_ei"]
    r, c, v = orc.normalize_adj_scipy(ei[0], ei[1], np.ones(ei.shape[1], np.float32), n)
    A = sp.csr_matrix((v, (r, c)), shape=(n, n)).toarray()
    d = np.array([2, 3, 3, 2, 1], dtype=np.float64)
    exp = np.zeros((n, n))
    for i, j in [(0, 1), (1, 2), (2, 3)]:
        exp[i, j] = exp[j, i] = 1 / np.sqrt(d[i] * d[j])
    for i in range(n):
        exp[i, i] = 1 / d[i]
    assert np.allclose(A, exp.astype(np.float32), rtol=0, atol=1e-7)
    assert A[4, 4] == 1.0


def test_normalize_adj_node0_selfloop_rule():
    # utils.py:425-426: node 0 has a self loop -> NO identity is added, nodes 2,3 keep zero diagonals
    n, ei, w = int(GOLD["loop0_n"]), GOLD["loop0_ei"], GOLD["loop0_w"]
    r, c, v = orc.normalize_adj_scipy(ei[0], ei[1], w, n)
    A = sp.csr_matrix((v, (r, c)), shape=(n, n)).toarray()
    assert A[2, 2] == 0 and A[3, 3] == 0
    assert np.isclose(A[0, 0], 1 / 2) and np.isclose(A[0, 1], 1 / np.sqrt(2 * 1))
    assert np.isclose(A[2, 3], 2 / np.sqrt(2 * 2))


def test_normalize_adj_sums_duplicates():
    n, ei, w = int(GOLD["dup_n"]), GOLD["dup_ei"], GOLD["dup_w"]
    r, c, v = orc.normalize_adj_scipy(ei[0], ei[1], w, n)
    A = sp.csr_matrix((v, (r, c)), shape=(n, n)).toarray()
    rs = np.array([0.75 + 1, 2 + 1, 1.5 + 1, 3.75 + 1])
    assert np.isclose(A[0, 1], 0.75 / np.sqrt(rs[0] * rs[1]))


@pytest.mark.parametrize("name", GRAPHS)
def test_golden_normalisation(name):
    n, ei = int(GOLD[f"{name}_n"]), GOLD[f"{name}_ei"]
    w = GOLD[f"{name}_w"] if f"{name}_w" in GOLD else None
    _, w1 = orc.gcn_norm_coo(ei, w, n, add_self_loops=False)
    assert np.array_equal(w1.view(np.uint32), GOLD[f"{name}_gcn_norm_noloop"].view(np.uint32))
    ei2, w2 = orc.gcn_norm_coo(ei, w, n, add_self_loops=True)
    assert np.array_equal(ei2, GOLD[f"{name}_gcn_norm_loop_ei"])
    assert np.array_equal(w2.view(np.uint32), GOLD[f"{name}_gcn_norm_loop"].view(np.uint32))
    vals = np.ones(ei.shape[1], np.float32) if w is None else w
    r, c, v = orc.normalize_adj_scipy(ei[0], ei[1], vals, n)
    assert np.array_equal(r, GOLD[f"{name}_normadj_row"]) and np.array_equal(c, GOLD[f"{name}_normadj_col"])
    assert np.array_equal(v.view(np.uint32), GOLD[f"{name}_normadj_val"].view(np.uint32))


@pytest.mark.parametrize("name", GRAPHS)
def test_golden_propagation(name):
    n = int(GOLD[f"{name}_n"])
    rowptr, col, val = GOLD[f"{name}_adjt_rowptr"], GOLD[f"{name}_adjt_col"], GOLD[f"{name}_adjt_val"]
    for F in (1, 3, 128):
        x = GOLD[f"{name}_x{F}"]
        for K in (1, 2):
            y = orc.khop_csr(rowptr, col, val, x, K, np.float64)
            assert np.allclose(y, GOLD[f"{name}_khop{K}_F{F}_f64"], rtol=1e-13, atol=1e-300)
    x = GOLD[f"{name}_x3"]
    assert np.array_equal(orc.max_aggregate(rowptr, col, x), GOLD[f"{name}_max_F3"])


def test_spmm_formulations_agree():
    # (i) scatter form (PyG non-fused route), (ii) scipy CSR float64, (iii) torch.sparse_csr, (iv) C port
    rowptr, col, val = random_csr(400, 400, 9, seed=3, hubs=[(5, 900)], weighted=True)
    x = np.random.default_rng(4).standard_normal((400, 37)).astype(np.float32)
    y64 = orc.spmm_csr(rowptr, col, val, x, "sum", np.float64)
    bound = orc.spmm_abs_bound(rowptr, col, val, x)
    row, _ = csr_to_coo(rowptr, col)
    # edge_index convention: out[col_e] += w * x[row_e]; with edge (src=col, dst=row) this is A x
    ys = orc.propagate_scatter(np.stack([col, row]), val, x, 400, np.float64)
    assert np.allclose(ys, y64, rtol=1e-12, atol=1e-12)
    yt = (torch.sparse_csr_tensor(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), (400, 400))
          @ torch.from_numpy(x)).numpy()
    assert_fwd_close(yt, y64, bound, what="torch.sparse_csr")
    yc = c_oracle.spmm_csr_f32(rowptr, col, val, x, "sum")
    assert_fwd_close(yc, y64, bound, what="C port")
    y64c, bc = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
    assert np.allclose(y64c, y64, rtol=1e-12, atol=1e-12) and np.allclose(bc, bound, rtol=1e-12)
    # the numpy float32 sequential restatement and the C port are the same algorithm
    small = slice(0, 40)
    rp = rowptr[: 41] - rowptr[0]
    ynp = orc.spmm_csr(rp, col[: rp[-1]], val[: rp[-1]], x, "sum", np.float32)
    assert np.array_equal(ynp.view(np.uint32), yc[small].view(np.uint32))


def test_mean_and_max_semantics():
    rowptr, col, val = random_csr(120, 150, 5, seed=8, hubs=[(3, 300)], weighted=True)
    x = np.random.default_rng(9).standard_normal((150, 11)).astype(np.float32)
    ym = c_oracle.spmm_csr_f32(rowptr, col, val, x, "mean")
    ys = c_oracle.spmm_csr_f32(rowptr, col, val, x, "sum")
    cnt = np.maximum(np.diff(rowptr), 1)[:, None].astype(np.float32)
    assert np.array_equal(ym, ys / cnt)
    mx = orc.max_aggregate(rowptr, col, x)
    t = torch.zeros(120, 11).scatter_reduce(0, torch.from_numpy(csr_to_coo(rowptr, col)[0])[:, None].expand(-1, 11),
                                            torch.from_numpy(x[col]), "amax", include_self=False)
    assert np.array_equal(mx, t.numpy())
    assert np.all(mx[np.diff(rowptr) == 0] == 0)          # rows without neighbours -> 0
    assert np.array_equal(c_oracle.spmm_csr_f32(rowptr, col, None, x, "max"), mx)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_scorer_oracle_matches_reference_vectors(tag):
    # pinned by the reference's own PGE_Edge outputs
    params = {k[len(tag) + 7:]: SCORER[k] for k in SCORER.files if k.startswith(f"{tag}_param_")}
    Z, S, L = SCORER[f"{tag}_Z"], SCORER[f"{tag}_S"], SCORER[f"{tag}_L"]
    n = Z.shape[0]
    emb = np.concatenate([np.repeat(Z, n, axis=0), np.tile(Z, (n, 1))], axis=1)
    logits = orc.pge_edge_logits(params, emb, np.float64).reshape(n, n)
    assert np.allclose(logits, L, rtol=2e-5, atol=2e-5)
    assert np.allclose(orc.sigmoid(logits), S, rtol=0, atol=2e-6)
    assert np.allclose(orc.pair_score_matrix(params, Z), S, rtol=0, atol=2e-6)


def test_edge_construction_variants():
    S = SCORER["a_S"]
    n = S.shape[0]
    thr = 0.9
    for variant in ("transductive", "inductive", "sampled"):
        ei, ew = orc.edge_construction(S, thr, variant)
        A = np.zeros((n, n), np.float32)
        A[ei[0], ei[1]] = ew
        assert np.all(np.diag(A) == 1)
        off = A[~np.eye(n, dtype=bool)]
        assert np.all((off == 0) | (off >= thr))
        assert np.array_equal(A, A.T)
    # literal restatement of LargeScaleCondensing.py:306-310 with torch
    a = torch.from_numpy(S.copy())
    a = (a + a.T) / 2
    a[a < thr] = 0
    a.diagonal().fill_(1)
    ti = torch.nonzero(a).T
    ei, ew = orc.edge_construction(S, thr, "transductive")
    assert np.array_equal(ei, ti.numpy()) and np.array_equal(ew, a[ti[0], ti[1]].numpy())


def test_set_up_restatements_on_a_hand_checked_edge_list():
    # edges 0->1 twice, 1->0, 2->2, 1->2: csr_matrix sums the doubled edge; to_undirected de-duplicates
    ei = np.array([[0, 1, 0, 2, 1], [1, 0, 1, 2, 2]])
    rp, col, data = orc.pyg2dpr_adjacency(ei, 3, undirected=False)
    assert rp.tolist() == [0, 1, 3, 4] and col.tolist() == [1, 0, 2, 2] and data.tolist() == [2.0, 1.0, 1.0, 1.0]
    und = orc.to_undirected(ei, 3)
    assert und.tolist() == [[0, 1, 1, 2, 2], [1, 0, 2, 1, 2]]
    rp, col, data = orc.pyg2dpr_adjacency(ei, 3, undirected=True)
    assert rp.tolist() == [0, 1, 3, 5] and data.tolist() == [1.0] * 5


# ---- A3 pinned by the reference's own code ------------------------------------------------------------------
NORM = np.load(os.path.join(ROOT, "tests", "golden", "norm_golden.npz"))


@pytest.mark.parametrize("name", [str(x) for x in NORM["names"]])
def test_normalize_adj_oracle_matches_reference_outputs(name):
    """tests/golden/norm_golden.npz holds what the REFERENCE's normalize_adj_tensor(adj, sparse=True) returned
    (tests/golden/make_norm_golden.py runs /root/reference/utils.py's own functions): the oracle must reproduce
    it -- same entries, same order after coalescing, values bit for bit in the float64 mode.  In the float32 mode
    (node 0 has a self-loop) numpy's float32 power is CPU dependent, so a few ulp are allowed there."""
    n, ei, w = int(NORM[f"{name}_n"]), NORM[f"{name}_ei"], NORM[f"{name}_w"]
    r, c, v = orc.normalize_adj_scipy(ei[0], ei[1], w, n)
    order = np.lexsort((c, r))
    ref_idx, ref_val = NORM[f"{name}_out_index"], NORM[f"{name}_out_value"]
    # the reference's result may hold explicit zeros (isolated nodes after + I have r = 1, not 0; zeros arise only
    # from inf -> 0): compare structure and values as stored
    assert np.array_equal(np.stack([r[order], c[order]]), ref_idx)
    f32_mode = bool(((ei[0] == 0) & (ei[1] == 0) & (w != 0)).any())
    d = np.abs(v[order].view(np.int32).astype(np.int64) - ref_val.view(np.int32).astype(np.int64))
    assert d.max() <= (4 if f32_mode else 0), f"{name}: {d.max()} ulp"


@pytest.mark.parametrize("tag", ["dense_a", "dense_b"])
def test_dense_normalisation_matches_reference_outputs(tag):
    from disco_amd import utils
    got = utils.normalize_adj_tensor(torch.from_numpy(NORM[f"{tag}_in"]), sparse=False).numpy()
    assert np.allclose(got, NORM[f"{tag}_out"], rtol=2e-7, atol=0) and np.allclose(orc.normalize_adj_dense(NORM[f"{tag}_in"]),
                                                                                     NORM[f"{tag}_out"], rtol=1e-6, atol=0)


def test_gcn_norm_sparse_branch_oracle_closed_form():
    """A2 restatement: on an unweighted symmetric graph with add_self_loops it is D^-1/2 (A + I) D^-1/2 with
    D = deg + 1; an existing self-loop is REPLACED (fill_diag), not doubled."""
    ei = np.array([[0, 1, 1, 2, 2, 3, 1], [1, 0, 2, 1, 3, 2, 1]])          # path 0-1-2-3 (+ a loop on 1), node 4 isolated
    r, c, v = orc.gcn_norm_sparse(ei[0], ei[1], None, 5)
    deg = np.array([2, 3, 3, 2, 1], np.float64)
    dense = np.zeros((5, 5))
    dense[r, c] = v
    want = np.zeros((5, 5))
    for a, b in [(0, 1), (1, 2), (2, 3)]:
        want[a, b] = want[b, a] = 1 / np.sqrt(deg[a] * deg[b])
    want[np.arange(5), np.arange(5)] = 1 / deg
    assert np.allclose(dense, want, rtol=3e-7)
    r2, c2, v2 = orc.gcn_norm_sparse(ei[0], ei[1], None, 5, add_self_loops=False)
    assert len(v2) == 7 and np.isclose(v2[(r2 == 1) & (c2 == 1)][0], 1 / 3.0, rtol=3e-7)


def test_anchor_max_restatement_matches_torch_indexing():
    """A8 (/root/reference/LargeScaleCondensing.py:299): the oracle's numpy form equals the reference's torch
    expression ``feat_transform[index[c]][anchor, :L*d].max(dim=1).values`` evaluated with stock torch."""
    import torch
    rng = np.random.default_rng(3)
    T = rng.standard_normal((200, 24)).astype(np.float32)
    index_c = rng.permutation(200)[:70]
    anchor = rng.integers(0, 70, (31, 3))
    anchor[0] = anchor[0, 0]
    want = torch.from_numpy(T)[torch.from_numpy(index_c)][torch.from_numpy(anchor), :16].max(dim=1).values.numpy()
    assert np.array_equal(orc.anchor_max(T, anchor, 16, index=index_c), want)
    assert np.array_equal(orc.anchor_max(T[index_c], anchor), torch.from_numpy(T[index_c])[torch.from_numpy(anchor)].max(dim=1).values.numpy())


def test_normalize_sparse_tensor_restatement_matches_stock_torch():
    """/root/reference/utils.py:437-455 evaluated with stock torch on the CPU (index_add_ is the serial scatter_add there)
    against the oracle's loop: same bits up to torch's pow(-0.5) (compared at 1 ulp)."""
    import torch
    rng = np.random.default_rng(4)
    n, e = 60, 400
    row, col = rng.integers(0, n, e), rng.integers(0, n, e)
    w = rng.random(e).astype(np.float32) + 0.5
    r, c, v = orc.normalize_sparse_tensor(row, col, w, n, 1.0)
    ei = torch.from_numpy(np.stack([np.concatenate([row, np.arange(n)]), np.concatenate([col, np.arange(n)])]))
    ew = torch.cat([torch.from_numpy(w), torch.ones(n)])
    deg = torch.zeros(n).index_add_(0, ei[0], ew)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    ref = (dis[ei[0]] * ew * dis[ei[1]]).numpy()
    assert np.array_equal(r, ei[0].numpy()) and np.array_equal(c, ei[1].numpy())
    ulp = np.abs(v.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max()
    assert ulp <= 2, ulp


def test_tolerance_helper_counts_nan_and_inf_as_failures():
    """A result holding NaN or inf compares false with every bound: the checker must fail on it, not pass vacuously."""
    from helpers import assert_fwd_close
    y64 = np.ones((4, 3))
    bound = np.full((4, 3), 2.0)
    assert_fwd_close(y64.astype(np.float32), y64, bound)
    for poison in (np.nan, np.inf, -np.inf):
        y = y64.astype(np.float32).copy()
        y[2, 1] = poison
        with pytest.raises(AssertionError):
            assert_fwd_close(y, y64, bound)
    y = y64.astype(np.float32).copy()
    y[0, 0] += 1e-3
    with pytest.raises(AssertionError):
        assert_fwd_close(y, y64, bound)
