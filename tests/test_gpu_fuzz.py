"""Seeded random sweep over the corners the hand-written cases do not enumerate together: shape, feature width, leading
dimension (16- / 8- / 4-byte aligned rows), rectangular operands, hub rows, empty rows, duplicate columns, weighted or
not, every reduce, schedule on / off, window table on / off with a random window size, and the windowed modes (2, 5).
Every case is compared with the CPU oracle through the C ABI: max bit for bit, sum / mean / y += A x within the float32
bars of tests/helpers.py; with a window table the result must also have the bits of the run without one on every row
below the hub threshold."""
import numpy as np
import pytest
import torch

from helpers import assert_fwd_close, random_csr
from oracle import c_oracle

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
CASES = list(range(96))


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 7, 130, 900, 2500]))
    square = bool(rng.random() < 0.6)
    m = n if square else int(rng.choice([5, 300, 4000]))
    F = int(rng.choice([1, 3, 4, 30, 41, 64, 100, 128, 129, 256, 301, 602]))
    pad = int(rng.choice([0, 1, 2, 4, 32]))
    ld = F if pad == 0 else (F + pad - 1) // pad * pad + (pad if rng.random() < 0.3 else 0)
    avg = float(rng.choice([0.5, 4, 30, 120]))
    hubs = []
    if n >= 130 and rng.random() < 0.6:
        hubs = [(int(rng.integers(0, n)), int(rng.choice([600, 3000, 9000]))) for _ in range(int(rng.integers(1, 4)))]
    return dict(n=n, m=m, F=F, ld=ld, avg=avg, hubs=hubs, weighted=bool(rng.random() < 0.75), dup=bool(rng.random() < 0.3),
                reduce=str(rng.choice(["sum", "mean", "max", "sum_acc"])), schedule=bool(rng.random() < 0.8),
                window=bool(rng.random() < 0.5), wrows=int(rng.choice([16, 97, 700])), mode=int(rng.choice([2, 5])), seed=seed)


@pytest.mark.parametrize("case", CASES)
def test_random_case_against_oracle(case):
    from disco_amd import SparseTensor, _lib, ops
    c = _draw(case)
    n, m, F, ld, reduce = c["n"], c["m"], c["F"], c["ld"], c["reduce"]
    rowptr, col, val = random_csr(n, m, c["avg"], seed=case, hubs=c["hubs"], weighted=c["weighted"], dup=c["dup"])
    rng = np.random.default_rng(case)
    x = rng.standard_normal((m, F)).astype(np.float32)
    y0 = rng.standard_normal((n, F)).astype(np.float32)
    xb = torch.zeros((m, ld), device=DEV)
    xb[:, :F] = torch.from_numpy(x).to(DEV)

    def run(window):
        adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV),
                           value=None if val is None else torch.from_numpy(val).to(DEV), sparse_sizes=(n, m), is_sorted=True)
        if reduce == "max":
            adj = adj.set_value(None)
        if c["schedule"]:
            wb = c["wrows"] * 4 * min(128, (F + 31) // 32 * 32)
            adj.schedule(2048, 1024, feat=F, sweep=window, window_bytes=wb)
        out = torch.full((n, ld), 7.0, device=DEV)
        out[:, :F] = torch.from_numpy(y0).to(DEV) if reduce == "sum_acc" else 0.0
        lib = _lib.load()
        lib.disco_set_tuning(b"sweep_mode", c["mode"])
        try:
            ops.spmm_raw(adj, xb[:, :F], reduce, out=out[:, :F], use_schedule=c["schedule"])
            torch.cuda.synchronize()
        finally:
            lib.disco_set_tuning(b"sweep_mode", 2)
        return out.cpu().numpy()

    got = run(c["window"] and c["schedule"])
    assert (got[:, F:] == 7.0).all(), f"{c}: wrote beyond column F"
    y = got[:, :F]
    v = None if reduce == "max" else val
    if reduce == "max":
        assert np.array_equal(y, c_oracle.spmm_csr_f32(rowptr, col, None, x, "max")), str(c)
    else:
        y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, v, x, want_bound=True)
        if reduce == "mean":
            cnt = np.maximum(np.diff(rowptr), 1)[:, None].astype(np.float64)
            y64, bound = y64 / cnt, bound / cnt
        if reduce == "sum_acc":
            y64, bound = y64 + y0.astype(np.float64), bound + np.abs(y0)
        # Without a schedule no row is split: a row of thousands of entries is then ONE strictly sequential float32 chain
        # (torch_sparse's own CSR order), whose rounding error grows with the row length -- 2.4e-5 of |y| was seen at
        # 9 000 entries, the same as the float32 CPU port gives.  The element-relative bar is applied where the default
        # path (schedule: hub rows cut into pieces and combined) or a short row is at work; the forward-error bar everywhere.
        long_seq = (not c["schedule"]) & (np.diff(rowptr) > 2048)
        assert_fwd_close(y[~long_seq], y64[~long_seq], bound[~long_seq], what=str(c))
        if long_seq.any():
            assert_fwd_close(y[long_seq], y64[long_seq], bound[long_seq], what=str(c) + " (long sequential rows)", rel_floor=np.inf)
            y32 = c_oracle.spmm_csr_f32(rowptr, col, v, x, "mean" if reduce == "mean" else "sum")
            if reduce == "sum_acc":
                y32 = y0 + y32
            e_gpu = np.abs(y[long_seq] - y64[long_seq]).max()
            e_cpu = np.abs(y32[long_seq].astype(np.float64) - y64[long_seq]).max()
            assert e_gpu <= 4 * e_cpu + 1e-30, f"{c}: long sequential rows are less accurate than the float32 CPU port ({e_gpu:.3e} vs {e_cpu:.3e})"
    if c["window"] and c["schedule"] and reduce != "sum_acc":
        plain = run(False)[:, :F]
        keep = np.diff(rowptr) <= 2048
        assert np.array_equal(y[keep], plain[keep]), f"{c}: windowed bits differ from the plain kernel below the hub threshold"


@pytest.mark.parametrize("case", list(range(24)))
def test_random_gcn_norm_bit_exact(case):
    """A1 (/root/reference/utils.py:228-247) over random edge lists: weighted or not, either flow, with and without the
    remaining self-loops, isolated nodes, duplicate edges, existing loops -- weights bit for bit against the oracle."""
    from disco_amd import utils
    from helpers import orc
    rng = np.random.default_rng(500 + case)
    n = int(rng.choice([1, 6, 200, 3000]))
    E = int(rng.choice([0, 1, 5 * n, 40 * n]))
    ei = rng.integers(0, n, (2, E))
    if E > 4:
        ei[:, :2] = ei[:, 2:4]                                  # duplicate edges
        ei[1, 4 % E] = ei[0, 4 % E]                             # an existing self-loop
    if n > 10:
        ei[ei == 3] = 4                                         # node 3 isolated
    weighted = bool(rng.random() < 0.6)
    loops = bool(rng.random() < 0.5)
    flow = "source_to_target" if loops or rng.random() < 0.5 else "target_to_source"
    w = (rng.random(E).astype(np.float32) + 0.05) if weighted else None
    ei_ref, ref = orc.gcn_norm_coo(ei, w, n, add_self_loops=loops, flow=flow)
    ei_got, got = utils.gcn_norm(torch.from_numpy(ei).to(DEV), None if w is None else torch.from_numpy(w).to(DEV), n,
                                 add_self_loops=loops, flow=flow)
    assert np.array_equal(ei_got.cpu().numpy(), ei_ref)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), np.asarray(ref, np.float32).view(np.uint32))


@pytest.mark.parametrize("case", list(range(24)))
def test_random_khop_against_oracle(case):
    """A4: K = 1..3 hops through disco_khop_csr_f32 (ping-pong buffers, optional column slices) on random square graphs
    with hub rows, against the float64 oracle."""
    from disco_amd import SparseTensor, ops
    from helpers import orc
    rng = np.random.default_rng(700 + case)
    n = int(rng.choice([3, 150, 1200]))
    F = int(rng.choice([1, 5, 64, 100, 128, 200]))
    K = int(rng.integers(1, 4))
    slices = int(rng.choice([1, 1, 2, 3]))
    hubs = [(int(rng.integers(0, n)), 3000)] if n >= 150 and rng.random() < 0.5 else []
    rowptr, col, val = random_csr(n, n, float(rng.choice([2, 20])), seed=700 + case, hubs=hubs, weighted=True)
    val = (val * np.float32(0.2)).astype(np.float32)
    x = rng.standard_normal((n, F)).astype(np.float32)
    adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV),
                       value=torch.from_numpy(val).to(DEV), sparse_sizes=(n, n), is_sorted=True)
    y = ops.khop(adj, torch.from_numpy(x).to(DEV), K, n_col_slices=slices).cpu().numpy()
    y64 = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert_fwd_close(y, y64, K * bound, what=f"case {case}: n={n} F={F} K={K} slices={slices}")


@pytest.mark.parametrize("case", list(range(16)))
def test_random_normalize_adj_tensor_against_oracle(case):
    """A3 (/root/reference/utils.py:473-484 -> :406-435) over random sparse COO tensors with duplicates, isolated nodes and
    either state of the mx[0, 0] rule: structure identical, values within 1 float32 ulp of the oracle's scipy restatement
    in the float64 mode (identity added) and 4 ulp in the float32 mode (numpy's float32 power is a few ulp off)."""
    from disco_amd import utils
    from helpers import orc
    rng = np.random.default_rng(900 + case)
    n = int(rng.choice([2, 9, 400, 2500]))
    E = min(int(rng.choice([n, 6 * n, 30 * n])), n * n)
    cells = rng.choice(n * n, E, replace=False)                # distinct cells ...
    row, col = cells // n, cells % n
    if E > 6:                                                  # ... plus three entries given twice: csr_matrix sums them (a sum
        row, col = np.concatenate([row, row[:3]]), np.concatenate([col, col[:3]])     # of two is order independent; scipy
    # sorts a row's entries with an unstable sort before folding, so the float32 sum of three or more has no defined order)
    loop0 = bool(case % 2)
    if loop0:
        row[-1], col[-1] = 0, 0                                # node 0 has a self-loop: no identity, float32 mode
    else:
        keep = ~((row == 0) & (col == 0))
        row, col = row[keep], col[keep]
    if n > 20:
        row[row == 5] = 6                                      # row 5 empty: rowsum 0 -> r = 0 (float32 mode) or 1 (identity)
    val = (rng.random(len(row)).astype(np.float32) + 0.1)
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.stack([row, col])), torch.from_numpy(val), (n, n)).to(DEV)
    out = utils.normalize_adj_tensor(adj, sparse=True)
    r_ref, c_ref, v_ref = orc.normalize_adj_scipy(row, col, val, n)
    idx, v = out.indices().cpu().numpy(), out.values().cpu().numpy()
    keep = v != 0
    order = np.lexsort((c_ref, r_ref))
    nz = v_ref[order] != 0
    assert np.array_equal(idx[0][keep], r_ref[order][nz]) and np.array_equal(idx[1][keep], c_ref[order][nz])
    ulp = np.abs(v[keep].view(np.int32).astype(np.int64) - v_ref[order][nz].view(np.int32).astype(np.int64)).max()
    assert ulp <= (4 if loop0 else 1), (case, int(ulp))


@pytest.mark.parametrize("case", list(range(12)))
def test_random_scorer_against_module_and_oracle(case):
    """A9 over random scorers: hidden width 32 / 64 / 128 / 256, odd node counts and feature widths, non-trivial BatchNorm
    running statistics.  Logits of the HIP scorer against the module's plain torch layers over explicit [z_i || z_j] rows
    (the reference's loop, LargeScaleCondensing.py:304-305); fused edge construction = dense path bit for bit; the dense
    path = the oracle's post-processing of the same scores, for a random variant and threshold."""
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    from helpers import orc
    rng = np.random.default_rng(300 + case)
    torch.manual_seed(300 + case)
    n = int(rng.choice([1, 2, 63, 65, 200, 333]))
    nfeat = int(rng.choice([3, 16, 50, 128]))
    H = int(rng.choice([32, 64, 128, 256]))
    m = PGE_Edge(nfeat=nfeat, nhid=H).to(DEV)
    with torch.no_grad():
        for bn in m.bns:
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.3, 1.7)
            bn.weight.normal_(1.0, 0.3)
            bn.bias.normal_(0, 0.2)
    m.eval()
    Z = torch.randn(n, nfeat, device=DEV)
    L = m.score_all_pairs(Z, logits=True)
    with torch.no_grad():
        pairs = torch.cat([Z[:, None, :].expand(n, n, nfeat), Z[None, :, :].expand(n, n, nfeat)], dim=2).reshape(n * n, 2 * nfeat)
        ref = m.logits(pairs).reshape(n, n)
    assert torch.allclose(L, ref, rtol=2e-5, atol=2e-5), float((L - ref).abs().max())
    S = m.score_all_pairs(Z)
    variant = str(rng.choice(["transductive", "inductive", "sampled"]))
    thr = float(np.quantile(S.cpu().numpy(), float(rng.choice([0.2, 0.7, 0.97])))) if n > 2 else 0.5
    ei_f, ew_f = m.edge_construction(Z, thr, variant)
    ei_d, ew_d = m.edge_construction(Z, thr, variant, fused=False)
    assert torch.equal(ei_f, ei_d) and torch.equal(ew_f, ew_d)
    rei, rew = orc.edge_construction(S.cpu().numpy(), thr, variant)
    assert np.array_equal(ei_d.cpu().numpy(), rei) and np.array_equal(ew_d.cpu().numpy(), rew)
    small = m.edge_construction(Z, thr, variant, capacity=1)          # overflow of the edge buffer: one retry, same list
    assert torch.equal(small[0], ei_f) and torch.equal(small[1], ew_f)


@pytest.mark.parametrize("case", list(range(16)))
def test_random_spmm_gradients_against_dense_autograd(case):
    """F4: dX = A^T dY through ops.spmm's autograd (sum and mean), rectangular shapes, hub rows, against float64 dense
    autograd of the same product."""
    from disco_amd import SparseTensor, ops
    rng = np.random.default_rng(400 + case)
    n, m_ = int(rng.choice([1, 40, 700])), int(rng.choice([3, 90, 700]))
    F = int(rng.choice([1, 7, 64, 100, 130]))
    reduce = "mean" if case % 2 else "sum"
    hubs = [(0, 1500)] if n >= 40 and rng.random() < 0.5 else []
    rowptr, col, val = random_csr(n, m_, float(rng.choice([1, 12])), seed=400 + case, hubs=hubs, weighted=True, dup=bool(case % 3 == 0))
    adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV),
                       value=torch.from_numpy(val).to(DEV), sparse_sizes=(n, m_), is_sorted=True)
    x = torch.randn(m_, F, device=DEV, requires_grad=True)
    g = torch.randn(n, F, device=DEV)
    y = ops.spmm(adj, x, reduce)
    y.backward(g)
    row = np.repeat(np.arange(n), np.diff(rowptr))
    A = torch.zeros(n, m_, dtype=torch.float64, device=DEV).index_put_(
        (torch.from_numpy(row).to(DEV), torch.from_numpy(col).to(DEV)), torch.from_numpy(val).double().to(DEV), accumulate=True)
    Aabs = A.abs()
    if reduce == "mean":
        cnt = torch.from_numpy(np.maximum(np.diff(rowptr), 1)).double().to(DEV)[:, None]
        A, Aabs = A / cnt, Aabs / cnt
    xd = x.detach().double().requires_grad_(True)
    yd = A @ xd
    yd.backward(g.double())
    by, bg = Aabs @ x.detach().abs().double(), Aabs.T @ g.abs().double()
    assert bool(((y.detach().double() - yd.detach()).abs() <= 1e-5 * by + 1e-30).all())
    assert bool(((x.grad.double() - xd.grad).abs() <= 1e-5 * bg + 1e-30).all())


@pytest.mark.parametrize("case", list(range(16)))
def test_random_coo_to_csr_and_transpose(case):
    """F3 + the transposed structure backward uses: disco_coo_to_csr in its three duplicate modes (with and without
    symmetrisation) against numpy / scipy on random rectangular edge lists, and SparseTensor.t() against scipy's transpose
    (entries of a column in original row order, values carried along)."""
    import scipy.sparse as sp
    from disco_amd import SparseTensor
    from disco_amd.sparse import coo_to_csr
    rng = np.random.default_rng(600 + case)
    n = int(rng.choice([1, 5, 300, 5000]))
    sym = bool(case % 4 == 0)
    m_ = n if sym else int(rng.choice([1, 7, 900]))
    E = int(rng.choice([0, 1, 4 * n, 25 * n]))
    r, c = rng.integers(0, n, E), rng.integers(0, m_, E)
    if E > 8:
        r[:4], c[:4] = r[4:8], c[4:8]
    w = rng.integers(1, 5, E).astype(np.float32)              # small integers: float32 sums are exact in any order
    dup = ["keep", "sum", "first"][case % 3]
    rp, ci, v = coo_to_csr(torch.from_numpy(r).to(DEV), torch.from_numpy(c).to(DEV), torch.from_numpy(w).to(DEV), (n, m_),
                           dup=dup, symmetrise=sym)
    rr, cc, ww = (np.concatenate([r, c]), np.concatenate([c, r]), np.concatenate([w, w])) if sym else (r, c, w)
    order = np.lexsort((np.arange(len(rr)), cc, rr))
    rr, cc, ww = rr[order], cc[order], ww[order]
    if dup != "keep" and len(rr):
        key = rr * m_ + cc
        first = np.concatenate([[True], key[1:] != key[:-1]])
        grp = np.cumsum(first) - 1
        ww = np.bincount(grp, weights=ww).astype(np.float32) if dup == "sum" else ww[first]
        rr, cc = rr[first], cc[first]
    ref_rp = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(rr, minlength=n), out=ref_rp[1:])
    assert np.array_equal(rp.cpu().numpy(), ref_rp)
    assert np.array_equal(ci.cpu().numpy().astype(np.int64), cc) and np.array_equal(v.cpu().numpy(), ww)
    # transpose of the resulting CSR
    adj = SparseTensor(rowptr=rp, col=ci, value=v, sparse_sizes=(n, m_), is_sorted=True)
    t = adj.t()
    trp, tci, tv = (a.cpu().numpy() for a in t.csr())
    row_of = np.repeat(np.arange(n), np.diff(ref_rp))
    o2 = np.lexsort((np.arange(len(cc)), row_of, cc))
    ref_trp = np.zeros(m_ + 1, np.int64)
    np.cumsum(np.bincount(cc, minlength=m_), out=ref_trp[1:])
    assert t.sparse_sizes() == (m_, n) and np.array_equal(trp, ref_trp)
    assert np.array_equal(tci, row_of[o2]) and np.array_equal(tv, ww[o2])
    if len(cc):
        dense = sp.csr_matrix((ww.astype(np.float64), cc, ref_rp), shape=(n, m_)).toarray()
        assert np.array_equal(sp.csr_matrix((tv.astype(np.float64), tci, trp), shape=(m_, n)).toarray(), dense.T)


@pytest.mark.parametrize("conv_name", ["GCNConv", "SGConv", "SAGEConv"])
def test_fresh_edge_lists_of_equal_size_never_hit_a_stale_cache(conv_name):
    """A layer called with a NEW int64 [2, E] edge list per step (same E: the allocator likes to hand out the address of
    the list that was just freed) must build a new structure every time -- forward and backward against dense float64."""
    from disco_amd.models import convs_large as C
    torch.manual_seed(0)
    n, E, d = 400, 5000, 24
    conv = getattr(C, conv_name)(d, 8).to(DEV)
    for step in range(6):
        g = torch.Generator(device=DEV).manual_seed(step)
        ei = torch.randint(0, n, (2, E), device=DEV, generator=g)
        w = torch.rand(E, device=DEV, generator=g) if conv_name != "SAGEConv" else None
        x = torch.randn(n, d, device=DEV, requires_grad=True)
        out = conv(x, ei, edge_weight=w) if w is not None else conv(x, ei)
        out.sum().backward()
        A = torch.zeros(n, n, dtype=torch.float64, device=DEV).index_put_(
            (ei[1], ei[0]), torch.ones(E, dtype=torch.float64, device=DEV) if w is None else w.double(), accumulate=True)
        if conv_name == "SAGEConv":
            A = A / A.sum(1, keepdim=True).clamp(min=1)
        xd = x.detach().double()
        with torch.no_grad():
            if conv_name == "GCNConv":
                ref = A @ (xd @ conv.lin.weight.double().T) + conv.bias.double()
            elif conv_name == "SGConv":
                h = xd
                for _ in range(conv.K):
                    h = A @ h
                ref = h @ conv.lin.weight.double().T + conv.lin.bias.double()
            else:
                ref = (A @ xd) @ conv.lin_l.weight.double().T + conv.lin_l.bias.double() + xd @ conv.lin_r.weight.double().T + conv.lin_r.bias.double()
        scale = ref.abs().max().item() + 1.0
        assert float((out.detach().double() - ref).abs().max()) <= 2e-5 * scale * (conv.K if conv_name == "SGConv" else 1) * 30, (conv_name, step)
        del ei, w, x, out, A


@pytest.mark.parametrize("case", list(range(16)))
def test_random_row_groups_keep_every_bit(case):
    """Row groups (random labels, the locality pass's labels, one giant group, one group per row) only reorder the rows a
    launch walks: the result must carry the bits of the degree-binned order, for every reduce, with and without windows."""
    from disco_amd import SparseTensor, ops
    rng = np.random.default_rng(4000 + case)
    n = int(rng.choice([3, 257, 1500, 6000]))
    F = int(rng.choice([1, 16, 41, 128, 200]))
    hubs = [(int(rng.integers(0, n)), 5000)] if n > 1000 and rng.random() < 0.5 else []
    rowptr, col, val = random_csr(n, n, float(rng.choice([2, 25])), seed=4000 + case, hubs=hubs, weighted=True, dup=False)
    reduce = str(rng.choice(["sum", "mean", "max"]))
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    window = bool(rng.random() < 0.5)

    def build():
        a = SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV),
                         value=None if reduce == "max" else torch.from_numpy(val).to(DEV), sparse_sizes=(n, n), is_sorted=True)
        return a

    base = build()
    base.schedule(2048, 1024, feat=F, sweep=window, window_bytes=64 * 4 * min(128, (F + 31) // 32 * 32))
    ref = ops.spmm_raw(base, x, reduce)
    kind = case % 4
    g = build()
    if kind == 0:
        labels = torch.from_numpy(rng.integers(0, max(1, n // 50), n)).to(DEV)
    elif kind == 1:
        labels = g.locality_groups(sweeps=4, max_group=int(rng.choice([0, 64, 1000])))
    elif kind == 2:
        labels = torch.zeros(n, dtype=torch.int64, device=DEV)
    else:
        labels = torch.arange(n, device=DEV).flip(0)
    g.set_row_groups(labels)
    g.schedule(2048, 1024, feat=F, sweep=window, window_bytes=64 * 4 * min(128, (F + 31) // 32 * 32))
    got = ops.spmm_raw(g, x, reduce)
    # hub rows: the grouped order walks tables wider than 160 floats in 128-column blocks, which changes the number of lane
    # groups that share a hub piece and with it the association of that piece's sum (max has no rounding)
    keep = torch.from_numpy((np.diff(rowptr) <= 2048) | (F <= 160) | (reduce == "max")).to(DEV)
    assert torch.equal(got[keep], ref[keep]), (case, kind, n, F, reduce, window)
    if not bool(keep.all()):
        y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x.cpu().numpy(), want_bound=True)
        if reduce == "mean":
            cnt = np.maximum(np.diff(rowptr), 1)[:, None].astype(np.float64)
            y64, bound = y64 / cnt, bound / cnt
        assert_fwd_close(got.cpu().numpy(), y64, bound, what=f"grouped order, hub rows {case}")
    g.set_row_groups(None)
    g.schedule(2048, 1024, feat=F, sweep=window, window_bytes=64 * 4 * min(128, (F + 31) // 32 * 32))
    assert torch.equal(ops.spmm_raw(g, x, reduce), ref), "back to the degree-binned order"


@pytest.mark.parametrize("case", list(range(16)))
def test_random_pair_gather_is_torch_indexing(case):
    """F3 (/root/reference/LargeScaleCondensing.py:190-191): concat(z[src[sel]], z[dst[sel]]) -- bit for bit against torch
    indexing over random widths, strides, selections (repeats, empty, every edge) and int32 / int64 edge lists."""
    from disco_amd import ops
    rng = np.random.default_rng(5000 + case)
    n = int(rng.choice([1, 9, 500, 20000]))
    w = int(rng.choice([1, 3, 4, 32, 41, 128, 300]))
    ld = w + int(rng.choice([0, 1, 3, 4]))
    E = int(rng.choice([0, 1, 77, 30000]))
    zb = torch.from_numpy(rng.standard_normal((n, ld)).astype(np.float32)).to(DEV)
    z = zb[:, :w]
    ei = torch.from_numpy(rng.integers(0, n, (2, E))).to(DEV)
    if rng.random() < 0.3:
        ei = ei.to(torch.int32)
    mode = case % 3
    sel = None if mode == 0 or E == 0 else torch.from_numpy(rng.integers(0, E, int(rng.choice([0, 1, 513, 4096])))).to(DEV)
    got = ops.pair_gather(z, ei, sel)
    s, d = ei[0].long(), ei[1].long()
    if sel is not None:
        s, d = s[sel], d[sel]
    ref = torch.cat((z[s], z[d]), dim=1)
    assert got.shape == ref.shape and torch.equal(got, ref), (case, n, w, ld, E, mode)
    if E > 0:
        bad = ei.clone()
        bad[1, int(rng.integers(0, E))] = n
        if sel is None or (sel.numel() and False):
            with pytest.raises(IndexError):
                ops.pair_gather(z, bad, None)


@pytest.mark.parametrize("case", list(range(12)))
def test_random_khop_column_slices_keep_every_bit(case):
    """A4 with n_col_slices > 1 (each slice runs its K hops before the next starts) must carry the bits of the unsliced
    call: the slices touch disjoint columns and the rows' sums do not depend on the slice."""
    from disco_amd import SparseTensor, ops
    rng = np.random.default_rng(6000 + case)
    n = int(rng.choice([5, 400, 3000]))
    F = int(rng.choice([4, 33, 128, 256, 300, 602]))
    K = int(rng.choice([1, 2, 3, 5]))
    rowptr, col, val = random_csr(n, n, float(rng.choice([1, 8, 40])), seed=6000 + case, hubs=[], weighted=True, dup=False)
    val = (val / np.maximum(np.add.reduceat(np.abs(val), np.minimum(rowptr[:-1], max(len(val) - 1, 0))).repeat(np.diff(rowptr)), 1e-3)
           ).astype(np.float32) if len(val) else val
    a = SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV), value=torch.from_numpy(val).to(DEV),
                     sparse_sizes=(n, n), is_sorted=True)
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    ref = ops.khop(a, x, K)
    for s in (2, 3, 7):
        assert torch.equal(ops.khop(a, x, K, n_col_slices=s), ref), (case, n, F, K, s)
    h = x
    for _ in range(K):
        h = ops.spmm_raw(a, h, "sum")
    assert torch.equal(h, ref), "K calls of the one-hop kernel give the same bits as the K-hop entry point"


def _dense_A(edge_index, n_src, size, w):
    """(A [n_dst, n_src] float64, entries per row) from a SparseTensor adj_t or an int64 [2, E] edge list + weights."""
    from disco_amd.models.convs_large.message_passing import as_adj_t
    adj_t = as_adj_t(edge_index)
    if adj_t is not None:
        rowptr, col, val = adj_t.csr()
        n_dst, n_src = adj_t.sparse_sizes()
        r = torch.repeat_interleave(torch.arange(n_dst, device=rowptr.device), rowptr.diff())
        c, vals = col.long(), (None if val is None else val.double())
    else:
        n_dst = n_src if size is None or size[1] is None else size[1]
        r, c, vals = edge_index[1].long(), edge_index[0].long(), (None if w is None else w.double())
    if vals is None:
        vals = torch.ones(r.numel(), dtype=torch.float64, device=r.device)
    A = torch.zeros(n_dst, n_src, dtype=torch.float64, device=r.device).index_put_((r, c), vals, accumulate=True)
    return A, torch.bincount(r, minlength=n_dst)


def _dense_propagate(self, edge_index, size=None, **kw):
    x = kw.get("x")
    x = x[0] if isinstance(x, (tuple, list)) else x
    A, cnt = _dense_A(edge_index, x.shape[0], size, kw.get("edge_weight"))
    out = A @ x.double()
    if self.aggr == "mean":
        out = out / cnt.clamp(min=1).double()[:, None]
    assert self.aggr != "max"
    return out


@pytest.mark.parametrize("case", list(range(30)))
def test_random_models_forward_backward_against_dense_float64(case, monkeypatch):
    """A5 / A6 / A10 callers end to end: a random BasicGNN (family, depth, width, norm, jumping knowledge), a random
    graph given as [2, E] + weights, [2, E] alone or a SparseTensor adj_t, in train or eval mode -- log-probabilities and
    every gradient (input and parameters) against the SAME module evaluated in float64 with every propagation replaced
    by a dense matrix product."""
    import copy
    from disco_amd import SparseTensor, ops, utils
    from disco_amd.models import basicgnn_large as M
    from disco_amd.models.convs_large import message_passing as mp
    rng = np.random.default_rng(7000 + case)
    torch.manual_seed(7000 + case)
    name = ["GCN", "SGC", "GraphSAGE", "GIN", "JKNet"][case % 5]
    n = int(rng.choice([40, 300, 1200]))
    d, nhid, C = int(rng.choice([5, 32, 100])), int(rng.choice([8, 33, 64])), int(rng.choice([2, 7, 40]))
    nlayers = int(rng.choice([1, 2, 3, 4]))
    norm = "BatchNorm" if rng.random() < 0.5 else None
    E = int(n * rng.choice([2, 12]))
    ei = torch.from_numpy(rng.integers(0, n, (2, E))).to(DEV)
    ew = torch.from_numpy(rng.random(E).astype(np.float32) + 0.05).to(DEV)
    if name in ("GCN", "SGC", "JKNet"):
        ei, ew = utils.gcn_norm(ei, ew, n, add_self_loops=bool(rng.random() < 0.5))
    kind = int(rng.integers(0, 3))          # 0: [2,E] + weights, 1: [2,E] alone, 2: SparseTensor adj_t
    train = bool(rng.random() < 0.5)
    if name == "SGC":
        model = M.SGC(nfeat=d, nhid=nhid, nclass=C, dropout=0.0, nlayers=max(nlayers, 1), sgc=True)
    elif name == "JKNet":
        model = M.JKNet(nfeat=d, nhid=nhid, nclass=C, dropout=0.0, nlayers=max(nlayers, 2), norm=norm, jk=str(rng.choice(["cat", "max"])))
    else:
        model = getattr(M, name)(nfeat=d, nhid=nhid, nclass=C, dropout=0.0, nlayers=nlayers, norm=norm)
    model = model.to(DEV)
    model.initialize()
    for m_ in model.modules():
        if isinstance(m_, torch.nn.BatchNorm1d):
            with torch.no_grad():
                m_.running_mean.normal_(0, 0.1)
                m_.running_var.uniform_(0.5, 1.5)
    model.train(train)
    ref_model = copy.deepcopy(model).double()
    x = torch.randn(n, d, device=DEV)
    g = torch.randn(n, C, device=DEV)

    def graph():
        if kind == 2:
            return SparseTensor(row=ei[1], col=ei[0], value=ew, sparse_sizes=(n, n)), {}
        return ei, ({"edge_weight": ew} if kind == 0 else {})

    def run(mdl, xin, gin):
        xin = xin.clone().requires_grad_(True)
        adj, kw = graph()
        out = mdl(xin, adj, **kw)
        (out * gin).sum().backward()
        grads = {k: p.grad for k, p in mdl.named_parameters() if p.grad is not None}
        return out.detach(), xin.grad, grads

    out, gx, gp = run(model, x, g)
    with monkeypatch.context() as mpc:
        mpc.setattr(mp.MessagePassing, "propagate", _dense_propagate)

        def khop64(adj_t, x_, K, **_):
            A, _c = _dense_A(adj_t, x_.shape[0], None, None)
            for _ in range(K):
                x_ = A @ x_.double()
            return x_
        mpc.setattr(ops, "khop", khop64)
        out64, gx64, gp64 = run(ref_model, x.double(), g.double())
    what = (case, name, n, d, nhid, C, nlayers, norm, kind, train)

    # a bias in front of a BatchNorm in train mode has gradient 0 exactly; float32 leaves rounding noise there, which is
    # measured against the largest parameter gradient instead of against 0
    gscale = max(float(v.abs().max()) for v in gp64.values())

    def close(a, b, tag, tol=2e-4):
        scale = float(b.abs().max()) + 1e-6
        if tag.startswith("d/d") and tag != "d/dx":
            scale = max(scale, 1e-2 * gscale)
        err = float((a.double() - b).abs().max())
        assert err <= tol * scale, f"{what} {tag}: {err:.3e} vs scale {scale:.3e}"

    assert torch.isfinite(out).all()
    close(out, out64, "log-probabilities")
    close(gx, gx64, "d/dx")
    assert set(gp) == set(gp64), what
    for k in gp:
        close(gp[k], gp64[k], "d/d" + k, tol=5e-4)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16, torch.int64])
def test_values_of_another_dtype_are_read_as_numbers_not_as_bytes(dtype):
    """A SparseTensor may carry float64 (scipy's default), half or integer values; every entry point that hands the
    value array to a float32 kernel converts it first (one of them, the K-hop call, used to pass the raw pointer)."""
    from disco_amd import SparseTensor, ops, utils
    rng = np.random.default_rng(11)
    n, F = 500, 40
    rowptr, col, val = random_csr(n, n, 6, seed=11, hubs=[], weighted=True, dup=False)
    v32 = torch.from_numpy(np.round(np.abs(val) * 4 + 1).astype(np.float32)).to(DEV)      # small integers: exact in every dtype
    mk = lambda v: SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV), value=v,
                                sparse_sizes=(n, n), is_sorted=True)
    a32, a = mk(v32), mk(v32.to(dtype))
    x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).to(DEV)
    assert torch.equal(ops.spmm_raw(a, x, "sum"), ops.spmm_raw(a32, x, "sum"))
    assert torch.equal(ops.khop(a, x, 2), ops.khop(a32, x, 2))
    assert torch.equal(ops.spmm_raw(a.t(), x, "sum"), ops.spmm_raw(a32.t(), x, "sum"))
    for with00 in (False, True):
        if with00:      # mx[0,0] != 0: normalize_adj adds no identity and reads the values as they are
            r = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), a32.coo()[0]])
            c = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), a32.coo()[1]])
            keep = torch.ones(r.numel(), dtype=torch.bool, device=DEV)
            keep[1:] = ~((r[1:] == 0) & (c[1:] == 0))
            vv = torch.cat([torch.full((1,), 3.0, device=DEV), v32])[keep]
            b32 = SparseTensor(row=r[keep], col=c[keep], value=vv, sparse_sizes=(n, n))
            b = b32.set_value(b32.storage.value().to(dtype))
        else:
            b32, b = a32, a
        assert torch.equal(utils.normalize_adj_csr(b).storage.value(), utils.normalize_adj_csr(b32).storage.value()), with00
    assert torch.equal(utils.gcn_norm(a, add_self_loops=False).storage.value(), utils.gcn_norm(a32, add_self_loops=False).storage.value())


def test_wrong_output_buffers_are_refused_before_any_launch():
    """An `out` / `scratch` of the wrong shape, type, stride or device would be overrun by the kernels: refused in Python."""
    from disco_amd import SparseTensor, ops
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    n, F = 50, 12
    rowptr, col, val = random_csr(n, n, 4, seed=3, hubs=[], weighted=True, dup=False)
    a = SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV), value=torch.from_numpy(val).to(DEV),
                     sparse_sizes=(n, n), is_sorted=True)
    x = torch.randn(n, F, device=DEV)
    bad = [torch.empty(n - 1, F, device=DEV), torch.empty(n, F - 1, device=DEV), torch.empty(n, F, device=DEV, dtype=torch.float64),
           torch.empty(F, n, device=DEV).t(), torch.empty(n, F), torch.empty(n * F, device=DEV)]
    for out in bad:
        with pytest.raises((ValueError, RuntimeError)):
            ops.spmm_raw(a, x, "sum", out=out)
        with pytest.raises((ValueError, RuntimeError)):
            ops.khop(a, x, 2, out=out)
    with pytest.raises(ValueError):
        ops.khop(a, x, 0)
    with pytest.raises(ValueError):
        ops.khop(a, x, 3, scratch=torch.empty(4 * n * 32, device=DEV, dtype=torch.float64))
    good = torch.full((n, 16), 5.0, device=DEV)
    ops.khop(a, x, 3, out=good[:, :F], scratch=torch.empty(2 * n * 32, device=DEV))
    assert torch.equal(good[:, :F], ops.khop(a, x, 3)) and bool((good[:, F:] == 5.0).all())
    m = PGE_Edge(nfeat=F, nhid=32, nlayers=3).to(DEV)
    Z = torch.randn(n, F, device=DEV)
    for out in (torch.empty(4, n, device=DEV), torch.empty(5, n - 1, device=DEV), torch.empty(5, n, device=DEV, dtype=torch.float64),
                torch.empty(n, 5, device=DEV).t()):
        with pytest.raises(ValueError):
            m.score_block(Z, (0, 5), out=out)
    with pytest.raises(ValueError):
        m.edges_from_scores(torch.zeros(4, 5, device=DEV), 0.5)
    S = m.score_all_pairs(Z)
    e1 = m.edges_from_scores(S, 0.5)
    e2 = m.edges_from_scores(S.t().contiguous().t(), 0.5)          # column-major view of the same matrix: copied, same edges
    assert torch.equal(e1[0], e2[0]) and torch.equal(e1[1], e2[1])


def test_results_do_not_change_from_run_to_run():
    """No atomics on any float: the same call gives the same bits every time -- plain, hub, windowed and K-hop propagation,
    the normalisations, and the fused edge construction (whose edges leave the kernel in arrival order and are sorted)."""
    from disco_amd import SparseTensor, ops, utils
    from disco_amd.models.parametrized_adj_lp import PGE_Edge
    n, F = 20000, 200
    rowptr, col, val = random_csr(n, n, 40, seed=77, hubs=[(5, 9000), (77, 3000)], weighted=True, dup=False)
    x = torch.from_numpy(np.random.default_rng(7).standard_normal((n, F)).astype(np.float32)).to(DEV)

    def adj(window):
        a = SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV), value=torch.from_numpy(val).to(DEV),
                         sparse_sizes=(n, n), is_sorted=True)
        a.schedule(2048, 1024, feat=F, sweep=window, window_bytes=1 << 20)
        return a

    side = torch.cuda.Stream()
    for window in (False, True):
        for reduce in ("sum", "mean", "max"):
            first = ops.spmm_raw(adj(window), x, reduce)
            for _ in range(3):
                assert torch.equal(ops.spmm_raw(adj(window), x, reduce), first), (window, reduce)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                other = ops.spmm_raw(adj(window), x, reduce)
            side.synchronize()
            assert torch.equal(other, first), (window, reduce, "another stream")
    a = adj(False)
    k1 = ops.khop(a, x, 3)
    assert torch.equal(ops.khop(a, x, 3), k1)
    ei = torch.from_numpy(np.random.default_rng(8).integers(0, n, (2, 300000))).to(DEV)
    w = torch.rand(300000, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) + 0.1
    g1 = utils.gcn_norm(ei, w, n)
    g2 = utils.gcn_norm(ei, w, n)
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    pos = a.set_value(a.storage.value().abs() + 0.1)              # negative row sums give NaN weights, as in the reference
    n1 = utils.normalize_adj_csr(pos)
    assert torch.equal(utils.normalize_adj_csr(pos).storage.value(), n1.storage.value()) and bool(torch.isfinite(n1.storage.value()).all())
    torch.manual_seed(0)
    m = PGE_Edge(nfeat=24, nhid=128, nlayers=3).to(DEV)
    Z = torch.randn(1500, 24, device=DEV)
    e1 = m.edge_construction(Z, 0.5)
    for _ in range(3):
        e2 = m.edge_construction(Z, 0.5)
        assert torch.equal(e1[0], e2[0]) and torch.equal(e1[1], e2[1])


@pytest.mark.parametrize("case", list(range(16)))
def test_random_misaligned_views_through_the_c_abi(case):
    """Operands that are column views starting 4, 8 or 12 bytes into a wider buffer (x, y or both), any leading dimension:
    the C entry point must fall back to the widest vector the pointers AND pitches allow, write nothing outside the view,
    and give the oracle's result -- for the plain, scheduled and windowed paths and K-hop scratch buffers alike."""
    from disco_amd import SparseTensor, _lib, ops
    rng = np.random.default_rng(8000 + case)
    n = int(rng.choice([50, 700, 3000]))
    F = int(rng.choice([1, 4, 31, 64, 100, 130, 256]))
    xoff, yoff = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    ldx = F + xoff + int(rng.choice([0, 1, 3, 4, 28]))
    ldy = F + yoff + int(rng.choice([0, 1, 3, 4, 28]))
    reduce = str(rng.choice(["sum", "mean", "max", "sum_acc"]))
    hubs = [(int(rng.integers(0, n)), 3000)] if n >= 700 and rng.random() < 0.5 else []
    rowptr, col, val = random_csr(n, n, float(rng.choice([3, 40])), seed=8000 + case, hubs=hubs, weighted=True, dup=False)
    x = rng.standard_normal((n, F)).astype(np.float32)
    y0 = rng.standard_normal((n, F)).astype(np.float32)
    xb = torch.full((n, ldx), 9.0, device=DEV)
    xb[:, xoff:xoff + F] = torch.from_numpy(x).to(DEV)
    yb = torch.full((n, ldy), 7.0, device=DEV)
    yv = yb[:, yoff:yoff + F]
    yv.copy_(torch.from_numpy(y0).to(DEV) if reduce == "sum_acc" else torch.zeros(n, F, device=DEV))
    adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(DEV), col=torch.from_numpy(col).to(DEV),
                       value=None if reduce == "max" else torch.from_numpy(val).to(DEV), sparse_sizes=(n, n), is_sorted=True)
    lib = _lib.load()
    window = bool(rng.random() < 0.4)
    use_sched = bool(rng.random() < 0.7)
    ws_ptr = info = sws = None
    sws_bytes = 0
    if use_sched:
        ws, info = adj.schedule(2048, 1024, feat=F, sweep=window, window_bytes=64 * 4 * min(128, (F + 31) // 32 * 32))
        ws_ptr = _lib.ptr(ws)
        nb = int(lib.disco_spmm_workspace_bytes(info, F))
        if nb:
            sws = torch.empty(nb, dtype=torch.uint8, device=DEV)
            sws_bytes = nb
    xv = xb[:, xoff:xoff + F]
    code = {"sum": 0, "mean": 1, "max": 2, "sum_acc": 3}[reduce]
    _lib.check(lib.disco_spmm_csr_f32(_lib.ptr(adj._rowptr), _lib.ptr(adj._col32), _lib.ptr(adj._value), n, xv.data_ptr(), ldx,
                                      yv.data_ptr(), ldy, F, code, ws_ptr, info, _lib.ptr(sws), sws_bytes, _lib.current_stream(torch.device(DEV))),
               "disco_spmm_csr_f32")
    torch.cuda.synchronize()
    got = yb.cpu().numpy()
    mask = np.ones((n, ldy), dtype=bool)
    mask[:, yoff:yoff + F] = False
    what = dict(case=case, n=n, F=F, xoff=xoff, yoff=yoff, ldx=ldx, ldy=ldy, reduce=reduce, sched=use_sched, window=window)
    assert (got[mask] == 7.0).all(), f"{what}: wrote outside the view"
    assert bool((xb[:, :xoff] == 9.0).all()) and bool((xb[:, xoff + F:] == 9.0).all())
    y = got[:, yoff:yoff + F]
    if reduce == "max":
        assert np.array_equal(y, c_oracle.spmm_csr_f32(rowptr, col, None, x, "max")), str(what)
    else:
        y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
        if reduce == "mean":
            cnt = np.maximum(np.diff(rowptr), 1)[:, None].astype(np.float64)
            y64, bound = y64 / cnt, bound / cnt
        if reduce == "sum_acc":
            y64, bound = y64 + y0.astype(np.float64), bound + np.abs(y0)
        long_seq = (not use_sched) & (np.diff(rowptr) > 2048)
        assert_fwd_close(y[~long_seq], y64[~long_seq], bound[~long_seq], what=str(what))
        assert_fwd_close(y[long_seq], y64[long_seq], bound[long_seq], what=str(what), rel_floor=np.inf)
