"""GPU parity of the HIP SpMM / K-hop / max path against the CPU oracle, through the C ABI
(disco_amd.ops -> ctypes -> libdisco_hip.so).  Tolerance: see helpers.RTOL (fp32, 1e-5 forward error)."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, assert_fwd_close, csr_to_coo, orc, random_csr
from oracle import c_oracle

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "graph_golden.npz"))
GRAPHS = ["path", "star", "loop0", "dup", "hub"]


def dev():
    return torch.device("cuda", 0)


def to_adj(rowptr, col, val, n_cols):
    from disco_amd import SparseTensor
    return SparseTensor(rowptr=torch.from_numpy(rowptr).to(dev()), col=torch.from_numpy(col).to(dev()),
                        value=None if val is None else torch.from_numpy(val).to(dev()),
                        sparse_sizes=(len(rowptr) - 1, n_cols), is_sorted=True)


def test_extension_is_loaded_from_tree():
    from disco_amd import _lib
    lib = _lib.load()
    assert lib.disco_abi_version() == 2
    maps = open("/proc/self/maps").read()
    assert "disco_amd/csrc/libdisco_hip.so" in maps
    assert len({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l}) == 1


@pytest.mark.parametrize("name", GRAPHS)
def test_golden_graphs(name):
    from disco_amd import ops
    n = int(GOLD[f"{name}_n"])
    rowptr, col, val = GOLD[f"{name}_adjt_rowptr"], GOLD[f"{name}_adjt_col"], GOLD[f"{name}_adjt_val"]
    adj = to_adj(rowptr, col, val, n)
    Fs = (1, 3, 100, 128, 602) if n <= 8 else (1, 3, 128)
    for F in Fs:
        x = GOLD[f"{name}_x{F}"]
        xd = torch.from_numpy(x).to(dev())
        bound1 = orc.spmm_abs_bound(rowptr, col, val, x)
        for K in (1, 2):
            y = ops.khop(adj, xd, K).cpu().numpy()
            bound = bound1 if K == 1 else orc.spmm_abs_bound(rowptr, col, val, bound1)
            assert_fwd_close(y, GOLD[f"{name}_khop{K}_F{F}_f64"], bound * K, what=f"{name} K={K} F={F}")
        if F in (3, 128):
            mx = ops.max_aggregate(adj, xd).cpu().numpy()
            assert np.array_equal(mx, GOLD[f"{name}_max_F{F}"]), "max aggregation must be bit exact"
            ym = ops.spmm(adj, xd, "mean").cpu().numpy()
            assert_fwd_close(ym, GOLD[f"{name}_mean_F{F}_f64"], bound1, what=f"{name} mean F={F}")


SHAPES = [  # n_rows, n_cols, avg_deg, F, hubs
    (300, 300, 6, 1, [(0, 100)]),
    (257, 400, 3, 3, []),
    (500, 500, 12, 16, [(4, 3000)]),          # 4 float4 per row: 4-lane groups (feature-partition slabs)
    (500, 600, 12, 24, [(4, 3000)]),          # 6 float4: 8-lane groups, two lanes idle
    (800, 800, 10, 32, [(7, 2600)]),          # 8 float4: 8-lane groups
    (1000, 1000, 14, 47, [(17, 5000)]),
    (777, 900, 9, 100, [(5, 2500), (6, 9000)]),
    (1500, 1500, 7, 128, [(1, 20000)]),
    (640, 640, 20, 256, [(2, 3000)]),
    (400, 512, 40, 602, [(3, 10000)]),
    (150, 200, 5, 1300, [(9, 2100)]),
    (90, 90, 4, 2600, []),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"n{s[0]}_F{s[3]}")
@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
@pytest.mark.parametrize("weighted", [True, False])
def test_spmm_parity(shape, reduce, weighted):
    from disco_amd import ops
    n_rows, n_cols, avg, F, hubs = shape
    rowptr, col, val = random_csr(n_rows, n_cols, avg, seed=n_rows + F, hubs=hubs, weighted=weighted)
    x = np.random.default_rng(F).standard_normal((n_cols, F)).astype(np.float32)
    adj = to_adj(rowptr, col, val, n_cols)
    xd = torch.from_numpy(x).to(dev())
    y = ops.spmm(adj, xd, reduce).cpu().numpy()
    assert y.shape == (n_rows, F)
    if reduce == "max":
        ref = c_oracle.spmm_csr_f32(rowptr, col, val, x, "max")
        assert np.array_equal(y, ref), "max is order independent: bit exact"
        return
    y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
    if reduce == "mean":
        cnt = np.maximum(np.diff(rowptr), 1)[:, None].astype(np.float64)
        y64, bound = y64 / cnt, bound / cnt
    assert_fwd_close(y, y64, bound, what=f"{reduce} {shape}")
    if not weighted and reduce == "sum":
        # unweighted sums of small integers are exact in fp32: must match bit for bit
        xi = np.random.default_rng(1).integers(-8, 8, (n_cols, F)).astype(np.float32)
        yi = ops.spmm(adj, torch.from_numpy(xi).to(dev()), "sum").cpu().numpy()
        assert np.array_equal(yi, c_oracle.spmm_csr_f32(rowptr, col, None, xi, "sum"))


@pytest.mark.parametrize("sched", [True, False])
def test_schedule_and_natural_order_agree(sched):
    from disco_amd import ops
    rowptr, col, val = random_csr(2000, 2000, 12, seed=11, hubs=[(4, 30000), (1999, 2049)], weighted=True)
    x = np.random.default_rng(2).standard_normal((2000, 96)).astype(np.float32)
    adj = to_adj(rowptr, col, val, 2000)
    y = ops.spmm_raw(adj, torch.from_numpy(x).to(dev()), "sum", use_schedule=sched).cpu().numpy()
    y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
    assert_fwd_close(y, y64, bound)
    if sched:
        ws, info = adj.schedule()
        # small graph: default threshold 256, pieces of <= 1024 entries -> 30 + 3 pieces, both rows multi-piece
        assert info.hub_threshold == 256 and info.hub_piece == 1024
        assert info.n_hub_rows == 2 and info.n_multi_rows == 2 and info.max_row_nnz == 30000
        assert info.n_hub_pieces == info.n_hub_partials == 33
        ws2, info2 = to_adj(rowptr, col, val, 2000).schedule(2048, 8192)
        assert info2.n_hub_rows == 2 and info2.n_multi_rows == 1 and info2.n_hub_pieces == 1 + info2.n_hub_partials == 5
        # deterministic: two launches give identical bits
        y2 = ops.spmm_raw(adj, torch.from_numpy(x).to(dev()), "sum").cpu().numpy()
        assert np.array_equal(y, y2)


def test_strided_and_unaligned_operands():
    # ld = 602 (8-byte rows -> 64-bit loads), ld = 601 (scalar loads), and a padded ld = 608 view
    from disco_amd import ops
    rowptr, col, val = random_csr(300, 300, 10, seed=5, hubs=[(7, 2500)], weighted=True)
    adj = to_adj(rowptr, col, val, 300)
    for F, ld in ((602, 602), (601, 601), (602, 608), (601, 604), (100, 100), (127, 127), (127, 128), (5, 8)):
        x = np.random.default_rng(F).standard_normal((300, F)).astype(np.float32)
        buf = torch.zeros((300, ld), device=dev())
        buf[:, :F] = torch.from_numpy(x).to(dev())
        xd = buf[:, :F]
        full = torch.full((300, ld), 7.0, device=dev())
        out = full[:, :F]
        y = ops.spmm_raw(adj, xd, "sum", out=out).cpu().numpy()
        y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
        assert_fwd_close(y, y64, bound, what=f"F={F} ld={ld}")
        # columns F.. of the caller's buffer belong to the caller: the last partial vector is stored masked
        assert bool((full[:, F:] == 7.0).all()), f"F={F} ld={ld}: padding columns were written"


@pytest.mark.parametrize("K,slices", [(1, 1), (2, 1), (2, 3), (3, 2), (4, 5)])
def test_khop_matches_repeated_oracle(K, slices):
    from disco_amd import ops
    n, F = 1200, 200
    rowptr, col, val = random_csr(n, n, 8, seed=21, hubs=[(0, 4000)], weighted=True)
    val = (val * 0.2).astype(np.float32)
    x = np.random.default_rng(3).standard_normal((n, F)).astype(np.float32)
    adj = to_adj(rowptr, col, val, n)
    y = ops.khop(adj, torch.from_numpy(x).to(dev()), K, n_col_slices=slices).cpu().numpy()
    y64 = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert_fwd_close(y, y64, bound * K, what=f"K={K} slices={slices}")
    # Slicing the columns does not change a bit of a single hop on rows below the hub threshold
    # (strict CSR order per row whatever the lane-group width).  The hub row's partial sums are grouped
    # by lane group, so it -- and from the second hop on, every row that gathers it -- moves by rounding.
    y1 = ops.khop(adj, torch.from_numpy(x).to(dev()), K, n_col_slices=1).cpu().numpy()
    if K == 1:
        assert np.array_equal(y[1:], y1[1:])
    assert_fwd_close(y1, y64, bound * K, what=f"K={K} slices=1")


def test_empty_and_degenerate_inputs():
    from disco_amd import SparseTensor, ops
    adj = SparseTensor(rowptr=torch.zeros(6, dtype=torch.int64, device=dev()), col=torch.zeros(0, dtype=torch.int64, device=dev()),
                       sparse_sizes=(5, 5), is_sorted=True)
    x = torch.randn(5, 16, device=dev())
    assert torch.equal(ops.spmm(adj, x, "sum"), torch.zeros(5, 16, device=dev()))
    assert torch.equal(ops.spmm(adj, x, "max"), torch.zeros(5, 16, device=dev()))
    assert ops.spmm(adj, torch.randn(5, 0, device=dev()), "sum").shape == (5, 0)
    adj0 = SparseTensor(rowptr=torch.zeros(1, dtype=torch.int64, device=dev()), col=torch.zeros(0, dtype=torch.int64, device=dev()),
                        sparse_sizes=(0, 5), is_sorted=True)
    assert ops.spmm(adj0, x, "sum").shape == (0, 16)
    with pytest.raises(ValueError):
        ops.spmm(adj, torch.randn(3, 16, device=dev()), "sum")
    with pytest.raises(TypeError):
        ops.spmm(adj, x.double(), "sum")


def test_full_size_properties_arxiv_shaped():
    """BASELINE config 1 (ogbn-arxiv shape) at full size: properties that need no oracle run --
    linearity, A*1 == row sums, and agreement of a 4096-row sample with the float64 oracle."""
    from disco_amd import SparseTensor, graphgen, ops, utils
    spec = graphgen.SPECS["arxiv"]
    row, col = graphgen.powerlaw_edges(spec.n_nodes, spec.nnz, spec.skew, spec.seed, device=dev())
    n = spec.n_nodes
    adj = SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True)
    adj = utils.normalize_adj_csr(adj)            # +I, row degree, float64 -> float32 (utils.py:406-435)
    F = spec.feat
    x = graphgen.features(n, F, spec.seed, device=dev())
    z = graphgen.features(n, F, spec.seed + 1, device=dev())
    y = ops.khop(adj, x, 2)
    yz = ops.khop(adj, z, 2)
    ylin = ops.khop(adj, 0.5 * x - 2.0 * z, 2)
    scale = ops.khop(adj.set_value(adj.storage.value().abs()), x.abs() + 2 * z.abs(), 2)
    assert bool(((ylin - (0.5 * y - 2.0 * yz)).abs() <= 4e-5 * scale + 1e-30).all())
    ones = torch.ones(n, 4, device=dev())
    rs = torch.zeros(n, dtype=torch.float64, device=dev()).index_add_(0, adj.storage.row(), adj.storage.value().double())
    a1 = ops.spmm(adj, ones, "sum")[:, 0].double()
    assert bool(((a1 - rs).abs() <= 1e-5 * rs.abs() + 1e-12).all())
    # sample of rows against the float64 oracle
    rp, c, v = (t.cpu().numpy() for t in adj.csr())
    xs = x.cpu().numpy()
    y1 = ops.spmm(adj, x, "sum").cpu().numpy()
    rows = np.random.default_rng(0).choice(n, 4096, replace=False)
    rows.sort()
    sub_rp = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(rp[rows + 1] - rp[rows], out=sub_rp[1:])
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
    y64, bound = c_oracle.spmm_csr_f64acc(sub_rp, c[idx], v[idx], xs, want_bound=True)
    assert_fwd_close(y1[rows], y64, bound, what="arxiv-shaped sample")


def test_full_size_properties_reddit_shaped():
    """BASELINE config 2 (Reddit shape: N = 232 965, nnz = 114.5 M, F = 602) at full size, the bench workload,
    on the default (windowed, L2-blocked) path: linearity, A*1 == row sums, the windowed result has the same
    bits as the plain kernel on every non-hub row, and a 512-row sample agrees with the float64 oracle."""
    from disco_amd import SparseTensor, graphgen, ops, utils
    spec = graphgen.SPECS["reddit"]
    n, F = spec.n_nodes, spec.feat
    row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev())
    adj = utils.normalize_adj_csr(SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True))
    del row, col
    x = graphgen.features(n, F, spec.seed, device=dev(), ld=608)
    z = graphgen.features(n, F, spec.seed + 1, device=dev(), ld=608)
    ws, info = adj.schedule(feat=F)
    assert info.sweep_ws and info.sweep_windows >= 2, "the Reddit shape must take the windowed path"
    y = ops.spmm_raw(adj, x, "sum")
    yz = ops.spmm_raw(adj, z, "sum")
    ylin = ops.spmm_raw(adj, 0.5 * x - 2.0 * z, "sum")
    scale = ops.spmm_raw(adj, x.abs() + 2 * z.abs(), "sum")          # weights are positive
    assert bool(((ylin - (0.5 * y - 2.0 * yz)).abs() <= 2e-5 * scale + 1e-30).all())
    del yz, ylin, scale, z
    rs = torch.zeros(n, dtype=torch.float64, device=dev()).index_add_(0, adj.storage.row(), adj.storage.value().double())
    a1 = ops.spmm_raw(adj, torch.ones(n, 4, device=dev()), "sum")[:, 0].double()
    assert bool(((a1 - rs).abs() <= 1e-5 * rs.abs() + 1e-12).all())
    # same bits as the plain kernel (same hub split) on the rows the windowed kernel owns
    plain = adj.set_value(adj.storage.value())
    ws_p, info_p = plain.schedule(int(info.hub_threshold), int(info.hub_piece), feat=F, sweep=False)
    assert not info_p.sweep_ws and adj.schedule(feat=F)[1].sweep_ws, "a copy's choice must not leak into its parent"
    y_plain = ops.spmm_raw(plain, x, "sum")
    assert not plain.schedule(feat=F)[1].sweep_ws, "sweep=False is sticky: spmm_raw must not re-attach the table"
    deg = adj.storage.rowptr()[1:] - adj.storage.rowptr()[:-1]
    keep = deg <= int(info.hub_threshold)
    assert int((~keep).sum()) == int(info.n_hub_rows) > 0
    assert torch.equal(y[keep], y_plain[keep])
    # sample against the float64 oracle
    rp, c, v = (t.cpu().numpy() for t in adj.csr())
    rows = np.sort(np.random.default_rng(0).choice(n, 512, replace=False))
    sub_rp = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(rp[rows + 1] - rp[rows], out=sub_rp[1:])
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
    y64, bound = c_oracle.spmm_csr_f64acc(sub_rp, c[idx], v[idx], x.cpu().numpy(), want_bound=True)
    assert_fwd_close(y.cpu().numpy()[rows], y64, bound, what="reddit-shaped sample")


@pytest.mark.parametrize("F", [64, 128, 602])
@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_sweep_kernel_same_bits_as_plain(F, reduce, mode):
    """The L2-blocked variants (1 = persistent sweep kernel, 2 = windowed launches) visit each row's entries
    in CSR order: same bits as the plain kernel on the rows they own (all rows below the hub threshold)."""
    from disco_amd import _lib, ops
    _lib.load().disco_set_tuning(b"sweep_mode", mode)
    n = 5000
    rowptr, col, val = random_csr(n, n, 60, seed=31, hubs=[(0, 5000), (7, 2100)], weighted=True)
    x = np.random.default_rng(8).standard_normal((n, F)).astype(np.float32)
    ld = (F + 31) // 32 * 32
    xb = torch.zeros((n, ld), device=dev())
    xb[:, :F] = torch.from_numpy(x).to(dev())
    xd = xb[:, :F]
    adj = to_adj(rowptr, col, val, n)
    ws, info = adj.schedule(feat=F, sweep=False)
    assert not info.sweep_ws
    y_plain = ops.spmm_raw(adj, xd, reduce, out=torch.empty((n, ld), device=dev())[:, :F]).cpu().numpy()
    adj2 = to_adj(rowptr, col, val, n)
    ws2, info2 = adj2.schedule(feat=F, sweep=True)
    assert info2.sweep_ws and info2.sweep_windows >= 1
    y_sweep = ops.spmm_raw(adj2, xd, reduce, out=torch.empty((n, ld), device=dev())[:, :F]).cpu().numpy()
    # rows 0 and 7 are hub rows: their partial sums follow the lane-group width, which the narrower column
    # blocks of the L2-blocked modes change; every other row must agree bit for bit
    keep = np.ones(n, dtype=bool)
    keep[[0, 7]] = False
    assert np.array_equal(y_plain[keep], y_sweep[keep])
    if reduce == "max":          # order independent: every row, and the oracle, bit for bit (modes 1/3 fall back to windows)
        assert np.array_equal(y_plain, y_sweep)
        assert np.array_equal(y_sweep, c_oracle.spmm_csr_f32(rowptr, col, val, x, "max"))
        _lib.load().disco_set_tuning(b"sweep_mode", 2)
        return
    y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
    if reduce == "mean":
        cnt = np.maximum(np.diff(rowptr), 1)[:, None].astype(np.float64)
        y64, bound = y64 / cnt, bound / cnt
    assert_fwd_close(y_sweep, y64, bound, what=f"sweep {reduce} F={F}")
    _lib.load().disco_set_tuning(b"sweep_mode", 2)


@pytest.mark.parametrize("F,wrows", [(128, 300), (128, 1500), (602, 97), (100, 640), (30, 4000)])
@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "sum_acc"])
def test_xcd_sliced_schedule_same_bits_as_windowed_launches(F, wrows, reduce):
    """sweep_mode 5 (XCD-sliced skewed schedule: in phase t the workgroups with blockIdx % 8 == j work on slice s_j and
    row group t - s_j): every row still visits the slices in ascending order, one per phase, so the bits are those
    of the windowed launches (mode 2) and, below the hub threshold, of the plain kernel -- with many slices, fewer
    slices than XCDs, empty rows, a masked tail column block and y += A x."""
    from disco_amd import _lib, ops
    lib = _lib.load()
    n = 5000
    rowptr, col, val = random_csr(n, n, 60, seed=77, hubs=[(0, 4800), (9, 2100)], weighted=True)
    x = np.random.default_rng(8).standard_normal((n, F)).astype(np.float32)
    ld = (F + 31) // 32 * 32
    xb = torch.zeros((n, ld), device=dev())
    xb[:, :F] = torch.from_numpy(x).to(dev())
    y0 = torch.from_numpy(np.random.default_rng(5).standard_normal((n, F)).astype(np.float32)).to(dev())

    def run(mode):
        adj = to_adj(rowptr, col, val, n)
        _, info = adj.schedule(8192, 4096, feat=F, sweep=True, window_bytes=wrows * 4 * min(128, (F + 31) // 32 * 32))
        assert info.sweep_ws and info.n_hub_rows == 0
        lib.disco_set_tuning(b"sweep_mode", mode)
        try:
            out = torch.full((n, ld), 7.0, device=dev())
            out[:, :F] = y0 if reduce == "sum_acc" else 0.0
            ops.spmm_raw(adj, xb[:, :F], reduce, out=out[:, :F])
            torch.cuda.synchronize()
        finally:
            lib.disco_set_tuning(b"sweep_mode", 2)
        return out, int(info.sweep_windows)

    y2, windows = run(2)
    y5, _ = run(5)
    assert windows == (n + wrows - 1) // wrows
    assert torch.equal(y2, y5)
    assert bool((y5[:, F:] == 7.0).all())
    plain = to_adj(rowptr, col, val, n)
    plain.schedule(8192, 4096, feat=F, sweep=False)
    out = torch.full((n, ld), 7.0, device=dev())
    out[:, :F] = y0 if reduce == "sum_acc" else 0.0
    ops.spmm_raw(plain, xb[:, :F], reduce, out=out[:, :F])
    if reduce != "sum_acc":      # per output element the same fma chain in CSR order, whatever the lane-group shape
        assert torch.equal(out, y5)  # (y += A x: the plain kernel adds y0 last, the windowed ones start from it)


def test_sweep_refuses_unsorted_columns():
    from disco_amd import ops
    rowptr, col, val = random_csr(300, 300, 20, seed=3, weighted=True, sort_cols=False)
    adj = to_adj(rowptr, col, val, 300)
    ws, info = adj.schedule(feat=128, sweep=True)
    assert not info.sweep_ws                       # plain kernel stays in use
    x = np.random.default_rng(1).standard_normal((300, 128)).astype(np.float32)
    y = ops.spmm_raw(adj, torch.from_numpy(x).to(dev()), "sum").cpu().numpy()
    y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
    assert_fwd_close(y, y64, bound)


@pytest.mark.parametrize("sweep", [False, True])
def test_sum_acc_accumulates_onto_existing_output(sweep):
    """DISCO_REDUCE_SUM_ACC (y += A x, the halo step of the row partition), plain and windowed: the running sum
    starts from the value already in y, so the result is fl(y0 + sum) with the sum continued in CSR order."""
    from disco_amd import ops
    n, F = 4000, 128
    rowptr, col, val = random_csr(n, n, 70, seed=41, hubs=[(3, 9000)], weighted=True)
    x = np.random.default_rng(9).standard_normal((n, F)).astype(np.float32)
    y0 = np.random.default_rng(10).standard_normal((n, F)).astype(np.float32)
    adj = to_adj(rowptr, col, val, n)
    ws, info = adj.schedule(feat=F, sweep=sweep)
    assert bool(info.sweep_ws) == sweep
    out = torch.from_numpy(y0.copy()).to(dev())
    ops.spmm_raw(adj, torch.from_numpy(x).to(dev()), "sum_acc", out=out)
    y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
    assert_fwd_close(out.cpu().numpy(), y64 + y0.astype(np.float64), bound + np.abs(y0), what=f"sum_acc sweep={sweep}")
    empty = np.diff(rowptr) == 0
    assert empty.any() and np.array_equal(out.cpu().numpy()[empty], y0[empty])      # rows without entries keep y0


@pytest.mark.parametrize("ld", [602, 601, 604])
def test_windowed_path_with_narrow_vectors(ld):
    """ld = 602: rows are only 8-byte aligned -> 64-bit gathers, still windowed; ld = 601: scalar loads, the
    window table is ignored and the plain kernel runs; ld = 604: 128-bit gathers with a partial last vector
    (F = 602), stored masked.  All must meet the forward-error bar and leave the padding columns alone."""
    from disco_amd import _lib, ops
    lib = _lib.load()
    n, F = 3000, min(ld, 602)
    rowptr, col, val = random_csr(n, n, 50, seed=51, hubs=[(1, 2500)], weighted=True)
    x = np.random.default_rng(3).standard_normal((n, F)).astype(np.float32)
    buf = torch.zeros((n, ld), device=dev())
    buf[:, :F] = torch.from_numpy(x).to(dev())
    adj = to_adj(rowptr, col, val, n)
    ws, info = adj.schedule(feat=F, sweep=True)
    assert info.sweep_ws
    sws = torch.empty(max(int(lib.disco_spmm_workspace_bytes(info, F)), 16), dtype=torch.uint8, device=dev())
    out = torch.full((n, ld), 7.0, device=dev())
    _lib.check(lib.disco_spmm_csr_f32(adj._rowptr.data_ptr(), adj._col32.data_ptr(), adj._value.data_ptr(), n,
                                      buf.data_ptr(), ld, out.data_ptr(), ld, F, 0, ws.data_ptr(), info, sws.data_ptr(),
                                      sws.numel(), _lib.current_stream(dev())), "spmm")
    y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x, want_bound=True)
    got = out.cpu().numpy()
    assert_fwd_close(got[:, :F], y64, bound, what=f"ld={ld}")
    assert np.all(got[:, F:] == 7.0)             # padding columns are never written


@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
def test_multi_window_resume_same_bits_as_plain(reduce):
    """Several windows on a small graph (window of 64 source rows -> 47 windows): the running sum passes through
    y between window launches and must come out with the same bits as the one-pass plain kernel."""
    from disco_amd import ops
    n, F = 3000, 96
    rowptr, col, val = random_csr(n, n, 90, seed=77, hubs=[(5, 2900)], weighted=True)
    x = torch.from_numpy(np.random.default_rng(4).standard_normal((n, F)).astype(np.float32)).to(dev())
    plain = to_adj(rowptr, col, val, n)
    _, ip = plain.schedule(8192, 4096, feat=F, sweep=False)
    win = to_adj(rowptr, col, val, n)
    _, iw = win.schedule(8192, 4096, feat=F, sweep=True, window_bytes=64 * 4 * F)
    assert not ip.sweep_ws and iw.sweep_ws and iw.sweep_windows >= 3 and iw.n_hub_rows == ip.n_hub_rows == 0
    y_p = ops.spmm_raw(plain, x, reduce)
    y_w = ops.spmm_raw(win, x, reduce)
    assert not plain.schedule(feat=F)[1].sweep_ws and win.schedule(feat=F)[1].sweep_windows == iw.sweep_windows
    assert torch.equal(y_p, y_w)


def test_windowed_path_with_output_block_beyond_4gib():
    """The windowed kernel's write-through stores address y with 32-bit offsets; an output whose rows span 4 GiB or
    more must take the plain-store variant -- same bits either way."""
    from disco_amd import ops
    n, F, ld = 280_000, 64, 4096                       # 280 000 rows x 16 KiB pitch = 4.6 GB
    rowptr, col, val = random_csr(n, n, 6, seed=13, weighted=True)
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((n, F)).astype(np.float32)).to(dev())
    small = to_adj(rowptr, col, val, n)
    small.schedule(feat=F, sweep=True, window_bytes=50_000 * 4 * F)
    y_small = ops.spmm_raw(small, x, "sum").clone()
    big = torch.full((n, ld), 3.0, device=dev())
    adj = to_adj(rowptr, col, val, n)
    _, info = adj.schedule(feat=F, sweep=True, window_bytes=50_000 * 4 * F)
    assert info.sweep_windows >= 5
    y = ops.spmm_raw(adj, x, "sum", out=big[:, :F])
    assert y.data_ptr() == big.data_ptr() and torch.equal(y, y_small)
    assert bool((big[::977, F:] == 3.0).all())
    y64, bound = c_oracle.spmm_csr_f64acc(rowptr, col, val, x.cpu().numpy(), want_bound=True)
    assert_fwd_close(y_small.cpu().numpy(), y64, bound)


def test_set_value_copies_do_not_share_schedule_state():
    """A set_value copy that attaches a window table for another feature width (and is then freed) must leave
    its parent's schedule untouched: every object owns its host-side schedule info."""
    import gc
    from disco_amd import ops
    n, F1, F2 = 4000, 128, 32
    rowptr, col, val = random_csr(n, n, 80, seed=91, weighted=True)
    base = to_adj(rowptr, col, val, n)
    x1 = torch.from_numpy(np.random.default_rng(5).standard_normal((n, F1)).astype(np.float32)).to(dev())
    x2 = torch.from_numpy(np.random.default_rng(6).standard_normal((n, F2)).astype(np.float32)).to(dev())
    _, info = base.schedule(feat=F1, sweep=True, window_bytes=256 * 4 * F1)
    windows, table = int(info.sweep_windows), int(info.sweep_ws)
    y_ref = ops.spmm_raw(base, x1, "sum").clone()
    child = base.set_value(None)
    _, ci = child.schedule(feat=F2, sweep=True, window_bytes=97 * 4 * F2)
    assert ci is not info and int(ci.sweep_windows) != windows
    ops.spmm_raw(child, x2, "max")
    off = base.set_value(base.storage.value())
    off.schedule(feat=F1, sweep=False)
    del child, ci, off
    gc.collect()
    torch.cuda.empty_cache()
    junk = torch.full((1 << 24,), 0x7f7f7f7f, dtype=torch.int32, device=dev())   # re-use whatever was freed
    _, again = base.schedule(feat=F1)
    assert int(again.sweep_windows) == windows and int(again.sweep_ws) == table
    assert torch.equal(ops.spmm_raw(base, x1, "sum"), y_ref)
    del junk


@pytest.mark.parametrize("F", [602, 256, 160])
@pytest.mark.parametrize("reduce", ["sum", "mean", "sum_acc"])
@pytest.mark.parametrize("weighted", [True, False])
def test_streaming_window_kernel_same_bits_as_plain(F, reduce, weighted):
    """sweep_mode 4 (k_spmm_stream: sums in registers through all windows, one flat gather stream per wave, paced by
    per-step counters): entries of a row are consumed in CSR order, so every row below the hub threshold must come
    out with the bits of the one-pass plain kernel -- with windows that hold 0, few and > 64 entries per wave, rows
    that end in the middle of a chunk, empty rows, and more rows than one sweep of the grid covers."""
    from disco_amd import _lib, ops
    lib = _lib.load()
    n = 9000
    rowptr, col, val = random_csr(n, n, 70, seed=123, hubs=[(5, 7000), (11, 3000), (8000, 900)], weighted=weighted)
    x = torch.from_numpy(np.random.default_rng(4).standard_normal((n, F)).astype(np.float32)).to(dev())
    ld = (F + 31) // 32 * 32
    xb = torch.zeros((n, ld), device=dev())
    xb[:, :F] = x
    y0 = torch.from_numpy(np.random.default_rng(5).standard_normal((n, F)).astype(np.float32)).to(dev())

    def run(adj):
        out = torch.full((n, ld), 7.0, device=dev())
        out[:, :F] = y0 if reduce == "sum_acc" else 0.0
        ops.spmm_raw(adj, xb[:, :F], reduce, out=out[:, :F])
        return out

    plain = to_adj(rowptr, col, val, n)
    plain.schedule(8192, 4096, feat=F, sweep=False)
    y_p = run(plain)
    win = to_adj(rowptr, col, val, n)
    _, iw = win.schedule(8192, 4096, feat=F, sweep=True, window_bytes=300 * 512)
    assert iw.sweep_ws and iw.sweep_windows >= 20 and iw.n_hub_rows == 0
    lib.disco_set_tuning(b"sweep_mode", 4)
    lib.disco_set_tuning(b"sweep_blocks_per_cu", 1)        # 256 workgroups x 4 waves x 8 rows < n: several sweeps
    try:
        y_s = run(win)
        lib.disco_set_tuning(b"stream_lead", -1)            # same result without pacing
        y_u = run(win)
    finally:
        lib.disco_set_tuning(b"sweep_mode", 2)
        lib.disco_set_tuning(b"sweep_blocks_per_cu", 0)
        lib.disco_set_tuning(b"stream_lead", 0)
    assert torch.equal(y_p, y_s) and torch.equal(y_p, y_u)
    assert bool((y_s[:, F:] == 7.0).all())


def _oracle_row_sample(adj, x, y, n_sample, seed, what):
    """Rows of y against the float64 oracle on the sub-matrix of those rows (columns renumbered to the rows of x they touch)."""
    rp = adj.storage.rowptr()
    n = adj.size(0)
    rows = torch.from_numpy(np.sort(np.random.default_rng(seed).choice(n, n_sample, replace=False))).to(rp.device)
    beg, end = rp[rows], rp[rows + 1]
    cnt = end - beg
    sub_rp = torch.zeros(n_sample + 1, dtype=torch.int64, device=rp.device)
    torch.cumsum(cnt, 0, out=sub_rp[1:])
    idx = torch.repeat_interleave(beg - sub_rp[:-1], cnt) + torch.arange(int(sub_rp[-1]), device=rp.device)
    cols = adj._col32[idx].to(torch.int64)
    vals = adj.storage.value()[idx]
    uniq, inv = torch.unique(cols, return_inverse=True)
    y64, bound = c_oracle.spmm_csr_f64acc(sub_rp.cpu().numpy(), inv.cpu().numpy(), vals.cpu().numpy(), x[uniq].cpu().numpy(),
                                          want_bound=True)
    assert_fwd_close(y[rows].cpu().numpy(), y64, bound, what=what)


def test_full_size_properties_products_shaped():
    """BASELINE config 3 (ogbn-products shape: N = 2 449 029, nnz = 61 859 140 + N self-loops, F = 100) at full size on
    one GPU: exact entry count, linearity, A*1 == row sums (hub row of several 10^5 entries included), a 2048-row
    sample against the float64 oracle, and the same bits with the XCD-aware block order."""
    from disco_amd import SparseTensor, _lib, graphgen, ops, utils
    spec = graphgen.SPECS["products"]
    n, F = spec.n_nodes, spec.feat
    row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev())
    assert row.numel() == spec.nnz
    adj = utils.normalize_adj_csr(SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True))
    del row, col
    assert adj.nnz() == spec.nnz + n
    x = graphgen.features(n, F, spec.seed, device=dev(), ld=128)
    z = graphgen.features(n, F, spec.seed + 1, device=dev(), ld=128)
    y = ops.spmm_raw(adj, x, "sum")
    yz = ops.spmm_raw(adj, z, "sum")
    ylin = ops.spmm_raw(adj, 0.5 * x - 2.0 * z, "sum")
    scale = ops.spmm_raw(adj, x.abs() + 2 * z.abs(), "sum")
    assert bool(((ylin - (0.5 * y - 2.0 * yz)).abs() <= 2e-5 * scale + 1e-30).all())
    del yz, ylin, scale, z
    # float64 row sums by a prefix sum (an index_add_ would serialise its atomics on the hub row)
    cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev()), torch.cumsum(adj.storage.value().double(), 0)])
    rs = cs[adj.storage.rowptr()[1:]] - cs[adj.storage.rowptr()[:-1]]
    a1 = ops.spmm_raw(adj, torch.ones(n, 4, device=dev()), "sum")[:, 0].double()
    assert bool(((a1 - rs).abs() <= 1e-5 * rs.abs() + 1e-9).all())
    _oracle_row_sample(adj, x, y, 2048, 0, "products-shaped sample")
    lib = _lib.load()
    lib.disco_set_tuning(b"xcd_map", 1)
    try:
        assert torch.equal(ops.spmm_raw(adj, x, "sum"), y)
    finally:
        lib.disco_set_tuning(b"xcd_map", -1)       # the library default: info->xcd_run decides
    y2 = ops.khop(adj, x, 2)
    assert bool(torch.isfinite(y2).all())


def test_papers100m_shaped_row_shard_with_global_columns():
    """BASELINE config 5: rank 3's row block of the 8-way partition of the ogbn-papers100M-shaped graph (13.9 M rows,
    ~202 M entries) with GLOBAL column ids into the 111 M-row feature table -- the rectangular operand a row partition
    hands to disco_spmm_csr_f32 (int32 columns up to 1.1e8, 64-bit gather addresses: the table is 14 GB at the F = 32 used
    here).  A*1 == row sums and a row sample against the float64 oracle."""
    from disco_amd import SparseTensor, graphgen, ops
    spec = graphgen.SPECS["papers100M"]
    n_global, world, rank, F = spec.n_nodes, 8, 3, 32
    row, col, offs = graphgen.partition_edges(n_global, spec.nnz, spec.skew, spec.seed, world, rank, device=dev())
    n_local = offs[rank + 1] - offs[rank]
    assert abs(row.numel() - spec.nnz // world) < 0.02 * spec.nnz // world and int(col.max()) > n_global - n_global // 64
    deg = torch.bincount(row, minlength=n_local).to(torch.float32)
    val = (1.0 / deg.clamp(min=1.0))[row]                      # row-stochastic weights: A * 1 == 1 on non-empty rows
    adj = SparseTensor(row=row, col=col, value=val, sparse_sizes=(n_local, n_global), is_sorted=True)
    del row, col, val
    x = graphgen.features(n_global, F, spec.seed, device=dev())
    y = ops.spmm_raw(adj, x, "sum")
    assert y.shape == (n_local, F)
    _oracle_row_sample(adj, x, y, 1024, 1, "papers100M-shaped shard sample")
    a1 = ops.spmm_raw(adj, torch.ones(n_global, 4, device=dev()), "sum")[:, 0]
    want = (deg > 0).to(torch.float32)
    assert bool(((a1 - want).abs() <= 1e-4).all())
