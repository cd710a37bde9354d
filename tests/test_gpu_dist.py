"""Row partition on the GPU: two ranks share the one card of the test box, the halo rows travel over gloo
(staged through host memory -- RCCL refuses two ranks on one device), the interior / halo SpMMs are the HIP
kernels (``sum`` then ``sum_acc``).  The assembled K-hop result must meet the forward-error bar against
the oracle's single-process result on the global matrix."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import RTOL, orc, random_csr
from test_dist_cpu import _free_port

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, F, K, result_dir, dense=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd.dist import RowPartition, balanced_row_offsets
        dev = torch.device("cuda", 0)
        rowptr, col, val = random_csr(n, n, 9, seed=78, hubs=[(3, 2600), (n - 2, 900)], weighted=True)
        x = np.random.default_rng(6).standard_normal((n, F)).astype(np.float32)
        offs = balanced_row_offsets(torch.from_numpy(rowptr), world)
        lo, hi = offs[rank], offs[rank + 1]
        rp = torch.from_numpy(rowptr[lo:hi + 1] - rowptr[lo]).to(dev)
        sl = slice(rowptr[lo], rowptr[hi])
        part = RowPartition(rp, torch.from_numpy(col[sl]).to(dev), torch.from_numpy(val[sl]).to(dev), offs, rank, world,
                            exchange="dense" if dense else "need")
        assert part.dense == dense
        ld = (F + 31) // 32 * 32
        cur = torch.zeros((hi - lo, ld), device=dev)[:, :F]
        cur.copy_(torch.from_numpy(x[lo:hi]))
        for _ in range(K):
            out = torch.empty((hi - lo, ld), device=dev)[:, :F]
            part.spmm(cur, out)
            cur = out
        torch.cuda.synchronize()
        np.save(os.path.join(result_dir, f"y{rank}.npy"), cur.cpu().numpy())
        np.save(os.path.join(result_dir, f"halo{rank}.npy"), np.array([part.n_halo, int(part.send_idx.numel())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("F,dense", [(40, False), (602, False), (602, True)])
def test_two_ranks_one_gpu_matches_global_oracle(F, dense, tmp_path):
    n, K, world = 3000, 2, 2
    mp.spawn(_worker, args=(world, _free_port(), n, F, K, str(tmp_path), dense), nprocs=world, join=True)
    rowptr, col, val = random_csr(n, n, 9, seed=78, hubs=[(3, 2600), (n - 2, 900)], weighted=True)
    x = np.random.default_rng(6).standard_normal((n, F)).astype(np.float32)
    ref = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert np.all(np.abs(got - ref) <= RTOL * K * bound + 1e-30)
    halo = np.array([np.load(tmp_path / f"halo{r}.npy") for r in range(world)])
    assert halo[:, 0].sum() > 0 and (dense or halo[:, 0].sum() == halo[:, 1].sum())   # need-only: received == sent


def _scorer_worker(rank, world, port, n, d, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd.dist import sharded_edge_construction
        from disco_amd.models.parametrized_adj_lp import PGE_Edge
        dev = torch.device("cuda", 0)
        torch.manual_seed(3)
        pge = PGE_Edge(nfeat=d, nhid=64, device=dev).to(dev).eval()
        Z = torch.randn(n, d, device=dev)
        for variant in ("transductive", "inductive", "sampled"):
            ei, ew = sharded_edge_construction(pge, Z, 0.5, variant)
            torch.save((ei.cpu(), ew.cpu()), os.path.join(result_dir, f"{variant}{rank}.pt"))
        if rank == 0:
            for variant in ("transductive", "inductive", "sampled"):
                ei, ew = pge.edge_construction(Z, 0.5, variant)
                torch.save((ei.cpu(), ew.cpu()), os.path.join(result_dir, f"{variant}_single.pt"))
    finally:
        dist.destroy_process_group()


def test_sharded_edge_construction_equals_single_gpu(tmp_path):
    n, d, world = 333, 12, 2          # 333 rows do not divide evenly: the last block is padded
    mp.spawn(_scorer_worker, args=(world, _free_port(), n, d, str(tmp_path)), nprocs=world, join=True)
    for variant in ("transductive", "inductive", "sampled"):
        ref_ei, ref_ew = torch.load(tmp_path / f"{variant}_single.pt", weights_only=True)
        assert ref_ei.shape[1] > n
        for r in range(world):
            ei, ew = torch.load(tmp_path / f"{variant}{r}.pt", weights_only=True)
            assert torch.equal(ei, ref_ei) and torch.equal(ew, ref_ew), (variant, r)


def _norm_worker(rank, world, port, n, with_loop0, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd import SparseTensor, utils
        from disco_amd.dist import balanced_row_offsets, normalize_adj_partition
        dev = torch.device("cuda", 0)
        rowptr, col, val = _norm_graph(n, with_loop0)
        offs = balanced_row_offsets(torch.from_numpy(rowptr), world)
        lo, hi = offs[rank], offs[rank + 1]
        sl = slice(rowptr[lo], rowptr[hi])
        rp, c, v = normalize_adj_partition(torch.from_numpy(rowptr[lo:hi + 1] - rowptr[lo]).to(dev), torch.from_numpy(col[sl]).to(dev),
                                           torch.from_numpy(val[sl]).to(dev), offs, rank, world)
        torch.save((rp.cpu(), c.cpu(), v.cpu(), lo, hi), os.path.join(result_dir, f"n{rank}.pt"))
        if rank == 0:
            adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(dev), col=torch.from_numpy(col).to(dev),
                               value=torch.from_numpy(val).to(dev), sparse_sizes=(n, n), is_sorted=True)
            g = utils.normalize_adj_csr(adj)
            torch.save(tuple(t.cpu() for t in g.csr()), os.path.join(result_dir, "single.pt"))
    finally:
        dist.destroy_process_group()


def _norm_graph(n, with_loop0):
    rowptr, col, val = random_csr(n, n, 7, seed=91, hubs=[(5, 700)], weighted=True)
    val = np.abs(val) + 0.1
    has = col[rowptr[0]:rowptr[1]].min() == 0 if rowptr[1] > rowptr[0] else False
    if with_loop0 and not has:          # give node 0 a self-loop: no identity is added, float32 mode
        col[rowptr[0]] = 0
    if not with_loop0 and has:
        val[rowptr[0]] = 0.0            # an explicit zero at (0,0) counts as "no self-loop" (mx[0,0] == 0)
    return rowptr, col, val


@pytest.mark.parametrize("with_loop0", [False, True])
def test_partitioned_normalize_adj_bitwise_equals_single_gpu(with_loop0, tmp_path):
    n, world = 2500, 2
    mp.spawn(_norm_worker, args=(world, _free_port(), n, with_loop0, str(tmp_path)), nprocs=world, join=True)
    rp, col, val = torch.load(tmp_path / "single.pt", weights_only=True)
    for r in range(world):
        prp, pc, pv, lo, hi = torch.load(tmp_path / f"n{r}.pt", weights_only=True)
        b, e = int(rp[lo]), int(rp[hi])
        assert torch.equal(prp, rp[lo:hi + 1] - rp[lo])
        assert torch.equal(pc, col[b:e]) and torch.equal(pv, val[b:e])


def _global_worker(rank, world, port, n, nnz, F, K, comm, mu, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd.dist import build_partitioned_graph
        dev = torch.device("cuda", 0)
        g = build_partitioned_graph(n, nnz, F, 0.7, 5, "normalize_adj", world, rank, dev, (F + 31) // 32 * 32,
                                    comm_size=comm, mu=mu, K=K)
        g.part.adj_int.schedule(feat=F)
        y = g.khop(K)
        torch.cuda.synchronize()
        torch.save({"y": y.cpu(), "x": g.x.cpu(), "dense": g.part.dense, "halo_nnz": g.part.halo_nnz,
                    "nnz_local": g.nnz_local, "nnz_global": g.nnz_global}, os.path.join(result_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("comm,mu,K", [(0, 1.0, 3), (256, 0.2, 2)])
def test_global_graph_partition_on_the_gpu(comm, mu, K, tmp_path):
    """bench.py's N > 1 path (dist.build_partitioned_graph: every rank generates and normalises its own rows of ONE
    global graph on the device, HIP kernels for the interior and halo blocks, K = 3 uses both ping-pong buffers):
    the assembled result meets the forward-error bar against the oracle on the global matrix."""
    from disco_amd import graphgen
    n, nnz, F, world = 6001, 150_000, 72, 2
    mp.spawn(_global_worker, args=(world, _free_port(), n, nnz, F, K, comm, mu, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f"g{r}.pt", weights_only=True) for r in range(world)]
    rows, cols = [], []
    for r in range(world):
        # same device as the ranks: the generator's random streams are per device type
        row, col, offs = graphgen.partition_edges(n, nnz, 0.7, 5, world, r, "cuda:0", comm, mu, add_diag=True)
        rows.append((row + offs[r]).cpu())
        cols.append(col.cpu())
    row, col = torch.cat(rows).numpy(), torch.cat(cols).numpy()
    assert len(row) == nnz + n == parts[0]["nnz_global"]
    deg = np.bincount(row, minlength=n).astype(np.float64)
    r_ = deg ** -0.5
    val = (r_[row] * r_[col]).astype(np.float32)
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(row, minlength=n), out=rowptr[1:])
    x = np.concatenate([p["x"].numpy() for p in parts])
    ref = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    got = np.concatenate([p["y"].numpy() for p in parts])
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert np.all(np.abs(got - ref) <= RTOL * K * bound + 1e-30)
    halo_frac = sum(p["halo_nnz"] for p in parts) / sum(p["nnz_local"] for p in parts)
    # the halo is an outcome: about half of the entries without locality (dense exchange chosen), about mu / 2 with it
    assert (halo_frac > 0.4 and all(p["dense"] for p in parts)) if comm == 0 else halo_frac < 0.2, halo_frac
    assert len({p["dense"] for p in parts}) == 1


def _feature_worker(rank, world, port, n, F, K, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd import SparseTensor
        from disco_amd.dist import build_feature_partitioned_graph
        dev = torch.device("cuda", 0)
        rowptr, col, val = random_csr(n, n, 9, seed=78, hubs=[(3, 2600)], weighted=True)
        x = torch.from_numpy(np.random.default_rng(6).standard_normal((n, F)).astype(np.float32)).to(dev)
        adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(dev), col=torch.from_numpy(col).to(dev),
                           value=torch.from_numpy(val).to(dev), sparse_sizes=(n, n), is_sorted=True)

        def slab(c0, c1, ld):
            buf = torch.zeros((n, ld), device=dev)
            buf[:, :c1 - c0] = x[:, c0:c1]
            return buf[:, :c1 - c0]

        g = build_feature_partitioned_graph(adj, slab, F, world, rank, K=K)
        y = g.khop(K)
        torch.cuda.synchronize()
        torch.save({"y": y.cpu(), "cols": g.cols}, os.path.join(result_dir, f"f{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("F,world,K", [(100, 2, 2), (13, 3, 3)])
def test_feature_partition_matches_global_oracle(F, world, K, tmp_path):
    """Feature (column) partition: the graph is replicated, every rank propagates its own column slab, nothing is
    exchanged; the slabs put side by side are the K-hop result (slabs of 52/48 and 4/4/5 columns: 4- and 16-lane groups)."""
    n = 3000
    mp.spawn(_feature_worker, args=(world, _free_port(), n, F, K, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f"f{r}.pt", weights_only=True) for r in range(world)]
    cols = [tuple(p["cols"]) for p in parts]
    assert cols[0][0] == 0 and cols[-1][1] == F and all(cols[i][1] == cols[i + 1][0] for i in range(world - 1))
    got = np.concatenate([p["y"].numpy() for p in parts], axis=1)
    rowptr, col, val = random_csr(n, n, 9, seed=78, hubs=[(3, 2600)], weighted=True)
    x = np.random.default_rng(6).standard_normal((n, F)).astype(np.float32)
    ref = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert np.all(np.abs(got - ref) <= RTOL * K * bound + 1e-30)


def _sweep_cases():
    """Seeded random row partitions: world 3, ragged / EMPTY row blocks, a rank without halo, narrow and wide features."""
    cases = []
    for seed in range(8):
        rng = np.random.default_rng(900 + seed)
        n = int(rng.choice([7, 400, 2500]))
        F = int(rng.choice([1, 5, 64, 130, 300]))
        K = int(rng.choice([1, 2, 3]))
        kind = seed % 4              # 0: balanced, 1: random cuts, 2: an empty block, 3: block-diagonal graph (no halo at all)
        cases.append(dict(seed=seed, n=n, F=F, K=K, kind=kind, avg=float(rng.choice([1, 6, 30]))))
    return cases


def _sweep_graph(c, world):
    from disco_amd.dist import balanced_row_offsets
    n = c["n"]
    rowptr, col, val = random_csr(n, n, c["avg"], seed=900 + c["seed"], hubs=[(min(3, n - 1), min(2600, 4 * n))] if n > 100 else [],
                                  weighted=True)
    rng = np.random.default_rng(c["seed"])
    if c["kind"] == 0:
        offs = balanced_row_offsets(torch.from_numpy(rowptr), world)
    elif c["kind"] == 1:
        offs = [0] + sorted(int(v) for v in rng.integers(0, n + 1, world - 1)) + [n]
    elif c["kind"] == 2:
        cut = int(rng.integers(1, n))
        offs = [0, cut, cut, n]
    else:
        offs = [0, n // 3, 2 * n // 3, n]
        r = np.repeat(np.arange(n), np.diff(rowptr))
        blk = np.searchsorted(np.array(offs[1:]), r, side="right")
        lo, hi = np.array(offs)[blk], np.array(offs)[blk + 1]
        col = (lo + (col - lo) % np.maximum(hi - lo, 1)).astype(col.dtype)        # fold every column into its row's own block
        order = np.lexsort((col, r))
        col, val = col[order], val[order]
    x = rng.standard_normal((n, c["F"])).astype(np.float32)
    return rowptr, col, val, offs, x


def _sweep_worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd.dist import RowPartition
        dev = torch.device("cuda", 0)
        for c in _sweep_cases():
            rowptr, col, val, offs, x = _sweep_graph(c, world)
            lo, hi = offs[rank], offs[rank + 1]
            F = c["F"]
            outs, first = {}, {}
            for exchange in ("need", "dense", "auto"):
                sl = slice(rowptr[lo], rowptr[hi])
                part = RowPartition(torch.from_numpy(rowptr[lo:hi + 1] - rowptr[lo]).to(dev), torch.from_numpy(col[sl]).to(dev),
                                    torch.from_numpy(val[sl]).to(dev), offs, rank, world, exchange=exchange)
                ld = (F + 3) // 4 * 4
                cur = torch.zeros((hi - lo, ld), device=dev)[:, :F]
                cur.copy_(torch.from_numpy(x[lo:hi]))
                for k in range(c["K"]):
                    out = torch.full((hi - lo, ld), 3.0, device=dev)[:, :F]
                    part.spmm(cur, out)
                    cur = out
                    if k == 0:
                        first[exchange] = out.cpu().numpy()
                torch.cuda.synchronize()
                outs[exchange] = cur.cpu().numpy()
                if c["kind"] == 3:
                    assert part.n_halo == 0 and part.halo_nnz == 0
            # Same bits in every exchange mode -- except on hub rows of tables wider than one column chunk: the dense mode
            # walks the halo block chunk by chunk, and the width of a chunk sets how many lane groups share a hub piece,
            # i.e. the association of that piece's sum (the oracle comparison below still holds them to the bar).
            deg = np.diff(rowptr)[lo:hi]
            same = (deg <= 256) | (F <= 128)
            for other in ("dense", "auto"):
                assert np.array_equal(first["need"][same], first[other][same]), (c, other)       # later hops consume the hub rows
                if F <= 128:
                    assert np.array_equal(outs["need"], outs[other]), (c, other)
            np.save(os.path.join(result_dir, f"s{c['seed']}_{rank}.npy"), outs["need"])
    finally:
        dist.destroy_process_group()


def test_random_row_partitions_three_ranks(tmp_path):
    """Every exchange mode on every case gives the same bits, and the assembled K-hop result meets the forward-error bar
    against the float64 oracle on the global matrix."""
    world = 3
    mp.spawn(_sweep_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for c in _sweep_cases():
        rowptr, col, val, offs, x = _sweep_graph(c, world)
        ref = orc.khop_csr(rowptr, col, val, x, c["K"], np.float64)
        bound = np.abs(x).astype(np.float64)
        for _ in range(c["K"]):
            bound = orc.spmm_abs_bound(rowptr, col, val, bound)
        got = np.concatenate([np.load(tmp_path / f"s{c['seed']}_{r}.npy") for r in range(world)])
        assert got.shape == ref.shape, c
        assert np.all(np.abs(got - ref) <= RTOL * c["K"] * bound + 1e-30), c
