"""Row partition + halo exchange on CPU with the gloo backend (world_size 2 and 3).  The SpMM of each
block is done by the ORACLE here (test infrastructure), so what is checked is the product's partition,
renumbering, request lists and all-to-all plumbing: the assembled result must equal the oracle's
single-process result on the global matrix."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import orc, random_csr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_spmm(adj, x, reduce, out):
    rp, col, val = (t.numpy() if t is not None else None for t in adj.csr())
    y = orc.spmm_csr(rp, col, val, x.numpy()[: adj.size(1)], "sum", np.float64).astype(np.float32)
    if reduce == "sum_acc":
        out += torch.from_numpy(y)
    else:
        out.copy_(torch.from_numpy(y))
    return out


def _worker(rank, world, port, n, F, K, result_dir, dense=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd.dist import RowPartition, balanced_row_offsets
        rowptr, col, val = random_csr(n, n, 6, seed=77, hubs=[(2, 400)], weighted=True)
        x = np.random.default_rng(5).standard_normal((n, F)).astype(np.float32)
        offs = balanced_row_offsets(torch.from_numpy(rowptr), world)
        lo, hi = offs[rank], offs[rank + 1]
        rp = torch.from_numpy(rowptr[lo:hi + 1] - rowptr[lo])
        sl = slice(rowptr[lo], rowptr[hi])
        part = RowPartition(rp, torch.from_numpy(col[sl]), torch.from_numpy(val[sl]), offs, rank, world, spmm_fn=_oracle_spmm,
                            exchange="dense" if dense else "need")
        assert part.adj_int.nnz() + part.adj_halo.nnz() == rowptr[hi] - rowptr[lo]
        assert part.dense == dense
        cur = torch.from_numpy(x[lo:hi].copy())
        for _ in range(K):
            out = torch.empty_like(cur)
            part.spmm(cur, out)
            cur = out
        np.save(os.path.join(result_dir, f"y{rank}.npy"), cur.numpy())
        np.save(os.path.join(result_dir, f"offs{rank}.npy"), np.array(offs))
        np.save(os.path.join(result_dir, f"halo{rank}.npy"), np.array([part.n_halo, int(part.send_idx.numel())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_partition_matches_global_oracle(world, tmp_path):
    n, F, K = 300, 9, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, F, K, str(tmp_path)), nprocs=world, join=True)
    rowptr, col, val = random_csr(n, n, 6, seed=77, hubs=[(2, 400)], weighted=True)
    x = np.random.default_rng(5).standard_normal((n, F)).astype(np.float32)
    ref = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    offs = np.load(tmp_path / "offs0.npy")
    assert offs[0] == 0 and offs[-1] == n and len(offs) == world + 1
    nnz_per = [rowptr[offs[r + 1]] - rowptr[offs[r]] for r in range(world)]
    assert max(nnz_per) - min(nnz_per) <= 450          # balanced by nnz up to one (hub) row
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert np.all(np.abs(got - ref) <= 1e-5 * K * bound + 1e-30)
    # need-only exchange: what is received is what somebody sent
    halo = np.array([np.load(tmp_path / f"halo{r}.npy") for r in range(world)])
    assert halo[:, 0].sum() == halo[:, 1].sum() and halo[:, 0].max() < n


def test_dense_halo_variant_matches_global_oracle(tmp_path):
    """exchange="dense": one all-gather of the (padded, here unequal) X shards; the halo block indexes the gathered table."""
    n, F, K, world = 300, 9, 2, 3
    mp.spawn(_worker, args=(world, _free_port(), n, F, K, str(tmp_path), True), nprocs=world, join=True)
    rowptr, col, val = random_csr(n, n, 6, seed=77, hubs=[(2, 400)], weighted=True)
    x = np.random.default_rng(5).standard_normal((n, F)).astype(np.float32)
    ref = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert np.all(np.abs(got - ref) <= 1e-5 * K * bound + 1e-30)


def _global_worker(rank, world, port, K, comm, mu, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd.dist import build_partitioned_graph
        g = build_partitioned_graph(1201, 14000, 7, 0.8, 99, "normalize_adj" if comm else "gcn_norm", world, rank, "cpu", 8,
                                    comm_size=comm, mu=mu, K=K, spmm_fn=_oracle_spmm)
        y = g.khop(K)
        p = g.part
        torch.save({"y": y.clone(), "x": g.x.clone(), "offs": p.col_offsets, "dense": p.dense, "desc": g.describe(),
                    "int": tuple(t.clone() if t is not None else None for t in p.adj_int.csr()),
                    "halo_nnz": p.halo_nnz, "nnz_global": g.nnz_global, "nnz_local": g.nnz_local},
                   os.path.join(result_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K,comm,mu", [(2, 3, 0, 1.0), (3, 2, 64, 0.2)])
def test_global_graph_partition_khop(world, K, comm, mu, tmp_path):
    """build_partitioned_graph: every rank generates its own rows of ONE global graph (no rank holds it), the
    K-hop result (K = 3 exercises the two ping-pong buffers) equals the oracle on the graph assembled from a
    single-process generation, and the halo fraction is an outcome: locality-free -> (world-1)/world of the
    entries and the dense exchange; community graph -> about mu * (world-1)/world and the need-only exchange."""
    from disco_amd import graphgen
    mp.spawn(_global_worker, args=(world, _free_port(), K, comm, mu, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f"g{r}.pt", weights_only=True) for r in range(world)]
    n, nnz = 1201, 14000
    norm_adj = comm > 0
    # the same global graph, generated block pair by block pair in one process
    rows, cols = [], []
    for r in range(world):
        row, col, offs = graphgen.partition_edges(n, nnz, 0.8, 99, world, r, "cpu", comm, mu, add_diag=norm_adj)
        rows.append(row + offs[r])
        cols.append(col)
    row, col = torch.cat(rows).numpy(), torch.cat(cols).numpy()
    assert len(row) == nnz + (n if norm_adj else 0) == parts[0]["nnz_global"]
    key = set((row * n + col).tolist())
    assert len(key) == len(row) and all((c * n + r) in key for r, c in zip(row[:4000], col[:4000]))     # symmetric
    deg = np.bincount(row, minlength=n).astype(np.float64)
    with np.errstate(divide="ignore"):
        if norm_adj:
            r_ = deg ** -0.5
            r_[np.isinf(r_)] = 0
            val = (r_[row] * r_[col]).astype(np.float32)
        else:
            d32 = (np.float32(1) / np.sqrt(deg.astype(np.float32)))
            d32[np.isinf(d32)] = 0
            val = (d32[row] * d32[col]).astype(np.float32)
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(row, minlength=n), out=rowptr[1:])
    x = np.concatenate([p["x"].numpy() for p in parts])
    ref = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    got = np.concatenate([p["y"].numpy() for p in parts])
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert np.all(np.abs(got - ref) <= 1e-5 * K * bound + 1e-30)
    halo_frac = sum(p["halo_nnz"] for p in parts) / sum(p["nnz_local"] for p in parts)
    if comm == 0:
        assert abs(halo_frac - (world - 1) / world) < 0.05 and all(p["dense"] for p in parts)
    else:
        assert halo_frac < 0.3 and not any(p["dense"] for p in parts)
    assert all(p["desc"]["halo_edge_fraction"] == round(p["halo_nnz"] / p["nnz_local"], 4) for p in parts)


def _c2r_worker(rank, world, port, n, F, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd.dist import columns_to_rows, feature_slabs
        Y = torch.arange(n * F, dtype=torch.float32).view(n, F)
        c0, c1 = feature_slabs(F, world)[rank]
        buf = torch.zeros(n, c1 - c0 + 3)                         # a strided slab, as the propagation leaves it
        buf[:, :c1 - c0] = Y[:, c0:c1]
        rows = columns_to_rows(buf[:, :c1 - c0], F)
        torch.save(rows, os.path.join(result_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,F", [(3, 50, 13), (2, 7, 602)])
def test_columns_to_rows_redistribution(world, n, F, tmp_path):
    """Feature partition -> row shards in one all-to-all (unequal slabs and row blocks)."""
    from disco_amd.graphgen import block_offsets
    mp.spawn(_c2r_worker, args=(world, _free_port(), n, F, str(tmp_path)), nprocs=world, join=True)
    Y = torch.arange(n * F, dtype=torch.float32).view(n, F)
    offs = block_offsets(n, world)
    for r in range(world):
        assert torch.equal(torch.load(tmp_path / f"r{r}.pt", weights_only=True), Y[offs[r]:offs[r + 1]])
