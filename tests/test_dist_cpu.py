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
    if dense:
        os.environ["DISCO_DENSE_HALO"] = "1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disco_amd.dist import RowPartition, balanced_row_offsets
        rowptr, col, val = random_csr(n, n, 6, seed=77, hubs=[(2, 400)], weighted=True)
        x = np.random.default_rng(5).standard_normal((n, F)).astype(np.float32)
        offs = balanced_row_offsets(torch.from_numpy(rowptr), world)
        if dense:
            offs = [g * (n // world) for g in range(world + 1)]       # the dense variant needs equal row blocks
        lo, hi = offs[rank], offs[rank + 1]
        rp = torch.from_numpy(rowptr[lo:hi + 1] - rowptr[lo])
        sl = slice(rowptr[lo], rowptr[hi])
        part = RowPartition(rp, torch.from_numpy(col[sl]), torch.from_numpy(val[sl]), offs, rank, world, spmm_fn=_oracle_spmm)
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
    """DISCO_DENSE_HALO=1: the exchange is one all-gather of X and the halo block indexes it by global column id."""
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
