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


def _worker(rank, world, port, n, F, K, result_dir):
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
        part = RowPartition(rp, torch.from_numpy(col[sl]).to(dev), torch.from_numpy(val[sl]).to(dev), offs, rank, world)
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


@pytest.mark.parametrize("F", [40, 602])
def test_two_ranks_one_gpu_matches_global_oracle(F, tmp_path):
    n, K, world = 3000, 2, 2
    mp.spawn(_worker, args=(world, _free_port(), n, F, K, str(tmp_path)), nprocs=world, join=True)
    rowptr, col, val = random_csr(n, n, 9, seed=78, hubs=[(3, 2600), (n - 2, 900)], weighted=True)
    x = np.random.default_rng(6).standard_normal((n, F)).astype(np.float32)
    ref = orc.khop_csr(rowptr, col, val, x, K, np.float64)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    bound = np.abs(x).astype(np.float64)
    for _ in range(K):
        bound = orc.spmm_abs_bound(rowptr, col, val, bound)
    assert np.all(np.abs(got - ref) <= RTOL * K * bound + 1e-30)
    halo = np.array([np.load(tmp_path / f"halo{r}.npy") for r in range(world)])
    assert halo[:, 0].sum() == halo[:, 1].sum() > 0


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
