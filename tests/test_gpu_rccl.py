"""RCCL itself, on the one GPU of the test box: a single-rank `nccl` process group runs the collectives of disco_amd.dist
through RCCL's own code path (device tensors, async work handles, uneven split lists) -- the path the N > 1 bench takes and
that the gloo rehearsals (host-staged) never touch.  Results must equal the single-GPU entry points bit for bit.  If RCCL
cannot be initialised on the box at all the test is skipped (environment), never silently passed."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import random_csr
from test_dist_cpu import _free_port

pytestmark = pytest.mark.gpu


def _worker(rank, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        probe = torch.ones(4, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as e:          # RCCL unusable on this box: report, do not judge the code by it
        open(os.path.join(result_dir, "skip"), "w").write(repr(e))
        return
    try:
        from disco_amd import dist as dd
        from disco_amd.models.parametrized_adj_lp import PGE_Edge
        assert dist.get_backend() == "nccl"
        # all-to-all with split lists and an async handle, as RowPartition.start_exchange issues it
        inp = torch.arange(12 * 8, dtype=torch.float32, device=dev)
        out = torch.empty_like(inp)
        work = dd._all_to_all(out, inp, [inp.numel()], [inp.numel()], async_op=True)
        assert work is not None
        work.wait()
        assert torch.equal(out, inp)
        # a rank that neither sends nor receives anything (halo-free block): empty buffers, zero splits
        e_in, e_out = torch.empty(0, device=dev), torch.empty(0, device=dev)
        work = dd._all_to_all(e_out, e_in, [0], [0], async_op=True)
        if work is not None:
            work.wait()
        cnt_in, cnt_out = torch.tensor([5], dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
        dd._all_to_all(cnt_out, cnt_in)                      # the int64 count exchange of RowPartition.__init__
        assert int(cnt_out) == 5
        full = torch.empty(24, device=dev)
        work = dd._all_gather_into(full, torch.arange(24, dtype=torch.float32, device=dev), async_op=True)
        work.wait()
        assert torch.equal(full, torch.arange(24, dtype=torch.float32, device=dev))
        # the row partition of a whole graph on one rank, both exchange modes forced through the collectives
        n, F = 700, 130
        rowptr, col, val = random_csr(n, n, 8, seed=5, hubs=[(3, 900)], weighted=True)
        x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)).to(dev)
        from disco_amd import SparseTensor, ops
        adj = SparseTensor(rowptr=torch.from_numpy(rowptr).to(dev), col=torch.from_numpy(col).to(dev), value=torch.from_numpy(val).to(dev),
                           sparse_sizes=(n, n), is_sorted=True)
        ref = ops.spmm_raw(adj, x, "sum")
        for exchange in ("need", "dense", "auto"):
            part = dd.RowPartition(torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev),
                                   [0, n], 0, 1, exchange=exchange)
            out = torch.empty((n, 132), device=dev)[:, :F]
            part.spmm(x, out)
            assert torch.equal(out, ref), exchange
        # column shards -> row shards (one all-to-all) and the sharded scorer (one all-gather)
        y = dd.columns_to_rows(x, F)
        assert torch.equal(y, x)
        torch.manual_seed(0)
        pge = PGE_Edge(nfeat=12, nhid=64, device=dev).to(dev).eval()
        Z = torch.randn(301, 12, device=dev)
        for variant in ("transductive", "inductive", "sampled"):
            a, b = dd.sharded_edge_construction(pge, Z, 0.5, variant), pge.edge_construction(Z, 0.5, variant)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), variant
        rp, c, v = dd.normalize_adj_partition(torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev),
                                              torch.from_numpy(np.abs(val) + 0.1).to(dev), [0, n], 0, 1)
        from disco_amd import utils
        g = utils.normalize_adj_csr(adj.set_value(torch.from_numpy(np.abs(val) + 0.1).to(dev)))
        assert torch.equal(rp, g.csr()[0]) and torch.equal(c.long(), g.csr()[1]) and torch.equal(v, g.csr()[2])
        torch.cuda.synchronize()
        open(os.path.join(result_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_runs_the_collectives_of_the_partitioned_paths(tmp_path):
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    if (tmp_path / "skip").exists():
        pytest.skip("RCCL could not be initialised on this box: " + (tmp_path / "skip").read_text()[:300])
    assert (tmp_path / "ok").exists()
