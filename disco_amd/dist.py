"""1-D row partition of the propagation across the GPUs of one node.

The reference is single-GPU (``--parallel_gpu_ids`` is parsed and never used,
/root/reference/LargeScaleCondensing.py:31); this module is new design for MI355X:

  * rank g owns a contiguous block of rows of CSR(Â^T) and the matching rows of X and Y;
  * its entries are split into an INTERIOR block (columns it owns, local ids) and a HALO block (remote
    columns, renumbered into a receive buffer that holds only the rows this rank really needs);
  * per hop: owned rows requested by peers are gathered into a send buffer and exchanged with ONE
    all-to-all(v) (RCCL over xGMI: every peer pair uses its own link) issued asynchronously, the
    interior SpMM runs meanwhile on the compute stream, then the halo SpMM accumulates onto the
    interior result (``y += A_halo @ recv``): interior first, halo second -- a fixed order, no atomics.

All index bookkeeping is stock torch; the SpMMs go through the C ABI.  ``spmm_fn`` is injectable so
the CPU (gloo) tests can check the partition / exchange logic against the oracle without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass, field  # noqa: F401
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from .sparse import SparseTensor


def balanced_row_offsets(rowptr: torch.Tensor, world: int) -> List[int]:
    """Row offsets that balance NNZ (not rows) across ranks."""
    n = rowptr.numel() - 1
    nnz = int(rowptr[-1])
    offs = [0]
    for g in range(1, world):
        target = nnz * g // world
        offs.append(int(torch.searchsorted(rowptr, torch.tensor([target], device=rowptr.device, dtype=rowptr.dtype)).item()))
    offs.append(n)
    for i in range(1, len(offs)):
        offs[i] = max(offs[i], offs[i - 1])
    return offs


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None, group=None, async_op=False):
    """``all_to_all_single``; device tensors under the gloo backend (rehearsals on a box without RCCL
    peers, tests/test_gpu_dist.py) are staged through host memory, synchronously."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(o, i, out_splits, in_splits, group=group)
        out.copy_(o)
        return None
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
        return None
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)


class RowPartition:
    """This rank's share of a row-partitioned square adjacency.

    rowptr/col/val: CSR of the LOCAL rows with GLOBAL column ids; ``col_offsets`` (world+1) gives the
    column (= row) ownership ranges.  ``exchange``: "need" = all-to-all(v) of exactly the rows each peer
    needs; "dense" = one all-gather of the X shards (blocks padded to the largest), the halo block then indexes
    the gathered table; "auto" picks "dense" when this rank needs more than half of all remote rows anyway
    (a graph without locality: the request lists, the index_select and the uneven all-to-all buy nothing).
    The choice is made collectively (every rank must run the same collective).
    """

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], col_offsets: List[int],
                 rank: int, world: int, group=None, spmm_fn: Optional[Callable] = None, exchange: str = "auto"):
        self.rank, self.world, self.group = rank, world, group
        dev = col.device
        self.device = dev
        self.col_offsets = list(col_offsets)
        lo, hi = col_offsets[rank], col_offsets[rank + 1]
        self.n_local = rowptr.numel() - 1
        self.n_global = col_offsets[-1]
        assert self.n_local == hi - lo, "row block must match the owned column range"
        self.nnz_local = int(col.numel())
        col = col.to(torch.int64)
        row = torch.repeat_interleave(torch.arange(self.n_local, device=dev), rowptr[1:] - rowptr[:-1])
        interior = (col >= lo) & (col < hi)
        # interior block: local column ids
        self.adj_int = SparseTensor(row=row[interior], col=col[interior] - lo,
                                    value=None if val is None else val[interior],
                                    sparse_sizes=(self.n_local, self.n_local), is_sorted=True)
        hrow, hcol = row[~interior], col[~interior]
        hval = None if val is None else val[~interior]
        del row, interior
        need = torch.unique(hcol)                                 # sorted global ids this rank needs
        self.n_halo = int(need.numel())
        self.halo_nnz = int(hcol.numel())
        if exchange == "auto":
            vote = torch.tensor([1 if self.n_halo * 2 > self.n_global - self.n_local else 0], dtype=torch.int64,
                                device=dev if world > 1 and dist.get_backend(group) != "gloo" else "cpu")
            if world > 1:
                dist.all_reduce(vote, group=group)
            exchange = "dense" if int(vote) * 2 > world else "need"
        self.dense = world > 1 and exchange == "dense"
        self._spmm = spmm_fn
        self._send_buf = self._recv_buf = self._full_buf = self._mine_buf = None
        self._dense_key = None
        # (before, after) event pairs around every wait for the exchange, recorded on the compute stream when
        # `time_waits` is set: their elapsed time is the part of the transfer that compute did NOT cover
        self.time_waits = False
        self.wait_events: List[tuple] = []
        bounds = torch.tensor(col_offsets, device=dev, dtype=torch.int64)
        if self.dense:
            # halo block over the all-gathered table: shard g sits at rows [g * per, g * per + size_g)
            self.per = max(col_offsets[g + 1] - col_offsets[g] for g in range(world))
            owner = torch.bucketize(hcol, bounds[1:], right=True)
            self.adj_halo = SparseTensor(row=hrow, col=owner * self.per + (hcol - bounds[owner]), value=hval,
                                         sparse_sizes=(self.n_local, world * self.per), is_sorted=True)
            self.adj_halo._active_col_frac = (world - 1) / world      # the rank's own shard of the table is never gathered
            self.send_idx = torch.empty(0, dtype=torch.int64, device=dev)
            self.send_splits = self.recv_splits = [0] * world
            return
        # need-only: columns renumbered into the receive buffer (sorted by global id => grouped by owner)
        inv = torch.searchsorted(need, hcol)
        self.adj_halo = SparseTensor(row=hrow, col=inv, value=hval, sparse_sizes=(self.n_local, max(self.n_halo, 1)),
                                     is_sorted=True)
        owner_start = torch.searchsorted(need, bounds)            # position of each owner's first needed row
        self.recv_splits = (owner_start[1:] - owner_start[:-1]).tolist()
        assert self.recv_splits[rank] == 0
        # tell every owner which of its rows we need
        send_counts = torch.zeros(world, dtype=torch.int64, device=dev)
        recv_counts = torch.tensor(self.recv_splits, dtype=torch.int64, device=dev)
        if world > 1:
            _all_to_all(send_counts, recv_counts, group=group)
        self.send_splits = send_counts.tolist()
        want_local = need - bounds[torch.bucketize(need, bounds[1:], right=True)]  # ids local to their owner
        self.send_idx = torch.empty(sum(self.send_splits), dtype=torch.int64, device=dev)
        if world > 1:
            _all_to_all(self.send_idx, want_local.contiguous(), self.send_splits, self.recv_splits, group=group)
        if self.send_idx.numel():
            assert int(self.send_idx.max()) < self.n_local and int(self.send_idx.min()) >= 0

    # ---- one propagation -------------------------------------------------------------------------
    def _spmm_call(self, adj, x, reduce, out):
        if self._spmm is not None:
            return self._spmm(adj, x, reduce, out)
        from . import ops
        return ops.spmm_raw(adj, x, reduce, out=out)

    @staticmethod
    def _xld(F: int) -> int:
        return (F + 3) // 4 * 4      # exchange buffers: 16-byte rows (128-bit gathers), no wider padding on the wire

    def _buffers(self, x: torch.Tensor):
        F = x.shape[1]
        ld = self._xld(F)
        if self._send_buf is None or self._send_buf.shape[1] != ld:
            self._send_buf = torch.empty((max(self.send_idx.numel(), 1), ld), dtype=x.dtype, device=x.device)
            self._recv_buf = torch.empty((max(self.n_halo, 1), ld), dtype=x.dtype, device=x.device)
        return self._send_buf[:, :F], self._recv_buf[:, :F]

    def start_exchange(self, x: torch.Tensor):
        """Gather the rows peers asked for and start the all-to-all; returns (work handle, recv view)."""
        send, recv = self._buffers(x)
        if self.world == 1:
            return None, recv
        ns = self.send_idx.numel()
        if ns:
            if x.is_cuda:
                from . import _lib
                with torch.cuda.device(x.device):
                    _lib.check(_lib.load().disco_gather_rows_f32(_lib.ptr(x), x.stride(0), x.shape[1], _lib.ptr(self.send_idx), ns,
                                                                 _lib.ptr(send), send.stride(0), _lib.current_stream(x.device)),
                               "disco_gather_rows_f32")
            else:                       # CPU: only the gloo tests of the bookkeeping come here
                torch.index_select(x, 0, self.send_idx, out=send[:ns])
        ld = self._send_buf.shape[1]
        sflat = self._send_buf[:ns].reshape(-1)
        rflat = self._recv_buf[:self.n_halo].reshape(-1)
        work = _all_to_all(rflat, sflat, [c * ld for c in self.recv_splits], [c * ld for c in self.send_splits],
                           group=self.group, async_op=True)
        return work, recv

    @staticmethod
    def _col_chunks(F: int, width: int = 128) -> List[tuple]:
        """Balanced column chunks of at most ~``width`` floats, multiples of 32 (the kernels' own column blocks)."""
        n = max(1, (F + width - 1) // width)
        per = ((F + n - 1) // n + 31) // 32 * 32
        return [(c0, min(c0 + per, F)) for c0 in range(0, F, per)]

    def _spmm_dense(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """All-gather of the X shards, pipelined by column chunk: chunk c + 1 is on the wire while the halo SpMM of
        chunk c runs, so only the first chunk's transfer (and what the SpMMs cannot cover) is exposed.  Every output
        column still gets interior first, halo second: below the hub threshold the bits do not depend on the chunking
        (a hub row's pieces are summed by 512 / G lane groups, and G follows the chunk's width: last-bit differences
        there once F exceeds one chunk)."""
        F, n = x.shape[1], self.n_local
        chunks = self._col_chunks(F)
        if self._dense_key != (F, len(chunks), x.dtype, x.device):
            self._dense_key = (F, len(chunks), x.dtype, x.device)
            self._full_buf = [torch.empty((self.world * self.per, self._xld(c1 - c0)), dtype=x.dtype, device=x.device) for c0, c1 in chunks]
            self._mine_buf = [torch.zeros((self.per, self._xld(c1 - c0)), dtype=x.dtype, device=x.device) for c0, c1 in chunks]
        works = []
        for (c0, c1), full, mine in zip(chunks, self._full_buf, self._mine_buf):
            mine[:n, :c1 - c0].copy_(x[:, c0:c1])
            works.append(_all_gather_into(full.reshape(-1), mine.reshape(-1), self.group, async_op=True))
        self._spmm_call(self.adj_int, x, "sum", out)              # overlaps with the transfers
        for (c0, c1), full, work in zip(chunks, self._full_buf, works):
            self._wait(work, x)
            if self.adj_halo.nnz():
                self._spmm_call(self.adj_halo, full[:, :c1 - c0], "sum_acc", out[:, c0:c1])
        return out

    def _wait(self, work, x: torch.Tensor) -> None:
        """Make the compute stream wait for an exchange; with `time_waits` the wait is bracketed by two events on the
        compute stream (exposed exchange time = what the interior SpMM did not cover)."""
        if work is None:
            return
        if self.time_waits and x.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work.wait()
            e1.record()
            self.wait_events.append((e0, e1))
        else:
            work.wait()

    def exposed_wait_ms(self, clear: bool = True) -> List[float]:
        """Elapsed ms of every recorded wait (call after a synchronize)."""
        out = [a.elapsed_time(b) for a, b in self.wait_events]
        if clear:
            self.wait_events = []
        return out

    def spmm(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """out = A_local @ X_global, X given by its local rows; halo rows are fetched from the peers."""
        if self.dense:
            return self._spmm_dense(x, out)
        work, recv = self.start_exchange(x)
        self._spmm_call(self.adj_int, x, "sum", out)              # overlaps with the exchange
        self._wait(work, x)
        if self.n_halo:
            self._spmm_call(self.adj_halo, recv, "sum_acc", out)
        return out

    def exchange_bytes_per_hop(self, F: int) -> int:
        """Bytes this rank RECEIVES per hop."""
        ld = self._xld(F)
        if self.dense:
            return (self.world - 1) * self.per * sum(self._xld(c1 - c0) for c0, c1 in self._col_chunks(F)) * 4
        return self.n_halo * ld * 4


@dataclass
class PartitionedGraph:
    """One rank's share of a BASELINE-shaped global graph, with its feature / output / ping-pong buffers."""
    part: RowPartition
    x: torch.Tensor
    y: torch.Tensor
    tmp: List[torch.Tensor]
    F: int
    nnz_local: int
    nnz_global: int
    n_global: int

    def khop(self, K: int, events=None):
        src = self.x
        for k in range(K):
            dst = self.y if k == K - 1 else self.tmp[k & 1]
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.part.spmm(src, dst)
            if events is not None:
                e1.record()
                events.append((e0, e1, src.shape[1]))
            src = dst
        return self.y

    def describe(self):
        p = self.part
        remote = self.n_global - p.n_local
        return {"exchange": ("all_gather_into_tensor of the X shards" if p.dense else "all_to_all_single of the needed rows") +
                            " (RCCL), async, overlapped with the interior SpMM",
                "halo_rows_needed": p.n_halo, "remote_rows": remote,
                "halo_row_fraction": round(p.n_halo / max(remote, 1), 4),
                "interior_nnz": p.adj_int.nnz(), "halo_nnz": p.halo_nnz,
                "halo_edge_fraction": round(p.halo_nnz / max(p.nnz_local, 1), 4),
                "recv_bytes_per_hop": p.exchange_bytes_per_hop(self.F)}


def build_partitioned_graph(n_global: int, nnz_global: int, F: int, skew: float, seed: int, norm: str, world: int, rank: int,
                            device, ld: int, comm_size: int = 0, mu: float = 1.0, exchange: str = "auto", K: int = 2,
                            group=None, spmm_fn: Optional[Callable] = None, max_degree: int = 0) -> PartitionedGraph:
    """Rank ``rank``'s row block of a GLOBAL symmetric graph (``graphgen.partition_edges``: every rank generates
    exactly the entries of its own rows, drawn over the global id range), normalised the way the reference
    normalises that dataset, wrapped in a ``RowPartition``.  How many halo rows / entries result is an OUTCOME
    of the graph and the partition (``describe()``), not an input.
      norm = "normalize_adj": +I, r = rowsum^-1/2 in float64, one rounding  (/root/reference/utils.py:406-435)
      norm = "gcn_norm"     : deg^-1/2 in float32, (dis[row] * 1) * dis[col]  (utils.py:242-247; the symmetric graph's
                              column degree equals its row degree)
    The degree vector is local under a row partition; r / dis of the remote columns needs ONE all-gather of an
    N-vector at set-up."""
    from . import graphgen
    row, col, offs = graphgen.partition_edges(n_global, nnz_global, skew, seed, world, rank, device, comm_size, mu,
                                              add_diag=(norm == "normalize_adj"), max_degree=max_degree)
    lo, hi = offs[rank], offs[rank + 1]
    n_local = hi - lo
    counts = torch.bincount(row, minlength=n_local)
    rowptr = torch.zeros(n_local + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    wdt = torch.float64 if norm == "normalize_adj" else torch.float32
    r_local = counts.to(wdt).pow(-0.5)
    r_local[torch.isinf(r_local)] = 0
    if world > 1:
        per = max(offs[g + 1] - offs[g] for g in range(world))
        mine = torch.zeros(per, dtype=wdt, device=device)
        mine[:n_local] = r_local
        padded = torch.empty(world * per, dtype=wdt, device=device)
        _all_gather_into(padded, mine, group)
        r_all = torch.cat([padded[g * per: g * per + offs[g + 1] - offs[g]] for g in range(world)])
    else:
        r_all = r_local
    val = (r_local[row] * r_all[col]).to(torch.float32)
    del r_all
    nnz_l = int(col.numel())
    part = RowPartition(rowptr, col, val, offs, rank, world, group=group, exchange=exchange, spmm_fn=spmm_fn)
    del row, col, val
    x = graphgen.features(n_local, F, seed + 31 * rank, device=device, ld=ld)
    y = torch.empty((n_local, ld), dtype=torch.float32, device=device)[:, :F]
    tmp = [torch.empty((n_local, ld), dtype=torch.float32, device=device)[:, :F] for _ in range(min(max(K - 1, 0), 2))]
    t = torch.tensor([nnz_l], dtype=torch.int64, device=device)
    if world > 1:
        if dist.get_backend(group) == "gloo":
            t = t.cpu()
        dist.all_reduce(t, group=group)
    return PartitionedGraph(part, x, y, tmp, F, nnz_l, int(t.item()), n_global)


def feature_slabs(F: int, world: int) -> List[tuple]:
    """Column ranges [c0, c1) of the ``world`` feature slabs: contiguous, in units of 4 floats (128-bit gathers), as
    equal as that allows; a rank may get an empty slab when F < 4 * world."""
    nv = (F + 3) // 4
    cuts = [min(F, 4 * (nv * g // world)) for g in range(world)] + [F]
    return [(cuts[g], cuts[g + 1]) for g in range(world)]


@dataclass
class FeaturePartitionedGraph:
    """Feature (column) partition of the K-hop propagation: EVERY rank holds the whole normalised graph -- a CSR of a
    few GB against 288 GB of HBM -- and one column slab of X.  A^K X is independent per feature column, so the K hops
    need NO exchange at all: xGMI moves tens of GB/s per link where HBM moves TB/s, and a row partition of a graph
    without locality ships nearly the whole feature table every hop (papers100M-shaped at 8 ranks: 50 GB received
    per rank and hop, ~10x the time of the SpMM itself).  What it costs: every rank streams the whole CSR (8 B per
    entry and hop), and slabs narrower than 32 floats gather less than a 128-byte line per request.  No counterpart
    in the reference (single GPU)."""
    adj: SparseTensor
    x: torch.Tensor
    y: torch.Tensor
    tmp: List[torch.Tensor]
    cols: tuple
    F: int
    nnz_global: int
    n_global: int

    def khop(self, K: int, events=None):
        from . import ops
        src = self.x
        for k in range(K):
            dst = self.y if k == K - 1 else self.tmp[k & 1]
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if src.shape[1]:
                ops.spmm_raw(self.adj, src, "sum", out=dst)
            if events is not None:
                e1.record()
                events.append((e0, e1, src.shape[1]))
            src = dst
        return self.y

    def describe(self):
        return {"exchange": "none: the graph is replicated, every rank propagates its own feature columns",
                "columns": list(self.cols), "csr_bytes_replicated": self.adj.nnz() * 8 + (self.n_global + 1) * 8}


def build_feature_partitioned_graph(adj: SparseTensor, x_slab_fn: Callable, F: int, world: int, rank: int, K: int = 2,
                                    ld_pad: int = 16) -> FeaturePartitionedGraph:
    """Wrap the (replicated, already normalised) adjacency with this rank's column slab.  ``x_slab_fn(c0, c1, ld)`` returns
    the [N, c1 - c0] slab of X as a column view of a buffer with leading dimension ``ld`` (from a loader, or synthetic)."""
    c0, c1 = feature_slabs(F, world)[rank]
    w = c1 - c0
    ld = max(ld_pad, (w + ld_pad - 1) // ld_pad * ld_pad)
    n = adj.size(0)
    dev = adj.device
    x = x_slab_fn(c0, c1, ld)
    assert x.shape == (n, w)
    y = torch.empty((n, ld), dtype=torch.float32, device=dev)[:, :w]
    tmp = [torch.empty((n, ld), dtype=torch.float32, device=dev)[:, :w] for _ in range(min(max(K - 1, 0), 2))]
    return FeaturePartitionedGraph(adj, x, y, tmp, (c0, c1), F, adj.nnz(), n)


def columns_to_rows(y_slab: torch.Tensor, F: int, group=None) -> torch.Tensor:
    """Turn the column-sharded result of the feature partition ([N, w_g] on rank g) into a row-sharded one
    ([N_g, F] on rank g, rows in ``graphgen.block_offsets(N, world)`` blocks) with ONE all-to-all: what a consumer
    that needs whole rows (SGConv's ``lin``, a classifier) wants -- once per K-hop call, not once per hop."""
    from .graphgen import block_offsets
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = y_slab.shape[0]
    slabs = feature_slabs(F, world)
    offs = block_offsets(n, world)
    w = slabs[rank][1] - slabs[rank][0]
    assert y_slab.shape[1] == w
    n_mine = offs[rank + 1] - offs[rank]
    send = y_slab.contiguous().reshape(-1)                        # row blocks are contiguous in a row-major slab
    recv = torch.empty(n_mine * F, dtype=y_slab.dtype, device=y_slab.device)
    in_splits = [(offs[r + 1] - offs[r]) * w for r in range(world)]
    out_splits = [n_mine * (slabs[g][1] - slabs[g][0]) for g in range(world)]
    _all_to_all(recv, send, out_splits, in_splits, group=group)
    out = torch.empty((n_mine, F), dtype=y_slab.dtype, device=y_slab.device)
    pos = 0
    for g in range(world):
        c0, c1 = slabs[g]
        out[:, c0:c1] = recv[pos: pos + n_mine * (c1 - c0)].view(n_mine, c1 - c0)
        pos += n_mine * (c1 - c0)
    return out


def sharded_edge_construction(pge_edge, Z: torch.Tensor, threshold: float, variant: str = "transductive", group=None):
    """``PGE_Edge.edge_construction`` over several GPUs (scorer row): the n x n pair grid is cut
    into row blocks, rank g scores rows [g*per, (g+1)*per) -- the n^2 x H x H GEMM, the only expensive part --
    and ONE all-gather of the blocks (n^2 float32 in total, 2.5 GB at n = 25 000: a few ms over xGMI against
    0.66 s / world of scoring) gives every rank the full matrix; the cheap symmetrise / threshold / nonzero
    tail then runs redundantly so that every rank ends up with the identical edge list, in
    ``torch.nonzero`` order.  Z (and the scorer) are replicated."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = Z.shape[0]
    per = (n + world - 1) // world
    r0, r1 = min(rank * per, n), min(rank * per + per, n)
    prepared = pge_edge._prepare(Z)
    block = torch.zeros((per, n), dtype=torch.float32, device=Z.device)
    if r1 > r0:
        pge_edge.score_block(Z, (r0, r1), out=block[:r1 - r0], _prepared=prepared)
    S = torch.empty((world * per, n), dtype=torch.float32, device=Z.device)
    _all_gather_into(S, block, group)
    return pge_edge.edges_from_scores(S[:n], threshold, variant)


def normalize_adj_partition(rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], col_offsets: List[int],
                            rank: int, world: int, group=None):
    """``normalize_adj`` (/root/reference/utils.py:406-435, the A3 semantics of ``utils.normalize_adj_csr``)
    on a row partition (gcn_norm row): the row sums are local, the ``mx[0,0] == 0`` self-loop
    rule is decided by the owner of row 0 and broadcast, and the only exchange is ONE all-gather of the
    N-vector of row sums (``r[col]`` is needed for remote columns).  Input: CSR of this rank's rows with
    GLOBAL column ids (duplicates already summed).  Returns (rowptr, col int64, val float32) of the
    normalised block -- feed it to ``RowPartition``.  Same kernels, same bits as the single-GPU path."""
    from . import _lib
    from .sparse import coo_to_csr
    dev = col.device
    assert dev.type == "cuda", "normalize_adj_partition runs on the GPU"
    lo, hi = col_offsets[rank], col_offsets[rank + 1]
    n_local, n_global = hi - lo, col_offsets[-1]
    assert rowptr.numel() == n_local + 1
    col = col.to(torch.int64)
    val = torch.ones(col.numel(), dtype=torch.float32, device=dev) if val is None else val.to(torch.float32)
    row = torch.repeat_interleave(torch.arange(n_local, device=dev), rowptr[1:] - rowptr[:-1])
    flag = torch.zeros(1, dtype=torch.int64)
    if lo == 0 and n_local > 0:
        flag[0] = int(bool(((row == 0) & (col == 0) & (val != 0)).any()))
    if world > 1:
        owner = next(g for g in range(world) if col_offsets[g + 1] > 0)
        if dist.get_backend(group) != "gloo":
            flag = flag.to(dev)                  # RCCL moves device tensors only
        dist.broadcast(flag, src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
    add_identity = 0 if int(flag[0]) else 1
    if add_identity:
        d = torch.arange(n_local, device=dev)
        rowptr, c32, val = coo_to_csr(torch.cat([row, d]), torch.cat([col, d + lo]),
                                      torch.cat([val, torch.zeros(n_local, dtype=torch.float32, device=dev)]),
                                      (n_local, n_global), dup="sum")
    else:
        c32 = col.to(torch.int32)
    lib = _lib.load()
    per = max(col_offsets[g + 1] - col_offsets[g] for g in range(world))
    with torch.cuda.device(dev):
        s = _lib.current_stream(dev)
        rs = torch.zeros(per, dtype=torch.float64, device=dev)
        _lib.check(lib.disco_csr_rowsum_f64(_lib.ptr(rowptr), _lib.ptr(val), n_local, add_identity, _lib.ptr(rs), s),
                   "disco_csr_rowsum_f64")
        if world > 1:
            padded = torch.empty(world * per, dtype=torch.float64, device=dev)
            _all_gather_into(padded, rs, group)
            r_all = torch.cat([padded[g * per: g * per + col_offsets[g + 1] - col_offsets[g]] for g in range(world)])
        else:
            r_all = rs[:n_local].clone()
        # the scaling kernel indexes r[] by global row and column: run it on the global row range with every
        # row outside this rank's block empty
        nnz = int(c32.numel())
        rp_g = torch.empty(n_global + 1, dtype=torch.int64, device=dev)
        rp_g[:lo] = 0
        rp_g[lo:hi + 1] = rowptr
        rp_g[hi + 1:] = nnz
        out = torch.empty(nnz, dtype=torch.float32, device=dev)
        _lib.check(lib.disco_gcn_norm_csr_f32(_lib.ptr(rp_g), _lib.ptr(c32), _lib.ptr(val), n_global, add_identity,
                                              _lib.ptr(r_all), _lib.ptr(out), s), "disco_gcn_norm_csr_f32")
    return rowptr, c32.to(torch.int64), out
