#!/usr/bin/env python3
"""Experiment: order of the hub pieces.  disco_schedule_build emits them row by row (consecutive workgroups walk disjoint
column ranges of ONE row); here they are re-ordered by their relative position inside their row, so that workgroups running
together walk the same column region of DIFFERENT hub rows (which share popular sources).  Pure re-ordering of the piece
arrays in the schedule workspace: results must not change by a bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from disco_amd import SparseTensor, _lib, graphgen, ops, utils  # noqa: E402

dev = torch.device("cuda:0")
name, div = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("products", 1)
spec = graphgen.SPECS[name]
n, nnz, F = spec.n_nodes // div, spec.nnz // div, spec.feat
row, col = graphgen.powerlaw_edges(n, nnz, spec.skew, spec.seed, device=dev)
adj = SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True)
del row, col
adj = utils.normalize_adj_csr(adj)
x = graphgen.features(n, F, 1, device=dev, ld=(F + 31) // 32 * 32)
y = torch.empty((n, x.stride(0)), device=dev)[:, :F]
ws, info = adj.schedule(feat=F, sweep=False)


def timed(reps=5):
    ops.spmm_raw(adj, x, "sum", out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.spmm_raw(adj, x, "sum", out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t0 = timed()
ref = y.clone()
P = int(info.n_hub_pieces)
prow = ws[info.off_piece_row: info.off_piece_row + 4 * P].view(torch.int32)
pbeg = ws[info.off_piece_begin: info.off_piece_begin + 8 * P].view(torch.int64)
pend = ws[info.off_piece_end: info.off_piece_end + 8 * P].view(torch.int64)
pslot = ws[info.off_piece_slot: info.off_piece_slot + 4 * P].view(torch.int32)
_, inv, cnt = torch.unique_consecutive(prow, return_inverse=True, return_counts=True)
first = torch.cumsum(cnt, 0) - cnt
k = torch.arange(P, device=dev) - first[inv]
for mode in ("fraction", "fraction, long rows first"):
    frac = (k.double() + 0.5) / cnt[inv].double()
    key = frac if mode == "fraction" else frac - 1e-9 * cnt[inv].double()
    order = torch.argsort(key, stable=True)
    for a in (prow, pbeg, pend, pslot):
        a.copy_(a[order].clone())
    t1 = timed()
    print(f"{name}/{div} n={n} nnz={adj.nnz()} hub rows={int(info.n_hub_rows)} pieces={P}: row-major {t0:.3f} ms/hop -> {mode} {t1:.3f} ms/hop, same bits: {torch.equal(y, ref)}")
    # restore row-major order for the next mode
    back = torch.argsort(order)
    for a in (prow, pbeg, pend, pslot):
        a.copy_(a[back].clone())
