#!/usr/bin/env python3
"""Calibration of SparseTensor.sweep_pays: windowed (DISCO_SWEEP=1) vs plain (0) on Reddit-sized graphs of falling
density (mean degree 25..200, 21 windows): the crossover lies near mean degree = 6 x windows; the rule uses 8."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from disco_amd import SparseTensor, graphgen, ops, utils
dev = torch.device("cuda", 0)
n = 232965
out = {}
for deg in (25, 50, 100, 200):
    row, col = graphgen.powerlaw_edges(n, n * deg, 0.4, 123 + deg, device=dev)
    base = utils.normalize_adj_csr(SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True))
    del row, col
    for F in (128, 602):
        ld = (F + 31) // 32 * 32
        x = torch.randn(n, ld, device=dev)[:, :F]
        y = torch.empty((n, ld), device=dev)[:, :F]
        for mode in ("1", "0"):
            os.environ["DISCO_SWEEP"] = mode
            adj = base.set_value(base.storage.value())
            ts = []
            for i in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.spmm_raw(adj, x, "sum", out=y); e1.record(); torch.cuda.synchronize()
                if i: ts.append(e0.elapsed_time(e1))
            out[f"deg{deg}_F{F}_sweep{mode}"] = round(float(np.median(ts)), 3)
print(json.dumps(out))
