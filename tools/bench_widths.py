#!/usr/bin/env python3
"""Reddit-shaped SpMM at the widths the evaluation models propagate (41 = nclass, 128, 256 = hidden, 602 = input):
windowed (DISCO_SWEEP=1) vs plain (DISCO_SWEEP=0) vs the automatic choice."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disco_amd import SparseTensor, graphgen, ops, utils  # noqa: E402

dev = torch.device("cuda", 0)
spec = graphgen.SPECS["reddit"]
n = spec.n_nodes
row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev)
base = utils.normalize_adj_csr(SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True))
del row, col
out = {}
for F in (41, 128, 256, 602):
    ld = (F + 31) // 32 * 32
    x = torch.randn(n, ld, device=dev)[:, :F]
    y = torch.empty((n, ld), device=dev)[:, :F]
    for mode in ("auto", "1", "0"):
        if mode == "auto":
            os.environ.pop("DISCO_SWEEP", None)
        else:
            os.environ["DISCO_SWEEP"] = mode
        adj = base.set_value(base.storage.value())
        ts = []
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.spmm_raw(adj, x, "sum", out=y)
            e1.record()
            torch.cuda.synchronize()
            if i:
                ts.append(e0.elapsed_time(e1))
        out[f"F{F}_sweep_{mode}_ms"] = round(float(np.median(ts)), 3)
print(json.dumps(out))
