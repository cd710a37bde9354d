#!/usr/bin/env python3
"""SpMM at the narrow widths GCNConv's last layer propagates (F = nclass: 41 Reddit, 47 products, 40 arxiv),
input straight from nn.Linear (row pitch = F floats, not 16-byte aligned)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disco_amd import SparseTensor, graphgen, ops, utils  # noqa: E402

dev = torch.device("cuda", 0)
out = {}
for wl, F in (("reddit", 41), ("products", 47)):
    spec = graphgen.SPECS[wl]
    n = spec.n_nodes
    row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev)
    adj = utils.normalize_adj_csr(SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True))
    del row, col
    x = torch.randn(n, F, device=dev)
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = ops.spmm_raw(adj, x, "sum")
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    out[f"{wl}_F{F}_ms"] = float(np.median(ts))
    out[f"{wl}_F{F}_Gedgefeat_s"] = adj.nnz() * F / np.median(ts) / 1e6
    del adj, x, y
print(json.dumps(out))
