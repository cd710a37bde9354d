#!/usr/bin/env python3
"""A7 at Reddit shape: T = max-aggregate(X) over the raw adjacency (LargeScaleCondensing.py:152-172),
windowed (default) vs plain kernel, HIP-event timing."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disco_amd import SparseTensor, graphgen, ops  # noqa: E402

dev = torch.device("cuda", 0)
spec = graphgen.SPECS["reddit"]
n, F = spec.n_nodes, spec.feat
row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev)
x = graphgen.features(n, F, spec.seed, device=dev, ld=608)
out = {}
res = {}
for name, env in (("windowed", "1"), ("plain", "0")):
    os.environ["DISCO_SWEEP"] = env
    adj = SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True)
    y = torch.empty((n, 608), device=dev)[:, :F]
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.spmm_raw(adj, x, "max", out=y)
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    out[name + "_ms"] = float(np.median(ts))
    res[name] = y.clone()
out["bit_identical"] = bool(torch.equal(res["windowed"], res["plain"]))
out["nnz"] = int(row.numel())
print(json.dumps(out))
