#!/usr/bin/env python3
"""Resident-graph evaluation: `model.predict(feat, adj)` on the Reddit-shaped graph for the
evaluation models the reference uses (LargeScaleCondensing.py:330-355 calls it 31 times per run, re-uploading
CSR + X every time; here both stay on the device).  HIP-event timing, median of 5."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disco_amd import SparseTensor, graphgen, utils  # noqa: E402
from disco_amd.models.basicgnn_large import GCN, SGC, GraphSAGE  # noqa: E402

dev = torch.device("cuda", 0)
spec = graphgen.SPECS["reddit"]
n, F, nclass = spec.n_nodes, spec.feat, 41
row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev)
adj_t = utils.normalize_adj_csr(SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True)).t()
del row, col
x = graphgen.features(n, F, spec.seed, device=dev, ld=608)
out = {"n": n, "nnz": adj_t.nnz(), "F": F}
for name, ctor in (("SGC_K2", lambda: SGC(nfeat=F, nhid=256, nclass=nclass, nlayers=2, sgc=True)),
                   ("GCN_2layer", lambda: GCN(nfeat=F, nhid=256, nclass=nclass, nlayers=2)),
                   ("SAGE_2layer", lambda: GraphSAGE(nfeat=F, nhid=256, nclass=nclass, nlayers=2))):
    torch.manual_seed(0)
    model = ctor().to(dev)
    ts = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = model.predict(x, adj_t)
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    assert y.shape == (n, nclass) and bool(torch.isfinite(y).all())
    out[name + "_ms"] = float(np.median(ts))
print(json.dumps(out))
