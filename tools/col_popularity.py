#!/usr/bin/env python3
"""Share of all gathers that goes to the k most popular source rows (column popularity CDF) of the bench graphs --
the measurement behind DESIGN.md's hot-row paragraph.  Usage: python tools/col_popularity.py > profiles/r2/col_popularity.log"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_amd import graphgen  # noqa: E402

dev = torch.device("cuda:0")
for name, div in (("reddit", 1), ("products", 1), ("papers100M", 8)):
    spec = graphgen.SPECS[name]
    n, nnz = spec.n_nodes // div, spec.nnz // div
    for comm in (0, graphgen.COMMUNITY_SIZE):
        row, col = graphgen.powerlaw_edges(n, nnz, spec.skew, spec.seed, device=dev, comm_size=comm,
                                           mu=graphgen.COMMUNITY_MU if comm else 1.0)
        pop = torch.bincount(col, minlength=n).sort(descending=True).values.double()
        cdf = pop.cumsum(0) / pop.sum()
        ks = [256, 1024, 4096, 16384, 65536, 262144]
        line = ", ".join(f"top {k}: {float(cdf[min(k, n) - 1]) * 100:.1f} %" for k in ks if k <= n)
        print(f"{name:18s} {'community' if comm else 'config   '} N={n} nnz={col.numel()} row bytes={spec.feat * 4}: {line}")
        del row, col, pop, cdf
