#!/usr/bin/env python3
"""Normalisation kernels against their HBM roofline ("gcn_norm bytes"):
  * A1  utils.gcn_norm(edge_index [2,E] int64, None, N, add_self_loops=False)  (_Sampled.py:491 call shape)
        algorithmic bytes: E*8 (col, degree) + E*16 (row, col) + E*4 (out) + N*12
  * A3  disco_csr_rowsum_f64 + disco_gcn_norm_csr_f32 on a built CSR: nnz*12 + N*16
HIP-event timing, median of reps.  (The CPU side of these kernels is timed in bench.py / the tests only: tools do
not touch oracle/.)"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from disco_amd import SparseTensor, _lib, graphgen, utils  # noqa: E402


def timed(fn, reps):
    ts = []
    for i in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-edges", type=int, default=0, help="unused (kept for old command lines)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    spec = graphgen.SPECS[args.workload]
    n = spec.n_nodes
    row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev)
    ei = torch.stack([row, col])
    E = ei.shape[1]
    out = {"workload": spec.name, "n": n, "E": E}
    ms, _ = timed(lambda: utils.gcn_norm(ei, None, n, add_self_loops=False), args.reps)
    b = E * 8 + E * 16 + E * 4 + n * 12
    out["gcn_norm_coo"] = {"ms": ms, "algorithmic_GB": b / 1e9, "GBs": b / ms / 1e6, "frac_of_8TBs": b / ms / 1e6 / 8000}
    adj = SparseTensor(row=row, col=col, value=None, sparse_sizes=(n, n), is_sorted=True)
    adj = adj.set_value(torch.ones(adj.nnz(), device=dev))
    lib = _lib.load()
    rs = torch.empty(n, dtype=torch.float64, device=dev)
    o = torch.empty(adj.nnz(), dtype=torch.float32, device=dev)
    s = _lib.current_stream(dev)

    def csr_kernels():
        _lib.check(lib.disco_csr_rowsum_f64(adj._rowptr.data_ptr(), adj._value.data_ptr(), n, 1, rs.data_ptr(), s), "rowsum")
        _lib.check(lib.disco_gcn_norm_csr_f32(adj._rowptr.data_ptr(), adj._col32.data_ptr(), adj._value.data_ptr(), n, 1,
                                              rs.data_ptr(), o.data_ptr(), s), "scale")
    ms, _ = timed(csr_kernels, args.reps)
    b = adj.nnz() * 12 + n * 16
    out["normalize_adj_csr_kernels"] = {"ms": ms, "algorithmic_GB": b / 1e9, "GBs": b / ms / 1e6, "frac_of_8TBs": b / ms / 1e6 / 8000}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
