"""Generates tests/golden/graph_golden.npz: tiny hand-checkable graphs with the quirks SURVEY.md
section 8c lists (isolated node, node 0 with / without self-loop, duplicate edge, zero in-degree
column, hub row, asymmetric weights) and the oracle's outputs on them for F in {1, 3, 100, 128, 602}.

The reference has no fixtures of its own (parity unpinned); these vectors pin the ORACLE so that a
later edit of oracle/ cannot silently move the target, and give the GPU tests fixed inputs.  The
normalised weights of the path/star graphs are additionally checked against closed forms in
tests/test_oracle.py.

Run:  python tests/golden/make_graph_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import disco_oracle as orc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_golden.npz")


def graphs():
    g = {}
    # path 0-1-2-3, symmetric, node 4 isolated
    g["path"] = (5, np.array([[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]]), None)
    # star with hub 1 (so node 0 has no self loop), plus self-loop on node 3
    ei = [[1, i] for i in range(2, 9)] + [[i, 1] for i in range(2, 9)] + [[3, 3]]
    g["star"] = (9, np.array(ei).T, None)
    # node 0 HAS a self loop -> normalize_adj adds no identity at all
    g["loop0"] = (4, np.array([[0, 0, 1, 2, 3], [0, 1, 0, 3, 2]]), np.array([1.0, 1.0, 1.0, 2.0, 2.0], dtype=np.float32))
    # duplicate edge (0->1 twice), zero in-degree column (node 3 is never a target), asymmetric weights
    g["dup"] = (4, np.array([[0, 0, 1, 2, 3, 3], [1, 1, 2, 0, 0, 1]]),
                np.array([0.5, 0.25, 2.0, 1.5, 0.75, 3.0], dtype=np.float32))
    # hub row: node 0 receives from everybody (70 in-edges)
    n = 71
    src = np.arange(1, n)
    g["hub"] = (n, np.stack([np.concatenate([src, np.zeros(n - 1, dtype=int)]), np.concatenate([np.zeros(n - 1, dtype=int), src])]), None)
    return g


def main():
    rng = np.random.default_rng(20250222)
    out = {}
    for name, (n, ei, w) in graphs().items():
        ei = ei.astype(np.int64)
        out[f"{name}_n"] = np.int64(n)
        out[f"{name}_ei"] = ei
        if w is not None:
            out[f"{name}_w"] = w
        # A1: gcn_norm(edge_index, w, n, add_self_loops=False) and with self loops
        _, w1 = orc.gcn_norm_coo(ei, w, n, add_self_loops=False)
        ei2, w2 = orc.gcn_norm_coo(ei, w, n, add_self_loops=True)
        out[f"{name}_gcn_norm_noloop"] = w1
        out[f"{name}_gcn_norm_loop_ei"] = ei2
        out[f"{name}_gcn_norm_loop"] = w2
        # A3: normalize_adj_tensor(adj, sparse=True)
        vals = np.ones(ei.shape[1], dtype=np.float32) if w is None else w
        r3, c3, v3 = orc.normalize_adj_scipy(ei[0], ei[1], vals, n)
        out[f"{name}_normadj_row"], out[f"{name}_normadj_col"], out[f"{name}_normadj_val"] = r3, c3, v3
        # adj_t = SparseTensor(row, col, value).t()
        rowptr, col, val = orc.adj_t_from_coo(r3, c3, v3, n)
        out[f"{name}_adjt_rowptr"], out[f"{name}_adjt_col"], out[f"{name}_adjt_val"] = rowptr, col, val
        for F in ((1, 3, 100, 128, 602) if n <= 8 else (1, 3, 128)):
            x = rng.standard_normal((n, F)).astype(np.float32)
            out[f"{name}_x{F}"] = x
            for K in (1, 2):
                out[f"{name}_khop{K}_F{F}_f64"] = orc.khop_csr(rowptr, col, val, x, K, np.float64)
            if F in (3, 128):
                out[f"{name}_max_F{F}"] = orc.max_aggregate(rowptr, col, x)
                out[f"{name}_mean_F{F}_f64"] = orc.spmm_csr(rowptr, col, val, x, "mean", np.float64)
                # Tensor edge_index convention on the un-normalised weights
                out[f"{name}_scatter_F{F}_f64"] = orc.propagate_scatter(ei, w, x, n, np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes", len(out), "arrays")


if __name__ == "__main__":
    main()
