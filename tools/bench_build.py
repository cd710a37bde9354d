#!/usr/bin/env python3
"""Set-up path on the device: raw edge list -> (to_undirected) -> summed CSR -> normalize_adj -> transposed CSR
(/root/reference/utils.py:99-106, :406-435, LargeScaleCondensing.py:478-488), timed with HIP events, beside
scipy doing the same on a bounded sample of the edges on the host."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from disco_amd import graphgen, utils  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-edges", type=int, default=4_000_000)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    spec = graphgen.SPECS[args.workload]
    n = spec.n_nodes
    row, col = graphgen.powerlaw_edges(n, spec.nnz, spec.skew, spec.seed, device=dev, symmetric=False)
    perm = torch.randperm(row.numel(), device=dev)
    ei = torch.stack([row[perm], col[perm]])          # raw, unsorted edge list as a dataset file holds it
    del row, col, perm
    E = ei.shape[1]
    times = {"adjacency": [], "normalize": [], "transpose": []}
    for rep in range(args.reps + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        adj = utils.adjacency_from_edge_index(ei, n, undirected=True)
        ev[1].record()
        nadj = utils.normalize_adj_csr(adj)
        ev[2].record()
        adj_t = nadj.t()
        adj_t._rowptr  # noqa: B018
        ev[3].record()
        torch.cuda.synchronize()
        if rep:
            for k, name in enumerate(times):
                times[name].append(ev[k].elapsed_time(ev[k + 1]))
    med = {k: float(np.median(v)) for k, v in times.items()}
    total = sum(med.values())
    out = {"workload": spec.name, "n": n, "raw_edges": E, "nnz": adj_t.nnz(), "ms": med, "total_ms": total,
           "raw_edges_per_s": E / total * 1e3}
    # host: scipy on a bounded sample (same steps, /root/reference/utils.py:104-106 + :406-435)
    import scipy.sparse as sp
    k = min(args.cpu_edges, E)
    sub = ei[:, :k].cpu().numpy()
    t0 = time.perf_counter()
    r = np.concatenate([sub[0], sub[1]])
    c = np.concatenate([sub[1], sub[0]])
    key = np.unique(r * n + c)
    m = sp.csr_matrix((np.ones(key.size), (key // n, key % n)), shape=(n, n))
    mx = m.tolil()
    if mx[0, 0] == 0:
        mx = mx + sp.eye(n)
    rs = np.array(mx.sum(1))
    with np.errstate(divide="ignore"):
        ri = np.power(rs, -0.5).flatten()
    ri[np.isinf(ri)] = 0
    mx = sp.diags(ri).dot(mx).dot(sp.diags(ri))
    mt = mx.tocoo().astype(np.float32)
    sp.csr_matrix((mt.data, (mt.col, mt.row)), shape=(n, n))
    cpu_s = time.perf_counter() - t0
    out["cpu_scipy"] = {"edges": k, "seconds": cpu_s, "raw_edges_per_s": k / cpu_s, "cores": 1}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
