"""Generates tests/golden/norm_golden.npz by RUNNING the reference's own normalisation functions.

/root/reference/utils.py cannot be imported here (its module-level imports need faiss / ogb / torch_geometric /
deeprobust / torch_sparse, all absent), but the functions of row A3 use nothing beyond numpy, scipy and torch:

    normalize_adj_tensor  (utils.py:473-494)   to_scipy        (:651-660)   is_sparse_tensor (:662-679)
    normalize_adj         (:406-435)           sparse_mx_to_torch_sparse_tensor (:632-639)

This script parses utils.py with ``ast``, compiles ONLY those function definitions (no module-level statement of
the file is executed) and runs them in a namespace holding the real numpy / scipy.sparse / torch -- no stand-ins.
Only inputs and the reference's outputs are stored; the reference's text never enters the repository.
Inputs: the five golden graphs of make_graph_golden.py plus seeded random graphs for both dtype modes of
``normalize_adj`` (node 0 without a self-loop: float64 arithmetic; node 0 with one: float32 arithmetic), with
duplicate entries, isolated nodes and non-unit weights; and dense inputs for the ``sparse=False`` branch.

Run (build container only, needs /root/reference):  python tests/golden/make_norm_golden.py
"""
import ast
import os
import sys
import warnings

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_graph_golden import graphs  # noqa: E402

REF = "/root/reference/utils.py"
WANTED = ("normalize_adj", "normalize_adj_tensor", "to_scipy", "sparse_mx_to_torch_sparse_tensor", "is_sparse_tensor")
OUT = os.path.join(HERE, "norm_golden.npz")


def reference_functions():
    tree = ast.parse(open(REF).read(), filename=REF)
    picked = [node for node in tree.body if isinstance(node, ast.FunctionDef) and node.name in WANTED]
    assert sorted(f.name for f in picked) == sorted(WANTED), [f.name for f in picked]
    ns = {"np": np, "sp": sp, "torch": torch, "__name__": "reference_utils_extract"}
    exec(compile(ast.Module(body=picked, type_ignores=[]), REF, "exec"), ns)
    return ns


def random_graph(rng, n, e, loop0, weighted):
    row, col = rng.integers(0, n, e), rng.integers(0, n, e)
    keep = ~((row == 0) & (col == 0))
    row, col = row[keep], col[keep]
    iso = rng.choice(np.arange(1, n), size=max(1, n // 10), replace=False)      # isolated nodes
    keep = ~np.isin(row, iso) & ~np.isin(col, iso)
    row, col = row[keep], col[keep]
    dup = rng.choice(len(row), size=len(row) // 8, replace=False)               # duplicate entries (summed)
    row, col = np.concatenate([row, row[dup]]), np.concatenate([col, col[dup]])
    if loop0:
        row, col = np.concatenate([[0], row]), np.concatenate([[0], col])
    val = (rng.random(len(row)) * 2 + 0.25).astype(np.float32) if weighted else np.ones(len(row), np.float32)
    p = rng.permutation(len(row))
    return np.stack([row[p], col[p]]).astype(np.int64), val[p]


def main():
    ref = reference_functions()
    rng = np.random.default_rng(20250223)
    cases = {}
    for name, (n, ei, w) in graphs().items():
        cases[name] = (n, ei.astype(np.int64), np.ones(ei.shape[1], np.float32) if w is None else w)
    for tag, n, e, loop0, weighted in (("rand_f64_unit", 60, 500, False, False), ("rand_f64_w", 90, 900, False, True),
                                       ("rand_f32_unit", 60, 500, True, False), ("rand_f32_w", 90, 900, True, True),
                                       ("rand_f32_big", 400, 9000, True, True)):
        ei, w = random_graph(rng, n, e, loop0, weighted)
        cases[tag] = (n, ei, w)
    out = {"names": np.array(sorted(cases))}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, (n, ei, w) in cases.items():
            adj = torch.sparse_coo_tensor(torch.from_numpy(ei), torch.from_numpy(w), (n, n))   # uncoalesced, as loaded
            res = ref["normalize_adj_tensor"](adj, sparse=True).coalesce()
            out[f"{name}_n"] = np.int64(n)
            out[f"{name}_ei"], out[f"{name}_w"] = ei, w
            out[f"{name}_out_index"] = res.indices().numpy()
            out[f"{name}_out_value"] = res.values().numpy()
            assert res.values().dtype == torch.float32
        for tag, n in (("dense_a", 7), ("dense_b", 33)):
            a = (rng.random((n, n)) < 0.3).astype(np.float32) * rng.random((n, n)).astype(np.float32)
            a[n // 2] = 0
            a[n // 2, n // 2] = -1.0        # row sum of A + I is 0: r = inf -> 0
            out[f"{tag}_in"] = a
            out[f"{tag}_out"] = ref["normalize_adj_tensor"](torch.from_numpy(a), sparse=False).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(cases), "sparse cases, 2 dense cases")


if __name__ == "__main__":
    main()
