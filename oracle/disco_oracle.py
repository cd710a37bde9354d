"""CPU oracle for the DisCo propagation hot path.  TEST INFRASTRUCTURE ONLY.

This module is a CPU restatement (numpy / scipy / stock torch-CPU) of what the
reference computes on the path BASELINE.json names.  It is imported by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` -- never by the product package ``disco_amd``.

PARITY STATUS: *parity unpinned by the reference* for everything except the
pair scorer.  The reference ships no tests, golden vectors or fixtures
 and its arithmetic lives in torch_geometric 2.1.0 /
torch_sparse 0.6.15 / torch_scatter 2.0.9, none of which is vendored under
/root/reference or importable here.  The functions below therefore follow the
reference's own call sites line by line (file:line cited per function) and the
published behaviour of those pinned packages.  ``PGE_Edge`` is the exception:
``/root/reference/models/parametrized_adj_lp.py`` is pure torch, and
``tests/golden/make_scorer_golden.py`` imports it in the build container to
generate the committed scorer vectors the restatement here is checked against.

Conventions
-----------
* ``edge_index`` is int64 ``[2, E]`` with ``row = edge_index[0]`` (source j) and
  ``col = edge_index[1]`` (target i); propagation is ``out[col] += w * x[row]``
  (PyG ``flow='source_to_target'``; /root/reference/models/convs_large/sg_conv.py:89-90).
* ``adj_t`` is a CSR triple ``(rowptr, col, val)``; propagation is ``adj_t @ x``
  (/root/reference/models/convs_large/sg_conv.py:92-93).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------
# A1: gcn_norm, edge-index branch
# --------------------------------------------------------------------------

def add_remaining_self_loops(edge_index, edge_weight, fill_value, num_nodes):
    """torch_geometric 2.1.0 ``utils.add_remaining_self_loops`` [upstream, restated].

    Keeps non-loop edges in order, then appends one loop per node whose weight is
    ``fill_value`` unless the input already had a loop on that node, in which case
    the (last) existing loop weight is kept.  Called from
    /root/reference/utils.py:236-240 (never enabled by any driver).
    """
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop_w = np.full((num_nodes,), fill_value, dtype=edge_weight.dtype)
    inv = ~mask
    loop_w[row[inv]] = edge_weight[inv]
    loops = np.arange(num_nodes, dtype=edge_index.dtype)
    ei = np.concatenate([edge_index[:, mask], np.stack([loops, loops])], axis=1)
    ew = np.concatenate([edge_weight[mask], loop_w])
    return ei, ew


def gcn_norm_coo(edge_index, edge_weight=None, num_nodes=None, improved=False,
                 add_self_loops=True, flow="source_to_target"):
    """/root/reference/utils.py:228-247 (identical copy at
    /root/reference/models/convs_large/gcn_conv.py:51-70), float32 throughout.

    deg = scatter_add(w, col)  -- in-degree by *column* (utils.py:243-244); the
    CPU scatter_add is a serial loop in edge order, restated with ``np.add.at``.
    deg^-0.5 with inf -> 0 (utils.py:245-246); torch's CPU ``pow_(-0.5)`` is
    ``1/sqrt(x)`` (two correctly rounded float32 ops) -- checked bitwise in
    tests/test_oracle.py.  Result ``dis[row] * w * dis[col]`` evaluated left to
    right (utils.py:247).
    """
    edge_index = np.asarray(edge_index, dtype=np.int64)
    E = edge_index.shape[1]
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if E else 0
    fill_value = 2.0 if improved else 1.0
    if edge_weight is None:
        edge_weight = np.ones((E,), dtype=np.float32)
    edge_weight = np.asarray(edge_weight, dtype=np.float32)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(
            edge_index, edge_weight, np.float32(fill_value), num_nodes)
    row, col = edge_index[0], edge_index[1]
    idx = col if flow == "source_to_target" else row
    deg = np.zeros((num_nodes,), dtype=np.float32)
    np.add.at(deg, idx, edge_weight)          # serial float32 adds in edge order
    with np.errstate(divide="ignore", invalid="ignore"):
        dis = np.float32(1.0) / np.sqrt(deg)
    dis[np.isinf(dis)] = np.float32(0.0)
    out = (dis[row] * edge_weight) * dis[col]
    return edge_index, out.astype(np.float32)


def gcn_norm_sparse(row, col, val, n, add_self_loops=True, improved=False):
    """/root/reference/utils.py:214-226: the SparseTensor branch of ``gcn_norm`` (dead code in the reference --
    no driver passes a SparseTensor -- restated for the A2 row; torch_sparse's ``fill_diag`` / ``sum`` / ``mul``
    are [upstream, recalled], parity unpinned).  float32 throughout:

    * no values -> all ones (:217-218);
    * ``fill_diag(adj_t, fill_value)``: the whole diagonal is set to ``fill_value`` -- existing diagonal entries
      are replaced, missing ones inserted (:219-220);
    * ``deg = sum(adj_t, dim=1)``: ROW sums, sequential float32 adds in CSR (row-major sorted) order (:221);
    * ``deg^-0.5`` (torch CPU: ``1/sqrt``), inf -> 0 (:222-223);
    * ``mul`` by ``dis[row]`` then by ``dis[col]`` (:224-225).
    Input COO (any order, no duplicates); returns ``(row, col, val)`` row-major sorted.
    """
    row, col = np.asarray(row, dtype=np.int64), np.asarray(col, dtype=np.int64)
    val = np.ones(len(row), np.float32) if val is None else np.asarray(val, dtype=np.float32)
    if add_self_loops:
        keep = row != col
        d = np.arange(n, dtype=np.int64)
        row, col = np.concatenate([row[keep], d]), np.concatenate([col[keep], d])
        val = np.concatenate([val[keep], np.full(n, 2.0 if improved else 1.0, np.float32)])
    order = np.lexsort((col, row))
    row, col, val = row[order], col[order], val[order]
    deg = np.zeros(n, np.float32)
    for r, v in zip(row, val):                    # sequential float32 accumulation in CSR order
        deg[r] = np.float32(deg[r] + v)
    with np.errstate(divide="ignore", invalid="ignore"):
        dis = np.float32(1.0) / np.sqrt(deg)
    dis[np.isinf(dis)] = np.float32(0.0)
    return row, col, ((val * dis[row]) * dis[col]).astype(np.float32)


# --------------------------------------------------------------------------
# A3: normalize_adj_tensor(adj, sparse=True)
# --------------------------------------------------------------------------

def normalize_adj_scipy(row, col, val, n):
    """/root/reference/utils.py:473-484 -> to_scipy (:651-656) -> normalize_adj
    (:406-435) -> sparse_mx_to_torch_sparse_tensor (:632-639).

    * ``sp.csr_matrix((values, indices))`` sums duplicate entries (:656).
    * ``if mx[0, 0] == 0: mx = mx + I`` -- self-loops are added to EVERY node iff
      node 0 has none (:425-426).
    * ``r = rowsum ** -1/2`` over ROW sums, inf -> 0 (:427-429);
      ``diag(r) @ mx @ diag(r)`` (:430-432); ``.astype(np.float32)`` (:634).
    * dtype quirk (scipy promotion, verified here): when the identity is added the
      matrix becomes float64 and all of the above runs in float64 with one final
      rounding; when it is NOT added (node 0 has a self-loop) the float32 matrix
      stays float32 -- sequential float32 row sums, numpy's float32 ``power``
      (a few ulp off correctly rounded, CPU dependent) and float32 products.
    Returns COO ``(row, col, val_f32)`` in scipy's ``tocoo()`` order.
    """
    mx = sp.csr_matrix((np.asarray(val), (np.asarray(row), np.asarray(col))), shape=(n, n))
    mx = mx.tolil()
    if mx[0, 0] == 0:
        mx = mx + sp.eye(mx.shape[0])
    rowsum = np.array(mx.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1 / 2).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    r_mat_inv = sp.diags(r_inv)
    mx = r_mat_inv.dot(mx)
    mx = mx.dot(r_mat_inv)
    coo = mx.tocoo().astype(np.float32)
    return coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data


def normalize_adj_dense(adj):
    """/root/reference/utils.py:485-494, dense branch: always adds I, float32."""
    adj = np.asarray(adj, dtype=np.float32)
    mx = adj + np.eye(adj.shape[0], dtype=np.float32)
    rowsum = mx.sum(1, dtype=np.float32)
    with np.errstate(divide="ignore"):
        r = np.power(rowsum, np.float32(-0.5), dtype=np.float32)
    r[np.isinf(r)] = 0.0
    return (np.diag(r) @ mx) @ np.diag(r)


# --------------------------------------------------------------------------
# CSR helpers (what SparseTensor(row, col, value).t() hands to the conv)
# --------------------------------------------------------------------------

def coo_to_csr(row, col, val, n_rows):
    """Sort by (row, col) keeping duplicates -- torch_sparse.SparseTensor
    construction [upstream, restated]; used at
    /root/reference/LargeScaleCondensing.py:486."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    order = np.lexsort((col, row))
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, row + 1, 1)
    rowptr = np.cumsum(rowptr)
    v = None if val is None else np.asarray(val)[order]
    return rowptr, col[order].astype(np.int64), v


def adj_t_from_coo(row, col, val, n):
    """``SparseTensor(row=row, col=col, value=val).t()``: CSR whose rows are the
    original columns (/root/reference/LargeScaleCondensing.py:486)."""
    return coo_to_csr(col, row, val, n)


def to_undirected(edge_index, num_nodes):
    """PyG ``to_undirected(edge_index, edge_attr=None, num_nodes=)`` [torch_geometric 2.1.0, restated]:
    concatenate both directions, ``coalesce`` = sort by row * N + col and drop repeats.  Call site:
    /root/reference/utils.py:100."""
    ei = np.asarray(edge_index, dtype=np.int64)
    r = np.concatenate([ei[0], ei[1]])
    c = np.concatenate([ei[1], ei[0]])
    key = np.unique(r * num_nodes + c)
    return np.stack([key // num_nodes, key % num_nodes])


def pyg2dpr_adjacency(edge_index, num_nodes, undirected=False):
    """``Pyg2Dpr.__init__`` (/root/reference/utils.py:99-106): optional to_undirected, then
    ``sp.csr_matrix((np.ones(E), (row, col)), shape=(n, n))`` -- scipy itself, float64 ones,
    duplicates summed; canonical (sorted) indices."""
    import scipy.sparse as sp
    ei = np.asarray(edge_index, dtype=np.int64)
    if undirected:
        ei = to_undirected(ei, num_nodes)
    m = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(num_nodes, num_nodes))
    m.sum_duplicates()
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int64), m.data


# --------------------------------------------------------------------------
# A4/A5/A10: propagation
# --------------------------------------------------------------------------

def spmm_csr(rowptr, col, val, x, reduce="sum", dtype=np.float64):
    """``torch_sparse.matmul(adj_t, x, reduce)`` [upstream, restated];
    call sites /root/reference/models/convs_large/sg_conv.py:92-93,
    gcn_conv.py:170-171, sage_conv.py:148-150, gin_conv.py:88-90.

    sum : y[r] = sum_e val[e] * x[col[e]]            (val None -> 1)
    mean: sum / max(row nnz, 1)
    max : running max of val[e] * x[col[e]]; rows without entries -> 0
    ``dtype=np.float64`` is the accuracy ground truth; ``np.float32`` accumulates
    sequentially in CSR order like the upstream CPU kernel.
    """
    rowptr = np.asarray(rowptr)
    col = np.asarray(col)
    x = np.asarray(x)
    n = len(rowptr) - 1
    F = x.shape[1]
    y = np.zeros((n, F), dtype=dtype)
    if reduce in ("sum", "add", "mean"):
        if dtype == np.float64:
            v = np.ones(len(col)) if val is None else np.asarray(val, dtype=np.float64)
            A = sp.csr_matrix((v, col, rowptr), shape=(n, x.shape[0]))
            y = A @ x.astype(np.float64)
        else:
            for r in range(n):
                acc = np.zeros(F, dtype=np.float32)
                for e in range(rowptr[r], rowptr[r + 1]):
                    w = np.float32(1.0) if val is None else np.float32(val[e])
                    acc = acc + w * x[col[e]].astype(np.float32)
                y[r] = acc
        if reduce == "mean":
            cnt = np.maximum(np.diff(rowptr), 1).astype(dtype)
            y = y / cnt[:, None]
        return y.astype(dtype)
    if reduce == "max":
        for r in range(n):
            b, e = rowptr[r], rowptr[r + 1]
            if e > b:
                g = x[col[b:e]].astype(dtype)
                if val is not None:
                    g = g * np.asarray(val[b:e], dtype=dtype)[:, None]
                y[r] = g.max(axis=0)
        return y
    raise ValueError(reduce)


def propagate_scatter(edge_index, edge_weight, x, num_nodes=None, dtype=np.float64):
    """PyG's non-fused route for a Tensor ``edge_index``: ``x_j = x[row]``,
    ``message = w * x_j`` (/root/reference/models/convs_large/sg_conv.py:89-90),
    ``aggregate = scatter_add over col``.  out = A^T x with A[row, col] = w."""
    row, col = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    x = np.asarray(x)
    n = x.shape[0] if num_nodes is None else num_nodes
    msg = x[row].astype(dtype)
    if edge_weight is not None:
        msg = msg * np.asarray(edge_weight, dtype=dtype)[:, None]
    out = np.zeros((n, x.shape[1]), dtype=dtype)
    np.add.at(out, col, msg)
    return out


def khop_csr(rowptr, col, val, x, K, dtype=np.float64):
    """``for k in range(K): x = propagate(adj_t, x=x)``
    (/root/reference/models/convs_large/sg_conv.py:74-77)."""
    y = np.asarray(x).astype(dtype)
    for _ in range(K):
        y = spmm_csr(rowptr, col, val, y, "sum", dtype)
    return y


def spmm_abs_bound(rowptr, col, val, x):
    """sum_e |w_e * x_e| per output element: the scale of the dot-product
    forward-error bound used by the fp32 tolerance."""
    v = None if val is None else np.abs(np.asarray(val, dtype=np.float64))
    return spmm_csr(rowptr, col, v, np.abs(np.asarray(x, dtype=np.float64)), "sum", np.float64)


def normalize_sparse_tensor(row, col, w, n, fill_value=1.0):
    """/root/reference/utils.py:437-455: append n self-loops of ``fill_value``, ``deg = scatter_add(w, row)``
    serially in float32 in edge order, ``deg^-1/2`` (inf -> 0, 1/sqrt correctly rounded), left-to-right float32
    product.  Returns (row, col, values) with the loops appended."""
    row = np.concatenate([np.asarray(row, np.int64), np.arange(n, dtype=np.int64)])
    col = np.concatenate([np.asarray(col, np.int64), np.arange(n, dtype=np.int64)])
    w = np.concatenate([np.asarray(w, np.float32), np.full(n, fill_value, np.float32)])
    deg = np.zeros(n, np.float32)
    for e in range(len(w)):                       # serial scatter_add, edge order
        deg[row[e]] = np.float32(deg[row[e]] + w[e])
    with np.errstate(divide="ignore"):
        dis = (np.float32(1.0) / np.sqrt(deg, dtype=np.float32)).astype(np.float32)
    dis[np.isinf(dis)] = 0
    return row, col, ((dis[row] * w).astype(np.float32) * dis[col]).astype(np.float32)


# --------------------------------------------------------------------------
# A7: max aggregation of the edge-translation link predictor
# --------------------------------------------------------------------------

def max_aggregate(rowptr, col, x):
    """``MessagePassing(aggr="max").propagate(aggr_adj, x=feat)``
    (/root/reference/LargeScaleCondensing.py:152-153,172): element-wise max over
    the in-neighbours given by the CSR rows of ``aggr_adj``; edge values are
    ignored, rows without neighbours give 0.  Order independent -> bit exact."""
    x = np.asarray(x, dtype=np.float32)
    return spmm_csr(rowptr, col, None, x, "max", np.float32)


def edge_translation_features(rowptr, col, x):
    """``concat((feat, feat_transform), dim=1)``
    (/root/reference/LargeScaleCondensing.py:173)."""
    x = np.asarray(x, dtype=np.float32)
    return np.concatenate([x, max_aggregate(rowptr, col, x)], axis=1)


def anchor_max(feat_transform_rows, anchor, width=None, index=None):
    """/root/reference/LargeScaleCondensing.py:299 (``_induct.py:293``, ``_Sampled.py:259-261``):
    ``feat_transform[index[c]][anchor, :L*d].max(dim=1).values`` -- per synthetic node the
    element-wise max over its anchors' rows; ``index`` selects the class's train rows first,
    ``width`` is the column cut ``:L*d``.  NaN propagates (torch.max / np.max alike)."""
    t = np.asarray(feat_transform_rows)
    if index is not None:
        t = t[np.asarray(index)]
    g = t[np.asarray(anchor)]
    if width is not None:
        g = g[:, :, :width]
    return g.max(axis=1)


# --------------------------------------------------------------------------
# A9: PGE_Edge pair scorer + the three post-processing orders
# --------------------------------------------------------------------------

def pge_edge_logits(params, edge_embed, dtype=np.float64, eps=1e-5):
    """/root/reference/models/parametrized_adj_lp.py:29-38 in eval mode: Linear,
    BatchNorm1d(running stats), ReLU, ..., final Linear; returns the pre-sigmoid
    logit.  ``params`` is the module's ``state_dict`` as numpy arrays
    (keys ``layers.{i}.{weight,bias}``, ``bns.{i}.{weight,bias,running_mean,running_var}``)."""
    h = np.asarray(edge_embed, dtype=dtype)
    nl = len([k for k in params if k.startswith("layers.") and k.endswith(".weight")])
    for i in range(nl):
        W = np.asarray(params[f"layers.{i}.weight"], dtype=dtype)
        b = np.asarray(params[f"layers.{i}.bias"], dtype=dtype)
        h = h @ W.T + b
        if i != nl - 1:
            g = np.asarray(params[f"bns.{i}.weight"], dtype=dtype)
            be = np.asarray(params[f"bns.{i}.bias"], dtype=dtype)
            mu = np.asarray(params[f"bns.{i}.running_mean"], dtype=dtype)
            var = np.asarray(params[f"bns.{i}.running_var"], dtype=dtype)
            h = (h - mu) / np.sqrt(var + eps) * g + be
            h = np.maximum(h, 0)
    return h.reshape(-1)


def sigmoid(z):
    z = np.asarray(z)
    return 1.0 / (1.0 + np.exp(-z))


def pair_score_matrix(params, Z, dtype=np.float64):
    """``adj_syn[i] = pge_edge.inference(cat([Z[[i]*n], Z[arange(n)]], 1))``
    (/root/reference/LargeScaleCondensing.py:304-305): S[i, j] = score([z_i || z_j])."""
    Z = np.asarray(Z, dtype=dtype)
    n = Z.shape[0]
    S = np.zeros((n, n), dtype=dtype)
    for i in range(n):
        emb = np.concatenate([np.repeat(Z[i:i + 1], n, axis=0), Z], axis=1)
        S[i] = sigmoid(pge_edge_logits(params, emb, dtype))
    return S


def edge_construction(S, threshold, variant):
    """The three post-processing orders applied to the score matrix ``S``:

    transductive /root/reference/LargeScaleCondensing.py:306-310:
        A=(S+S^T)/2; A[A<thr]=0; diag=1; nonzero.
    inductive    /root/reference/LargeScaleCondensing_induct.py:300-304:
        A=(S+S^T)/2; diag=1; A[A<thr]=0; nonzero.
    sampled      /root/reference/LargeScaleCondensing_Sampled.py:271-285:
        only j>i, A_ij=(S_ij+S_ji)/2, threshold, to_undirected, add_self_loops(1).
    Returns (edge_index int64 [2,E] in row-major order, weight float32 [E]).
    """
    S = np.asarray(S, dtype=np.float32)
    n = S.shape[0]
    A = (S + S.T) / np.float32(2)
    if variant == "transductive":
        A[A < threshold] = 0
        np.fill_diagonal(A, 1)
    elif variant == "inductive":
        np.fill_diagonal(A, 1)
        A[A < threshold] = 0
    elif variant == "sampled":
        # to_undirected() returns the coalesced (row-major sorted) symmetric
        # edge list; add_self_loops() then APPENDS the n unit loops.
        U = np.triu(A, 1)
        U[U < threshold] = 0
        A = U + U.T
        r, c = np.nonzero(A)
        loops = np.arange(n)
        ei = np.stack([np.concatenate([r, loops]), np.concatenate([c, loops])])
        ew = np.concatenate([A[r, c], np.ones(n, dtype=np.float32)])
        return ei.astype(np.int64), ew.astype(np.float32)
    else:
        raise ValueError(variant)
    r, c = np.nonzero(A)
    return np.stack([r, c]).astype(np.int64), A[r, c].astype(np.float32)
