"""Adjacency normalisation with the reference's function names and semantics
(/root/reference/utils.py), computed by the HIP kernels of libdisco_hip.so.

Only the normalisation block of the reference's ``utils.py`` is on the hot path (SURVEY.md
section 2 row 3); data loading, metrics and the nettack helpers are out of scope.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from . import _lib
from .sparse import SparseTensor, coo_to_csr


def is_sparse_tensor(tensor) -> bool:
    """/root/reference/utils.py:662-679."""
    return isinstance(tensor, torch.Tensor) and tensor.layout == torch.sparse_coo


def maybe_num_nodes(edge_index: torch.Tensor, num_nodes: Optional[int] = None) -> int:
    if num_nodes is not None:
        return int(num_nodes)
    return int(edge_index.max()) + 1 if edge_index.numel() else 0


def add_self_loops(edge_index, edge_weight=None, fill_value=1, num_nodes=None):
    """Interface of /root/reference/utils.py:457-471: one (i, i) edge per node is APPENDED after the given edges
    (existing loops are kept, nothing is de-duplicated); weights, when given, get ``fill_value`` for the loops."""
    n = maybe_num_nodes(edge_index, num_nodes)
    ids = torch.arange(n, dtype=torch.long, device=edge_index.device)
    out_index = torch.cat((edge_index, torch.stack((ids, ids))), dim=1)
    if edge_weight is None:
        return out_index, None
    if edge_weight.numel() != edge_index.size(1):
        raise ValueError("edge_weight must have one entry per edge")
    return out_index, torch.cat((edge_weight, torch.full_like(edge_weight[:1], fill_value).expand(n)))


def add_remaining_self_loops(edge_index, edge_weight, fill_value, num_nodes):
    """torch_geometric 2.1.0 semantics (called at /root/reference/utils.py:237): keep non-loop
    edges, append one loop per node, existing loop weights win over ``fill_value``."""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop_w = edge_weight.new_full((num_nodes,), fill_value)
    inv = ~mask
    # several loops on one node: the LAST one in edge order wins, as the serial CPU assignment of the reference's stack
    # does -- an indexed assignment with repeated indices has no defined winner on the device, so pick it explicitly
    pos = torch.nonzero(inv).flatten()
    if pos.numel():
        last = torch.full((num_nodes,), -1, dtype=torch.long, device=edge_index.device)
        last.scatter_reduce_(0, row[pos], pos, reduce="amax", include_self=True)
        has = last >= 0
        loop_w[has] = edge_weight[last[has]]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    ei = torch.cat([edge_index[:, mask], torch.stack([loops, loops])], dim=1)
    ew = torch.cat([edge_weight[mask], loop_w])
    return ei, ew


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if t.device.type != "cuda":
        raise RuntimeError(f"{what}: tensors must live on the GPU (got {t.device}); disco_amd has no CPU fallback")


def _staging_device(t: torch.Tensor, what: str):
    """The reference calls its normalisation functions on HOST tensors (LargeScaleCondensing.py:480,
    LargeScaleCondensing_Sampled.py:491).  There is still no CPU arithmetic here: a host operand is copied to the
    current GPU, the kernels run there, and the result goes back to the operand's device like the reference's does.
    Returns the GPU to compute on, or None when the operand already lives on one."""
    if t.device.type == "cuda":
        return None
    if not torch.cuda.is_available():
        raise RuntimeError(f"{what}: no GPU is visible and disco_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def gcn_norm(edge_index: Union[torch.Tensor, SparseTensor], edge_weight: Optional[torch.Tensor] = None,
             num_nodes: Optional[int] = None, improved: bool = False, add_self_loops: bool = True,
             flow: str = "source_to_target", dtype=None):
    """/root/reference/utils.py:209-247 (copy at models/convs_large/gcn_conv.py:32-70).

    Tensor branch (:228-247): ``deg = scatter_add(w, col)`` (by COLUMN for source_to_target),
    ``dis = deg^-1/2`` with inf -> 0, returns ``(edge_index, dis[row] * w * dis[col])``.
      * edge_weight None: integer degree counts (exact, any order);
      * weighted: the degree is summed per node in original edge order in float32, i.e. the
        result of the reference's serial CPU scatter_add, bit for bit.
    SparseTensor branch (:214-226): fill_value(1) if no values, optional unit diagonal, ROW degree.
    """
    fill_value = 2.0 if improved else 1.0
    lib = _lib.load()
    if isinstance(edge_index, SparseTensor):
        assert flow in ["source_to_target"]
        adj_t = edge_index
        _require_cuda(adj_t._col32, "gcn_norm")
        if add_self_loops:
            adj_t = fill_diag(adj_t, fill_value)
        # float32 throughout, as torch_sparse computes it: sequential float32 row sums in CSR order, deg^-1/2 as
        # 1/sqrt (two correctly rounded float32 ops = torch's pow_(-0.5)), inf -> 0, (val * dis[row]) * dis[col]
        n = adj_t.size(0)
        dev = adj_t.device
        with torch.cuda.device(dev):
            s = _lib.current_stream(dev)
            rs = torch.empty(n, dtype=torch.float64, device=dev)
            val = adj_t._value
            if val is not None and val.dtype != torch.float32:
                val = val.to(torch.float32)
            _lib.check(lib.disco_csr_rowsum_f64(_lib.ptr(adj_t._rowptr), _lib.ptr(val), n, 0, _lib.ptr(rs), s), "disco_csr_rowsum_f64")
            dis = rs.to(torch.float32)                      # exact: the kernel's sums are float32 values
            _lib.check(lib.disco_inv_sqrt_f32(_lib.ptr(dis), n, s), "disco_inv_sqrt_f32")
            row, col, _ = adj_t.coo()
            out = torch.empty(adj_t.nnz(), dtype=torch.float32, device=dev)
            _lib.check(lib.disco_gcn_norm_coo_apply_f32(_lib.ptr(row), _lib.ptr(col), _lib.ptr(val), adj_t.nnz(), _lib.ptr(dis),
                                                        _lib.ptr(out), s), "disco_gcn_norm_coo_apply_f32")
        return adj_t.set_value(out)

    assert flow in ["source_to_target", "target_to_source"]
    stage = _staging_device(edge_index, "gcn_norm")
    if stage is not None:
        home = edge_index.device
        ei, ew = gcn_norm(edge_index.to(stage), None if edge_weight is None else edge_weight.to(stage), num_nodes, improved,
                          add_self_loops, flow, dtype)
        return (edge_index if not add_self_loops else ei.to(home)), ew.to(home)
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    if add_self_loops:
        if edge_weight is None:
            edge_weight = torch.ones((edge_index.size(1),), dtype=dtype or torch.float32, device=edge_index.device)
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, num_nodes)
    edge_index = edge_index.contiguous()
    row, col = edge_index[0], edge_index[1]
    idx = col if flow == "source_to_target" else row
    E = edge_index.size(1)
    dev = edge_index.device
    with torch.cuda.device(dev):
        s = _lib.current_stream(dev)
        deg = torch.empty(num_nodes, dtype=torch.float32, device=dev)
        if edge_weight is None:
            if E >= (1 << 20):      # many edges: sort + run lengths beats one global atomic per edge (same integers)
                ws = torch.empty(max(int(lib.disco_degree_count_sorted_workspace_bytes(E, num_nodes)), 256), dtype=torch.uint8, device=dev)
                _lib.check(lib.disco_degree_count_sorted_f32(_lib.ptr(idx), E, num_nodes, _lib.ptr(deg), _lib.ptr(ws), ws.numel(), s),
                           "disco_degree_count_sorted_f32")
                del ws
            else:
                _lib.check(lib.disco_degree_count_f32(_lib.ptr(idx), E, num_nodes, _lib.ptr(deg), s), "disco_degree_count_f32")
            w = None
        elif E == 0:
            w = edge_weight.to(torch.float32).contiguous()
            deg.zero_()
        else:
            w = edge_weight.to(torch.float32).contiguous()
            order = torch.argsort(idx, stable=True)
            segptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dev)
            torch.cumsum(torch.bincount(idx, minlength=num_nodes), 0, out=segptr[1:])
            _lib.check(lib.disco_degree_ordered_f32(_lib.ptr(segptr), _lib.ptr(order), _lib.ptr(w), num_nodes,
                                                    _lib.ptr(deg), s), "disco_degree_ordered_f32")
        _lib.check(lib.disco_inv_sqrt_f32(_lib.ptr(deg), num_nodes, s), "disco_inv_sqrt_f32")
        out = torch.empty(E, dtype=torch.float32, device=dev)
        _lib.check(lib.disco_gcn_norm_coo_apply_f32(_lib.ptr(row), _lib.ptr(col), _lib.ptr(w), E, _lib.ptr(deg),
                                                    _lib.ptr(out), s), "disco_gcn_norm_coo_apply_f32")
    return edge_index, out


def fill_diag(adj: SparseTensor, fill_value: float) -> SparseTensor:
    """torch_sparse.fill_diag [upstream, restated]: set the whole diagonal to ``fill_value``
    (existing diagonal entries are replaced).  Used only by the dead SparseTensor branch of
    gcn_norm (/root/reference/utils.py:219-220)."""
    row, col, val = adj.coo()
    n = min(adj.sparse_sizes())
    if val is None:
        val = torch.ones(adj.nnz(), dtype=torch.float32, device=adj.device)
    keep = row != col
    d = torch.arange(n, device=adj.device)
    return SparseTensor(row=torch.cat([row[keep], d]), col=torch.cat([col[keep], d]),
                        value=torch.cat([val[keep], val.new_full((n,), fill_value)]), sparse_sizes=adj.sparse_sizes())


def _with_diagonal(adj: SparseTensor, row: torch.Tensor, col: torch.Tensor, val: torch.Tensor) -> SparseTensor:
    """CSR of the same matrix with a (zero-valued) diagonal entry in every row: the structure of ``mx + I``.
    Columns sorted inside the rows (what every constructor of this package produces): a device merge,
    ``disco_csr_insert_diagonal``; otherwise one more sort through ``coo_to_csr``."""
    import ctypes
    n, dev = adj.size(0), adj.device
    lib = _lib.load()
    # strictly increasing columns inside every row; duplicate entries take the sort path, which sums them
    unsorted = bool(((col[1:] <= col[:-1]) & (row[1:] == row[:-1])).any()) if col.numel() > 1 else False
    if adj.size(0) == adj.size(1) and not unsorted:
        with torch.cuda.device(dev):
            nb = lib.disco_csr_insert_diagonal_workspace_bytes(n)
            ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device=dev)
            rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
            c32 = torch.empty(adj.nnz() + n, dtype=torch.int32, device=dev)
            v = torch.empty(adj.nnz() + n, dtype=torch.float32, device=dev)
            nnz = ctypes.c_int64(0)
            _lib.check(lib.disco_csr_insert_diagonal(_lib.ptr(adj._rowptr), _lib.ptr(adj._col32), _lib.ptr(val), n, adj.nnz(),
                                                     _lib.ptr(rp), _lib.ptr(c32), _lib.ptr(v), ctypes.byref(nnz), _lib.ptr(ws),
                                                     ws.numel(), _lib.current_stream(dev)), "disco_csr_insert_diagonal")
        k = int(nnz.value)
        return SparseTensor(rowptr=rp, col=c32[:k], value=v[:k], sparse_sizes=adj.sparse_sizes(), is_sorted=True)
    d = torch.arange(n, device=dev)
    rp, c32, v = coo_to_csr(torch.cat([row, d]), torch.cat([col, d]),
                            torch.cat([val, torch.zeros(n, dtype=torch.float32, device=dev)]), adj.sparse_sizes(), dup="sum")
    return SparseTensor(rowptr=rp, col=c32, value=v, sparse_sizes=adj.sparse_sizes(), is_sorted=True)


def normalize_adj_csr(adj: SparseTensor) -> SparseTensor:
    """Core of ``normalize_adj`` (/root/reference/utils.py:406-435) on a CSR whose duplicates are
    already summed: ``if mx[0,0] == 0: mx += I``; r = rowsum^-1/2 (inf -> 0) in float64;
    val = float32(r[row] * val * r[col])."""
    _require_cuda(adj._col32, "normalize_adj")
    n = adj.size(0)
    row, col, val = adj.coo()
    if val is None:
        val = torch.ones(adj.nnz(), dtype=torch.float32, device=adj.device)
    has00 = bool(((row == 0) & (col == 0) & (val != 0)).any()) if n > 0 else True
    add_identity = 0 if has00 else 1
    if add_identity:
        # Structure of mx + I: a zero-valued diagonal entry is merged into every row (x + 0 is exact in
        # float32); the kernels add the 1 in float64, as scipy's `mx + sp.eye(n)` does.
        adj = _with_diagonal(adj, row, col, val.to(torch.float32).contiguous())
    elif adj._value is not None and adj._value.dtype != torch.float32:
        adj = adj.set_value(adj._value.to(torch.float32))       # the kernels read float32 values
    lib = _lib.load()
    with torch.cuda.device(adj.device):
        s = _lib.current_stream(adj.device)
        rs = torch.empty(n, dtype=torch.float64, device=adj.device)
        _lib.check(lib.disco_csr_rowsum_f64(_lib.ptr(adj._rowptr), _lib.ptr(adj._value), n, add_identity, _lib.ptr(rs), s),
                   "disco_csr_rowsum_f64")
        out = torch.empty(adj.nnz(), dtype=torch.float32, device=adj.device)
        _lib.check(lib.disco_gcn_norm_csr_f32(_lib.ptr(adj._rowptr), _lib.ptr(adj._col32), _lib.ptr(adj._value), n,
                                              add_identity, _lib.ptr(rs), _lib.ptr(out), s), "disco_gcn_norm_csr_f32")
    return adj.set_value(out)


def normalize_adj_tensor(adj: torch.Tensor, sparse: bool = False):
    """/root/reference/utils.py:473-494.

    sparse=True : adj is a torch sparse COO tensor; duplicates are summed (``to_scipy`` ->
                  ``csr_matrix``, :651-656), then normalize_adj (:406-435); returns a torch sparse
                  COO float32 tensor on adj's device like the reference (:484).
    sparse=False: dense ``D^-1/2 (A + I) D^-1/2`` in float32 with stock torch ops (:486-493) --
                  a small dense matrix product, not part of the accelerated path.
    """
    if sparse:
        stage = _staging_device(adj, "normalize_adj_tensor")
        if stage is not None:
            return normalize_adj_tensor(adj.to(stage), sparse=True).to(adj.device)
        st = SparseTensor.from_torch_sparse_coo_tensor(adj)
        out = normalize_adj_csr(st)
        row, col, val = out.coo()
        return torch.sparse_coo_tensor(torch.stack([row, col]), val, adj.shape).coalesce()
    # dense branch: r = rowsum(A + I)^-1/2 with inf -> 0, result diag(r) (A + I) diag(r) -- the two diagonal
    # products are row / column scalings (same float32 products as multiplying by the diagonal matrices)
    a_hat = adj + torch.eye(adj.shape[0], dtype=adj.dtype, device=adj.device)
    r = a_hat.sum(dim=1).pow(-0.5)
    r = torch.where(torch.isinf(r), torch.zeros_like(r), r)
    return (r.unsqueeze(1) * a_hat) * r.unsqueeze(0)


def normalize_sparse_tensor(adj: torch.Tensor, fill_value=1):
    """/root/reference/utils.py:437-455 (the variant the reference keeps one comment away from use, :481): n self-loops
    of weight ``fill_value`` are APPENDED (existing loops stay), ``deg = scatter_add(w, row)`` -- ROW degree, summed in
    edge order in float32 like the serial CPU scatter_add --, ``deg^-1/2`` with inf -> 0, and
    ``(deg_inv_sqrt[row] * w) * deg_inv_sqrt[col]`` left to right.  Returns an (uncoalesced) torch sparse COO tensor like
    the reference.  Same kernels as ``gcn_norm`` (degree by row = ``flow="target_to_source"``)."""
    stage = _staging_device(adj, "normalize_sparse_tensor")
    if stage is not None:
        return normalize_sparse_tensor(adj.to(stage), fill_value).to(adj.device)
    edge_index, edge_weight = add_self_loops(adj._indices(), adj._values(), fill_value, adj.size(0))
    edge_index, values = gcn_norm(edge_index, edge_weight, adj.size(0), add_self_loops=False, flow="target_to_source")
    return torch.sparse_coo_tensor(edge_index, values, adj.shape)


def sparse_mx_to_torch_sparse_tensor(sparse_mx) -> torch.Tensor:
    """/root/reference/utils.py:632-639: scipy sparse matrix -> torch sparse COO float32 tensor (entries in the COO
    order scipy hands back, values rounded to float32 once, int64 indices)."""
    coo = sparse_mx.tocoo().astype("float32")
    idx = torch.stack([torch.from_numpy(coo.row.astype("int64")), torch.from_numpy(coo.col.astype("int64"))])
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data), torch.Size(coo.shape))


def to_scipy(tensor: torch.Tensor):
    """/root/reference/utils.py:651-660: dense or sparse torch tensor -> scipy CSR on the host (duplicate COO entries
    are summed by ``csr_matrix``)."""
    import scipy.sparse as sp
    if is_sparse_tensor(tensor):
        idx, val = tensor._indices(), tensor._values()
    else:
        idx = tensor.nonzero().t()
        val = tensor[idx[0], idx[1]]
    return sp.csr_matrix((val.cpu().numpy(), idx.cpu().numpy()), shape=tuple(tensor.shape))


def to_tensor(adj, device="cpu") -> torch.Tensor:
    """/root/reference/utils.py:322-341: adjacency as a torch tensor on ``device`` -- scipy sparse -> sparse COO float32,
    anything else -> dense float32 (what ``LargeScaleCondensing.py:378`` feeds ``normalize_adj_tensor``)."""
    import scipy.sparse as sp
    out = sparse_mx_to_torch_sparse_tensor(adj) if sp.issparse(adj) else torch.as_tensor(adj, dtype=torch.float32)
    return out.to(device)


def to_undirected(edge_index: torch.Tensor, edge_attr=None, num_nodes: Optional[int] = None) -> torch.Tensor:
    """PyG ``to_undirected(edge_index, edge_attr=None, num_nodes=)`` as the reference calls it
    (/root/reference/utils.py:100): both directions of every edge, duplicates removed, sorted by
    (row, col).  One device radix sort (``disco_coo_to_csr`` with SYMMETRISE | DUP_FIRST)."""
    if edge_attr is not None:
        raise NotImplementedError("the reference only symmetrises unweighted edge lists (utils.py:100)")
    _require_cuda(edge_index, "to_undirected")
    n = maybe_num_nodes(edge_index, num_nodes)
    rp, c32, _ = coo_to_csr(edge_index[0], edge_index[1], None, (n, n), dup="first", symmetrise=True, want_value=False)
    row = torch.repeat_interleave(torch.arange(n, device=edge_index.device), rp[1:] - rp[:-1])
    return torch.stack([row, c32.to(torch.int64)])


def adjacency_from_edge_index(edge_index: torch.Tensor, num_nodes: int, undirected: bool = False) -> SparseTensor:
    """``Pyg2Dpr``'s adjacency (/root/reference/utils.py:99-106): optional ``to_undirected``, then
    ``csr_matrix((ones, (row, col)), shape=(n, n))`` -- duplicate edges are SUMMED (a doubled edge
    weighs 2).  Built on the device; the result feeds ``normalize_adj_csr``."""
    _require_cuda(edge_index, "adjacency_from_edge_index")
    if undirected:
        edge_index = to_undirected(edge_index, None, num_nodes)
    rp, c32, v = coo_to_csr(edge_index[0], edge_index[1], None, (num_nodes, num_nodes), dup="sum", want_value=True)
    return SparseTensor(rowptr=rp, col=c32, value=v, sparse_sizes=(num_nodes, num_nodes), is_sorted=True)


def to_adj_t(adj) -> SparseTensor:
    """``SparseTensor(row=adj._indices()[0], col=adj._indices()[1], value=adj._values(),
    sparse_sizes=adj.size()).t()`` (/root/reference/LargeScaleCondensing.py:486)."""
    if isinstance(adj, SparseTensor):
        return adj.t()
    return SparseTensor(row=adj._indices()[0], col=adj._indices()[1], value=adj._values(),
                        sparse_sizes=tuple(adj.size())).t()
