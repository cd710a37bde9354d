"""Minimal ``MessagePassing`` base with the call surface the reference's convolutions rely on
(torch_geometric 2.1.0 ``MessagePassing.propagate`` as used at
/root/reference/models/convs_large/sg_conv.py:76, gcn_conv.py:159 and
/root/reference/LargeScaleCondensing.py:152-172), backed by the HIP SpMM.

``propagate(edge_index, x=..., edge_weight=None, size=None)`` accepts
  * an ``adj_t`` (``disco_amd.SparseTensor`` or a torch sparse CSR/COO tensor): out = adj_t @ x with
    the layer's ``aggr`` (fused ``message_and_aggregate`` route, sg_conv.py:92-93);
  * an int64 ``[2, E]`` tensor (+ optional float32 ``[E]`` weights): out[col] (+)= w * x[row]
    (``message`` at sg_conv.py:89-90 followed by scatter aggregation).  The edge list is turned into
    the CSR of A^T once per distinct edge_index tensor (stable in edge order) and cached.
"""
from __future__ import annotations

from typing import Optional, Union

import torch
from torch import Tensor

from ... import ops
from ...sparse import SparseTensor

Adj = Union[Tensor, SparseTensor]


def as_adj_t(edge_index) -> Optional[SparseTensor]:
    if isinstance(edge_index, SparseTensor):
        return edge_index
    if isinstance(edge_index, Tensor) and edge_index.layout == torch.sparse_csr:
        return SparseTensor.from_torch_sparse_csr_tensor(edge_index)
    if isinstance(edge_index, Tensor) and edge_index.layout == torch.sparse_coo:
        return SparseTensor.from_torch_sparse_coo_tensor(edge_index)
    return None


class _EdgeListCache:
    """CSR(A^T) structure of one ``[2, E]`` edge_index tensor: stable argsort by target."""

    def __init__(self):
        self.key = None
        self.perm = None
        self.struct: Optional[SparseTensor] = None
        self.held: Optional[Tensor] = None

    def get(self, edge_index: Tensor, n_src: int, n_dst: int):
        # The key names the tensor's memory and version.  The cache keeps the tensor itself alive (`held`): otherwise the
        # allocator may hand the same address to a NEW edge list of the same size (version 0 again) and the stale
        # structure would be served for it.
        key = (edge_index.data_ptr(), edge_index.shape[1], edge_index._version, n_src, n_dst, edge_index.device)
        if key != self.key:
            self.held = edge_index
            row, col = edge_index[0], edge_index[1]
            perm = torch.argsort(col, stable=True)
            counts = torch.bincount(col, minlength=n_dst)
            rowptr = torch.zeros(n_dst + 1, dtype=torch.int64, device=edge_index.device)
            torch.cumsum(counts, 0, out=rowptr[1:])
            self.struct = SparseTensor(rowptr=rowptr, col=row[perm], value=None, sparse_sizes=(n_dst, n_src), is_sorted=True)
            self.perm = perm
            self.key = key
        return self.struct, self.perm


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr: Optional[str] = "add", flow: str = "source_to_target", node_dim: int = -2, **kwargs):
        super().__init__()
        if aggr not in ("add", "sum", "mean", "max"):
            raise ValueError(f"unsupported aggr {aggr!r}")
        if flow != "source_to_target":
            raise NotImplementedError("only flow='source_to_target' is used by the reference")
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim
        self._edge_cache = _EdgeListCache()

    def propagate(self, edge_index: Adj, size=None, **kwargs) -> Tensor:
        x = kwargs.get("x")
        edge_weight = kwargs.get("edge_weight")
        if isinstance(x, (tuple, list)):
            x = x[0]
        adj_t = as_adj_t(edge_index)
        if adj_t is not None:
            return self.message_and_aggregate(adj_t, x)
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        n_src = x.shape[0]
        n_dst = n_src if size is None else (size[1] if size[1] is not None else n_src)
        struct, perm = self._edge_cache.get(edge_index, n_src, n_dst)
        if edge_weight is not None and edge_weight.requires_grad and torch.is_grad_enabled():
            # the reference detaches the condensed graph's weights (LargeScaleCondensing.py:309-310); a gradient
            # that would silently be dropped is an error here
            raise NotImplementedError("edge_weight.requires_grad: this path returns gradients for x only")
        w = None if edge_weight is None else edge_weight[perm]
        if struct.device.type == "cuda" and struct.size(0) > 0:
            struct.schedule(feat=x.shape[1])     # built once on the cached structure, inherited by set_value
        adj = struct.set_value(w)
        reduce = "sum" if self.aggr == "add" else self.aggr
        if reduce == "max" and w is not None:
            return ops.spmm(adj, x, "max")
        return ops.spmm(adj, x, reduce)

    def message_and_aggregate(self, adj_t: SparseTensor, x: Tensor) -> Tensor:
        if self.aggr == "max":
            # base MessagePassing: message(x_j) = x_j, edge values are ignored
            # (/root/reference/LargeScaleCondensing.py:152,172)
            return ops.max_aggregate(adj_t, x)
        return ops.spmm(adj_t, x, "sum" if self.aggr == "add" else self.aggr)
