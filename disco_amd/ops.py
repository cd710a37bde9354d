"""Python entry points of the propagation path: thin wrappers that check operands, fetch the
current HIP stream and call the C ABI (include/disco_hip.h).  No CPU fallback exists.

spmm       <- torch_sparse.matmul(adj_t, x, reduce)      sg_conv.py:92-93, gcn_conv.py:170-171
khop       <- for k in range(K): x = propagate(adj_t, x)  sg_conv.py:74-77
max_aggregate <- MessagePassing(aggr="max").propagate     LargeScaleCondensing.py:152-172
anchor_max <- feat_transform[index[c]][anchor, :L*d].max(dim=1).values   LargeScaleCondensing.py:299
(paths relative to /root/reference)
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .sparse import SparseTensor

_REDUCE = {"sum": 0, "add": 0, "mean": 1, "max": 2, "sum_acc": 3}


def _as_operand(x: torch.Tensor) -> torch.Tensor:
    if x.dtype != torch.float32:
        raise TypeError(f"disco_amd propagates float32 features, got {x.dtype}")
    if x.dim() != 2:
        raise ValueError("features must be [N, F]")
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        x = x.contiguous()
    if x.shape[1] >= 4 and (x.stride(0) % 4 or x.data_ptr() % 16) and x.shape[0] > 0:
        # rows that are not 16-byte aligned (e.g. the [N, 41] output of GCNConv's last `lin`,
        # /root/reference/models/convs_large/gcn_conv.py:157) would be gathered with scalar loads: one copy
        # of N*F floats into a padded buffer buys 128-bit gathers for all nnz*F of them
        buf = _padded_empty(x.shape[0], x.shape[1], x.device)
        buf.copy_(x)
        x = buf
    return x


def _require_cuda(adj: SparseTensor, x: torch.Tensor) -> None:
    if adj.device.type != "cuda" or x.device.type != "cuda":
        raise RuntimeError("disco_amd runs on the GPU only (adjacency on %s, features on %s); there is no CPU "
                           "fallback" % (adj.device, x.device))
    if adj.device != x.device:
        raise RuntimeError(f"adjacency on {adj.device} but features on {x.device}")


def _padded_empty(n: int, F: int, device, pad: int = 4) -> torch.Tensor:
    """[n, F] view of a buffer whose leading dimension is a multiple of `pad` floats (16-byte rows)."""
    ld = (F + pad - 1) // pad * pad
    return torch.empty((n, ld), dtype=torch.float32, device=device)[:, :F]


def _check_out(out: torch.Tensor, n: int, F: int, device, what: str) -> None:
    """The kernels write n rows of F floats at out.stride(0): a buffer of another shape, type or place would be
    overrun silently."""
    if (out.dim() != 2 or out.shape[0] != n or out.shape[1] != F or out.dtype != torch.float32 or out.device != device
            or (F > 1 and out.stride(1) != 1) or (n > 1 and out.stride(0) < F)):
        raise ValueError(f"{what}: out must be a float32 [{n}, {F}] tensor on {device} with unit column stride, got "
                         f"{tuple(out.shape)} {out.dtype} on {out.device}, strides {tuple(out.stride())}")


def _hub_workspace(adj: SparseTensor, info, F: int, lib, device):
    """Partial-row workspace of the hub kernels, kept on the adjacency and grown on demand (one allocation per
    graph instead of one per call).  Calls that overlap on different streams must not share an adjacency."""
    need = int(lib.disco_spmm_workspace_bytes(info, F))
    if not need:
        return None, 0
    ws = adj._hub_ws
    if ws is None or ws.numel() < need or ws.device != device:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        adj._hub_ws = ws
    return ws, ws.numel()


def spmm_raw(adj: SparseTensor, x: torch.Tensor, reduce: str = "sum", out: Optional[torch.Tensor] = None,
             use_schedule: bool = True) -> torch.Tensor:
    """y = reduce_e val[e] * x[col[e]] through disco_spmm_csr_f32 (no autograd)."""
    x = _as_operand(x)
    _require_cuda(adj, x)
    n, m = adj.sparse_sizes()
    if x.shape[0] < m:
        raise ValueError(f"adjacency has {m} columns but x has {x.shape[0]} rows")
    F = x.shape[1]
    if out is None:
        out = _padded_empty(n, F, x.device)
    else:
        _check_out(out, n, F, x.device, "spmm_raw")
    lib = _lib.load()
    val = adj._value
    if val is not None and val.dtype != torch.float32:
        val = val.to(torch.float32)
    with torch.cuda.device(x.device):
        stream = _lib.current_stream(x.device)
        ws_ptr, info_ref, sws, sws_bytes = None, None, None, 0
        if use_schedule and n > 0:
            ws, info = adj.schedule(feat=F)
            ws_ptr, info_ref = _lib.ptr(ws), info
            sws, sws_bytes = _hub_workspace(adj, info, F, lib, x.device)
        _lib.check(lib.disco_spmm_csr_f32(_lib.ptr(adj._rowptr), _lib.ptr(adj._col32), _lib.ptr(val), n,
                                          _lib.ptr(x), x.stride(0), _lib.ptr(out), out.stride(0), F, _REDUCE[reduce],
                                          ws_ptr, info_ref, _lib.ptr(sws), sws_bytes, stream), "disco_spmm_csr_f32")
    return out


class _SpmmFn(torch.autograd.Function):
    """Autograd for reduce in {sum, mean}: dX = A^T dY (gradient to x only, as in the reference where
    edge weights are detached, /root/reference/LargeScaleCondensing.py:309-310)."""

    @staticmethod
    def forward(ctx, x, adj, reduce):
        ctx.adj, ctx.reduce = adj, reduce
        ctx.x_rows = x.shape[0]
        return spmm_raw(adj, x, reduce)

    @staticmethod
    def backward(ctx, gy):
        adj, reduce = ctx.adj, ctx.reduce
        gy = gy.contiguous()
        if reduce == "mean":
            cnt = (adj._rowptr[1:] - adj._rowptr[:-1]).clamp(min=1).to(gy.dtype)
            gy = gy / cnt[:, None]
        gx = spmm_raw(adj.t(), gy, "sum")
        if gx.shape[0] < ctx.x_rows:
            gx = torch.cat([gx, gx.new_zeros(ctx.x_rows - gx.shape[0], gx.shape[1])])
        return gx, None, None


def spmm(adj: SparseTensor, x: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
    """``matmul(adj_t, x, reduce)``; differentiable w.r.t. x for sum/mean."""
    reduce = "sum" if reduce == "add" else reduce
    if reduce not in ("sum", "mean", "max"):
        raise ValueError(f"unsupported reduce {reduce!r}")
    if x.requires_grad and torch.is_grad_enabled():
        if reduce == "max":
            raise NotImplementedError("reduce='max' is forward-only on this path")
        return _SpmmFn.apply(x, adj, reduce)
    return spmm_raw(adj, x, reduce)


matmul = spmm


def khop(adj: SparseTensor, x: torch.Tensor, K: int, n_col_slices: int = 1, out: Optional[torch.Tensor] = None,
         scratch: Optional[torch.Tensor] = None, use_schedule: bool = True) -> torch.Tensor:
    """y = A^K x through disco_khop_csr_f32 (forward only; use repeated spmm() under autograd).

    `scratch` may be a caller-owned float32 buffer of at least ``min(K-1, 2) * n * ld`` elements."""
    x = _as_operand(x)
    _require_cuda(adj, x)
    n, m = adj.sparse_sizes()
    if n != m or x.shape[0] != n:
        raise ValueError("khop needs a square adjacency matching x")
    F = x.shape[1]
    if K < 1:
        raise ValueError("khop needs K >= 1")
    if out is None:
        out = _padded_empty(n, F, x.device)
    else:
        _check_out(out, n, F, x.device, "khop")
    if scratch is not None and (scratch.dtype != torch.float32 or scratch.device != x.device or not scratch.is_contiguous()):
        raise ValueError("khop: scratch must be a contiguous float32 buffer on the features' device")
    ldt = (F + 31) // 32 * 32
    ntmp = min(K - 1, 2)
    need = ntmp * n * ldt
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty(max(need, 1), dtype=torch.float32, device=x.device)
    t0 = scratch[: n * ldt] if ntmp >= 1 else None
    t1 = scratch[n * ldt: 2 * n * ldt] if ntmp >= 2 else None
    lib = _lib.load()
    val = adj._value
    if val is not None and val.dtype != torch.float32:      # the kernels read float32 values (as spmm_raw)
        val = val.to(torch.float32)
    with torch.cuda.device(x.device):
        stream = _lib.current_stream(x.device)
        ws_ptr, info_ref, sws, sws_bytes = None, None, None, 0
        if use_schedule and n > 0:
            ws, info = adj.schedule(feat=F)
            ws_ptr, info_ref = _lib.ptr(ws), info
            sws, sws_bytes = _hub_workspace(adj, info, F, lib, x.device)
        _lib.check(lib.disco_khop_csr_f32(_lib.ptr(adj._rowptr), _lib.ptr(adj._col32), _lib.ptr(val), n, K,
                                          _lib.ptr(x), x.stride(0), _lib.ptr(out), out.stride(0), _lib.ptr(t0),
                                          _lib.ptr(t1), ldt, F, n_col_slices, ws_ptr, info_ref, _lib.ptr(sws),
                                          sws_bytes, stream), "disco_khop_csr_f32")
    return out


def max_aggregate(adj: SparseTensor, x: torch.Tensor) -> torch.Tensor:
    """T[i] = max over the entries of CSR row i of x[col]; values ignored; empty rows give 0."""
    if adj.has_value():
        adj = adj.set_value(None)
    return spmm_raw(adj, x, "max")


def pair_gather(z: torch.Tensor, edge_index: torch.Tensor, sel: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, err_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``concat((z[edge_index[0][sel]], z[edge_index[1][sel]]), dim=1)`` in one kernel
    (/root/reference/LargeScaleCondensing.py:190-191), ``sel=None`` = every edge in order.
    ``err_flag`` (int32 device scalar, optional) is set to 1 if an index is out of range; without it the
    indices are checked here on the host side of the call (one device reduction)."""
    if not (z.is_cuda and edge_index.is_cuda):
        raise RuntimeError("disco_amd.pair_gather needs CUDA (HIP) tensors: there is no CPU fallback")
    z = _as_operand(z)
    ei = edge_index.to(torch.int64)
    src, dst = ei[0].contiguous(), ei[1].contiguous()
    n, w = z.shape
    p = int(sel.numel()) if sel is not None else int(src.numel())
    if sel is not None:
        sel = sel.to(torch.int64).contiguous()
    if out is None:
        out = torch.empty((p, 2 * w), dtype=torch.float32, device=z.device)
    assert out.shape == (p, 2 * w) and out.stride(1) == 1 and out.dtype == torch.float32
    own_flag = err_flag is None
    if own_flag:
        err_flag = torch.zeros(1, dtype=torch.int32, device=z.device)
    lib = _lib.load()
    with torch.cuda.device(z.device):
        _lib.check(lib.disco_pair_gather_f32(_lib.ptr(z), n, z.stride(0), w, _lib.ptr(sel), _lib.ptr(src), _lib.ptr(dst),
                                             int(src.numel()), p, _lib.ptr(out), out.stride(0), _lib.ptr(err_flag),
                                             _lib.current_stream(z.device)), "disco_pair_gather_f32")
    if own_flag and int(err_flag.item()):
        raise IndexError("pair_gather: edge selection or node index out of range")
    return out


def anchor_max(feat_transform: torch.Tensor, anchor: torch.Tensor, width: Optional[int] = None,
               index: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
               out_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``feat_transform[index][anchor, :width].max(dim=1).values`` in one kernel -- the anchor neighbour features of
    ``edge_construction`` (/root/reference/LargeScaleCondensing.py:293-300, _induct.py:284-294, _Sampled.py:252-262).

    anchor   int64 [n_c, k]: per synthetic node the k nearest real nodes (k = 3 in the reference), as positions in
             ``index`` when that is given (faiss searches the class's train rows), else as rows of feat_transform;
    width    the column cut ``:(aggr_layer)*d`` (default: every column);
    out / out_rows  write row i to ``out[out_rows[i]]`` (``feat_syn_neighbor[index_syn[c]] = ...``); without them a
             fresh [n_c, width] tensor is returned."""
    if not (feat_transform.is_cuda and anchor.is_cuda):
        raise RuntimeError("disco_amd.anchor_max needs CUDA (HIP) tensors: there is no CPU fallback")
    if anchor.dim() != 2:
        raise ValueError("anchor must be [n_c, k]")
    t = _as_operand(feat_transform)
    n, full = t.shape
    width = full if width is None else int(width)
    if not 0 <= width <= full:
        raise ValueError(f"width {width} outside [0, {full}]")
    anchor = anchor.to(torch.int64).contiguous()
    n_c, k = anchor.shape
    if index is not None:
        index = index.to(device=t.device, dtype=torch.int64).contiguous()
    if out_rows is not None:
        if out is None:
            raise ValueError("out_rows needs out")
        out_rows = out_rows.to(device=t.device, dtype=torch.int64).contiguous()
        if out_rows.numel() != n_c:
            raise ValueError("out_rows must name one destination row per anchor row")
    if out is None:
        out = _padded_empty(n_c, width, t.device)
    if out.dim() != 2 or out.shape[1] < width or out.stride(1) != 1 or out.dtype != torch.float32 or out.device != t.device:
        raise ValueError("out must be a float32 [rows, >= width] matrix on the same device")
    if out_rows is None and out.shape[0] < n_c:
        raise ValueError("out has fewer rows than anchor")
    if n_c == 0 or width == 0:
        return out if out_rows is not None else out[:n_c, :width]
    if k < 1:
        raise ValueError("anchor needs at least one column")
    err = torch.zeros(1, dtype=torch.int32, device=t.device)
    lib = _lib.load()
    with torch.cuda.device(t.device):
        _lib.check(lib.disco_anchor_max_f32(_lib.ptr(t), n, t.stride(0), width, _lib.ptr(index), 0 if index is None else index.numel(),
                                            _lib.ptr(anchor), n_c, k, _lib.ptr(out_rows), out.shape[0], _lib.ptr(out), out.stride(0),
                                            _lib.ptr(err), _lib.current_stream(t.device)), "disco_anchor_max_f32")
    if int(err.item()):
        raise IndexError("anchor_max: an anchor, index or destination row is out of range")
    return out if out_rows is not None else out[:n_c, :width]
