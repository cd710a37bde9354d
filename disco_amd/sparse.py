"""CSR adjacency container with the subset of ``torch_sparse.SparseTensor`` the reference uses.

The reference hands ``SparseTensor(row=, col=, value=, sparse_sizes=).t()`` to its convolutions
(/root/reference/LargeScaleCondensing.py:153,486-488) and touches ``.to()``, ``.cuda()``,
``.cpu()``, ``.t()``, ``.storage.row()/col()/rowptr()/value()``, ``.csr()``, ``.coo()``,
``.size()``, ``.sparse_sizes()``, ``.nnz()``.  torch_sparse is not available on MI355X boxes, so
this class provides that surface and owns what the HIP kernels need: int64 ``rowptr``, int32
``col``, float32 ``value`` on the device, plus the lazily built degree-binned schedule and the
transposed structure used by backward.

Index bookkeeping (sort, cumsum, bincount) is done with stock torch ops: it is layout plumbing,
not the accelerated path.  All arithmetic on the path goes through libdisco_hip.so.
"""
from __future__ import annotations

from typing import Optional, Tuple

import math
import weakref

import torch

from . import _lib


WIN_COLS = 128           # column block width of the windowed launches (libdisco_hip default)
WINDOW_BYTES = 5632 << 10   # 5.5 MiB: flat optimum 5-6 MiB on Reddit (profiles/r1/tune_window_fine.log)


_SWEEP_FIELDS = ("sweep_ws", "sweep_windows", "sweep_window_rows", "sweep_count_padded", "sweep_n_cols")
WIN_HUB_THRESHOLD, WIN_HUB_PIECE = 8192, 4096   # hub split used with windowed launches (tune_hub_windowed.log)


def sweep_window_rows(feat: int, window_bytes: int = WINDOW_BYTES, cap: int = WIN_COLS) -> int:
    """Source rows per L2 window for feature width `feat`: the kernels cut wide features into balanced
    column blocks of at most `cap` floats (multiples of 32); one window of one block takes `window_bytes`."""
    nblk = max(1, (feat + cap - 1) // cap)
    blk = max(32, ((feat + nblk - 1) // nblk + 31) // 32 * 32)
    return max(64, window_bytes // (4 * blk))


def coo_to_csr(row: torch.Tensor, col: torch.Tensor, value: Optional[torch.Tensor], sparse_sizes: Tuple[int, int],
               dup: str = "keep", symmetrise: bool = False, want_value: Optional[bool] = None):
    """(rowptr int64, col int32, value float32 | None) from raw COO edges with ``disco_coo_to_csr``
    (device radix sort; replaces scipy ``csr_matrix`` / PyG ``to_undirected`` / the torch_sparse
    constructor's ordering: /root/reference/utils.py:99-106, LargeScaleCondensing.py:486).
    dup: "keep" (stable order), "sum" (float64 fold, csr_matrix), "first" (coalesce of unit weights)."""
    assert row.is_cuda and col.is_cuda, "coo_to_csr runs on the GPU"
    lib = _lib.load()
    flags = {"keep": _lib.COO_DUP_KEEP, "sum": _lib.COO_DUP_SUM, "first": _lib.COO_DUP_FIRST}[dup]
    if symmetrise:
        flags |= _lib.COO_SYMMETRISE
    row = row.to(torch.int64).contiguous()
    col = col.to(torch.int64).contiguous()
    if value is not None and value.dtype != torch.float32:
        raise TypeError("coo_to_csr carries float32 values")
    if value is not None:
        value = value.contiguous()           # kept alive in this frame until the call below has been enqueued
    want_value = (value is not None) if want_value is None else want_value
    e, (m, n) = int(row.numel()), sparse_sizes
    cap = 2 * e if symmetrise else e
    dev = row.device
    with torch.cuda.device(dev):
        nb = lib.disco_coo_to_csr_workspace_bytes(e, flags, int(want_value))
        ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device=dev)
        rowptr = torch.empty(m + 1, dtype=torch.int64, device=dev)
        col_out = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        val_out = torch.empty(max(cap, 1), dtype=torch.float32, device=dev) if want_value else None
        import ctypes
        nnz = ctypes.c_int64(0)
        _lib.check(lib.disco_coo_to_csr(_lib.ptr(row), _lib.ptr(col), _lib.ptr(value),
                                        e, m, n, flags, _lib.ptr(rowptr), _lib.ptr(col_out), _lib.ptr(val_out),
                                        ctypes.byref(nnz), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)),
                   "disco_coo_to_csr")
    k = int(nnz.value)
    if k < cap // 2:       # do not pin a mostly empty allocation
        return rowptr, col_out[:k].clone(), None if val_out is None else val_out[:k].clone()
    return rowptr, col_out[:k], None if val_out is None else val_out[:k]


class _Storage:
    """``adj.storage`` facade: ``row()``, ``col()``, ``rowptr()``, ``value()``."""

    def __init__(self, owner: "SparseTensor"):
        self._o = owner

    def row(self) -> torch.Tensor:
        return self._o._row()

    def col(self) -> torch.Tensor:
        return self._o._col64()

    def rowptr(self) -> torch.Tensor:
        return self._o._rowptr

    def value(self) -> Optional[torch.Tensor]:
        return self._o._value

    def sparse_sizes(self) -> Tuple[int, int]:
        return self._o._sizes


class _StructCache:
    """Caches that depend on the (immutable) structure only and are therefore shared by every ``set_value`` copy:
    the transposed structure and the permutation that carries values into it."""
    __slots__ = ("t_struct", "t_perm", "sched", "sweep")

    def __init__(self):
        self.t_struct: Optional["SparseTensor"] = None
        self.t_perm: Optional[torch.Tensor] = None
        self.sched = {}     # (hub_threshold, hub_piece) -> (workspace tensor, ScheduleInfo template without window table)
        self.sweep = {}     # ((hub_threshold, hub_piece), window_rows) -> (workspace tensor | None, window fields)


class SparseTensor:
    def __init__(self, row: Optional[torch.Tensor] = None, rowptr: Optional[torch.Tensor] = None,
                 col: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None,
                 sparse_sizes: Optional[Tuple[int, int]] = None, is_sorted: bool = False):
        assert col is not None and (row is not None or rowptr is not None)
        device = col.device
        if sparse_sizes is None:
            m = int(row.max()) + 1 if row is not None and row.numel() else (rowptr.numel() - 1 if rowptr is not None else 0)
            n = int(col.max()) + 1 if col.numel() else 0
            sparse_sizes = (m, n)
        sparse_sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        if rowptr is None and not is_sorted and col.is_cuda and row.numel() and (value is None or value.dtype == torch.float32):
            # device build: stable radix sort, duplicates kept (torch_sparse.SparseTensor construction)
            rowptr, col, value = coo_to_csr(row, col, value, sparse_sizes, dup="keep")
            is_sorted = True
        if rowptr is None:
            row = row.to(torch.int64)
            col = col.to(torch.int64)
            if not is_sorted and row.numel():
                # row-major order, duplicates kept -- torch_sparse.SparseTensor construction
                key = row * sparse_sizes[1] + col
                perm = torch.argsort(key, stable=True)
                row, col = row[perm], col[perm]
                if value is not None:
                    value = value[perm]
            counts = torch.bincount(row, minlength=sparse_sizes[0])
            rowptr = torch.zeros(sparse_sizes[0] + 1, dtype=torch.int64, device=device)
            torch.cumsum(counts, 0, out=rowptr[1:])
        if col.numel() and (int(col.max()) >= sparse_sizes[1] or int(col.min()) < 0):
            raise ValueError("column index outside sparse_sizes")  # the kernels trust col[] (checked once here)
        if rowptr.numel() != sparse_sizes[0] + 1 or int(rowptr[-1]) != col.numel():
            raise ValueError("rowptr inconsistent with col / sparse_sizes")
        self._rowptr = rowptr.to(torch.int64).contiguous()
        self._col32 = col.to(torch.int32).contiguous()
        self._value = None if value is None else value.contiguous()
        self._sizes = sparse_sizes
        self._init_caches()

    def _init_caches(self):
        self._sched = None          # (ws tensor [immutable, shared by set_value copies], ScheduleInfo [own copy])
        self._sched_key = None
        self._sched_auto = False
        self._sweep = None          # (window_rows, workspace tensor or None)
        self._sweep_force: Optional[bool] = None   # sticky schedule(sweep=True/False) choice
        self._window_bytes: Optional[int] = None   # sticky schedule(window_bytes=) override
        self._hub_ws: Optional[torch.Tensor] = None    # partial-row workspace of the hub kernels (grown on demand)
        self._row_groups: Optional[torch.Tensor] = None   # locality groups: major key of the row order
        self._t: Optional["SparseTensor"] = None      # the transposed matrix once built (strong: it is derived from this one)
        self._t_of = None                              # weakref to the matrix this one was transposed FROM (see t())
        self._on_device = {}        # host-resident object: its device copies, one per GPU (see to())
        self._shared = _StructCache()

    @classmethod
    def _from_parts(cls, rowptr: torch.Tensor, col32: torch.Tensor, value: Optional[torch.Tensor],
                    sparse_sizes: Tuple[int, int]) -> "SparseTensor":
        """Wrap already validated CSR arrays (int64 rowptr, int32 col) without any device read-back."""
        out = cls.__new__(cls)
        out._rowptr, out._col32 = rowptr, col32
        out._value = None if value is None else value.contiguous()
        out._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        out._init_caches()
        return out

    # ---- constructors ---------------------------------------------------------------------
    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor] = None,
                        sparse_sizes: Optional[Tuple[int, int]] = None, is_sorted: bool = False) -> "SparseTensor":
        return cls(row=edge_index[0], col=edge_index[1], value=edge_attr, sparse_sizes=sparse_sizes, is_sorted=is_sorted)

    @classmethod
    def from_torch_sparse_coo_tensor(cls, mat: torch.Tensor, has_value: bool = True) -> "SparseTensor":
        mat = mat.coalesce()
        idx = mat.indices()
        return cls(row=idx[0], col=idx[1], value=mat.values() if has_value else None,
                   sparse_sizes=tuple(mat.shape[:2]), is_sorted=True)

    @classmethod
    def from_torch_sparse_csr_tensor(cls, mat: torch.Tensor, has_value: bool = True) -> "SparseTensor":
        return cls(rowptr=mat.crow_indices(), col=mat.col_indices(), value=mat.values() if has_value else None,
                   sparse_sizes=tuple(mat.shape[:2]), is_sorted=True)

    @classmethod
    def from_scipy(cls, mat, has_value: bool = True) -> "SparseTensor":
        mat = mat.tocsr()
        mat.sort_indices()
        return cls(rowptr=torch.from_numpy(mat.indptr.astype("int64")), col=torch.from_numpy(mat.indices.astype("int64")),
                   value=torch.from_numpy(mat.data.astype("float32")) if has_value else None,
                   sparse_sizes=mat.shape, is_sorted=True)

    # ---- accessors --------------------------------------------------------------------------
    @property
    def storage(self) -> _Storage:
        """``adj.storage`` facade, made on demand: an attribute pointing back at its owner would be a reference cycle, and a
        SparseTensor in a cycle keeps its device arrays until Python's cycle collector runs (every ``set_value`` copy a
        propagation makes would linger)."""
        return _Storage(self)

    @property
    def device(self) -> torch.device:
        return self._col32.device

    def _row(self) -> torch.Tensor:
        n = self._sizes[0]
        return torch.repeat_interleave(torch.arange(n, device=self.device), self._rowptr[1:] - self._rowptr[:-1])

    def _col64(self) -> torch.Tensor:
        return self._col32.to(torch.int64)

    def csr(self):
        return self._rowptr, self._col64(), self._value

    def coo(self):
        return self._row(), self._col64(), self._value

    def size(self, dim: int) -> int:
        return self._sizes[dim]

    def sizes(self):
        return list(self._sizes)

    def sparse_sizes(self) -> Tuple[int, int]:
        return self._sizes

    def sparse_size(self, dim: int) -> int:
        return self._sizes[dim]

    def nnz(self) -> int:
        return self._col32.numel()

    def has_value(self) -> bool:
        return self._value is not None

    def set_value(self, value: Optional[torch.Tensor], layout: Optional[str] = None) -> "SparseTensor":
        """Same structure, other values.  The copy gets its OWN host-side schedule info (the struct is mutated
        when a window table is attached or detached); the device workspaces it points to are immutable once
        built and stay referenced by every object whose info points at them."""
        if value is not None and value.numel() != self.nnz():
            raise ValueError("value must have one entry per stored element")
        out = SparseTensor._from_parts(self._rowptr, self._col32, value, self._sizes)
        if self._sched is not None:
            ws, info = self._sched
            out._sched = (ws, _lib.ScheduleInfo.from_buffer_copy(info))
        out._sched_key, out._sched_auto = self._sched_key, self._sched_auto
        out._sweep, out._sweep_force, out._window_bytes = self._sweep, self._sweep_force, self._window_bytes
        out._shared = self._shared
        out._row_groups = self._row_groups
        out._xcd_run = getattr(self, "_xcd_run", 0)
        return out

    def fill_value(self, fill: float, dtype=torch.float32) -> "SparseTensor":
        return self.set_value(torch.full((self.nnz(),), fill, dtype=dtype, device=self.device))

    def to_torch_sparse_coo_tensor(self) -> torch.Tensor:
        row, col, val = self.coo()
        if val is None:
            val = torch.ones(self.nnz(), device=self.device)
        return torch.sparse_coo_tensor(torch.stack([row, col]), val, self._sizes)

    def to_dense(self) -> torch.Tensor:
        return self.to_torch_sparse_coo_tensor().to_dense()

    # ---- movement -----------------------------------------------------------------------------
    def to(self, device, non_blocking: bool = False) -> "SparseTensor":
        device = torch.device(device) if not isinstance(device, torch.device) else device
        if device == self.device or (device.type == "cuda" and device.index is None and self.device.type == "cuda"):
            return self
        # The reference keeps `adj` on the host and writes `adj.to(device)` in front of each of its 31 evaluations
        # (/root/reference/LargeScaleCondensing.py:344): the first call uploads, later calls return the SAME device object,
        # with its schedule and transpose already built (the graph stays resident).  The structure of a
        # SparseTensor is immutable (set_value makes a new object), so the copy cannot go stale.
        host_to_gpu = self.device.type == "cpu" and device.type == "cuda"
        key = device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else -1)
        if host_to_gpu and key in self._on_device:
            return self._on_device[key]
        out = SparseTensor(rowptr=self._rowptr.to(device, non_blocking=non_blocking),
                           col=self._col32.to(device, non_blocking=non_blocking),
                           value=None if self._value is None else self._value.to(device, non_blocking=non_blocking),
                           sparse_sizes=self._sizes, is_sorted=True)
        if host_to_gpu:
            self._on_device[key] = out
        return out

    def cuda(self) -> "SparseTensor":
        return self.to(torch.device("cuda", torch.cuda.current_device()))

    def cpu(self) -> "SparseTensor":
        return self.to(torch.device("cpu"))

    # ---- transpose ------------------------------------------------------------------------------
    def t(self) -> "SparseTensor":
        """Transposed matrix as CSR (row-major sorted, stable in the original entry order).  The transposed
        structure and the value permutation are built once per structure and shared by ``set_value`` copies
        (the backward of every training step on the condensed graph re-uses them)."""
        if self._t is None and self._t_of is not None:
            origin = self._t_of()
            if origin is not None:
                return origin
        if self._t is None:
            sh = self._shared
            if sh.t_struct is None:
                sizes_t = (self._sizes[1], self._sizes[0])
                nnz = self.nnz()
                row, col = self._row(), self._col64()
                if self._col32.is_cuda and 0 < nnz < 2 ** 31:
                    # the stable device sort carries a 32-bit payload unchanged: send the entry index through it
                    ids = torch.arange(nnz, dtype=torch.int32, device=self.device).view(torch.float32)
                    rp, c32, pv = coo_to_csr(col, row, ids, sizes_t, dup="keep")
                    perm = pv.view(torch.int32)
                else:
                    perm = torch.argsort(col * max(self._sizes[0], 1) + row, stable=True)
                    counts = torch.bincount(col, minlength=sizes_t[0])
                    rp = torch.zeros(sizes_t[0] + 1, dtype=torch.int64, device=self.device)
                    torch.cumsum(counts, 0, out=rp[1:])
                    c32 = row[perm].to(torch.int32)
                sh.t_struct, sh.t_perm = SparseTensor._from_parts(rp, c32.contiguous(), None, sizes_t), perm
            out = sh.t_struct.set_value(None if self._value is None else torch.index_select(self._value, 0, sh.t_perm))
            # a.t().t() is a -- but through a WEAK back-pointer: the reference writes `SparseTensor(...).t()` and drops the
            # untransposed object at once (LargeScaleCondensing.py:486); a strong pair would be a reference cycle that keeps
            # both CSR copies (2 x 13.8 GB at papers100M size) alive until Python's cycle collector happens to run
            out._t_of = weakref.ref(self)
            self._t = out
        return self._t

    # ---- device-side schedule -------------------------------------------------------------------
    def schedule(self, hub_threshold: int = 0, hub_piece: int = 0, feat: Optional[int] = None, sweep: Optional[bool] = None,
                 window_bytes: Optional[int] = None):
        """Degree-binned schedule (device workspace + host info), built once per structure.

        With ``feat`` (the feature width about to be propagated) the L2-blocked sweep structure is
        attached as well when the graph re-uses source rows enough for it to pay (``sweep=None``: automatic, or
        the last explicit choice; True / False force it and STAY in force for later ``schedule(feat=...)`` calls
        such as the ones ``ops.spmm_raw`` makes).  ``window_bytes`` overrides the window size (sticky as well; tests)."""
        lib = _lib.load()
        if sweep is None:
            sweep = self._sweep_force
        else:
            self._sweep_force = sweep
        if window_bytes is None:
            window_bytes = self._window_bytes
        else:
            self._window_bytes = window_bytes
        if hub_threshold == 0 and hub_piece == 0:
            if self._sched is not None:
                hub_threshold, hub_piece = self._sched_key      # keep the choice made first (automatic or explicit)
            elif feat is not None and sweep is not False and (sweep or self.sweep_pays(feat)):
                # windowed launches visit a row once per window, so only much longer rows need splitting
                hub_threshold, hub_piece = WIN_HUB_THRESHOLD, WIN_HUB_PIECE
            auto = True
        else:
            auto = False
        key = (hub_threshold, hub_piece)
        sh = self._shared
        if self._sched is None or self._sched_key != key:
            assert self.device.type == "cuda", "schedules live on the GPU"
            if key not in sh.sched:
                n, nnz = self._sizes[0], self.nnz()
                with torch.cuda.device(self.device):
                    info = _lib.ScheduleInfo()
                    if self._row_groups is not None:
                        nbytes = lib.disco_schedule_grouped_workspace_bytes(n, nnz, hub_threshold, hub_piece)
                        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
                        _lib.check(lib.disco_schedule_build_grouped(_lib.ptr(self._rowptr), n, nnz, hub_threshold, hub_piece,
                                                                    _lib.ptr(self._row_groups), _lib.ptr(ws), ws.numel(), info,
                                                                    _lib.current_stream(self.device)),
                                   "disco_schedule_build_grouped")
                        if getattr(self, "_xcd_run", 0):
                            info.xcd_run = self._xcd_run
                    else:
                        nbytes = lib.disco_schedule_workspace_bytes(n, nnz, hub_threshold, hub_piece)
                        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
                        _lib.check(lib.disco_schedule_build(_lib.ptr(self._rowptr), n, nnz, hub_threshold, hub_piece,
                                                            _lib.ptr(ws), ws.numel(), info, _lib.current_stream(self.device)),
                                   "disco_schedule_build")
                    # keep only the part the SpMM kernels read
                    keep = int(info.off_multi_first + 8 * (info.n_multi_rows + 2))
                    keep = min(max(keep, 256), ws.numel())
                    if keep < ws.numel() // 2:
                        ws = ws[:keep].clone()
                info.n_cols = self._sizes[1]      # every col[] entry is < sizes[1] (checked at construction)
                sh.sched[key] = (ws, info)
            ws, tmpl = sh.sched[key]
            # own copy of the host struct: attaching / detaching a window table mutates it
            self._sched, self._sched_key, self._sched_auto = (ws, _lib.ScheduleInfo.from_buffer_copy(tmpl)), key, auto
            self._sweep = None
        ws, info = self._sched
        if feat is not None and sweep is not False:
            window_rows = sweep_window_rows(feat, window_bytes) if window_bytes else sweep_window_rows(feat)
            built = self._sweep
            if built is None or built[0] != window_rows:
                want = sweep if sweep is not None else self.sweep_pays(feat)
                self._sweep = (window_rows, None)
                info.sweep_ws, info.sweep_windows = None, 0
                if want and self._sizes[0] > 0:
                    skey = (key, window_rows)
                    if skey not in sh.sweep:
                        sw, fields = None, None
                        with torch.cuda.device(self.device):
                            nb = lib.disco_sweep_workspace_bytes(info, self._sizes[1], window_rows)
                            if nb:
                                sw = torch.empty(int(nb), dtype=torch.uint8, device=self.device)
                                rc = lib.disco_sweep_build(_lib.ptr(self._rowptr), _lib.ptr(self._col32), _lib.ptr(ws), info,
                                                           self._sizes[1], window_rows, _lib.ptr(sw), sw.numel(),
                                                           _lib.current_stream(self.device))
                                if rc == -2:      # columns not sorted inside the rows: plain kernel stays in use
                                    sw = None
                                else:
                                    _lib.check(rc, "disco_sweep_build")
                                    fields = tuple(getattr(info, f) for f in _SWEEP_FIELDS)
                        sh.sweep[skey] = (sw, fields)
                    sw, fields = sh.sweep[skey]
                    if sw is not None:
                        for f, v in zip(_SWEEP_FIELDS, fields):
                            setattr(info, f, v)
                    # the object keeps the table alive as long as its info points at it
                    self._sweep = (window_rows, sw)
        elif sweep is False:
            info.sweep_ws, info.sweep_windows = None, 0
            self._sweep = None
        return ws, info

    # ---- locality-aware row order ----------------------------------------------------------------------------
    def locality_groups(self, sweeps: int = 8, max_group: int = 0) -> torch.Tensor:
        """Community label per node from ``disco_locality_labels`` (device-side label propagation, square graphs).
        ``max_group`` > 0 bounds the size of a group (choose it so that a group's rows of x fit one XCD's L2)."""
        assert self.device.type == "cuda" and self._sizes[0] == self._sizes[1], "locality pass: square graph on the GPU"
        lib = _lib.load()
        n = self._sizes[0]
        labels = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            ws = torch.empty(max(int(lib.disco_locality_workspace_bytes(n)), 256), dtype=torch.uint8, device=self.device)
            _lib.check(lib.disco_locality_labels(_lib.ptr(self._rowptr), _lib.ptr(self._col32), n, int(sweeps), int(max_group), _lib.ptr(labels),
                                                 _lib.ptr(ws), ws.numel(), _lib.current_stream(self.device)), "disco_locality_labels")
        return labels[:n]

    def set_row_groups(self, groups: Optional[torch.Tensor]) -> None:
        """Process rows group by group (``disco_schedule_build_grouped``); None returns to the degree-binned order.
        Rows below the hub threshold keep every bit: only the order in which rows are processed changes.  (Hub rows, too,
        up to F = 160; wider tables are walked in 128-column blocks here, which changes the association of a hub piece's
        sum -- 1 ulp)"""
        if groups is not None:
            assert groups.numel() == self._sizes[0] and groups.device == self.device
            groups = groups.to(torch.int32).contiguous()
        self._row_groups = groups
        # run length of the XCD-aware workgroup order: the size of the group a typical row lives in (sum size^2 / sum size),
        # as a power of two in [256, 8192] rows
        self._xcd_run = 0
        if groups is not None and groups.numel():
            sizes = torch.bincount(groups.long()).double()
            typical = float((sizes * sizes).sum() / sizes.sum())
            self._xcd_run = int(min(8192, max(256, 2 ** round(math.log2(max(typical, 1.0))))))
        self._sched, self._sweep = None, None
        old, self._shared = self._shared, _StructCache()          # schedules are per row order; the transpose is not
        self._shared.t_struct, self._shared.t_perm = old.t_struct, old.t_perm

    def sweep_pays(self, feat: int = WIN_COLS) -> bool:
        """L2-blocked (windowed) propagation re-reads and re-writes the output block once per window, so it
        pays when a row has many entries per window: mean row length >= 8 x number of windows.
        (Reddit: 492 entries/row, 19 windows -> yes; products / arxiv / papers100M -> no.)
        DISCO_SWEEP=0 disables it, DISCO_SWEEP=1 forces it."""
        import os
        env = os.environ.get("DISCO_SWEEP")
        if env is not None:
            return env == "1"
        n, m = self._sizes
        if n == 0 or m == 0 or self.nnz() >= 2 ** 32 - 1:
            return False
        windows = (m + sweep_window_rows(feat) - 1) // sweep_window_rows(feat)
        # windows no row has entries in cost next to nothing (a visit without entries leaves the row alone): the halo block
        # over an all-gathered table never touches the rank's own shard (dist.RowPartition sets the fraction)
        busy = max(1.0, windows * float(self.__dict__.get("_active_col_frac", 1.0)))
        if not (windows >= 2 and self.nnz() / n >= 8.0 * busy):
            return False
        # ... and only when a row's entries are spread over the windows.  On a graph whose ids already follow its
        # communities a row has most of its entries in ONE window: a launch per window then leaves most lane groups
        # with nothing to do (Reddit-shaped, communities in id order: 97 ms windowed vs 45 ms plain,
        # profiles/r2/reddit_community_modes.log).
        return self.window_concentration(sweep_window_rows(feat)) < 0.25

    def window_concentration(self, window_rows: int, sample: int = 2048) -> float:
        """Share of the entries of (evenly sampled) rows that fall into the row's fullest window of ``window_rows``
        source rows: about 1/windows + noise when ids carry no locality, about the share of intra-community entries
        when they do.  Deterministic; cached per window size."""
        cache = self.__dict__.setdefault("_conc", {})
        if window_rows in cache:
            return cache[window_rows]
        n, m = self._sizes
        windows = (m + window_rows - 1) // window_rows
        rows = torch.unique(torch.linspace(0, n - 1, min(sample, n), device=self.device).long())
        b, e = self._rowptr[rows], self._rowptr[rows + 1]
        cnt = e - b
        total = int(cnt.sum())
        if total == 0 or windows < 2:
            cache[window_rows] = 0.0
            return 0.0
        owner = torch.repeat_interleave(torch.arange(rows.numel(), device=self.device), cnt)
        pos = torch.arange(total, device=self.device) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt) + b[owner]
        win = (self._col32[pos].long() // window_rows)
        hist = torch.zeros(rows.numel() * windows, dtype=torch.int64, device=self.device)
        hist.index_add_(0, owner * windows + win, torch.ones_like(win))
        conc = float(hist.view(rows.numel(), windows).max(dim=1).values.sum()) / total
        cache[window_rows] = conc
        return conc

    def __repr__(self) -> str:
        return f"SparseTensor(sizes={self._sizes}, nnz={self.nnz()}, has_value={self.has_value()}, device={self.device})"
