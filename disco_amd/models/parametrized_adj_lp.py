"""PGE_Edge with the reference's interface and state_dict keys
(/root/reference/models/parametrized_adj_lp.py:11-52) plus the HIP pair scorer.

``forward`` / ``inference`` on explicit pair embeddings keep the reference's behaviour with stock torch
layers (training needs autograd on a small dense MLP: out of the accelerated path, SURVEY.md section 2
row 13).  What the reference spends its time on -- scoring ALL n^2 pairs of condensed nodes in a Python
loop (LargeScaleCondensing.py:304-305) and post-processing the n x n matrix -- is ``score_all_pairs``
and ``edge_construction`` here, computed by libdisco_hip.so (fp32 MFMA for the 256 x 256 layer).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

_VARIANTS = {"transductive": 0, "inductive": 1, "sampled": 2}


class PGE_Edge(nn.Module):
    """Pair scorer MLP.  Interface contract taken from the reference (constructor arguments, ``forward`` /
    ``inference`` signatures, parameter names ``layers.<i>`` / ``bns.<i>`` so that saved
    ``pge_aggr_max_*.pt`` files load, /root/reference/LargeScaleCondensing.py:391-394): ``nlayers`` linear
    layers ``2*nfeat -> nhid -> ... -> 1`` with BatchNorm1d + ReLU after every layer but the last, sigmoid on
    the single logit."""

    def __init__(self, nfeat, nhid=256, nlayers=3, device=None, args=None):
        super().__init__()
        if nlayers < 2:
            raise ValueError("PGE_Edge needs at least an input and an output layer")
        widths = [2 * nfeat] + [nhid] * (nlayers - 1) + [1]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))
        self.bns = nn.ModuleList(nn.BatchNorm1d(nhid) for _ in range(nlayers - 1))
        self.device, self.args, self.cnt = device, args, 0
        self.reset_parameters()      # the reference re-draws the initial weights once more after construction

    def reset_parameters(self):
        for m in list(self.layers) + list(self.bns):
            m.reset_parameters()

    def logits(self, edge_embed: torch.Tensor) -> torch.Tensor:
        h = edge_embed
        for lin, bn in zip(self.layers[:-1], self.bns):
            h = F.relu(bn(lin(h)))
        return self.layers[-1](h).reshape(-1)

    def forward(self, edge_embed, inference=False):
        """Edge probability of every row ``[z_src || z_dst]`` of ``edge_embed`` (``inference`` is accepted
        for call compatibility and changes nothing, as in the reference)."""
        return torch.sigmoid(self.logits(edge_embed))

    @torch.no_grad()
    def inference(self, edge_embed):
        """Eval-mode (running BatchNorm statistics) forward; leaves the module in eval mode like the
        reference's ``inference`` does (parametrized_adj_lp.py:41-44)."""
        self.eval()
        return self.forward(edge_embed, inference=True)

    # ---- HIP path -------------------------------------------------------------------------------
    def _prepare(self, Z: torch.Tensor):
        if len(self.layers) != 3:
            raise NotImplementedError("the HIP scorer implements the 3-layer PGE_Edge every driver instantiates")
        if Z.device.type != "cuda":
            raise RuntimeError("PGE_Edge.score_all_pairs runs on the GPU only; there is no CPU fallback")
        if Z.dtype != torch.float32 or Z.dim() != 2:
            raise TypeError("Z must be float32 [n, nfeat]")
        Z = Z.contiguous() if Z.stride(1) != 1 else Z
        n, D = Z.shape
        H = self.layers[0].out_features
        if self.layers[0].in_features != 2 * D:
            raise ValueError(f"Z has {D} features but the scorer was built for nfeat={self.layers[0].in_features // 2}")
        lib = _lib.load()
        f32 = lambda t: t.detach().to(device=Z.device, dtype=torch.float32).contiguous()  # noqa: E731
        p = [f32(self.layers[0].weight), f32(self.layers[0].bias), f32(self.bns[0].weight), f32(self.bns[0].bias),
             f32(self.bns[0].running_mean), f32(self.bns[0].running_var), f32(self.layers[1].weight), f32(self.layers[1].bias),
             f32(self.bns[1].weight), f32(self.bns[1].bias), f32(self.bns[1].running_mean), f32(self.bns[1].running_var),
             f32(self.layers[2].weight), f32(self.layers[2].bias)]
        with torch.cuda.device(Z.device):
            nb = lib.disco_pairscore_workspace_bytes(n, H)
            ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device=Z.device)
            _lib.check(lib.disco_pairscore_prepare_f32(_lib.ptr(Z), n, D, Z.stride(0), H, *[_lib.ptr(t) for t in p],
                                                       float(self.bns[0].eps), _lib.ptr(ws), ws.numel(),
                                                       _lib.current_stream(Z.device)), "disco_pairscore_prepare_f32")
        return ws, n, H, p

    @torch.no_grad()
    def score_block(self, Z: torch.Tensor, rows: Tuple[int, int], cols: Optional[Tuple[int, int]] = None, logits: bool = False,
                    out: Optional[torch.Tensor] = None, _prepared=None) -> torch.Tensor:
        """S[i, j] = inference([z_i || z_j]) for i in rows, j in cols (eval-mode BatchNorm)."""
        ws, n, H, keep = _prepared or self._prepare(Z)
        i0, i1 = rows
        j0, j1 = cols if cols is not None else (0, n)
        if out is None:
            out = torch.empty((max(i1 - i0, 0), max(j1 - j0, 0)), dtype=torch.float32, device=Z.device)
        elif (tuple(out.shape) != (i1 - i0, j1 - j0) or out.dtype != torch.float32 or out.device != Z.device
              or (out.numel() and out.stride(1) != 1)):
            raise ValueError(f"out must be a float32 [{i1 - i0}, {j1 - j0}] tensor on {Z.device} with unit column stride")
        lib = _lib.load()
        with torch.cuda.device(Z.device):
            _lib.check(lib.disco_pairscore_block_f32(_lib.ptr(ws), n, H, i0, i1, j0, j1, _lib.ptr(out), out.stride(0),
                                                     0 if logits else 1, _lib.current_stream(Z.device)), "disco_pairscore_block_f32")
        return out

    @torch.no_grad()
    def score_all_pairs(self, Z: torch.Tensor, logits: bool = False) -> torch.Tensor:
        """The full n x n score matrix of LargeScaleCondensing.py:304-305 (stays on the GPU)."""
        return self.score_block(Z, (0, Z.shape[0]), None, logits)

    @torch.no_grad()
    def edge_construction(self, Z: torch.Tensor, threshold: float, variant: str = "transductive", fused: bool = True,
                          capacity: Optional[int] = None):
        """(edge_index int64 [2, E], edge_weight float32 [E]) of the condensed graph, on the GPU: scores ->
        (S + S^T)/2 -> threshold / unit diagonal in the order of `variant` -> entries in ``torch.nonzero`` order
        (/root/reference/LargeScaleCondensing.py:304-310, _induct.py:298-304, _Sampled.py:266-285).

        fused=True (default): ``disco_pairscore_edges_f32`` scores both directions of every pair, thresholds in the
        kernel and emits only the surviving edges; the n x n matrix (2.5 GB at n = 25 000) is never written or
        re-read.  fused=False is the dense path (score matrix + ``edges_from_scores``); both give identical bits."""
        if not fused:
            return self.edges_from_scores(self.score_all_pairs(Z), threshold, variant)
        v = _VARIANTS[variant]
        ws, n, H, keep = self._prepare(Z)
        lib = _lib.load()
        dev = Z.device
        cap = int(capacity) if capacity is not None else min(n * n, max(1 << 22, 64 * n))
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            while True:
                erow = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
                ecol = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
                evl = torch.empty(max(cap, 1), dtype=torch.float32, device=dev)
                _lib.check(lib.disco_pairscore_edges_f32(_lib.ptr(ws), n, H, float(threshold), _lib.ptr(erow), _lib.ptr(ecol),
                                                         _lib.ptr(evl), cap, _lib.ptr(count), _lib.current_stream(dev)),
                           "disco_pairscore_edges_f32")
                E = int(count.item())
                if E <= cap:
                    break
                cap = E                      # denser than expected: one more pass with exactly enough room
        erow, ecol, evl = erow[:E], ecol[:E], evl[:E]
        # unit diagonal: transductive sets it after thresholding, inductive before (a unit entry survives unless 1 < thr)
        if v == 0 or (v == 1 and not (1.0 < float(threshold))):
            d = torch.arange(n, device=dev)
            erow, ecol = torch.cat([erow, d]), torch.cat([ecol, d])
            evl = torch.cat([evl, torch.ones(n, dtype=torch.float32, device=dev)])
        if erow.numel():
            from ..sparse import coo_to_csr
            rp, c32, vals = coo_to_csr(erow, ecol, evl, (n, n), dup="keep")       # device radix sort by (row, col)
            rows = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
            ei, ew = torch.stack([rows, c32.to(torch.int64)]), vals
        else:
            ei, ew = torch.empty((2, 0), dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.float32, device=dev)
        if variant == "sampled":
            loops = torch.arange(n, device=dev)
            ei = torch.cat([ei, torch.stack([loops, loops])], dim=1)
            ew = torch.cat([ew, torch.ones(n, device=dev)])
        return ei, ew

    @staticmethod
    @torch.no_grad()
    def edges_from_scores(S: torch.Tensor, threshold: float, variant: str = "transductive"):
        """Symmetrise / threshold / unit diagonal / ``nonzero`` of a full score matrix (the tail of the three
        ``edge_construction`` variants), ``disco_sym_threshold_count/fill``."""
        v = _VARIANTS[variant]
        if S.dim() != 2 or S.shape[0] != S.shape[1] or S.dtype != torch.float32 or S.device.type != "cuda":
            raise ValueError("S must be a square float32 score matrix on the GPU")
        if S.numel() and S.stride(1) != 1:
            S = S.contiguous()
        Z = S
        n = S.shape[0]
        lib = _lib.load()
        dev = Z.device
        with torch.cuda.device(dev):
            s = _lib.current_stream(dev)
            cnt = torch.empty(n, dtype=torch.int64, device=dev)
            _lib.check(lib.disco_sym_threshold_count(_lib.ptr(S), n, S.stride(0), float(threshold), v, _lib.ptr(cnt), s),
                       "disco_sym_threshold_count")
            start = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            torch.cumsum(cnt, 0, out=start[1:])
            E = int(start[-1])
            ei = torch.empty((2, E), dtype=torch.int64, device=dev)
            ew = torch.empty(E, dtype=torch.float32, device=dev)
            _lib.check(lib.disco_sym_threshold_fill(_lib.ptr(S), n, S.stride(0), float(threshold), v, _lib.ptr(start), E,
                                                    _lib.ptr(ei), _lib.ptr(ew), s), "disco_sym_threshold_fill")
        if variant == "sampled":
            # add_self_loops(edge_index, edge_weight, num_nodes=n): n unit loops appended (_Sampled.py:285)
            loops = torch.arange(n, device=dev)
            ei = torch.cat([ei, torch.stack([loops, loops])], dim=1)
            ew = torch.cat([ew, torch.ones(n, device=dev)])
        return ei, ew
