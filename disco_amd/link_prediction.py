"""Pre-training of the edge-translation link predictor without leaving the device.

Mirrors ``link_prediction(pge_edge, nedges)`` of /root/reference/LargeScaleCondensing.py:146-218:
  * ``feat_transform = [X | max-aggregate(X)]``                                  (:150-173)  -> ops.max_aggregate
  * ``neg_edge_index = negative_sampling(edge_index, N, 3 E)``                   (:178)
  * per iteration: ``nedges`` random positives + ``3 nedges`` random negatives, pair embeddings
    ``[z_src | z_dst]``, ``PGE_Edge.forward`` (training-mode BatchNorm), ``BCELoss``, Adam   (:187-200)
The reference moves every batch device -> host -> device (:190-192); here the batch is written by one gather
kernel (``disco_pair_gather_f32``) into a buffer that is re-used across iterations.  The dense MLP itself
stays stock torch (rocBLAS GEMMs), as in the reference.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F

from . import ops
from .sparse import SparseTensor


def edge_translation_features(adj_t: SparseTensor, feat: torch.Tensor) -> torch.Tensor:
    """``concat((feat, MessagePassing(aggr="max").propagate(adj_t, x=feat)), dim=1)``
    (/root/reference/LargeScaleCondensing.py:152-153,172-173)."""
    return torch.cat([feat, ops.max_aggregate(adj_t, feat)], dim=1)


def anchor_neighbor_features(feat_syn: torch.Tensor, feat_transform: torch.Tensor, anchors, index, index_syn,
                             width: Optional[int] = None) -> torch.Tensor:
    """``feat_syn_transform`` of ``edge_construction`` (/root/reference/LargeScaleCondensing.py:292-300,
    _induct.py:284-295, _Sampled.py:252-262): per class c the synthetic nodes ``index_syn[c]`` take the element-wise
    maximum, over their ``neighbor`` (3) nearest real train nodes ``index[c][anchors[c]]``, of the first ``width``
    (= aggr_layer * d) columns of ``feat_transform``; the result is concatenated behind ``feat_syn``.

    anchors / index / index_syn: dicts (or sequences) keyed by class, as the reference's ``knn_class[c].search``
    result, ``index[c]`` and ``index_syn[c]``; classes missing from ``anchors`` keep zeros (``num_class_dict``).
    The k-NN search itself (faiss, CPU) is outside the path.  One ``disco_anchor_max_f32`` call per class writes
    straight into the output rows: no row-subset copy, no [n_c, 3, L*d] intermediate."""
    width = feat_transform.shape[1] if width is None else int(width)
    out = torch.zeros((feat_syn.shape[0], feat_syn.shape[1] + width), dtype=torch.float32, device=feat_syn.device)
    out[:, : feat_syn.shape[1]] = feat_syn
    neigh = out[:, feat_syn.shape[1]:]
    keys = anchors.keys() if hasattr(anchors, "keys") else range(len(anchors))
    for c in keys:
        a = torch.as_tensor(anchors[c], device=feat_syn.device)
        if a.numel() == 0:
            continue
        ops.anchor_max(feat_transform, a, width, index=torch.as_tensor(index[c]), out=neigh,
                       out_rows=torch.as_tensor(index_syn[c]))
    return out


def negative_sampling(edge_index: torch.Tensor, num_nodes: int, num_neg_samples: int,
                      generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """PyG ``negative_sampling(edge_index, num_nodes, num_neg_samples)`` [upstream, restated]
    (/root/reference/LargeScaleCondensing.py:178): uniformly random node pairs that are NOT edges
    (self pairs excluded), at most ``num_neg_samples`` of them.  Random by construction, so parity is a
    property (no returned pair is an edge; pairs are distinct), not a bit pattern.  Index bookkeeping
    in stock torch on the device."""
    dev = edge_index.device
    n = int(num_nodes)
    pos = torch.unique(edge_index[0].to(torch.int64) * n + edge_index[1].to(torch.int64))
    want = int(num_neg_samples)
    got = torch.empty(0, dtype=torch.int64, device=dev)
    for _ in range(4):                      # PyG also retries a few times and may return fewer
        k = int((want - got.numel()) * 1.2) + 16
        cand = torch.randint(0, n * n, (k,), device=dev, generator=generator)
        cand = cand[(cand // n) != (cand % n)]
        at = torch.searchsorted(pos, cand).clamp_(max=max(pos.numel() - 1, 0))
        keep = pos[at] != cand if pos.numel() else torch.ones_like(cand, dtype=torch.bool)
        got = torch.unique(torch.cat([got, cand[keep]]))
        if got.numel() >= want:
            break
    got = got[torch.randperm(got.numel(), device=dev, generator=generator)[:want]]
    return torch.stack([got // n, got % n])


def link_prediction(pge_edge, feat_transform: torch.Tensor, edge_index: torch.Tensor, nedges: int, epochs: int,
                    lr: float = 0.001, neg_edge_index: Optional[torch.Tensor] = None, num_nodes: Optional[int] = None,
                    generator: Optional[torch.Generator] = None, log_every: int = 100,
                    on_best: Optional[Callable] = None, verbose: bool = False):
    """The training loop of /root/reference/LargeScaleCondensing.py:176-214 with the batch assembled on the
    device.  Returns ``(best_accuracy, last_loss)``; ``on_best(pge_edge, accuracy)`` replaces the
    reference's ``torch.save`` of the best state_dict (:208-213)."""
    dev = feat_transform.device
    n = int(num_nodes if num_nodes is not None else feat_transform.shape[0])
    edge_index = edge_index.to(dev)
    if neg_edge_index is None:
        neg_edge_index = negative_sampling(edge_index, n, 3 * edge_index.shape[1], generator)    # :178
    neg_edge_index = neg_edge_index.to(dev)
    optimizer = torch.optim.Adam(pge_edge.parameters(), lr=lr)                                     # :148
    w = feat_transform.shape[1]
    batch = torch.empty((4 * nedges, 2 * w), dtype=torch.float32, device=dev)
    y = torch.cat([torch.ones(nedges, device=dev), torch.zeros(3 * nedges, device=dev)])          # :193
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    best_acc, loss = 0.0, None
    pge_edge.train()
    for i in range(epochs):
        index = torch.randint(0, edge_index.shape[1], (nedges,), device=dev, generator=generator)           # :188
        neg_index = torch.randint(0, neg_edge_index.shape[1], (3 * nedges,), device=dev, generator=generator)  # :189
        ops.pair_gather(feat_transform, edge_index, index, out=batch[:nedges], err_flag=err)           # :190
        ops.pair_gather(feat_transform, neg_edge_index, neg_index, out=batch[nedges:], err_flag=err)   # :191
        optimizer.zero_grad()
        pred = pge_edge.forward(batch)                                                             # :196
        loss = F.binary_cross_entropy(pred, y)                                                     # :197-198
        loss.backward()
        optimizer.step()
        if log_every and i % log_every == 0:                                                       # :202-214
            acc = float((torch.round(pred) == y).float().mean())
            if acc > best_acc:
                best_acc = acc
                if on_best is not None:
                    on_best(pge_edge, acc)
            if verbose:
                print("Acc:", acc)
    if int(err.item()):
        raise IndexError("link_prediction: an edge index pointed outside feat_transform")
    return best_acc, (float(loss.detach()) if loss is not None else None)
