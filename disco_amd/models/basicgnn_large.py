"""Callers of the propagation path with the reference's interface
(/root/reference/models/basicgnn_large.py:27-512): ``BasicGNN`` and ``GCN / GraphSAGE / GIN / SGC / JKNet``.

Only what sits between the drivers and the convolutions is provided: the layer loop
(``forward`` :161-207), ``predict`` (:222-233) and the loader-driven ``forward_sampler`` / ``inference`` /
``large_inference`` (:209-220, :235-318), the last three written from their interface contract (loader protocol,
slice-after-propagate, where ``lin`` applies) around one shared per-depth step.  Convolutions are the HIP-backed ones of ``disco_amd.models.convs_large``; dense layers,
BatchNorm, dropout and log_softmax are stock torch (already rocBLAS / MIOpen).  state_dict keys follow
the reference (``convs.{i}.lin.weight``, ``convs.{i}.bias``, ``norms.{i}.*``, ``lin.*``).

On MI355X the graph stays resident: ``predict(x, adj)`` with a device-resident ``adj`` re-uploads nothing
(the reference ships CSR + X to the GPU for each of its 31 evaluations, LargeScaleCondensing.py:344),
and ``full_inference`` replaces the sampler round trips of ``large_inference`` by full-graph propagation.
"""
from __future__ import annotations

import copy
from typing import Any, Callable, Dict, List, Optional, Union

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn import Linear, ModuleList

from .convs_large import GCNConv, GINConv, SAGEConv, SGConv

_ACTS = {"relu": torch.nn.ReLU, "elu": torch.nn.ELU, "leaky_relu": torch.nn.LeakyReLU, "tanh": torch.nn.Tanh,
         "sigmoid": torch.nn.Sigmoid, "gelu": torch.nn.GELU, "prelu": torch.nn.PReLU}


def activation_resolver(act, **kwargs):
    if act is None or callable(act):
        return act
    return _ACTS[str(act).lower()](**kwargs)


def normalization_resolver(norm, channels: int, **kwargs):
    if callable(norm) and not isinstance(norm, str):
        return norm
    name = str(norm).lower()
    if name in ("batchnorm", "batch_norm", "batchnorm1d"):
        return torch.nn.BatchNorm1d(channels, **kwargs)
    if name in ("layernorm", "layer_norm"):
        return torch.nn.LayerNorm(channels, **kwargs)
    raise ValueError(f"unsupported norm {norm!r}")


class MLP(torch.nn.Module):
    """Plain MLP used inside GINConv (``lins`` / ``norms`` like torch_geometric.nn.MLP; last layer plain)."""

    def __init__(self, channel_list: List[int], act="relu", act_first: bool = False, norm=None, norm_kwargs=None):
        super().__init__()
        self.act = activation_resolver(act)
        self.act_first = act_first
        self.lins = ModuleList(Linear(a, b) for a, b in zip(channel_list[:-1], channel_list[1:]))
        self.norms = ModuleList()
        for c in channel_list[1:-1]:
            self.norms.append(normalization_resolver(norm, c, **(norm_kwargs or {})) if norm is not None else torch.nn.Identity())

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        for n in self.norms:
            if hasattr(n, "reset_parameters"):
                n.reset_parameters()

    def forward(self, x: Tensor) -> Tensor:
        for i, lin in enumerate(self.lins):
            x = lin(x)
            if i < len(self.lins) - 1:
                if self.act is not None and self.act_first:
                    x = self.act(x)
                x = self.norms[i](x)
                if self.act is not None and not self.act_first:
                    x = self.act(x)
        return x


class JumpingKnowledge(torch.nn.Module):
    def __init__(self, mode: str, channels: int, num_layers: int):
        super().__init__()
        if mode not in ("cat", "max"):
            raise NotImplementedError("jk modes 'cat' and 'max' are supported ('lstm' is unused by the reference drivers)")
        self.mode = mode

    def reset_parameters(self):
        pass

    def forward(self, xs: List[Tensor]) -> Tensor:
        if self.mode == "cat":
            return torch.cat(xs, dim=-1)
        return torch.stack(xs, dim=-1).max(dim=-1)[0]


class BasicGNN(torch.nn.Module):
    supports_edge_weight = True
    supports_edge_attr = False

    def __init__(self, nfeat: int, nhid: int, nlayers: int, nclass: Optional[int] = None, dropout: float = 0.0,
                 act: Union[str, Callable, None] = "relu", act_first: bool = False, act_kwargs: Optional[Dict[str, Any]] = None,
                 norm: Union[str, Callable, None] = None, norm_kwargs: Optional[Dict[str, Any]] = None,
                 jk: Optional[str] = None, sgc: bool = False, **kwargs):
        super().__init__()
        self.nfeat, self.nhid = nfeat, nhid
        self.temp_layers = nlayers                      # basicgnn_large.py:80-85: SGC is ONE conv with K = nlayers
        self.nlayers = 1 if sgc else nlayers
        nlayers = self.nlayers
        self.dropout = dropout
        self.act = activation_resolver(act, **(act_kwargs or {}))
        self.jk_mode = jk
        self.act_first = act_first
        self.norm = norm if isinstance(norm, str) else None
        self.norm_kwargs = norm_kwargs
        self.sgc = sgc
        self.nclass = nclass if nclass is not None else nhid

        self.convs = ModuleList()
        if nlayers > 1:
            self.convs.append(self.init_conv(nfeat, nhid, **kwargs))
            nfeat = nhid
        for _ in range(nlayers - 2):
            self.convs.append(self.init_conv(nfeat, nhid, **kwargs))
            nfeat = nhid
        if nclass is not None and jk is None:
            self._is_conv_to_out = True
            self.convs.append(self.init_conv(nfeat, nclass, **kwargs))
        else:
            self.convs.append(self.init_conv(nfeat, nhid, **kwargs))

        self.norms = None
        if norm is not None and not sgc:
            layer = normalization_resolver(norm, nhid, **(norm_kwargs or {}))
            self.norms = ModuleList(copy.deepcopy(layer) for _ in range(nlayers - 1))
            if jk is not None:
                self.norms.append(copy.deepcopy(layer))
        if jk is not None and jk != "last":
            self.jk = JumpingKnowledge(jk, nhid, nlayers)
        if jk is not None:
            self.lin = Linear(nlayers * nhid if jk == "cat" else nhid, self.nclass)

    def init_conv(self, nfeat, nclass, **kwargs):
        raise NotImplementedError

    def initialize(self):
        for conv in self.convs:
            conv.reset_parameters()
        for norm in self.norms or []:
            norm.reset_parameters()
        if hasattr(self, "jk"):
            self.jk.reset_parameters()
        if hasattr(self, "lin"):
            self.lin.reset_parameters()

    def _post(self, x: Tensor, i: int) -> Tensor:
        if self.act is not None and self.act_first:
            x = self.act(x)
        if self.norms is not None:
            x = self.norms[i](x)
        if self.act is not None and not self.act_first:
            x = self.act(x)
        return x

    def forward(self, x: Tensor, edge_index, *, edge_weight: Optional[Tensor] = None, edge_attr=None) -> Tensor:
        # basicgnn_large.py:161-207
        xs: List[Tensor] = []
        for i in range(self.nlayers):
            x = self.convs[i](x, edge_index, edge_weight=edge_weight)
            if i == self.nlayers - 1 and self.jk_mode is None:
                break
            x = self._post(x, i)
            x = F.dropout(x, p=self.dropout, training=self.training)
            if hasattr(self, "jk"):
                xs.append(x)
        x = self.jk(xs) if hasattr(self, "jk") else x
        x = self.lin(x) if hasattr(self, "lin") else x
        return F.log_softmax(x, dim=1)

    def forward_sampler(self, x, adjs):
        """Training-time forward over sampled neighbourhoods (reference interface: basicgnn_large.py:209-220).
        ``adjs`` holds one hop per entry, outermost hop first; a hop is an ``(adj_t, e_id, size)`` triple whose first
        member is the bipartite adjacency (targets x sources) that the convolution of that depth consumes.  SGC has a
        single K-hop convolution which walks the hops itself.  Hidden layers are followed by norm, activation and
        dropout; the result is log-probabilities."""
        if self.sgc:
            h = self.convs[0].forward_sampler(x, adjs)
        else:
            h, last = x, self.nlayers - 1
            for depth, hop in enumerate(adjs):
                h = self.convs[depth](h, hop[0])
                if depth < last:
                    if self.norms is not None:
                        h = self.norms[depth](h)
                    if self.act is not None:
                        h = self.act(h)
                    h = F.dropout(h, self.dropout, training=self.training)
        return F.log_softmax(h, dim=1)

    @torch.no_grad()
    def predict(self, x: Tensor, edge_index, *, edge_weight: Optional[Tensor] = None, edge_attr=None) -> Tensor:
        self.eval()
        return self.forward(x, edge_index, edge_weight=edge_weight, edge_attr=edge_attr)

    def _sampled_layer(self, depth: int, x: Tensor, adj_t, keep: int, raw_final: bool) -> Tensor:
        """One depth of sampled evaluation on a bipartite block whose first ``keep`` output rows are the targets
        (NeighborSampler puts a batch's targets first): propagate, THEN slice.
          SGC      : hop ``depth`` of the K-hop convolution; its ``lin`` is applied once, after the last hop;
          otherwise: convolution ``depth``; hidden layers get act / norm; the final layer stays raw unless a
                     jumping-knowledge head exists and the caller wants it applied (``raw_final`` False)."""
        if self.sgc:
            conv = self.convs[0]
            h = conv.propagate(adj_t, x=x)[:keep]
            return conv.lin(h) if depth == self.temp_layers - 1 else h
        h = self.convs[depth](x, adj_t)[:keep]
        final = depth == self.nlayers - 1
        if final and raw_final:
            return h
        h = self._post(h, depth)
        return self.lin(h) if final and hasattr(self, "lin") else h

    @torch.no_grad()
    def inference(self, x_all: Tensor, loader, device=None, progress_bar: bool = False) -> Tensor:
        """Layer-wise sampled evaluation (reference interface: basicgnn_large.py:235-281).  The loader is walked once
        per depth and yields ``(batch_size, n_id, block)`` with ``block.adj_t`` the batch's bipartite adjacency; every
        pass maps the features of ALL nodes to the next depth, batch by batch (outputs collected on the host in loader
        order, which must cover the nodes in order), so no depth ever needs more than one batch on the device."""
        self.eval()
        assert self.jk_mode is None or self.jk_mode == "last"
        feats = x_all
        for depth in range(self.temp_layers):
            feats = torch.cat([self._sampled_layer(depth, feats[n_id].to(device), block.adj_t.to(device), batch_size,
                                                   raw_final=self.jk_mode is None).cpu()
                               for batch_size, n_id, block in loader], dim=0)
        return F.log_softmax(feats, dim=1).to(device)

    @torch.no_grad()
    def large_inference(self, x_all, loader, device=None, progress_bar: bool = False) -> Tensor:
        """Batch-wise sampled evaluation (reference interface: basicgnn_large.py:283-318).  The loader yields
        ``(batch_size, n_id, blocks)`` with one ``(adj_t, e_id, size)`` block per depth; ``size[1]`` is the number of
        target rows of that block.  A batch runs through every depth on the device and only its final rows return to
        the host; the final layer's output is left raw (no head), as the reference does on this route."""
        self.eval()
        assert self.jk_mode is None or self.jk_mode == "last"
        outs: List[Tensor] = []
        for _batch_size, n_id, blocks in loader:
            h = torch.as_tensor(x_all[n_id], dtype=torch.float32).to(device)
            for depth in range(self.temp_layers):
                block = blocks[depth]
                h = self._sampled_layer(depth, h, block.adj_t.to(device), int(block[2][1]), raw_final=True)
            outs.append(h.cpu())
        return F.log_softmax(torch.cat(outs, dim=0), dim=1)

    @torch.no_grad()
    def full_inference(self, x: Tensor, adj_t) -> Tensor:
        """MI355X replacement for the sampler loops above: the whole graph and X are resident in the
        288 GB of HBM, so evaluation is `predict` on the full graph -- one K-hop call per layer."""
        return self.predict(x, adj_t)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.nfeat}, {self.nclass}, nlayers={self.nlayers})"


class GCN(BasicGNN):
    def init_conv(self, nfeat, nclass, **kwargs):
        return GCNConv(nfeat, nclass, **kwargs)


class GraphSAGE(BasicGNN):
    def init_conv(self, nfeat, nclass, **kwargs):
        return SAGEConv(nfeat, nclass, project=False, **kwargs)


class GIN(BasicGNN):
    def init_conv(self, nfeat, nclass, **kwargs):
        mlp = MLP([nfeat, nclass, nclass], act=self.act, act_first=self.act_first, norm=self.norm, norm_kwargs=self.norm_kwargs)
        return GINConv(mlp, train_eps=True, **kwargs)


class SGC(BasicGNN):
    def init_conv(self, nfeat, nclass, **kwargs):
        return SGConv(nfeat, nclass, self.temp_layers, **kwargs)


class JKNet(BasicGNN):
    def init_conv(self, nfeat, nclass, **kwargs):
        return GCNConv(nfeat, nclass, **kwargs)


__all__ = ["GCN", "GraphSAGE", "GIN", "SGC", "JKNet"]
