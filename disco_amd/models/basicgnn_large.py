"""Callers of the propagation path with the reference's interface
(/root/reference/models/basicgnn_large.py:27-512): ``BasicGNN`` and ``GCN / GraphSAGE / GIN / SGC / JKNet``.

Only what sits between the drivers and the convolutions is reproduced: the layer loop
(``forward`` :161-207), ``predict`` (:222-233) and the loader-driven ``inference`` / ``large_inference``
(:235-318).  Convolutions are the HIP-backed ones of ``disco_amd.models.convs_large``; dense layers,
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
        # basicgnn_large.py:209-220
        if not self.sgc:
            for i, (adj, _, size) in enumerate(adjs):
                x = self.convs[i](x, adj)
                if i != self.nlayers - 1:
                    x = self.norms[i](x)
                    x = self.act(x)
                    x = F.dropout(x, self.dropout, training=self.training)
        else:
            x = self.convs[0].forward_sampler(x, adjs)
        return F.log_softmax(x, dim=1)

    @torch.no_grad()
    def predict(self, x: Tensor, edge_index, *, edge_weight: Optional[Tensor] = None, edge_attr=None) -> Tensor:
        self.eval()
        return self.forward(x, edge_index, edge_weight=edge_weight, edge_attr=edge_attr)

    @torch.no_grad()
    def inference(self, x_all: Tensor, loader, device=None, progress_bar: bool = False) -> Tensor:
        """Layer-wise sampled inference (basicgnn_large.py:235-281).  `loader` yields
        ``(batch_size, n_id, adj)`` with ``adj.adj_t`` a bipartite adjacency (any object this package's
        convolutions accept)."""
        self.eval()
        assert self.jk_mode is None or self.jk_mode == "last"
        for i in range(self.temp_layers):
            xs: List[Tensor] = []
            for batch_size, n_id, adj in loader:
                x = x_all[n_id].to(device)
                edge_index = adj.adj_t.to(device)
                if not self.sgc:
                    x = self.convs[i](x, edge_index)[:batch_size]
                    if i == self.nlayers - 1 and self.jk_mode is None:
                        xs.append(x.cpu())
                        continue
                    x = self._post(x, i)
                    if i == self.nlayers - 1 and hasattr(self, "lin"):
                        x = self.lin(x)
                else:
                    x = self.convs[0].propagate(edge_index, x=x)[:batch_size]
                    if i == self.temp_layers - 1:
                        x = self.convs[0].lin(x)
                xs.append(x.cpu())
            x_all = torch.cat(xs, dim=0)
        return F.log_softmax(x_all, dim=1).to(device)

    @torch.no_grad()
    def large_inference(self, x_all, loader, device=None, progress_bar: bool = False) -> Tensor:
        """Batch-wise sampled inference (basicgnn_large.py:283-318); `loader` yields
        ``(batch_size, n_id, adjs)`` with one ``(adj_t, e_id, size)``-like entry per layer."""
        self.eval()
        assert self.jk_mode is None or self.jk_mode == "last"
        xs: List[Tensor] = []
        for batch_size, n_id, adjs in loader:
            x = torch.as_tensor(x_all[n_id], dtype=torch.float32).to(device)
            for i in range(self.temp_layers):
                target_num = adjs[i][2][1]
                edge_index = adjs[i].adj_t.to(device)
                if not self.sgc:
                    x = self.convs[i](x, edge_index)[:target_num]
                    if i == self.nlayers - 1:
                        continue
                    x = self._post(x, i)
                else:
                    x = self.convs[0].propagate(edge_index, x=x)[:target_num]
                    if i == self.temp_layers - 1:
                        x = self.convs[0].lin(x)
            xs.append(x.cpu())
        return F.log_softmax(torch.cat(xs, dim=0), dim=1)

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
