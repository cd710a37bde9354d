"""SGConv with the reference's interface (/root/reference/models/convs_large/sg_conv.py:12-97):
K rounds of ``propagate`` on a PRE-NORMALISED adjacency, then ``lin``."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from ... import ops
from .message_passing import Adj, MessagePassing, as_adj_t


class SGConv(MessagePassing):
    def __init__(self, in_channels: int, out_channels: int, K: int = 3, bias: bool = True, **kwargs):
        kwargs.setdefault("aggr", "add")
        super().__init__(**kwargs)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.K = K
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()

    def forward(self, x: Tensor, edge_index: Adj, edge_weight: Optional[Tensor] = None) -> Tensor:
        # sg_conv.py:71-79
        adj_t = as_adj_t(edge_index)
        fused = (adj_t is not None and self.aggr in ("add", "sum") and self.K >= 1
                 and adj_t.size(0) == adj_t.size(1) and not (x.requires_grad and torch.is_grad_enabled()))
        if fused:
            # all K hops in one C-ABI call with ping-pong buffers (no [N, F] allocation per hop)
            x = ops.khop(adj_t, x, self.K)
        else:
            for _ in range(self.K):
                x = self.propagate(edge_index, x=x, edge_weight=edge_weight, size=None)
        return self.lin(x)

    def forward_sampler(self, x: Tensor, edge_index, edge_weight: Optional[Tensor] = None) -> Tensor:
        # sg_conv.py:81-87: edge_index is a list of (adj, e_id, size) per hop
        for adj, _, size in edge_index:
            x = self.propagate(adj, x=x, edge_weight=edge_weight)
        return self.lin(x)

    def message(self, x_j: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, K={self.K})"
