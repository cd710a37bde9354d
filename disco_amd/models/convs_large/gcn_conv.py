"""GCNConv with the reference's interface (/root/reference/models/convs_large/gcn_conv.py:73-171):
``lin`` (no bias, glorot) -> one ``propagate`` on a PRE-NORMALISED adjacency -> ``+ bias``."""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor
from torch.nn import Parameter

from ...utils import gcn_norm  # noqa: F401  (re-exported like the reference module does, gcn_conv.py:32)
from .message_passing import Adj, MessagePassing


def glorot(w: Tensor) -> None:
    stdv = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
    w.data.uniform_(-stdv, stdv)


class GCNConv(MessagePassing):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, **kwargs):
        kwargs.setdefault("aggr", "add")
        super().__init__(**kwargs)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=False)
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.lin.weight)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x: Tensor, edge_index: Adj, edge_weight: Optional[Tensor] = None) -> Tensor:
        # gcn_conv.py:153-165
        x = self.lin(x)
        out = self.propagate(edge_index, x=x, edge_weight=edge_weight, size=None)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"
