"""SAGEConv with the reference's interface (/root/reference/models/convs_large/sage_conv.py:71-169):
mean aggregation through the HIP SpMM (``reduce='mean'``, sage_conv.py:148-150), then
``lin_l(agg) + lin_r(x_root)``.  Only the ``aggr`` strings the kernels implement are accepted."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from .message_passing import Adj, MessagePassing


class SAGEConv(MessagePassing):
    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int, aggr: Optional[str] = "mean",
                 root_weight: bool = True, project: bool = False, bias: bool = True, **kwargs):
        super().__init__(aggr=aggr, **kwargs)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.root_weight = root_weight
        self.project = project
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        if self.project:
            self.lin = torch.nn.Linear(in_channels[0], in_channels[0], bias=True)
        self.lin_l = torch.nn.Linear(in_channels[0], out_channels, bias=bias)
        if self.root_weight:
            self.lin_r = torch.nn.Linear(in_channels[1], out_channels, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        if self.project:
            self.lin.reset_parameters()
        self.lin_l.reset_parameters()
        if self.root_weight:
            self.lin_r.reset_parameters()

    def forward(self, x, edge_index: Adj, edge_weight: Optional[Tensor] = None, size=None) -> Tensor:
        # sage_conv.py:123-143
        if isinstance(x, Tensor):
            x = (x, x)
        if self.project and hasattr(self, "lin"):
            x = (self.lin(x[0]).relu(), x[1])
        out = self.propagate(edge_index, x=x, edge_weight=edge_weight, size=size)
        out = self.lin_l(out)
        x_r = x[1] if x[1].shape[0] == out.shape[0] else x[1][:out.shape[0]]
        if self.root_weight and x_r is not None:
            out = out + self.lin_r(x_r)
        return out

    def message(self, x_j: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, aggr={self.aggr})"
