"""GINConv with the reference's interface (/root/reference/models/convs_large/gin_conv.py:53-93):
``nn(aggregate(x) + (1 + eps) * x_root)`` with the reference's default ``aggr='mean'`` (gin_conv.py:55)
computed by the HIP SpMM (gin_conv.py:88-90)."""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import Tensor

from .message_passing import Adj, MessagePassing


def _reset(nn) -> None:
    if hasattr(nn, "children") and len(list(nn.children())) > 0:
        for item in nn.children():
            _reset(item)
    elif hasattr(nn, "reset_parameters"):
        nn.reset_parameters()


class GINConv(MessagePassing):
    def __init__(self, nn: Callable, eps: float = 0., train_eps: bool = False, **kwargs):
        kwargs.setdefault("aggr", "mean")
        super().__init__(**kwargs)
        self.nn = nn
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))
        self.reset_parameters()

    def reset_parameters(self):
        _reset(self.nn)
        self.eps.data.fill_(self.initial_eps)

    def forward(self, x, edge_index: Adj, edge_weight: Optional[Tensor] = None, size=None) -> Tensor:
        # gin_conv.py:70-83
        if isinstance(x, Tensor):
            x = (x, x)
        out = self.propagate(edge_index, x=x, edge_weight=edge_weight, size=size)
        x_r = x[1]
        if x_r is not None:
            out = out + (1 + self.eps) * x_r[:out.shape[0]]
        return self.nn(out)

    def message(self, x_j: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(nn={self.nn})"
