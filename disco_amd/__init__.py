"""disco_amd -- MI355X-native implementation of DisCo's GCN-normalised sparse propagation path.

Host side mirrors the reference's Python module API (``models.convs_large``, ``utils.gcn_norm``,
``models.parametrized_adj_lp.PGE_Edge``); all arithmetic on the path runs in hand-written HIP kernels
behind the C ABI of ``include/disco_hip.h`` (``disco_amd/csrc/libdisco_hip.so``).
"""
from .sparse import SparseTensor  # noqa: F401
from .ops import khop, matmul, max_aggregate, pair_gather, spmm  # noqa: F401

__version__ = "0.1.0"
