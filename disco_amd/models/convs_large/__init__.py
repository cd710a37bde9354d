"""Drop-in for ``models.convs_large`` (/root/reference/models/convs_large/__init__.py:1-11)."""
from .gcn_conv import GCNConv
from .gin_conv import GINConv
from .message_passing import MessagePassing
from .sage_conv import SAGEConv
from .sg_conv import SGConv

__all__ = ["GCNConv", "SAGEConv", "GINConv", "SGConv", "MessagePassing"]
