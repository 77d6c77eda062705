"""dgl_amd — MI355X-native g-SpMM / g-SDDMM message-passing path behind DGL's operator API.

Only the hot path is here: hand-written gfx950 HIP kernels in
``dgl_amd/csrc`` exposed through a C ABI (``include/dgl_amd.h``), and the thin Python host
side that mirrors ``dgl.ops`` / ``DGLGraph.update_all``.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError if libdgl_amd.so has not been built)
from ._lib import DGLAMDError

__version__ = "0.1.0"

from . import function  # noqa: E402,F401
from . import ops  # noqa: E402,F401
from .heterograph import (DGLGraph, ETYPE, NTYPE, create_block, from_networkx, graph, heterograph,  # noqa: E402,F401
                          rand_bipartite, rand_graph, reverse, to_heterogeneous, to_homogeneous)
from . import udf  # noqa: E402,F401
from . import nn  # noqa: E402,F401
from . import sampling  # noqa: E402,F401
from .transforms import (add_reverse_edges, add_self_loop, batch, bipartite_from_scipy, edge_subgraph,  # noqa: E402,F401
                         from_scipy, in_subgraph, node_subgraph, remove_edges, remove_self_loop, reorder_graph,
                         to_bidirected, to_simple, unbatch)
from .readout import *  # noqa: E402,F401,F403
from . import sparse  # noqa: E402,F401
from .ops import edge_softmax  # noqa: E402,F401
from .sampling import EID, NID, NeighborSampler, to_block  # noqa: E402,F401
from .mm import gather_mm, segment_mm  # noqa: E402,F401
from .segment import scatter_add, segment_reduce, segment_softmax  # noqa: E402,F401
from .sparse_kernels import release_static, set_auto_edge_operand, static_features  # noqa: E402,F401

from .edge_order import edge_order_handoff, set_edge_order_handoff  # noqa: E402,F401
from .capture import CapturedStep  # noqa: E402,F401

DGLError = DGLAMDError
