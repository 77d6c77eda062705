"""``dgl_amd.nn`` — only what is NEW above the operator API.  The layers themselves (GraphConv, SAGEConv, GATConv,
RelGraphConv, TypedLinear, HeteroGraphConv ...) are the reference's own ``python/dgl/nn/pytorch/*.py``: they stay
untouched above ``update_all`` / ``dgl.ops`` (docs/DESIGN_detail_r1_r5.md §1) and run on this package through the ``dgl`` -> ``dgl_amd``
alias (``tools/ref_suite/run.py --suite nn`` imports them unmodified and runs the reference's layer tests).

``gat_attention`` is the one addition: GATConv's attention block (python/dgl/nn/pytorch/conv/gatconv.py:330-347 —
``u_add_v`` -> ``leaky_relu`` -> ``edge_softmax`` -> ``u_mul_e_sum``) as ONE operator, so that a layer can hand the whole
block to the fused kernel (csrc/gat_attention.hip) instead of four launches that write and re-read three (E, H) tensors.
"""
import torch.nn.functional as F

from . import edge_order as _eo
from . import function as fn
from .ops import edge_softmax

__all__ = ["functional", "gat_attention"]


class functional:  # noqa: N801  (a namespace: python/dgl/nn/functional/__init__.py exports exactly this)
    """``dgl.nn.functional``."""
    edge_softmax = staticmethod(edge_softmax)


def gat_attention(graph, ft, el, er, negative_slope=0.2, fused=None, handoff=False):
    """``out[v] = sum_{u->v} softmax_v(leaky_relu(el[u] + er[v])) * ft[u]`` per head.

    ft: (N_src, H, D); el: (N_src, H, 1); er: (N_dst, H, 1) -> (N_dst, H, D).  ``fused=None`` takes the one-pass kernel
    (``dgl_amd.ops.gat_attention``) whenever it applies and the composed operators otherwise; ``fused=False`` forces the
    composition (the reference's own sequence, the parity yardstick); ``handoff=True`` runs the composition inside
    ``dgl_amd.edge_order_handoff()`` (opt-in: entering that scope installs edge_order's process-wide shims)."""
    from . import ops

    if fused is None:
        fused = ops.gat_attention_applies(graph, ft, el, er)
    if fused:
        return ops.gat_attention(graph, ft, el, er, negative_slope)
    with graph.local_scope():
        graph.srcdata.update({"ft": ft, "el": el})
        graph.dstdata.update({"er": er})
        with _eo.edge_order_handoff(bool(handoff)):
            graph.apply_edges(fn.u_add_v("el", "er", "e"))
            graph.edata["a"] = edge_softmax(graph, F.leaky_relu(graph.edata.pop("e"), negative_slope))
            graph.update_all(fn.u_mul_e("ft", "a", "m"), fn.sum("m", "ft"))
            return graph.dstdata["ft"]
