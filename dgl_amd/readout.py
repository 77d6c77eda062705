"""Graph-level readout over the graphs of a batch (python/dgl/readout.py): the callers of ``segment_reduce`` /
``segment_softmax`` — one segment per batched graph, lengths = ``batch_num_nodes`` / ``batch_num_edges``.

``readout_nodes / readout_edges`` and the ``sum / mean / max`` forms run the segment-reduce kernels (csrc/segment.hip),
``softmax_nodes / softmax_edges`` the segment softmax built on them, ``broadcast_*`` is a row repeat and ``topk_*`` a
padded sort (torch, as in the reference: readout.py:531-657)."""
import torch

from . import segment as _segment
from ._lib import DGLAMDError

__all__ = ["readout_nodes", "readout_edges", "sum_nodes", "sum_edges", "mean_nodes", "mean_edges", "max_nodes", "max_edges",
           "softmax_nodes", "softmax_edges", "broadcast_nodes", "broadcast_edges", "topk_nodes", "topk_edges"]


def _node_side(graph, ntype):
    return graph.nodes[ntype].data, graph.batch_num_nodes(ntype)


def _edge_side(graph, etype):
    return graph.edges[etype].data, graph.batch_num_edges(etype)


def _readout(data, lens, feat, weight, op):
    x = data[feat]
    if weight is not None:
        x = x * data[weight]
    return _segment.segment_reduce(lens, x, reducer=op)


def readout_nodes(graph, feat, weight=None, *, op="sum", ntype=None):
    """One row per batched graph: ``op`` over its nodes' ``feat`` (times ``weight``) (readout.py:26-101)."""
    data, lens = _node_side(graph, ntype)
    return _readout(data, lens, feat, weight, op)


def readout_edges(graph, feat, weight=None, *, op="sum", etype=None):
    data, lens = _edge_side(graph, etype)
    return _readout(data, lens, feat, weight, op)


def sum_nodes(graph, feat, weight=None, *, ntype=None):
    return readout_nodes(graph, feat, weight, ntype=ntype, op="sum")


def sum_edges(graph, feat, weight=None, *, etype=None):
    return readout_edges(graph, feat, weight, etype=etype, op="sum")


def mean_nodes(graph, feat, weight=None, *, ntype=None):
    return readout_nodes(graph, feat, weight, ntype=ntype, op="mean")


def mean_edges(graph, feat, weight=None, *, etype=None):
    return readout_edges(graph, feat, weight, etype=etype, op="mean")


def max_nodes(graph, feat, weight=None, *, ntype=None):
    return readout_nodes(graph, feat, weight, ntype=ntype, op="max")


def max_edges(graph, feat, weight=None, *, etype=None):
    return readout_edges(graph, feat, weight, etype=etype, op="max")


def softmax_nodes(graph, feat, *, ntype=None):
    """Softmax of ``feat`` over the nodes of each batched graph (readout.py:248-305)."""
    data, lens = _node_side(graph, ntype)
    return _segment.segment_softmax(lens, data[feat])


def softmax_edges(graph, feat, *, etype=None):
    data, lens = _edge_side(graph, etype)
    return _segment.segment_softmax(lens, data[feat])


def _broadcast(graph, graph_feat, lens):
    bs = getattr(graph, "batch_size", 1)
    if graph_feat.shape[0] != bs and bs == 1:
        graph_feat = graph_feat.unsqueeze(0)          # (the reference warns: use a (1, *) tensor for a single graph)
    return torch.repeat_interleave(graph_feat, lens.long(), dim=0)


def broadcast_nodes(graph, graph_feat, *, ntype=None):
    """Row i of ``graph_feat`` repeated for every node of batched graph i (readout.py:374-448)."""
    return _broadcast(graph, graph_feat, graph.batch_num_nodes(ntype))


def broadcast_edges(graph, graph_feat, *, etype=None):
    return _broadcast(graph, graph_feat, graph.batch_num_edges(etype))


def _topk_on(feat, lens, k, descending, sortby):
    if feat.dim() > 2:
        raise DGLAMDError("Only support feature with dimension less than or equal to 2")
    if feat.dim() == 1:
        feat = feat.unsqueeze(-1)
    lens = lens.long()
    bs, hidden = lens.shape[0], feat.shape[-1]
    length = max(int(lens.max()) if bs else 0, k)
    fill = float("-inf") if descending else float("inf")
    padded = feat.new_full((bs, length, hidden), fill)
    seg = torch.repeat_interleave(torch.arange(bs, device=feat.device), lens)
    pos = torch.arange(feat.shape[0], device=feat.device) - (torch.cumsum(lens, 0) - lens)[seg]
    padded[seg, pos] = feat
    if sortby is not None:
        idx = padded[..., sortby].topk(k, -1, largest=descending)[1]                 # (bs, k)
        out = torch.gather(padded, 1, idx.unsqueeze(-1).expand(bs, k, hidden))
    else:
        idx = torch.argsort(padded, 1, descending=descending)[:, :k]                 # (bs, k, hidden): per column
        out = torch.gather(padded, 1, idx)
    return torch.masked_fill(out, torch.isinf(out), 0), idx


def topk_nodes(graph, feat, k, *, descending=True, sortby=None, ntype=None):
    """Per batched graph the ``k`` largest (smallest) node features — whole rows ranked by column ``sortby``, or every
    column on its own — zero-padded when the graph has fewer nodes; also the node positions (readout.py:660-772)."""
    data, lens = _node_side(graph, ntype)
    return _topk_on(data[feat], lens, k, descending, sortby)


def topk_edges(graph, feat, k, *, descending=True, sortby=None, etype=None):
    data, lens = _edge_side(graph, etype)
    return _topk_on(data[feat], lens, k, descending, sortby)
