"""The three layers BASELINE.json's configs are quoted on, as CALLERS of the hot path (SURVEY.md §8 "who calls it").

``GraphConv`` (configs[0]; python/dgl/nn/pytorch/conv/graphconv.py:262-470), ``SAGEConv`` with the mean / gcn / pool
aggregators (configs[3]; sageconv.py:100-290) and ``GATConv`` (configs[2]; gatconv.py:135-370): constructor arguments,
parameter names and shapes, forward semantics and error cases follow the reference, the message passing goes through
``DGLGraph.update_all`` / ``apply_edges`` / ``dgl_amd.ops.edge_softmax`` — nothing here is a kernel.

What ``GATConv`` adds to the kernels' side: its attention block (``u_add_v`` -> ``leaky_relu`` -> ``edge_softmax`` ->
dropout -> ``u_mul_e_sum``) runs inside ``dgl_amd.edge_order_handoff()``, the opt-in scope in which edge tensors travel
between the operators in the CSC's position order (dgl_amd/edge_order.py; GATConv forward + backward behind DGL's
edge-id map 19 ms instead of 30 ms at 62 M edges).  The module is a region that controls everything the tagged tensors
meet — four torch functions, all on the sweep of tests/test_edge_order_sweep.py — and NOTHING tagged leaves it: the
node output never was, and ``get_attention=True`` hands out the plain edge-id-ordered tensor.  (Entering the scope
installs edge_order's two process-wide shims, like any opt-in.)  ``GATConv.handoff = False`` (class or instance) keeps
the block on plain tensors.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import edge_order as _eo
from . import function as fn
from ._lib import DGLAMDError
from .ops import edge_softmax

__all__ = ["GraphConv", "SAGEConv", "GATConv", "functional"]


class functional:  # noqa: N801  (a namespace: python/dgl/nn/functional/__init__.py exports exactly this)
    """``dgl.nn.functional``."""
    edge_softmax = staticmethod(edge_softmax)


_ZERO_IN_DEGREE = ("There are 0-in-degree nodes in the graph, output for those nodes will be invalid. This is harmful "
                   "for some applications, causing silent performance regression. Adding self-loop on the input graph "
                   "by calling `g = dgl.add_self_loop(g)` will resolve the issue. Setting ``allow_zero_in_degree`` to "
                   "be `True` when constructing this module will suppress the check and let the code run.")


def _expand_as_pair(feat, graph):
    """(source features, destination features): a pair as given; on a block the destination nodes are the first
    ``num_dst_nodes`` source nodes (python/dgl/utils/internal.py expand_as_pair)."""
    if isinstance(feat, tuple):
        return feat
    if graph.is_block:
        return feat, feat[: graph.number_of_dst_nodes()]
    return feat, feat


def _check_in_degrees(graph, allow):
    if not allow and bool((graph.in_degrees() == 0).any()):
        raise DGLAMDError(_ZERO_IN_DEGREE)


class GraphConv(nn.Module):
    """``h_i = b + sum_j c_ji h_j W`` with ``norm`` in {'both', 'right', 'left', 'none'} (graphconv.py)."""

    def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None,
                 allow_zero_in_degree=False):
        super().__init__()
        if norm not in ("none", "both", "right", "left"):
            raise DGLAMDError('Invalid norm value. Must be either "none", "both", "right" or "left". '
                              'But got "{}".'.format(norm))
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        if weight:
            self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        else:
            self.register_parameter("weight", None)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_feats))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self._activation = activation

    def reset_parameters(self):
        if self.weight is not None:
            nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def forward(self, graph, feat, weight=None, edge_weight=None):
        with graph.local_scope():
            _check_in_degrees(graph, self._allow_zero_in_degree)
            aggregate_fn = fn.copy_u("h", "m")
            if edge_weight is not None:
                assert edge_weight.shape[0] == graph.num_edges()
                graph.edata["_edge_weight"] = edge_weight
                aggregate_fn = fn.u_mul_e("h", "_edge_weight", "m")
            feat_src, feat_dst = _expand_as_pair(feat, graph)
            if self._norm in ("left", "both"):
                degs = graph.out_degrees().to(feat_src).clamp(min=1)
                norm = torch.pow(degs, -0.5) if self._norm == "both" else 1.0 / degs
                feat_src = feat_src * norm.reshape(norm.shape + (1,) * (feat_src.dim() - 1))
            if weight is not None:
                if self.weight is not None:
                    raise DGLAMDError("External weight is provided while at the same time the module has defined its "
                                      "own weight parameter. Please create the module with flag weight=False.")
            else:
                weight = self.weight
            if self._in_feats > self._out_feats:      # multiply first: the aggregation then moves the narrower rows
                if weight is not None:
                    feat_src = torch.matmul(feat_src, weight)
                graph.srcdata["h"] = feat_src
                graph.update_all(aggregate_fn, fn.sum(msg="m", out="h"))
                rst = graph.dstdata["h"]
            else:
                graph.srcdata["h"] = feat_src
                graph.update_all(aggregate_fn, fn.sum(msg="m", out="h"))
                rst = graph.dstdata["h"]
                if weight is not None:
                    rst = torch.matmul(rst, weight)
            if self._norm in ("right", "both"):
                degs = graph.in_degrees().to(feat_dst).clamp(min=1)
                norm = torch.pow(degs, -0.5) if self._norm == "both" else 1.0 / degs
                rst = rst * norm.reshape(norm.shape + (1,) * (feat_dst.dim() - 1))
            if self.bias is not None:
                rst = rst + self.bias
            if self._activation is not None:
                rst = self._activation(rst)
            return rst


class SAGEConv(nn.Module):
    """GraphSAGE layer, aggregators 'mean', 'gcn' and 'pool' (sageconv.py; 'lstm' — a recurrent reducer over a node's
    mailbox — is a user-defined reduce function in the reference and is not offered here)."""

    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0.0, bias=True, norm=None, activation=None):
        super().__init__()
        if aggregator_type not in ("mean", "gcn", "pool"):
            raise DGLAMDError("Invalid aggregator_type. Must be one of {}. But got {!r} instead.".format(
                ("mean", "gcn", "pool"), aggregator_type))
        self._in_src_feats, self._in_dst_feats = (in_feats if isinstance(in_feats, tuple) else (in_feats, in_feats))
        self._out_feats, self._aggre_type = out_feats, aggregator_type
        self.norm, self.activation = norm, activation
        self.feat_drop = nn.Dropout(feat_drop)
        if aggregator_type == "pool":
            self.fc_pool = nn.Linear(self._in_src_feats, self._in_src_feats)
        self.fc_neigh = nn.Linear(self._in_src_feats, out_feats, bias=False)
        if aggregator_type != "gcn":
            self.fc_self = nn.Linear(self._in_dst_feats, out_feats, bias=bias)
        elif bias:
            self.bias = nn.Parameter(torch.zeros(out_feats))
        else:
            self.register_buffer("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        if self._aggre_type == "pool":
            nn.init.xavier_uniform_(self.fc_pool.weight, gain=gain)
        if self._aggre_type != "gcn":
            nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def forward(self, graph, feat, edge_weight=None):
        with graph.local_scope():
            if isinstance(feat, tuple):
                feat_src, feat_dst = self.feat_drop(feat[0]), self.feat_drop(feat[1])
            else:
                feat_src = feat_dst = self.feat_drop(feat)
                if graph.is_block:
                    feat_dst = feat_src[: graph.number_of_dst_nodes()]
            msg_fn = fn.copy_u("h", "m")
            if edge_weight is not None:
                assert edge_weight.shape[0] == graph.num_edges()
                graph.edata["_edge_weight"] = edge_weight
                msg_fn = fn.u_mul_e("h", "_edge_weight", "m")
            h_self = feat_dst
            if graph.num_edges() == 0:
                graph.dstdata["neigh"] = torch.zeros(feat_dst.shape[0], self._in_src_feats).to(feat_dst)
            lin_before_mp = self._in_src_feats > self._out_feats     # multiply first when that narrows the rows
            if self._aggre_type == "mean":
                graph.srcdata["h"] = self.fc_neigh(feat_src) if lin_before_mp else feat_src
                graph.update_all(msg_fn, fn.mean("m", "neigh"))
                h_neigh = graph.dstdata["neigh"]
                if not lin_before_mp:
                    h_neigh = self.fc_neigh(h_neigh)
            elif self._aggre_type == "gcn":
                graph.srcdata["h"] = self.fc_neigh(feat_src) if lin_before_mp else feat_src
                graph.dstdata["h"] = (self.fc_neigh(feat_dst) if lin_before_mp else feat_dst) if isinstance(feat, tuple) \
                    else graph.srcdata["h"][: graph.num_dst_nodes()]
                graph.update_all(msg_fn, fn.sum("m", "neigh"))
                degs = graph.in_degrees().to(feat_dst)
                h_neigh = (graph.dstdata["neigh"] + graph.dstdata["h"]) / (degs.unsqueeze(-1) + 1)
                if not lin_before_mp:
                    h_neigh = self.fc_neigh(h_neigh)
            else:   # pool
                graph.srcdata["h"] = F.relu(self.fc_pool(feat_src))
                graph.update_all(msg_fn, fn.max("m", "neigh"))
                h_neigh = self.fc_neigh(graph.dstdata["neigh"])
            if self._aggre_type == "gcn":
                rst = h_neigh
                if self.bias is not None:
                    rst = rst + self.bias
            else:
                rst = self.fc_self(h_self) + h_neigh
            if self.activation is not None:
                rst = self.activation(rst)
            if self.norm is not None:
                rst = self.norm(rst)
            return rst


class GATConv(nn.Module):
    """Graph attention layer (gatconv.py): ``e_ij = LeakyReLU(a_l . W h_j + a_r . W h_i)``, softmax over the incoming
    edges of ``i``, ``h_i' = sum_j alpha_ij W h_j`` per head."""

    handoff = True   # run the attention block inside dgl_amd.edge_order_handoff() (see the module docstring)

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0.0, attn_drop=0.0, negative_slope=0.2,
                 residual=False, activation=None, allow_zero_in_degree=False, bias=True):
        super().__init__()
        self._num_heads, self._out_feats = num_heads, out_feats
        self._in_src_feats, self._in_dst_feats = (in_feats if isinstance(in_feats, tuple) else (in_feats, in_feats))
        self._allow_zero_in_degree = allow_zero_in_degree
        if isinstance(in_feats, tuple):
            self.fc_src = nn.Linear(self._in_src_feats, out_feats * num_heads, bias=False)
            self.fc_dst = nn.Linear(self._in_dst_feats, out_feats * num_heads, bias=False)
        else:
            self.fc = nn.Linear(self._in_src_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.feat_drop, self.attn_drop = nn.Dropout(feat_drop), nn.Dropout(attn_drop)
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        self.has_linear_res = self.has_explicit_bias = False
        if residual:
            if self._in_dst_feats != out_feats * num_heads:
                self.res_fc = nn.Linear(self._in_dst_feats, num_heads * out_feats, bias=bias)
                self.has_linear_res = True
            else:
                self.res_fc = nn.Identity()
        else:
            self.register_buffer("res_fc", None)
        if bias and not self.has_linear_res:
            self.bias = nn.Parameter(torch.empty(num_heads * out_feats))
            self.has_explicit_bias = True
        else:
            self.register_buffer("bias", None)
        self.reset_parameters()
        self.activation = activation

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        for lin in ("fc", "fc_src", "fc_dst"):
            if hasattr(self, lin):
                nn.init.xavier_normal_(getattr(self, lin).weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)
        if self.has_explicit_bias:
            nn.init.constant_(self.bias, 0)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)
            if self.res_fc.bias is not None:
                nn.init.constant_(self.res_fc.bias, 0)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def forward(self, graph, feat, edge_weight=None, get_attention=False):
        with graph.local_scope():
            _check_in_degrees(graph, self._allow_zero_in_degree)
            heads, d = self._num_heads, self._out_feats
            if isinstance(feat, tuple):
                src_prefix, dst_prefix = feat[0].shape[:-1], feat[1].shape[:-1]
                h_src, h_dst = self.feat_drop(feat[0]), self.feat_drop(feat[1])
                if hasattr(self, "fc_src"):
                    feat_src = self.fc_src(h_src).view(*src_prefix, heads, d)
                    feat_dst = self.fc_dst(h_dst).view(*dst_prefix, heads, d)
                else:
                    feat_src = self.fc(h_src).view(*src_prefix, heads, d)
                    feat_dst = self.fc(h_dst).view(*dst_prefix, heads, d)
            else:
                src_prefix = dst_prefix = feat.shape[:-1]
                h_src = h_dst = self.feat_drop(feat)
                feat_src = feat_dst = self.fc(h_src).view(*src_prefix, heads, d)
                if graph.is_block:
                    feat_dst = feat_src[: graph.number_of_dst_nodes()]
                    h_dst = h_dst[: graph.number_of_dst_nodes()]
                    dst_prefix = (graph.number_of_dst_nodes(),) + tuple(dst_prefix[1:])
            # "first projection then addition": a^T [W h_i || W h_j] = a_l . W h_j + a_r . W h_i (gatconv.py:311-321)
            el = (feat_src * self.attn_l).sum(dim=-1).unsqueeze(-1)
            er = (feat_dst * self.attn_r).sum(dim=-1).unsqueeze(-1)
            graph.srcdata.update({"ft": feat_src, "el": el})
            graph.dstdata.update({"er": er})
            # ---- the attention block: edge tensors may travel in the CSC's position order in here, and only in here ----
            with _eo.edge_order_handoff(bool(self.handoff)):
                graph.apply_edges(fn.u_add_v("el", "er", "e"))
                e = self.leaky_relu(graph.edata.pop("e"))
                a = self.attn_drop(edge_softmax(graph, e))
                if edge_weight is not None:
                    a = a * edge_weight.tile(1, heads, 1).transpose(0, 2)
                graph.edata["a"] = a
                graph.update_all(fn.u_mul_e("ft", "a", "m"), fn.sum("m", "ft"))
                rst = graph.dstdata["ft"]
                attention = _eo.plain(a) if get_attention else None     # what leaves is plain, edge-id ordered
            if self.res_fc is not None and h_dst.numel() != 0:
                rst = rst + self.res_fc(h_dst).view(*dst_prefix, -1, d)
            if self.has_explicit_bias:
                rst = rst + self.bias.view(*((1,) * len(dst_prefix)), heads, d)
            if self.activation:
                rst = self.activation(rst)
            return (rst, attention) if get_attention else rst
