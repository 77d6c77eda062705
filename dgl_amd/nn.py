"""The three layers BASELINE.json's configs are quoted on, as CALLERS of the hot path (SURVEY.md §8 "who calls it").

``GraphConv`` (configs[0]; python/dgl/nn/pytorch/conv/graphconv.py:262-470), ``SAGEConv`` with the mean / gcn / pool
aggregators (configs[3]; sageconv.py:100-290), ``GATConv`` (configs[2]; gatconv.py:135-370) and, for configs[4] (R-GCN),
``TypedLinear`` (linear.py:13-225 — the caller of ``segment_mm`` / ``gather_mm``), ``RelGraphConv``
(relgraphconv.py:10-215) and ``HeteroGraphConv`` / ``HeteroLinear`` / ``HeteroEmbedding`` (hetero.py): constructor arguments,
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

__all__ = ["GraphConv", "SAGEConv", "GATConv", "TypedLinear", "RelGraphConv", "HeteroGraphConv", "HeteroLinear",
           "HeteroEmbedding", "EdgeWeightNorm", "functional"]


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
    """GraphSAGE layer, aggregators 'mean', 'gcn', 'pool' (one g-SpMM each) and 'lstm' (sageconv.py: a recurrent reducer
    over a node's mailbox — a user-defined reduce function there and here: degree bucketing, dgl_amd/udf.py)."""

    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0.0, bias=True, norm=None, activation=None):
        super().__init__()
        if aggregator_type not in ("mean", "gcn", "pool", "lstm"):
            raise DGLAMDError("Invalid aggregator_type. Must be one of {}. But got {!r} instead.".format(
                {"mean", "gcn", "pool", "lstm"}, aggregator_type))
        self._in_src_feats, self._in_dst_feats = (in_feats if isinstance(in_feats, tuple) else (in_feats, in_feats))
        self._out_feats, self._aggre_type = out_feats, aggregator_type
        self.norm, self.activation = norm, activation
        self.feat_drop = nn.Dropout(feat_drop)
        if aggregator_type == "pool":
            self.fc_pool = nn.Linear(self._in_src_feats, self._in_src_feats)
        if aggregator_type == "lstm":
            self.lstm = nn.LSTM(self._in_src_feats, self._in_src_feats, batch_first=True)
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
        if self._aggre_type == "lstm":
            self.lstm.reset_parameters()
        if self._aggre_type != "gcn":
            nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def _lstm_reducer(self, nodes):
        m = nodes.mailbox["m"]                                   # (nodes of one in-degree, that degree, D)
        h = (m.new_zeros((1, m.shape[0], self._in_src_feats)), m.new_zeros((1, m.shape[0], self._in_src_feats)))
        _, (rst, _) = self.lstm(m, h)
        return {"neigh": rst.squeeze(0)}

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
            elif self._aggre_type == "pool":
                graph.srcdata["h"] = F.relu(self.fc_pool(feat_src))
                graph.update_all(msg_fn, fn.max("m", "neigh"))
                h_neigh = self.fc_neigh(graph.dstdata["neigh"])
            else:   # lstm
                graph.srcdata["h"] = feat_src
                graph.update_all(msg_fn, self._lstm_reducer)
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


class TypedLinear(nn.Module):
    """``y_i = x_i W_{t_i}`` with an optional basis / block-diagonal decomposition of the weights (linear.py:13-225).
    Unsorted types go through ``gather_mm`` (rows grouped by type through a permutation the MFMA kernels read through),
    sorted ones through ``segment_mm`` with the segment lengths kept on the device (the reference reads them back:
    linear.py:203-207 "cause device synchronize")."""

    def __init__(self, in_size, out_size, num_types, regularizer=None, num_bases=None):
        super().__init__()
        self.in_size, self.out_size, self.num_types = in_size, out_size, num_types
        if regularizer is None:
            self.W = nn.Parameter(torch.empty(num_types, in_size, out_size))
        elif regularizer == "basis":
            if num_bases is None:
                raise ValueError('Missing "num_bases" for basis regularization.')
            self.W = nn.Parameter(torch.empty(num_bases, in_size, out_size))
            self.coeff = nn.Parameter(torch.empty(num_types, num_bases))
            self.num_bases = num_bases
        elif regularizer == "bdd":
            if num_bases is None:
                raise ValueError('Missing "num_bases" for bdd regularization.')
            if in_size % num_bases != 0 or out_size % num_bases != 0:
                raise ValueError("Input and output sizes must be divisible by num_bases.")
            self.submat_in, self.submat_out = in_size // num_bases, out_size // num_bases
            self.W = nn.Parameter(torch.empty(num_types, num_bases * self.submat_in * self.submat_out))
            self.num_bases = num_bases
        else:
            raise ValueError('Supported regularizer options: "basis", "bdd", but got {}'.format(regularizer))
        self.regularizer = regularizer
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            fan = self.submat_in if self.regularizer == "bdd" else self.in_size
            nn.init.uniform_(self.W, -1 / fan ** 0.5, 1 / fan ** 0.5)
            if self.regularizer == "basis":
                nn.init.xavier_uniform_(self.coeff, gain=nn.init.calculate_gain("relu"))

    def get_weight(self):
        if self.regularizer == "basis":
            W = self.W.view(self.num_bases, self.in_size * self.out_size)
            return (self.coeff @ W).view(self.num_types, self.in_size, self.out_size)
        return self.W

    def forward(self, x, x_type, sorted_by_type=False):
        from .mm import gather_mm, segment_mm

        w = self.get_weight()
        if self.regularizer == "bdd":
            w = w.index_select(0, x_type.long()).view(-1, self.submat_in, self.submat_out)
            return torch.bmm(x.reshape(-1, 1, self.submat_in), w).view(-1, self.out_size)
        if sorted_by_type:
            pos_l = torch.searchsorted(x_type, torch.arange(self.num_types, device=x.device, dtype=x_type.dtype))
            pos_r = torch.cat([pos_l[1:], torch.tensor([len(x_type)], device=x.device, dtype=pos_l.dtype)])
            return segment_mm(x, w, seglen_a=pos_r - pos_l)          # (lengths stay on the device)
        return gather_mm(x, w, idx_b=x_type)

    def __repr__(self):
        extra = "" if self.regularizer is None else ", regularizer={}, num_bases={}".format(self.regularizer, self.num_bases)
        return "TypedLinear(in_size={}, out_size={}, num_types={}{})".format(self.in_size, self.out_size, self.num_types, extra)


class RelGraphConv(nn.Module):
    """R-GCN layer on a homogeneous graph with an edge-type vector: ``h_i = sum_r sum_{j in N_r(i)} e_ji W_r h_j + W_0 h_i``
    (relgraphconv.py:10-215).  The message is the reference's own — the typed linear map of the gathered source rows,
    a user-defined function over the edge batch — and the reduce is the built-in sum (g-SpMM ``copy_e``)."""

    def __init__(self, in_feat, out_feat, num_rels, regularizer=None, num_bases=None, bias=True, activation=None,
                 self_loop=True, dropout=0.0, layer_norm=False):
        super().__init__()
        if regularizer is not None and num_bases is None:
            num_bases = num_rels
        self.linear_r = TypedLinear(in_feat, out_feat, num_rels, regularizer, num_bases)
        self.bias, self.activation, self.self_loop, self.layer_norm = bias, activation, self_loop, layer_norm
        if self.bias:
            self.h_bias = nn.Parameter(torch.zeros(out_feat))
        if self.layer_norm:
            self.layer_norm_weight = nn.LayerNorm(out_feat, elementwise_affine=True)
        if self.self_loop:
            self.loop_weight = nn.Parameter(torch.empty(in_feat, out_feat))
            nn.init.xavier_uniform_(self.loop_weight, gain=nn.init.calculate_gain("relu"))
        self.dropout = nn.Dropout(dropout)

    def message(self, edges):
        m = self.linear_r(edges.src["h"], edges.data["etype"], self.presorted)
        if "norm" in edges.data:
            m = m * edges.data["norm"]
        return {"m": m}

    def forward(self, g, feat, etypes, norm=None, *, presorted=False):
        self.presorted = presorted
        with g.local_scope():
            g.srcdata["h"] = feat
            if norm is not None:
                g.edata["norm"] = norm
            g.edata["etype"] = etypes if isinstance(etypes, torch.Tensor) else torch.as_tensor(etypes, device=feat.device)
            g.update_all(self.message, fn.sum("m", "h"))
            h = g.dstdata["h"]
            if self.layer_norm:
                h = self.layer_norm_weight(h)
            if self.bias:
                h = h + self.h_bias
            if self.self_loop:
                h = h + feat[: g.num_dst_nodes()] @ self.loop_weight
            if self.activation:
                h = self.activation(h)
            return self.dropout(h)


def _max_reduce(inputs, dim):
    return torch.max(inputs, dim=dim)[0]


def _min_reduce(inputs, dim):
    return torch.min(inputs, dim=dim)[0]


def _sum_reduce(inputs, dim):
    return torch.sum(inputs, dim=dim)


def _mean_reduce(inputs, dim):
    return torch.mean(inputs, dim=dim)


def _stack_agg(inputs, dsttype):      # noqa: ARG001
    return torch.stack(inputs, dim=1) if inputs else None


def _agg(inputs, dsttype, fn_):       # noqa: ARG001
    return fn_(torch.stack(inputs, dim=0), 0) if inputs else None


def get_aggregate_fn(agg):
    """hetero.py:253-287 (module-level functions + ``partial``: the layer pickles)."""
    from functools import partial

    table = {"sum": _sum_reduce, "max": _max_reduce, "min": _min_reduce, "mean": _mean_reduce}
    if agg == "stack":
        return _stack_agg
    if agg in table:
        return partial(_agg, fn_=table[agg])
    raise DGLAMDError('Invalid cross type aggregator. Must be one of "sum", "max", "min", "mean" or "stack". '
                      "But got {!r}".format(agg))


class HeteroGraphConv(nn.Module):
    """One module per relation, run on that relation's slice of the graph, results aggregated per destination type
    (hetero.py:12-222).  With the layers of this file every slice ends in one g-SpMM launch."""

    def __init__(self, mods, aggregate="sum"):
        super().__init__()
        self.mod_dict = mods
        self.mods = nn.ModuleDict({str(k): v for k, v in mods.items()})
        for v in self.mods.values():
            setter = getattr(v, "set_allow_zero_in_degree", None)
            if callable(setter):
                setter(True)
        self.agg_fn = get_aggregate_fn(aggregate) if isinstance(aggregate, str) else aggregate

    def _get_module(self, etype):
        mod = self.mod_dict.get(etype, None)
        if mod is not None:
            return mod
        if isinstance(etype, tuple):
            return self.mod_dict[etype[1]]
        raise KeyError("Cannot find module with edge type %s" % (etype,))

    def forward(self, g, inputs, mod_args=None, mod_kwargs=None):
        mod_args, mod_kwargs = mod_args or {}, mod_kwargs or {}
        outputs = {nty: [] for nty in g.dsttypes}
        if isinstance(inputs, tuple) or g.is_block:
            if isinstance(inputs, tuple):
                src_inputs, dst_inputs = inputs
            else:
                src_inputs = inputs
                dst_inputs = {k: v[: g.number_of_dst_nodes(k)] for k, v in inputs.items()}
        else:
            src_inputs = dst_inputs = inputs
        for stype, etype, dtype in g.canonical_etypes:
            if stype not in src_inputs or dtype not in dst_inputs:
                continue
            rel_graph = g[stype, etype, dtype]
            dstdata = self._get_module((stype, etype, dtype))(rel_graph, (src_inputs[stype], dst_inputs[dtype]),
                                                              *mod_args.get(etype, ()), **mod_kwargs.get(etype, {}))
            outputs[dtype].append(dstdata)
        return {nty: self.agg_fn(alist, nty) for nty, alist in outputs.items() if len(alist) != 0}


class HeteroLinear(nn.Module):
    """One linear map per key (hetero.py:290-342)."""

    def __init__(self, in_size, out_size, bias=True):
        super().__init__()
        self.linears = nn.ModuleDict({str(typ): nn.Linear(size, out_size, bias=bias) for typ, size in in_size.items()})

    def forward(self, feat):
        return {typ: self.linears[str(typ)](typ_feat) for typ, typ_feat in feat.items()}


class HeteroEmbedding(nn.Module):
    """One embedding table per key (hetero.py:345-430)."""

    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.embeds = nn.ModuleDict({str(k): nn.Embedding(n, embedding_dim) for k, n in num_embeddings.items()})
        self.raw_keys = {str(k): k for k in num_embeddings}

    @property
    def weight(self):
        return {self.raw_keys[typ]: emb.weight for typ, emb in self.embeds.items()}

    def reset_parameters(self):
        for emb in self.embeds.values():
            nn.init.xavier_uniform_(emb.weight)

    def forward(self, input_ids):
        return {typ: self.embeds[str(typ)](ids) for typ, ids in input_ids.items()}


from .transforms import EdgeWeightNorm  # noqa: E402,F401  (graphconv.py:17-130; lives with the other graph helpers)
