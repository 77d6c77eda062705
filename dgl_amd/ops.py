"""Operator API of the hot path: the ``dgl.ops`` surface for g-SpMM / g-SDDMM / edge softmax.

Same names, argument meaning and semantics as python/dgl/ops/spmm.py (``gspmm`` :39-116 and
the generated ``u_mul_e_sum`` ... ``copy_e_mean`` aliases :224-241), python/dgl/ops/sddmm.py
(``gsddmm`` :40-98, ``u_add_v`` ... ``copy_e`` :146-207) and python/dgl/ops/edge_softmax.py
(:12).  ``g`` is a :class:`dgl_amd.heterograph.DGLGraph`.
"""
import sys

import torch

from . import autograd as _F
from ._lib import DGLAMDError

__all__ = ["gspmm", "gsddmm", "edge_softmax", "copy_u", "copy_v", "copy_e", "gat_attention", "gat_attention_applies"]


def _reshape_for_broadcast(op, lhs, rhs):
    """Pad the shorter feature shape with 1s right after the node/edge axis so both operands
    have the same rank (ops/spmm.py:13-36); `dot` keeps the last (reduced) axis aligned."""
    ls, rs = lhs.shape, rhs.shape
    if len(ls) == len(rs):
        return lhs, rhs
    if len(ls) < len(rs):
        lhs = lhs.reshape((ls[0],) + (1,) * (len(rs) - len(ls)) + tuple(ls[1:]))
    else:
        rhs = rhs.reshape((rs[0],) + (1,) * (len(ls) - len(rs)) + tuple(rs[1:]))
    return lhs, rhs


def gspmm(g, op, reduce_op, lhs_data, rhs_data):
    r"""Generalized SpMM: messages ``op(lhs[src], rhs[edge])`` reduced into destination nodes.

    op in {add, sub, mul, div, copy_lhs, copy_rhs}; reduce_op in {sum, max, min, mean}.
    For graphs with several relations pass tuples/dicts as the reference does (lhs indexed by
    source node type, rhs by edge type)."""
    if op not in ("add", "sub", "mul", "div", "copy_lhs", "copy_rhs"):
        raise DGLAMDError("Unsupported SpMM binary operator: {}".format(op))
    if reduce_op not in ("sum", "max", "min", "mean"):
        raise DGLAMDError("Unsupported SpMM reducer: {}".format(reduce_op))
    gidx = g._graph
    if gidx.number_of_etypes() == 1:
        if op not in ("copy_lhs", "copy_rhs"):
            lhs_data, rhs_data = _reshape_for_broadcast(op, lhs_data, rhs_data)
        if reduce_op == "mean" and gidx.relations[0].allowed("csc"):
            # the division by clamp(in_degree, 1) happens inside the kernel (DGLA_MEAN)
            return _F.gspmm_mean(gidx, op, lhs_data, rhs_data, g.in_degrees())
        ret = _F.gspmm(gidx, op, "sum" if reduce_op == "mean" else reduce_op, lhs_data, rhs_data)
    else:
        lhs_t = _to_type_tuple(g, lhs_data, "ntype") if op != "copy_rhs" else ()
        rhs_t = _to_type_tuple(g, rhs_data, "etype") if op != "copy_lhs" else ()
        if op not in ("copy_lhs", "copy_rhs"):
            lhs_t, rhs_t = list(lhs_t), list(rhs_t)
            for et in range(gidx.number_of_etypes()):
                s, _ = gidx.metagraph.find_edge(et)
                if lhs_t[s] is not None and rhs_t[et] is not None:
                    lhs_t[s], rhs_t[et] = _reshape_for_broadcast(op, lhs_t[s], rhs_t[et])
        ret = _F.gspmm_hetero(gidx, op, "sum" if reduce_op == "mean" else reduce_op,
                              len(lhs_t), *(tuple(lhs_t) + tuple(rhs_t)))
    if reduce_op == "mean":
        # mean = sum / clamp(in_degree, 1)  (ops/spmm.py:109-114)
        def div(x, deg):
            shp = (x.shape[0],) + (1,) * (x.dim() - 1)
            return x / deg.to(x.dtype).clamp(min=1).reshape(shp)

        if gidx.number_of_etypes() == 1:
            ret = div(ret, g.in_degrees())
        else:
            raise DGLAMDError("Reduce op 'mean' is not supported on graphs with several relations "
                              "(python/dgl/heterograph.py:5133-5138).")
    return ret


def gsddmm(g, op, lhs_data, rhs_data, lhs_target="u", rhs_target="v"):
    r"""Generalized SDDMM: ``out[e] = op(lhs[target_l(e)], rhs[target_r(e)])``.

    op in {add, sub, mul, div, dot, copy_lhs, copy_rhs}; targets in {u, v, e}."""
    if op not in ("add", "sub", "mul", "div", "dot", "copy_lhs", "copy_rhs"):
        raise DGLAMDError("Unsupported SDDMM binary operator: {}".format(op))
    if lhs_target not in "uve" or rhs_target not in "uve":
        raise DGLAMDError("targets must be one of 'u', 'v', 'e'")
    gidx = g._graph
    if gidx.number_of_etypes() == 1:
        if op not in ("copy_lhs", "copy_rhs"):
            lhs_data, rhs_data = _reshape_for_broadcast(op, lhs_data, rhs_data)
        return _F.gsddmm(gidx, op, lhs_data, rhs_data, lhs_target, rhs_target)
    kind = lambda t: "etype" if t == "e" else "ntype"
    lhs_t = _to_type_tuple(g, lhs_data, kind(lhs_target)) if op != "copy_rhs" else ()
    rhs_t = _to_type_tuple(g, rhs_data, kind(rhs_target)) if op != "copy_lhs" else ()
    return _F.gsddmm_hetero(gidx, op, len(lhs_t), lhs_target, rhs_target,
                            *(tuple(lhs_t) + tuple(rhs_t)))


def _to_type_tuple(g, data, kind):
    """dict keyed by type name (or tuple/list already in type-id order) -> tuple in id order."""
    if isinstance(data, (tuple, list)):
        return tuple(data)
    n = len(g.ntypes) if kind == "ntype" else len(g.canonical_etypes)
    out = [None] * n
    for k, v in data.items():
        idx = g.get_ntype_id(k) if kind == "ntype" else g.get_etype_id(k)
        out[idx] = v
    return tuple(out)


def edge_softmax(graph, logits, eids=None, norm_by="dst"):
    r"""Softmax of edge scores over the edges that share a destination (or source) node
    (python/dgl/ops/edge_softmax.py:12-140)."""
    if norm_by not in ("dst", "src"):
        raise DGLAMDError("norm_by must be 'dst' or 'src'")
    gidx = graph._graph
    if gidx.number_of_etypes() == 1:
        if eids is not None and not isinstance(eids, torch.Tensor):
            eids = torch.as_tensor(eids)
        return _F.edge_softmax(gidx, logits, eids, norm_by)
    # several relations: the normaliser runs over ALL incoming edges of a node regardless of
    # relation (EdgeSoftmax_hetero, sparse.py:750-850): concatenate per destination type
    scores = _to_type_tuple(graph, logits, "etype")
    if eids is not None:
        # the reference hands ``eids`` to ``gidx.edge_subgraph([eids], True)`` (sparse.py:771-772), i.e. ONE id array
        # per relation: here a dict {etype: ids} / a sequence in edge-type order (None = all edges of that type); a
        # single tensor is accepted when exactly one relation carries a score.  The softmax then runs on the
        # edge-induced subgraph (same nodes), scores listed in the order of the ids.
        if isinstance(eids, torch.Tensor):
            with_score = [et for et, sc in enumerate(scores) if sc is not None]
            if len(with_score) != 1:
                raise DGLAMDError("edge_softmax on a graph with several relations: give eids as a dict "
                                  "{edge type: ids} (a single id tensor is ambiguous)")
            per = [None] * len(scores)
            per[with_score[0]] = eids
        elif isinstance(eids, dict):
            per = [None] * len(scores)
            for k, v in eids.items():
                per[graph.get_etype_id(k)] = v
        else:
            per = list(eids) + [None] * (len(scores) - len(eids))
        full = []
        for et, ids in enumerate(per):
            if ids is None:
                ids = torch.arange(gidx.num_edges(et), dtype=gidx.dtype, device=gidx.ctx)
            elif not isinstance(ids, torch.Tensor):
                ids = torch.as_tensor(ids)
            full.append(ids.to(device=gidx.ctx, dtype=gidx.dtype))
        gidx = gidx.edge_subgraph(full)
    outs = _F.edge_softmax_hetero(gidx, None, norm_by, *scores)
    if isinstance(logits, dict):
        return {graph.canonical_etypes[et]: o for et, o in enumerate(outs) if o is not None}
    return tuple(outs)


# ---- GAT attention block as one operator ------------------------------------------------
def gat_attention_applies(graph, ft, el, er):
    """Whether the one-pass kernel takes this call: one relation with an in-edge CSC, fp32 operands on the GPU of
    shapes (N_src, H, D), (N_src, H, 1), (N_dst, H, 1) with D a power of two >= 4 and H * D <= 256 (one 16-byte slab
    per lane); anything else runs the composed operators (dgl_amd.nn.gat_attention)."""
    gidx = graph._graph
    if gidx.number_of_etypes() != 1 or not gidx.relations[0].allowed("csc"):
        return False
    if "dgl_amd._CAPI_GATAttentionForward" not in _registered():
        return False
    if ft.dim() != 3 or el.dim() != 3 or er.dim() != 3 or el.shape[2] != 1 or er.shape[2] != 1:
        return False
    if not (ft.dtype == el.dtype == er.dtype == torch.float32) or type(ft) is not torch.Tensor:
        return False
    h, d = int(ft.shape[1]), int(ft.shape[2])
    return (ft.is_cuda and el.shape[1] == h and er.shape[1] == h and d >= 4 and d & (d - 1) == 0 and h * d <= 256 and
            ft.shape[0] == el.shape[0] == graph.num_src_nodes() and er.shape[0] == graph.num_dst_nodes())


_registry_names = []


def _registered():
    if not _registry_names:
        from . import _ffi

        _registry_names.append(frozenset(_ffi.list_global_func_names()))
    return _registry_names[0]


def gat_attention(graph, ft, el, er, negative_slope=0.2):
    """``out[v] = sum_{u->v} softmax_v(leaky_relu(el[u] + er[v])) * ft[u]`` per head in ONE pass over the in-edges
    (csrc/gat_attention.hip): no (E, H) tensor is written or read; the backward recomputes the attention weights from
    the per-row (max, sum) the forward saved.  The composition it replaces: gatconv.py:330-347."""
    if not gat_attention_applies(graph, ft, el, er):
        raise DGLAMDError("gat_attention: the fused kernel does not take these operands (dgl_amd.nn.gat_attention "
                          "falls back to the composed operators)")
    return _F.gat_attention(graph._graph, ft, el, er, float(negative_slope))


# ---- generated aliases -----------------------------------------------------------------
def _make_spmm(binary, reduce_op):
    name = "u_{}_e_{}".format(binary, reduce_op)

    def f(g, x, y):
        return gspmm(g, binary, reduce_op, x, y)

    f.__name__ = name
    f.__doc__ = "Generalized SpMM: message = u {} e, reducer = {} (ops/spmm.py:119-170).".format(
        binary, reduce_op)
    return name, f


def _make_copy_spmm(which, reduce_op):
    name = "copy_{}_{}".format(which, reduce_op)
    op = "copy_lhs" if which == "u" else "copy_rhs"

    def f(g, x):
        return gspmm(g, op, reduce_op, x if which == "u" else None, None if which == "u" else x)

    f.__name__ = name
    f.__doc__ = "Generalized SpMM: message = copy_{}, reducer = {} (ops/spmm.py:173-221).".format(
        which, reduce_op)
    return name, f


def _make_sddmm(lhs, binary, rhs):
    name = "{}_{}_{}".format(lhs, binary, rhs)

    def f(g, x, y):
        return gsddmm(g, binary, x, y, lhs_target=lhs, rhs_target=rhs)

    f.__name__ = name
    f.__doc__ = "Generalized SDDMM: out[e] = {} {} {} (ops/sddmm.py:101-143).".format(lhs, binary, rhs)
    return name, f


_mod = sys.modules[__name__]
for _b in ("add", "sub", "mul", "div"):
    for _r in ("sum", "max", "min", "mean"):
        _n, _f = _make_spmm(_b, _r)
        setattr(_mod, _n, _f)
        __all__.append(_n)
for _w in ("u", "e"):
    for _r in ("sum", "max", "min", "mean"):
        _n, _f = _make_copy_spmm(_w, _r)
        setattr(_mod, _n, _f)
        __all__.append(_n)
for _l in "uve":
    for _r in "uve":
        for _b in ("add", "sub", "mul", "div", "dot"):
            _n, _f = _make_sddmm(_l, _b, _r)
            setattr(_mod, _n, _f)
            __all__.append(_n)


def copy_u(g, x):
    """Edge feature = source node feature (ops/sddmm.py:146-165)."""
    return gsddmm(g, "copy_lhs", x, None)


def copy_v(g, x):
    """Edge feature = destination node feature (ops/sddmm.py:168-187)."""
    return gsddmm(g, "copy_rhs", None, x)


def copy_e(g, x):
    """Identity on edge features (ops/sddmm.py:190-207)."""
    return x


# python/dgl/ops/__init__.py also re-exports the segment and gather / segment matmul operators
# (ops/segment.py:6, ops/gather_mm.py:6): dgl.ops.segment_reduce, segment_softmax, segment_mm, gather_mm
from .mm import gather_mm, segment_mm  # noqa: E402,F401
from .segment import segment_reduce, segment_softmax  # noqa: E402,F401

__all__ += ["segment_reduce", "segment_softmax", "segment_mm", "gather_mm"]
