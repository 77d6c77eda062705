"""A thin ``DGLGraph`` for the message-passing hot path.

Keeps the user-facing calls that reach the g-SpMM / g-SDDMM kernels — ``update_all``
(python/dgl/heterograph.py:5018-5158), ``apply_edges`` (:4597-4712), the feature frames
(``ndata`` / ``edata`` / ``srcdata`` / ``dstdata``), degrees and type lookups — and the routing
of built-in functions to fused operators (python/dgl/core.py:273-425).  The reference's graph
engine (construction from files, sampling, transforms, batching, user-defined functions with
degree bucketing, ...) is out of scope; user-defined message/reduce functions raise.
"""
from collections.abc import MutableMapping
from contextlib import contextmanager

import torch

from . import function as fn
from . import ops
from ._lib import DGLAMDError
from .graph_index import GraphIndex, Relation

__all__ = ["DGLGraph", "graph", "heterograph", "rand_graph", "rand_bipartite", "create_block",
           "reverse"]


class _Frame(dict):
    """Column store of one node / edge type; checks the leading dimension on insert."""

    def __init__(self, num_rows):
        super().__init__()
        self.num_rows = num_rows

    def __setitem__(self, key, val):
        if val.shape[0] != self.num_rows:
            raise DGLAMDError("Expect number of features to match number of nodes/edges (len(u)). "
                              "Got {} and {} instead.".format(val.shape[0], self.num_rows))
        super().__setitem__(key, val)


class _TypedView(MutableMapping):
    """``g.ndata`` / ``g.edata`` / ``g.srcdata`` / ``g.dstdata``: a plain mapping when there is a
    single type on that side, otherwise ``view[key]`` is a dict keyed by type name and
    assignment takes such a dict (python/dgl/view.py)."""

    def __init__(self, frames, names):
        self._frames, self._names = frames, names

    def __getitem__(self, key):
        if len(self._frames) == 1:
            return self._frames[0][key]
        out = {n: f[key] for n, f in zip(self._names, self._frames) if key in f}
        if not out:
            raise KeyError(key)
        return out

    def __setitem__(self, key, val):
        if len(self._frames) == 1:
            self._frames[0][key] = val
            return
        if not isinstance(val, dict):
            raise DGLAMDError("Current graph has more than one type; please give a dict of type -> tensor.")
        for n, v in val.items():
            self._frames[self._names.index(n)][key] = v

    def __delitem__(self, key):
        for f in self._frames:
            f.pop(key, None)

    def __iter__(self):
        seen = []
        for f in self._frames:
            for k in f:
                if k not in seen:
                    seen.append(k)
        return iter(seen)

    def __len__(self):
        return len(list(iter(self)))


class _TypeIndexer:
    def __init__(self, g, kind):
        self._g, self._kind = g, kind

    def __getitem__(self, key):
        g = self._g
        if self._kind == "node":
            f = g._node_frames[g.get_ntype_id(key)]
        elif self._kind == "srcnode":
            f = g._node_frames[g.get_ntype_id_from_src(key)]
        elif self._kind == "dstnode":
            f = g._node_frames[g.get_ntype_id_from_dst(key)]
        else:
            f = g._edge_frames[g.get_etype_id(key)]
        return type("_DataHolder", (), {"data": f})()


class DGLGraph:
    is_block = False

    def __init__(self, gidx, ntypes, canonical_etypes, node_frames=None, edge_frames=None,
                 src_ntypes=None, dst_ntypes=None):
        self._graph = gidx
        self._ntypes = list(ntypes)
        self._canonical_etypes = [tuple(c) for c in canonical_etypes]
        self._node_frames = node_frames or [_Frame(gidx.num_nodes(i)) for i in range(len(ntypes))]
        self._edge_frames = edge_frames or [_Frame(gidx.num_edges(i)) for i in range(len(canonical_etypes))]
        # blocks / uni-bipartite graphs keep separate source and destination type lists
        self._src_ntype_ids = src_ntypes
        self._dst_ntype_ids = dst_ntypes

    # ---- types ---------------------------------------------------------------------
    @property
    def ntypes(self):
        return list(self._ntypes)

    @property
    def etypes(self):
        return [c[1] for c in self._canonical_etypes]

    @property
    def canonical_etypes(self):
        return list(self._canonical_etypes)

    @property
    def is_unibipartite(self):
        return self._src_ntype_ids is not None

    @property
    def srctypes(self):
        ids = self._src_ntype_ids if self.is_unibipartite else range(len(self._ntypes))
        return [self._ntypes[i] for i in ids]

    @property
    def dsttypes(self):
        ids = self._dst_ntype_ids if self.is_unibipartite else range(len(self._ntypes))
        return [self._ntypes[i] for i in ids]

    def get_ntype_id(self, ntype):
        if ntype is None:
            if len(self._ntypes) != 1:
                raise DGLAMDError("Node type name must be specified if there are more than one node types.")
            return 0
        if ntype not in self._ntypes:
            raise DGLAMDError('Node type "{}" does not exist.'.format(ntype))
        if self.is_unibipartite and self._ntypes.count(ntype) > 1:
            raise DGLAMDError('Node type "{}" is ambiguous in a block; use srcdata / dstdata.'.format(ntype))
        return self._ntypes.index(ntype)

    def get_ntype_id_from_src(self, ntype):
        ids = self._src_ntype_ids if self.is_unibipartite else list(range(len(self._ntypes)))
        if ntype is None:
            if len(ids) != 1:
                raise DGLAMDError("SRC node type name must be specified if there are more than one SRC node types.")
            return ids[0]
        for i in ids:
            if self._ntypes[i] == ntype:
                return i
        raise DGLAMDError('SRC node type "{}" does not exist.'.format(ntype))

    def get_ntype_id_from_dst(self, ntype):
        ids = self._dst_ntype_ids if self.is_unibipartite else list(range(len(self._ntypes)))
        if ntype is None:
            if len(ids) != 1:
                raise DGLAMDError("DST node type name must be specified if there are more than one DST node types.")
            return ids[0]
        for i in ids:
            if self._ntypes[i] == ntype:
                return i
        raise DGLAMDError('DST node type "{}" does not exist.'.format(ntype))

    def to_canonical_etype(self, etype):
        if etype is None:
            if len(self._canonical_etypes) != 1:
                raise DGLAMDError("Edge type name must be specified if there are more than one edge types.")
            return self._canonical_etypes[0]
        if isinstance(etype, tuple):
            if etype not in self._canonical_etypes:
                raise DGLAMDError('Edge type "{}" does not exist.'.format(etype))
            return etype
        hits = [c for c in self._canonical_etypes if c[1] == etype]
        if len(hits) != 1:
            raise DGLAMDError('Edge type "{}" {}.'.format(etype, "does not exist" if not hits else "is ambiguous"))
        return hits[0]

    def get_etype_id(self, etype):
        return self._canonical_etypes.index(self.to_canonical_etype(etype))

    # ---- sizes / properties ----------------------------------------------------------
    @property
    def idtype(self):
        return self._graph.dtype

    @property
    def device(self):
        return self._graph.ctx

    def num_nodes(self, ntype=None):
        if ntype is None:
            return sum(self._graph.num_nodes(i) for i in range(len(self._ntypes)))
        return self._graph.num_nodes(self.get_ntype_id(ntype))

    number_of_nodes = num_nodes

    def num_src_nodes(self, ntype=None):
        if ntype is None:
            ids = self._src_ntype_ids if self.is_unibipartite else range(len(self._ntypes))
            return sum(self._graph.num_nodes(i) for i in ids)
        return self._graph.num_nodes(self.get_ntype_id_from_src(ntype))

    number_of_src_nodes = num_src_nodes

    def num_dst_nodes(self, ntype=None):
        if ntype is None:
            ids = self._dst_ntype_ids if self.is_unibipartite else range(len(self._ntypes))
            return sum(self._graph.num_nodes(i) for i in ids)
        return self._graph.num_nodes(self.get_ntype_id_from_dst(ntype))

    number_of_dst_nodes = num_dst_nodes

    def num_edges(self, etype=None):
        if etype is None:
            return sum(self._graph.num_edges(i) for i in range(len(self._canonical_etypes)))
        return self._graph.num_edges(self.get_etype_id(etype))

    number_of_edges = num_edges

    def in_degrees(self, v=None, etype=None):
        d = self._graph.relations[self.get_etype_id(etype)].in_degrees()
        return d if v is None else d[v]

    def out_degrees(self, u=None, etype=None):
        d = self._graph.relations[self.get_etype_id(etype)].out_degrees()
        return d if u is None else d[u]

    @property
    def edges(self):
        """``g.edges(etype=...)`` -> (src, dst) in edge-id order; ``g.edges[etype].data`` -> frame."""
        return _EdgeView(self)

    def formats(self, formats=None):
        """Restrict the allowed sparse formats (``g.formats(['csr'])``), like the reference."""
        if formats is None:
            return {"created": [f for f in ("coo", "csr", "csc") if all(r.has(f) for r in self._graph.relations)],
                    "not created": []}
        if isinstance(formats, str):
            formats = [formats]
        rels = []
        for r in self._graph.relations:
            keep = {f: getattr(r, f)() for f in formats if f in ("csr", "csc")}
            nr = Relation(r.num_src, r.num_dst, csr=keep.get("csr"), csc=keep.get("csc"),
                          idtype=r.idtype, device=r.device, formats=tuple(formats))
            if "coo" in formats or not keep:
                nr._coo = r.coo()
            rels.append(nr)
        gidx = GraphIndex([self._graph.num_nodes(i) for i in range(len(self._ntypes))],
                          self._graph.metagraph.edges, rels)
        return DGLGraph(gidx, self._ntypes, self._canonical_etypes, self._node_frames,
                        self._edge_frames, self._src_ntype_ids, self._dst_ntype_ids)

    def to(self, device):
        device = torch.device(device)
        gidx = GraphIndex([self._graph.num_nodes(i) for i in range(len(self._ntypes))],
                          self._graph.metagraph.edges, [r.to(device) for r in self._graph.relations])
        mv = lambda frames: [self._copy_frame(f, lambda t: t.to(device)) for f in frames]
        return DGLGraph(gidx, self._ntypes, self._canonical_etypes, mv(self._node_frames),
                        mv(self._edge_frames), self._src_ntype_ids, self._dst_ntype_ids)

    def astype(self, idtype):
        if idtype == self.idtype:
            return self
        gidx = GraphIndex([self._graph.num_nodes(i) for i in range(len(self._ntypes))],
                          self._graph.metagraph.edges, [r.astype(idtype) for r in self._graph.relations])
        return DGLGraph(gidx, self._ntypes, self._canonical_etypes, self._node_frames,
                        self._edge_frames, self._src_ntype_ids, self._dst_ntype_ids)

    def int(self):
        return self.astype(torch.int32)

    def long(self):
        return self.astype(torch.int64)

    @staticmethod
    def _copy_frame(f, fn_=lambda t: t):
        nf = _Frame(f.num_rows)
        for k, v in f.items():
            dict.__setitem__(nf, k, fn_(v))
        return nf

    # ---- feature access ----------------------------------------------------------------
    @property
    def ndata(self):
        return _TypedView(self._node_frames, self._ntypes)

    @property
    def edata(self):
        return _TypedView(self._edge_frames, self._canonical_etypes)

    @property
    def srcdata(self):
        ids = self._src_ntype_ids if self.is_unibipartite else list(range(len(self._ntypes)))
        return _TypedView([self._node_frames[i] for i in ids], [self._ntypes[i] for i in ids])

    @property
    def dstdata(self):
        ids = self._dst_ntype_ids if self.is_unibipartite else list(range(len(self._ntypes)))
        return _TypedView([self._node_frames[i] for i in ids], [self._ntypes[i] for i in ids])

    @property
    def nodes(self):
        return _TypeIndexer(self, "node")

    @property
    def srcnodes(self):
        return _TypeIndexer(self, "srcnode")

    @property
    def dstnodes(self):
        return _TypeIndexer(self, "dstnode")

    def _edge_indexer(self):
        return _TypeIndexer(self, "edge")

    @contextmanager
    def local_scope(self):
        """Feature changes inside the scope are discarded on exit (heterograph.py local_scope)."""
        old_n, old_e = self._node_frames, self._edge_frames
        self._node_frames = [self._copy_frame(f) for f in old_n]
        self._edge_frames = [self._copy_frame(f) for f in old_e]
        try:
            yield
        finally:
            self._node_frames, self._edge_frames = old_n, old_e

    def local_var(self):
        return DGLGraph(self._graph, self._ntypes, self._canonical_etypes,
                        [self._copy_frame(f) for f in self._node_frames],
                        [self._copy_frame(f) for f in self._edge_frames],
                        self._src_ntype_ids, self._dst_ntype_ids)

    def __getitem__(self, key):
        """``g[etype]``: the relation slice sharing feature storage (heterograph.py __getitem__)."""
        cet = self.to_canonical_etype(key)
        et = self._canonical_etypes.index(cet)
        s, d = self._graph.metagraph.find_edge(et)
        sub = self._graph.get_relation_graph(et)
        if s == d and not self.is_unibipartite:
            return DGLGraph(sub, [cet[0]], [cet], [self._node_frames[s]], [self._edge_frames[et]])
        return DGLGraph(sub, [cet[0], cet[2]], [cet], [self._node_frames[s], self._node_frames[d]],
                        [self._edge_frames[et]], [0], [1])

    # ---- message passing -----------------------------------------------------------------
    def apply_edges(self, func, edges=None, etype=None):
        """Write ``func`` of the end points / edge features of every edge into ``edata``
        (python/dgl/heterograph.py:4597-4712, built-in functions only)."""
        if edges is not None:
            raise DGLAMDError("apply_edges on an edge subset is outside the accelerated path")
        _require_builtin(func)
        if etype is None and len(self._canonical_etypes) > 1:
            out = _invoke_gsddmm(self, func)
            for et, cet in enumerate(self._canonical_etypes):
                for k, v in out.items():
                    if v[et] is not None:
                        self._edge_frames[et][k] = v[et]
            return
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        for k, v in _invoke_gsddmm(g, func).items():
            g._edge_frames[0][k] = v

    def update_all(self, message_func, reduce_func, apply_node_func=None, etype=None):
        """Send messages along all edges and reduce them at the destination nodes
        (python/dgl/heterograph.py:5018-5158)."""
        if apply_node_func is not None:
            raise DGLAMDError("apply_node_func (a user-defined function) is outside the accelerated path")
        _require_builtin(message_func)
        _require_builtin(reduce_func)
        if etype is None and len(self._canonical_etypes) > 1:
            if reduce_func.name == "mean":
                raise NotImplementedError(
                    "Cannot set both intra-type and inter-type reduce operators as 'mean' using "
                    "update_all. Please use multi_update_all instead.")
            out = _message_passing(self, message_func, reduce_func)
            for key, per_type in out.items():
                for d, val in enumerate(per_type):
                    if val is None:
                        continue
                    if reduce_func.name in ("max", "min"):
                        val = _replace_inf_with_zero(val)
                    self._node_frames[d][key] = val
            return
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        ndata = _message_passing(g, message_func, reduce_func)
        dst_frame = g._node_frames[g.get_ntype_id_from_dst(None)]
        for k, v in ndata.items():
            if reduce_func.name in ("max", "min"):
                v = _replace_inf_with_zero(v)  # heterograph.py:5115-5122
            dst_frame[k] = v

    def multi_update_all(self, etype_dict, cross_reducer, apply_node_func=None):
        """Per-relation update_all followed by a cross-relation reducer
        (python/dgl/heterograph.py multi_update_all); built-ins only."""
        if apply_node_func is not None:
            raise DGLAMDError("apply_node_func is outside the accelerated path")
        if cross_reducer not in ("sum", "min", "max", "mean", "stack"):
            raise DGLAMDError("Invalid cross type reducer. Must be one of 'sum', 'min', 'max', 'mean' or 'stack'.")
        collected = {}
        for etype, (mfunc, rfunc) in etype_dict.items():
            g = self[etype]
            cet = self.to_canonical_etype(etype)
            d = self.get_ntype_id(cet[2])
            nd = _message_passing(g, mfunc, rfunc)
            for k, v in nd.items():
                if rfunc.name in ("max", "min"):
                    v = _replace_inf_with_zero(v)
                collected.setdefault((d, k), []).append(v)
        for (d, k), vals in collected.items():
            st = torch.stack(vals, 0 if cross_reducer != "stack" else 1)
            if cross_reducer == "sum":
                st = st.sum(0)
            elif cross_reducer == "mean":
                st = st.mean(0)
            elif cross_reducer == "max":
                st = st.max(0)[0]
            elif cross_reducer == "min":
                st = st.min(0)[0]
            self._node_frames[d][k] = st

    def __repr__(self):
        return "DGLGraph(num_nodes={}, num_edges={}, ntypes={}, etypes={})".format(
            {n: self._graph.num_nodes(i) for i, n in enumerate(self._ntypes)},
            {c: self._graph.num_edges(i) for i, c in enumerate(self._canonical_etypes)},
            self._ntypes, self.etypes)


class _EdgeView:
    def __init__(self, g):
        self._g = g

    def __call__(self, etype=None):
        g = self._g
        row, col, old = g._graph.relations[g.get_etype_id(etype)].coo()
        if old is not None:
            order = torch.argsort(old)
            return row[order], col[order]
        return row, col

    def __getitem__(self, key):
        return self._g._edge_indexer()[key]


def _require_builtin(func):
    if not isinstance(func, fn.BuiltinFunction):
        raise DGLAMDError(
            "Only built-in functions (dgl_amd.function.*) are supported on the accelerated "
            "message-passing path; user-defined functions use the reference's degree-bucketing "
            "executor (python/dgl/core.py:99-174), which is out of scope here.")


def _replace_inf_with_zero(x):
    return torch.where(torch.isinf(x), torch.zeros_like(x), x)


def _field(g, code, name):
    view = [g.srcdata, g.dstdata, g.edata][code]
    return view[name]


def _as_tuple(g, data, target):
    """Feature of a multi-relation graph as a tuple in type-id order (core.data_dict_to_list)."""
    if not isinstance(data, dict):
        return data
    if target == "e":
        out = [None] * len(g.canonical_etypes)
        for k, v in data.items():
            out[g.get_etype_id(k)] = v
    else:
        out = [None] * len(g.ntypes)
        for k, v in data.items():
            out[g.get_ntype_id(k)] = v
    return tuple(out)


def _invoke_gsddmm(g, func):
    multi = g._graph.number_of_etypes() > 1
    if isinstance(func, fn.BinaryMessageFunction):
        x, y = _field(g, func.lhs, func.lhs_field), _field(g, func.rhs, func.rhs_field)
        lt, _, rt = func.name.split("_", 2)
        if multi:
            x, y = _as_tuple(g, x, lt), _as_tuple(g, y, rt)
        z = getattr(ops, func.name)(g, x, y)
    else:
        x = _field(g, func.target, func.in_field)
        if multi:
            x = _as_tuple(g, x, "u" if func.name == "copy_u" else "e")
        z = getattr(ops, func.name)(g, x)
    return {func.out_field: z}


def _invoke_gspmm(g, mfunc, rfunc, edata=None):
    if mfunc.out_field != rfunc.msg_field:
        raise DGLAMDError(
            "Invalid message ({}) and reduce ({}) function pairs. The output field of the message "
            "function must be equal to the message field of the reduce function.".format(mfunc, rfunc))
    multi = g._graph.number_of_etypes() > 1
    views = [g.srcdata, g.dstdata, g.edata if edata is None else edata]
    op = getattr(ops, "{}_{}".format(mfunc.name, rfunc.name))
    if isinstance(mfunc, fn.BinaryMessageFunction):
        x, y = views[mfunc.lhs][mfunc.lhs_field], views[mfunc.rhs][mfunc.rhs_field]
        if multi:
            lt, _, rt = mfunc.name.split("_", 2)
            x, y = _as_tuple(g, x, lt), _as_tuple(g, y, rt)
        z = op(g, x, y)
    else:
        x = views[mfunc.target][mfunc.in_field]
        if multi:
            x = _as_tuple(g, x, "u" if mfunc.name == "copy_u" else "e")
        z = op(g, x)
    return {rfunc.out_field: z}


def _message_passing(g, mfunc, rfunc):
    """Fused when ``ops.<msg>_<reduce>`` exists (copy_u, copy_e, u_{add,sub,mul,div}_e), else
    g-SDDMM to materialise the messages followed by copy_e + reduce (core.py:392-413)."""
    if getattr(ops, "{}_{}".format(mfunc.name, rfunc.name), None) is not None:
        return _invoke_gspmm(g, mfunc, rfunc)
    msg = _invoke_gsddmm(g, mfunc)
    m = rfunc.msg_field
    if mfunc.out_field != m:
        raise DGLAMDError("Invalid message ({}) and reduce ({}) function pairs.".format(mfunc, rfunc))
    return _invoke_gspmm(g, fn.copy_e(m, m), rfunc, edata=msg)


# ---- constructors ----------------------------------------------------------------------
def _as_index(x, idtype, device):
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(x)
    return t.to(device=device, dtype=idtype).contiguous()


def graph(data, num_nodes=None, idtype=None, device=None):
    """Homogeneous graph from ``(src, dst)`` (dgl.graph)."""
    u, v = data
    if idtype is None:
        idtype = u.dtype if isinstance(u, torch.Tensor) and u.dtype in (torch.int32, torch.int64) else torch.int64
    if device is None:
        device = u.device if isinstance(u, torch.Tensor) else torch.device("cpu")
    u, v = _as_index(u, idtype, device), _as_index(v, idtype, device)
    if num_nodes is None:
        num_nodes = int(max(int(u.max()) if u.numel() else -1, int(v.max()) if v.numel() else -1)) + 1
    rel = Relation(num_nodes, num_nodes, u, v, idtype=idtype, device=torch.device(device))
    return DGLGraph(GraphIndex([num_nodes], [(0, 0)], [rel]), ["_N"], [("_N", "_E", "_N")])


def heterograph(data_dict, num_nodes_dict=None, idtype=None, device=None):
    """Graph with several node / edge types from ``{(srctype, etype, dsttype): (src, dst)}``."""
    # relations are ordered by the FULL canonical tuple (create_metagraph_index,
    # python/dgl/heterograph_index.py:1238-1240 `sorted(canonical_etypes)`): edge-type ids,
    # g.canonical_etypes and every per-etype tuple argument follow this order
    cets = sorted(data_dict.keys())
    ntypes = sorted({c[0] for c in cets} | {c[2] for c in cets})
    first = next(iter(data_dict.values()))[0]
    if idtype is None:
        idtype = first.dtype if isinstance(first, torch.Tensor) and first.dtype in (torch.int32, torch.int64) else torch.int64
    if device is None:
        device = first.device if isinstance(first, torch.Tensor) else torch.device("cpu")
    pairs = {c: (_as_index(u, idtype, device), _as_index(v, idtype, device)) for c, (u, v) in data_dict.items()}
    if num_nodes_dict is None:
        num_nodes_dict = {n: 0 for n in ntypes}
        for (s, _, d), (u, v) in pairs.items():
            if u.numel():
                num_nodes_dict[s] = max(num_nodes_dict[s], int(u.max()) + 1)
                num_nodes_dict[d] = max(num_nodes_dict[d], int(v.max()) + 1)
    src_types = {c[0] for c in cets}
    dst_types = {c[2] for c in cets}
    unibipartite = len(src_types & dst_types) == 0 and len(cets) >= 1
    if unibipartite:
        # like the reference, a graph whose source and destination types are disjoint keeps
        # separate SRC / DST type lists (is_unibipartite)
        ntypes = sorted(src_types) + sorted(dst_types)
        src_ids = list(range(len(src_types)))
        dst_ids = list(range(len(src_types), len(ntypes)))
    rels, meta = [], []
    for c in cets:
        u, v = pairs[c]
        s, d = ntypes.index(c[0]), ntypes.index(c[2])
        rels.append(Relation(num_nodes_dict[c[0]], num_nodes_dict[c[2]], u, v, idtype=idtype,
                             device=torch.device(device)))
        meta.append((s, d))
    gidx = GraphIndex([num_nodes_dict[n] for n in ntypes], meta, rels)
    if unibipartite:
        return DGLGraph(gidx, ntypes, cets, src_ntypes=src_ids, dst_ntypes=dst_ids)
    return DGLGraph(gidx, ntypes, cets)


def create_block(data, num_src_nodes, num_dst_nodes, idtype=None, device=None):
    """Bipartite message-flow block ``(src, dst)`` with ``num_src >= num_dst`` (dgl.create_block)."""
    g = heterograph({("_N", "_E", "_N_dst"): data},
                    {"_N": num_src_nodes, "_N_dst": num_dst_nodes}, idtype, device)
    g._ntypes = ["_N", "_N"]
    g._canonical_etypes = [("_N", "_E", "_N")]
    g.is_block = True
    return g


def rand_graph(num_nodes, num_edges, idtype=torch.int64, device="cpu", seed=None):
    gen = torch.Generator()
    if seed is not None:
        gen.manual_seed(seed)
    u = torch.randint(0, num_nodes, (num_edges,), generator=gen)
    v = torch.randint(0, num_nodes, (num_edges,), generator=gen)
    return graph((u, v), num_nodes=num_nodes, idtype=idtype, device=device)


def rand_bipartite(utype, etype, vtype, num_src_nodes, num_dst_nodes, num_edges,
                   idtype=torch.int64, device="cpu", seed=None):
    gen = torch.Generator()
    if seed is not None:
        gen.manual_seed(seed)
    u = torch.randint(0, num_src_nodes, (num_edges,), generator=gen)
    v = torch.randint(0, num_dst_nodes, (num_edges,), generator=gen)
    return heterograph({(utype, etype, vtype): (u, v)},
                       {utype: num_src_nodes, vtype: num_dst_nodes}, idtype, device)


def reverse(g, copy_ndata=True, copy_edata=False):
    """Graph with every edge reversed (dgl.reverse); node features are shared."""
    gidx = g._graph.reverse()
    cets = [(c[2], c[1], c[0]) for c in g.canonical_etypes]
    return DGLGraph(gidx, g.ntypes, cets, g._node_frames if copy_ndata else None,
                    g._edge_frames if copy_edata else None, g._dst_ntype_ids, g._src_ntype_ids)
