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
from . import udf as _udf
from ._lib import DGLAMDError
from .graph_index import GraphIndex, Relation

__all__ = ["DGLGraph", "graph", "heterograph", "rand_graph", "rand_bipartite", "create_block",
           "reverse", "to_homogeneous", "to_heterogeneous", "from_networkx", "NID", "EID", "NTYPE", "ETYPE"]

NID, EID, NTYPE, ETYPE = "_ID", "_ID", "_TYPE", "_TYPE"   # python/dgl/base.py:14-17


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

    def __call__(self, ntype=None):
        """``g.nodes()`` / ``g.srcnodes()`` / ``g.dstnodes()``: the node ids of one type."""
        g = self._g
        tid = {"node": g.get_ntype_id, "srcnode": g.get_ntype_id_from_src,
               "dstnode": g.get_ntype_id_from_dst}[self._kind](ntype)
        return torch.arange(g._graph.num_nodes(tid), dtype=g.idtype, device=g.device)

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
    batch_size = 1          # (``dgl.batch`` sets the number of graphs on its result)

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
        if self.is_unibipartite and isinstance(ntype, str) and ntype[:4] in ("SRC/", "DST/"):
            # "SRC/<type>" / "DST/<type>" name one side of a block (python/dgl/heterograph.py get_ntype_id)
            return (self.get_ntype_id_from_src if ntype[:4] == "SRC/" else self.get_ntype_id_from_dst)(ntype[4:])
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

    def all_edges(self, form="uv", order="eid", etype=None):
        """``g.edges(form, order, etype)`` under its older name (heterograph.py all_edges)."""
        return self.edges(form, order, etype)

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
        return self._same_kind(DGLGraph(gidx, self._ntypes, self._canonical_etypes, self._node_frames,
                                        self._edge_frames, self._src_ntype_ids, self._dst_ntype_ids))

    def to(self, device):
        device = torch.device(device)
        gidx = GraphIndex([self._graph.num_nodes(i) for i in range(len(self._ntypes))],
                          self._graph.metagraph.edges, [r.to(device) for r in self._graph.relations])
        mv = lambda frames: [self._copy_frame(f, lambda t: t.to(device)) for f in frames]
        return self._same_kind(DGLGraph(gidx, self._ntypes, self._canonical_etypes, mv(self._node_frames),
                                        mv(self._edge_frames), self._src_ntype_ids, self._dst_ntype_ids))

    def astype(self, idtype):
        if idtype == self.idtype:
            return self
        gidx = GraphIndex([self._graph.num_nodes(i) for i in range(len(self._ntypes))],
                          self._graph.metagraph.edges, [r.astype(idtype) for r in self._graph.relations])
        return self._same_kind(DGLGraph(gidx, self._ntypes, self._canonical_etypes, self._node_frames,
                                        self._edge_frames, self._src_ntype_ids, self._dst_ntype_ids))

    def int(self):
        return self.astype(torch.int32)

    def long(self):
        return self.astype(torch.int64)

    def _same_kind(self, other):
        """``other`` — a re-hosted / re-typed copy of this graph — is a block / a batch if this one is."""
        if self.is_block:
            other.is_block = True
        for a in ("_batch_num_nodes", "_batch_num_edges", "batch_size"):
            if a in self.__dict__:
                v = getattr(self, a)
                setattr(other, a, {k: t.to(other.device) for k, t in v.items()} if isinstance(v, dict) else v)
        return other

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

    def adj_external(self, transpose=False, ctx=None, scipy_fmt=None, etype=None):
        from .transforms import adj_external
        return adj_external(self, transpose, ctx, scipy_fmt, etype)

    def batch_num_nodes(self, ntype=None):
        """Nodes per batched graph (``dgl.batch``); a graph that is not a batch is a batch of one."""
        b = getattr(self, "_batch_num_nodes", None)
        n = self._ntypes[self.get_ntype_id(ntype)]
        return b[n] if b is not None else torch.tensor([self.num_nodes(n)], dtype=self.idtype, device=self.device)

    def batch_num_edges(self, etype=None):
        b = getattr(self, "_batch_num_edges", None)
        c = self.to_canonical_etype(etype)
        return b[c] if b is not None else torch.tensor([self.num_edges(c)], dtype=self.idtype, device=self.device)

    def local_var(self):
        return self._same_kind(DGLGraph(self._graph, self._ntypes, self._canonical_etypes,
                                        [self._copy_frame(f) for f in self._node_frames],
                                        [self._copy_frame(f) for f in self._edge_frames],
                                        self._src_ntype_ids, self._dst_ntype_ids))

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
    def _ids(self, x):
        """Node / edge ids given as int, list or tensor -> 1-D tensor in the graph's idtype on its device."""
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=self.idtype)
        else:
            t = torch.as_tensor(x, dtype=self.idtype, device=self.device)
        return t.reshape(-1)

    def _set_n_repr(self, ntid, rows, data):
        """Write columns of node type ``ntid``: whole columns (``rows`` None) or the given rows — missing
        columns start from the zero initializer, rows are written out of place (frame.update_row)."""
        frame = self._node_frames[ntid]
        for k, v in data.items():
            if rows is None:
                frame[k] = v
            else:
                col = frame.get(k)
                if col is None or col.shape[1:] != v.shape[1:] or col.dtype != v.dtype:
                    col = torch.zeros((frame.num_rows,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                frame[k] = col.index_copy(0, rows.long(), v)

    def _set_e_repr(self, etid, rows, data):
        frame = self._edge_frames[etid]
        for k, v in data.items():
            if rows is None:
                frame[k] = v
            else:
                col = frame.get(k)
                if col is None or col.shape[1:] != v.shape[1:] or col.dtype != v.dtype:
                    col = torch.zeros((frame.num_rows,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                frame[k] = col.index_copy(0, rows.long(), v)

    def apply_nodes(self, func, v=None, ntype=None):
        """Update node features with a user-defined function (python/dgl/heterograph.py:4533-4595)."""
        ntid = self.get_ntype_id(ntype)
        rows = None if v is None else self._ids(v)
        self._set_n_repr(ntid, rows, _udf.invoke_node_udf(self, func, ntid, nodes=rows))

    def apply_edges(self, func, edges=None, etype=None):
        """Write ``func`` of the end points / edge features of every edge into ``edata``
        (python/dgl/heterograph.py:4597-4712).  Built-in functions run the g-SDDMM kernels, any other
        callable is invoked on an :class:`dgl_amd.udf.EdgeBatch`."""
        builtin = _udf.is_builtin(func)
        if etype is None and len(self._canonical_etypes) > 1:
            if not builtin:
                raise DGLAMDError("User defined functions are not yet supported in apply_edges for heterogeneous "
                                  "graphs. Please use (apply_edges(func), etype = rel) instead.")
            if edges is not None:
                raise DGLAMDError("apply_edges on an edge subset of a multi-relation graph needs an edge type")
            out = _invoke_gsddmm(self, func)
            for et, cet in enumerate(self._canonical_etypes):
                for k, v in out.items():
                    if v[et] is not None:
                        self._edge_frames[et][k] = v[et]
            return
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        eid = None if edges is None else g._parse_edges(edges)
        if builtin:
            if eid is None:
                edata = _invoke_gsddmm(g, func)
            else:
                edata = _invoke_gsddmm(g._edge_subgraph(eid), func)
        else:
            edata = _udf.invoke_edge_udf(g, func, eid)
        g._set_e_repr(0, eid, edata)

    def update_all(self, message_func, reduce_func, apply_node_func=None, etype=None):
        """Send messages along all edges and reduce them at the destination nodes
        (python/dgl/heterograph.py:5018-5158).  A pair of built-in functions runs the fused kernels; any
        other callable takes the reference's edge-batch / degree-bucketing route (:mod:`dgl_amd.udf`)."""
        builtin = _udf.is_builtin(message_func) and _udf.is_builtin(reduce_func)
        if etype is None and len(self._canonical_etypes) > 1:
            if not builtin:
                raise DGLAMDError("User defined functions are not yet supported in update_all for heterogeneous "
                                  "graphs. Please use multi_update_all instead.")
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
            if apply_node_func is not None:
                for d in sorted({self._graph.metagraph.find_edge(et)[1] for et in range(len(self._canonical_etypes))}):
                    self.apply_nodes(apply_node_func, None, self._ntypes[d])
            return
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        ndata = _udf.message_passing(g, message_func, reduce_func, apply_node_func, _message_passing)
        if _udf.is_builtin(reduce_func) and reduce_func.name in ("max", "min") and ndata:
            key = next(iter(ndata))
            ndata[key] = _replace_inf_with_zero(ndata[key])   # heterograph.py:5115-5122 (the first key, as there)
        g._set_n_repr(g.get_ntype_id_from_dst(None), None, ndata)

    def multi_update_all(self, etype_dict, cross_reducer, apply_node_func=None):
        """Per-relation update_all followed by a cross-relation reducer
        (python/dgl/heterograph.py multi_update_all)."""
        if not callable(cross_reducer) and cross_reducer not in ("sum", "min", "max", "mean", "stack"):
            raise DGLAMDError("Invalid cross type reducer. Must be one of 'sum', 'min', 'max', 'mean' or 'stack'.")
        collected, last_rfunc = {}, None
        for etype, args in etype_dict.items():
            args = tuple(args)
            if len(args) not in (2, 3):
                raise DGLAMDError('Invalid arguments for edge type "{}". Should be (msg_func, reduce_func, '
                                  "[apply_node_func])".format(etype))
            mfunc, rfunc, afunc = (args + (None,))[:3]
            last_rfunc = rfunc
            g = self[etype]
            cet = self.to_canonical_etype(etype)
            d = self.get_ntype_id_from_dst(cet[2])     # (on a block the destination side has its own node frames)
            nd = _udf.message_passing(g, mfunc, rfunc, afunc, _message_passing)
            for k, v in nd.items():
                collected.setdefault((d, k), []).append((self.get_etype_id(etype), v))
        touched = set()
        for (d, k), vals in collected.items():
            vals = [v for _, v in (sorted(vals, key=lambda t: t[0]) if cross_reducer == "stack" else vals)]
            if callable(cross_reducer):
                st = cross_reducer(vals)
            elif cross_reducer == "stack":
                st = torch.stack(vals, 1)
            elif len(vals) == 1:
                st = vals[0]
            else:
                st = torch.stack(vals, 0)
                if cross_reducer == "sum":
                    st = st.sum(0)
                elif cross_reducer == "mean":
                    st = st.mean(0)
                elif cross_reducer == "max":
                    st = st.max(0)[0]
                else:
                    st = st.min(0)[0]
            if _udf.is_builtin(last_rfunc) and last_rfunc.name in ("max", "min"):
                st = _replace_inf_with_zero(st)
            self._node_frames[d][k] = st
            touched.add(d)
        if apply_node_func is not None:
            for d in sorted(touched):
                self.apply_nodes(apply_node_func, None, self._ntypes[d])

    # ---- partial message passing: pull / push / send_and_recv (heterograph.py:4714-5016) ----------
    def _parse_edges(self, edges):
        """Edge ids from ``eid`` tensor / list or an ``(u, v)`` pair (utils.parse_edges_arg_to_eid)."""
        if isinstance(edges, tuple) and len(edges) == 2:
            return self.edge_ids(edges[0], edges[1])
        return self._ids(edges)

    def _edge_subgraph(self, eid):
        """Same nodes, the given edges (in that order), their edge features; shares node frames."""
        u, v = self.edges()
        sel = eid.long()
        rel0 = self._graph.relations[0]
        rel = Relation(rel0.num_src, rel0.num_dst, u[sel].contiguous(), v[sel].contiguous(), idtype=self.idtype,
                       device=self.device)
        rel.transient = True
        gidx = GraphIndex([self._graph.num_nodes(i) for i in range(len(self._ntypes))], self._graph.metagraph.edges,
                          [rel])
        ef = _Frame(int(sel.shape[0]))
        for k, col in self._edge_frames[0].items():
            dict.__setitem__(ef, k, col[sel])
        return DGLGraph(gidx, self._ntypes, self._canonical_etypes, self._node_frames, [ef],
                        self._src_ntype_ids, self._dst_ntype_ids)

    def _compute_graph(self, u, v, eid, recv_nodes=None):
        """The bipartite graph of the ACTIVE edges with relabelled end points and the matching feature rows
        (``_create_compute_graph``, heterograph.py:6679-6750): original ids in ``srcdata[NID]`` / ``dstdata[NID]`` /
        ``edata[EID]``.  Returns (graph, receiving node ids)."""
        uniq_s, new_u = torch.unique(u, return_inverse=True)
        base = v if recv_nodes is None else recv_nodes
        uniq_d = torch.unique(base)
        new_v = torch.searchsorted(uniq_d, v)
        rel = Relation(int(uniq_s.shape[0]), int(uniq_d.shape[0]), new_u.to(self.idtype).contiguous(),
                       new_v.to(self.idtype).contiguous(), idtype=self.idtype, device=self.device)
        rel.transient = True
        gidx = GraphIndex([rel.num_src, rel.num_dst], [(0, 1)], [rel])
        s_t, d_t = self.get_ntype_id_from_src(None), self.get_ntype_id_from_dst(None)

        def sub(frame, rows, key):
            f = _Frame(int(rows.shape[0]))
            for k, col in frame.items():
                dict.__setitem__(f, k, col[rows.long()])
            dict.__setitem__(f, key, rows)
            return f

        cet = self._canonical_etypes[0]
        cg = DGLGraph(gidx, [cet[0], cet[2]], [cet],
                      [sub(self._node_frames[s_t], uniq_s, NID), sub(self._node_frames[d_t], uniq_d, NID)],
                      [sub(self._edge_frames[0], eid, EID)], [0], [1])
        return cg, uniq_d

    def _partial_message_passing(self, g, u, v, eid, recv, mfunc, rfunc, afunc):
        cg, dstnodes = g._compute_graph(u, v, eid, recv)
        ndata = _udf.message_passing(cg, mfunc, rfunc, afunc, _message_passing)
        ndata.pop(NID, None)
        g._set_n_repr(g.get_ntype_id_from_dst(None), dstnodes, ndata)

    def pull(self, v, message_func, reduce_func, apply_node_func=None, etype=None):
        """Pull messages from the predecessors of ``v`` and update ``v`` (heterograph.py:4842-4938)."""
        v = self._ids(v)
        if v.numel() == 0:
            return
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        src, dst, eid = g.in_edges(v, form="all")
        self._partial_message_passing(g, src, dst, eid, v, message_func, reduce_func, apply_node_func)

    def push(self, u, message_func, reduce_func, apply_node_func=None, etype=None):
        """Send messages from ``u`` along their out-edges and update the successors (heterograph.py:4940-5016)."""
        u = self._ids(u)
        if u.numel() == 0:
            return
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        src, dst, eid = g.out_edges(u, form="all")
        if eid.numel() == 0:
            return
        self._partial_message_passing(g, src, dst, eid, None, message_func, reduce_func, apply_node_func)

    def send_and_recv(self, edges, message_func, reduce_func, apply_node_func=None, etype=None):
        """Message passing along the given edges only (heterograph.py:4714-4840)."""
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        eid = g._parse_edges(edges)
        if eid.numel() == 0:
            return
        u, v = g.find_edges(eid)
        self._partial_message_passing(g, u, v, eid, None, message_func, reduce_func, apply_node_func)

    # ---- structure queries used by the calls above ------------------------------------------------
    def _gather_ranges(self, fmt, ids):
        """All (position, owner) pairs of rows ``ids`` of a compressed format."""
        indptr, indices, emap = fmt
        ids_l = ids.long()
        ip = indptr.long()
        degs = ip[ids_l + 1] - ip[ids_l]
        total = int(degs.sum())
        owner = torch.repeat_interleave(torch.arange(ids_l.shape[0], device=ids.device), degs, output_size=total)
        start_excl = torch.cumsum(degs, 0) - degs
        pos = ip[ids_l][owner] + (torch.arange(total, device=ids.device) - start_excl[owner])
        other = indices[pos]
        eid = pos.to(self.idtype) if emap is None else emap[pos]
        return other, ids[owner], eid

    def in_edges(self, v, form="uv", etype=None):
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        src, dst, eid = g._gather_ranges(g._graph.relations[0].csc(), g._ids(v))
        return {"uv": (src, dst), "eid": eid, "all": (src, dst, eid)}[form]

    def out_edges(self, u, form="uv", etype=None):
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        dst, src, eid = g._gather_ranges(g._graph.relations[0].csr(), g._ids(u))
        return {"uv": (src, dst), "eid": eid, "all": (src, dst, eid)}[form]

    def find_edges(self, eid, etype=None):
        u, v = self.edges(etype=etype)
        sel = self._ids(eid).long()
        return u[sel], v[sel]

    def edge_ids(self, u, v, etype=None):
        """The id of ONE edge between each (u, v) pair (the smallest, if there are several)."""
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        u, v = g._ids(u), g._ids(v)
        src, dst = g.edges()
        n_dst = g._graph.relations[0].num_dst
        key = src.long() * max(n_dst, 1) + dst.long()
        order = torch.argsort(key, stable=True)
        want = u.long() * max(n_dst, 1) + v.long()
        at = torch.searchsorted(key[order], want)
        ok = (at < key.numel()) & (key[order][at.clamp(max=max(key.numel() - 1, 0))] == want) if key.numel() else \
            torch.zeros_like(want, dtype=torch.bool)
        if not bool(ok.all()):
            raise DGLAMDError("edge_ids: some (u, v) pairs are not edges of the graph")
        return order[at].to(g.idtype)

    def has_edges_between(self, u, v, etype=None):
        """Whether an edge u -> v exists, pair by pair (heterograph.py has_edges_between)."""
        g = self if len(self._canonical_etypes) == 1 else self[etype]
        u, v = g._ids(u), g._ids(v)
        src, dst = g.edges()
        n_dst = max(g._graph.relations[0].num_dst, 1)
        key = torch.sort(src.long() * n_dst + dst.long())[0]
        want = u.long() * n_dst + v.long()
        if key.numel() == 0:
            return torch.zeros_like(want, dtype=torch.bool)
        at = torch.searchsorted(key, want).clamp(max=key.numel() - 1)
        return key[at] == want

    # ---- mutation (used by the reference's suites to build their inputs) -------------------------------
    def _grow_frame(self, frame, extra, data=None):
        nf = _Frame(frame.num_rows + extra)
        for k, col in frame.items():
            add = data[k] if data is not None and k in data else \
                torch.zeros((extra,) + tuple(col.shape[1:]), dtype=col.dtype, device=col.device)
            dict.__setitem__(nf, k, torch.cat([col, add.to(col.device)], 0))
        if data is not None:
            for k, add in data.items():
                if k not in frame:
                    pad = torch.zeros((frame.num_rows,) + tuple(add.shape[1:]), dtype=add.dtype, device=add.device)
                    dict.__setitem__(nf, k, torch.cat([pad, add], 0))
        return nf

    def _rebuild(self, num_nodes, pairs):
        rels = []
        for et, (s, d) in enumerate(self._graph.metagraph.edges):
            u, v = pairs[et]
            rels.append(Relation(num_nodes[s], num_nodes[d], u.contiguous(), v.contiguous(), idtype=self.idtype,
                                 device=self.device))
        self._graph = GraphIndex(num_nodes, self._graph.metagraph.edges, rels)

    def add_nodes(self, num, data=None, ntype=None):
        """Append ``num`` nodes of one type, in place (heterograph.py add_nodes); new feature rows are zero."""
        ntid = self.get_ntype_id(ntype)
        counts = [self._graph.num_nodes(i) for i in range(len(self._ntypes))]
        counts[ntid] += int(num)
        pairs = [self.edges(etype=c) for c in self._canonical_etypes]
        self._rebuild(counts, pairs)
        self._node_frames[ntid] = self._grow_frame(self._node_frames[ntid], int(num), data)

    def add_edges(self, u, v, data=None, etype=None):
        """Append edges ``u -> v`` of one type, in place (heterograph.py add_edges); nodes are added as needed."""
        etid = self.get_etype_id(etype)
        u, v = self._ids(u), self._ids(v)
        if u.numel() == 1 and v.numel() > 1:
            u = u.expand(v.numel())
        if v.numel() == 1 and u.numel() > 1:
            v = v.expand(u.numel())
        s, d = self._graph.metagraph.find_edge(etid)
        counts = [self._graph.num_nodes(i) for i in range(len(self._ntypes))]
        old = list(counts)
        if u.numel():
            counts[s] = max(counts[s], int(u.max()) + 1)
            counts[d] = max(counts[d], int(v.max()) + 1)
        pairs = [list(self.edges(etype=c)) for c in self._canonical_etypes]
        pairs[etid] = [torch.cat([pairs[etid][0], u]), torch.cat([pairs[etid][1], v])]
        self._rebuild(counts, pairs)
        for i, (a, b) in enumerate(zip(old, counts)):
            if b > a:
                self._node_frames[i] = self._grow_frame(self._node_frames[i], b - a)
        self._edge_frames[etid] = self._grow_frame(self._edge_frames[etid], int(u.numel()), data)

    def __repr__(self):
        return "DGLGraph(num_nodes={}, num_edges={}, ntypes={}, etypes={})".format(
            {n: self._graph.num_nodes(i) for i, n in enumerate(self._ntypes)},
            {c: self._graph.num_edges(i) for i, c in enumerate(self._canonical_etypes)},
            self._ntypes, self.etypes)


class _EdgeView:
    def __init__(self, g):
        self._g = g

    def __call__(self, form="uv", order="eid", etype=None):
        g = self._g
        if form not in ("uv", "eid", "all"):        # g.edges(etype) — positional edge type (older call sites)
            form, etype = "uv", form
        row, col, old = g._graph.relations[g.get_etype_id(etype)].coo()
        if old is not None:
            perm = torch.argsort(old)
            row, col = row[perm], col[perm]
        if form == "uv":
            return row, col
        eid = torch.arange(row.shape[0], dtype=g.idtype, device=row.device)
        return eid if form == "eid" else (row, col, eid)

    def __getitem__(self, key):
        return self._g._edge_indexer()[key]


def _replace_inf_with_zero(x):
    return torch.where(torch.isinf(x), torch.zeros_like(x), x)


def _field(g, code, name):
    view = [g.srcdata, g.dstdata, g.edata][code]
    return view[name]


def _as_tuple(g, data, target):
    """Feature of a multi-relation graph as a tuple in type-id order (core.data_dict_to_list)."""
    if isinstance(data, torch.Tensor) and target != "e":
        # a side with ONE node type hands out a tensor, not a dict (g.srcdata on a graph whose relations all leave
        # the same type): it belongs to that type's slot
        side = (g._src_ntype_ids if target == "u" else g._dst_ntype_ids) if g.is_unibipartite else range(len(g.ntypes))
        side = list(side)
        if len(side) == 1:
            out = [None] * len(g.ntypes)
            out[side[0]] = data
            return tuple(out)
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


def graph(data, num_nodes=None, idtype=None, device=None, row_sorted=False, col_sorted=False):  # noqa: ARG001
    """Homogeneous graph from ``(src, dst)`` or a list of ``(u, v)`` pairs (dgl.graph).  ``row_sorted`` / ``col_sorted``
    are the reference's hints about the order of the ids (python/dgl/convert.py:69-75); the formats are built by a stable
    sort either way, so they change nothing here."""
    if isinstance(data, list):
        # the reference takes a LIST as (u, v) pairs and a TUPLE as (src ids, dst ids); a two-element list whose
        # elements are themselves id sequences / tensors is read as (src ids, dst ids) too
        def is_scalar_pair(p):
            return isinstance(p, (tuple, list)) and len(p) == 2 and not isinstance(p[0], (list, tuple, torch.Tensor)) \
                and not isinstance(p[1], (list, tuple, torch.Tensor))
        if len(data) == 0 or all(is_scalar_pair(p) for p in data):
            data = ([p[0] for p in data], [p[1] for p in data])
        elif len(data) != 2:
            raise DGLAMDError("dgl.graph: expected (src ids, dst ids) or a list of (u, v) pairs")
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


# ---- the few transforms the reference's suites use to build inputs (python/dgl/convert.py) ------------------
def from_networkx(nx_graph, idtype=None, device=None):
    """Graph from a networkx(-like) graph: ``number_of_nodes()``, ``edges()``, ``is_directed()``; an undirected
    graph gives both directions of every edge (dgl.from_networkx on ``nx.Graph``)."""
    pairs = [(int(a), int(b)) for a, b in nx_graph.edges()]
    if not nx_graph.is_directed():
        pairs = pairs + [(b, a) for a, b in pairs if a != b]
    u = torch.tensor([p[0] for p in pairs], dtype=torch.int64)
    v = torch.tensor([p[1] for p in pairs], dtype=torch.int64)
    return graph((u, v), num_nodes=int(nx_graph.number_of_nodes()), idtype=idtype or torch.int64,
                 device=device or torch.device("cpu"))


def to_homogeneous(G, ndata=None, edata=None, store_type=True, return_count=False):
    """One node / edge type: nodes concatenated in node-type order, edges in edge-type order; ``ndata[NTYPE]`` /
    ``ndata[NID]`` / ``edata[ETYPE]`` / ``edata[EID]`` remember the origin (dgl.to_homogeneous); ``ndata`` /
    ``edata`` name the feature fields to carry over (concatenated)."""
    counts = [G._graph.num_nodes(i) for i in range(len(G._ntypes))]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    dev, it = G.device, G.idtype
    us, vs, ety, eid = [], [], [], []
    for et, cet in enumerate(G._canonical_etypes):
        s, d = G._graph.metagraph.find_edge(et)
        u, v = G.edges(etype=cet)
        us.append(u + offs[s])
        vs.append(v + offs[d])
        ety.append(torch.full((u.shape[0],), et, dtype=it, device=dev))
        eid.append(torch.arange(u.shape[0], dtype=it, device=dev))
    cat = lambda xs: torch.cat(xs) if xs else torch.zeros(0, dtype=it, device=dev)
    hg = graph((cat(us), cat(vs)), num_nodes=offs[-1], idtype=it, device=dev)
    if store_type:
        hg.ndata[NTYPE] = torch.cat([torch.full((c,), i, dtype=it, device=dev) for i, c in enumerate(counts)]) \
            if counts else torch.zeros(0, dtype=it, device=dev)
    nid = torch.cat([torch.arange(c, dtype=it, device=dev) for c in counts]) if counts else torch.zeros(0, dtype=it, device=dev)
    hg._node_frames[0][NID] = nid
    if store_type:
        hg._edge_frames[0][ETYPE] = cat(ety)
    hg._edge_frames[0][EID] = cat(eid)
    for k in (ndata or []):
        hg._node_frames[0][k] = torch.cat([f[k] for f in G._node_frames])
    for k in (edata or []):
        hg._edge_frames[0][k] = torch.cat([f[k] for f in G._edge_frames])
    if return_count:    # python/dgl/convert.py to_homogeneous: (graph, nodes per type, edges per type)
        return hg, counts, [int(t.shape[0]) for t in eid]
    return hg


def to_heterogeneous(G, ntypes, etypes, ntype_field=NTYPE, etype_field=ETYPE, metagraph=None):
    """Split a homogeneous graph by ``ndata[ntype_field]`` / ``edata[etype_field]`` (dgl.to_heterogeneous): node ids
    inside a type follow their order in ``G``; every other feature field is sliced along (autograd kept)."""
    if metagraph is not None:
        raise DGLAMDError("to_heterogeneous: a caller-given metagraph is not supported; the metagraph is derived from the "
                       "type fields")
    ntype_ids = G.ndata[ntype_field].long()
    etype_ids = G.edata[etype_field].long()
    dev, it = G.device, G.idtype
    # new node id of every node inside its type
    local = torch.zeros_like(ntype_ids)
    node_sel = []
    for i in range(len(ntypes)):
        m = (ntype_ids == i).nonzero(as_tuple=True)[0]
        local[m] = torch.arange(m.shape[0], device=dev)
        node_sel.append(m)
    u, v = G.edges()
    u, v = u.long(), v.long()
    data, edge_sel, cets = {}, {}, []
    for ei, et in enumerate(etypes):
        m = (etype_ids == ei).nonzero(as_tuple=True)[0]
        if m.numel() == 0:
            continue
        st = ntype_ids[u[m]]
        dt = ntype_ids[v[m]]
        combos = torch.unique(torch.stack([st, dt], 1), dim=0).tolist()
        for s_i, d_i in combos:
            mm = m[(st == s_i) & (dt == d_i)]
            cet = (ntypes[s_i], et, ntypes[d_i])
            data[cet] = (local[u[mm]].to(it), local[v[mm]].to(it))
            edge_sel[cet] = mm
            cets.append(cet)
    hg = heterograph(data, {nt: int(node_sel[i].shape[0]) for i, nt in enumerate(ntypes)}, idtype=it, device=dev)
    for i, nt in enumerate(ntypes):
        f = hg._node_frames[hg.get_ntype_id(nt)]
        for k, col in G._node_frames[0].items():
            if k not in (ntype_field, NID):
                f[k] = col[node_sel[i]]
    for cet, mm in edge_sel.items():
        f = hg._edge_frames[hg.get_etype_id(cet)]
        for k, col in G._edge_frames[0].items():
            if k not in (etype_field, EID):
                f[k] = col[mm]
    # the ids the per-type nodes / edges had in G (python/dgl/convert.py:878-887: hg.ndata[dgl.NID] / edata[dgl.EID])
    for i, nt in enumerate(ntypes):
        hg._node_frames[hg.get_ntype_id(nt)][NID] = node_sel[i].to(it)
    for cet, mm in edge_sel.items():
        hg._edge_frames[hg.get_etype_id(cet)][EID] = mm.to(it)
    return hg
