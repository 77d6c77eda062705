"""User-defined message / reduce / apply functions — the reference's NON-accelerated path, in plain torch.

``update_all`` / ``apply_edges`` / ``pull`` / ``apply_nodes`` accept any Python callable next to the built-in
functions (python/dgl/core.py:372-425).  A built-in pair is routed to the fused g-SpMM / g-SDDMM kernels; a
callable runs here, the way the reference runs it:

  * an edge function gets an :class:`EdgeBatch` — ``edges.src[k]`` / ``edges.dst[k]`` are the end points' feature
    rows gathered per edge, ``edges.data[k]`` the edge features (core.py:53-96, udf.py:5-238);
  * a reduce function is applied by DEGREE BUCKETING (core.py:99-174): destination nodes are grouped by in-degree,
    every group gets a :class:`NodeBatch` whose ``mailbox[k]`` is ``(nodes in the group, degree, ...)`` with a
    node's messages ordered by edge id; nodes without in-edges are skipped and keep the zero initializer;
  * an apply function gets a :class:`NodeBatch` over all (receiving) nodes (core.py:17-50).

Nothing here is a kernel and nothing here is timed: it is ``index_select`` / ``reshape`` / ``index_copy`` on the
caller's device, differentiable through autograd.  It exists so that (a) mixed user code drops in unchanged and
(b) the parity tests have the reference's OWN second oracle — its suites compare every built-in against the
same computation written as UDFs (tests/python/common/ops/test_ops.py:87-181) — with no use of ``oracle/``.
"""
import torch

from ._lib import DGLAMDError

NID = EID = "_ID"   # python/dgl/base.py:14-15


class _Rows:
    """Read-only ``{field: frame[field][index]}`` evaluated lazily (only the fields a UDF touches are gathered)."""

    def __init__(self, frame, index):
        self._frame, self._index, self._cache = frame, index, {}

    def __getitem__(self, key):
        if key not in self._cache:
            col = self._frame[key]
            self._cache[key] = col if self._index is None else col[self._index]
        return self._cache[key]

    def __contains__(self, key):
        return key in self._frame

    def __iter__(self):
        return iter(self._frame)

    def __len__(self):
        return len(self._frame)

    def keys(self):
        return self._frame.keys()

    def items(self):
        return ((k, self[k]) for k in self._frame)

    def values(self):
        return (self[k] for k in self._frame)

    def get(self, key, default=None):
        return self[key] if key in self._frame else default


class EdgeBatch:
    """A batch of edges handed to an edge UDF (python/dgl/udf.py:5-238)."""

    def __init__(self, graph, eid, etype, src_data, edge_data, dst_data, endpoints):
        self._graph, self._eid, self._etype = graph, eid, etype
        self._src, self._edge, self._dst = src_data, edge_data, dst_data
        self._u, self._v = endpoints

    @property
    def src(self):
        return self._src

    @property
    def dst(self):
        return self._dst

    @property
    def data(self):
        return self._edge

    def edges(self):
        return self._u, self._v, self._eid

    def batch_size(self):
        return int(self._eid.shape[0])

    def __len__(self):
        return self.batch_size()

    @property
    def canonical_etype(self):
        return self._etype


class NodeBatch:
    """A batch of nodes handed to a reduce / apply UDF (python/dgl/udf.py:241-417)."""

    def __init__(self, graph, nodes, ntype, data, msgs=None):
        self._graph, self._nodes, self._ntype, self._data, self._msgs = graph, nodes, ntype, data, msgs

    @property
    def data(self):
        return self._data

    @property
    def mailbox(self):
        if self._msgs is None:
            raise DGLAMDError("NodeBatch.mailbox is only available inside a reduce function")
        return self._msgs

    def nodes(self):
        return self._nodes

    def batch_size(self):
        return int(self._nodes.shape[0])

    def __len__(self):
        return self.batch_size()

    @property
    def ntype(self):
        return self._ntype


def is_builtin(func):
    from . import function as fn

    return isinstance(func, fn.BuiltinFunction)


def _check_result(res, what, rows):
    if not isinstance(res, dict):
        raise DGLAMDError("User-defined %s function must return a dict of str -> tensor." % what)
    for k, v in res.items():
        if not isinstance(v, torch.Tensor) or v.dim() == 0 or v.shape[0] != rows:
            raise DGLAMDError("Expect number of features to match number of %s. Got %s and %d instead (field %r)."
                              % ("edges" if what == "message" else "nodes",
                                 "a non-tensor" if not isinstance(v, torch.Tensor) else str(tuple(v.shape)), rows, k))
    return res


def invoke_edge_udf(g, func, eid=None, orig_eid=None):
    """``func(EdgeBatch)`` over all edges (``eid`` None) or the given edge ids of the single relation of ``g``
    (core.py:53-96).  Returns ``{field: (number of edges, ...)}`` in the order of ``eid``.  ``orig_eid``: the ids the
    batch reports (``edges.edges()[2]``) when ``g`` is an extracted compute graph (core.py:65, 90)."""
    s_t, d_t = g.get_ntype_id_from_src(None), g.get_ntype_id_from_dst(None)
    u, v = g.edges()
    if eid is None:
        ids = torch.arange(g.num_edges(), dtype=g.idtype, device=g.device)
        sel = None
    else:
        ids = eid
        sel = eid.long()
        u, v = u[sel], v[sel]
    ebatch = EdgeBatch(g, ids if orig_eid is None else orig_eid, g.canonical_etypes[0],
                       _Rows(g._node_frames[s_t], u.long()), _Rows(g._edge_frames[0], sel),
                       _Rows(g._node_frames[d_t], v.long()), (u, v))
    return _check_result(func(ebatch), "message", int(ids.shape[0]))


def invoke_node_udf(g, func, ntid, nodes=None, ndata=None, orig_nid=None):
    """``func(NodeBatch)`` on all nodes of type ``ntid`` (``nodes`` None) or on the given ones (core.py:17-50)."""
    if nodes is None:
        ids = torch.arange(g._graph.num_nodes(ntid), dtype=g.idtype, device=g.device)
        data = g._node_frames[ntid] if ndata is None else ndata
    else:
        ids = nodes
        data = _Rows(g._node_frames[ntid], nodes.long()) if ndata is None else ndata
    nb = NodeBatch(g, ids if orig_nid is None else orig_nid, g._ntypes[ntid], data)
    return _check_result(func(nb), "apply", int(ids.shape[0]))


def _in_edge_table(rel):
    """(indptr, edge id per CSC position) of the in-edge CSR, as int64."""
    indptr, _, emap = rel.csc()
    n = int(indptr.shape[0]) - 1
    if emap is None:
        emap = torch.arange(rel.num_edges, device=indptr.device)
    return indptr.long(), emap.long(), n


def invoke_udf_reduce(g, func, msgdata, orig_nid=None):
    """Degree bucketing (core.py:99-174): returns ``{field: (num_dst, ...)}``, zero rows for nodes without
    in-edges (the frame's default initializer)."""
    rel = g._graph.relations[0]
    d_t = g.get_ntype_id_from_dst(None)
    dst_frame = g._node_frames[d_t]
    indptr, emap, n = _in_edge_table(rel)
    degs = indptr[1:] - indptr[:-1]
    nodes = torch.arange(n, device=degs.device)
    if orig_nid is None:
        orig_nid = nodes.to(g.idtype)
    results, bucket_nodes = [], []
    for deg in torch.unique(degs).tolist():          # ascending, like F.unique(sorted_val)
        if deg == 0:
            continue                                 # zero-degree nodes: the reduce function is not invoked
        nb = nodes[degs == deg]                       # node ids ascending inside a bucket (stable sort in the reference)
        pos = indptr[nb].unsqueeze(1) + torch.arange(deg, device=nb.device).unsqueeze(0)
        eid = torch.sort(emap[pos], dim=1)[0]         # a node's incoming edges ordered by edge id
        flat = eid.reshape(-1)
        mail = {k: m[flat].reshape((nb.shape[0], deg) + tuple(m.shape[1:])) for k, m in msgdata.items()}
        batch = NodeBatch(g, orig_nid[nb], g._ntypes[d_t], _Rows(dst_frame, nb), msgs=mail)
        results.append(_check_result(func(batch), "reduce", int(nb.shape[0])))
        bucket_nodes.append(nb)
    out = {}
    if results:
        merged_nodes = torch.cat(bucket_nodes)
        for k in results[0]:
            val = torch.cat([r[k] for r in results], dim=0)
            base = torch.zeros((n,) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device)
            out[k] = base.index_copy(0, merged_nodes, val)      # out of place: differentiable (F.scatter_row)
    return out


def message_passing(g, mfunc, rfunc, afunc, fused):
    """core.message_passing (core.py:372-425) on a single-relation graph ``g``.  ``fused(g, mfunc, rfunc)`` is the
    built-in route (g-SpMM, or g-SDDMM + copy_e g-SpMM); it is taken whenever both functions are built-ins."""
    from . import function as fn
    from .heterograph import _invoke_gsddmm, _invoke_gspmm   # (the package re-exports a FUNCTION called heterograph)

    if is_builtin(mfunc) and is_builtin(rfunc):
        ndata = fused(g, mfunc, rfunc)
    else:
        # message phase
        if is_builtin(mfunc):
            msgdata = _invoke_gsddmm(g, mfunc)
        else:
            msgdata = invoke_edge_udf(g, mfunc, orig_eid=g._edge_frames[0].get(EID))          # core.py:405-407
        # reduce phase
        if is_builtin(rfunc):
            m = rfunc.msg_field
            if m not in msgdata:
                raise DGLAMDError("Invalid message ({}) and reduce ({}) function pairs. The message function must "
                                  "produce the field the reduce function reads.".format(mfunc, rfunc))
            ndata = _invoke_gspmm(g, fn.copy_e(m, m), rfunc, edata=msgdata)
        else:
            ndata = invoke_udf_reduce(g, rfunc, msgdata,
                                      orig_nid=g._node_frames[g.get_ntype_id_from_dst(None)].get(NID))   # core.py:414-415
    if afunc is not None:
        d_t = g.get_ntype_id_from_dst(None)
        full = dict(g._node_frames[d_t])             # include original node features (core.py:416-419)
        full.update(ndata)
        ndata = invoke_node_udf(g, afunc, d_t, ndata=full, orig_nid=g._node_frames[d_t].get(NID))   # core.py:421-423
    return ndata
