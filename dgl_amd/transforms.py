"""Graph construction helpers around the hot path, in torch index arithmetic on the caller's device.

``add_self_loop`` / ``remove_self_loop`` / ``remove_edges`` / ``reorder_graph`` / ``add_reverse_edges`` / ``to_simple`` /
``to_bidirected`` (python/dgl/transforms/functional.py), ``node_subgraph`` / ``edge_subgraph`` / ``in_subgraph``
(python/dgl/subgraph.py),
``batch`` (python/dgl/batch.py), ``from_scipy`` / ``bipartite_from_scipy`` (python/dgl/convert.py), ``adj_external``
(heterograph.py): the handful the reference's own layer tests
(tests/python/pytorch/nn/test_nn.py, tests/utils/graph_cases.py) build their graphs with — tools/ref_suite runs those
files unmodified.  None of this is a kernel and none of it is timed; every graph these return builds its CSR / CSC with
the library's COO -> CSR kernel the first time an operator asks for it, like any other graph.
"""
import torch

from . import function as fn
from ._lib import DGLAMDError
from .graph_index import GraphIndex, Relation
from .heterograph import DGLGraph, _Frame, graph, heterograph

EID = NID = "_ID"


def _single(g, what):
    if len(g.canonical_etypes) != 1:
        raise DGLAMDError("%s: an edge type must be given on a graph with several edge types" % what)


def _with_edges(g, etid, u, v, edge_frame):
    """Copy of ``g`` whose relation ``etid`` has the edges (u, v) and the given edge frame; everything else shared."""
    rels = list(g._graph.relations)
    old = rels[etid]
    rels[etid] = Relation(old.num_src, old.num_dst, u.contiguous(), v.contiguous(), idtype=g.idtype, device=g.device)
    gidx = GraphIndex([g._graph.num_nodes(i) for i in range(len(g._ntypes))], list(g._graph.metagraph.edges), rels)
    eframes = [g._copy_frame(f) for f in g._edge_frames]
    eframes[etid] = edge_frame
    return DGLGraph(gidx, g._ntypes, g._canonical_etypes, [g._copy_frame(f) for f in g._node_frames], eframes,
                    g._src_ntype_ids, g._dst_ntype_ids)


def add_self_loop(g, edge_feat_names=None, fill_data=1.0, etype=None):
    """One edge i -> i per node, appended behind the existing edges; their features are ``fill_data`` — a number, or
    'sum' / 'mean' / 'max' / 'min' of the node's incoming edges' features (transforms/functional.py add_self_loop)."""
    etid = g.get_etype_id(etype)
    s, d = g._graph.metagraph.find_edge(etid)
    if s != d:
        raise DGLAMDError("add_self_loop does not support unidirectional bipartite graphs: {}. Please make sure the types "
                          "of head node and tail node are identical.".format(g.canonical_etypes[etid]))
    n = g._graph.num_nodes(s)
    u, v = g.edges(etype=g.canonical_etypes[etid])
    loop = torch.arange(n, dtype=g.idtype, device=g.device)
    old = g._edge_frames[etid]
    frame = _Frame(int(u.shape[0]) + n)
    for k, col in old.items():
        shape = (n,) + tuple(col.shape[1:])
        if edge_feat_names is not None and k not in edge_feat_names:
            # not named: the feature stays, the new self-loop rows are zero (the reference goes through add_edges(data=...))
            dict.__setitem__(frame, k, torch.cat([col, torch.zeros(shape, dtype=col.dtype, device=col.device)], 0))
            continue
        if isinstance(fill_data, str):
            if fill_data not in ("sum", "mean", "max", "min"):
                raise DGLAMDError("Unsupported aggregation: {}".format(fill_data))
            idx = v.long().view((-1,) + (1,) * (col.dim() - 1)).expand_as(col)
            op = {"sum": "sum", "mean": "mean", "max": "amax", "min": "amin"}[fill_data]
            add = torch.zeros(shape, dtype=col.dtype, device=col.device).scatter_reduce(0, idx, col, op, include_self=False)
        else:
            add = torch.full(shape, fill_data, dtype=col.dtype, device=col.device)
        dict.__setitem__(frame, k, torch.cat([col, add], 0))
    return _with_edges(g, etid, torch.cat([u, loop]), torch.cat([v, loop]), frame)


def remove_edges(g, eids, etype=None, store_ids=False):
    """Copy of ``g`` without the given edges of one type; the others keep their relative order (functional.py)."""
    etid = g.get_etype_id(etype)
    u, v = g.edges(etype=g.canonical_etypes[etid])
    keep = torch.ones(u.shape[0], dtype=torch.bool, device=g.device)
    keep[torch.as_tensor(eids, device=g.device).long()] = False
    sel = torch.nonzero(keep).reshape(-1)
    frame = _Frame(int(sel.shape[0]))
    for k, col in g._edge_frames[etid].items():
        dict.__setitem__(frame, k, col[sel])
    if store_ids:
        dict.__setitem__(frame, EID, sel.to(g.idtype))
    out = _with_edges(g, etid, u[sel], v[sel], frame)
    if store_ids:
        for i in range(len(out._ntypes)):
            out._node_frames[i][NID] = torch.arange(out._graph.num_nodes(i), dtype=g.idtype, device=g.device)
    return out


def remove_self_loop(g, etype=None):
    etid = g.get_etype_id(etype)
    u, v = g.edges(etype=g.canonical_etypes[etid])
    return remove_edges(g, torch.nonzero(u == v).reshape(-1), etype=etype)


def reorder_graph(g, node_permute_algo=None, edge_permute_algo="src", store_ids=True, permute_config=None):
    """Edges re-ordered by source / destination id or by a given permutation (functional.py reorder_graph; node
    re-ordering — rcmk / metis — is the reference's graph-partitioning tool chain and is not offered)."""
    if node_permute_algo is not None:
        raise DGLAMDError("reorder_graph: node_permute_algo is not supported (only edges are re-ordered)")
    _single(g, "reorder_graph")
    u, v = g.edges()
    if edge_permute_algo == "src":
        perm = torch.argsort(u, stable=True)
    elif edge_permute_algo == "dst":
        perm = torch.argsort(v, stable=True)
    elif edge_permute_algo == "custom":
        if not permute_config or "edges_perm" not in permute_config:
            raise DGLAMDError("edge_permute_algo='custom' needs permute_config['edges_perm']")
        perm = torch.as_tensor(permute_config["edges_perm"], device=g.device)
        if perm.shape[0] != u.shape[0]:
            raise DGLAMDError("edges_perm must hold one entry per edge")
    else:
        raise DGLAMDError("Unexpected edge_permute_algo is specified: {}. Expected algos: ['src', 'dst', 'custom']"
                          .format(edge_permute_algo))
    perm = perm.long()
    frame = _Frame(int(u.shape[0]))
    for k, col in g._edge_frames[0].items():
        dict.__setitem__(frame, k, col[perm])
    if store_ids:
        dict.__setitem__(frame, EID, perm.to(g.idtype))
    return _with_edges(g, 0, u[perm], v[perm], frame)


def batch(graphs, ndata="__ALL__", edata="__ALL__"):
    """Disjoint union with node / edge ids shifted graph by graph (python/dgl/batch.py); ``batch_num_nodes()`` /
    ``batch_num_edges()`` give the pieces."""
    if len(graphs) == 0:
        raise DGLAMDError("The input list of graphs cannot be empty.")
    g0 = graphs[0]
    for g in graphs[1:]:
        if g.ntypes != g0.ntypes or g.canonical_etypes != g0.canonical_etypes:
            raise DGLAMDError("All graphs should have the same node types and edge types to be batched.")
    nts, cets = g0.ntypes, g0.canonical_etypes
    counts = {n: [g.num_nodes(n) for g in graphs] for n in nts}
    offs = {n: [sum(counts[n][:i]) for i in range(len(graphs))] for n in nts}
    data = {}
    for c in cets:
        us, vs = [], []
        for i, g in enumerate(graphs):
            u, v = g.edges(etype=c)
            us.append(u + offs[c[0]][i])
            vs.append(v + offs[c[2]][i])
        data[c] = (torch.cat(us), torch.cat(vs))
    total = {n: sum(counts[n]) for n in nts}
    if len(cets) == 1 and len(nts) == 1:
        out = graph(data[cets[0]], num_nodes=total[nts[0]], idtype=g0.idtype, device=g0.device)
    else:
        out = heterograph(data, total, idtype=g0.idtype, device=g0.device)
    def merge(frames, keys):
        merged = {}
        names = list(frames[0].keys()) if keys == "__ALL__" else (keys or [])
        for k in names:
            if all(k in f for f in frames):
                merged[k] = torch.cat([f[k] for f in frames], 0)
        return merged
    for i, n in enumerate(out._ntypes):
        for k, val in merge([g._node_frames[g.get_ntype_id(n)] for g in graphs], ndata).items():
            out._node_frames[i][k] = val
    for i, c in enumerate(out._canonical_etypes):
        for k, val in merge([g._edge_frames[g.get_etype_id(c)] for g in graphs], edata).items():
            out._edge_frames[i][k] = val
    out._batch_num_nodes = {n: torch.tensor(counts[n], dtype=g0.idtype, device=g0.device) for n in nts}
    out._batch_num_edges = {c: torch.tensor([g.num_edges(c) for g in graphs], dtype=g0.idtype, device=g0.device) for c in cets}
    out.batch_size = len(graphs)
    return out


def unbatch(g, node_split=None, edge_split=None):
    """The graphs a batch was made of, with their slices of the features (python/dgl/batch.py unbatch)."""
    nts, cets = g.ntypes, g.canonical_etypes
    nsplit = {n: (g.batch_num_nodes(n) if node_split is None else torch.as_tensor(
        node_split[n] if isinstance(node_split, dict) else node_split)).tolist() for n in nts}
    esplit = {c: (g.batch_num_edges(c) if edge_split is None else torch.as_tensor(
        edge_split[c] if isinstance(edge_split, dict) else edge_split)).tolist() for c in cets}
    k = len(nsplit[nts[0]])
    noff = {n: [sum(nsplit[n][:i]) for i in range(k + 1)] for n in nts}
    eoff = {c: [sum(esplit[c][:i]) for i in range(k + 1)] for c in cets}
    out = []
    for i in range(k):
        data = {}
        for c in cets:
            u, v = g.edges(etype=c)
            a, b = eoff[c][i], eoff[c][i + 1]
            data[c] = (u[a:b] - noff[c[0]][i], v[a:b] - noff[c[2]][i])
        counts = {n: nsplit[n][i] for n in nts}
        if len(cets) == 1 and len(nts) == 1:
            sg = graph(data[cets[0]], num_nodes=counts[nts[0]], idtype=g.idtype, device=g.device)
        else:
            sg = heterograph(data, counts, idtype=g.idtype, device=g.device)
        for n in nts:
            fr = g._node_frames[g.get_ntype_id(n)]
            for key, val in fr.items():
                sg._node_frames[sg.get_ntype_id(n)][key] = val[noff[n][i]:noff[n][i + 1]]
        for c in cets:
            fr = g._edge_frames[g.get_etype_id(c)]
            for key, val in fr.items():
                sg._edge_frames[sg.get_etype_id(c)][key] = val[eoff[c][i]:eoff[c][i + 1]]
        out.append(sg)
    return out


def from_scipy(sp_mat, eweight_name=None, idtype=None, device=None):
    """Graph with an edge row -> column per nonzero of a square scipy sparse matrix (convert.py from_scipy)."""
    if sp_mat.shape[0] != sp_mat.shape[1]:
        raise DGLAMDError("Expect the number of rows to be the same as the number of columns for sp_mat, got {:d} and {:d}."
                          .format(sp_mat.shape[0], sp_mat.shape[1]))
    coo = sp_mat.tocoo()
    g = graph((torch.as_tensor(coo.row).long(), torch.as_tensor(coo.col).long()), num_nodes=sp_mat.shape[0],
              idtype=idtype or torch.int64, device=device or torch.device("cpu"))
    if eweight_name is not None:
        g.edata[eweight_name] = torch.as_tensor(coo.data).to(g.device)
    return g


def bipartite_from_scipy(sp_mat, utype, etype, vtype, eweight_name=None, idtype=None, device=None):
    coo = sp_mat.tocoo()
    g = heterograph({(utype, etype, vtype): (torch.as_tensor(coo.row).long(), torch.as_tensor(coo.col).long())},
                    {utype: sp_mat.shape[0], vtype: sp_mat.shape[1]}, idtype=idtype or torch.int64,
                    device=device or torch.device("cpu"))
    if eweight_name is not None:
        g.edata[eweight_name] = torch.as_tensor(coo.data).to(g.device)
    return g


def adj_external(g, transpose=False, ctx=None, scipy_fmt=None, etype=None):
    """Adjacency matrix as a torch sparse COO tensor — rows = source nodes unless ``transpose`` — or a scipy matrix
    (heterograph.py adj_external)."""
    c = g.to_canonical_etype(etype)
    u, v = g.edges(etype=c)
    n_src, n_dst = g.num_src_nodes(c[0]) if g.is_unibipartite else g.num_nodes(c[0]), \
        g.num_dst_nodes(c[2]) if g.is_unibipartite else g.num_nodes(c[2])
    if scipy_fmt is not None:
        import numpy as np
        import scipy.sparse as sp

        rows, cols, shape = (v, u, (n_dst, n_src)) if transpose else (u, v, (n_src, n_dst))
        mat = sp.coo_matrix((np.ones(int(u.shape[0])), (rows.cpu().numpy(), cols.cpu().numpy())), shape=shape)
        return mat.asformat(scipy_fmt)
    idx = torch.stack([v, u] if transpose else [u, v]).long()
    shape = (n_dst, n_src) if transpose else (n_src, n_dst)
    out = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1], device=idx.device), shape)
    return out.to(ctx) if ctx is not None else out


def _homogeneous(g, what):
    if len(g.ntypes) != 1 or len(g.canonical_etypes) != 1:
        raise DGLAMDError("%s only supports homogeneous graphs; convert with to_homogeneous first" % what)


def add_reverse_edges(g, readonly=None, copy_ndata=True, copy_edata=False, ignore_bipartite=False, exclude_self=True):  # noqa: ARG001
    """Every edge (u, v) followed by its reverse (v, u), appended behind the existing edges — self-loops are not doubled
    when ``exclude_self`` (transforms/functional.py add_reverse_edges); reverse edges get a copy of their edge's features
    when ``copy_edata``."""
    _homogeneous(g, "add_reverse_edges")
    u, v = g.edges()
    keep = (u != v) if exclude_self else torch.ones_like(u, dtype=torch.bool)
    ru, rv = v[keep], u[keep]
    out = graph((torch.cat([u, ru]), torch.cat([v, rv])), num_nodes=g.num_nodes(), idtype=g.idtype, device=g.device)
    if copy_ndata:
        for k, val in g._node_frames[0].items():
            out._node_frames[0][k] = val
    if copy_edata:
        for k, val in g._edge_frames[0].items():
            out._edge_frames[0][k] = torch.cat([val, val[keep]], 0)
    return out


def to_simple(g, return_counts="count", writeback_mapping=False, copy_ndata=True, copy_edata=False):  # noqa: ARG001
    """Parallel edges merged into one, edges ordered by (source, destination); ``edata[return_counts]`` = multiplicity
    (functional.py to_simple).  With ``writeback_mapping`` also returns, per original edge, the id of its merged edge."""
    _homogeneous(g, "to_simple")
    u, v = g.edges()
    n = max(g.num_nodes(), 1)
    key = u.long() * n + v.long()
    uniq, inverse, counts = torch.unique(key, sorted=True, return_inverse=True, return_counts=True)
    out = graph(((uniq // n).to(g.idtype), (uniq % n).to(g.idtype)), num_nodes=g.num_nodes(), idtype=g.idtype, device=g.device)
    if return_counts is not None:
        out.edata[return_counts] = counts
    if copy_ndata:
        for k, val in g._node_frames[0].items():
            out._node_frames[0][k] = val
    return (out, inverse.to(g.idtype)) if writeback_mapping else out


def to_bidirected(g, copy_ndata=False, readonly=None):  # noqa: ARG001
    """A simple graph with both directions of every edge (functional.py to_bidirected): reverse edges added, parallel edges
    merged; edge features are not carried (there is no single value for a merged edge)."""
    _homogeneous(g, "to_bidirected")
    both = add_reverse_edges(g, copy_ndata=copy_ndata, copy_edata=False, exclude_self=True)
    return to_simple(both, return_counts=None, copy_ndata=copy_ndata)


def _induced(g, node_ids_per_type, edge_ids_per_etype, relabel_nodes, store_ids):
    """Subgraph from per-etype edge ids (and, when relabelling, per-type node ids in their new order)."""
    nts, cets = g.ntypes, g.canonical_etypes
    dev, idt = g.device, g.idtype
    if relabel_nodes:
        maps = {}
        for n in nts:
            ids = node_ids_per_type[n].long()
            m = torch.full((max(g.num_nodes(n), 1),), -1, dtype=torch.long, device=dev)
            m[ids] = torch.arange(ids.shape[0], device=dev)
            maps[n] = m
        counts = {n: int(node_ids_per_type[n].shape[0]) for n in nts}
    else:
        counts = {n: g.num_nodes(n) for n in nts}
    data = {}
    for c in cets:
        u, v = g.edges(etype=c)
        e = edge_ids_per_etype[c].long()
        uu, vv = u[e].long(), v[e].long()
        if relabel_nodes:
            uu, vv = maps[c[0]][uu], maps[c[2]][vv]
        data[c] = (uu.to(idt), vv.to(idt))
    if len(cets) == 1 and len(nts) == 1:
        out = graph(data[cets[0]], num_nodes=counts[nts[0]], idtype=idt, device=dev)
    else:
        out = heterograph(data, counts, idtype=idt, device=dev)
    for n in nts:
        fr = g._node_frames[g.get_ntype_id(n)]
        of = out._node_frames[out.get_ntype_id(n)]
        for k, val in fr.items():
            of[k] = val[node_ids_per_type[n].long()] if relabel_nodes else val
        if store_ids and relabel_nodes:
            of[NID] = node_ids_per_type[n].to(idt)
    for c in cets:
        fr = g._edge_frames[g.get_etype_id(c)]
        of = out._edge_frames[out.get_etype_id(c)]
        for k, val in fr.items():
            of[k] = val[edge_ids_per_etype[c].long()]
        if store_ids:
            of[EID] = edge_ids_per_etype[c].to(idt)
    return out


def _per_type(g, x, names, what):
    if isinstance(x, dict):
        return {k: torch.as_tensor(v, device=g.device) for k, v in x.items()}
    if len(names) != 1:
        raise DGLAMDError("%s: a dict keyed by type is needed on a graph with several types" % what)
    return {names[0]: torch.as_tensor(x, device=g.device)}


def _ids_of(t, n):
    return torch.nonzero(t).reshape(-1) if t.dtype == torch.bool else t.reshape(-1)


def node_subgraph(g, nodes, relabel_nodes=True, store_ids=True, output_device=None):
    """Subgraph induced on the given nodes (ids or a boolean mask; a dict per node type on heterographs): every edge whose
    two ends are kept, in edge-id order (python/dgl/subgraph.py node_subgraph)."""
    nodes = _per_type(g, nodes, g.ntypes, "node_subgraph")
    ids = {n: _ids_of(nodes[n], g.num_nodes(n)) if n in nodes else torch.empty(0, dtype=torch.long, device=g.device) for n in g.ntypes}
    inside = {}
    for n in g.ntypes:
        m = torch.zeros(max(g.num_nodes(n), 1), dtype=torch.bool, device=g.device)
        m[ids[n].long()] = True
        inside[n] = m
    eids = {}
    for c in g.canonical_etypes:
        u, v = g.edges(etype=c)
        eids[c] = torch.nonzero(inside[c[0]][u.long()] & inside[c[2]][v.long()]).reshape(-1)
    out = _induced(g, ids, eids, relabel_nodes, store_ids)
    return out.to(output_device) if output_device is not None else out


def edge_subgraph(g, edges, relabel_nodes=True, store_ids=True, output_device=None):
    """Subgraph of the given edges (ids or a boolean mask; a dict per edge type), its nodes the edges' end points in
    ascending id order when relabelled (subgraph.py edge_subgraph)."""
    edges = _per_type(g, edges, g.canonical_etypes if len(g.canonical_etypes) == 1 else g.etypes, "edge_subgraph")
    eids = {}
    for c in g.canonical_etypes:
        t = edges.get(c, edges.get(c[1]))
        eids[c] = _ids_of(t, g.num_edges(c)) if t is not None else torch.empty(0, dtype=torch.long, device=g.device)
    ids = {}
    for n in g.ntypes:
        parts = []
        for c in g.canonical_etypes:
            u, v = g.edges(etype=c)
            e = eids[c].long()
            if c[0] == n:
                parts.append(u[e].long())
            if c[2] == n:
                parts.append(v[e].long())
        ids[n] = torch.unique(torch.cat(parts)) if parts else torch.empty(0, dtype=torch.long, device=g.device)
    out = _induced(g, ids, eids, relabel_nodes, store_ids)
    return out.to(output_device) if output_device is not None else out


def in_subgraph(g, nodes, relabel_nodes=False, store_ids=True, output_device=None):
    """Every inbound edge of the given nodes, all nodes kept unless ``relabel_nodes`` (subgraph.py in_subgraph)."""
    nodes = _per_type(g, nodes, g.ntypes, "in_subgraph")
    eids = {}
    for c in g.canonical_etypes:
        u, v = g.edges(etype=c)
        if c[2] in nodes:
            m = torch.zeros(max(g.num_nodes(c[2]), 1), dtype=torch.bool, device=g.device)
            m[_ids_of(nodes[c[2]], 0).long()] = True
            eids[c] = torch.nonzero(m[v.long()]).reshape(-1)
        else:
            eids[c] = torch.empty(0, dtype=torch.long, device=g.device)
    if not relabel_nodes:
        out = _induced(g, None, eids, False, store_ids)
    else:
        ids = {}
        for n in g.ntypes:
            parts = [_ids_of(nodes[n], 0).long()] if n in nodes else []
            for c in g.canonical_etypes:
                u, v = g.edges(etype=c)
                e = eids[c].long()
                if c[0] == n:
                    parts.append(u[e].long())
                if c[2] == n:
                    parts.append(v[e].long())
            ids[n] = torch.unique(torch.cat(parts)) if parts else torch.empty(0, dtype=torch.long, device=g.device)
        out = _induced(g, ids, eids, True, store_ids)
    return out.to(output_device) if output_device is not None else out


__all__ = ["add_self_loop", "remove_self_loop", "remove_edges", "reorder_graph", "add_reverse_edges", "to_simple", "to_bidirected",
           "node_subgraph", "edge_subgraph", "in_subgraph", "batch", "unbatch", "from_scipy", "bipartite_from_scipy",
           "adj_external"]
