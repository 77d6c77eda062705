"""Uniform neighbour sampling and message-flow blocks for mini-batch training.

Mirror of ``dgl.sampling.sample_neighbors`` (python/dgl/sampling/neighbor.py:222-395), ``dgl.to_block``
(python/dgl/transforms/functional.py) and ``dgl.dataloading.NeighborSampler.sample_blocks``
(python/dgl/dataloading/neighbor_sampler.py): the steps that run in front of the g-SpMM for every
mini-batch of GraphSAGE (BASELINE config 4).  The kernels are in csrc/sampling.hip; the blocks come out
with their in-edge CSR already built (rows = seeds), i.e. in the format the SpMM consumes — no COO
round trip.  Graphs with several relations (round 5): ``nodes`` / ``fanout`` / ``dst_nodes`` are given per
node / edge type as in the reference, every relation goes through the same two kernels, and a
block keeps one source and one destination node set per node type (R-GCN mini-batches).
"""
import torch

from . import _capi
from ._lib import DGLAMDError as _DGLError
from .graph_index import GraphIndex, Relation
from .heterograph import DGLGraph

NID = "_ID"   # dgl.NID / dgl.EID
EID = "_ID"


def _csc_of(g):
    rel = g._graph.relations[0]
    indptr, indices, eids = rel.csc()
    return rel, _capi.make_csr(indptr, indices, eids, rel.num_src), (indptr, indices, eids)


def _node_map(g, device, ntid=None):
    """Per-graph (per node type) dense node -> local-id scratch (int32, all -1 between calls)."""
    if ntid is None:
        m = getattr(g, "_sampling_node_map", None)
        if m is None or m.device != device:
            m = torch.full((g.num_nodes(),), -1, dtype=torch.int32, device=device)
            g._sampling_node_map = m
        return m
    maps = g.__dict__.setdefault("_sampling_node_maps", {})
    m = maps.get(ntid)
    if m is None or m.device != device:
        m = maps[ntid] = torch.full((max(1, g._graph.num_nodes(ntid)),), -1, dtype=torch.int32, device=device)
    return m


def _prob_of(prob, frame, rel, who):
    """The sampling weights of one relation as a contiguous float vector (None: uniform)."""
    if prob is None:
        return None
    p = frame[prob] if isinstance(prob, str) else prob
    # the kernels index `prob` by EDGE ID of g: anything else is an out-of-bounds device read
    # (reference: CHECK on the probability array's length, src/array/array.cc RowWiseSampling)
    if p.dim() == 0 or p.shape[0] != rel.num_edges or p.numel() != rel.num_edges:
        raise _DGLError("%s: prob must hold one value per edge of the graph (%d), got shape %s"
                        % (who, rel.num_edges, tuple(p.shape)))
    p = p.to(rel.device)
    return (p if p.dtype in (torch.float32, torch.float64) else p.float()).contiguous().reshape(-1)


def _per_etype(g, value, what):
    """``value`` (an int, or a dict keyed by edge type name / canonical edge type) -> one entry per relation."""
    if not isinstance(value, dict):
        return [value] * len(g.canonical_etypes)
    out = []
    for c in g.canonical_etypes:
        if c in value:
            out.append(value[c])
        elif c[1] in value:
            out.append(value[c[1]])
        else:
            raise _DGLError("%s is not given for edge type %s (a dict must name every edge type, neighbor.py:330-340)"
                            % (what, (c,)))
    return out


def _sample_neighbors_hetero(g, nodes, fanout, edge_dir, prob, replace, seed):
    """sample_neighbors on a graph with several relations (python/dgl/sampling/neighbor.py:300-395): relation by
    relation through the same kernels; a relation whose seed side has no nodes in ``nodes``, or whose fanout is 0, comes
    out empty.  The result keeps every node type and node count of ``g``."""
    if not isinstance(nodes, dict):
        if len(g.ntypes) != 1:
            raise _DGLError("Must specify node type when the graph is not homogeneous.")
        nodes = {g.ntypes[0]: nodes}
    for n in nodes:
        if n not in g.ntypes:
            raise _DGLError('Node type "{}" does not exist.'.format(n))
    fanouts = _per_etype(g, fanout, "fanout")
    rels = []
    picked = []
    for etid, (s_t, _, d_t) in enumerate(g.canonical_etypes):
        rel = g._graph.relations[etid]
        dev, idt = rel.device, rel.idtype
        seeds = nodes.get(d_t if edge_dir == "in" else s_t)
        f = int(fanouts[etid])
        if seeds is None or f == 0 or rel.num_edges == 0 or torch.as_tensor(seeds).numel() == 0:
            empty = torch.empty(0, dtype=idt, device=dev)
            rels.append(Relation(rel.num_src, rel.num_dst, empty, empty, idtype=idt, device=dev))
            picked.append(empty)
            continue
        seeds = torch.as_tensor(seeds).reshape(-1).to(device=dev, dtype=idt).contiguous()
        fmt = rel.csc() if edge_dir == "in" else rel.csr()
        csr = _capi.make_csr(fmt[0], fmt[1], fmt[2], rel.num_src if edge_dir == "in" else rel.num_dst)
        p = _prob_of(prob, g._edge_frames[etid], rel, "sample_neighbors")
        rng = int(seed) * 131 + etid
        if p is None:
            indptr, nbr, eids = _capi.sample_neighbors(csr, seeds, f, replace, rng)
        elif f < 0:         # every edge that CAN be picked: all of them, minus those of zero weight
            indptr, nbr, eids = _capi.sample_neighbors(csr, seeds, -1, False, rng)
        else:
            indptr, nbr, eids = _capi.sample_neighbors_weighted(csr, p, seeds, f, replace, rng)
        n_e = int(indptr[-1])
        own = torch.repeat_interleave(seeds, (indptr[1:] - indptr[:-1]).long(), output_size=n_e)
        nbr, eids = nbr[:n_e], eids[:n_e]
        if p is not None and f < 0:
            keep_e = p[eids.long()] > 0
            own, nbr, eids = own[keep_e], nbr[keep_e], eids[keep_e]
        src, dst = (nbr.contiguous(), own) if edge_dir == "in" else (own, nbr.contiguous())
        rels.append(Relation(rel.num_src, rel.num_dst, src, dst, idtype=idt, device=dev))
        picked.append(eids)
    gi = g._graph
    out = DGLGraph(GraphIndex([gi.num_nodes(i) for i in range(len(g._ntypes))], list(gi.metagraph.edges), rels),
                   g._ntypes, g.canonical_etypes, src_ntypes=g._src_ntype_ids, dst_ntypes=g._dst_ntype_ids)
    for etid, e in enumerate(picked):
        out._edge_frames[etid][EID] = e
    return out


def sample_neighbors(g, nodes, fanout, edge_dir="in", prob=None, replace=False, seed=None):
    """Frontier graph on the nodes of `g` holding, for every node in `nodes`, ``fanout`` of its
    inbound edges picked uniformly (all of them when it has fewer, or ``fanout == -1``).
    ``edata[dgl.EID]`` carries the original edge ids.

    ``seed=None`` (default) draws a fresh stream for every call from torch's global CPU
    generator, like the reference's advancing global RNG (so ``torch.manual_seed`` makes a run
    reproducible); an explicit integer pins the picks of this call."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    if edge_dir not in ("in", "out"):
        raise ValueError("edge_dir must be 'in' or 'out'")
    if len(g.canonical_etypes) != 1 or isinstance(nodes, dict) or isinstance(fanout, dict):
        return _sample_neighbors_hetero(g, nodes, fanout, edge_dir, prob, replace, seed)
    if edge_dir == "in":
        rel, csr, keep = _csc_of(g)
    else:  # outbound edges: the same kernel over the out-edge CSR (rows = source nodes)
        rel = g._graph.relations[0]
        keep = rel.csr()
        csr = _capi.make_csr(keep[0], keep[1], keep[2], rel.num_dst)
    if isinstance(prob, str):  # name of an edge feature, as in the reference
        prob = g.edata[prob]
    nodes = torch.as_tensor(nodes).reshape(-1).to(device=rel.device, dtype=rel.idtype).contiguous()   # (ids, a list or one id)
    if prob is None:
        indptr, nbr, eids = _capi.sample_neighbors(csr, nodes, int(fanout), replace, int(seed))
    else:
        # the kernels index `prob` by EDGE ID of g: anything else is an out-of-bounds device read
        # (reference: CHECK on the probability array's length, src/array/array.cc RowWiseSampling)
        if prob.dim() == 0 or prob.shape[0] != rel.num_edges or prob.numel() != rel.num_edges:
            raise _DGLError("sample_neighbors: prob must hold one value per edge of the graph "
                            "(%d), got shape %s" % (rel.num_edges, tuple(prob.shape)))
        p = prob.to(rel.device)
        if p.dtype not in (torch.float32, torch.float64):
            p = p.float()
        p = p.contiguous().reshape(-1)
        if fanout < 0:      # every edge that CAN be picked: all of them, minus those of zero weight
            indptr, nbr, eids = _capi.sample_neighbors(csr, nodes, -1, False, int(seed))
        else:
            indptr, nbr, eids = _capi.sample_neighbors_weighted(csr, p, nodes, int(fanout), replace, int(seed))
    n_e = int(indptr[-1])
    own = torch.repeat_interleave(nodes, (indptr[1:] - indptr[:-1]).long(), output_size=n_e)
    nbr, eids = nbr[:n_e], eids[:n_e]
    if prob is not None and fanout < 0:
        keep_e = p[eids.long()] > 0
        own, nbr, eids = own[keep_e], nbr[keep_e], eids[keep_e]
        n_e = int(eids.shape[0])
    src, dst = (nbr[:n_e].contiguous(), own) if edge_dir == "in" else (own, nbr[:n_e].contiguous())
    r = Relation(rel.num_src, rel.num_dst, src, dst, idtype=rel.idtype, device=rel.device)
    out = DGLGraph(GraphIndex([g.num_nodes()], [(0, 0)], [r]), ["_N"], [("_N", "_E", "_N")])
    out.edata[EID] = eids[:n_e]
    return out


def sample_neighbors_fused(g, nodes, fanout, edge_dir="in", prob=None, replace=False, seed=None):
    """``sample_neighbors`` followed by ``to_block`` on the seeds (python/dgl/sampling/neighbor.py
    sample_neighbors_fused, a CPU-only fusion of the two there): the block of one sampling step, its edge ids those of
    ``g``."""
    if edge_dir != "in":
        raise _DGLError("sample_neighbors_fused: only inbound sampling yields a block whose destinations are the seeds")
    frontier = sample_neighbors(g, nodes, fanout, edge_dir=edge_dir, prob=prob, replace=replace, seed=seed)
    if isinstance(nodes, dict):
        seeds = {n: torch.as_tensor(v, device=g.device).reshape(-1) for n, v in nodes.items()}
    else:
        seeds = torch.as_tensor(nodes, device=g.device).reshape(-1)
    blk = to_block(frontier, seeds)
    for etid in range(len(g.canonical_etypes)):
        fe = frontier._edge_frames[etid][EID]
        blk._edge_frames[etid][EID] = fe[blk._edge_frames[etid][EID].long()]
    return blk


def _make_block(indptr, local_src, num_src, num_dst, idtype, device):
    rel = Relation(num_src, num_dst, csc=(indptr, local_src, None), idtype=idtype, device=device)
    rel.transient = True
    blk = DGLGraph(GraphIndex([num_src, num_dst], [(0, 1)], [rel]), ["_N", "_N"], [("_N", "_E", "_N")],
                   src_ntypes=[0], dst_ntypes=[1])
    blk.is_block = True
    return blk


def _renumber(seeds, src, g, dev, ntid=None):
    """Block-local ids of ``src`` with ``seeds`` first and the new sources after them in ascending order: the
    ``dgla_to_block`` kernel on the GPU; for a graph that lives on the CPU (graph construction in tests and data
    pipelines, where the reference runs its own CPU ``ToBlock``) the same convention in torch index arithmetic."""
    if dev.type == "cuda":
        return _capi.to_block(seeds, src, _node_map(g, dev, ntid))
    n_nodes = g.num_nodes() if ntid is None else g._graph.num_nodes(ntid)
    mark = torch.full((max(1, n_nodes),), -1, dtype=torch.long)
    mark[seeds.long()] = torch.arange(seeds.shape[0])
    new = torch.unique(src[mark[src.long()] < 0])
    mark[new.long()] = seeds.shape[0] + torch.arange(new.shape[0])
    return mark[src.long()].to(seeds.dtype), torch.cat([seeds, new.to(seeds.dtype)]), int(seeds.shape[0] + new.shape[0])


def _renumber_into(src_nodes, src, n_nodes):
    """Block-local ids of ``src`` when the block's source nodes are GIVEN (``to_block(..., src_nodes=)``)."""
    mark = torch.full((max(1, n_nodes),), -1, dtype=torch.long, device=src.device)
    mark[src_nodes.long()] = torch.arange(src_nodes.shape[0], device=src.device)
    local = mark[src.long()]
    if bool((local < 0).any()):
        raise _DGLError("to_block: src_nodes does not hold every source node of the frontier")
    return local.to(src_nodes.dtype), src_nodes, int(src_nodes.shape[0])


def _copy_block_features(g, blk, src_nodes, dst_nodes, induced):
    """Node / edge features of the frontier follow its nodes and edges into the block, behind the ids
    (utils.extract_node_subframes_for_block / extract_edge_subframes; the ids win over a feature of the same name)."""
    t = len(g.ntypes)
    for i, n in enumerate(g.ntypes):
        fr = g._node_frames[i]
        for key, val in fr.items():
            if key == NID:
                continue
            blk._node_frames[i][key] = val[src_nodes[n].long()]
            blk._node_frames[t + i][key] = val[dst_nodes[n].long()]
    for etid in range(len(g.canonical_etypes)):
        fr = g._edge_frames[etid]
        for key, val in fr.items():
            if key == EID:
                continue
            blk._edge_frames[etid][key] = val[induced[etid].long()]


def _default_dst_nodes(g):
    """``to_block(g)`` without destination nodes: per node type the nodes with an inbound edge, ascending
    (transforms/functional.py to_block: ``F.unique`` of the relations' destination ids)."""
    out = {}
    for c in g.canonical_etypes:
        out.setdefault(c[2], []).append(g.edges(etype=c)[1])
    return {n: (torch.unique(torch.cat(v)) if v else torch.empty(0, dtype=g.idtype, device=g.device)) for n, v in out.items()}


def to_block(g, dst_nodes=None, include_dst_in_src=True, src_nodes=None):
    """``dgl.to_block`` (python/dgl/transforms/functional.py to_block): the block's destination nodes are ``dst_nodes``
    in the given order (default: per type the nodes with an inbound edge, ascending); its source nodes start with them
    (``include_dst_in_src``) followed by the other sources in ascending order, or are ``src_nodes`` as given.
    ``srcdata / dstdata[dgl.NID]`` and ``edata[dgl.EID]`` hold the ids in ``g``; the features of ``g`` follow its nodes
    and edges into the block.  Every edge of ``g`` must point at one of ``dst_nodes``.  (:class:`NeighborSampler` then
    replaces ``edata[dgl.EID]`` by the ORIGINAL graph's edge ids, like neighbor_sampler.py:170-172.)"""
    if dst_nodes is None:
        dst_nodes = _default_dst_nodes(g)
        if len(g.ntypes) == 1:
            dst_nodes = dst_nodes.get(g.ntypes[0], torch.empty(0, dtype=g.idtype, device=g.device))
    return _to_block_hetero(g, dst_nodes, include_dst_in_src, src_nodes)


def _rows_of(rel, dst_nodes):
    """In-edges of ``dst_nodes`` (in that order) out of the relation's in-edge CSR: (block indptr, CSC positions)."""
    dev, idt = rel.device, rel.idtype
    indptr = rel.csc()[0]
    deg = (indptr[1:] - indptr[:-1])[dst_nodes.long()]
    total = int(deg.sum())
    if total != rel.num_edges:
        raise ValueError("to_block: some edges of the frontier do not end in dst_nodes")
    blk_ptr = torch.zeros(dst_nodes.shape[0] + 1, dtype=idt, device=dev)
    blk_ptr[1:] = torch.cumsum(deg, 0)
    starts = indptr[:-1][dst_nodes.long()].long()
    pos = torch.repeat_interleave(starts - blk_ptr[:-1].long(), deg.long(), output_size=total) + \
        torch.arange(total, device=dev)
    return blk_ptr, pos


def _to_block_hetero(g, dst_nodes, include_dst_in_src=True, given_src=None):
    """``dgl.to_block`` on a frontier with several node / edge types (python/dgl/transforms/functional.py to_block,
    src/graph/transform/cuda/cuda_to_block.cu): ``dst_nodes`` = {node type: ids}; per node type the block's source nodes
    start with that type's destination nodes (include_dst_in_src), followed by the new sources of EVERY relation leaving
    that type; one relation per edge type, rows = the destination nodes of its destination type."""
    if not isinstance(dst_nodes, dict):
        if len(g.ntypes) != 1:
            raise _DGLError("to_block: dst_nodes must be a dict of node type -> ids on a graph with several node types")
        dst_nodes = {g.ntypes[0]: dst_nodes}
    if g.is_block or len(set(g.ntypes)) != len(g.ntypes):
        raise _DGLError("to_block: the frontier must have one node space per type (not itself a block)")
    nts, cets = g.ntypes, g.canonical_etypes
    dev, idt = g.device, g.idtype
    dst = {n: torch.as_tensor(dst_nodes[n]).to(device=dev, dtype=idt).contiguous() if n in dst_nodes
           else torch.empty(0, dtype=idt, device=dev) for n in nts}
    rows, src_global, orig = [], [], []
    for etid, (s_t, _, d_t) in enumerate(cets):
        rel = g._graph.relations[etid]
        if rel.num_edges == 0 or dst[d_t].shape[0] == 0:
            # no destination node of this relation's destination type: the relation comes out empty and its sources are
            # not collected (src/graph/transform/to_block.cc:271-277)
            ptr = torch.zeros(dst[d_t].shape[0] + 1, dtype=idt, device=dev)
            pos = torch.empty(0, dtype=torch.long, device=dev)
        else:
            ptr, pos = _rows_of(rel, dst[d_t])
        indices, eids = rel.csc()[1], rel.csc()[2]
        rows.append(ptr)
        src_global.append(indices[pos].contiguous())
        orig.append(pos.to(idt) if eids is None else eids[pos])     # ids in g (no map: edge id == CSC position)
    src_nodes, local = {}, [None] * len(cets)
    for ntid, n in enumerate(nts):
        ets = [i for i, c in enumerate(cets) if c[0] == n]
        cat = torch.cat([src_global[i] for i in ets]) if ets else torch.empty(0, dtype=idt, device=dev)
        if given_src is not None:
            gs = given_src[n] if isinstance(given_src, dict) else given_src
            loc, sn, _ = _renumber_into(torch.as_tensor(gs).to(device=dev, dtype=idt), cat, g._graph.num_nodes(ntid))
        elif dst[n].shape[0] == 0 and cat.shape[0] == 0:
            src_nodes[n] = dst[n]
            for i in ets:
                local[i] = cat
            continue
        else:
            first = dst[n] if include_dst_in_src else dst[n][:0]
            loc, sn, _ = _renumber(first, cat, g, dev, ntid)
        src_nodes[n] = sn
        off = 0
        for i in ets:
            k = src_global[i].shape[0]
            local[i] = loc[off:off + k].contiguous()
            off += k
    t = len(nts)
    rels = []
    for etid, (s_t, _, d_t) in enumerate(cets):
        r = Relation(src_nodes[s_t].shape[0], dst[d_t].shape[0], csc=(rows[etid], local[etid], None), idtype=idt, device=dev)
        r.transient = True
        rels.append(r)
    meta = [(nts.index(c[0]), t + nts.index(c[2])) for c in cets]
    gidx = GraphIndex([src_nodes[n].shape[0] for n in nts] + [dst[n].shape[0] for n in nts], meta, rels)
    blk = DGLGraph(gidx, nts + nts, cets, src_ntypes=list(range(t)), dst_ntypes=list(range(t, 2 * t)))
    blk.is_block = True
    for i, n in enumerate(nts):
        blk._node_frames[i][NID] = src_nodes[n]
        blk._node_frames[t + i][NID] = dst[n]
    for etid in range(len(cets)):
        blk._edge_frames[etid][EID] = orig[etid]
    _copy_block_features(g, blk, src_nodes, dst, orig)
    return blk


class NeighborSampler:
    """``NeighborSampler([15, 10])``: one block per layer, built from the output nodes inwards
    (neighbor_sampler.py: sample_blocks).  ``sample_blocks(g, seed_nodes)`` returns
    ``(input_nodes, output_nodes, blocks)``; ``blocks[i].srcdata[dgl.NID]`` /
    ``dstdata[dgl.NID]`` / ``edata[dgl.EID]`` hold the original ids."""

    def __init__(self, fanouts, replace=False, seed=0, prob=None):
        self.fanouts = [f if isinstance(f, dict) else int(f) for f in fanouts]
        self.replace = bool(replace)
        self.seed = int(seed)
        self.prob = prob  # name of an edge feature holding sampling weights (the reference's `prob`)
        self._calls = 0

    def _sample_blocks_hetero(self, g, seed_nodes):
        """Several relations: ``sample_neighbors`` + ``to_block`` per layer, as neighbor_sampler.py:150-175 composes
        them; fanouts may be dicts keyed by edge type."""
        if not isinstance(seed_nodes, dict):
            if len(g.ntypes) != 1:
                raise _DGLError("seed_nodes must be a dict of node type -> ids on a graph with several node types")
            seed_nodes = {g.ntypes[0]: seed_nodes}
        seeds = {n: torch.as_tensor(v).to(device=g.device, dtype=g.idtype).contiguous() for n, v in seed_nodes.items()}
        output_nodes = dict(seeds)
        blocks = []
        for layer, fanout in enumerate(reversed(self.fanouts)):
            rng = (self.seed * 1000003 + self._calls) * 64 + layer
            frontier = sample_neighbors(g, seeds, fanout, prob=self.prob, replace=self.replace, seed=rng)
            blk = to_block(frontier, seeds)
            for etid in range(len(g.canonical_etypes)):          # ids of the ORIGINAL graph (neighbor_sampler.py:170-172)
                fe = frontier._edge_frames[etid][EID]
                blk._edge_frames[etid][EID] = fe[blk._edge_frames[etid][EID].long()]
            blocks.insert(0, blk)
            seeds = {n: blk.srcnodes[n].data[NID] for n in g.ntypes}
        self._calls += 1
        return seeds, output_nodes, blocks

    def sample_blocks(self, g, seed_nodes):
        if len(g.canonical_etypes) != 1 or len(g.ntypes) != 1 or isinstance(seed_nodes, dict):
            return self._sample_blocks_hetero(g, seed_nodes)
        rel, csr, keep = _csc_of(g)
        dev, idt = rel.device, rel.idtype
        node_map = _node_map(g, dev)
        seeds = seed_nodes.to(device=dev, dtype=idt).contiguous()
        output_nodes = seeds
        blocks = []
        for layer, fanout in enumerate(reversed(self.fanouts)):
            rng = (self.seed * 1000003 + self._calls) * 64 + layer
            if self.prob is None:
                if not (fanout > 0 and seeds.shape[0] > 0):
                    indptr, src, eids = _capi.sample_neighbors(csr, seeds, fanout, self.replace, rng)
            else:
                p = g.edata[self.prob] if isinstance(self.prob, str) else self.prob
                if p.dim() == 0 or p.shape[0] != rel.num_edges or p.numel() != rel.num_edges:
                    raise _DGLError("NeighborSampler: prob must hold one value per edge of the graph "
                                    "(%d), got shape %s" % (rel.num_edges, tuple(p.shape)))
                p = p.to(dev)
                p = (p if p.dtype in (torch.float32, torch.float64) else p.float()).contiguous().reshape(-1)
                indptr, src, eids = _capi.sample_neighbors_weighted(csr, p, seeds, fanout, self.replace, rng)
            n = seeds.shape[0]
            if self.prob is None and fanout > 0 and n > 0:
                # ONE read-back per layer instead of two: renumber the whole fixed-size pick buffer
                # (its unused tail is filled with the first seed, which adds no node) and fetch the
                # number of picks and of source nodes together.  Same picks, same local ids.
                indptr, src, eids = _capi.sample_neighbors_padded(csr, seeds, None, fanout, self.replace, rng, None,
                                                                  sink_rows=1)
                local, src_nodes, num = _capi.to_block_padded(seeds, None, src, node_map, num_nodes=rel.num_src)
                n_e, num_src = (int(v) for v in torch.stack([indptr[n].long(), num[0]]).tolist())
                indptr, local, src_nodes = indptr[: n + 1], local[:n_e], src_nodes[:num_src]
            else:
                n_e = int(indptr[-1])   # one read-back here, one in to_block
                local, src_nodes, num_src = _capi.to_block(seeds, src[:n_e], node_map)
            blk = _make_block(indptr, local, num_src, seeds.shape[0], idt, dev)
            blk.srcdata[NID] = src_nodes
            blk.dstdata[NID] = seeds
            blk.edata[EID] = eids[:n_e]
            blocks.insert(0, blk)
            seeds = src_nodes
        self._calls += 1
        return seeds, output_nodes, blocks

    def sample_blocks_padded(self, g, seed_nodes, num_valid=None):
        """Static-shape ``sample_blocks``: no size is read back, every tensor has a shape that depends
        only on ``len(seed_nodes)`` and the fanouts, so the call — and the training step around it —
        can be captured in one hipGraph (``torch.cuda.graph``) and replayed.

        ``seed_nodes`` has B slots of which the first ``num_valid`` (int64 device tensor, None = all)
        are real.  Layer by layer (output side first) a block has D destination SLOTS + 64 SINK rows
        and D + D * fanout source slots: real picks first, then the sink rows' edges (pointing at the
        real destination nodes in turn); source slots past the block's ``num_src`` (device tensor) are
        padding holding node 0.  Real rows are built exactly as :meth:`sample_blocks` builds them (same picks
        for the FIRST call of a fresh sampler; afterwards this method draws from its device-side counter's
        stream, :meth:`sample_blocks` from its host-side one); padded / sink rows produce values nobody reads.
        Every layer needs at least 64 pick slots (``slots * fanout >= 64``) — the sink rows' edges live there;
        a smaller batch is refused.  Returns ``(input_nodes,
        num_input, output_nodes, blocks)``; ``blocks[i].num_src_valid`` / ``.num_dst_valid`` are the
        device-side counts.  Each call advances a DEVICE-side draw counter (``self.counter``), so a
        replayed graph samples fresh neighbours."""
        if len(g.canonical_etypes) != 1 or any(isinstance(f, dict) for f in self.fanouts):
            raise _DGLError("sample_blocks_padded: single-relation graphs only (one static shape per layer)")
        if min(self.fanouts) < 1:
            raise _DGLError("sample_blocks_padded needs positive fanouts")
        rel, csr, keep = _csc_of(g)
        dev, idt = rel.device, rel.idtype
        node_map = _node_map(g, dev)
        if getattr(self, "counter", None) is None or self.counter.device != dev:
            self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        seeds = seed_nodes.to(device=dev, dtype=idt).contiguous()
        output_nodes, nv = seeds, num_valid
        p = None
        if self.prob is not None:
            p = g.edata[self.prob] if isinstance(self.prob, str) else self.prob
            if p.dim() == 0 or p.shape[0] != rel.num_edges or p.numel() != rel.num_edges:
                raise _DGLError("NeighborSampler: prob must hold one value per edge of the graph "
                                "(%d), got shape %s" % (rel.num_edges, tuple(p.shape)))
            p = p.to(dev)
            p = (p if p.dtype in (torch.float32, torch.float64) else p.float()).contiguous().reshape(-1)
        blocks = []
        for layer, fanout in enumerate(reversed(self.fanouts)):
            rng = (self.seed * 1000003) * 64 + layer
            if seeds.shape[0] * fanout < _capi.SINK_ROWS:
                raise _DGLError("sample_blocks_padded: %d slots x fanout %d leave fewer than %d pick slots for the "
                                "sink rows (a block would have more destination than source slots); use at least "
                                "%d seed slots" % (seeds.shape[0], fanout, _capi.SINK_ROWS,
                                                   -(-_capi.SINK_ROWS // fanout)))
            indptr, src, eids = _capi.sample_neighbors_padded(csr, seeds, nv, fanout, self.replace, rng,
                                                              self.counter, prob=p)
            local, src_nodes, num_src = _capi.to_block_padded(seeds, nv, src, node_map, num_nodes=rel.num_src)
            d = seeds.shape[0] + _capi.SINK_ROWS
            blk = _make_block(indptr, local, src_nodes.shape[0], d, idt, dev)
            blk.srcdata[NID] = src_nodes
            blk.dstdata[NID] = src_nodes[:d]
            blk.edata[EID] = eids
            blk.num_dst_valid, blk.num_src_valid = nv, num_src
            blocks.insert(0, blk)
            seeds, nv = src_nodes, num_src
        self.counter += 1
        return seeds, nv, output_nodes, blocks
