"""Minimal graph index for the message-passing hot path.

The reference's ``HeteroGraphIndex`` (python/dgl/heterograph_index.py:19) is a full graph
engine; of it the g-SpMM / g-SDDMM path touches only: number_of_etypes / number_of_ntypes,
num_nodes, num_edges, dtype, ctx, metagraph.find_edge, reverse(), edge_subgraph and the
per-relation COO / CSR / CSC matrices (src/graph/unit_graph.cc:1418-1450,1588-1600,1748-1758).
That surface is what this module provides.  Index arrays are torch tensors on the ROCm
device; the C++ side only borrows their pointers through a unit-graph handle.

Format conversion (COO -> CSR / CSC) is done here with torch sorting primitives; it is
outside the timed hot path exactly like the reference's lazy ``GetInCSR`` and is listed as
the next row to move into HIP.
"""
import torch

from . import _ffi
from ._lib import DGLAMDError

ALL_FORMATS = ("coo", "csr", "csc")


def _idbits(dtype):
    return 32 if dtype == torch.int32 else 64


class Relation:
    """One edge type: a bipartite unit graph with lazily materialised sparse formats."""

    def __init__(self, num_src, num_dst, row=None, col=None, csr=None, csc=None,
                 idtype=torch.int64, device=None, formats=ALL_FORMATS):
        self.num_src, self.num_dst = int(num_src), int(num_dst)
        self.idtype = idtype
        self._coo = None if row is None else (row, col, None)  # (row, col, eids | None)
        self._csr = csr  # (indptr, indices, eids | None), rows = source nodes
        self._csc = csc  # (indptr, indices, eids | None), rows = destination nodes
        any_fmt = self._coo or self._csr or self._csc
        self.device = device if device is not None else any_fmt[0].device
        self.formats = tuple(formats)
        self._handle = None
        self._ws = None
        self._esm_ws = None
        self._set = set()
        self._rev = None
        self._degs = {}
        self.transient = False  # a sampled block, rebuilt every step (sparse_kernels._spmm_format)

    # ---- sizes ---------------------------------------------------------------------
    @property
    def num_edges(self):
        for f in (self._coo, self._csr, self._csc):
            if f is not None:
                return int(f[1].shape[0])
        return 0

    # ---- formats -------------------------------------------------------------------
    def coo(self):
        if self._coo is None:
            src_fmt = self._csr if self._csr is not None else self._csc
            indptr, indices, eids = src_fmt
            counts = (indptr[1:] - indptr[:-1]).long()
            major = torch.repeat_interleave(  # (output_size: no read-back of the total)
                torch.arange(counts.numel(), device=self.device, dtype=self.idtype), counts,
                output_size=int(indices.shape[0]))
            if self._csr is not None:
                self._coo = (major, indices, eids)
            else:
                self._coo = (indices, major, eids)
        return self._coo

    def _compress(self, major, minor, eids, n_major, n_minor=0):
        # stable sort by the major index keeps edge-id order inside each row, which fixes the
        # CSR position order (and therefore arg-max/min tie-breaking) deterministically
        if major.is_cuda:
            # native path: one radix sort + one fused gather / indptr kernel (csrc/coo2csr.hip
            # ≙ aten::COOToCSR<kDGLCUDA>, src/array/cuda/coo2csr.cu:28-110)
            from . import _capi
            return _capi.coo_to_csr(major.contiguous(), minor.contiguous(),
                                    None if eids is None else eids.contiguous(), n_major, n_minor)
        order = torch.argsort(major, stable=True)
        counts = torch.bincount(major.long(), minlength=n_major)
        indptr = torch.zeros(n_major + 1, dtype=self.idtype, device=self.device)
        indptr[1:] = torch.cumsum(counts, 0).to(self.idtype)
        new_eids = order.to(self.idtype) if eids is None else eids[order]
        return indptr, minor[order].contiguous(), new_eids.contiguous()

    _IDENTITY_CHECK_MIN_EDGES = 1 << 20

    @staticmethod
    def _drop_identity_map(fmt):
        """An edge-id map that says "edge id == position" (a COO that was already sorted by this
        format's major index, e.g. after dgl.reorder_graph(edge_permute_algo='dst')) is dropped:
        the kernels then skip the map altogether.  One pass + one synchronisation, at format
        build time (where the reference synchronises too) — only for graphs large enough for the
        map to cost anything: on sampled mini-batch blocks (built every step) the check's
        synchronisation would cost more than the map ever does."""
        indptr, indices, eids = fmt
        if eids is not None and eids.numel() >= Relation._IDENTITY_CHECK_MIN_EDGES and eids.is_cuda:
            ident = torch.arange(eids.numel(), device=eids.device, dtype=eids.dtype)
            if bool(torch.equal(eids, ident)):
                return indptr, indices, None
        return fmt

    def csr(self):
        if self._csr is None:
            row, col, eids = self.coo()
            self._csr = self._drop_identity_map(self._compress(row, col, eids, self.num_src, self.num_dst))
        return self._csr

    def csc(self):
        if self._csc is None:
            row, col, eids = self.coo()
            self._csc = self._drop_identity_map(self._compress(col, row, eids, self.num_dst, self.num_src))
        return self._csc

    def has(self, fmt):
        return getattr(self, "_" + fmt) is not None

    def allowed(self, fmt):
        return fmt in self.formats

    def in_degrees(self):
        if "in" not in self._degs:
            if self._csc is not None:
                ip = self._csc[0]
                self._degs["in"] = (ip[1:] - ip[:-1])
            else:
                self._degs["in"] = torch.bincount(self.coo()[1].long(), minlength=self.num_dst).to(self.idtype)
        return self._degs["in"]

    def out_degrees(self):
        if "out" not in self._degs:
            if self._csr is not None:
                ip = self._csr[0]
                self._degs["out"] = (ip[1:] - ip[:-1])
            else:
                self._degs["out"] = torch.bincount(self.coo()[0].long(), minlength=self.num_src).to(self.idtype)
        return self._degs["out"]

    # ---- the C++ handle ------------------------------------------------------------
    def handle(self, need):
        """Unit-graph handle with format `need` ('coo' | 'csr' | 'csc') registered."""
        if self._handle is None:
            self._handle = _ffi.get_global_func("dgl_amd._CAPI_UnitGraphCreate")(
                self.num_src, self.num_dst, _idbits(self.idtype))
        if need not in self._set:
            a, b, d = getattr(self, need)()
            fn = {"coo": "SetCOO", "csr": "SetCSR", "csc": "SetCSC"}[need]
            nd = _ffi.NDArray
            _ffi.get_global_func("dgl_amd._CAPI_UnitGraph" + fn)(
                self._handle, nd(a), nd(b), None if d is None else nd(d))
            self._set.add(need)
        return self._handle

    def ensure_workspace(self, nbytes):
        if nbytes and (self._ws is None or self._ws.numel() < nbytes):
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            _ffi.get_global_func("dgl_amd._CAPI_UnitGraphSetWorkspace")(
                self._handle, _ffi.NDArray(self._ws))

    def ensure_softmax_workspace(self, nbytes):
        """Scratch of the merge-path edge softmax (own plan, kept valid between calls)."""
        if nbytes and (self._esm_ws is None or self._esm_ws.numel() < nbytes):
            self._esm_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            _ffi.get_global_func("dgl_amd._CAPI_UnitGraphSetSoftmaxWorkspace")(
                self._handle, _ffi.NDArray(self._esm_ws))

    def __del__(self):
        try:
            if self._handle is not None:
                _ffi.get_global_func("dgl_amd._CAPI_UnitGraphFree")(self._handle)
        except Exception:  # interpreter shutdown
            pass

    # ---- derived graphs ------------------------------------------------------------
    def reverse(self):
        """Swap the roles of source and destination (UnitGraph::Reverse swaps in/out CSR,
        src/graph/unit_graph.cc:1748-1758): formats already built are shared, not copied."""
        if self._rev is None:
            r = Relation.__new__(Relation)
            r.num_src, r.num_dst, r.idtype, r.device = self.num_dst, self.num_src, self.idtype, self.device
            r._coo = None if self._coo is None else (self._coo[1], self._coo[0], self._coo[2])
            r._csr, r._csc = self._csc, self._csr
            r.formats = tuple({"csr": "csc", "csc": "csr", "coo": "coo"}[f] for f in self.formats)
            r._handle, r._ws, r._esm_ws, r._set, r._degs = None, None, None, set(), {}
            r.transient = self.transient
            r._rev = self
            self._rev = r
        return self._rev

    def edge_subgraph(self, eids):
        row, col, old = self.coo()
        if old is not None:
            raise DGLAMDError("edge_subgraph needs a COO in original edge order")
        eids = eids.to(self.device).long()
        return Relation(self.num_src, self.num_dst, row[eids].contiguous(), col[eids].contiguous(),
                        idtype=self.idtype, device=self.device, formats=self.formats)

    def to(self, device):
        mv = lambda t: None if t is None else tuple(None if x is None else x.to(device) for x in t)
        r = Relation(self.num_src, self.num_dst, csr=mv(self._csr), csc=mv(self._csc),
                     idtype=self.idtype, device=torch.device(device), formats=self.formats,
                     **({} if self._coo is None else
                        {"row": self._coo[0].to(device), "col": self._coo[1].to(device)}))
        # a COO derived from a CSR / CSC lists the edges in that format's order and carries the
        # edge ids as its third member: keep the full triple (as astype does)
        r._coo = mv(self._coo)
        return r

    def astype(self, idtype):
        cv = lambda t: None if t is None else tuple(None if x is None else x.to(idtype) for x in t)
        r = Relation(self.num_src, self.num_dst, csr=cv(self._csr), csc=cv(self._csc),
                     idtype=idtype, device=self.device, formats=self.formats)
        r._coo = cv(self._coo)
        return r


def stack_csc(cscs, num_rows, idtype):
    """Row-wise concatenation of several in-edge CSRs over the same destination nodes: row r
    lists relation 0's edges into r, then relation 1's, ...  Returns
    ``(indptr, indices, eids, rel)`` with ``eids`` = relation-local edge id and ``rel`` = uint8
    relation index of every stacked position.  Done once per (graph, relation set) with torch
    primitives and cached, like the reference's lazily built CSC (unit_graph.cc:1418-1450)."""
    dev = cscs[0][0].device
    degs = [(ip[1:] - ip[:-1]).long() for ip, _, _ in cscs]
    total = torch.stack(degs).sum(0)
    indptr = torch.zeros(num_rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(total, 0, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = torch.empty(nnz, dtype=idtype, device=dev)
    eids = torch.empty(nnz, dtype=idtype, device=dev)
    rel = torch.empty(nnz, dtype=torch.uint8, device=dev)
    before = torch.zeros(num_rows, dtype=torch.int64, device=dev)
    rows = torch.arange(num_rows, device=dev)
    for k, ((ip, ix, ei), deg) in enumerate(zip(cscs, degs)):
        n_k = int(ix.shape[0])
        if n_k == 0:
            continue
        r = torch.repeat_interleave(rows, deg)
        pos = (indptr[:-1] + before)[r] + (torch.arange(n_k, device=dev) - ip.long()[r])
        indices[pos] = ix
        eids[pos] = ei if ei is not None else torch.arange(n_k, dtype=idtype, device=dev)
        rel[pos] = k
        before += deg
    return indptr.to(idtype), indices, eids, rel


class MetaGraph:
    def __init__(self, edges):
        self.edges = list(edges)  # etype id -> (src ntype id, dst ntype id)

    def find_edge(self, etype):
        return self.edges[etype]


class GraphIndex:
    """The part of HeteroGraphIndex the hot path reads."""

    def __init__(self, num_nodes_per_type, meta_edges, relations):
        self._num_nodes = [int(n) for n in num_nodes_per_type]
        self.metagraph = MetaGraph(meta_edges)
        self.relations = list(relations)
        self._rev = None
        self._stacked = {}
        self._hetero_handle = None
        self._hetero_fmts = None

    def stacked(self, etypes):
        """(Relation over the stacked CSC, uint8 rel array) for relations `etypes`, which must
        share their destination node type; built on first use and cached."""
        key = tuple(etypes)
        if key not in self._stacked:
            d = self.metagraph.find_edge(key[0])[1]
            rels = [self.relations[et] for et in key]
            n = self._num_nodes[d]
            indptr, indices, eids, rel = stack_csc([r.csc() for r in rels], n, rels[0].idtype)
            stk = Relation(max(r.num_src for r in rels), n, csc=(indptr, indices, eids),
                           idtype=rels[0].idtype, device=rels[0].device, formats=("csc",))
            self._stacked[key] = (stk, rel)
        return self._stacked[key]

    def hetero_handle(self, fmts):
        """Heterograph handle (dgl_amd._CAPI_HeteroGraphCreate) over the relations' unit-graph
        handles, with format ``fmts[et]`` registered for relation ``et`` — what the reference's
        HeteroGraphRef argument gives its C++ side: NumEdgeTypes, meta_graph()->FindEdge and the
        per-relation matrices (src/array/kernel.cc:563-601)."""
        handles = [r.handle(f) for r, f in zip(self.relations, fmts)]
        if self._hetero_handle is None:
            flat = []
            for h, (s, d) in zip(handles, self.metagraph.edges):
                flat += [h, int(s), int(d)]
            self._hetero_handle = _ffi.get_global_func("dgl_amd._CAPI_HeteroGraphCreate")(
                len(self._num_nodes), *flat)
        return self._hetero_handle

    def __del__(self):
        try:
            if self._hetero_handle is not None:
                _ffi.LIB.DGLObjectFree(self._hetero_handle.handle)
        except Exception:  # interpreter shutdown
            pass

    def number_of_etypes(self):
        return len(self.relations)

    def number_of_ntypes(self):
        return len(self._num_nodes)

    def num_nodes(self, ntype):
        return self._num_nodes[ntype]

    def num_edges(self, etype):
        return self.relations[etype].num_edges

    @property
    def dtype(self):
        return self.relations[0].idtype if self.relations else torch.int64

    @property
    def ctx(self):
        return self.relations[0].device if self.relations else torch.device("cpu")

    def reverse(self):
        if self._rev is None:
            g = GraphIndex(self._num_nodes, [(d, s) for s, d in self.metagraph.edges],
                           [r.reverse() for r in self.relations])
            g._rev = self
            self._rev = g
        return self._rev

    def edge_subgraph(self, eids_per_etype, preserve_nodes=True):
        assert preserve_nodes
        return GraphIndex(self._num_nodes, self.metagraph.edges,
                          [r.edge_subgraph(e) for r, e in zip(self.relations, eids_per_etype)])

    def get_relation_graph(self, etype):
        s, d = self.metagraph.find_edge(etype)
        if s == d:
            return GraphIndex([self._num_nodes[s]], [(0, 0)], [self.relations[etype]])
        return GraphIndex([self._num_nodes[s], self._num_nodes[d]], [(0, 1)], [self.relations[etype]])
