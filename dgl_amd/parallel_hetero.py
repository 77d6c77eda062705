"""BASELINE configs[4] across GPUs: the per-relation g-SpMM of an R-GCN layer (sum over the
relations that share a destination node type) sharded by destination rows, one process per GPU.

Reference flow being replaced: the per-relation loop of SpMMCsrHetero
(src/array/cuda/spmm_hetero.cu:26-200, accumulation into the shared destination buffer
:150-158) on every partition, with the halo construction of partition_graph_with_halo
(python/dgl/partition.py:139-186) and a feature pull (python/dgl/cuda/nccl.py:98-183) in front.

Design (VERDICT r2 Missing #2):
  * every node type is partitioned (node -> rank); a rank owns the destination rows of its nodes
    in EVERY relation and the feature rows of its nodes of every type;
  * per destination type the rank's rows of all relations are stacked row-wise into TWO CSRs —
    own-column edges and halo-column edges — each with a relation byte per edge, so one
    dgla_spmm_csr_stacked launch per block replaces the relation loop (csrc/spmm_csr.hip.h MULTI);
  * halo rows are kept per SOURCE node type: the union over all relations reading that type, so a
    remote row wanted by several relations travels once per step;
  * step = [pack + all-to-all per source type (RCCL stream)] || own-column stacked launch ->
    halo-column stacked launch accumulating (dgl_amd.parallel.HaloExchange does the transport,
    chunked pipeline included).
Nothing here falls back to the CPU: the default kernel backend is the library's stacked kernel;
tests inject a torch backend under gloo exactly like tests/test_sharded_gloo.py.
"""
import torch

from .parallel import HaloExchange, SimulatedExchange


def _renumber(node_part, k):
    """new id of every node so that parts are contiguous ranges (reshuffle=True), bounds."""
    node_part = node_part.long()
    orig_id = torch.argsort(node_part, stable=True)
    new_id = torch.empty_like(orig_id)
    new_id[orig_id] = torch.arange(orig_id.numel(), device=orig_id.device)
    counts = torch.bincount(node_part, minlength=k)
    bounds = torch.zeros(k + 1, dtype=torch.int64, device=node_part.device)
    bounds[1:] = torch.cumsum(counts, 0)
    return orig_id, new_id, bounds


def shard_hetero_from_partition(num_nodes, meta, rels, node_parts, k, rank):
    """Rank ``rank``'s shard of a heterograph.

    ``num_nodes[t]`` nodes of type t; ``meta[r] = (src type, dst type)``; ``rels[r] = (indptr,
    indices)`` the in-edge CSR of relation r (rows = destination nodes of its dst type, columns =
    source ids of its src type); ``node_parts[t]`` the node -> rank assignment of type t.

    Returns a dict:
      ``rows[t]``      old ids of the owned nodes of type t (new order) — the local feature / output rows
      ``bounds[t]``    k + 1 range boundaries of type t in new ids
      ``blocks[d]``    for every destination type d that has relations: dict with
                       ``rels`` (relation ids stacked, in order), ``src`` (their source types),
                       ``local`` / ``halo`` = (indptr, indices, relid) stacked CSRs over the owned rows of d
                       (local columns = owner-local row of the source type, halo columns = row of the
                       source type's halo buffer)
      ``requests[s]``  {owner: owner-local rows of type s} in halo order (grouped by owner)
      ``n_halo[s]``, ``n_local[t]``, ``cut_edges``, ``nnz``
    """
    from .graph_index import stack_csc

    dev = rels[0][0].device
    T = len(num_nodes)
    ren = [_renumber(node_parts[t].to(dev), k) for t in range(T)]
    bounds_h = [[int(v) for v in ren[t][2].tolist()] for t in range(T)]
    lo = [bounds_h[t][rank] for t in range(T)]
    hi = [bounds_h[t][rank + 1] for t in range(T)]
    rows = [ren[t][0][lo[t]:hi[t]] for t in range(T)]
    n_local = [hi[t] - lo[t] for t in range(T)]

    # pass 1: every relation's rows in new order, columns in new ids, split own / remote
    per_rel = []
    remote_by_src = [[] for _ in range(T)]
    nnz = cut = 0
    for r, ((s, d), (indptr, indices)) in enumerate(zip(meta, rels)):
        ip = indptr.long()
        rows_old = rows[d]
        deg = (ip[1:] - ip[:-1])[rows_old]
        ptr = torch.zeros(n_local[d] + 1, dtype=torch.int64, device=dev)
        ptr[1:] = torch.cumsum(deg, 0)
        m = int(ptr[-1])
        pos = torch.repeat_interleave(ip[:-1][rows_old] - ptr[:-1], deg) + torch.arange(m, device=dev)
        cols = ren[s][1][indices[pos].long()]
        row_of = torch.repeat_interleave(torch.arange(n_local[d], device=dev), deg)
        order = torch.argsort(row_of * max(num_nodes[s], 1) + cols, stable=True)  # columns ascending inside a row
        cols = cols[order]
        own = (cols >= lo[s]) & (cols < hi[s])
        per_rel.append((row_of, cols, own))
        remote_by_src[s].append(cols[~own])
        nnz += m
        cut += int((~own).sum())

    # pass 2: one halo index per source type (union over the relations reading it)
    halo_ids, requests, n_halo = [], [], []
    for s in range(T):
        if remote_by_src[s]:
            uniq = torch.unique(torch.cat(remote_by_src[s]))     # ascending new ids = grouped by owner
        else:
            uniq = torch.empty(0, dtype=torch.int64, device=dev)
        halo_ids.append(uniq)
        n_halo.append(int(uniq.numel()))
        owner = torch.searchsorted(ren[s][2][1:], uniq, right=True)
        req = {}
        for p in range(k):
            mk = owner == p
            if p != rank and bool(mk.any()):
                req[p] = uniq[mk] - bounds_h[s][p]
        requests.append(req)

    # pass 3: stack the relations of every destination type
    idt = rels[0][0].dtype
    blocks = {}
    for d in range(T):
        rl = [r for r, (_, dd) in enumerate(meta) if dd == d]
        if not rl:
            continue

        def csr_of(row_of, new_cols, mask):
            cnt = torch.zeros(n_local[d], dtype=torch.int64, device=dev)
            cnt.index_add_(0, row_of[mask], torch.ones(int(mask.sum()), dtype=torch.int64, device=dev))
            ptr = torch.zeros(n_local[d] + 1, dtype=torch.int64, device=dev)
            ptr[1:] = torch.cumsum(cnt, 0)
            return ptr.to(idt), new_cols.to(idt), None

        loc, hal = [], []
        for r in rl:
            s = meta[r][0]
            row_of, cols, own = per_rel[r]
            loc.append(csr_of(row_of, cols[own] - lo[s], own))
            hal.append(csr_of(row_of, torch.searchsorted(halo_ids[s], cols[~own]), ~own))
        l_ip, l_ix, _, l_rel = stack_csc(loc, n_local[d], idt)
        h_ip, h_ix, _, h_rel = stack_csc(hal, n_local[d], idt)
        blocks[d] = {"rels": rl, "src": [meta[r][0] for r in rl],
                     "local": (l_ip, l_ix, l_rel), "halo": (h_ip, h_ix, h_rel)}
    return {"rows": rows, "bounds": [ren[t][2].cpu() for t in range(T)], "blocks": blocks,
            "requests": requests, "n_halo": n_halo, "n_local": n_local, "cut_edges": cut, "nnz": nnz,
            "num_types": T, "k": k, "rank": rank}


def _gpu_stacked_factory(device):
    """The product backend: dgla_spmm_csr_stacked on a stacked block, workspace + merge plan +
    pointer table cached per block."""
    from . import _capi
    state = {}

    def run(tag, block, n_cols, xs, out, accumulate):
        indptr, indices, relid = block
        if indices.numel() == 0:
            if not accumulate:
                out.zero_()
            return
        # ONE csr + workspace per block (keyed by the tag alone: activations get new addresses every layer
        # and step, a key made of data_ptr values grew without bound and pinned a workspace per entry,
        # ADVICE r3); only the small pointer table is rebuilt when the operand pointers change
        ptrs = tuple(int(x.data_ptr()) for x in xs)
        ent = state.get(tag)
        if ent is None:
            csr = _capi.make_csr(indptr, indices, None, n_cols)
            ws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", csr, xs[0], None, out),
                             dtype=torch.uint8, device=out.device)
            tabs = _capi.spmm_csr_stacked("copy_lhs", csr, relid, xs, None, out, ws, accumulate=accumulate)
            state[tag] = [csr, ws, tabs[0], ptrs]
            return
        csr, ws, tab, old = ent
        if old != ptrs:
            tab = None   # rebuilt by the call below
        tabs = _capi.spmm_csr_stacked("copy_lhs", csr, relid, xs, None, out, ws, u_table=tab, accumulate=accumulate,
                                      plan_valid=True)
        ent[2], ent[3] = tabs[0], ptrs

    return run


class SimulatedHeteroExchange:
    """In-process stand-in for the per-source-type HaloExchange objects over simulated ranks."""

    def __init__(self, shards, chunks=1):
        T = shards[0]["num_types"]
        self.per_type = [SimulatedExchange([{"requests": sh["requests"][s], "n_halo": sh["n_halo"][s]}
                                            for sh in shards], chunks) for s in range(T)]

    def bind(self, rank, x_local):
        for s, ex in enumerate(self.per_type):
            ex.bind(rank, x_local[s])


class ShardedHeteroSpMM:
    """One rank's part of ``out[d] = sum_{r: dst(r) = d} A_r @ X[src(r)]`` (copy_u / sum over the
    relations of an R-GCN layer).  ``step(x_local, out_local)`` takes / fills one tensor per node
    type (``None`` where a type has no features / receives nothing)."""

    def __init__(self, shard, feat_shape, dtype, device, group=None, backend=None, exchange=None, chunks=1):
        self.shard = shard
        self.device = torch.device(device)
        self.T = shard["num_types"]
        self.rank = shard["rank"]
        fs = tuple(feat_shape)
        self.halo = [torch.empty((shard["n_halo"][s],) + fs, dtype=dtype, device=self.device)
                     for s in range(self.T)]
        self.simulated = exchange
        srcs = sorted({s for b in shard["blocks"].values() for s in b["src"]})
        self.src_types = srcs
        if exchange is None:
            import torch.distributed as dist
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.exchange = {s: HaloExchange(shard["n_local"][s], shard["n_halo"][s], 1, self.device,
                                             requests=shard["requests"][s], group=group, chunks=chunks)
                             for s in srcs} if world > 1 else {}
        else:
            self.exchange = {}
        if any(getattr(ex, "chunks", 1) > 1 for ex in self.exchange.values()) or \
                (exchange is not None and exchange.per_type[0].chunks > 1):
            # the halo blocks' column ids follow the chunk-major layout of their source type
            for d, b in shard["blocks"].items():
                ip, ix, rel = b["halo"]
                ix = ix.clone()
                for j, s in enumerate(b["src"]):
                    o2n = (self.exchange[s].halo_old2new if exchange is None
                           else exchange.per_type[s].layout(self.rank)[1]).to(ix.device)
                    m = rel == j
                    if bool(m.any()):
                        ix[m] = o2n[ix[m].long()].to(ix.dtype)
                b["halo_cm"] = (ip, ix, rel)
        if backend is None:
            if self.device.type != "cuda":
                raise RuntimeError("ShardedHeteroSpMM: the kernel backend runs on a ROCm GPU (no CPU fallback)")
            backend = _gpu_stacked_factory(self.device)
        self.backend = backend

    def step(self, x_local, out_local):
        sh = self.shard
        works = {}
        if self.simulated is not None:
            for s in self.src_types:
                if sh["n_halo"][s]:
                    self.simulated.per_type[s].pull_into(self.rank, self.halo[s])
        else:
            for s, ex in self.exchange.items():
                works[s] = ex.pull_async(x_local[s], self.halo[s])
        for d, b in sh["blocks"].items():
            self.backend(("local", d), b["local"], max(sh["n_local"][s] for s in b["src"]) or 1,
                         [x_local[s] for s in b["src"]], out_local[d], False)
        for w in works.values():
            if w is not None:
                w.wait()
        for d, b in sh["blocks"].items():
            blk = b.get("halo_cm", b["halo"])
            if blk[1].numel():
                self.backend(("halo", d), blk, max(max(sh["n_halo"][s] for s in b["src"]), 1),
                             [self.halo[s] for s in b["src"]], out_local[d], True)
        return out_local
