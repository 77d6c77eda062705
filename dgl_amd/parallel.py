"""Multi-GPU side of the hot path: halo feature pull over RCCL (one process per GPU).

The SpMM shards by destination rows: rank p owns rows R_p, their CSR slice
and output rows, and needs the feature rows of every column its slice references — its own
rows plus HALO rows owned by other ranks.  With a static graph the index lists are exchanged
once; per step only feature rows move, with a single ``all_to_all_single`` (on ROCm the
"nccl" backend is RCCL; grouped point-to-point traffic uses all xGMI links at once, unlike a
ring all-gather).

Reference counterpart: ``sparse_all_to_all_pull`` (python/dgl/cuda/nccl.py:98-183), which
re-sends split counts and indices on every call.
"""
import torch
import torch.distributed as dist


def _all_to_all(out, inp, out_splits, in_splits, group=None):
    dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


class HaloExchange:
    """Pulls remote feature rows into the tail of a local feature buffer.

    The local feature buffer has ``n_local + n_halo`` rows: rows ``[0, n_local)`` are owned
    by this rank, rows ``[n_local, n_local + n_halo)`` mirror rows of the peers, grouped by
    owner rank in ascending order.  ``requests[p]`` lists the owner-local row ids wanted from
    rank ``p`` in the order they appear in the halo block.
    """

    def __init__(self, n_local, n_halo, feat, device, seed=0, requests=None, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.n_local, self.n_halo, self.feat = int(n_local), int(n_halo), int(feat)
        self.device = torch.device(device)
        if requests is None:
            requests = synthetic_requests(self.rank, self.world, n_local, n_halo, seed)
        self.recv_splits = [int(requests[p].numel()) if p in requests else 0
                            for p in range(self.world)]
        assert sum(self.recv_splits) == self.n_halo, "requests must cover the halo block"
        # one-time index exchange: tell every owner which of its rows we need
        want_counts = torch.tensor(self.recv_splits, dtype=torch.int64, device=self.device)
        serve_counts = torch.empty_like(want_counts)
        _all_to_all(serve_counts, want_counts, None, None, group)
        self.send_splits = [int(v) for v in serve_counts.tolist()]
        want = torch.cat([requests[p].to(self.device, torch.int64) if p in requests
                          else torch.empty(0, dtype=torch.int64, device=self.device)
                          for p in range(self.world)]) if self.n_halo else \
            torch.empty(0, dtype=torch.int64, device=self.device)
        serve = torch.empty(sum(self.send_splits), dtype=torch.int64, device=self.device)
        _all_to_all(serve, want, self.send_splits, self.recv_splits, group)
        if serve.numel():
            assert int(serve.min()) >= 0 and int(serve.max()) < self.n_local
        self.serve_rows = serve          # local rows to send, grouped by destination rank
        self._send_buf = None

    def pull(self, x):
        """Fill ``x[n_local:]`` with the peers' current values of the requested rows."""
        assert x.shape[0] == self.n_local + self.n_halo and x.is_contiguous()
        if self.world == 1:
            return x
        if self._send_buf is None or self._send_buf.dtype != x.dtype or \
                self._send_buf.shape[1:] != x.shape[1:]:
            self._send_buf = torch.empty((self.serve_rows.numel(),) + tuple(x.shape[1:]),
                                         dtype=x.dtype, device=x.device)
        torch.index_select(x[: self.n_local], 0, self.serve_rows, out=self._send_buf)
        _all_to_all(x[self.n_local:], self._send_buf, self.recv_splits, self.send_splits,
                    self.group)
        return x

    def bytes_per_step(self, elem_size=4):
        return self.n_halo * self.feat * elem_size


def synthetic_requests(rank, world, n_local, n_halo, seed):
    """Equal share of the halo block from every peer; row ids sorted (what a partitioner's
    ``inner_node`` mask would give after relabelling, python/dgl/distributed/partition.py:114-130)."""
    if world == 1 or n_halo == 0:
        return {}
    per = n_halo // (world - 1)
    assert per * (world - 1) == n_halo, "n_halo must be a multiple of world - 1"
    g = torch.Generator()
    g.manual_seed(seed * 1000003 + rank)
    req = {}
    for p in range(world):
        if p == rank:
            continue
        req[p] = torch.sort(torch.randperm(n_local, generator=g)[:per])[0]
    return req


def partition_rows(indptr, world):
    """Contiguous destination-row ranges with (nearly) equal edge counts: the row split a
    METIS-relabelled graph reduces to once every partition is a contiguous id range
    (reference: reshuffle=True in python/dgl/distributed/partition.py).  Returns
    ``world + 1`` row boundaries."""
    nnz = int(indptr[-1])
    targets = torch.arange(0, world + 1, dtype=torch.float64) * (nnz / world)
    bounds = torch.searchsorted(indptr.to(torch.float64).cpu(), targets)
    bounds[0] = 0
    bounds[-1] = indptr.numel() - 1
    return bounds.to(torch.int64)


def shard_csr(indptr, indices, eids, bounds, rank):
    """Rank's slice of a CSR (rows = destination nodes) plus the halo bookkeeping.

    Returns a dict with the local CSR whose column ids are relabelled to
    ``[0, n_local)`` for owned columns followed by halo columns grouped by owner, and
    ``requests`` = {owner: owner-local row ids} suitable for :class:`HaloExchange`.
    Assumes a square graph whose columns are owned by the same row ranges.
    """
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    s, e = int(indptr[lo]), int(indptr[hi])
    loc_indptr = (indptr[lo:hi + 1] - indptr[lo]).clone()
    cols = indices[s:e].to(torch.int64)
    owner = torch.searchsorted(bounds[1:].to(cols.device), cols, right=True)
    is_local = owner == rank
    new_cols = torch.empty_like(cols)
    new_cols[is_local] = cols[is_local] - lo
    requests = {}
    base = hi - lo
    world = bounds.numel() - 1
    for p in range(world):
        if p == rank:
            continue
        m = owner == p
        if not bool(m.any()):
            continue
        uniq, inv = torch.unique(cols[m], return_inverse=True)
        new_cols[m] = base + inv
        requests[p] = uniq - int(bounds[p])
        base += uniq.numel()
    return {
        "indptr": loc_indptr.to(indptr.dtype), "indices": new_cols.to(indices.dtype),
        "eids": None if eids is None else eids[s:e].clone(),
        "n_local": hi - lo, "n_halo": base - (hi - lo), "requests": requests,
        "row_range": (lo, hi),
    }


# ---------------------------------------------------------------------------------------
# Node-cut partitioning (the reference: metis_partition_assignment + reshuffle,
# python/dgl/partition.py:278-397, python/dgl/distributed/partition.py reshuffle=True)
# ---------------------------------------------------------------------------------------
def partition_assignment(indptr, indices, k, balance_edges=True, imbalance=0.03, seed=0):
    """Part id of every node of a square graph given as a CSR (rows = destination nodes), from
    the native multilevel partitioner in libdgl_amd.so (csrc/partition.cc), which stands where
    METIS stands in the reference.  Host-side preprocessing: tensors are taken on the CPU.

    Returns ``(node_part int64[N], stats)`` with ``stats`` = cut edges, cut fraction,
    heaviest / average part weight and the number of multilevel levels."""
    import ctypes

    from ._lib import LIB, check_call

    ip = indptr.detach().cpu().contiguous()
    ix = indices.detach().cpu().to(ip.dtype).contiguous()
    n = ip.numel() - 1
    bits = {torch.int32: 32, torch.int64: 64}[ip.dtype]
    part = torch.empty(n, dtype=torch.int64)
    st = (ctypes.c_int64 * 4)()
    check_call(LIB.dgla_partition_kway(bits, n, ip.data_ptr(), ix.data_ptr(), int(k),
                                       float(imbalance), 1 if balance_edges else 0, int(seed),
                                       part.data_ptr(), ctypes.cast(st, ctypes.c_void_p)))
    nnz = max(int(ix.numel()), 1)
    stats = {"cut_edges": int(st[0]), "cut_fraction": int(st[0]) / nnz,
             "max_part_weight": int(st[1]), "avg_part_weight": int(st[2]), "levels": int(st[3])}
    return part, stats


def reshuffle(node_part, k):
    """Relabelling that makes every part a contiguous id range (reshuffle=True in the
    reference): returns ``(orig_id, new_id, bounds)`` with ``orig_id[new] = old``,
    ``new_id[old] = new`` and ``bounds`` the ``k + 1`` range boundaries.  Stable inside a
    part, so node order within a part is preserved."""
    node_part = node_part.cpu()
    orig_id = torch.argsort(node_part, stable=True)
    new_id = torch.empty_like(orig_id)
    new_id[orig_id] = torch.arange(orig_id.numel(), dtype=orig_id.dtype)
    counts = torch.bincount(node_part, minlength=k)
    bounds = torch.zeros(k + 1, dtype=torch.int64)
    bounds[1:] = torch.cumsum(counts, 0)
    return orig_id, new_id, bounds


def relabel_csr(indptr, indices, eids, orig_id, new_id):
    """The same square graph with node ``old`` renamed ``new_id[old]`` (rows reordered, column
    ids renamed and re-sorted inside each row; edge ids follow their edges).  Internal only:
    results are mapped back with ``orig_id``."""
    indptr, indices = indptr.cpu().long(), indices.cpu().long()
    deg = indptr[1:] - indptr[:-1]
    new_deg = deg[orig_id]
    new_indptr = torch.zeros_like(indptr)
    new_indptr[1:] = torch.cumsum(new_deg, 0)
    # gather the old rows in the new order
    starts = indptr[:-1][orig_id]
    pos = torch.repeat_interleave(starts - new_indptr[:-1], new_deg) + torch.arange(int(new_indptr[-1]))
    cols = new_id[indices[pos]]
    e = pos if eids is None else eids.cpu().long()[pos]
    rows = torch.repeat_interleave(torch.arange(orig_id.numel()), new_deg)
    order = torch.argsort(rows * orig_id.numel() + cols, stable=True)
    return new_indptr, cols[order], e[order]


def halo_fraction(indptr, indices, bounds):
    """Fraction of stored edges whose column is owned by another part and, per part, the number
    of distinct remote rows it has to pull (the per-step exchange volume in rows)."""
    indptr, indices = indptr.cpu().long(), indices.cpu().long()
    k = bounds.numel() - 1
    rows = torch.repeat_interleave(torch.arange(indptr.numel() - 1), indptr[1:] - indptr[:-1])
    row_part = torch.searchsorted(bounds[1:], rows, right=True)
    col_part = torch.searchsorted(bounds[1:], indices, right=True)
    remote = row_part != col_part
    halo_rows = [int(torch.unique(indices[remote & (row_part == p)]).numel()) for p in range(k)]
    return float(remote.float().mean()) if indices.numel() else 0.0, halo_rows
