"""Synthetic graphs for the parity tests and bench.py.

Everything is generated from fixed seeds; nothing reads the reference tree or a dataset.
"""
import numpy as np
import torch

C2_NODES = 2_449_029   # ogbn-products (examples/pytorch/correct_and_smooth/README.md:27)
C2_EDGES = 61_859_140
C2_FEAT = 100


def lognormal_degrees(n, e, seed=20250824, mu=2.3, sigma=1.25, dmax=17_500):
    """Degree sequence with a heavy tail like ogbn-products: d_i ~ lognormal(mu, sigma),
    truncated to [0, dmax], rescaled and rounded so that sum(d) == e exactly."""
    rng = np.random.default_rng(seed)
    raw = np.minimum(rng.lognormal(mu, sigma, size=n), dmax)
    d = np.floor(raw * (e / raw.sum())).astype(np.int64)
    d = np.minimum(d, dmax)
    short = int(e - d.sum())
    if short > 0:  # hand the remainder to random rows (keeps the shape of the tail)
        idx = rng.integers(0, n, size=short)
        np.add.at(d, idx, 1)
    elif short < 0:
        nz = np.flatnonzero(d > 0)
        idx = rng.choice(nz, size=-short, replace=False)
        d[idx] -= 1
    assert d.sum() == e and d.min() >= 0
    return d


def synth_csr(n_rows, n_cols, n_edges, variant="U", seed=20250824, device="cpu",
              idtype=torch.int32, with_eids=False, sort_cols=True):
    """CSR in the "CSC role" (rows = destination nodes).

    variant U: column ids i.i.d. uniform (worst-case locality).
    variant L: 80 % of a row's neighbours within +-32k of the row id (community locality).
    variant C: 64 planted communities, node ids shuffled (90 % of a row's neighbours from its own community).
    Returns dict(indptr, indices, eids|None, num_rows, num_cols, nnz).
    """
    deg = lognormal_degrees(n_rows, n_edges, seed)
    indptr_np = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr_np[1:])
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    deg_t = torch.from_numpy(deg).to(dev)
    rows = torch.repeat_interleave(torch.arange(n_rows, device=dev), deg_t)
    cols = torch.randint(0, n_cols, (n_edges,), device=dev, generator=g)
    if variant == "L":
        local = torch.rand(n_edges, device=dev, generator=g) < 0.8
        off = torch.randint(-32768, 32769, (n_edges,), device=dev, generator=g)
        near = (rows * n_cols // max(n_rows, 1) + off).clamp_(0, n_cols - 1)
        cols = torch.where(local, near, cols)
        del local, off, near
    elif variant == "C":
        # 64 planted communities with SHUFFLED node ids: 90 % of a row's neighbours come from the row's own community
        # (uniform inside it), the rest are uniform — contiguous ranges cut this graph like variant U, a partitioner
        # that finds the communities does not (the graph a node-cut partitioner actually helps; VERDICT r5 Next #6b)
        ncomm, p_in = 64, 0.9
        comm = torch.randint(0, ncomm, (n_cols,), device=dev, generator=g)
        order = torch.argsort(comm, stable=True)                 # members of a community, contiguous in `order`
        start = torch.searchsorted(comm[order].contiguous(), torch.arange(ncomm + 1, device=dev))
        rc = comm[(rows * n_cols // max(n_rows, 1)).clamp_(max=n_cols - 1)]
        inside = torch.rand(n_edges, device=dev, generator=g) < p_in
        size = (start[1:] - start[:-1])[rc]
        pick = start[rc] + (torch.rand(n_edges, device=dev, generator=g) * size).long()
        cols = torch.where(inside, order[pick.clamp_(max=n_cols - 1)], cols)
        del comm, order, rc, inside, size, pick
    elif variant != "U":
        raise ValueError(variant)
    if sort_cols:
        key = rows * n_cols + cols
        key, _ = torch.sort(key)
        cols = key - rows * n_cols
        del key
    del rows
    out = {
        "indptr": torch.from_numpy(indptr_np).to(dev).to(idtype),
        "indices": cols.to(idtype),
        "eids": None,
        "num_rows": n_rows, "num_cols": n_cols, "nnz": n_edges,
    }
    if with_eids:
        out["eids"] = torch.randperm(n_edges, device=dev, generator=g).to(idtype)
    return out


def coo_to_csc(src, dst, num_dst, idtype=np.int32):
    """numpy helper: in-edge CSR (rows = dst) with the edge-id map, stable in edge order —
    what the reference's COO->CSC conversion produces (src/graph/unit_graph.cc:1418-1450)."""
    src = np.asarray(src)
    dst = np.asarray(dst)
    order = np.lexsort((src, dst))
    indptr = np.zeros(num_dst + 1, dtype=np.int64)
    np.add.at(indptr, dst + 1, 1)
    indptr = np.cumsum(indptr)
    return indptr.astype(idtype), src[order].astype(idtype), order.astype(idtype)


def coo_to_csr(src, dst, num_src, idtype=np.int32):
    """out-edge CSR (rows = src) with the edge-id map."""
    indptr, indices, eids = coo_to_csc(dst, src, num_src, idtype)
    return indptr, indices, eids
