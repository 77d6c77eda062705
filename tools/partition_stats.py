#!/usr/bin/env python
"""Cut, exchanged rows (communication volume) and wall time of the native k-way partitioner
(csrc/partition.cc, host code) under both objectives, against contiguous edge-balanced row ranges, on the
bench graphs and on a planted-community graph with SHUFFLED ids (where ranges carry no information).  CPU only.

    python tools/partition_stats.py [--scale S] [--k 8] [--variants U,L,C]

One JSON line per graph: for every method the cut fraction, the halo rows summed over the parts (`volume`: what
a row-sharded SpMM pulls per step) and the largest part's halo rows, all RECOUNTED from the assignment.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd.parallel import partition_assignment, partition_rows  # noqa: E402
from tests.graphgen import C2_EDGES, C2_NODES, lognormal_degrees, synth_csr  # noqa: E402


def community_graph(n, e, communities, p_in, seed):
    """C2's degree sequence; a fraction p_in of every row's neighbours from the row's own community, the rest
    uniform; node ids SHUFFLED.  Returns (indptr, indices) int64."""
    rng = np.random.default_rng(seed)
    deg = lognormal_degrees(n, e, seed)
    comm = rng.integers(0, communities, n)
    order = np.argsort(comm, kind="stable")                 # members of a community, contiguous in `order`
    start = np.searchsorted(comm[order], np.arange(communities + 1))
    rows = np.repeat(np.arange(n), deg)
    inside = rng.random(e) < p_in
    size = (start[1:] - start[:-1])[comm[rows]]
    pick = start[comm[rows]] + (rng.random(e) * size).astype(np.int64)
    cols = np.where(inside, order[np.minimum(pick, n - 1)], rng.integers(0, n, e))
    key = np.argsort(rows * n + cols, kind="stable")
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    return torch.from_numpy(indptr), torch.from_numpy(cols[key])


def recount(ip, ix, part, k):
    ipn, ixn, p = ip.numpy(), ix.numpy(), part.numpy()
    n = len(p)
    rows = np.repeat(np.arange(n), np.diff(ipn))
    remote = p[rows] != p[ixn]
    keys = np.unique(p[rows][remote].astype(np.int64) * n + ixn[remote])
    per = np.bincount(keys // n, minlength=k)
    w = np.bincount(p, weights=np.diff(ipn) + 1, minlength=k)
    return {"cut": round(float(remote.mean()), 4), "volume_rows": int(per.sum()), "halo_rows_per_rank_max": int(per.max()),
            "halo_rows_per_rank_mean": int(per.mean()), "load_max_over_mean": round(float(w.max() / w.mean()), 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--variants", default="U,L,C")
    args = ap.parse_args()
    n, e = C2_NODES // args.scale, C2_EDGES // args.scale
    for variant in args.variants.split(","):
        if variant == "C":
            ip, ix = community_graph(n, e, 64, 0.9, 7)
            what = "64 planted communities, 90 % of a row's neighbours inside its community, ids shuffled"
        else:
            g = synth_csr(n, n, e, variant, device=torch.device("cpu"))
            ip, ix = g["indptr"].long(), g["indices"].long()
            what = "SURVEY 8d variant " + variant
        out = {"variant": variant, "graph": what, "nodes": n, "edges": e, "k": args.k,
               "threads": os.environ.get("DGLA_PARTITION_THREADS", "all")}
        bounds = partition_rows(ip, args.k)
        rng_part = torch.searchsorted(bounds[1:].contiguous(), torch.arange(n), right=True)
        out["contiguous_ranges"] = recount(ip, ix, rng_part, args.k)
        for name, kw in (("multilevel_cut", dict(objtype="cut", order_aware=False)),
                         ("multilevel_vol", dict(objtype="vol", order_aware=False)),
                         ("order_aware_vol", dict(objtype="vol", order_aware=True))):
            t0 = time.time()
            part, st = partition_assignment(ip, ix, args.k, seed=1, **kw)
            r = recount(ip, ix, part, args.k)
            r.update(seconds=round(time.time() - t0, 1), method=st["method"], refine_moves=st["refine_moves"])
            assert r["volume_rows"] == st["volume"]
            out[name] = r
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
