#!/usr/bin/env python
"""Cut, halo rows per rank and wall time of the native k-way partitioner (csrc/partition.cc, host
code) against contiguous edge-balanced row ranges on the bench graphs.  CPU only.

    python tools/partition_stats.py [--scale S] [--k 8] [--variants U,L]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd.parallel import halo_fraction, partition_assignment, partition_rows, relabel_csr, reshuffle  # noqa: E402
from tests.graphgen import C2_EDGES, C2_NODES, synth_csr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--variants", default="U,L")
    args = ap.parse_args()
    n, e = C2_NODES // args.scale, C2_EDGES // args.scale
    for variant in args.variants.split(","):
        g = synth_csr(n, n, e, variant, device=torch.device("cpu"))
        ip, ix = g["indptr"].long(), g["indices"].long()
        cut_r, halo_r = halo_fraction(ip, ix, partition_rows(ip, args.k))
        t0 = time.time()
        part, st = partition_assignment(ip, ix, args.k, seed=1, order_aware=False)
        dt = time.time() - t0
        orig_id, new_id, bounds = reshuffle(part, args.k)
        ip2, ix2, _ = relabel_csr(ip, ix, None, orig_id, new_id)
        cut_m, halo_m = halo_fraction(ip2, ix2, bounds)
        print(json.dumps({"variant": variant, "nodes": n, "edges": e, "k": args.k,
                          "multilevel": {"seconds": round(dt, 1), "cut": round(float(cut_m), 4),
                                         "halo_rows_per_rank_max": int(max(halo_m)), "stats": st},
                          "contiguous_ranges": {"cut": round(float(cut_r), 4),
                                                "halo_rows_per_rank_max": int(max(halo_r))},
                          "threads": os.environ.get("DGLA_PARTITION_THREADS", "all")}), flush=True)


if __name__ == "__main__":
    main()
