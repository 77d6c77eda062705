#!/usr/bin/env python
"""Strong-scaling bench of the partitioned g-SpMM with SIMULATED ranks on ONE GPU.

The pool has single-GPU boxes only, so the k ranks of `bench.py --gpus k` are run one after the
other on the same device: partition -> per-rank shard -> ShardedSpMM.step with the halo rows
copied in-process.  What this measures is every rank's COMPUTE (own-column launch + halo-column
launch) and the exchange VOLUME; what it cannot measure is the RCCL all-to-all itself, so the
`modeled_*` fields price the exchange at a stated xGMI rate and say so.  One JSON line per
(variant, k).

    python benchmarks/bench_sharded_sim.py [--scale S] [--ks 1,2,4,8] [--partitioner range|kway]
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
from tests.graphgen import C2_EDGES, C2_FEAT, C2_NODES, synth_csr  # noqa: E402

# 7 xGMI links per GPU.  Two stated rates INTO a GPU (both MODELS; nothing here ran on a multi-GPU node):
#   conservative: 64 GB/s per direction and link at 70 % = 314 GB/s (round 2 / 3's figure)
#   optimistic:   153.6 GB/s bidirectional per link = 76.8 GB/s per direction, at 85 % = 457 GB/s
XGMI_IN_GBPS = 7 * 64 * 0.7
XGMI_IN_GBPS_HI = 7 * 76.8 * 0.85


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--ks", default="1,2,4,8")
    ap.add_argument("--variants", default="L,U")
    ap.add_argument("--partitioner", default="range", choices=["range", "kway"],
                    help="kway = the native partitioner under the communication-volume objective, order-aware")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from dgl_amd.parallel import (ShardedSpMM, SimulatedExchange, partition_assignment,
                                  partition_rows, shard_from_partition)

    dev = torch.device("cuda:0")
    n, e, f = C2_NODES // args.scale, C2_EDGES // args.scale, C2_FEAT
    for variant in args.variants.split(","):
        if variant == "C":   # 64 planted communities, ids shuffled (tools/partition_stats.py): ranges are useless here
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from partition_stats import community_graph
            ipc, ixc = community_graph(n, e, 64, 0.9, 7)
            g = {"indptr": ipc.to(torch.int32).to(dev), "indices": ixc.to(torch.int32).to(dev)}
        else:
            g = synth_csr(n, n, e, variant, seed=20250824, device=dev)
        torch.manual_seed(12345)
        x = torch.rand(n, f, device=dev) + 1
        base = None
        for k in [int(v) for v in args.ks.split(",")]:
            t0 = time.perf_counter()
            if args.partitioner == "kway" and k > 1:
                part, _ = partition_assignment(g["indptr"], g["indices"], k, seed=1, objtype="vol")
            else:
                bounds = partition_rows(g["indptr"].cpu(), k)
                part = torch.searchsorted(bounds[1:].contiguous(), torch.arange(n), right=True)
            t_part = time.perf_counter() - t0
            shards = [shard_from_partition(g["indptr"], g["indices"], part, k, r) for r in range(k)]
            ex = SimulatedExchange(shards)
            xs = [x[s["rows"]].contiguous() for s in shards]
            for r in range(k):
                ex.bind(r, xs[r])
            per_rank = []
            for r, s in enumerate(shards):
                op = ShardedSpMM(s, (f,), x.dtype, dev, exchange=ex, rank=r)
                out = torch.empty(s["n_local"], f, device=dev)
                op.step(xs[r], out)
                op.step(xs[r], out)
                torch.cuda.synchronize()

                def timed(fn):
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
                    ev[0].record()
                    for i in range(args.reps):
                        fn()
                        ev[i + 1].record()
                    torch.cuda.synchronize()
                    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(args.reps)]))

                t_loc = timed(lambda: op.spmm("local", s["local"], s["n_local"], xs[r], out, False))
                t_halo = timed(lambda: op.spmm("halo", s["halo"], s["n_halo"], op.halo, out, True)) \
                    if s["n_halo"] else 0.0
                per_rank.append({"edges": s["nnz"], "cut_edges": s["cut_edges"], "halo_rows": s["n_halo"],
                                 "local_ms": t_loc, "halo_ms": t_halo,
                                 "exchange_ms_modeled": s["n_halo"] * f * 4 / (XGMI_IN_GBPS * 1e6)})
                del op, out
            comp = max(p["local_ms"] + p["halo_ms"] for p in per_rank)
            # overlapped schedule: the exchange hides behind the own-column launch
            modeled = max(max(p["local_ms"], p["exchange_ms_modeled"]) + p["halo_ms"] for p in per_rank)
            hi = XGMI_IN_GBPS / XGMI_IN_GBPS_HI
            modeled_hi = max(max(p["local_ms"], p["exchange_ms_modeled"] * hi) + p["halo_ms"] for p in per_rank)
            if k == 1:
                base = comp
            print(json.dumps({
                "variant": variant, "k": k, "partitioner": args.partitioner, "partition_s": round(t_part, 2),
                "cut_fraction": sum(p["cut_edges"] for p in per_rank) / e,
                "halo_rows_max": max(p["halo_rows"] for p in per_rank),
                "exchange_MB_max_rank": max(p["halo_rows"] for p in per_rank) * f * 4 / 1e6,
                "compute_ms_max_rank": round(comp, 4),
                "compute_only_speedup_vs_k1": round(base / comp, 3) if base else None,
                "G_edges_per_s_compute_only": round(e / comp / 1e6, 2),
                "modeled_step_ms": round(modeled, 4),
                "modeled_speedup_vs_k1": round(base / modeled, 3) if base else None,
                "modeled_exchange_rate_GBps": XGMI_IN_GBPS,
                "modeled_step_ms_at_457GBps": round(modeled_hi, 4),
                "modeled_speedup_vs_k1_at_457GBps": round(base / modeled_hi, 3) if base else None,
                "per_rank": [{k2: (round(v, 4) if isinstance(v, float) else v) for k2, v in p.items()}
                             for p in per_rank]}), flush=True)
            del shards, ex, xs
        del g, x


if __name__ == "__main__":
    main()
