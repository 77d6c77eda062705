#!/usr/bin/env python
"""BASELINE configs[4]: R-GCN per-relation g-SpMM on a synthetic heterograph (8 edge types x 12.5 M
edges on 10 M nodes, feat = 256, bf16), on 1 GPU (ONE stacked launch) or sharded over N GPUs
(dgl_amd.parallel_hetero.ShardedHeteroSpMM: destination rows + features sharded by node ranges, one
halo per source node type pulled with all_to_all over RCCL, two stacked launches per step).

    python benchmarks/bench_rgcn.py --gpus N [--steps K] [--warmup W] [--scale S] [--chunks C]

`--gpus N` without a launcher starts its own N ranks (like bench.py).  Prints one JSON line:
edges/s over all ranks (strong scaling of the one graph), halo rows / bytes per rank, parity of
every rank's rows against the single-GPU stacked launch (16-bit tolerance).  Reference flow:
src/array/cuda/spmm_hetero.cu:26-200 per partition + python/dgl/partition.py:139-186 halos.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from tests.graphgen import synth_csr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=1, help="pipeline chunks of the halo exchange")
    args = ap.parse_args()
    gloo = os.environ.get("DGLA_BENCH_BACKEND", "nccl") == "gloo"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import bench

        raise SystemExit(bench.spawn_ranks(args.gpus, gloo, script=__file__))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench_rgcn.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if local_rank >= torch.cuda.device_count():
        if not gloo:
            raise SystemExit("bench_rgcn.py: rank %d has no GPU of its own" % local_rank)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo") if gloo else dist.init_process_group("nccl", device_id=dev)

    from dgl_amd import _capi
    from dgl_amd.graph_index import stack_csc
    from dgl_amd.parallel_hetero import ShardedHeteroSpMM, shard_hetero_from_partition

    n, e, f, r = 10_000_000 // args.scale, 12_500_000 // args.scale, 256, 8
    torch.manual_seed(3)
    x = (torch.rand(n, f, device=dev) + 1).to(torch.bfloat16)
    gs = [synth_csr(n, n, e, "U", seed=100 + k, device=dev) for k in range(r)]   # same graph on every rank
    rels = [(g["indptr"], g["indices"]) for g in gs]
    meta = [(0, 0)] * r

    # the single-GPU answer (also the N = 1 measurement): one stacked launch over all relations
    ip, ix, ei, relid = stack_csc([(a, b, None) for a, b in rels], n, torch.int32)
    scsr = _capi.make_csr(ip, ix, ei, n)
    full = torch.empty(n, f, device=dev, dtype=torch.bfloat16)
    sws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", scsr, x, None, full), dtype=torch.uint8,
                      device=dev)
    tabs = _capi.spmm_csr_stacked("copy_lhs", scsr, relid, [x] * r, None, full, sws)

    if world == 1:
        def step():
            _capi.spmm_csr_stacked("copy_lhs", scsr, relid, [x] * r, None, full, sws, u_table=tabs[0], plan_valid=True)
        info = {}
    else:
        bounds = torch.linspace(0, n, world + 1).long()
        part = torch.searchsorted(bounds[1:].contiguous(), torch.arange(n), right=True).to(dev)  # contiguous ranges
        sh = shard_hetero_from_partition([n], meta, rels, [part], world, rank)
        op = ShardedHeteroSpMM(sh, (f,), torch.bfloat16, dev, chunks=args.chunks)
        x_loc = [x[sh["rows"][0]].contiguous()]
        out = [torch.empty(sh["n_local"][0], f, device=dev, dtype=torch.bfloat16)]

        def step():
            op.step(x_loc, out)
        info = {"rank": rank, "rows": sh["n_local"][0], "edges": sh["nnz"], "cut_edges": sh["cut_edges"],
                "halo_rows": sh["n_halo"][0], "halo_bytes": sh["n_halo"][0] * f * 2}
        del ip, ix, ei, relid, scsr, sws

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    res = None
    if dist is not None:
        sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
        from bench_sage import reduce_host

        dt = reduce_host([dt], dist.ReduceOp.MAX, dist, dev)[0]
        ref = full[sh["rows"][0]].float()
        err = float(((out[0].float() - ref).abs() / ref.abs().clamp_min(1e-30)).max())
        err = reduce_host([err], dist.ReduceOp.MAX, dist, dev)[0]
        infos = [None] * world
        dist.all_gather_object(infos, info)
    if rank == 0:
        ms = dt / args.steps * 1e3
        res = {"workload": "configs[4]: R-GCN hetero copy_u+sum, %d relations x %d edges on %d nodes, feat=%d, bf16"
                           % (r, e, n, f),
               "n_gpus": world, "steps": args.steps, "ms_per_step": ms, "edges_per_s": r * e / (ms * 1e-3),
               "scaling": "strong", "dtype": "bf16"}
        if world > 1:
            res.update({"parallelism": "%d-way contiguous node ranges, rows + features sharded, halo per source "
                                       "type by all_to_all (chunks=%d)" % (world, args.chunks),
                        "parity_max_rel_err_vs_single_gpu_stacked_launch": err, "per_rank": infos,
                        "cut_fraction": sum(i["cut_edges"] for i in infos) / (r * e),
                        "halo_rows_max": max(i["halo_rows"] for i in infos)})
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
