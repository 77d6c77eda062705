#!/usr/bin/env python
"""Mini-batch GraphSAGE end to end (BASELINE.json configs[3] mechanics on the C2-shaped graph):
NeighborSampler(15, 10) -> feature gather -> 2 x SAGE-mean (copy_u + mean through update_all,
dense part in torch) -> cross-entropy -> backward (g-SpMM on the reversed blocks) -> SGD.

    python benchmarks/bench_sage.py [--batch 1024] [--steps 50]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 benchmarks/bench_sage.py

MI355X-first data placement: the whole graph AND the whole feature matrix are replicated on
every GPU (ogbn-products: 0.5 GB of CSC + 1 GB of features; even papers100M — 13 GB of CSC,
28 GB of bf16 features — fits the 288 GB of one MI355X several times over), so mini-batch
training shards by SEED NODES only and the one collective per step is the gradient all-reduce
(the reference shards features with METIS partitions and pulls halo rows per batch because
its target GPUs cannot hold them).  Prints one JSON line: seeds/s and sampled edges/s over
all ranks.
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

from tests.graphgen import C2_EDGES, C2_NODES, synth_csr  # noqa: E402


def sync_grads(grads, dist, world):
    """Gradient all-reduce (mean) of one step: ONE flat buffer, one collective — what DDP's bucket
    does for a model this small (reference: DistributedDataParallel in
    benchmarks/benchmarks/multigpu/bench_multigpu_sage.py:86-90).  Under the gloo flow-test backend
    device tensors are staged through host memory."""
    if dist is None or world == 1:
        return list(grads)
    flat = torch.cat([gr.reshape(-1) for gr in grads])
    if flat.is_cuda and dist.get_backend() == "gloo":
        h = flat.cpu()
        dist.all_reduce(h)
        flat = h.to(flat.device)
    else:
        dist.all_reduce(flat)
    flat /= world
    off, synced = 0, []
    for gr in grads:
        synced.append(flat[off:off + gr.numel()].view_as(gr))
        off += gr.numel()
    return synced


def reduce_host(values, op, dist, dev):
    """all-reduce a short list of Python floats with whatever the backend takes (RCCL: device
    tensors; gloo: host tensors)."""
    t = torch.tensor(values, dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
    dist.all_reduce(t, op=op)
    return [float(v) for v in t.tolist()]


class FeatureStore:
    """Input features of the mini-batch: ``replicated`` (every GPU holds all N rows — what 288 GB of
    HBM make natural), ``sharded`` (row i lives on rank i % world, local row i // world — the
    reference's NDArrayPartition 'remainder' mode, python/dgl/partition.py:474-640 — and every
    batch pulls its input rows with sparse_all_to_all_pull, python/dgl/cuda/nccl.py:98-183) or ``owner``
    (BASELINE configs[3] as worded: a node partition — here the native partitioner under the
    communication-volume objective — every rank trains on the seeds it OWNS, features live with their owner
    (reshuffled to contiguous ranges, python/dgl/partition.py:139-186 ``partition_graph_with_halo`` /
    python/dgl/distributed/partition.py:114-130 ``inner_node``) and a batch pulls only its NON-owned input rows)."""

    def __init__(self, feat_full, mode, rank, world, node_part=None, range_partition=None):
        from dgl_amd.parallel import NDArrayPartition

        self.mode, self.world, self.rank = mode, world, rank
        self.remote_rows = self.total_rows = 0
        self.owned = None
        if mode == "sharded" and world > 1:
            self.part = NDArrayPartition(feat_full.shape[0], world, mode="remainder")
            self.local = feat_full[rank::world].contiguous()
        elif mode == "owner" and world > 1:
            from dgl_amd.parallel import reshuffle
            orig_id, new_id, bounds = reshuffle(node_part.cpu(), world)
            dev = feat_full.device
            self.new_id = new_id.to(dev)
            self.lo, self.hi = int(bounds[rank]), int(bounds[rank + 1])
            self.owned = orig_id[self.lo:self.hi].to(dev)                    # old ids of my nodes, in new order
            self.local = feat_full[self.owned].contiguous()                  # my rows only
            self.part = range_partition(bounds) if range_partition is not None else \
                NDArrayPartition(feat_full.shape[0], world, mode="range", part_ranges=bounds)
        else:
            self.part, self.local = None, feat_full

    def fetch(self, ids):
        if self.mode == "owner" and self.part is not None:
            from dgl_amd.parallel import sparse_all_to_all_pull

            nid = self.new_id[ids]
            mine = (nid >= self.lo) & (nid < self.hi)
            out = torch.empty((ids.shape[0],) + tuple(self.local.shape[1:]), dtype=self.local.dtype,
                              device=self.local.device)
            out[mine] = self.local[nid[mine] - self.lo]
            remote = nid[~mine].contiguous()
            out[~mine] = sparse_all_to_all_pull(remote, self.local, self.part)   # (a collective: every rank calls it)
            self.remote_rows += int(remote.numel())
            self.total_rows += int(ids.numel())
            return out
        if self.part is None:
            if ids.numel() >= 32768 and self.local.is_cuda:   # the library's row gather: 29 vs 45 us at 180 k rows
                from dgl_amd import _capi
                return _capi.gather_rows(self.local, ids)
            return self.local[ids]
        from dgl_amd.parallel import sparse_all_to_all_pull

        return sparse_all_to_all_pull(ids, self.local, self.part)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--features", default="replicated", choices=["replicated", "sharded", "owner"],
                    help="sharded: features partitioned over the ranks by id %% world, pulled per batch; owner: node "
                         "partition, every rank trains on the seeds it owns and pulls only the non-owned input rows "
                         "(BASELINE configs[3] as worded)")
    ap.add_argument("--shape", default="products", choices=["products", "papers100m"],
                    help="papers100m: BASELINE configs[3]'s own size — 111 M nodes, 1.6 G edges, F = 128, 172 classes "
                         "(13 GB of int64 CSC + 57 GB of fp32 features resident on the one GPU)")
    ap.add_argument("--mode", default="both", choices=["eager", "graph", "both"],
                    help="graph: the whole step (sampling included) captured once in a hipGraph over padded, "
                         "static-shape blocks (NeighborSampler.sample_blocks_padded) and replayed")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    gloo = os.environ.get("DGLA_BENCH_BACKEND", "nccl") == "gloo"   # flow test: ranks may share a GPU
    if local_rank >= torch.cuda.device_count():
        if not gloo:
            raise SystemExit("bench_sage.py: rank %d has no GPU of its own" % local_rank)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import dgl_amd as dgl
    import dgl_amd.function as fn
    from dgl_amd._capi import SINK_ROWS
    from dgl_amd.graph_index import GraphIndex, Relation
    from dgl_amd.heterograph import DGLGraph

    n, e, f, classes = C2_NODES // args.scale, C2_EDGES // args.scale, 100, 47
    if args.shape == "papers100m":   # ogbn-papers100M: 111 059 956 nodes, 1 615 685 872 edges, 128 features, 172 classes
        n, e, f, classes = 111_059_956 // args.scale, 1_615_685_872 // args.scale, 128, 172
    # (columns need not be sorted inside a row for sampling; at 1.6 G edges the sort is the slow part)
    gs = synth_csr(n, n, e, "L", seed=20250824, device=dev, idtype=torch.int64,
                   sort_cols=args.shape != "papers100m")   # same graph on every rank
    rel = Relation(n, n, csc=(gs["indptr"], gs["indices"], None), idtype=torch.int64, device=dev)
    g = DGLGraph(GraphIndex([n], [(0, 0)], [rel]), ["_N"], [("_N", "_E", "_N")])
    torch.manual_seed(0)
    node_part, part_stats = None, None
    if args.features == "owner" and world > 1:
        from dgl_amd.parallel import partition_assignment
        node_part = torch.empty(n, dtype=torch.int64, device=dev)
        if rank == 0:   # host code, once per graph; the order-aware entry keeps contiguous ranges when they are better
            p0, part_stats = partition_assignment(gs["indptr"], gs["indices"], world, seed=1, objtype="vol")
            node_part.copy_(p0)
        if gloo:
            h = node_part.cpu()
            dist.broadcast(h, src=0)
            node_part = h.to(dev)
        else:
            dist.broadcast(node_part, src=0)
    store = FeatureStore(torch.rand(n, f, device=dev), args.features, rank, world, node_part=node_part)
    labels = torch.randint(0, classes, (n,), device=dev)
    params = [torch.randn(f, args.hidden, device=dev) * 0.05, torch.randn(f, args.hidden, device=dev) * 0.05,
              torch.randn(args.hidden, classes, device=dev) * 0.05, torch.randn(args.hidden, classes, device=dev) * 0.05]
    for p in params:
        p.requires_grad_(True)
    sampler = dgl.NeighborSampler([15, 10], seed=1 + rank)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)

    def sage(blk, h, ws, wn):
        with blk.local_scope():
            blk.srcdata["h"] = h
            blk.update_all(fn.copy_u("h", "m"), fn.mean("m", "n"))
            return h[: blk.num_dst_nodes()] @ ws + blk.dstdata["n"] @ wn

    edges = [0]

    def draw_seeds():
        if store.owned is not None:   # owner mode: a rank trains on the seeds its partition owns
            return store.owned[torch.randint(0, store.owned.numel(), (args.batch,), device=dev, generator=gen)]
        return torch.randint(0, n, (args.batch,), device=dev, generator=gen)

    def step():
        seeds = draw_seeds().unique()
        inp, out, blocks = sampler.sample_blocks(g, seeds)
        edges[0] += sum(b.num_edges() for b in blocks)
        h = store.fetch(inp)
        h = torch.relu(sage(blocks[0], h, params[0], params[1]))
        logits = sage(blocks[1], h, params[2], params[3])
        loss = torch.nn.functional.cross_entropy(logits, labels[out.long()])
        grads = torch.autograd.grad(loss, params)
        grads = sync_grads(grads, dist, world)
        with torch.no_grad():
            for p, gr in zip(params, grads):
                p -= 0.1 * gr
        return loss

    # ---- the same step on padded, static-shape blocks: nothing is read back, so everything from the
    # neighbour draws to the SGD update is ONE captured hipGraph; per step the host copies a fresh
    # seed batch into the static buffer, replays, and (N > 1) all-reduces the gradients.
    static_seeds = torch.zeros(args.batch, dtype=torch.int64, device=dev)
    edge_ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    static_grads = [torch.zeros_like(p) for p in params]
    static_loss = torch.zeros((), device=dev)

    def padded_body():
        inp, n_inp, out, blocks = sampler.sample_blocks_padded(g, static_seeds)
        for b in blocks:
            ip = b._graph.relations[0].csc()[0]
            d = b.num_dst_nodes() - SINK_ROWS
            edge_ctr.add_(ip[d:d + 1].long())         # real picks of the layer (where the sink rows start)
        h = store.fetch(inp)
        h = torch.relu(sage(blocks[0], h, params[0], params[1]))
        h = h[: blocks[1].num_src_nodes()]            # drop the outer block's sink row
        logits = sage(blocks[1], h, params[2], params[3])[: args.batch]
        loss = torch.nn.functional.cross_entropy(logits, labels[out.long()])
        grads = torch.autograd.grad(loss, params)
        if world == 1:
            with torch.no_grad():
                for p, gr in zip(params, grads):
                    p -= 0.1 * gr
        else:
            for sg, gr in zip(static_grads, grads):
                sg.copy_(gr)
        static_loss.copy_(loss.detach())

    graph = None

    def step_graph():
        static_seeds.copy_(torch.randint(0, n, (args.batch,), device=dev, generator=gen))
        graph.replay()
        if world > 1:
            grads = sync_grads(static_grads, dist, world)
            with torch.no_grad():
                for p, gr in zip(params, grads):
                    p -= 0.1 * gr
        return static_loss

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        edges[0] = 0
        edge_ctr.zero_()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0, loss

    results = {}
    if args.features in ("sharded", "owner") and world > 1:
        # the per-batch feature pull is a collective with data-dependent split sizes: not capturable
        if args.mode == "graph":
            raise SystemExit("--mode graph needs --features replicated (the sharded pull is not capturable)")
        args.mode = "eager"
    if args.mode in ("graph", "both"):  # (captured before any eager step has put autograd nodes on other streams)
        static_seeds.copy_(torch.randint(0, n, (args.batch,), device=dev, generator=gen))
        graph = dgl.CapturedStep(lambda: padded_body(), {}).graph   # warm-up on a side stream, then one recording
        dt, loss = timed(step_graph)
        results["graph"] = (dt, float(edge_ctr.item()), loss)
    if args.mode in ("eager", "both"):
        dt, loss = timed(step)
        results["eager"] = (dt, float(edges[0]), loss)
    best = "graph" if "graph" in results else "eager"
    dt, tot_edges, loss = results[best]
    # replicas must hold the same parameters after the same synchronised steps
    chk = [float(p.detach().double().sum()) for p in params]
    spread = 0.0
    if dist is not None:
        dt = reduce_host([dt], dist.ReduceOp.MAX, dist, dev)[0]
        tot_edges = reduce_host([tot_edges], dist.ReduceOp.SUM, dist, dev)[0]
        hi = reduce_host(chk, dist.ReduceOp.MAX, dist, dev)
        lo = reduce_host(chk, dist.ReduceOp.MIN, dist, dev)
        spread = max(abs(a - b) / max(abs(a), 1e-30) for a, b in zip(hi, lo))
    remote_frac = None
    if dist is not None and args.features == "owner":
        tot = reduce_host([float(store.remote_rows), float(store.total_rows)], dist.ReduceOp.SUM, dist, dev)
        remote_frac = tot[0] / max(tot[1], 1.0)
    if rank == 0:
        print(json.dumps({
            "workload": "2-layer GraphSAGE-mean mini-batch training step, fanouts (15, 10), batch %d per GPU, "
                        "graph N=%d E=%d (variant L), F=%d -> %d -> %d, fp32; graph replicated per GPU"
                        % (args.batch, n, e, f, args.hidden, classes),
            "features": args.features if world > 1 else "local",
            "remote_input_row_fraction": (remote_frac if world > 1 and args.features == "owner" else None),
            "remote_input_row_fraction_if_seeds_and_rows_were_spread_uniformly": ((world - 1) / world if world > 1 else None),
            "partition": part_stats,
            "resident_GB": {"csc": round((gs["indptr"].numel() + gs["indices"].numel()) * 8 / 1e9, 1),
                            "features": round(store.local.numel() * store.local.element_size() / 1e9, 1)},
            "param_checksum_rel_spread_across_ranks": spread,
            "step": "one hipGraph replay over padded static-shape blocks (sampling, block building, gather, "
                    "forward, backward, SGD inside the graph)" if best == "graph" else "eager (sizes read back per layer)",
            "ms_per_step_by_mode": {k: v[0] / args.steps * 1e3 for k, v in results.items()},
            "n_gpus": world, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
            "seeds_per_s": args.batch * world * args.steps / dt,
            "sampled_edges_per_s": tot_edges / dt, "final_loss": float(loss.detach())}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
