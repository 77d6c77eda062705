"""BASELINE configs[3] with more than one rank: the gradient all-reduce of
benchmarks/bench_sage.py under gloo on the CPU, and the whole mini-batch training flow with two
ranks sharing the test box's one GPU — features replicated and features sharded over the ranks
(NDArrayPartition 'remainder' + sparse_all_to_all_pull per batch, the reference's layout,
python/dgl/cuda/nccl.py:98-183)."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
        import bench_sage

        g = torch.Generator().manual_seed(10 + rank)
        grads = [torch.rand(3, 4, generator=g, dtype=torch.float64), torch.rand(5, generator=g, dtype=torch.float64)]
        synced = bench_sage.sync_grads(grads, dist, world)
        # the mean over the ranks, shapes kept, identical on every rank
        want = []
        for k, shape in enumerate([(3, 4), (5,)]):
            acc = torch.zeros(shape, dtype=torch.float64)
            for r in range(world):
                gr = torch.Generator().manual_seed(10 + r)
                parts = [torch.rand(3, 4, generator=gr, dtype=torch.float64), torch.rand(5, generator=gr, dtype=torch.float64)]
                acc += parts[k]
            want.append(acc / world)
        for a, b in zip(synced, want):
            assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-14)
        assert bench_sage.reduce_host([float(rank)], dist.ReduceOp.MAX, dist, "cpu") == [float(world - 1)]
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_bench_sage_gradient_sync_under_gloo(world):
    port = 25100 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


class _RangeDouble:
    """torch stand-in for NDArrayPartition(mode='range') on CPU tensors (the product class is GPU-only)."""

    def __init__(self, bounds):
        self.b = torch.as_tensor(bounds).long()

    def generate_permutation(self, idx):
        part = torch.searchsorted(self.b[1:].contiguous(), idx.long(), right=True)
        perm = torch.argsort(part, stable=True)
        return perm, torch.bincount(part, minlength=self.b.numel() - 1).to(torch.int64)

    def map_to_local(self, idx):
        part = torch.searchsorted(self.b[1:].contiguous(), idx.long(), right=True)
        return idx - self.b[part]


def _owner_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import cpu_backends
    cpu_backends.install()
    try:
        sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
        import bench_sage

        n, f = 5000, 7
        torch.manual_seed(0)
        feat = torch.rand(n, f, dtype=torch.float64)                  # the same matrix on every rank
        g = torch.Generator().manual_seed(3)
        node_part = torch.randint(0, world, (n,), generator=g)         # an arbitrary owner per node
        store = bench_sage.FeatureStore(feat, "owner", rank, world, node_part=node_part, range_partition=_RangeDouble)
        assert torch.equal(torch.sort(store.owned)[0], (node_part == rank).nonzero().flatten())
        assert store.local.shape[0] == int((node_part == rank).sum())  # only my rows are resident
        for it in range(3):
            gi = torch.Generator().manual_seed(50 + 10 * it + rank)
            ids = torch.randint(0, n, (400 + 37 * rank,), generator=gi)
            got = store.fetch(ids)
            assert torch.equal(got, feat[ids])                          # owned rows locally, the rest pulled
        own = float((node_part[torch.cat([torch.randint(0, n, (400 + 37 * rank,),
                                                        generator=torch.Generator().manual_seed(50 + 10 * it + rank))
                                          for it in range(3)])] == rank).double().mean())
        assert abs(store.remote_rows / store.total_rows - (1 - own)) < 1e-12
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_owner_feature_store_pulls_only_non_owned_rows_under_gloo(world):
    port = 25300 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_owner_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("features", ["replicated", "sharded", "owner"])
def test_bench_sage_two_ranks_on_one_gpu(features):
    port = 28000 + os.getpid() % 1500 + {"replicated": 0, "sharded": 7, "owner": 13}[features]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DGLA_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "benchmarks", "bench_sage.py"), "--steps", "6", "--warmup", "2",
             "--scale", "16" if features == "owner" else "64", "--batch", "256", "--features", features],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=800) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["features"] == features and r["steps"] == 6
    assert r["seeds_per_s"] > 0 and r["sampled_edges_per_s"] > 0
    assert r["final_loss"] == r["final_loss"] and r["final_loss"] < 10       # finite
    # per-rank seeds, synchronised gradients: the replicas stay identical
    assert r["param_checksum_rel_spread_across_ranks"] < 1e-6
    if features == "owner":
        # variant L, owner seeds: most input rows of a batch are the rank's own (VERDICT r3 Next #6)
        assert 0 < r["remote_input_row_fraction"] < 0.5 * (2 - 1) / 2 + 0.2, r["remote_input_row_fraction"]
        assert r["remote_input_row_fraction"] < r["remote_input_row_fraction_if_seeds_and_rows_were_spread_uniformly"]
        assert r["partition"]["volume"] > 0
