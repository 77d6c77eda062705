"""BASELINE configs[4] across ranks: the stacked multi-relation
g-SpMM sharded by destination rows with one halo per SOURCE node type.

CPU: simulated ranks in one process and world-2 gloo, with the torch stand-in as kernel backend
(host logic: shard construction, halo union, exchange, stacking).  GPU: simulated ranks on the real
stacked kernel, bf16, against the single-GPU stacked launch."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

NUM_NODES = [500, 230, 90]
META = [(0, 0), (1, 0), (2, 0), (1, 2), (0, 0), (0, 1)]   # relations 0,1,2,4 reduce into type 0
EDGES = [4000, 2500, 700, 900, 3000, 1200]


def _hetero(seed, dev="cpu", idtype=torch.int64):
    from tests.graphgen import coo_to_csc

    rng = np.random.default_rng(seed)
    rels = []
    for (s, d), ne in zip(META, EDGES):
        src = rng.integers(0, NUM_NODES[s], ne)
        dst = np.minimum((rng.random(ne) ** 2 * NUM_NODES[d]).astype(np.int64), NUM_NODES[d] - 1)
        ip, ix, _ = coo_to_csc(src, dst, NUM_NODES[d], np.int64)
        rels.append((torch.from_numpy(ip).to(idtype).to(dev), torch.from_numpy(ix).to(idtype).to(dev)))
    return rels


def _dense(rels, xs):
    outs = [None] * len(NUM_NODES)
    for (s, d), (ip, ix) in zip(META, rels):
        ip = ip.long()
        row_of = torch.repeat_interleave(torch.arange(NUM_NODES[d], device=ip.device), ip[1:] - ip[:-1])
        if outs[d] is None:
            outs[d] = torch.zeros((NUM_NODES[d],) + tuple(xs[s].shape[1:]), dtype=torch.float64, device=ip.device)
        outs[d].index_add_(0, row_of, xs[s].double()[ix.long()])
    return outs


def _parts(k, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, k, (n,), generator=g) for n in NUM_NODES]


@pytest.mark.parametrize("k,chunks", [(2, 1), (3, 1), (4, 2)])
def test_sharded_hetero_simulated_ranks_equal_unpartitioned(k, chunks):
    from dgl_amd.parallel_hetero import (ShardedHeteroSpMM, SimulatedHeteroExchange,
                                         shard_hetero_from_partition)
    from tests.cpu_backends import torch_stacked_backend

    rels = _hetero(1)
    torch.manual_seed(0)
    xs = [torch.rand(n, 6, dtype=torch.float64) for n in NUM_NODES]
    want = _dense(rels, xs)
    parts = _parts(k, 5)
    shards = [shard_hetero_from_partition(NUM_NODES, META, rels, parts, k, r) for r in range(k)]
    assert sum(sh["nnz"] for sh in shards) == sum(EDGES)
    ex = SimulatedHeteroExchange(shards, chunks)
    x_loc = [[xs[t][sh["rows"][t]].contiguous() for t in range(3)] for sh in shards]
    for r in range(k):
        ex.bind(r, x_loc[r])
    for r, sh in enumerate(shards):
        # a remote row wanted by several relations is in the halo ONCE
        for s in range(3):
            allreq = torch.cat([v for v in sh["requests"][s].values()]) if sh["requests"][s] else torch.empty(0)
            assert allreq.numel() == sh["n_halo"][s]
        op = ShardedHeteroSpMM(sh, (6,), torch.float64, "cpu", backend=torch_stacked_backend(), exchange=ex)
        out = [torch.full((sh["n_local"][t], 6), 7.0, dtype=torch.float64) for t in range(3)]
        op.step(x_loc[r], out)
        for d in (0, 1, 2):
            np.testing.assert_allclose(out[d].numpy(), want[d][sh["rows"][d]].numpy(), rtol=1e-12)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import cpu_backends
    cpu_backends.install()   # torch stand-ins for the row kernels: the package has no CPU path
    try:
        from dgl_amd.parallel_hetero import ShardedHeteroSpMM, shard_hetero_from_partition
        from tests.cpu_backends import torch_stacked_backend

        rels = _hetero(2)
        torch.manual_seed(1)
        xs = [torch.rand(n, 5, dtype=torch.float64) for n in NUM_NODES]
        want = _dense(rels, xs)
        parts = _parts(world, 9)
        sh = shard_hetero_from_partition(NUM_NODES, META, rels, parts, world, rank)
        op = ShardedHeteroSpMM(sh, (5,), torch.float64, "cpu", backend=torch_stacked_backend(), chunks=2)
        x_loc = [xs[t][sh["rows"][t]].contiguous() for t in range(3)]
        out = [torch.empty(sh["n_local"][t], 5, dtype=torch.float64) for t in range(3)]
        op.step(x_loc, out)
        for d in range(3):
            np.testing.assert_allclose(out[d].numpy(), want[d][sh["rows"][d]].numpy(), rtol=1e-12)
        op.step([2 * t for t in x_loc], out)       # the owners changed their rows: no stale halo
        for d in range(3):
            np.testing.assert_allclose(out[d].numpy(), 2 * want[d][sh["rows"][d]].numpy(), rtol=1e-12)
        ret[rank] = (sh["cut_edges"], sh["nnz"])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hetero_gloo(world):
    port = 24100 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    got = dict(ret)
    assert sorted(got) == list(range(world))
    assert sum(v[1] for v in got.values()) == sum(EDGES) and sum(v[0] for v in got.values()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("k", [2, 4])
def test_sharded_hetero_gpu_equals_single_gpu_stacked_launch(dev, dtype, k):
    """Simulated ranks on the real stacked kernel ≡ ONE stacked launch over the whole graph
    (per destination type), 16-bit tolerance for bf16."""
    from dgl_amd import _capi
    from dgl_amd.graph_index import stack_csc
    from dgl_amd.parallel_hetero import ShardedHeteroSpMM, SimulatedHeteroExchange, shard_hetero_from_partition

    rels = _hetero(3, dev, torch.int32)
    torch.manual_seed(2)
    f = 64
    xs = [(torch.rand(n, f, device=dev) + 1).to(dtype) for n in NUM_NODES]
    # single-GPU stacked launch per destination type
    full = {}
    for d in range(3):
        rl = [r for r, (_, dd) in enumerate(META) if dd == d]
        if not rl:
            continue
        ip, ix, ei, rel = stack_csc([(rels[r][0], rels[r][1], None) for r in rl], NUM_NODES[d], torch.int32)
        csr = _capi.make_csr(ip, ix, ei, max(NUM_NODES))
        o = torch.empty(NUM_NODES[d], f, device=dev, dtype=dtype)
        xl = [xs[META[r][0]] for r in rl]
        ws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", csr, xl[0], None, o), dtype=torch.uint8, device=dev)
        _capi.spmm_csr_stacked("copy_lhs", csr, rel, xl, None, o, ws)
        full[d] = o
    exact = _dense(rels, xs)
    parts = [p.to(dev) for p in _parts(k, 7)]
    shards = [shard_hetero_from_partition(NUM_NODES, META, rels, parts, k, r) for r in range(k)]
    ex = SimulatedHeteroExchange(shards)
    x_loc = [[xs[t][sh["rows"][t]].contiguous() for t in range(3)] for sh in shards]
    for r in range(k):
        ex.bind(r, x_loc[r])
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
    for r, sh in enumerate(shards):
        op = ShardedHeteroSpMM(sh, (f,), dtype, dev, exchange=ex)
        out = [torch.empty(sh["n_local"][t], f, device=dev, dtype=dtype) for t in range(3)]
        op.step(x_loc[r], out)
        op.step(x_loc[r], out)   # cached plan / tables
        for d in full:
            got = out[d].double()
            torch.testing.assert_close(got, full[d][sh["rows"][d]].double(), rtol=tol, atol=0)
            torch.testing.assert_close(got, exact[d][sh["rows"][d]], rtol=tol, atol=0)
