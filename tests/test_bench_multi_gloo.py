"""bench.py's N > 1 set-up (same graph on every rank, rank-0 partition + broadcast, per-rank
shard, ShardedSpMM step, per-rank bookkeeping) run under gloo on the CPU with the oracle as the
kernel backend: what the driver launches on 2/4/8 GPUs, minus the kernels and the timing."""
import os
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, partitioner, ret, variant="L", scale=512):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import cpu_backends
    cpu_backends.install()   # torch stand-ins for the row kernels: the package has no CPU path
    try:
        import bench
        import oracle
        from tests.test_sharded_gloo import oracle_backend

        args = types.SimpleNamespace(variant=variant, partitioner=partitioner)
        n, e, f = bench.C2_NODES // scale, bench.C2_EDGES // scale, 16
        step, ctx = bench.multi_gpu(args, torch.device("cpu"), n, e, f, rank, world, dist,
                                    spmm=oracle_backend())
        step()
        step()
        g, x = ctx["g"], ctx["x"]
        full, _, _ = oracle.spmm_csr("copy_lhs", "sum", g["indptr"].numpy(), g["indices"].numpy(), None,
                                     x.numpy(), None)
        # both sides are the oracle's SEQUENTIAL fp32 sums, in different orders (own columns first,
        # then halo columns): on the 17k-edge hub rows of this degree distribution two such orders
        # differ by up to ~2e-5; the GPU kernels (blocked partial sums) are held to 1e-5 in
        # test_gpu_sharded.py
        np.testing.assert_allclose(ctx["out"].numpy(), full[ctx["shard"]["rows"].numpy()], rtol=1e-4)
        ctx["step_replicated"]()      # rows sharded, features replicated: no exchange, same rows
        np.testing.assert_allclose(ctx["out_replicated"].numpy(), full[ctx["shard"]["rows"].numpy()], rtol=1e-4)
        # the N > 1 line explains itself: exchange alone, exposed wait, overlap, halo MB — per rank (VERDICT r4 Next #8)
        prof = bench.exchange_profile(ctx["op"], ctx["x_loc"], ctx["out"], dist, torch.device("cpu"), reps=2)
        for key in ("step_ms", "local_ms", "wait_ms", "halo_ms", "exchange_alone_ms", "overlap_frac", "profiled_steps"):
            assert key in prof, (key, prof)
        assert prof["profiled_steps"] == 2 and prof["exchange_alone_ms"] > 0 and 0.0 <= prof["overlap_frac"] <= 1.0
        np.testing.assert_allclose(ctx["out"].numpy(), full[ctx["shard"]["rows"].numpy()], rtol=1e-4)
        infos = ctx["infos"]
        assert [i["rank"] for i in infos] == list(range(world))
        assert sum(i["edges"] for i in infos) == e and sum(i["rows"] for i in infos) == n
        assert ctx["alg_bytes"] == bench.algorithmic_bytes(ctx["rows"], ctx["edges"], f)
        ret[rank] = sum(i["cut_edges"] for i in infos) / e
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,partitioner,variant,scale", [(2, "kway", "L", 512), (4, "range", "L", 512),
                                                             (8, "kway", "C", 64), (8, "kway", "U", 64)])
def test_bench_multi_gpu_setup_under_gloo(world, partitioner, variant, scale):
    """world 8 at 1/64 of C2 (VERDICT r5 Next #6d): the set-up the driver's 8-GPU run executes — graph on every rank,
    rank-0 k-way partition + broadcast, shards, the exchange, per-rank bookkeeping — for the uniform graph and for the
    planted-community graph the partitioner helps."""
    port = 26000 + (os.getpid() % 2000) + world + {"L": 0, "C": 20, "U": 40}[variant]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, partitioner, ret, variant, scale), nprocs=world, join=True)
    cuts = dict(ret)
    assert sorted(cuts) == list(range(world))
    assert len(set(cuts.values())) == 1 and 0 < cuts[0] < 1
    if variant == "C":
        assert cuts[0] < 0.3          # 10 % of the edges leave their community by construction; ranges would cut ~7/8
