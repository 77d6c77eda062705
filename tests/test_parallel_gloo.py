"""Multi-process (gloo, world_size = 2, CPU) tests of the halo exchange and the row sharding:
the sharded SpMM (oracle on each shard + halo pull) must equal the single-partition result."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import cpu_backends
    cpu_backends.install()   # torch stand-ins for the row kernels: the package has no CPU path
    try:
        import oracle
        from dgl_amd.parallel import HaloExchange, partition_rows, shard_csr
        from tests.graphgen import synth_csr

        n, e, f = 3000, 40000, 12
        g = synth_csr(n, n, e, "L", seed=5, idtype=torch.int64)
        torch.manual_seed(1)
        x_full = torch.rand(n, f, dtype=torch.float64)
        full, _, _ = oracle.spmm_csr("copy_lhs", "sum", g["indptr"].numpy(), g["indices"].numpy(),
                                     None, x_full.numpy(), None)
        bounds = partition_rows(g["indptr"], world)
        sh = shard_csr(g["indptr"], g["indices"], None, bounds, rank)
        lo, hi = sh["row_range"]
        hx = HaloExchange(sh["n_local"], sh["n_halo"], f, "cpu", requests=sh["requests"])
        x = torch.zeros(sh["n_local"] + sh["n_halo"], f, dtype=torch.float64)
        x[: sh["n_local"]] = x_full[lo:hi]
        hx.pull(x)
        # every halo row must equal the owner's row
        off = sh["n_local"]
        for p in sorted(sh["requests"]):
            rows = sh["requests"][p] + int(bounds[p])
            assert torch.equal(x[off: off + rows.numel()], x_full[rows])
            off += rows.numel()
        loc, _, _ = oracle.spmm_csr("copy_lhs", "sum", sh["indptr"].numpy(), sh["indices"].numpy(),
                                    None, x.numpy(), None)
        np.testing.assert_allclose(loc, full[lo:hi], rtol=1e-12)
        # second pull after the owners changed their rows sees the new values
        x[: sh["n_local"]] *= 2
        hx.pull(x)
        assert torch.equal(x[sh["n_local"]:][:1] if sh["n_halo"] else x[:0],
                           (2 * x_full[sh["requests"][min(sh["requests"])][:1] + int(bounds[min(sh["requests"])])])
                           if sh["n_halo"] else x[:0])
        # synthetic (bench-style) requests also round-trip
        hx2 = HaloExchange(100, 40 * (world - 1), 3, "cpu", seed=3)
        y = torch.arange((100 + 40 * (world - 1)) * 3, dtype=torch.float32).reshape(-1, 3) + 1000 * rank
        hx2.pull(y)
        assert float(y[100:].min()) >= 1000 * (1 - rank)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_halo_exchange_world2():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_partition_rows_balances_edges():
    from dgl_amd.parallel import partition_rows
    from tests.graphgen import synth_csr

    g = synth_csr(5000, 5000, 120000, "U", seed=9, idtype=torch.int64)
    b = partition_rows(g["indptr"], 8)
    assert b[0] == 0 and b[-1] == 5000 and torch.all(b[1:] >= b[:-1])
    per = (g["indptr"][b[1:]] - g["indptr"][b[:-1]]).double()
    assert per.max() / per.mean() < 1.1
