"""Multi-process (gloo, CPU) tests of the partitioned g-SpMM schedule bench.py --gpus N runs:
partition -> shard (local / halo column blocks) -> overlapped halo pull -> two SpMM launches,
plus the gradient push and the general sparse all-to-all.  The kernel backend is replaced by
the CPU oracle here (tests may use it; the product backend is GPU-only and is covered by the
simulated-rank tests in test_gpu_sharded.py); everything else — index math, splits, collectives,
ordering — is the code the GPU ranks run."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def oracle_backend():
    import oracle

    def run(tag, csr_pair, n_cols, x, out, accumulate):
        indptr, indices = csr_pair
        res, _, _ = oracle.spmm_csr("copy_lhs", "sum", indptr.numpy(), indices.numpy(), None,
                                    x.numpy(), None)
        res = torch.from_numpy(np.ascontiguousarray(res)).reshape(out.shape)
        if accumulate:
            out += res
        else:
            out.copy_(res)
    return run


class RemainderDouble:
    """numpy stand-in for NDArrayPartition(mode='remainder') on CPU tensors (the product class is
    GPU-only, like the reference's); duck-typed for sparse_all_to_all_push / _pull."""

    def __init__(self, k):
        self.k = k

    def generate_permutation(self, idx):
        part = idx % self.k
        perm = torch.argsort(part, stable=True)
        return perm, torch.bincount(part, minlength=self.k).to(torch.int64)

    def map_to_local(self, idx):
        return idx // self.k


def _worker(rank, world, port, ret, chunks=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import cpu_backends
    cpu_backends.install()   # torch stand-ins for the row kernels: the package has no CPU path
    try:
        import oracle
        from dgl_amd.parallel import (HaloExchange, ShardedSpMM, partition_assignment,
                                      shard_from_partition, sparse_all_to_all_pull,
                                      sparse_all_to_all_push)
        from tests.graphgen import synth_csr

        n, e, f = 4000, 60000, 12
        g = synth_csr(n, n, e, "L", seed=11, idtype=torch.int64)
        torch.manual_seed(3)
        x_full = torch.rand(n, f, dtype=torch.float64)
        full, _, _ = oracle.spmm_csr("copy_lhs", "sum", g["indptr"].numpy(), g["indices"].numpy(),
                                     None, x_full.numpy(), None)
        full = torch.from_numpy(full)
        # the same deterministic partition on every rank (bench.py: rank 0 computes, broadcasts)
        part, stats = partition_assignment(g["indptr"], g["indices"], world, seed=1)
        sh = shard_from_partition(g["indptr"], g["indices"], part, world, rank)
        assert sh["nnz"] == int(sh["local"][0][-1]) + int(sh["halo"][0][-1])
        rows = sh["rows"]
        op = ShardedSpMM(sh, (f,), torch.float64, "cpu", spmm=oracle_backend(), chunks=chunks)
        assert op.chunks == chunks and len(op.halo_blocks) == (chunks if sh["n_halo"] else 1)
        if chunks > 1 and sh["n_halo"]:
            # the chunk blocks partition the halo-column block: same edges, columns inside their chunk
            assert sum(int(b[0][-1]) for b in op.halo_blocks) == int(sh["halo"][0][-1])
            for c, (bp, bi) in enumerate(op.halo_blocks):
                if bi.numel():
                    assert int(bi.min()) >= op.exchange.chunk_bounds[c] and int(bi.max()) < op.exchange.chunk_bounds[c + 1]
        x_loc = x_full[rows].contiguous()
        out = torch.empty(sh["n_local"], f, dtype=torch.float64)
        op.step(x_loc, out)
        np.testing.assert_allclose(out.numpy(), full[rows].numpy(), rtol=1e-12)
        # a second step after the owners changed their rows sees the new values (no stale halo)
        op.step(2 * x_loc, out)
        np.testing.assert_allclose(out.numpy(), 2 * full[rows].numpy(), rtol=1e-12)

        # gradient push = transpose of the pull: <pull(x), h> == <x, push(h)>
        hx = op.exchange
        assert isinstance(hx, HaloExchange)
        h = torch.rand(sh["n_halo"], f, dtype=torch.float64)
        pulled = torch.empty(sh["n_halo"], f, dtype=torch.float64)
        w = hx.pull_async(x_loc, pulled)
        w.wait()
        back = hx.push(h, torch.zeros(sh["n_local"], f, dtype=torch.float64))
        lhs = torch.tensor([float((pulled * h).sum())], dtype=torch.float64)
        rhs = torch.tensor([float((x_loc * back).sum())], dtype=torch.float64)
        dist.all_reduce(lhs)
        dist.all_reduce(rhs)
        assert abs(float(lhs) - float(rhs)) <= 1e-9 * abs(float(lhs))
        # push of ones counts how many ranks pulled each row
        cnt = hx.push(torch.ones(sh["n_halo"], 1, dtype=torch.float64),
                      torch.zeros(sh["n_local"], 1, dtype=torch.float64))
        want = torch.zeros(n, 1, dtype=torch.float64)
        for r in range(world):
            o = shard_from_partition(g["indptr"], g["indices"], part, world, r)
            for p, req in o["requests"].items():
                if p == rank:
                    want[req + int(sh["bounds"][rank])] += 1
        assert torch.equal(cnt, want[int(sh["bounds"][rank]): int(sh["bounds"][rank + 1])])

        # general sparse all-to-all (nccl.py docstring example, remainder partition)
        part_obj = RemainderDouble(world)
        table = torch.arange(n, dtype=torch.float64).reshape(-1, 1) * 10
        mine = table[rank::world].contiguous()          # rows this rank owns: i % world == rank
        gen = torch.Generator().manual_seed(100 + rank)
        req = torch.randint(0, n, (777,), generator=gen)
        got = sparse_all_to_all_pull(req, mine, part_obj)
        assert torch.equal(got, table[req])
        idx = torch.randint(0, n, (500,), generator=gen)
        val = torch.rand(500, 3, dtype=torch.float64, generator=gen)
        r_idx, r_val = sparse_all_to_all_push(idx, val, part_obj)
        assert bool((r_idx % world == rank).all())
        tot = torch.tensor([float(val.sum()), float(idx.sum())], dtype=torch.float64)
        rec = torch.tensor([float(r_val.sum()), float(r_idx.sum())], dtype=torch.float64)
        dist.all_reduce(tot)
        dist.all_reduce(rec)
        assert torch.allclose(tot, rec, rtol=1e-12)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,chunks", [(2, 1), (4, 1), (4, 3), (2, 4)])
def test_sharded_spmm_and_exchange(world, chunks):
    """chunks > 1: the pipelined schedule (chunk-major halo block, one all-to-all per chunk, the
    halo-column block cut per chunk) must give what the unpartitioned graph gives, pull and push
    stay transposes of each other (VERDICT r2 Next #3)."""
    port = 23000 + (os.getpid() % 2000) + world * 8 + chunks
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret, chunks), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


def test_chunk_layout_is_a_permutation_and_owner_derivable():
    from dgl_amd.parallel import chunk_layout

    counts = [0, 7, 3, 0, 10, 1]
    for C in (1, 2, 3, 5, 16):
        pieces, bounds, o2n = chunk_layout(counts, C)
        assert sorted(o2n.tolist()) == list(range(sum(counts)))
        assert bounds[0] == 0 and bounds[-1] == sum(counts) and len(pieces) == C
        assert [sum(pieces[c][p] for c in range(C)) for p in range(len(counts))] == counts
        # inside a chunk: peers ascending, each peer's piece contiguous and in request order
        off = 0
        for p, cnt in enumerate(counts):
            pos = o2n[off: off + cnt]
            assert bool((pos[1:] > pos[:-1]).all()) if cnt > 1 else True
            off += cnt
    assert torch.equal(chunk_layout(counts, 1)[2], torch.arange(sum(counts)))


def test_shard_from_partition_matches_relabel_then_shard():
    """The per-rank construction equals reshuffle -> relabel_csr -> shard_csr on the whole graph."""
    from dgl_amd.parallel import (partition_assignment, relabel_csr, reshuffle, shard_csr,
                                  shard_from_partition)
    from tests.graphgen import synth_csr

    n, e, k = 3000, 45000, 4
    g = synth_csr(n, n, e, "U", seed=4, idtype=torch.int32)
    part, _ = partition_assignment(g["indptr"], g["indices"], k, seed=2)
    orig_id, new_id, bounds = reshuffle(part, k)
    ip, ix, _ = relabel_csr(g["indptr"], g["indices"], None, orig_id, new_id)
    total_cut = 0
    for r in range(k):
        ref = shard_csr(ip, ix, None, bounds, r)
        sh = shard_from_partition(g["indptr"], g["indices"], part, k, r)
        assert torch.equal(sh["rows"], orig_id[int(bounds[r]): int(bounds[r + 1])])
        assert sh["n_local"] == ref["n_local"] and sh["n_halo"] == ref["n_halo"]
        assert sorted(sh["requests"]) == sorted(ref["requests"])
        for p in ref["requests"]:
            assert torch.equal(sh["requests"][p].long(), ref["requests"][p].long())
        # merging the two column blocks row by row gives the reference shard's rows
        lp, li = (t.long() for t in sh["local"])
        hp, hi = (t.long() for t in sh["halo"])
        rp, ri = ref["indptr"].long(), ref["indices"].long()
        assert torch.equal(lp + hp, rp)
        for row in (0, 1, sh["n_local"] // 2, sh["n_local"] - 1):
            merged = torch.cat([li[lp[row]: lp[row + 1]], hi[hp[row]: hp[row + 1]] + sh["n_local"]])
            assert torch.equal(torch.sort(merged)[0], torch.sort(ri[rp[row]: rp[row + 1]])[0])
        total_cut += sh["cut_edges"]
    assert total_cut > 0
