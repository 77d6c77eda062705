"""The native k-way partitioner (csrc/partition.cc; stands where METIS stands in the
reference, python/dgl/partition.py:278-397) and the relabelling that turns its answer into the
contiguous row ranges the sharded SpMM uses.  Host code: runs without a GPU."""
import numpy as np
import pytest
import torch

import oracle
from dgl_amd.parallel import (halo_fraction, partition_assignment, partition_rows, relabel_csr,
                              reshuffle, shard_csr)
from tests.graphgen import coo_to_csc


def planted(k, per, deg, p_in, seed, shuffle=True):
    """k communities of `per` nodes; every node draws `deg` in-neighbours, a fraction p_in of
    them from its own community; node ids shuffled so that ranges carry no information."""
    rng = np.random.default_rng(seed)
    n = k * per
    comm = np.repeat(np.arange(k), per)
    dst = np.repeat(np.arange(n), deg)
    inside = rng.random(n * deg) < p_in
    src = np.where(inside, comm[dst] * per + rng.integers(0, per, n * deg), rng.integers(0, n, n * deg))
    perm = rng.permutation(n) if shuffle else np.arange(n)
    src, dst = perm[src], perm[dst]
    truth = np.empty(n, dtype=np.int64)
    truth[perm] = comm
    indptr, indices, _ = coo_to_csc(src, dst, n, np.int64)
    return torch.from_numpy(indptr), torch.from_numpy(indices), truth


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_recovers_planted_communities(idtype):
    k = 8
    indptr, indices, truth = planted(k, 1500, 12, 0.9, seed=1)
    part, st = partition_assignment(indptr.to(idtype), indices.to(idtype), k, imbalance=0.05, seed=3)
    assert part.shape == (k * 1500,) and int(part.min()) == 0 and int(part.max()) == k - 1
    # planted cut: ~10 % of the edges leave their community (+ the share of random edges that
    # happen to land inside): a contiguous split of the shuffled ids would cut 87.5 %
    rows = np.repeat(np.arange(len(truth)), np.diff(indptr.numpy()))
    planted_cut = float((truth[rows] != truth[indices.numpy()]).mean())
    assert 0.05 < planted_cut < 0.12
    assert st["cut_fraction"] <= 1.25 * planted_cut, st
    assert st["max_part_weight"] <= 1.06 * st["avg_part_weight"] + 20, st
    # stats agree with a recount
    p = part.numpy()
    sym_cut = (p[rows] != p[indices.numpy()]).sum()
    assert sym_cut >= st["cut_edges"] * 0.99  # merged parallel edges count once per pair weight
    # deterministic
    part2, _ = partition_assignment(indptr.to(idtype), indices.to(idtype), k, imbalance=0.05, seed=3)
    assert torch.equal(part, part2)


def test_degenerate_inputs():
    # one part, more parts than nodes with edges, isolated nodes, self loops only
    indptr = torch.tensor([0, 1, 2, 2, 3], dtype=torch.int64)
    indices = torch.tensor([0, 0, 3], dtype=torch.int64)
    part, st = partition_assignment(indptr, indices, 1)
    assert torch.equal(part, torch.zeros(4, dtype=torch.int64)) and st["cut_edges"] == 0
    part, st = partition_assignment(indptr, indices, 3, imbalance=0.5)
    assert part.numel() == 4 and 0 <= int(part.min()) and int(part.max()) < 3
    empty, _ = partition_assignment(torch.zeros(1, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 4)
    assert empty.numel() == 0


def test_partitioned_spmm_equals_unpartitioned():
    """partition -> reshuffle -> relabel -> shard per rank -> oracle SpMM per shard with halo
    rows filled from the owners -> map back: equals the SpMM of the original graph."""
    k = 4
    indptr, indices, _ = planted(k, 300, 8, 0.85, seed=5)
    n = indptr.numel() - 1
    rng = np.random.default_rng(0)
    x = rng.random((n, 6))
    full, _, _ = oracle.spmm_csr("copy_lhs", "sum", indptr.numpy(), indices.numpy(), None, x, None)
    part, st = partition_assignment(indptr, indices, k, seed=1)
    orig_id, new_id, bounds = reshuffle(part, k)
    assert torch.equal(new_id[orig_id], torch.arange(n)) and int(bounds[-1]) == n
    ip2, ix2, e2 = relabel_csr(indptr, indices, None, orig_id, new_id)
    # e2 names the original edge position of every relabelled edge
    assert torch.equal(torch.sort(e2)[0], torch.arange(indices.numel()))
    x2 = x[orig_id.numpy()]
    out2 = np.zeros_like(x2)
    for r in range(k):
        sh = shard_csr(ip2, ix2, None, bounds, r)
        lo, hi = sh["row_range"]
        xl = np.zeros((sh["n_local"] + sh["n_halo"], x.shape[1]))
        xl[: sh["n_local"]] = x2[lo:hi]
        off = sh["n_local"]
        for p in sorted(sh["requests"]):
            rows = sh["requests"][p].numpy() + int(bounds[p])
            xl[off: off + len(rows)] = x2[rows]
            off += len(rows)
        loc, _, _ = oracle.spmm_csr("copy_lhs", "sum", sh["indptr"].numpy(), sh["indices"].numpy(),
                                    None, xl, None)
        out2[lo:hi] = loc
    back = np.empty_like(out2)
    back[orig_id.numpy()] = out2
    np.testing.assert_allclose(back, full, rtol=1e-12)
    # the partitioner's cut is what the halo exchange pays for: far below the naive split
    frac, halo_rows = halo_fraction(ip2, ix2, bounds)
    naive = partition_rows(indptr, k)
    naive_frac, naive_rows = halo_fraction(indptr, indices, naive)
    assert frac < 0.5 * naive_frac and sum(halo_rows) < sum(naive_rows)
    assert abs(frac - st["cut_fraction"]) < 0.05


def test_banded_graph_multilevel_is_no_worse_than_vertex_order_ranges():
    """A banded graph (neighbours within a window of the row id — ids assigned by locality) is cut
    best by contiguous ranges in id order.  Round 2's multilevel answer lost to them (0.56 vs 0.47 of
    the edges at C2 scale) and a fallback picked the ranges; with the recursive-bisection initial
    partition the multilevel answer itself must be at least as good (VERDICT r2 Next #7), and the
    order-aware entry point still returns the better of the two."""
    rng = np.random.default_rng(5)
    n, deg, k = 40_000, 12, 8
    dst = np.repeat(np.arange(n), deg)
    src = np.clip(dst + rng.integers(-300, 301, n * deg), 0, n - 1)
    indptr, indices, _ = coo_to_csc(src, dst, n, np.int64)
    ip, ix = torch.from_numpy(indptr), torch.from_numpy(indices)
    rows = np.repeat(np.arange(n), np.diff(indptr))
    bounds = partition_rows(ip, k)
    rng_part = torch.searchsorted(bounds[1:].contiguous(), torch.arange(n), right=True).numpy()
    range_cut = float((rng_part[rows] != rng_part[indices]).mean())
    part_ml, st_ml = partition_assignment(ip, ix, k, seed=1, order_aware=False)
    assert st_ml["method"] == "multilevel"
    p = part_ml.numpy()
    ml_cut = float((p[rows] != p[indices]).mean())
    assert ml_cut <= 1.05 * range_cut, (ml_cut, range_cut)
    counts = torch.bincount(part_ml, weights=torch.from_numpy(np.diff(indptr) + 1).double(), minlength=k)
    assert float(counts.max()) <= 1.04 * float(counts.sum()) / k + 50
    part, st = partition_assignment(ip, ix, k, seed=1)           # order-aware: the better of the two
    assert st["cut_fraction"] <= min(ml_cut, range_cut) + 1e-9 and st["cut_fraction"] < 0.04, st
    q = part.numpy()
    assert int((q[rows] != q[indices]).sum()) == st["cut_edges"]
    # shuffled ids (test_recovers_planted_communities' graph): ranges would cut 87.5 %, the multilevel answer is kept
    ip2, ix2, _ = planted(8, 1500, 12, 0.9, seed=1)
    _, st2 = partition_assignment(ip2, ix2, 8, imbalance=0.05, seed=3)
    assert st2["method"] == "multilevel"


def test_thread_count_does_not_change_the_answer():
    import subprocess
    import sys

    code = ("import sys, torch; sys.path.insert(0, %r);"
            "from tests.test_partition import planted; from dgl_amd.parallel import partition_assignment;"
            "ip, ix, _ = planted(4, 4000, 10, 0.8, seed=2);"
            "p, st = partition_assignment(ip, ix, 4, seed=5, order_aware=False);"
            "print(int((p * torch.arange(1, p.numel() + 1)).sum()), st['cut_edges'])")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for t in ("1", "3", "8"):
        r = subprocess.run([sys.executable, "-c", code % root], env=dict(os.environ, DGLA_PARTITION_THREADS=t),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] == outs[2], outs


def _recount_volume(indptr, indices, part, k):
    """sum over parts of the DISTINCT remote columns their rows read (what a row-sharded SpMM pulls)."""
    ip, ix, p = indptr.numpy(), indices.numpy(), part.numpy()
    rows = np.repeat(np.arange(len(p)), np.diff(ip))
    remote = p[rows] != p[ix]
    keys = np.unique(p[rows][remote].astype(np.int64) * len(p) + ix[remote])
    per_part = np.bincount(keys // len(p), minlength=k)
    return int(per_part.sum()), int(per_part.max())


def test_volume_objective_minimises_what_is_exchanged():
    """objtype='vol' (python/dgl/partition.py:278-312): the partition's communication volume — distinct remote
    columns per part, recounted here from the assignment — is no larger than under objtype='cut', far below
    contiguous ranges of the SHUFFLED ids, and the reported statistics are that recount."""
    k = 8
    indptr, indices, truth = planted(k * 4, 600, 14, 0.85, seed=11)       # 32 communities, shuffled ids
    n = indptr.numel() - 1
    p_cut, st_cut = partition_assignment(indptr, indices, k, imbalance=0.05, seed=2, objtype="cut")
    p_vol, st_vol = partition_assignment(indptr, indices, k, imbalance=0.05, seed=2, objtype="vol")
    v_cut, _ = _recount_volume(indptr, indices, p_cut, k)
    v_vol, h_vol = _recount_volume(indptr, indices, p_vol, k)
    assert (st_vol["volume"], st_vol["max_halo_rows"]) == (v_vol, h_vol)
    assert st_cut["volume"] == v_cut
    assert v_vol <= v_cut, (v_vol, v_cut)
    bounds = partition_rows(indptr, k)
    rng_part = torch.searchsorted(bounds[1:].contiguous(), torch.arange(n), right=True)
    v_rng, _ = _recount_volume(indptr, indices, rng_part, k)
    assert v_vol < 0.5 * v_rng, (v_vol, v_rng)
    w = torch.bincount(p_vol, weights=(indptr[1:] - indptr[:-1] + 1).double(), minlength=k)
    assert float(w.max()) <= 1.06 * float(w.sum()) / k + 20
    # refinement never makes the objective worse than where it started: ranges refined <= ranges
    from dgl_amd._lib import LIB, check_call
    import ctypes
    out = torch.empty(n, dtype=torch.int64)
    st = (ctypes.c_int64 * 8)()
    init = rng_part.to(torch.int64).contiguous()
    check_call(LIB.dgla_partition_kway_ex(64, n, indptr.data_ptr(), indices.data_ptr(), k, 0.05, 1, 0, 1, 0, None,
                                          init.data_ptr(), out.data_ptr(), ctypes.cast(st, ctypes.c_void_p)))
    assert _recount_volume(indptr, indices, out, k)[0] == st[4] <= v_rng and st[7] > 0


def test_vertex_order_graph_volume_not_above_ranges():
    """A graph in locality order with a share of uniformly random neighbours (SURVEY §8d's variant L): the
    order-aware entry returns a partition whose exchanged rows are BELOW those of contiguous ranges."""
    from tests.graphgen import synth_csr

    n, e, k = 60_000, 1_200_000, 8
    g = synth_csr(n, n, e, "L", seed=3, idtype=torch.int64)
    ip, ix = g["indptr"], g["indices"]
    bounds = partition_rows(ip, k)
    rng_part = torch.searchsorted(bounds[1:].contiguous(), torch.arange(n), right=True)
    v_rng, h_rng = _recount_volume(ip, ix, rng_part, k)
    part, st = partition_assignment(ip, ix, k, seed=1, objtype="vol")
    v, h = _recount_volume(ip, ix, part, k)
    assert (v, h) == (st["volume"], st["max_halo_rows"])
    assert v < v_rng, (v, v_rng, st)


def test_balance_ntypes_balances_every_type():
    """balance_ntypes (python/dgl/partition.py:330-352): every node type is spread evenly over the parts."""
    k = 4
    indptr, indices, truth = planted(k, 2000, 10, 0.9, seed=4)
    n = indptr.numel() - 1
    rng = np.random.default_rng(8)
    # type 1 nodes are concentrated in two of the communities: a partition by community alone cannot balance them
    ntype = torch.from_numpy(((truth < 2) & (rng.random(n) < 0.6)).astype(np.int64))
    part, st = partition_assignment(indptr, indices, k, imbalance=0.05, seed=2, balance_ntypes=ntype)
    assert st["ntype_excess"] <= 0, st
    for t in (0, 1):
        cnt = torch.bincount(part[ntype == t], minlength=k).double()
        assert float(cnt.max()) <= 1.05 * float(cnt.sum()) / k + 2, (t, cnt)
    plain, _ = partition_assignment(indptr, indices, k, imbalance=0.05, seed=2)
    cnt = torch.bincount(plain[ntype == 1], minlength=k).double()
    assert float(cnt.max()) > 1.5 * float(cnt.sum()) / k      # without the constraint type 1 sits in two parts


def test_counter_table_cap_skips_the_directed_refinement(monkeypatch):
    """The k-way refiner's dense n x k counter table is capped (DGLA_PARTITION_TABLE_MAX_BYTES): above the cap the
    multilevel result stands and the volume statistics are reported as unknown, instead of an allocation the size of the
    graph times k."""
    import torch
    from dgl_amd.parallel import partition_assignment

    g = torch.Generator().manual_seed(0)
    n, e = 4000, 40000
    row = torch.sort(torch.randint(n, (e,), generator=g))[0]
    indptr = torch.zeros(n + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0)
    indices = torch.randint(n, (e,), generator=g)
    part, stats = partition_assignment(indptr, indices, 4, seed=1, objtype="vol")
    assert stats["volume"] > 0
    monkeypatch.setenv("DGLA_PARTITION_TABLE_MAX_BYTES", "1000")
    part2, stats2 = partition_assignment(indptr, indices, 4, seed=1, objtype="vol")
    assert stats2["volume"] == -1 and part2.shape == part.shape and int(part2.max()) == 3
    assert torch.bincount(part2, minlength=4).max() <= 1.1 * n / 4 + 2
