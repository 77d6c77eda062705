"""COO -> CSR / CSC on the device against the definition the reference's
conversion satisfies (aten::COOToCSR, src/array/cuda/coo2csr.cu:28-110): rows compressed,
edges in COO order inside a row, `data` = original edge id of every position.  Integer work:
bit-exact.  The definition itself is pinned to the reference: its own COOToCSR<kDGLCPU>
(src/array/cpu/spmat_op_impl_coo.cc, built in place into oracle/_ref) equals it on every case,
and its outputs are committed as tests/golden/reference_coo2csr_outputs.npz."""
import os

import numpy as np
import pytest
import torch

from oracle import ref
from tests.coo_cases import all_cases, stable_definition as _expect

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_coo2csr_outputs.npz")


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libdglref.so not built")
def test_reference_build_equals_the_stable_definition():
    """Pins the definition the GPU path is held to: the reference's own COOToCSR<kDGLCPU> (all
    four of its algorithms: sorted / small / sparse / dense, chosen by shape) returns exactly
    the stable compression, bit for bit, for both id widths and with explicit edge ids."""
    for threads in (1, 4):
        ref.set_num_threads(threads)
        for c in all_cases():
            got = ref.coo_to_csr(c["row"], c["col"], c["eids"], c["num_rows"], c["num_cols"])
            want = _expect(c["row"], c["col"], c["eids"], c["num_rows"])
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g, w, err_msg=c["name"])
    ref.set_num_threads(os.cpu_count() or 1)


def test_golden_fixture_matches_the_definition():
    gold = np.load(GOLDEN)
    n = 0
    for c in all_cases():
        if c["name"] + "/out/indptr" not in gold.files:
            continue
        want = _expect(c["row"], c["col"], c["eids"], c["num_rows"])
        for k, w in zip(("indptr", "indices", "eids"), want):
            np.testing.assert_array_equal(gold["%s/out/%s" % (c["name"], k)], w)
        n += 1
    assert n >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("c", [c for c in all_cases() if c["row"].size <= 5000], ids=lambda c: c["name"])
def test_gpu_matches_reference_outputs(dev, c):
    from dgl_amd import _capi

    gold = np.load(GOLDEN)
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    ip, ix, ei = _capi.coo_to_csr(t(c["row"]), t(c["col"]), t(c["eids"]), c["num_rows"])
    for k, g in (("indptr", ip), ("indices", ix), ("eids", ei)):
        np.testing.assert_array_equal(g.cpu().numpy(), gold["%s/out/%s" % (c["name"], k)], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("idtype", [np.int32, np.int64])
@pytest.mark.parametrize("n,m,e", [(1, 1, 0), (5, 7, 0), (1, 1, 1), (30, 40, 300), (1000, 10, 50000),
                                   (3, 100000, 200000), (70000, 70000, 1), (1 << 17, 333, 1 << 20)])
@pytest.mark.parametrize("with_eids", [False, True])
def test_coo_to_csr_bit_exact(dev, idtype, n, m, e, with_eids):
    from dgl_amd import _capi

    rng = np.random.default_rng(e + n)
    row = rng.integers(0, n, e).astype(idtype)
    if e > 10 and n > 8:
        row[rng.integers(0, e, e // 3)] = n - 1      # a hub row and empty rows in the same graph
        row[row == 2] = 3
    col = rng.integers(0, m, e).astype(idtype)
    eids = rng.permutation(e).astype(idtype) if with_eids else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    wip, wix, wei = _expect(row, col, eids, n)
    # without a bound on the minor ids (int64: permutation form) and with it (packed 32-bit sort)
    for num_minor in (0, m):
        ip, ix, ei = _capi.coo_to_csr(t(row), t(col), t(eids), n, num_minor)
        np.testing.assert_array_equal(ip.cpu().numpy(), wip)
        np.testing.assert_array_equal(ix.cpu().numpy(), wix)
        np.testing.assert_array_equal(ei.cpu().numpy(), wei)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["sorted_by_row", "one_row_holds_everything", "three_levels_large_items",
                                   "two_buckets_only", "reverse_sorted", "rows_not_a_power_of_two_with_empty_tail"])
def test_coo_to_csr_shapes_the_bucket_sort_is_sensitive_to(dev, shape):
    """csrc/sort.hip.h: every lane of a wave in ONE bin (sorted input, a mega row), buckets that are empty or hold
    everything, three digit levels with 8 192-element items (> 2^19 edges), row counts just above a power of two."""
    from dgl_amd import _capi

    g = torch.Generator(device=dev).manual_seed(5)
    idt = torch.int32
    if shape == "sorted_by_row":
        n, e = 300_000, 1_200_000
        row = torch.sort(torch.randint(0, n, (e,), device=dev, generator=g))[0]
    elif shape == "reverse_sorted":
        n, e = 300_000, 1_200_000
        row = torch.sort(torch.randint(0, n, (e,), device=dev, generator=g), descending=True)[0]
    elif shape == "one_row_holds_everything":
        n, e = 70_000, 900_000
        row = torch.full((e,), 12_345, device=dev, dtype=torch.int64)
        row[::1000] = torch.randint(0, n, (e // 1000,), device=dev, generator=g)
    elif shape == "three_levels_large_items":
        n, e = 2_100_000, 3_000_000                    # 22 bits of key: levels of 8 / 7 / 7 bits, items of 8 192
        row = torch.randint(0, n, (e,), device=dev, generator=g)
    elif shape == "two_buckets_only":
        n, e = 1 << 20, 700_000
        row = torch.where(torch.rand(e, device=dev, generator=g) < 0.5, 3, n - 2) + torch.randint(0, 2, (e,), device=dev, generator=g)
    else:
        n, e = (1 << 16) + 3, 800_000
        row = torch.randint(0, (1 << 16) - 100, (e,), device=dev, generator=g)
    row = row.to(idt).contiguous()
    col = torch.randint(0, n, (e,), device=dev, generator=g).to(idt)
    order = torch.argsort(row, stable=True)
    counts = torch.bincount(row.long(), minlength=n)
    want_ip = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    want_ip[1:] = torch.cumsum(counts, 0)
    for num_minor in (n, 0):
        ip, ix, ei = _capi.coo_to_csr(row, col, None, n, num_minor)
        assert torch.equal(ip.long(), want_ip) and torch.equal(ix, col[order]) and torch.equal(ei.long(), order)


@pytest.mark.gpu
def test_graph_formats_are_built_natively_and_round_trip(dev):
    """The DGLGraph shim builds CSR / CSC through the native path; COO -> CSC -> COO keeps
    every edge, and the conversion runs on the current stream."""
    import dgl_amd as dgl

    g = dgl.rand_graph(500, 20000, device=dev, seed=3)
    s, d = g.edges()
    rel = g._graph.relations[0]
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        ip, ix, ei = rel.csc()
    side.synchronize()
    assert int(ip[-1]) == 20000
    rows = torch.repeat_interleave(torch.arange(500, device=dev), (ip[1:] - ip[:-1]).long())
    assert torch.equal(rows, d.long()[ei.long()]) and torch.equal(ix.long(), s.long()[ei.long()])
    # stable: edge ids ascend inside every row
    same_row = rows[1:] == rows[:-1]
    assert bool((ei[1:][same_row] > ei[:-1][same_row]).all())
