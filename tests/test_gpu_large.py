"""Maximum-size edge case: a CSR with MORE THAN 2^31 edges (int64 ids), the regime of
ogbn-papers100M-scale graphs that 288 GB of HBM makes a single-GPU problem.  Every 32-bit
offset in the merge plan, the LDS staging, the carry workspace or the fix-up would show up
here.  Checked exactly: small-integer features make every fp32 partial sum exact, so the result
is order-independent and compared bit for bit with an independent torch evaluation."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_spmm_with_more_than_2_31_edges(dev):
    from dgl_amd import _capi

    free, _ = torch.cuda.mem_get_info(dev)
    if free < 80 << 30:
        pytest.skip("needs ~60 GB of free HBM")
    n = 1 << 20
    deg = 2049                       # n * deg = 2^31 + 2^20 edges
    e = n * deg
    assert e > 2 ** 31
    gen = torch.Generator(device=dev).manual_seed(1)
    indices = torch.randint(0, n, (e,), device=dev, dtype=torch.int64, generator=gen)
    indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * deg
    # make the rows uneven around the 2^31 boundary: move the row ends (keeps indptr monotone)
    shift = torch.randint(-deg // 2, deg // 2, (n - 1,), device=dev, generator=gen)
    indptr[1:-1] += shift
    x = torch.randint(0, 4, (n, 1), device=dev, generator=gen).float()   # sums < 2^24: exact in fp32
    csr = _capi.make_csr(indptr, indices, None, n)
    for red in ("sum", "max"):
        out = torch.empty(n, 1, device=dev)
        arg = torch.empty(n, 1, dtype=torch.int64, device=dev) if red == "max" else None
        ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", red, csr, x.dtype, x, None, out),
                         dtype=torch.uint8, device=dev)
        _capi.spmm_csr("copy_lhs", red, csr, x, None, out, arg, None, ws)
        torch.cuda.synchronize()
        del ws
        # independent evaluation in slabs of rows (keeps the temporary small)
        step = 1 << 16
        for r0 in range(0, n, step):
            r1 = min(n, r0 + step)
            lo, hi = int(indptr[r0]), int(indptr[r1])
            vals = x[indices[lo:hi], 0]
            seg = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), (indptr[r0 + 1:r1 + 1] - indptr[r0:r1]))
            if red == "sum":
                want = torch.zeros(r1 - r0, device=dev).index_add_(0, seg, vals)
                assert torch.equal(out[r0:r1, 0], want), (red, r0)
            else:
                want = torch.full((r1 - r0,), -float("inf"), device=dev).scatter_reduce(0, seg, vals, "amax")
                assert torch.equal(out[r0:r1, 0], want), (red, r0)
                # arg: the FIRST position attaining the maximum, as a global column id
                pos = torch.arange(lo, hi, device=dev)
                is_max = vals == want[seg]
                first = torch.full((r1 - r0,), hi, device=dev, dtype=torch.int64).scatter_reduce(
                    0, seg[is_max], pos[is_max], "amin")
                assert torch.equal(arg[r0:r1, 0], indices[first]), (red, r0)
            del vals, seg, want


def _need(dev, gib):
    free, _ = torch.cuda.mem_get_info(dev)
    if free < gib << 30:
        pytest.skip("needs ~%d GB of free HBM" % gib)


def test_sddmm_and_edge_softmax_with_more_than_2_31_edges(dev):
    from dgl_amd import _capi

    _need(dev, 120)
    n = 1 << 16
    e = (1 << 31) + (1 << 12)
    gen = torch.Generator(device=dev).manual_seed(2)
    row = torch.randint(0, n, (e,), device=dev, dtype=torch.int64, generator=gen)
    col = torch.randint(0, n, (e,), device=dev, dtype=torch.int64, generator=gen)
    x = torch.randint(0, 100, (n, 1), device=dev, generator=gen).float()
    y = torch.randint(0, 100, (n, 1), device=dev, generator=gen).float()
    coo = _capi.make_coo(row, col, None, n, n)
    out = torch.empty(e, 1, device=dev)
    _capi.sddmm_coo("add", coo, x, y, out, _capi.TARGETS["u"], _capi.TARGETS["v"])
    step = 1 << 28
    for s in range(0, e, step):
        t = min(e, s + step)
        assert torch.equal(out[s:t, 0], x[row[s:t], 0] + y[col[s:t], 0]), s
    del out, x, y
    # edge softmax over an in-edge CSR with > 2^31 positions and an explicit edge-id map that
    # reverses the order (ids up to 2^31 + 4095): rows sum to one, out follows the ids
    deg = e // n
    indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * deg
    indptr[-1] = e
    eids = torch.arange(e - 1, -1, -1, device=dev, dtype=torch.int64)
    csr = _capi.make_csr(indptr, col, eids, n)
    score = torch.randint(-3, 4, (e, 1), device=dev, generator=gen).float()
    sm = torch.empty_like(score)
    _capi.edge_softmax_forward(csr, score, sm, None)
    torch.cuda.synchronize()
    for r in (0, 1, n // 2, n - 2, n - 1):
        lo, hi = int(indptr[r]), int(indptr[r + 1])
        ids = eids[lo:hi]
        want = torch.softmax(score[ids, 0].double(), 0)
        assert torch.allclose(sm[ids, 0].double(), want, rtol=1e-5, atol=1e-9), r
    total = float(sm.double().sum())
    assert abs(total - n) < 1e-3 * n


def test_segment_reduce_and_coo_to_csr_with_more_than_2_31_rows(dev):
    from dgl_amd import _capi

    _need(dev, 160)
    n_seg = 1 << 20
    rows = (1 << 31) + (1 << 10)
    gen = torch.Generator(device=dev).manual_seed(3)
    seg_of = torch.randint(0, n_seg, (rows,), device=dev, dtype=torch.int64, generator=gen)
    # COO -> CSR of (major = seg_of, minor = position): indptr = segment offsets, eids = the
    # stable order of the rows -> doubles as the segment-reduce input
    pos = torch.arange(rows, device=dev, dtype=torch.int64)
    indptr, indices, eids = _capi.coo_to_csr(seg_of, pos, None, n_seg)
    counts = torch.bincount(seg_of, minlength=n_seg)
    want_ptr = torch.zeros(n_seg + 1, dtype=torch.int64, device=dev)
    want_ptr[1:] = torch.cumsum(counts, 0)
    assert torch.equal(indptr, want_ptr)
    assert torch.equal(indices, eids)                      # minor = position = edge id
    sorted_seg = seg_of[eids]
    assert bool((sorted_seg[1:] >= sorted_seg[:-1]).all())
    same = sorted_seg[1:] == sorted_seg[:-1]
    assert not bool((same & (eids[1:] <= eids[:-1])).any())  # stable inside a row
    del sorted_seg, same, pos, indices
    feat = torch.randint(0, 8, (rows, 1), device=dev, generator=gen).float()   # exact sums
    out = torch.empty(n_seg, 1, device=dev)
    # segments are contiguous runs of `feat`: reduce feat in ITS order with offsets = indptr
    _capi.segment_reduce("sum", feat, indptr, out)
    want = torch.zeros(n_seg, device=dev)
    step = 1 << 28
    ip = indptr
    bounds = torch.searchsorted(ip, torch.arange(0, rows + step, step, device=dev).clamp(max=rows))
    for k in range(len(bounds) - 1):
        a, b = int(bounds[k]), int(bounds[k + 1])
        if b <= a:
            continue
        lo, hi = int(ip[a]), int(ip[b])
        seg = torch.repeat_interleave(torch.arange(a, b, device=dev), ip[a + 1:b + 1] - ip[a:b])
        want.index_add_(0, seg, feat[lo:hi, 0])
    assert torch.equal(out[:, 0], want)
    arg = torch.empty(n_seg, 1, dtype=torch.int64, device=dev)
    _capi.segment_reduce("max", feat, indptr, out, arg)
    probe = torch.tensor([0, 1, n_seg // 3, n_seg - 1], device=dev)
    for r in probe.tolist():
        lo, hi = int(ip[r]), int(ip[r + 1])
        if hi > lo:
            m = feat[lo:hi, 0].max()
            assert float(out[r, 0]) == float(m)
            assert int(arg[r, 0]) == lo + int(torch.nonzero(feat[lo:hi, 0] == m)[0])


def test_segment_mm_with_more_than_2_31_elements(dev):
    from dgl_amd import _capi

    _need(dev, 40)
    rows, k, n, r = 10_000_000, 256, 256, 8          # A and C: 2.56e9 elements each
    seglen = torch.full((r,), rows // r, dtype=torch.int64)
    torch.manual_seed(0)
    a = (torch.rand(rows, k, device=dev) - 0.5).to(torch.bfloat16)
    b = (torch.rand(r, k, n, device=dev) - 0.5).to(torch.bfloat16)
    c = torch.empty(rows, n, device=dev, dtype=torch.bfloat16)
    _capi.segment_mm(a, b, c, seglen)
    off = 0
    for i in range(r):
        m = int(seglen[i])
        want = a[off:off + m].float() @ b[i].float()
        err = (c[off:off + m].float() - want).abs().max()
        assert float(err) < 0.08, (i, float(err))     # |c| ~ 1.3: bf16 rounding of the result
        off += m
    db = torch.empty(r, k, n, device=dev, dtype=torch.bfloat16)
    _capi.segment_mm_backward_b(a, c, db, seglen)
    i, m = r - 1, int(seglen[-1])
    want = a[rows - m:].float().T @ c[rows - m:].float()
    assert torch.allclose(db[i].float(), want, rtol=2e-2, atol=2.0)
