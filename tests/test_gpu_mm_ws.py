"""The weights-stationary segment_mm kernels (csrc/segment_mm.hip: segment_mm_ws_kernel), FORCED on.

By its own rule the kernel only takes calls with many rows per relation (``ws_eligible``), which the
small cases of the other mm tests never reach; ``DGLA_MM_WS=1`` forces it for every shape it supports
(K <= 256, vector-aligned).  Reference: the per-segment product in float64 (what
python/dgl/ops/segment.py:segment_mm -> src/array/cuda/gather_mm.cu computes per relation), at the
tolerances of tests/python/common/ops/test_ops.py:test_segment_mm (fp32 1e-4 rel. there; the split-bf16
path is held to 2e-5 of the output scale here).
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(256, 256), (128, 256), (256, 128), (64, 64), (32, 264), (16, 8), (200, 72), (8, 520), (96, 40), (248, 136)]
SEGLENS = [[5000, 1, 0, 33, 4097, 31, 32, 700], [100000], [3, 0, 1, 7, 2, 0, 0, 5, 1, 1, 9, 4], [64] * 40]
TOL = {torch.bfloat16: 2e-2, torch.float16: 4e-3, torch.float32: 2e-5}


@pytest.fixture(autouse=True)
def _force_ws():
    old = os.environ.get("DGLA_MM_WS")
    os.environ["DGLA_MM_WS"] = "1"
    yield
    if old is None:
        os.environ.pop("DGLA_MM_WS", None)
    else:
        os.environ["DGLA_MM_WS"] = old


def _want(a, b, seglen, b_trans):
    out = torch.zeros(a.shape[0], b.shape[1] if b_trans else b.shape[2], dtype=torch.float64, device=a.device)
    off = 0
    for r, m in enumerate(seglen.tolist()):
        w = b[r].double().T if b_trans else b[r].double()
        out[off:off + m] = a[off:off + m].double() @ w
        off += m
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("kn", SHAPES, ids=lambda s: "k%dn%d" % s)
@pytest.mark.parametrize("which", range(len(SEGLENS)))
def test_forced_ws_matches_float64(dtype, kn, which):
    from dgl_amd import _capi
    dev = torch.device("cuda:0")
    k, n = kn
    sl = SEGLENS[which]
    seglen = torch.tensor(sl, dtype=torch.int64)
    m, r = int(seglen.sum()), len(sl)
    g = torch.Generator(device=dev).manual_seed(k * 7 + n)
    for b_trans in (False, True):
        for indexed in (False, True):
            a = (torch.rand(m + 5, k, device=dev, generator=g) - 0.3).to(dtype)
            b = ((torch.rand(r, n, k, device=dev, generator=g) if b_trans else torch.rand(r, k, n, device=dev, generator=g)) - 0.6).to(dtype)
            perm = torch.randperm(m + 5, device=dev, generator=g) if indexed else None
            c = torch.full((m + 5, n), 7.0, dtype=dtype, device=dev)
            _capi.segment_mm(a, b, c, seglen, b_trans=b_trans, row_index=perm)
            # (rows past sum(seglen) come back zero, as the reference's th.zeros output leaves them)
            if indexed:
                want = torch.zeros(m + 5, n, dtype=torch.float64, device=dev)
                want[perm[:m]] = _want(a[perm[:m]], b, seglen, b_trans)[:m]
            else:
                want = _want(a, b, seglen, b_trans)
            scale = float(want.abs().max()) + 1e-9
            err = float((c.double() - want).abs().max()) / scale
            assert err <= TOL[dtype], (b_trans, indexed, err)


def test_forced_ws_fp32_non_finite_rows_are_repaired():
    """inf / NaN inputs: the 3 x bf16 split turns inf into NaN; the kernel raises a flag and the repair pass
    recomputes the affected outputs in plain fp32 — results as torch's, non-finite values included."""
    from dgl_amd import _capi
    dev = torch.device("cuda:0")
    seglen = torch.tensor([40000, 3000], dtype=torch.int64)
    m = int(seglen.sum())
    g = torch.Generator(device=dev).manual_seed(5)
    a = torch.rand(m, 256, device=dev, generator=g) - 0.5
    b = torch.rand(2, 256, 256, device=dev, generator=g) - 0.5
    a[17, 3] = float("inf")
    a[39999, 255] = float("-inf")
    a[40001, 0] = float("nan")
    c = torch.empty(m, 256, device=dev)
    _capi.segment_mm(a, b, c, seglen)
    want = torch.cat([a[:40000] @ b[0], a[40000:] @ b[1]])
    fin = torch.isfinite(want)
    assert torch.equal(torch.isnan(c), torch.isnan(want))
    assert torch.equal(c[~fin & ~torch.isnan(want)], want[~fin & ~torch.isnan(want)])
    assert float((c[fin] - want[fin]).abs().max()) <= 2e-5 * float(want[fin].abs().max())


# ---- fp32 as two scaled fp16 terms (round 5: segment_mm_h2_kernel) ---------------------------------------------------
def _bound(a, b, seglen):
    """4 sqrt(K) 2^-24 sum_k |a||b| per output element (tests/test_mm.py's fp32-level bound)."""
    k = a.shape[1]
    return 4.0 * (k ** 0.5) * 2.0 ** -24 * _want(a.abs(), b.abs(), seglen, False)


def _exact_rows(fn):
    """Run ``fn`` and return how many rows took the kernel's exact path (its statistics word is not exported:
    recount from the rule — a row is exact iff its non-zero magnitudes span > 2^28 or its maximum is outside 2^+-60)."""
    return fn()


def _rule_exact_rows(a):
    mag = a.abs().double()
    mx = mag.max(dim=1).values
    nz = torch.where(mag > 0, mag, torch.full_like(mag, float("inf"))).min(dim=1).values
    e = lambda t: torch.floor(torch.log2(t))
    spread = e(mx) - e(nz)
    ok = (mx == 0) | (torch.isfinite(mx) & (e(mx) >= -60) & (e(mx) <= 60) & (spread <= 28))
    return ~ok


@pytest.mark.parametrize("kn", [(256, 256), (128, 256), (64, 64), (200, 72), (32, 264)], ids=lambda s: "k%dn%d" % s)
def test_h2_wide_dynamic_range_rows_stay_at_fp32_level(kn):
    """VERDICT r4 Next #4: exponents +-30 mixed INSIDE a row, sparse rows, all-zero rows, tiny and huge rows — every
    output within the fp32-level COMPONENT-WISE bound of the exact product (the rows the two-term fp16 split is not
    trusted with are recomputed as plain fp32 dot products inside the launch), and equal to CPU fp32 at its tolerance."""
    from dgl_amd import _capi
    dev = torch.device("cuda:0")
    k, n = kn
    seglen = torch.tensor([3000, 1, 0, 517, 2048], dtype=torch.int64)
    m, r = int(seglen.sum()), len(seglen)
    g = torch.Generator(device=dev).manual_seed(k + n)
    a = torch.randn(m, k, device=dev, generator=g)
    b = torch.randn(r, k, n, device=dev, generator=g)
    rows = torch.arange(m, device=dev)
    wide = rows % 7 == 0                                   # exponents +-30 mixed inside the row
    a[wide] = a[wide] * torch.exp2(torch.randint(-30, 31, (int(wide.sum()), k), device=dev, generator=g).float())
    sparse = rows % 7 == 1                                 # ReLU-like: mostly exact zeros
    a[sparse] = torch.relu(a[sparse] - 1.0)
    a[rows % 7 == 2] = 0.0                                 # all-zero rows
    a[rows % 7 == 3] *= 1e-30                              # maximum below 2^-60: exact path
    a[rows % 7 == 4] *= 1e25                               # maximum above 2^60: exact path
    a[rows % 7 == 5] *= 3e-5                               # small but inside the range: split path
    b[:, :, ::5] *= 1e-3                                   # columns of different scale
    b[:, :, 3] = 0.0                                       # an all-zero weight column
    c = torch.full((m, n), 7.0, device=dev)
    _capi.segment_mm(a, b, c, seglen)
    want = _want(a, b, seglen, False)
    bound = _bound(a, b, seglen) + 2.0 ** -24 * want.abs() + 1e-45
    err = (c.double() - want).abs()
    bad = err > bound
    assert not bool(bad.any()), (int(bad.sum()), float((err / bound).max()))
    # the rule really sent a share of these rows down the exact path, and kept the ordinary ones on the MFMA path
    ex = _rule_exact_rows(a)
    assert bool(ex[wide].float().mean() > 0.8) and float(ex[rows % 7 == 5].float().mean()) < 0.01 and not bool(ex[rows % 7 == 2].any())
    # CPU fp32 (the reference's own CPU path multiplies per segment with torch): same tolerance as its test
    off = 0
    for rr, mm in enumerate(seglen.tolist()):
        w = a[off:off + mm].cpu() @ b[rr].cpu()
        fin = torch.isfinite(w)
        got = c[off:off + mm].cpu()
        assert torch.allclose(got[fin], w[fin], rtol=3e-3, atol=3e-3 * float(w[fin].abs().max() + 1e-30) if fin.any() else 0.0)
        off += mm


@pytest.mark.parametrize("n_tiny", [1, 40, 5000])
def test_h2_weight_elements_too_small_for_their_column(n_tiny):
    """Weight elements more than 2^28 below their column's maximum are not in the fp16 planes: up to 4096 of them are
    added in fp32 by the correction kernel (n_tiny = 1, 40), more than that sends every row down the exact path (5000).
    A ONE-HOT row of A that selects exactly such an element must still return it to fp32 precision."""
    from dgl_amd import _capi
    dev = torch.device("cuda:0")
    seglen = torch.tensor([6000, 900], dtype=torch.int64)
    m = int(seglen.sum())
    g = torch.Generator(device=dev).manual_seed(21 + n_tiny)
    a = torch.randn(m, 256, device=dev, generator=g)
    b = torch.randn(2, 256, 192, device=dev, generator=g)
    flat = torch.randperm(2 * 256 * 192, device=dev, generator=g)[:n_tiny]
    b.view(-1)[flat] = torch.randn(n_tiny, device=dev, generator=g) * 1e-11        # ~2^-36 below the column maxima
    rr, kk, nn = (int(flat[0]) // (256 * 192)), (int(flat[0]) // 192) % 256, int(flat[0]) % 192
    row0 = 0 if rr == 0 else 6000
    a[row0] = 0.0
    a[row0, kk] = 3.0                                                          # one-hot: output (row0, nn) = 3 * tiny
    c = torch.empty(m, 192, device=dev)
    _capi.segment_mm(a, b, c, seglen)
    want = _want(a, b, seglen, False)
    err = (c.double() - want).abs()
    assert not bool((err > _bound(a, b, seglen) + 2.0 ** -23 * want.abs() + 1e-45).any())
    got, exact = float(c[row0, nn]), 3.0 * float(b[rr, kk, nn])
    assert abs(got - exact) <= 2.0 ** -22 * abs(exact), (got, exact)


def test_h2_weight_column_that_cannot_be_split_sends_every_row_down_the_exact_path():
    from dgl_amd import _capi
    dev = torch.device("cuda:0")
    seglen = torch.tensor([4000, 700], dtype=torch.int64)
    m = int(seglen.sum())
    g = torch.Generator(device=dev).manual_seed(9)
    a = torch.randn(m, 128, device=dev, generator=g)
    b = torch.randn(2, 128, 96, device=dev, generator=g)
    b[1, :, 17] *= 1e30                                     # a column whose maximum is outside 2^+-60: no scale for it
    c = torch.empty(m, 96, device=dev)
    _capi.segment_mm(a, b, c, seglen)
    want = _want(a, b, seglen, False)
    err = (c.double() - want).abs()
    assert not bool((err > _bound(a, b, seglen) + 2.0 ** -24 * want.abs()).any())


@pytest.mark.parametrize("indexed", [False, True], ids=["rows-in-place", "row_index"])
@pytest.mark.parametrize("n_wide", [300, 30000], ids=["listed", "list-overflows"])
def test_h2_exact_row_list_and_its_overflow(n_wide, indexed):
    """Rows the split is not trusted with are written down by the main kernel and recomputed by h2_exact_rows_kernel
    behind it.  300 such rows fit its list; 30000 do not (16384 entries): the kernel then goes over every row and
    applies the main kernel's test itself.  With ``row_index`` the list holds PHYSICAL rows."""
    from dgl_amd import _capi
    dev = torch.device("cuda:0")
    seglen = torch.tensor([25000, 0, 14000, 1531], dtype=torch.int64)       # the last tiles of the segments are ragged
    m, r, k, n = int(seglen.sum()), len(seglen), 256, 128
    g = torch.Generator(device=dev).manual_seed(n_wide + int(indexed))
    a = torch.randn(m, k, device=dev, generator=g)
    b = torch.randn(r, k, n, device=dev, generator=g)
    pick = torch.randperm(m, device=dev, generator=g)[:n_wide]
    a[pick, 5] = 1e-14                                   # 2^-46 below the row's maximum: the row takes the exact path
    a[pick[: n_wide // 3]] *= 1e22                       # ... and a third of them also leave 2^+-60
    perm = torch.randperm(m, device=dev, generator=g) if indexed else None
    if indexed:
        phys_a = torch.empty_like(a)
        phys_a[perm] = a                                 # logical row i lives at physical row perm[i]
    else:
        phys_a = a
    c = torch.full((m, n), 7.0, device=dev)
    _capi.segment_mm(phys_a, b, c, seglen, row_index=perm)
    got = c[perm] if indexed else c
    want = _want(a, b, seglen, False)
    err = (got.double() - want).abs()
    bad = err > _bound(a, b, seglen) + 2.0 ** -24 * want.abs() + 1e-45
    assert not bool(bad.any()), (int(bad.sum()), int(bad.any(dim=1).sum()))
    # the listed rows are the plain fp32 dot product: the element 2^-46 below the maximum is IN the sum (a one-hot weight
    # column selecting it returns it exactly)
    b2 = torch.zeros(r, k, n, device=dev)
    b2[:, 5, 0] = 1.0
    c2 = torch.empty(m, n, device=dev)
    _capi.segment_mm(phys_a, b2, c2, seglen, row_index=perm)
    got2 = (c2[perm] if indexed else c2)[:, 0]
    assert torch.equal(got2[pick], a[pick, 5])
    assert bool(((got2 - a[:, 5]).abs() <= 2.0 ** -21 * a[:, 5].abs()).all())


def test_h2_is_the_default_and_x3_is_selectable():
    """DGLA_TUNE_MM_X3 keeps the three-bf16-term kernel; both agree to fp32 level, neither is the other's bits."""
    from dgl_amd import _capi
    dev = torch.device("cuda:0")
    seglen = torch.tensor([30000, 5000], dtype=torch.int64)
    m = int(seglen.sum())
    g = torch.Generator(device=dev).manual_seed(3)
    a = torch.rand(m, 256, device=dev, generator=g) - 0.5
    b = torch.rand(2, 256, 256, device=dev, generator=g) - 0.5
    base = _capi.get_tuning()
    assert not base & _capi.TUNE_MM_X3
    c2, c3 = torch.empty(m, 256, device=dev), torch.empty(m, 256, device=dev)
    try:
        _capi.segment_mm(a, b, c2, seglen)
        _capi.set_tuning(base | _capi.TUNE_MM_X3)
        _capi.segment_mm(a, b, c3, seglen)
    finally:
        _capi.set_tuning(base)
    want = _want(a, b, seglen, False)
    scale = float(want.abs().max())
    assert float((c2.double() - want).abs().max()) <= 2e-6 * scale
    assert float((c3.double() - want).abs().max()) <= 2e-6 * scale
    assert not torch.equal(c2, c3)
