"""fp32 weight gradient of segment_mm on TWO scaled fp16 terms (csrc/segment_mm.hip: segment_mm_bwd_b_h2_kernel, round 6).

Reference: SegmentMMBackwardB (src/array/cuda/gather_mm.cu:248-291 — one cuBLAS fp32 GEMM A_r^T . dC_r per relation);
tolerance of the reference's own test (tests/python/common/ops/test_ops.py:test_segment_mm: 1e-4 relative) tightened to
the fp32-level COMPONENT-WISE bound tests/test_mm.py uses: 4 sqrt(m) 2^-24 sum_m |a||dc| against the float64 product.

The kernel estimates a power-of-two scale per (relation, column) from a row sample and verifies it while converting; these
tests drive every route: estimate holds (the usual case), element list (values far below their column's maximum), estimate
fails (outlier rows the sample did not see, zero-sample columns, Inf / NaN) -> the three-bf16-term kernel redoes the call.
``dgla_segment_mm_backward_b_last_route`` tells which route ran, so that a silent fall-back cannot pass for the fast path.
"""
import numpy as np
import pytest
import torch

import os

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _force_two_term_route():
    """By its own rule the kernel takes calls of >= 16 384 rows; DGLA_MM_BWD_H2=1 forces it for the small shapes here."""
    old = os.environ.get("DGLA_MM_BWD_H2")
    os.environ["DGLA_MM_BWD_H2"] = "1"
    yield
    if old is None:
        os.environ.pop("DGLA_MM_BWD_H2", None)
    else:
        os.environ["DGLA_MM_BWD_H2"] = old


def _want(a, dc, seglen):
    r = len(seglen)
    out = torch.zeros(r, a.shape[1], dc.shape[1], dtype=torch.float64, device=a.device)
    mag = torch.zeros_like(out)
    off = 0
    for i, m in enumerate(int(v) for v in seglen):
        x, y = a[off:off + m].double(), dc[off:off + m].double()
        out[i] = x.T @ y
        mag[i] = x.abs().T @ y.abs()
        off += m
    return out, mag


def _check(db, a, dc, seglen, what=""):
    want, mag = _want(a, dc, seglen)
    mmax = max(int(v) for v in seglen)
    bound = 4.0 * (max(mmax, 1) ** 0.5) * 2.0 ** -24 * mag + 1e-37
    err = (db.double() - want).abs()
    worst = float((err / bound).max())
    assert worst <= 1.0, (what, worst)


def _run(a, dc, seglen):
    from dgl_amd import _capi
    db = torch.full((len(seglen), a.shape[1], dc.shape[1]), 7.0, device=a.device)
    _capi.segment_mm_backward_b(a, dc, db, torch.tensor(seglen, dtype=torch.int64))
    return db, _capi.segment_mm_backward_b_last_route()


@pytest.mark.parametrize("d12", [(256, 256), (128, 256), (256, 128), (64, 64), (200, 72), (36, 264), (4, 4), (132, 516)],
                         ids=lambda s: "d%dx%d" % s)
@pytest.mark.parametrize("seglen", [[5000, 1, 0, 33, 4097, 31], [70000], [3, 0, 1, 7, 2, 0, 0, 5], [2048, 2049, 16, 15, 17]],
                         ids=["ragged", "long", "tiny", "edges"])
def test_two_term_route_matches_float64(d12, seglen):
    d1, d2 = d12
    g = torch.Generator(device=DEV).manual_seed(d1 * 131 + d2 + len(seglen))
    m = sum(seglen)
    a = torch.randn(m + 3, d1, device=DEV, generator=g)
    dc = torch.randn(m + 3, d2, device=DEV, generator=g) * 3e-3
    db, (fell_back, listed) = _run(a, dc, seglen)
    assert fell_back == 0, "Gaussian data must stay on the two-term route"
    _check(db, a, dc, seglen)


def test_layout_identity():
    """A = rows of the identity against an ASYMMETRIC dC: a swapped row / column or a wrong lane -> row mapping of the
    conversion stage cannot pass; integers below 2^11 are exact in the high plane alone."""
    d1, d2, m = 256, 256, 256
    a = torch.eye(m, d1, device=DEV)
    dc = (torch.arange(m * d2, device=DEV).reshape(m, d2) % 251).float()
    db, (fell_back, _) = _run(a, dc, [m])
    assert fell_back == 0
    assert torch.equal(db[0], dc)          # dB = I^T dC


def test_column_scales_differ_by_sixty_binades():
    """Every column of A and of dC at its own magnitude (2^-30 .. 2^30): the per-column scales absorb it."""
    g = torch.Generator(device=DEV).manual_seed(3)
    seglen = [9000, 4000]
    m = sum(seglen)
    a = torch.randn(m, 256, device=DEV, generator=g) * torch.exp2(torch.randint(-30, 31, (256,), device=DEV, generator=g).float())
    dc = torch.randn(m, 128, device=DEV, generator=g) * torch.exp2(torch.randint(-30, 31, (128,), device=DEV, generator=g).float())
    db, (fell_back, listed) = _run(a, dc, seglen)
    assert fell_back == 0
    _check(db, a, dc, seglen)


def test_elements_far_below_their_column_go_through_the_list():
    """Column 5 of A: one large value in a row whose dC is zero, tiny values (2^-30 of it) elsewhere — the output row is made
    of the tiny elements alone, each of which would keep < 21 bits under the column's scale.  They are listed and added in
    fp32: the row is correct to fp32 level although the column's scale is set by a value 10^9 times larger."""
    g = torch.Generator(device=DEV).manual_seed(4)
    m, d1, d2 = 6000, 128, 128
    a = torch.randn(m, d1, device=DEV, generator=g)
    dc = torch.randn(m, d2, device=DEV, generator=g)
    a[:, 5] = 0.0
    a[0, 5] = 1024.0
    dc[0] = 0.0
    rows = torch.tensor([17, 1000, 1001, 4099, 5999], device=DEV)
    a[rows, 5] = torch.tensor([3.1, -2.7, 1.9, 2.2, -3.3], device=DEV) * 2.0 ** -20      # 2^-30 of the column's maximum
    a[4001, 5] = 2.0 ** -32                                                              # high term zero, low term not
    # the same on the dC side (large value in row 2, which the sample of 2 048 of 6 000 rows sees), and rows where BOTH
    # operands hold a listed element
    dc[:, 9] = 0.0
    dc[2, 9] = 512.0
    a[2] = 0.0
    dc[rows, 9] = torch.tensor([1.1, 1.7, -1.3, 2.9, 0.7], device=DEV) * 2.0 ** -21
    db, (fell_back, listed) = _run(a, dc, [m])
    assert fell_back == 0 and listed == 11, (fell_back, listed)
    _check(db, a, dc, [m])
    want, _ = _want(a, dc, [m])
    assert float((db[0, 5].double() - want[0, 5]).abs().max() / want[0, 5].abs().max()) < 1e-6
    assert float((db[0, :, 9].double() - want[0, :, 9]).abs().max() / want[0, :, 9].abs().max()) < 1e-6


def test_wide_dynamic_range_rows():
    """Rows scaled by 2^+-20 on both operands (segments longer than the sample): whatever route the call takes — list,
    list overflow, estimate failure — the result stays inside the fp32-level bound."""
    g = torch.Generator(device=DEV).manual_seed(5)
    seglen = [30000, 20000, 700]
    m = sum(seglen)
    a = torch.randn(m, 192, device=DEV, generator=g) * torch.exp2(torch.randint(-20, 21, (m, 1), device=DEV, generator=g).float())
    dc = torch.randn(m, 160, device=DEV, generator=g) * torch.exp2(torch.randint(-20, 21, (m, 1), device=DEV, generator=g).float())
    db, route = _run(a, dc, seglen)
    _check(db, a, dc, seglen, route)


@pytest.mark.parametrize("which", ["outlier", "zero_sample", "inf", "nan", "huge", "denormal_column"])
def test_inputs_outside_the_estimate_are_redone_by_the_three_term_kernel(which):
    g = torch.Generator(device=DEV).manual_seed(6)
    seglen = [100000, 5000]
    m = sum(seglen)
    a = torch.randn(m, 256, device=DEV, generator=g)
    dc = torch.randn(m, 256, device=DEV, generator=g)
    # rows the sample (2 048 evenly spread rows: floor(k len / 2048)) does not see: 100000 / 2048 = 48.8 -> row 1 is never one
    if which == "outlier":
        a[1, 7] = 1e6
    elif which == "zero_sample":
        dc[:100000, 11] = 0.0
        dc[1, 11] = 1e-20
    elif which == "inf":
        a[1, 0] = float("inf")
    elif which == "nan":
        dc[100001, 3] = float("nan")
    elif which == "huge":
        a[:, 2] *= 2.0 ** 70                      # a sampled maximum outside 2^+-60
    elif which == "denormal_column":
        dc[:, 4] = 0.0
        dc[100001, 4] = 1e-42                     # an fp32 denormal in a column whose sample is all zero
    db, (fell_back, _) = _run(a, dc, seglen)
    assert fell_back != 0, which
    want = torch.stack([a[:100000].T @ dc[:100000], a[100000:].T @ dc[100000:]])
    fin = torch.isfinite(want)
    assert torch.equal(torch.isnan(db), torch.isnan(want))
    assert torch.equal(db[~fin & ~torch.isnan(want)], want[~fin & ~torch.isnan(want)])
    want64, mag = _want(torch.nan_to_num(a, nan=0.0, posinf=0.0, neginf=0.0), torch.nan_to_num(dc, nan=0.0, posinf=0.0, neginf=0.0), seglen)
    bound = 4.0 * (100000 ** 0.5) * 2.0 ** -24 * mag + 1e-37
    ok = fin
    assert float(((db.double() - want64).abs() / bound)[ok].max()) <= 1.0


def test_three_term_and_plain_fp32_routes_stay_selectable_and_agree():
    from dgl_amd import _capi
    from dgl_amd._lib import DGLA_TUNE_MM_F32, DGLA_TUNE_MM_X3
    g = torch.Generator(device=DEV).manual_seed(7)
    seglen = [20000, 3000, 5]
    m = sum(seglen)
    a = torch.randn(m, 256, device=DEV, generator=g)
    dc = torch.randn(m, 256, device=DEV, generator=g)
    default = _capi.get_tuning()
    got = {}
    try:
        for name, flags in (("h2", default), ("x3", default | DGLA_TUNE_MM_X3), ("f32", default | DGLA_TUNE_MM_F32)):
            _capi.set_tuning(flags)
            got[name], _ = _run(a, dc, seglen)
            _check(got[name], a, dc, seglen, name)
    finally:
        _capi.set_tuning(default)
    scale = float(got["f32"].abs().max())
    assert float((got["h2"] - got["f32"]).abs().max()) <= 2e-5 * scale
    assert float((got["x3"] - got["f32"]).abs().max()) <= 2e-5 * scale


def test_small_calls_stay_on_the_three_term_kernel():
    """Without the override a call below 16 384 rows does not take the two-term route (its statistics stay those of the
    previous two-term launch), a larger one does."""
    from dgl_amd import _capi
    g = torch.Generator(device=DEV).manual_seed(8)
    a = torch.randn(40000, 64, device=DEV, generator=g)
    dc = torch.randn(40000, 64, device=DEV, generator=g)
    a[5, 5] = 0.0
    os.environ.pop("DGLA_MM_BWD_H2", None)
    db, route = _run(a, dc, [40000])
    assert route == (0, 0)
    _check(db, a, dc, [40000])
    a[5, 5] = 2.0 ** -40          # would be listed by the two-term route
    db, route = _run(a[:900], dc[:900], [128, 1, 127, 129, 0, 0, 256, 259])
    assert route == (0, 0)        # untouched: the three-term kernel ran
    for i, (o, m) in enumerate(((0, 128), (128, 1), (129, 127))):
        want, mag = _want(a[o:o + m], dc[o:o + m], [m])
        assert float(((db[i].double() - want[0]).abs() / (4.0 * m ** 0.5 * 2.0 ** -24 * mag[0] + 1e-37)).max()) <= 1.0
    db, route = _run(a, dc, [40000])
    assert route == (0, 1)


def test_empty_and_degenerate_calls():
    a = torch.randn(10, 64, device=DEV)
    dc = torch.randn(10, 64, device=DEV)
    db, _ = _run(a, dc, [0, 0, 0])
    assert float(db.abs().max()) == 0.0
    z = torch.zeros(5000, 64, device=DEV)
    db, (fell_back, listed) = _run(z, dc.repeat(500, 1), [5000])
    assert fell_back == 0 and listed == 0 and float(db.abs().max()) == 0.0
