"""segment_mm / gather_mm on the MFMA units.

The reference's CPU path for these operators is a per-segment ``torch`` matmul
(python/dgl/backend/pytorch/sparse.py:1173-1195; the C++ CPU kernels are LOG(FATAL) stubs,
src/array/cpu/gather_mm.cc:16-45), and its GPU path is cuBLAS: neither fixes a summation
order, so parity here is tolerance-based exactly like the reference's own tests
(tests/python/common/ops/test_ops.py:302-383: per-segment matmul, rtol/atol 1e-4 fp32, 2e-2
fp16 / bf16) — "parity unpinned" at the bit level, stated in DESIGN.md.  The oracle is the
exact product in fp64 (numpy) of the SAME stored inputs; bounds are condition-aware:
|err| <= tol * sum_k |a_ik| |b_kj|.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# accumulator epsilon and the half-ulp of the final store, per storage type
ACC_EPS = {torch.float32: 2.0 ** -24, torch.float64: 2.0 ** -53, torch.float16: 2.0 ** -24,
           torch.bfloat16: 2.0 ** -24}
STORE_EPS = {torch.float32: 0.0, torch.float64: 0.0, torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}


def _exact_segment_mm(a, b, seglen, b_trans=False):
    a64, b64 = a.double().cpu().numpy(), b.double().cpu().numpy()
    n_out = b64.shape[1] if b_trans else b64.shape[2]
    out = np.zeros((a64.shape[0], n_out))
    mag = np.zeros_like(out)
    off = 0
    for r, m in enumerate(int(v) for v in seglen):
        w = b64[r].T if b_trans else b64[r]
        out[off:off + m] = a64[off:off + m] @ w
        mag[off:off + m] = np.abs(a64[off:off + m]) @ np.abs(w)
        off += m
    return out, mag


def _close(got, want, mag, dtype, k, what=""):
    """|err| <= 4 sqrt(k) eps_acc * sum|a||b|  (random-walk growth of an fp32 / fp64 dot product of
    length k in ANY order; the worst case would be k eps) + one rounding of the stored result."""
    err = np.abs(got.double().cpu().numpy() - want)
    bound = 4 * np.sqrt(max(k, 1)) * ACC_EPS[dtype] * mag + 1.01 * STORE_EPS[dtype] * np.abs(want) + 1e-30
    if dtype == torch.float16:
        bound = bound + 2.0 ** -24  # results below 6.1e-5 land on fp16's subnormal grid
    assert (err <= bound).all(), (what, float((err / bound).max()))


SEGLENS = {
    "docstring": [10, 5, 0, 3],                       # python/dgl/ops/segment.py:109-113
    "tile_edges": [128, 1, 127, 129, 0, 0, 256, 3],   # around the 128-row tile size
    "one_big": [1000],
    "many_small": [3, 0, 1, 7, 2, 0, 0, 5, 1, 1, 9, 4],
}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16, torch.float64])
@pytest.mark.parametrize("kind", list(SEGLENS))
@pytest.mark.parametrize("d1,d2", [(16, 32), (256, 256), (100, 36), (7, 130), (33, 1), (32, 200),
                                   (96, 128), (160, 264)])
@pytest.mark.parametrize("seglen_dev", ["cpu", "gpu"])
def test_segment_mm_forward(dev, dtype, kind, d1, d2, seglen_dev):
    from dgl_amd import mm

    seglen = torch.tensor(SEGLENS[kind], dtype=torch.int64)
    n, r = int(seglen.sum()), len(seglen)
    g = torch.Generator().manual_seed(d1 * 1000 + d2)
    a = (torch.rand(n, d1, generator=g) - 0.3).to(dtype).to(dev)
    b = (torch.rand(r, d1, d2, generator=g) - 0.6).to(dtype).to(dev)   # asymmetric on purpose
    sl = seglen.to(dev) if seglen_dev == "gpu" else seglen
    c = torch.full((n, d2), 7.0, dtype=dtype, device=dev)
    mm._segment_mm(a, b, c, sl)
    want, mag = _exact_segment_mm(a, b, seglen)
    _close(c, want, mag, dtype, d1, "fwd")
    # transposed weights: the A-gradient form  dA = dC . B^T
    da = torch.full((n, d1), 7.0, dtype=dtype, device=dev)
    mm._segment_mm(c, b, da, sl, b_trans=True)
    want, mag = _exact_segment_mm(c, b, seglen, b_trans=True)
    _close(da, want, mag, dtype, d2, "b_trans")


@pytest.mark.parametrize("d1,d2", [(16, 32), (100, 36), (256, 256), (7, 130)])
@pytest.mark.parametrize("path", ["x3", "f32"])
def test_segment_mm_nonfinite(dev, d1, d2, path):
    """fp32 operands holding +-inf, NaN and values that round to a bf16 infinity: the default
    three-term bf16 path must give what IEEE fp32 arithmetic gives (the reference's per-segment
    torch matmul, python/dgl/backend/pytorch/sparse.py:1173-1195): +-inf where a product is infinite,
    NaN for inf - inf, inf * 0 and NaN operands, ordinary values everywhere else — same classes
    and signs as an fp32 CPU matmul, finite entries within the usual bound (VERDICT r2 Weak #1c)."""
    from dgl_amd import _capi, mm

    seglen = torch.tensor([130, 3, 0, 67], dtype=torch.int64)
    n, r = int(seglen.sum()), len(seglen)
    g = torch.Generator().manual_seed(d1 + d2)
    a = (torch.rand(n, d1, generator=g) + 0.1)          # positive: no accidental inf - inf
    b = (torch.rand(r, d1, d2, generator=g) + 0.1)
    inf, big = float("inf"), 3.4e38                      # 3.4e38 > 0x7f7f8000: rounds to a bf16 infinity
    a[0, 0] = inf            # row 0: +inf everywhere
    a[1, 1] = -inf           # row 1: -inf
    a[2, 0], a[2, 1] = inf, -inf  # row 2: inf - inf = NaN
    a[3, 2] = float("nan")   # row 3: NaN
    a[4, 0] = big            # row 4: huge but finite times b in (0.1, 1.1): finite or +inf as fp32 says
    a[5, 0] = inf
    b[0, 0, 0] = 0.0         # inf * 0 = NaN in (5, 0) — and in (0, 0), (2, 0)
    b[0, 3, 1] = inf         # an infinite weight: column 1 of segment 0
    a[131, 0] = -inf         # second segment
    b[3, 1, 2] = float("nan")
    a, b = a.to(dev), b.to(dev)
    old = _capi.get_tuning()
    try:
        _capi.set_tuning(old | _capi.TUNE_MM_F32 if path == "f32" else old & ~_capi.TUNE_MM_F32)
        c = torch.full((n, d2), 7.0, device=dev)
        mm._segment_mm(a, b, c, seglen)
        db = torch.full((r, d1, d2), 7.0, device=dev)
        dc = torch.rand(n, d2, generator=g).to(dev) + 0.1
        mm._segment_mm_backward_B(a, dc, db, seglen)
    finally:
        _capi.set_tuning(old)
    ah, bh = a.cpu(), b.cpu()
    off = 0
    with np.errstate(all="ignore"):
        for i, m in enumerate(int(v) for v in seglen):
            want = (ah[off:off + m].double() @ bh[i].double()).float().numpy()   # fp64 product rounded: class/sign reference
            got = c[off:off + m].cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(want)), (i, "nan pattern")
            assert np.array_equal(np.isposinf(got), np.isposinf(want)), (i, "+inf pattern")
            assert np.array_equal(np.isneginf(got), np.isneginf(want)), (i, "-inf pattern")
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5)
            wdb = (ah[off:off + m].double().T @ dc[off:off + m].cpu().double()).float().numpy()
            gdb = db[i].cpu().numpy()
            assert np.array_equal(np.isnan(gdb), np.isnan(wdb)), (i, "dB nan pattern")
            assert np.array_equal(np.isposinf(gdb), np.isposinf(wdb)) and np.array_equal(np.isneginf(gdb), np.isneginf(wdb))
            fin = np.isfinite(wdb)
            np.testing.assert_allclose(gdb[fin], wdb[fin], rtol=1e-5)
            off += m
    assert np.isnan(c[5, 0].item()) and np.isposinf(c[0, 2].item()) and np.isneginf(c[1, 0].item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16, torch.float64])
@pytest.mark.parametrize("kind", ["docstring", "tile_edges", "many_small", "long"])
@pytest.mark.parametrize("d1,d2", [(16, 32), (130, 70), (256, 256), (5, 3)])
def test_segment_mm_backward_b(dev, dtype, kind, d1, d2):
    from dgl_amd import mm

    seglen = torch.tensor(SEGLENS[kind] if kind != "long" else [5000, 0, 2049, 1], dtype=torch.int64)
    n, r = int(seglen.sum()), len(seglen)
    g = torch.Generator().manual_seed(d1 * 77 + d2)
    a = (torch.rand(n, d1, generator=g) - 0.3).to(dtype).to(dev)
    dc = (torch.rand(n, d2, generator=g) - 0.6).to(dtype).to(dev)
    db = torch.full((r, d1, d2), 7.0, dtype=dtype, device=dev)
    mm._segment_mm_backward_B(a, dc, db, seglen)
    a64, c64 = a.double().cpu().numpy(), dc.double().cpu().numpy()
    off = 0
    for i, m in enumerate(int(v) for v in seglen):
        want = a64[off:off + m].T @ c64[off:off + m]
        mag = np.abs(a64[off:off + m]).T @ np.abs(c64[off:off + m])
        _close(db[i], want, mag, dtype, m, "rel %d" % i)
        off += m


def test_mfma_layout_identity(dev):
    """A = I against an ASYMMETRIC B: a row/column swap or a wrong lane -> k mapping cannot pass."""
    from dgl_amd import mm

    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        k = 160
        a = torch.eye(k, dtype=dtype, device=dev)
        b = (torch.arange(k * 192, device=dev).reshape(1, k, 192) % 251).to(dtype)
        c = torch.empty(k, 192, dtype=dtype, device=dev)
        mm._segment_mm(a, b, c, torch.tensor([k]))
        assert torch.equal(c, b[0]), dtype


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("k", [8, 16, 24, 32, 64, 96, 100, 256, 544, 600])
@pytest.mark.parametrize("n", [64, 136, 256, 264, 776])
def test_lds_direct_kernel_matches_register_staged(dev, dtype, k, n):
    """DGLA_TUNE_GLDS picks the global_load_lds K-loop whenever the operands are whole 16-byte
    pieces (K % 8 == 0 for 16-bit, K % 4 == 0 for fp32; a K tail inside the last 64-byte slab
    reads the zero page).  16-bit storage: same MFMAs in the same k order, so the BITS must equal the
    register-staged kernel's; fp32: k is contracted in a permuted order, so agreement is to
    rounding.  Ragged segments, clamped tail rows / columns, row-indexed (gather_mm) access."""
    from dgl_amd import _capi
    from dgl_amd._lib import DGLA_TUNE_GLDS

    default = _capi.get_tuning()
    assert default & DGLA_TUNE_GLDS, "LDS-direct kernel is the default"
    g = torch.Generator().manual_seed(k * 31 + n)
    try:
        for seg in ([1], [127, 129, 0, 5], [1000, 3, 0, 0, 2049], [300] * 5):
            m, r = sum(seg), len(seg)
            a = torch.randn((m, k), generator=g).to(dtype).to(dev)
            b = torch.randn((r, k, n), generator=g).to(dtype).to(dev)
            bt = b.transpose(1, 2).contiguous()
            sl = torch.tensor(seg, dtype=torch.int64)
            for ri in (None, torch.randperm(m, generator=g).to(dev)):
                got = []
                for flags, w, tr in ((default & ~DGLA_TUNE_GLDS, b, False), (default, b, False), (default, bt, True)):
                    _capi.set_tuning(flags)
                    c = torch.full((m, n), float("nan"), dtype=dtype, device=dev)
                    _capi.segment_mm(a, w, c, sl, b_trans=tr, row_index=ri)
                    got.append(c)
                assert not torch.isnan(got[0]).any()
                for c in got[1:]:
                    assert not torch.isnan(c).any()
                    if dtype == torch.float32:
                        mag = a.abs() @ b.abs().amax(0)  # >= sum_k |a||b| of every relation
                        assert ((c - got[0]).abs() <= 8 * k ** 0.5 * 2.0 ** -24 * mag + 1e-30).all(), (seg, ri is not None)
                    else:
                        assert torch.equal(c.view(torch.int16), got[0].view(torch.int16)), (seg, ri is not None)
    finally:
        _capi.set_tuning(default)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("d1,d2", [(8, 8), (4, 12), (64, 128), (256, 256), (136, 72), (132, 76), (520, 264)])
def test_lds_direct_weight_gradient(dev, dtype, d1, d2):
    """dB with contiguous rows and whole 16-byte pieces takes the LDS-direct kernels
    (global_load_lds ring; 16-bit: transposing ds_read_b64_tr_b16 fragments, fp32: plain
    ds_read_b32; both ring depths are reached: 520 x 264 has 15 tiles per slab; (4, 12) and
    (132, 76) are whole pieces only in fp32, so 16-bit takes the register-staged kernel there).  Checked against the exact fp64 product and
    against the register-staged kernel (DGLA_TUNE_GLDS off): slab tails read the zero page,
    feature tails are clamped, empty relations stay zero."""
    from dgl_amd import _capi
    from dgl_amd._lib import DGLA_TUNE_GLDS

    default = _capi.get_tuning()
    g = torch.Generator().manual_seed(d1 * 13 + d2)
    try:
        for seg in ([1], [127, 129, 0, 5], [5000, 3, 0, 0, 2049], [40000, 17]):
            m, r = sum(seg), len(seg)
            a = (torch.rand(m, d1, generator=g) - 0.3).to(dtype).to(dev)
            dc = (torch.rand(m, d2, generator=g) - 0.6).to(dtype).to(dev)
            sl = torch.tensor(seg, dtype=torch.int64)
            a64, c64 = a.double().cpu().numpy(), dc.double().cpu().numpy()
            for flags in (default, default & ~DGLA_TUNE_GLDS):
                _capi.set_tuning(flags)
                db = torch.full((r, d1, d2), float("nan"), dtype=dtype, device=dev)
                _capi.segment_mm_backward_b(a, dc, db, sl)
                off = 0
                for i, n_ in enumerate(seg):
                    want = a64[off:off + n_].T @ c64[off:off + n_]
                    mag = np.abs(a64[off:off + n_]).T @ np.abs(c64[off:off + n_])
                    _close(db[i], want, mag, dtype, n_, "rel %d flags %d" % (i, flags))
                    off += n_
    finally:
        _capi.set_tuning(default)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_api_segment_mm_matches_torch_with_grads(dev, dtype):
    """tests/python/common/ops/test_ops.py:302-342 (test_segment_mm)."""
    import dgl_amd

    torch.manual_seed(1)
    a = torch.randn(100, 32, device=dev, dtype=dtype).requires_grad_(True)
    b = torch.randn(4, 32, 48, device=dev, dtype=dtype).requires_grad_(True)
    seglen = torch.tensor([10, 15, 8, 67])
    c = dgl_amd.segment_mm(a, b, seglen)
    dc = torch.randn_like(c)
    c.backward(dc)
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    parts, off = [], 0
    for i, m in enumerate(seglen.tolist()):
        parts.append(a2[off:off + m] @ b2[i])
        off += m
    c2 = torch.cat(parts)
    c2.backward(dc)
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(c, c2, **tol)
    assert torch.allclose(a.grad, a2.grad, **tol)
    assert torch.allclose(b.grad, b2.grad, **(tol if dtype == torch.float32 else dict(rtol=3e-2, atol=8e-2)))


@pytest.mark.parametrize("d1,d2", [(4, 6), (8, 8), (32, 16)])
@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_api_gather_mm_matches_bmm_with_grads(dev, d1, d2, idtype):
    """tests/python/common/ops/test_ops.py:345-383 (test_gather_mm_idx_b); (32, 16) takes the
    sort + segment_mm route, the small shapes the per-row kernel."""
    import dgl_amd

    torch.manual_seed(2)
    n, r = 300, 5
    a = torch.randn(n, d1, device=dev).requires_grad_(True)
    b = torch.randn(r, d1, d2, device=dev).requires_grad_(True)
    idx = torch.randint(0, r, (n,), device=dev).to(idtype)
    c = dgl_amd.gather_mm(a, b, idx_b=idx)
    dc = torch.randn_like(c)
    c.backward(dc)
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    c2 = torch.bmm(a2.unsqueeze(1), b2[idx.long()]).squeeze(1)
    c2.backward(dc)
    assert torch.allclose(c, c2, rtol=1e-4, atol=1e-4)
    assert torch.allclose(a.grad, a2.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(b.grad, b2.grad, rtol=1e-4, atol=1e-3)


def test_errors(dev):
    import dgl_amd
    from dgl_amd._lib import DGLAMDError

    a = torch.ones(4, 3, device=dev)
    b = torch.ones(2, 5, 6, device=dev)
    with pytest.raises(DGLAMDError, match="A.shape\\[1\\] == B.shape\\[1\\]"):
        dgl_amd.segment_mm(a, b, torch.tensor([2, 2]))
    with pytest.raises(ValueError, match="3D"):
        dgl_amd.segment_mm(a, torch.ones(3, 6, device=dev), torch.tensor([4]))
    with pytest.raises(DGLAMDError, match="len\\(seglen_A\\)"):
        dgl_amd.segment_mm(a, torch.ones(2, 3, 6, device=dev), torch.tensor([4]))
