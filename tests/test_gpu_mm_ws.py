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
