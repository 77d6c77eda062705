"""Segment reduce / scatter add / backward-of-segment-max.

CPU (`-m "not gpu"`): the C restatement in oracle/ is pinned to the reference's own
src/array/cpu/segment_reduce.cc (oracle/_ref) bit for bit on an exhaustive sweep, and to the
committed golden outputs of that build.  GPU (`-m gpu`): the HIP path, through the C ABI and
through the registry names, against the golden outputs and the oracle — arg and max/min
values bit-exact, fp32 sums within 1e-5, plus the reference's own API-level checks
(tests/python/common/ops/test_ops.py:226-262) against plain torch.
"""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import ref
from tests.segment_cases import all_cases, run_case

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                      "reference_segment_outputs.npz")
CASES = all_cases(full=False)


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def _outputs(golden, name):
    pre = name + "/out/"
    return {k[len(pre):]: golden[k] for k in golden.files if k.startswith(pre)}


# ---- CPU: oracle pinned to the reference -------------------------------------------------
@pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libdglref.so not built")
def test_oracle_equals_reference_build_bit_exact():
    ref.set_num_threads(1)
    bad = []
    cases = all_cases(full=True)
    assert len(cases) > 350
    for c in cases:
        got, want = run_case(oracle, c), run_case(ref, c)
        for k in want:
            if not _same(got[k], want[k]):
                bad.append((c["name"], k))
    ref.set_num_threads(os.cpu_count() or 1)
    assert not bad, bad[:20]


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_golden(golden, c):
    for k in c:
        if isinstance(c[k], np.ndarray):
            assert _same(golden["%s/in/%s" % (c["name"], k)], c[k]), "inputs drifted: " + k
    want = _outputs(golden, c["name"])
    got = {k: v for k, v in run_case(oracle, c).items() if v is not None}
    assert set(got) == set(want)
    for k in want:
        assert _same(got[k], want[k]), k


def test_docstring_example():
    out, _ = oracle.segment_reduce("sum", np.ones((10, 3), np.float32), np.array([0, 1, 1, 6, 10]))
    np.testing.assert_array_equal(out, np.array([[1.] * 3, [0.] * 3, [5.] * 3, [4.] * 3], np.float32))
    out, arg = oracle.segment_reduce("max", np.arange(10, dtype=np.float64).reshape(10, 1),
                                     np.array([0, 1, 1, 6, 10], np.int32))
    assert out[1, 0] == -np.inf and arg[1, 0] == -1 and arg[3, 0] == 9 and arg.dtype == np.int32


def _assert_sum_close(got, want, c):
    """The features are standard normal, so segment sums cancel: the error of ANY summation
    order is bounded by n * eps * sum|x| (not by a multiple of the result).  Bar: within
    (1e-5 + 2 n eps) * sum|x| of the reference value, n = longest segment — for positive data
    this is the north-star's 1e-5 relative."""
    f64 = c["feat"].astype(np.float64)
    mag, _ = oracle.segment_reduce("sum", np.abs(f64), c["offsets"])
    longest = int(np.diff(c["offsets"]).max()) if len(c["offsets"]) > 1 else 0
    eps = 2.0 ** -24 if c["feat"].dtype == np.float32 else 2.0 ** -53
    bound = (1e-5 * (eps / 2.0 ** -24) + 2 * longest * eps) * mag + 1e-30
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert (err <= bound).all(), (c["name"], float((err / bound).max()))


# ---- GPU ---------------------------------------------------------------------------------
def _gpu_run(dev, c, via):
    from dgl_amd import _capi
    from dgl_amd import segment as S

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if c["kind"] == "segment_reduce":
        feat, off = t(c["feat"]), t(c["offsets"])
        if via == "registry":
            out, arg = S._segment_reduce(c["reduce"], feat, off)
        else:
            n = off.shape[0] - 1
            out = torch.full((n,) + tuple(feat.shape[1:]), 3.0, dtype=feat.dtype, device=dev)
            arg = None if c["reduce"] == "sum" else torch.full(out.shape, 77, dtype=off.dtype, device=dev)
            ws = None
            if via == "workspace":
                ws = torch.empty(max(1, _capi.segment_reduce_workspace_bytes(c["reduce"], feat, off, out)),
                                 dtype=torch.uint8, device=dev)
            _capi.segment_reduce(c["reduce"], feat, off, out, arg, ws)
            if ws is not None:  # cached plan: same bits
                out2 = torch.full_like(out, 5.0)
                _capi.segment_reduce(c["reduce"], feat, off, out2, arg, ws, plan_valid=True)
                assert torch.equal(out.view(torch.uint8), out2.view(torch.uint8))
        res = {"out": out.cpu().numpy(), "arg": None if arg is None else arg.cpu().numpy()}
        if arg is not None:
            dy = t((np.arange(out.numel(), dtype=np.float64).reshape(out.shape) / 7 + 1).astype(c["feat"].dtype))
            if via == "registry":
                back = S._bwd_segment_cmp(dy, arg, feat.shape[0])
            else:
                back = torch.zeros_like(feat)
                if back.numel():
                    _capi.backward_segment_cmp(dy, arg, back)
            res["back"] = back.cpu().numpy()
        return res
    feat, idx = t(c["feat"]), t(c["idx"])
    if via == "registry":
        out = S._scatter_add(feat, idx, c["m"])
    else:
        out = torch.zeros((c["m"],) + tuple(feat.shape[1:]), dtype=feat.dtype, device=dev)
        _capi.scatter_add(feat, idx, out)
    return {"out": out.cpu().numpy()}


@pytest.mark.gpu
@pytest.mark.parametrize("via", ["seam", "workspace", "registry"])
@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_gpu_matches_reference_outputs(dev, golden, c, via):
    want = _outputs(golden, c["name"])
    got = {k: v for k, v in _gpu_run(dev, c, via).items() if v is not None}
    assert set(got) == set(want), (set(got), set(want))
    for k, v in want.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
        if c["kind"] == "segment_reduce" and c["reduce"] == "sum":
            _assert_sum_close(got[k], v, c)
        else:  # max/min values, args, scattered gradients, exact-by-construction scatter sums
            np.testing.assert_array_equal(got[k], v, err_msg="%s/%s" % (c["name"], k))


@pytest.mark.gpu
@pytest.mark.parametrize("idtype", [np.int32, np.int64])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("reduce", ["sum", "max", "min"])
def test_gpu_full_sweep_vs_oracle(dev, reduce, dtype, idtype):
    """Every shape/kind of the exhaustive sweep for one (reduce, dtype, idtype)."""
    for c in all_cases(full=True):
        if c["kind"] != "segment_reduce" or c["reduce"] != reduce or c["feat"].dtype != dtype or \
                c["offsets"].dtype != idtype:
            continue
        want = run_case(oracle, c)
        got = _gpu_run(dev, c, "seam")
        if reduce == "sum":
            exact = run_case(oracle, dict(c, feat=c["feat"].astype(np.float64)))["out"]
            _assert_sum_close(got["out"], exact, c)
        else:
            for k in ("out", "arg", "back"):
                np.testing.assert_array_equal(got[k], want[k], err_msg="%s/%s" % (c["name"], k))


@pytest.mark.gpu
@pytest.mark.parametrize("tdtype", [torch.float16, torch.bfloat16])
def test_gpu_half_precision(dev, tdtype):
    """16-bit storage: fp32 accumulation, one rounding at the end (the reference's
    accum_dtype rule, src/array/cuda/segment_reduce.cuh:37); max/min exact."""
    from dgl_amd import _capi

    rng = np.random.default_rng(3)
    seglen = rng.integers(0, 40, 200)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(seglen)])).to(dev)
    feat = torch.from_numpy(rng.standard_normal((int(seglen.sum()), 24)).astype(np.float32)).to(dev).to(tdtype)
    out = torch.empty(200, 24, dtype=tdtype, device=dev)
    _capi.segment_reduce("sum", feat, off, out)
    want, _ = oracle.segment_reduce("sum", feat.float().cpu().numpy(), off.cpu().numpy())
    np.testing.assert_allclose(out.float().cpu().numpy(), want, rtol=1e-2, atol=5e-2)
    arg = torch.empty(200, 24, dtype=torch.int64, device=dev)
    _capi.segment_reduce("max", feat, off, out, arg)
    want, warg = oracle.segment_reduce("max", feat.float().cpu().numpy(), off.cpu().numpy())
    np.testing.assert_array_equal(arg.cpu().numpy(), warg)
    w = torch.from_numpy(want)
    ok = torch.isinf(w) | (out.float().cpu() == w)
    assert bool(ok.all())
    # fp16 max of an empty segment is the reference's largest finite value, bf16 -inf
    # scatter add in 16-bit storage: exact for small integers
    idx = torch.from_numpy(rng.integers(0, 9, 300)).to(dev)
    x = torch.from_numpy(rng.integers(-3, 4, (300, 10)).astype(np.float32)).to(dev).to(tdtype)
    acc = torch.zeros(9, 10, dtype=tdtype, device=dev)
    _capi.scatter_add(x, idx, acc)
    want = torch.zeros(9, 10).index_add_(0, idx.cpu(), x.float().cpu())
    assert torch.equal(acc.float().cpu(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("reducer", ["sum", "max", "min", "mean"])
def test_api_segment_reduce_forward_backward(dev, reducer):
    """tests/python/common/ops/test_ops.py:226-262: forward and gradient against a per-segment
    torch loop."""
    import dgl_amd

    torch.manual_seed(0)
    seglen = torch.tensor([2, 0, 7, 1, 0, 640, 3], device=dev)
    v1 = torch.randn(int(seglen.sum()), 5, device=dev, requires_grad=True)
    v2 = v1.detach().clone().requires_grad_(True)
    out = dgl_amd.segment_reduce(seglen, v1, reducer=reducer)
    chunks = torch.split(v2, seglen.tolist())
    fn = {"sum": lambda c: c.sum(0), "mean": lambda c: c.mean(0), "max": lambda c: c.max(0)[0],
          "min": lambda c: c.min(0)[0]}[reducer]
    want = torch.stack([fn(c) if len(c) else torch.zeros(5, device=dev) for c in chunks])
    assert torch.allclose(out, want, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(out)
    out.backward(g)
    want.backward(g)
    assert torch.allclose(v1.grad, v2.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_api_segment_softmax_and_docstrings(dev):
    import dgl_amd

    val = torch.ones(10, 3, device=dev)
    seg = torch.tensor([1, 0, 5, 4], device=dev)
    assert torch.equal(dgl_amd.segment_reduce(seg, val).cpu(),
                       torch.tensor([[1.] * 3, [0.] * 3, [5.] * 3, [4.] * 3]))
    sm = dgl_amd.segment_softmax(seg, val).cpu()
    want = torch.tensor([1.0] + [0.2] * 5 + [0.25] * 4)[:, None].expand(10, 3)
    assert torch.allclose(sm, want)
    # gradient of scatter_add is a gather
    x = torch.randn(6, 4, device=dev, requires_grad=True)
    idx = torch.tensor([2, 0, 2, 1, 0, 2], device=dev)
    y = dgl_amd.scatter_add(x, idx, 3)
    assert torch.allclose(y, torch.zeros(3, 4, device=dev).index_add_(0, idx, x.detach()))
    y.sum().backward()
    assert torch.equal(x.grad, torch.ones_like(x))


@pytest.mark.gpu
def test_errors(dev):
    from dgl_amd import _capi
    from dgl_amd._lib import DGLAMDError

    feat = torch.ones(4, 3, device=dev)
    off = torch.tensor([0, 2, 4], device=dev)
    out = torch.empty(2, 3, device=dev)
    with pytest.raises(DGLAMDError, match="Unsupported reduce function"):
        _capi.segment_reduce("prod", feat, off, out)
    with pytest.raises(DGLAMDError, match="arg is required"):
        _capi.segment_reduce("max", feat, off, out)
    with pytest.raises(DGLAMDError, match="different feature shapes"):
        _capi.segment_reduce("sum", feat, off, torch.empty(2, 4, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("tdtype,dim", [(torch.float32, 8), (torch.float32, 100), (torch.float64, 5), (torch.bfloat16, 16)])
def test_gpu_scatter_add_sorted_path(dev, idtype, tdtype, dim):
    """Above 2^20 elements scatter add groups the rows by target and sums them with the merge-path
    kernel (no atomics): exact for exactly representable inputs, keeps what `out` already held,
    leaves untouched targets alone, and is reproducible bit for bit."""
    from dgl_amd import _capi

    n = (1 << 20) // dim + 4097
    m = 3001
    g = torch.Generator().manual_seed(dim)
    idx = torch.randint(0, m, (n,), generator=g)
    idx[idx == 7] = 8                                   # an untouched target
    idx[: n // 4] = 11                                  # a hub target
    x = torch.randint(-4, 5, (n, dim), generator=g).to(tdtype)
    base = torch.randint(-2, 3, (m, dim), generator=g).to(tdtype)
    out = base.clone().to(dev)
    _capi.scatter_add(x.to(dev), idx.to(idtype).to(dev), out)
    want = base.double().clone().index_add_(0, idx, x.double())   # .double() aliases an fp64 base
    if tdtype == torch.bfloat16:
        # the hub sums exceed bf16's exact-integer range: compare after the same final rounding
        assert torch.equal(out.cpu(), want.to(torch.bfloat16))
    else:
        assert torch.equal(out.cpu().double(), want)
    assert torch.equal(out[7].cpu(), base[7])
    out2 = base.clone().to(dev)
    _capi.scatter_add(x.to(dev), idx.to(idtype).to(dev), out2)
    assert torch.equal(out.view(torch.uint8), out2.view(torch.uint8))
