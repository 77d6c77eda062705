"""Merge-path edge softmax kernels (balanced walk reduce, csrc/edge_softmax.hip) on the shapes
that stress their internals: units with more segments than one round of the LDS tables holds
(runs of 1-edge rows), empty rows, hub rows cut by many units, every feature width up to 16,
with and without an edge-id map — against the oracle (the reference's CPU kernel restated,
src/array/cpu/spmm.h:484-570) — and the accuracy of the hardware-exp based exponential."""
import numpy as np
import pytest
import torch

import oracle
from tests.tolerance import assert_fp32_sum

pytestmark = pytest.mark.gpu


def _graph(kind, rng):
    if kind == "ones":          # 5000 rows of exactly one edge: 128 segments per 256-item unit
        deg = np.ones(5000, dtype=np.int64)
    elif kind == "tiny":        # degrees 0..3: > 64 segments per unit, empty rows in between
        deg = rng.integers(0, 4, size=6000)
    elif kind == "hub":         # two hubs cut by many units among small rows
        deg = rng.integers(0, 12, size=3000)
        deg[17] = 9000
        deg[2500] = 3001
    elif kind.startswith("const"):  # every row the same length L: unit boundaries (960 items apart,
        L = int(kind[5:])           # moved forward by <= 63 edges to a row end) meet rows at every phase
        deg = np.full(max(20_000 // (L + 1), 40), L, dtype=np.int64)
    elif kind == "slack":       # row lengths 56..72 around the 63-edge slack, then rows around a unit's capacity
        deg = np.concatenate([rng.integers(56, 73, size=600), rng.integers(940, 1040, size=40),
                              rng.integers(0, 3, size=500), rng.integers(56, 73, size=300)])
    else:                       # lognormal-ish mix
        deg = np.minimum(rng.lognormal(2.0, 1.3, size=4000), 3000).astype(np.int64)
    indptr = np.zeros(deg.size + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    return indptr


@pytest.mark.parametrize("kind", ["ones", "tiny", "hub", "mix", "slack", "const62", "const63", "const64", "const65",
                                  "const958", "const959", "const960", "const1023", "const1024", "const1025",
                                  "const2500"])
@pytest.mark.parametrize("dim", [1, 3, 4, 8, 16])
@pytest.mark.parametrize("with_eids", [False, True])
def test_merge_softmax_against_oracle(dev, kind, dim, with_eids):
    from dgl_amd import _capi

    rng = np.random.default_rng(sum(map(ord, kind)) * 131 + dim)
    indptr = _graph(kind, rng)
    e, n = int(indptr[-1]), indptr.size - 1
    eids = rng.permutation(e).astype(np.int32) if with_eids else None
    score = (rng.standard_normal((e, dim)) * 3).astype(np.float32)
    ip = torch.from_numpy(indptr.astype(np.int32)).to(dev)
    ix = torch.zeros(e, dtype=torch.int32, device=dev)
    ei = None if eids is None else torch.from_numpy(eids).to(dev)
    csr = _capi.make_csr(ip, ix, ei, n)
    x = torch.from_numpy(score).to(dev)
    ws = torch.empty(_capi.edge_softmax_workspace_bytes(csr, x.dtype, dim), dtype=torch.uint8, device=dev)
    assert ws.numel() > 0
    out = torch.full_like(x, float("nan"))
    _capi.edge_softmax_forward(csr, x, out, ws)
    again = torch.full_like(x, float("nan"))
    _capi.edge_softmax_forward(csr, x, again, ws, plan_valid=True)
    assert torch.equal(out, again)                      # deterministic, plan re-used
    ref = oracle.edge_softmax_fwd(indptr.astype(np.int32), eids, score)
    # the reference adds a row's exponentials sequentially in fp32: on the 9000-edge hub its own
    # sum is ~1e-5 off; same rule as for the SpMM sums (tests/tolerance.py)
    exact = oracle.edge_softmax_fwd(indptr.astype(np.int32), eids, score.astype(np.float64))
    deg = np.diff(indptr)
    edge_row_len = np.empty(e, dtype=np.int64)       # softmax output is per edge: its row's length
    edge_row_len[np.arange(e) if eids is None else eids] = np.repeat(deg, deg)
    assert_fp32_sum(out.cpu().numpy(), ref, exact, rtol=1e-5, atol=1e-7, row_len=edge_row_len)
    sds = (rng.standard_normal((e, dim))).astype(np.float32) * ref
    back = torch.full_like(x, float("nan"))
    _capi.edge_softmax_backward(csr, out, torch.from_numpy(sds).to(dev), back, ws, plan_valid=True)
    ref_b = oracle.edge_softmax_bwd(indptr.astype(np.int32), eids, out.cpu().numpy(), sds)
    np.testing.assert_allclose(back.cpu().numpy(), ref_b, rtol=1e-4, atol=2e-6)
    # the scratch-free lane-group kernel agrees
    lg = torch.empty_like(x)
    _capi.edge_softmax_forward(csr, x, lg, None)
    np.testing.assert_allclose(lg.cpu().numpy(), out.cpu().numpy(), rtol=5e-6, atol=1e-8)


def test_hardware_exp_accuracy(dev):
    """One row whose scores sweep [-88, 0]: softmax = exp(x) / sum, against fp64."""
    from dgl_amd import _capi

    e = 200_000
    x64 = np.linspace(-88.0, 0.0, e)
    ip = torch.tensor([0, e], dtype=torch.int32, device=dev)
    csr = _capi.make_csr(ip, torch.zeros(e, dtype=torch.int32, device=dev), None, 1)
    x = torch.from_numpy(x64.astype(np.float32)).to(dev).reshape(e, 1)
    ws = torch.empty(_capi.edge_softmax_workspace_bytes(csr, x.dtype, 1), dtype=torch.uint8, device=dev)
    out = torch.empty_like(x)
    _capi.edge_softmax_forward(csr, x, out, ws)
    xe = x.double().cpu().numpy().ravel()
    want = np.exp(xe - xe.max())
    want /= want.sum()
    got = out.double().cpu().numpy().ravel()
    big = want > 1e-30                       # below: fp32 subnormal territory
    rel = np.abs(got[big] - want[big]) / want[big]
    assert rel.max() < 2e-6, rel.max()
