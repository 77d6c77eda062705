"""GPU parity tests of the graph-free C seam (dgla_*) against the CPU oracle.

Modelled on the reference's operator tests (tests/python/common/ops/test_ops.py:87-299):
same graphs (rand_graph(30,100), rand_bipartite(30,40,300)), the same broadcast shape pairs,
every op x reducer x idtype x dtype; plus the edge cases the merge-path kernel has to survive
(rows longer than a unit, isolated nodes, empty graphs, star graphs).

Bar: arg_u / arg_e and max/min values bit-exact.  fp32 sums: within 1e-5 relative of the
exact sum (the oracle run in fp64 on the same fp32 inputs) — the north-star tolerance — and
within 1e-5 + 2 * max_degree * 2^-24 of the oracle's own fp32 result, whose sequential
`out[k] += x` loop (src/array/cpu/spmm.h:60-70) carries up to max_degree * eps of rounding
error itself (a 3000-edge row differs from its exact sum by ~2e-5).  fp64 sums: 1e-12.
"""
import numpy as np
import pytest
import torch

import oracle
from tests.tolerance import assert_fp32_sum
from tests.graphgen import coo_to_csc, coo_to_csr, synth_csr

pytestmark = pytest.mark.gpu

SPMM_SHAPES = [  # tests/python/common/ops/test_ops.py:93-100
    ((1, 2, 1, 3, 1), (4, 1, 3, 1, 1)),
    ((5, 3, 1, 7), (1, 3, 7, 1)),
    ((1, 3, 1), (4, 1, 3)),
    ((3, 3), (1, 3)),
    ((1,), (3,)),
    ((3,), (1,)),
    ((1,), (1,)),
    ((), ()),
]
SDDMM_SHAPES = [  # test_ops.py:102-108
    ((1, 2, 1, 3, 1), (4, 1, 3, 1, 1)),
    ((5, 3, 1, 7), (1, 3, 7, 7)),
    ((1, 3, 3), (4, 1, 3)),
    ((3,), (3,)),
    ((1,), (1,)),
]


def rand_graph(n_src, n_dst, n_edges, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, n_src, n_edges), rng.integers(0, n_dst, n_edges)


GRAPHS = {
    "homo": lambda: (30, 30) + rand_graph(30, 30, 100, 1),          # dgl.rand_graph(30, 100)
    "bipartite": lambda: (30, 40) + rand_graph(30, 40, 300, 2),     # dgl.rand_bipartite(..,30,..,40,300)
}


def _tol(dtype):
    return dict(rtol=1e-5, atol=1e-6) if dtype == np.float32 else dict(rtol=1e-12, atol=1e-12)


def run_spmm(dev, op, reduce, n_src, n_dst, src, dst, ufeat, efeat, idtype, use_eids=True,
             accumulate_into=None):
    from dgl_amd import _capi

    indptr, indices, eids = coo_to_csc(src, dst, n_dst, idtype)
    if not use_eids:
        # positions as edge ids: permute efeat on the host so both sides see the same data
        if efeat is not None:
            efeat = efeat[eids]
        eids = None
    ref, ref_u, ref_e = oracle.spmm_csr(op, reduce, indptr, indices, eids, ufeat, efeat)
    exact = None
    if reduce == "sum" and ref.dtype == np.float32:
        f64 = lambda a: None if a is None else a.astype(np.float64)
        exact = oracle.spmm_csr(op, reduce, indptr, indices, eids, f64(ufeat), f64(efeat))[0]
    maxdeg = np.diff(indptr)   # per-row edge counts (the plain 1e-5 bar applies to rows under 1000 edges)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    keep = (t(indptr), t(indices), t(eids))
    csr = _capi.make_csr(keep[0], keep[1], keep[2], n_src)
    tu, te = t(ufeat), t(efeat)
    out = torch.full(ref.shape, 7.0, dtype=(tu if tu is not None else te).dtype, device=dev)
    if accumulate_into is not None:
        out = t(accumulate_into.astype(ref.dtype))
    tidt = torch.int32 if idtype == np.int32 else torch.int64
    arg_u = torch.full(ref.shape, -5, dtype=tidt, device=dev) if reduce != "sum" else None
    arg_e = torch.full(ref.shape, -5, dtype=tidt, device=dev) if reduce != "sum" else None
    nbytes = _capi.spmm_csr_workspace_bytes(op, reduce, csr, out.dtype, tu, te, out)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    _capi.spmm_csr(op, reduce, csr, tu, te, out, arg_u, arg_e, ws,
                   accumulate=accumulate_into is not None)
    # second call re-using the cached plan must give identical bits
    out2 = out.clone() if accumulate_into is None else t(accumulate_into.astype(ref.dtype))
    _capi.spmm_csr(op, reduce, csr, tu, te, out2, arg_u, arg_e, ws,
                   accumulate=accumulate_into is not None, plan_valid=True)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    return out.cpu().numpy(), (None if arg_u is None else arg_u.cpu().numpy()), \
        (None if arg_e is None else arg_e.cpu().numpy()), ref, ref_u, ref_e, exact, maxdeg


def check_spmm(res, reduce, dtype):
    out, au, ae, ref, ref_u, ref_e, exact, maxdeg = res
    if reduce == "sum":
        if exact is not None:
            # flat 1e-5 against the exact sum; against the reference's sequential fp32 value
            # 1e-5 or "closer to exact than the reference is" (tests/tolerance.py)
            assert_fp32_sum(out, ref, exact, row_len=maxdeg if out.shape[0] == len(maxdeg) else None)
        else:
            np.testing.assert_allclose(out, ref, **_tol(dtype))
    else:
        # max/min pick one of the candidates: must be the same bits, and the same edge
        np.testing.assert_array_equal(out, ref)
        if ref_u is not None:
            np.testing.assert_array_equal(au, ref_u)
        if ref_e is not None:
            np.testing.assert_array_equal(ae, ref_e)


@pytest.mark.parametrize("gname", list(GRAPHS))
@pytest.mark.parametrize("shp", SPMM_SHAPES)
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "copy_lhs", "copy_rhs"])
@pytest.mark.parametrize("reduce", ["sum", "min", "max"])
@pytest.mark.parametrize("idtype", [np.int32, np.int64])
def test_spmm_matrix(dev, gname, shp, op, reduce, idtype):
    n_src, n_dst, src, dst = GRAPHS[gname]()
    rng = np.random.default_rng(12345)
    dtype = np.float32
    u = (rng.random((n_src,) + shp[0]) + 1).astype(dtype)
    e = (rng.random((len(src),) + shp[1]) + 1).astype(dtype)
    res = run_spmm(dev, op, reduce, n_src, n_dst, src, dst,
                   u if op != "copy_rhs" else None, e if op != "copy_lhs" else None, idtype)
    check_spmm(res, reduce, dtype)


@pytest.mark.parametrize("op", ["mul", "copy_lhs", "copy_rhs", "add"])
@pytest.mark.parametrize("reduce", ["sum", "max", "min"])
@pytest.mark.parametrize("feat", [(100,), (8, 16), (4,), (1,), (33,), (300,), (260,)])
def test_spmm_f64_and_widths(dev, op, reduce, feat):
    """fp64 and feature widths that hit every access path: 16-byte lanes (100, 128, 4),
    element-wise lanes (1, 33), more than one 64-lane chunk (300, 260)."""
    n_src, n_dst, src, dst = GRAPHS["bipartite"]()
    rng = np.random.default_rng(7)
    for dtype in (np.float32, np.float64):
        u = (rng.random((n_src,) + feat) + 1).astype(dtype)
        e = (rng.random((len(src),) + feat) + 1).astype(dtype)
        res = run_spmm(dev, op, reduce, n_src, n_dst, src, dst,
                       u if op != "copy_rhs" else None, e if op != "copy_lhs" else None, np.int32)
        check_spmm(res, reduce, dtype)


@pytest.mark.parametrize("reduce", ["sum", "max", "min"])
@pytest.mark.parametrize("use_eids", [True, False])
def test_spmm_gat_broadcast(dev, reduce, use_eids):
    """u_mul_e with (N,H,D) x (E,H,1) — the GATConv pattern (gatconv.py:346), and scalar e."""
    n_src, n_dst, src, dst = GRAPHS["homo"]()
    rng = np.random.default_rng(3)
    for ushape, eshape in (((8, 16), (8, 1)), ((64,), (1,)), ((4, 8), (1, 1))):
        u = (rng.random((n_src,) + ushape) + 1).astype(np.float32)
        e = (rng.random((len(src),) + eshape) + 1).astype(np.float32)
        res = run_spmm(dev, "mul", reduce, n_src, n_dst, src, dst, u, e, np.int32, use_eids)
        check_spmm(res, reduce, np.float32)


def _edge_case_graphs():
    rng = np.random.default_rng(99)
    cases = {}
    # star: 900 leaves -> node 0 (tests/python/common/ops/test_ops.py:193-216)
    cases["star900"] = (901, 901, np.arange(1, 901), np.zeros(900, dtype=np.int64))
    # one row far longer than a 512-item unit, plus isolated nodes before and after it
    src = rng.integers(0, 50, 3000)
    dst = np.full(3000, 20)
    cases["long_row"] = (50, 60, src, dst)
    # many isolated destination nodes (more row-end items than edges), edges at the end
    cases["isolated"] = (10, 2000, rng.integers(0, 10, 40), rng.integers(1990, 2000, 40))
    # mixture: a few heavy rows among light ones, unit boundaries fall inside rows
    dst = np.concatenate([rng.integers(0, 300, 2000), np.full(700, 7), np.full(1300, 150)])
    cases["mixed"] = (200, 300, rng.integers(0, 200, len(dst)), dst)
    # exact multiples of the unit size
    cases["exact512"] = (16, 256, rng.integers(0, 16, 256), np.arange(256))
    cases["no_edges"] = (5, 7, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    cases["single"] = (1, 1, np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64))
    return cases


@pytest.mark.parametrize("name", list(_edge_case_graphs()))
@pytest.mark.parametrize("reduce", ["sum", "max", "min"])
@pytest.mark.parametrize("op", ["copy_lhs", "mul"])
@pytest.mark.parametrize("feat", [100, 8, 1])
def test_spmm_edge_cases(dev, name, reduce, op, feat):
    n_src, n_dst, src, dst = _edge_case_graphs()[name]
    rng = np.random.default_rng(5)
    u = (rng.random((n_src, feat)) + 1).astype(np.float32)
    e = (rng.random((len(src), feat)) + 1).astype(np.float32)
    if len(src) == 0:
        e = np.zeros((0, feat), np.float32)
    res = run_spmm(dev, op, reduce, n_src, n_dst, src, dst, u, e if op == "mul" else None,
                   np.int64)
    check_spmm(res, reduce, np.float32)


def test_spmm_ties_pick_first_position(dev):
    """All candidates equal: arg must be the first edge in CSR order (strict compare,
    src/array/cpu/spmm_binary_ops.h:121,137), also across unit / lane-group boundaries."""
    n_src, n_dst = 40, 3
    src = np.tile(np.arange(40), 50)          # 2000 edges into 3 rows, long rows
    dst = np.repeat(np.arange(3), [1200, 1, 799])
    u = np.ones((n_src, 100), np.float32)
    for reduce in ("max", "min"):
        res = run_spmm(dev, "copy_lhs", reduce, n_src, n_dst, src, dst, u, None, np.int32)
        check_spmm(res, reduce, np.float32)


def test_spmm_nan_never_wins(dev):
    n_src, n_dst, src, dst = GRAPHS["homo"]()
    rng = np.random.default_rng(8)
    u = (rng.random((n_src, 12)) + 1).astype(np.float32)
    u[::3, ::2] = np.nan
    for reduce in ("max", "min"):
        res = run_spmm(dev, "copy_lhs", reduce, n_src, n_dst, src, dst, u, None, np.int32)
        check_spmm(res, reduce, np.float32)


def test_spmm_accumulate_flag(dev):
    """DGLA_ACCUMULATE: out += result (the reference's contract, spmm.cuh:528-534)."""
    n_src, n_dst, src, dst = _edge_case_graphs()["mixed"]
    rng = np.random.default_rng(11)
    u = (rng.random((n_src, 100)) + 1).astype(np.float32)
    base = rng.random((n_dst, 100)).astype(np.float32)
    out, _, _, ref, _, _, _, _ = run_spmm(dev, "copy_lhs", "sum", n_src, n_dst, src, dst, u,
                                          None, np.int32, accumulate_into=base)
    np.testing.assert_allclose(out, ref + base, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tdtype,rtol,atol", [(torch.float16, 1e-3, 0.5), (torch.bfloat16, 4e-3, 2.0)])
def test_spmm_half(dev, tdtype, rtol, atol):
    """test_ops.py:184-223: star graph 900 -> 1, feat 32, fp16 / bf16 inputs, fp32 accumulate."""
    from dgl_amd import _capi

    n = 901
    src, dst = np.arange(1, 901), np.zeros(900, dtype=np.int64)
    indptr, indices, eids = coo_to_csc(src, dst, n, np.int32)
    rng = np.random.default_rng(4)
    u32 = (rng.random((n, 32)) + 1).astype(np.float32)
    e32 = (rng.random((900, 32)) + 1).astype(np.float32)
    tu = torch.from_numpy(u32).to(dev).to(tdtype)
    te = torch.from_numpy(e32).to(dev).to(tdtype)
    ti = [torch.from_numpy(a).to(dev) for a in (indptr, indices, eids)]
    csr = _capi.make_csr(ti[0], ti[1], ti[2], n)
    for op, uu, ee in (("copy_lhs", tu, None), ("mul", tu, te)):
        out = torch.empty((n, 32), dtype=tdtype, device=dev)
        ws = torch.empty(max(_capi.spmm_csr_workspace_bytes(op, "sum", csr, tdtype, uu, ee, out), 1),
                         dtype=torch.uint8, device=dev)
        _capi.spmm_csr(op, "sum", csr, uu, ee, out, None, None, ws)
        # oracle on the rounded inputs, accumulating in fp32/fp64
        ref, _, _ = oracle.spmm_csr(op, "sum", indptr, indices, eids,
                                    tu.float().cpu().numpy().astype(np.float64),
                                    None if ee is None else te.float().cpu().numpy().astype(np.float64))
        np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=rtol, atol=atol)


# --------------------------------------------------------------------------------------
# SDDMM
# --------------------------------------------------------------------------------------
def run_sddmm(dev, fmt, op, n_src, n_dst, src, dst, lhs, rhs, lt, rt, idtype, dtype):
    from dgl_amd import _capi

    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if fmt == "coo":
        row, col = src.astype(idtype), dst.astype(idtype)
        ref = oracle.sddmm_coo(op, row, col, None, lhs, rhs, lt, rt)
        keep = (t(row), t(col))
        sp = _capi.make_coo(keep[0], keep[1], None, n_src, n_dst)
    else:
        indptr, indices, eids = coo_to_csr(src, dst, n_src, idtype)
        ref = oracle.sddmm_csr(op, indptr, indices, eids, lhs, rhs, lt, rt)
        keep = (t(indptr), t(indices), t(eids))
        sp = _capi.make_csr(keep[0], keep[1], keep[2], n_dst)
    tl, tr = t(lhs), t(rhs)
    out = torch.full(ref.shape, -3.0, dtype=torch.float32 if dtype == np.float32 else torch.float64,
                     device=dev)
    fn = _capi.sddmm_coo if fmt == "coo" else _capi.sddmm_csr
    fn(op, sp, tl, tr, out, _capi.TARGETS[lt], _capi.TARGETS[rt])
    torch.cuda.synchronize()
    return out.cpu().numpy(), ref


@pytest.mark.parametrize("gname", list(GRAPHS))
@pytest.mark.parametrize("fmt", ["coo", "csr"])
@pytest.mark.parametrize("shp", SDDMM_SHAPES)
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "dot", "copy_lhs", "copy_rhs"])
@pytest.mark.parametrize("lt", ["u", "v", "e"])
@pytest.mark.parametrize("rt", ["u", "v", "e"])
def test_sddmm_matrix(dev, gname, fmt, shp, op, lt, rt):
    n_src, n_dst, src, dst = GRAPHS[gname]()
    rng = np.random.default_rng(12345)
    n = {"u": n_src, "v": n_dst, "e": len(src)}
    dtype = np.float32
    lhs = (rng.random((n[lt],) + shp[0]) + 1).astype(dtype)
    rhs = (rng.random((n[rt],) + shp[1]) + 1).astype(dtype)
    if op == "dot" and shp[0][-1:] != shp[1][-1:]:
        pytest.skip("dot needs equal last dims")
    out, ref = run_sddmm(dev, fmt, op, n_src, n_dst, src, dst,
                         lhs if op != "copy_rhs" else None, rhs if op != "copy_lhs" else None,
                         lt, rt, np.int32, dtype)
    if op == "dot":
        np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)
    else:
        np.testing.assert_array_equal(out, ref)  # one rounding per element: bit-exact


@pytest.mark.parametrize("hd", [(8, 32), (8, 8), (1, 128), (4, 64), (3, 20), (2, 256), (16, 4)])
@pytest.mark.parametrize("idtype", [np.int32, np.int64])
def test_sddmm_dot_heads(dev, hd, idtype):
    """u_dot_v with (N,H,D): fast shuffle path (D/4 power of two) and general path."""
    n_src, n_dst, src, dst = GRAPHS["bipartite"]()
    rng = np.random.default_rng(2)
    for dtype in (np.float32, np.float64):
        lhs = (rng.random((n_src,) + hd) - 0.5).astype(dtype)
        rhs = (rng.random((n_dst,) + hd) - 0.5).astype(dtype)
        for fmt in ("coo", "csr"):
            out, ref = run_sddmm(dev, fmt, "dot", n_src, n_dst, src, dst, lhs, rhs, "u", "v",
                                 idtype, dtype)
            np.testing.assert_allclose(out, ref, rtol=1e-5 if dtype == np.float32 else 1e-12,
                                       atol=1e-6 if dtype == np.float32 else 1e-12)


# --------------------------------------------------------------------------------------
# COO SpMM, edge softmax, errors
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("op", ["copy_lhs", "mul", "add", "copy_rhs"])
@pytest.mark.parametrize("reduce", ["sum", "max", "min"])
def test_spmm_coo(dev, op, reduce):
    from dgl_amd import _capi

    n_src, n_dst, src, dst = GRAPHS["bipartite"]()
    rng = np.random.default_rng(6)
    for shp in (((5,), (5,)), ((3, 4), (3, 1))):
        u = (rng.random((n_src,) + shp[0]) + 1).astype(np.float32)
        e = (rng.random((len(src),) + shp[1]) + 1).astype(np.float32)
        uu = u if op != "copy_rhs" else None
        ee = e if op != "copy_lhs" else None
        row, col = src.astype(np.int32), dst.astype(np.int32)
        ref, ru, re_ = oracle.spmm_coo(op, reduce, row, col, None, n_dst, uu, ee)
        t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
        keep = (t(row), t(col))
        coo = _capi.make_coo(keep[0], keep[1], None, n_src, n_dst)
        out = torch.full(ref.shape, 9.0, device=dev)
        au = torch.full(ref.shape, -1, dtype=torch.int32, device=dev) if reduce != "sum" else None
        ae = torch.full(ref.shape, -1, dtype=torch.int32, device=dev) if reduce != "sum" else None
        _capi.spmm_coo(op, reduce, coo, t(uu), t(ee), out, au, ae)
        if reduce == "sum":  # atomics: order differs, tolerance only
            np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
        else:
            np.testing.assert_array_equal(out.cpu().numpy(), ref)
            if ru is not None:
                np.testing.assert_array_equal(au.cpu().numpy(), ru)
            if re_ is not None:
                np.testing.assert_array_equal(ae.cpu().numpy(), re_)


def test_spmm_coo_deterministic_flag(dev, monkeypatch):
    """USE_DETERMINISTIC_ALG (src/array/cuda/spmm.cu:33-35): under the flag a COO sum must give
    the same bits run after run — and, being a position-ordered sum like the CSR kernel's, the
    bits of the CSC path over the same edges."""
    from dgl_amd import _capi

    rng = np.random.default_rng(8)
    n_src, n_dst, e, f = 500, 40, 60_000, 33          # ~1500 edges per destination: heavy atomic contention
    src = rng.integers(0, n_src, e).astype(np.int32)
    dst = rng.integers(0, n_dst, e).astype(np.int32)
    x = torch.from_numpy((rng.random((n_src, f)) * 100).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.random((e, 1)).astype(np.float32)).to(dev)
    row, col = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
    coo = _capi.make_coo(row, col, None, n_src, n_dst)
    monkeypatch.setenv("USE_DETERMINISTIC_ALG", "1")
    outs = []
    for _ in range(4):
        o = torch.empty(n_dst, f, device=dev)
        _capi.spmm_coo("mul", "sum", coo, x, w, o)
        outs.append(o)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    indptr, indices, eids = _capi.coo_to_csr(col, row, None, n_dst)
    csr = _capi.make_csr(indptr, indices, eids, n_src)
    ref = torch.empty(n_dst, f, device=dev)
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("mul", "sum", csr, x.dtype, x, w, ref), dtype=torch.uint8, device=dev)
    _capi.spmm_csr("mul", "sum", csr, x, w, ref, None, None, ws)
    assert torch.equal(outs[0], ref)
    monkeypatch.delenv("USE_DETERMINISTIC_ALG")
    o = torch.empty(n_dst, f, device=dev)
    _capi.spmm_coo("mul", "sum", coo, x, w, o)              # the atomic route: same value to 1e-5
    torch.testing.assert_close(o, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("heads", [1, 4, 8])
@pytest.mark.parametrize("merge", [False, True])
def test_edge_softmax_fused(dev, heads, merge):
    from dgl_amd import _capi

    n_src, n_dst, src, dst = _edge_case_graphs()["mixed"]
    indptr, indices, eids = coo_to_csc(src, dst, n_dst, np.int32)
    rng = np.random.default_rng(1)
    score = rng.standard_normal((len(src), heads, 1)).astype(np.float32) * 3
    grad = rng.standard_normal((len(src), heads, 1)).astype(np.float32)
    ref = oracle.edge_softmax_fwd(indptr, eids, score)
    ref_b = oracle.edge_softmax_bwd(indptr, eids, ref, ref * grad)
    t = lambda a: torch.from_numpy(a).to(dev)
    keep = (t(indptr), t(indices), t(eids))
    csr = _capi.make_csr(keep[0], keep[1], keep[2], n_src)
    out = torch.empty_like(t(score))
    ws = torch.empty(_capi.edge_softmax_workspace_bytes(csr, out.dtype, heads), dtype=torch.uint8,
                     device=dev) if merge else None
    _capi.edge_softmax_forward(csr, t(score), out, ws)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-7)
    back = torch.empty_like(out)
    _capi.edge_softmax_backward(csr, t(ref), t(ref * grad), back, ws, plan_valid=merge)
    np.testing.assert_allclose(back.cpu().numpy(), ref_b, rtol=1e-5, atol=1e-6)


def test_errors_are_reported(dev):
    from dgl_amd import DGLAMDError, _capi

    n_src, n_dst, src, dst = GRAPHS["homo"]()
    indptr, indices, eids = coo_to_csc(src, dst, n_dst, np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    keep = (t(indptr), t(indices), t(eids))
    csr = _capi.make_csr(keep[0], keep[1], keep[2], n_src)
    u = torch.ones((n_src, 4), device=dev)
    out = torch.empty((n_dst, 4), device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    with pytest.raises(DGLAMDError, match="Unsupported SpMM binary operator"):
        _capi.spmm_csr("pow", "sum", csr, u, None, out, None, None, ws)
    with pytest.raises(DGLAMDError, match="Unsupported SpMM reducer"):
        _capi.spmm_csr("copy_lhs", "prod", csr, u, None, out, None, None, ws)
    with pytest.raises(DGLAMDError, match="first dimension"):
        _capi.spmm_csr("copy_lhs", "sum", csr, u[:-1].contiguous(), None, out, None, None, ws)
    with pytest.raises(DGLAMDError, match="workspace"):
        _capi.spmm_csr("copy_lhs", "sum", csr, u, None, out, None, None, ws[:8])
    with pytest.raises(DGLAMDError, match="arg_u is required"):
        _capi.spmm_csr("copy_lhs", "max", csr, u, None, out, None, None, ws)


def test_scaled_c2_properties(dev):
    """Scaled-down headline config (N/16, E/16, F=100) at full kernel geometry: result vs the
    oracle, plus size-independent properties used at full size by bench.py: linearity
    (A(x+y) = Ax + Ay within fp32 tolerance) and the column-sum identity
    sum_r out[r] == sum_c outdeg[c] * x[c]."""
    from dgl_amd import _capi
    from tests.graphgen import C2_EDGES, C2_NODES

    n, e, f = C2_NODES // 16, C2_EDGES // 16, 100
    g = synth_csr(n, n, e, "U", device=dev)
    torch.manual_seed(12345)
    x = torch.rand(n, f, device=dev) + 1
    csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
    out = torch.empty(n, f, device=dev)
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out),
                     dtype=torch.uint8, device=dev)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws)
    ref = oracle.copy_u_sum_csr(g["indptr"].cpu().numpy(), g["indices"].cpu().numpy(), x.cpu().numpy())
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    # max with args, bit exact
    au = torch.empty(n, f, dtype=torch.int32, device=dev)
    ae = torch.empty(n, f, dtype=torch.int32, device=dev)
    outm = torch.empty(n, f, device=dev)
    wsm = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "max", csr, x.dtype, x, None, outm),
                      dtype=torch.uint8, device=dev)
    _capi.spmm_csr("copy_lhs", "max", csr, x, None, outm, au, ae, wsm)
    r, ru, _ = oracle.spmm_csr("copy_lhs", "max", g["indptr"].cpu().numpy(),
                               g["indices"].cpu().numpy(), None, x.cpu().numpy(), None)
    np.testing.assert_array_equal(outm.cpu().numpy(), r)
    np.testing.assert_array_equal(au.cpu().numpy(), ru)
    # column-sum identity in fp64
    outdeg = torch.bincount(g["indices"].long(), minlength=n).double()
    lhs = out.double().sum(0)
    rhs = (outdeg[:, None] * x.double()).sum(0)
    assert torch.allclose(lhs, rhs, rtol=1e-6)


# --------------------------------------------------------------------------------------
# Access-width paths added after the first profile: 8-byte lanes (bf16/fp16 rows of
# F % 4 == 0, fp32 rows of even length), generalised dot lanes, edge-softmax lane groups
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("tdtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("feat", [100, 36, 6, 10, 260])
@pytest.mark.parametrize("op,reduce", [("copy_lhs", "sum"), ("mul", "sum"), ("copy_lhs", "max"),
                                       ("add", "min")])
def test_spmm_8byte_lanes(dev, tdtype, feat, op, reduce):
    """F=100 in bf16 is a 200-byte row (8-byte aligned only); F=6/10 fp32 rows are 8-byte
    aligned.  Values: sums against the oracle in fp64 on the rounded inputs; max/min pick an
    input (or one rounded op result) so they must match the oracle run in the same storage
    precision exactly, and so must arg_u / arg_e."""
    from dgl_amd import _capi

    n_src, n_dst, src, dst = GRAPHS["bipartite"]()
    indptr, indices, eids = coo_to_csc(src, dst, n_dst, np.int32)
    rng = np.random.default_rng(feat)
    tu = torch.from_numpy((rng.random((n_src, feat)) + 1).astype(np.float32)).to(dev).to(tdtype)
    te = torch.from_numpy((rng.random((len(src), feat)) + 1).astype(np.float32)).to(dev).to(tdtype)
    uu = tu if op != "copy_rhs" else None
    ee = te if op != "copy_lhs" else None
    ti = [torch.from_numpy(a).to(dev) for a in (indptr, indices, eids)]
    csr = _capi.make_csr(ti[0], ti[1], ti[2], n_src)
    out = torch.empty((n_dst, feat), dtype=tdtype, device=dev)
    au = torch.full((n_dst, feat), -3, dtype=torch.int32, device=dev) if reduce != "sum" else None
    ae = torch.full((n_dst, feat), -3, dtype=torch.int32, device=dev) if reduce != "sum" else None
    ws = torch.empty(max(_capi.spmm_csr_workspace_bytes(op, reduce, csr, tdtype, uu, ee, out), 1),
                     dtype=torch.uint8, device=dev)
    _capi.spmm_csr(op, reduce, csr, uu, ee, out, au, ae, ws)
    f64 = lambda t: None if t is None else t.float().cpu().numpy().astype(np.float64)
    ref, ru, re_ = oracle.spmm_csr(op, reduce, indptr, indices, eids, f64(uu), f64(ee))
    got = out.float().cpu().numpy()
    if reduce == "sum":
        tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1e-2}[tdtype]
        np.testing.assert_allclose(got, ref, rtol=tol, atol=tol)
    else:
        # round the fp64 op result to storage precision the way the kernel does
        want = torch.from_numpy(ref).to(tdtype).float().numpy()
        if op == "copy_lhs":
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(au.cpu().numpy(), ru)
        else:
            np.testing.assert_allclose(got, want, rtol={torch.float32: 1e-6}.get(tdtype, 1e-2))


@pytest.mark.parametrize("hd", [(1, 100), (2, 50), (1, 7), (3, 33), (1, 300), (5, 12)])
def test_sddmm_dot_any_width(dev, hd):
    """dot over D that is neither a power of two nor a multiple of the vector width."""
    n_src, n_dst, src, dst = GRAPHS["homo"]()
    rng = np.random.default_rng(3)
    for dtype in (np.float32, np.float64):
        lhs = (rng.random((n_src,) + hd) - 0.5).astype(dtype)
        rhs = (rng.random((n_dst,) + hd) - 0.5).astype(dtype)
        for fmt in ("coo", "csr"):
            out, ref = run_sddmm(dev, fmt, "dot", n_src, n_dst, src, dst, lhs, rhs, "u", "v",
                                 np.int32, dtype)
            np.testing.assert_allclose(out, ref, rtol=1e-5 if dtype == np.float32 else 1e-12,
                                       atol=1e-6 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("n_dst,n_edges,dim", [
    (2000, 3000, 1),      # mean degree 1-2: 64 rows per wave
    (500, 12000, 8),      # mean degree 24: 8 heads x 8 edge slots = one row per wave
    (40, 30000, 4),       # rows of ~750 edges: far beyond the register cache
    (300, 9000, 70),      # dim > 64: feature loop
    (1000, 16000, 3),     # non power-of-two dim: idle feature lanes
    (3, 20000, 8),        # three hub rows spanning ~25 merge units each
    (5000, 200, 2),       # almost all rows empty
    (700, 40000, 16),     # widest merge-path case
])
@pytest.mark.parametrize("idtype", [np.int32, np.int64])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("merge", [False, True])
def test_edge_softmax_geometries(dev, n_dst, n_edges, dim, idtype, dtype, merge):
    """merge=True hands the kernels a workspace: the degree-balanced merge-path pair runs
    (for dim <= 16; fp64: <= 8), otherwise the scratch-free lane-group kernel."""
    from dgl_amd import _capi

    rng = np.random.default_rng(n_edges)
    src = rng.integers(0, 50, n_edges)
    dst = np.minimum((rng.random(n_edges) ** 2 * n_dst).astype(np.int64), n_dst - 1)  # skewed
    indptr, indices, eids = coo_to_csc(src, dst, n_dst, idtype)
    score = (rng.standard_normal((n_edges, dim)) * 4).astype(dtype)
    grad = rng.standard_normal((n_edges, dim)).astype(dtype)
    # exact reference: the oracle in fp64 on the same inputs (the reference's own fp32 loop
    # carries ~degree * 2^-24 of rounding in its sequential sums; rows here reach 7k edges)
    ref = oracle.edge_softmax_fwd(indptr, eids, score.astype(np.float64)).astype(dtype)
    ref_b = oracle.edge_softmax_bwd(indptr, eids, ref.astype(np.float64),
                                    ref.astype(np.float64) * grad).astype(dtype)
    t = lambda a: torch.from_numpy(a).to(dev)
    keep = (t(indptr), t(indices), t(eids))
    csr = _capi.make_csr(keep[0], keep[1], keep[2], 50)
    ts = t(score)
    ws = None
    if merge:
        nbytes = _capi.edge_softmax_workspace_bytes(csr, ts.dtype, dim)
        assert (nbytes > 0) == (dim <= (8 if dtype == np.float64 else 16))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    out = torch.full_like(ts, 7.0)
    _capi.edge_softmax_forward(csr, ts, out, ws)
    tol = dict(rtol=2e-5, atol=1e-7) if dtype == np.float32 else dict(rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(out.cpu().numpy(), ref, **tol)
    if merge:  # cached plan: identical bits
        out2 = torch.full_like(ts, 5.0)
        _capi.edge_softmax_forward(csr, ts, out2, ws, plan_valid=True)
        assert torch.equal(out, out2)
    back = torch.full_like(ts, 7.0)
    _capi.edge_softmax_backward(csr, t(ref), t(ref * grad), back, ws, plan_valid=merge)
    np.testing.assert_allclose(back.cpu().numpy(), ref_b, rtol=tol["rtol"] * 10, atol=1e-6 if dtype == np.float32 else 1e-13)
    # softmax rows sum to one (size-independent property)
    rows = np.repeat(np.arange(n_dst), np.diff(indptr))
    sums = np.zeros((n_dst, dim))
    np.add.at(sums, rows, out.cpu().numpy()[eids].astype(np.float64))
    nz = np.diff(indptr) > 0
    np.testing.assert_allclose(sums[nz], 1.0, rtol=1e-5)


# --------------------------------------------------------------------------------------
# Tuning bits (dgla_set_tuning): XCD-contiguous order, non-temporal streams, split-row layout.
# They only steer the memory system: every setting must give the SAME BITS as flags = 0.
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("feat,tdtype", [(100, torch.float32), (36, torch.float32),
                                         (100, torch.bfloat16), (200, torch.float16),
                                         (50, torch.float64),
                                         # edge layout of the split rows: 128 k + t bytes, k >= 2
                                         (68, torch.float32), (76, torch.float32), (132, torch.float32),
                                         (252, torch.float32), (136, torch.bfloat16), (38, torch.float64),
                                         # straddle layout: 8-byte-aligned rows gathered with 8-byte lanes
                                         (50, torch.float32), (25, torch.float64), (84, torch.bfloat16),
                                         (300, torch.float16)])
@pytest.mark.parametrize("op,reduce", [("copy_lhs", "sum"), ("mul", "sum"), ("copy_lhs", "max")])
def test_spmm_tuning_bits_do_not_change_results(dev, feat, tdtype, op, reduce):
    from dgl_amd import _capi

    # large enough that the split-row layout is eligible for 400-byte rows
    # (N_src * row bytes >= 64 MiB, E >= 4 N_src); columns != rows to catch mix-ups
    n_dst, n_src, e = 50_000, 180_000, 800_000
    g = synth_csr(n_dst, n_src, e, "U", seed=77, device=dev, with_eids=True)
    torch.manual_seed(5)
    x = (torch.rand(n_src, feat, device=dev) + 1).to(tdtype)
    w = (torch.rand(e, 1, device=dev) + 1).to(tdtype) if op == "mul" else None
    csr = _capi.make_csr(g["indptr"], g["indices"], g["eids"], n_src)
    results = {}
    default = _capi.get_tuning()
    try:
        # the full cross product of the layout / memory-system bits: XCD, SPLIT, SPLIT_FORCE, NO_STAGE_W (scalar edge weights
        # staged in LDS or read per gather batch: only the `mul` cases differ)
        for flags in [a | d | f | sw for a in (0, 1) for d in (0, 8) for f in (0, 64) for sw in ((0, 8192) if op == "mul" else (0,))]:
            _capi.set_tuning(flags)
            assert _capi.get_tuning() == flags
            out = torch.full((n_dst, feat), -3.0, dtype=tdtype, device=dev)
            au = torch.full((n_dst, feat), -9, dtype=torch.int32, device=dev) if reduce != "sum" else None
            ae = None
            ws = torch.empty(_capi.spmm_csr_workspace_bytes(op, reduce, csr, x.dtype, x, w, out),
                             dtype=torch.uint8, device=dev)
            _capi.spmm_csr(op, reduce, csr, x, w, out, au, ae, ws)
            # second call with the cached plan (and, for split, a fresh re-layout of X)
            _capi.spmm_csr(op, reduce, csr, x, w, out, au, ae, ws, plan_valid=True)
            torch.cuda.synchronize()
            results[flags] = (out.clone(), None if au is None else au.clone(), ws.numel())
    finally:
        _capi.set_tuning(default)
    base = results[0]
    for flags, (o, a, _) in results.items():
        assert torch.equal(o.view(torch.uint8), base[0].view(torch.uint8)), "flags=%d" % flags
        if a is not None:
            assert torch.equal(a, base[1]), "flags=%d" % flags
    row_bytes = feat * x.element_size()
    if row_bytes % 128 and row_bytes >= 128 and row_bytes % 16 == 0 and row_bytes * n_src >= 64 << 20 \
            and row_bytes <= 1024:
        # edge layout (rows of two or more whole lines): one side line + the dense tail per row;
        # shorter rows are gathered in place
        side = 128 + row_bytes % 128 if row_bytes >= 256 else 0
        assert results[0][2] + n_src * side <= results[8][2] < results[0][2] + n_src * side + (1 << 20)
    # and the shared result is the right one
    host = [t.cpu().numpy() for t in (g["indptr"], g["indices"], g["eids"])]
    if tdtype in (torch.float32, torch.float64):
        ref, ru, _ = oracle.spmm_csr(op, reduce, host[0], host[1], host[2], x.cpu().numpy(),
                                     None if w is None else w.cpu().numpy())
        if reduce == "sum":
            np.testing.assert_allclose(base[0].cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
        else:
            np.testing.assert_array_equal(base[0].cpu().numpy(), ref)
            np.testing.assert_array_equal(base[1].cpu().numpy(), ru)


@pytest.mark.parametrize("n_dst", [2_000, 20_000])  # rows of 200 edges (the hint applies) / 20 (it does not)
@pytest.mark.parametrize("feat,tdtype", [(100, torch.float32), (1, torch.float32), (50, torch.bfloat16),
                                         (25, torch.float64), (7, torch.float16)])
def test_position_ordered_operand_nontemporal_loads_same_bits(dev, feat, tdtype, n_dst):
    """copy_rhs over long rows WITHOUT an edge-id map (a readout-like segment reduce) loads the rows
    non-temporally (spmm_nt_stream(), csrc/spmm_csr.hip.h) — a cache-policy hint, so the bits must equal
    those of the default-load kernel, reached here through an IDENTITY edge-id map (no reference
    counterpart)."""
    from dgl_amd import _capi

    n_src, e = 30_000, 400_000
    g = synth_csr(n_dst, n_src, e, "U", seed=78, device=dev, with_eids=False)
    csr = _capi.make_csr(g["indptr"], g["indices"], None, n_src)
    torch.manual_seed(6)
    w = (torch.rand(e, feat, device=dev) + 1).to(tdtype)
    x = (torch.rand(n_src, feat, device=dev) + 1).to(tdtype)
    off = g["indptr"].to(torch.int64)
    ident = torch.arange(e, dtype=g["indices"].dtype, device=dev)
    got = {}
    if True:
        for on in (0, 1):
            csr = _capi.make_csr(g["indptr"], g["indices"], None if on else ident, n_src)
            seg = torch.full((n_dst, feat), -3.0, dtype=tdtype, device=dev)
            _capi.segment_reduce("sum", w, off, seg)
            smax = torch.full((n_dst, feat), -3.0, dtype=tdtype, device=dev)
            amax = torch.full((n_dst, feat), -7, dtype=torch.int64, device=dev)
            _capi.segment_reduce("max", w, off, smax, amax)
            mul = torch.full((n_dst, feat), -3.0, dtype=tdtype, device=dev)
            ws = torch.empty(_capi.spmm_csr_workspace_bytes("mul", "sum", csr, x.dtype, x, w, mul), dtype=torch.uint8,
                             device=dev)
            _capi.spmm_csr("mul", "sum", csr, x, w, mul, None, None, ws)
            cpy = torch.full((n_dst, feat), -3.0, dtype=tdtype, device=dev)
            ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_rhs", "sum", csr, w.dtype, None, w, cpy),
                             dtype=torch.uint8, device=dev)
            _capi.spmm_csr("copy_rhs", "sum", csr, None, w, cpy, None, None, ws)
            torch.cuda.synchronize()
            got[on] = (seg, mul, cpy, smax, amax)
    for a, b in zip(got[0], got[1]):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    assert torch.equal(got[1][0].view(torch.uint8), got[1][2].view(torch.uint8))  # segment sum == copy_e sum
    deg = (off[1:] - off[:-1])
    rows_of = torch.repeat_interleave(torch.arange(n_dst, device=dev), deg)
    ref_max = torch.full((n_dst, feat), float("-inf"), dtype=torch.float64, device=dev).index_reduce_(
        0, rows_of, w.double(), "amax", include_self=True)
    has = (deg > 0)[:, None].expand(-1, feat)
    assert torch.equal(got[1][3].double()[has], ref_max[has])
    if tdtype in (torch.float32, torch.float64):
        host = [t.cpu().numpy() for t in (g["indptr"], g["indices"])]
        ref, _, _ = oracle.spmm_csr("mul", "sum", host[0], host[1], None, x.cpu().numpy(), w.cpu().numpy())
        np.testing.assert_allclose(got[1][1].cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("feat,tdtype", [(100, torch.float32), (32, torch.float32), (64, torch.bfloat16), (128, torch.float16),
                                         (40, torch.float64), (300, torch.float32), (8, torch.float32)])
@pytest.mark.parametrize("op,reduce", [("copy_lhs", "sum"), ("mul", "sum"), ("copy_lhs", "max"), ("mul", "min"),
                                       ("copy_rhs", "sum")])
def test_in_kernel_fixup_gives_the_bits_of_the_fixup_kernel(dev, feat, tdtype, op, reduce, monkeypatch):
    """Round 4: rows that straddle slots are finished INSIDE the merge launch by the slot that draws the last
    ticket, combining the parts in slot order — the order of the separate fix-up kernel, so the results must be
    bit-identical to it (DGLA_SPMM_FUSE_FIXUP=0), values and winners, on a graph with hub rows that span many
    slots, over repeated launches on one workspace (the arrival counters must come back to zero), with mean and
    accumulate.  Shapes cover one slot per wave (F = 100 fp32), several lane groups, and the shapes that keep the
    separate kernel (feature chunks > 1: F = 300; more lane groups than counters: F = 8)."""
    from dgl_amd import _capi

    n_dst, n_src, e = 6_000, 9_000, 400_000
    g = synth_csr(n_dst, n_src, e, "U", seed=91, device=dev, with_eids=True)
    ip = g["indptr"].clone()
    # two hub rows of ~60 k edges: rows 10 and 4000 swallow their neighbours' edges (indptr stays monotone)
    ip[11:1200] = ip[1200]
    ip[4001:5500] = ip[5500]
    torch.manual_seed(7)
    x = (torch.rand(n_src, feat, device=dev) + 1).to(tdtype)
    w = (torch.rand(e, 1, device=dev) + 1).to(tdtype) if op in ("mul",) else \
        ((torch.rand(e, feat, device=dev) + 1).to(tdtype) if op == "copy_rhs" else None)
    u = None if op == "copy_rhs" else x
    csr = _capi.make_csr(ip, g["indices"], g["eids"], n_src)
    res = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("DGLA_SPMM_FUSE_FIXUP", fuse)
        out = torch.full((n_dst, feat), -3.0, dtype=tdtype, device=dev)
        au = torch.full((n_dst, feat), -9, dtype=torch.int32, device=dev) if reduce != "sum" else None
        ae = torch.full((n_dst, feat), -9, dtype=torch.int32, device=dev) if reduce != "sum" else None
        ws = torch.empty(_capi.spmm_csr_workspace_bytes(op, reduce, csr, tdtype, u, w, out), dtype=torch.uint8, device=dev)
        _capi.spmm_csr(op, reduce, csr, u, w, out, au, ae, ws)
        outs = [out.clone()]
        for _ in range(2):                       # cached plan: the counters were left at zero
            _capi.spmm_csr(op, reduce, csr, u, w, out, au, ae, ws, plan_valid=True)
            outs.append(out.clone())
        extra = []
        if reduce == "sum":
            acc = torch.full((n_dst, feat), 0.5, dtype=tdtype, device=dev)
            _capi.spmm_csr(op, reduce, csr, u, w, acc, None, None, ws, accumulate=True, plan_valid=True)
            extra.append(acc)
            if op == "copy_lhs":
                mean = torch.empty((n_dst, feat), dtype=tdtype, device=dev)
                _capi.spmm_csr(op, reduce, csr, u, w, mean, None, None, ws, plan_valid=True, mean=True)
                extra.append(mean)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        res[fuse] = [outs[0]] + extra + ([au, ae] if au is not None else [])
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a.view(torch.uint8) if a.dtype.is_floating_point else a,
                           b.view(torch.uint8) if b.dtype.is_floating_point else b)


def test_prepare_only_is_the_producers_half_of_a_call(dev):
    """DGLA_PREPARE_ONLY: plan + side copy of the operand's ragged row ends, no output; the consumer's call with
    DGLA_SPLIT_VALID then launches the merge kernel alone and gives the bits of an ordinary call."""
    from dgl_amd import _capi

    n_dst, n_src, e, feat = 50_000, 180_000, 800_000, 100
    g = synth_csr(n_dst, n_src, e, "U", seed=77, device=dev)
    torch.manual_seed(5)
    x = torch.rand(n_src, feat, device=dev) + 1
    csr = _capi.make_csr(g["indptr"], g["indices"], None, n_src)
    out = torch.full((n_dst, feat), -3.0, device=dev)
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out), dtype=torch.uint8, device=dev)
    want = torch.empty_like(out)
    ws2 = torch.empty_like(ws)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, want, None, None, ws2)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, split_keep=True, prepare_only=True)
    torch.cuda.synchronize()
    assert bool((out == -3.0).all())                                   # nothing written
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True, split_keep=True, split_valid=True)
    assert torch.equal(out, want)
    with pytest.raises(Exception):
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, prepare_only=True)   # needs split_keep
