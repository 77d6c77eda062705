"""The narrow-feature copy_e kernels (csrc/narrow_reduce.hip: `copy_rhs` with 1 ... 8 fp32 columns per edge — a lane owns
four EDGES, a wave 256 CSR positions, rows cut by a unit boundary are finished by a second kernel) against torch index
arithmetic: values (sums against float64, max / min exact), winners' edge ids (first position wins a tie, as in the
reference's sequential loop: src/array/cpu/spmm.h:150-200), rows without an edge, rows spanning many units, boundaries that
fall on unit boundaries, with and without an edge-id map, int32 / int64 ids.  `dgla_narrow_reduce_calls` proves which
kernel family took the call."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _csr(dev, degs, idtype, seed, with_eids):
    deg = torch.as_tensor(degs, dtype=torch.int64)
    indptr = torch.zeros(deg.numel() + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(deg, 0)
    nnz = int(indptr[-1])
    g = torch.Generator().manual_seed(seed)
    indices = torch.randint(0, max(1, deg.numel()), (nnz,), generator=g)
    eids = torch.randperm(nnz, generator=g) if with_eids else None
    to = lambda t: None if t is None else t.to(device=dev, dtype=idtype)
    return to(indptr), to(indices), to(eids), nnz


def _reference(indptr, eids, w, red):
    n, nnz, f = indptr.numel() - 1, w.shape[0], w.shape[1]
    dev = w.device
    row = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
    vi = eids.long() if eids is not None else torch.arange(nnz, device=dev)
    vals = w[vi]
    if red == "sum":
        return torch.zeros(n, f, dtype=torch.float64, device=dev).index_add(0, row, vals.double()), None
    ident = float("-inf") if red == "max" else float("inf")
    out = torch.full((n, f), ident, device=dev)
    out = out.scatter_reduce(0, row.view(-1, 1).expand(-1, f), vals, "amax" if red == "max" else "amin", include_self=True)
    # first position that attains the winner, per (row, column)
    pos = torch.arange(nnz, device=dev).view(-1, 1).expand(-1, f)
    hit = vals == out[row]
    first = torch.full((n, f), nnz, dtype=torch.long, device=dev).scatter_reduce(
        0, row.view(-1, 1).expand(-1, f), torch.where(hit, pos, torch.full_like(pos, nnz)), "amin", include_self=True)
    arg = torch.where(first < nnz, vi[first.clamp(max=max(nnz - 1, 0))], torch.zeros_like(first))
    return out, arg


def _run(dev, degs, f, red, idtype, with_eids, seed=0, ties=False, misaligned=False):
    from dgl_amd import _capi

    indptr, indices, eids, nnz = _csr(dev, degs, idtype, seed, with_eids)
    n = indptr.numel() - 1
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    w = torch.randn(nnz, f, device=dev, generator=g)
    if ties:
        w = torch.round(w * 2) / 2                                   # many equal values: the FIRST position must win
    if misaligned:                                                   # rows that start 4 bytes off a 16-byte boundary
        buf = torch.empty(nnz * f + 1, device=dev)
        buf[1:] = w.reshape(-1)
        w = buf[1:].view(nnz, f)
        assert w.data_ptr() % 16 == 4 and w.is_contiguous()
    csr = _capi.make_csr(indptr, indices, eids, n)
    out = torch.full((n, f), 7.0, device=dev)
    arg = torch.full((n, f), -5, dtype=idtype, device=dev) if red != "sum" else None
    ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes("copy_rhs", red, csr, out.dtype, None, w, out)), dtype=torch.uint8, device=dev)
    before = _capi.narrow_reduce_calls()
    _capi.spmm_csr("copy_rhs", red, csr, None, w, out, None, arg, ws)
    assert _capi.narrow_reduce_calls() == before + 1, "the narrow kernels did not take this call"
    want, want_arg = _reference(indptr, eids, w, red)
    if red == "sum":
        scale = torch.zeros(n, f, dtype=torch.float64, device=dev).index_add(
            0, torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long()),
            w[eids.long() if eids is not None else torch.arange(nnz, device=dev)].abs().double())
        assert bool(((out.double() - want).abs() <= 1e-6 * scale + 1e-30).all())
    else:
        assert torch.equal(out, want)
        assert torch.equal(arg.long(), want_arg)
    # same bits on a second launch over the same workspace (the records are re-written, the order is fixed)
    out2 = torch.empty_like(out)
    arg2 = torch.empty_like(arg) if arg is not None else None
    _capi.spmm_csr("copy_rhs", red, csr, None, w, out2, None, arg2, ws)
    assert torch.equal(out, out2) and (arg is None or torch.equal(arg, arg2))


def _degree_cases():
    g = torch.Generator().manual_seed(5)
    rnd = torch.randint(0, 60, (3000,), generator=g).tolist()
    return {
        "random-with-empty-rows": rnd,
        "hub-rows": [3, 0, 5000, 1, 0, 0, 700, 256, 2, 12000, 0, 4],
        "boundaries-on-unit-boundaries": [256, 256, 512, 128, 128, 0, 256, 1024, 1],
        "boundaries-on-large-units": [1024, 512, 1024, 2048, 0, 512, 512, 3072, 1, 1023, 1],   # (units of 512 / 1024 edges at F <= 4)
        "one-row": [1000],
        "fewer-than-a-unit": [3, 0, 0, 2, 5],
        "empty-rows-at-both-ends": [0, 0, 0] + [7] * 100 + [0] * 50,
        "long-then-many-empties": [2000] + [0] * 600 + [1] * 300,
        "singletons": [1] * 1500,
    }


@pytest.mark.parametrize("case", list(_degree_cases()))
@pytest.mark.parametrize("red", ["sum", "max", "min"])
@pytest.mark.parametrize("f", [1, 2, 3, 4, 8])
def test_narrow_copy_e_matches_index_arithmetic(dev, case, red, f):
    _run(dev, _degree_cases()[case], f, red, torch.int64 if f % 2 else torch.int32, with_eids=(f in (2, 3, 8)), seed=f)


@pytest.mark.parametrize("case", ["random-with-empty-rows", "hub-rows", "boundaries-on-unit-boundaries", "fewer-than-a-unit",
                                  "long-then-many-empties", "singletons"])
@pytest.mark.parametrize("red", ["sum", "max"])
@pytest.mark.parametrize("f", [4, 8])
@pytest.mark.parametrize("misaligned", [False, True])
def test_edge_rows_in_position_order_staged_through_lds(dev, case, red, f, misaligned):
    """Edge rows of 4 / 8 columns WITHOUT an edge-id map are fetched as whole 1 KB wavefront loads and handed to their lanes
    through LDS (narrow_reduce_kernel<..., STAGED>): full units, a last unit cut short by the end of the edge list, ties; rows
    that are not 16-byte aligned take the lane-by-lane loads and must give the same bits."""
    for idtype in (torch.int32, torch.int64):
        _run(dev, _degree_cases()[case], f, red, idtype, with_eids=False, seed=60 + f, ties=(red == "max"), misaligned=misaligned)


@pytest.mark.parametrize("red", ["sum", "max", "min"])
@pytest.mark.parametrize("f", [1, 4, 7, 8])
def test_rows_cut_into_many_units_take_the_long_row_fix_up(dev, red, f):
    """A row of 12 000 edges is 47 units, one of 200 000 is 782: the fix-up kernel walks at most four pieces per thread and
    lists longer rows for a kernel that merges 64 pieces per step and wavefront (narrow_fixup_long_kernel) — sums to the same
    tolerance, winners and their FIRST positions exact with many ties, the same bits on a second launch (checked in _run)."""
    degs = [5, 200000, 0, 3, 70000, 1, 1024 + 256 * 4, 256 * 5, 256 * 6 + 1, 0, 9, 16640, 2]
    for idtype, with_eids in ((torch.int32, False), (torch.int64, True)):
        _run(dev, degs, f, red, idtype, with_eids, seed=70 + f, ties=(red != "sum"))


@pytest.mark.parametrize("red", ["max", "min"])
@pytest.mark.parametrize("with_eids", [False, True])
def test_first_position_wins_a_tie(dev, red, with_eids):
    _run(dev, _degree_cases()["random-with-empty-rows"], 5, red, torch.int32, with_eids, seed=11, ties=True)
    _run(dev, _degree_cases()["hub-rows"], 7, red, torch.int64, with_eids, seed=12, ties=True)


def test_at_size_and_through_the_operator_api(dev):
    """62 k rows x 1.5 M edges, 6 columns; and `dgl.ops.copy_e_max` / autograd on top of it (the route is transparent)."""
    import dgl_amd as dgl
    from dgl_amd import _capi

    g = torch.Generator().manual_seed(3)
    degs = torch.randint(0, 50, (62000,), generator=g).tolist()
    _run(dev, degs, 6, "sum", torch.int32, True, seed=21)
    _run(dev, degs, 6, "max", torch.int64, False, seed=22)
    n, e = 5000, 60000
    u, v = torch.randint(n, (e,), generator=g).to(dev), torch.randint(n, (e,), generator=g).to(dev)
    gr = dgl.graph((u, v), num_nodes=n)
    w = torch.randn(e, 4, device=dev, requires_grad=True)
    before = _capi.narrow_reduce_calls()
    y = dgl.ops.copy_e_sum(gr, w)
    assert _capi.narrow_reduce_calls() > before
    want = torch.zeros(n, 4, device=dev, dtype=torch.float64).index_add(0, v, w.double())
    assert torch.allclose(y.double(), want, atol=1e-5)
    m = dgl.ops.copy_e_max(gr, w)
    wm = torch.full((n, 4), float("-inf"), device=dev).scatter_reduce(0, v.view(-1, 1).expand(-1, 4), w.detach(), "amax")
    assert torch.equal(m.detach(), wm)
    (gw,) = torch.autograd.grad(m[torch.isfinite(m)].sum(), [w])
    hit = (w.detach() == wm[v]).double()
    assert torch.allclose(gw.double().sum(0), hit.sum(0).clamp(max=1e9) * 0 + gw.double().sum(0))   # (shape check)
    assert float(gw.sum()) == float(torch.isfinite(wm).sum())        # one unit of gradient per finite output element


def test_the_merge_plan_is_in_the_workspace_after_a_narrow_call(dev):
    """Contract of dgla_spmm_csr: after a successful call the workspace holds the graph's merge plan and the next call —
    whatever its operator — may say DGLA_PLAN_VALID.  A FIRST call served by the narrow kernels must leave that plan
    behind too (found by tests/test_gpu_api.py: copy_u_sum after copy_e_sum read a plan nobody had built)."""
    import dgl_amd as dgl
    from dgl_amd import _capi

    g = torch.Generator().manual_seed(0)
    n, e = 3000, 40000
    u, v = torch.randint(n, (e,), generator=g).to(dev), torch.randint(n, (e,), generator=g).to(dev)
    gr = dgl.graph((u, v), num_nodes=n)
    x, w = torch.rand(n, 4, device=dev), torch.rand(e, 4, device=dev)
    before = _capi.narrow_reduce_calls()
    y = dgl.ops.copy_e_sum(gr, w)                                     # first SpMM on this graph: narrow route
    assert _capi.narrow_reduce_calls() == before + 1
    z = dgl.ops.copy_u_sum(gr, x)                                     # merge kernel, plan taken as valid
    assert torch.allclose(z, torch.zeros(n, 4, device=dev).index_add_(0, v, x[u]), atol=1e-5)
    assert torch.allclose(y, torch.zeros(n, 4, device=dev).index_add_(0, v, w), atol=1e-5)
    big = torch.rand(n, 100, device=dev)                              # a wider operator re-sizes the shared workspace
    assert torch.allclose(dgl.ops.copy_u_sum(gr, big), torch.zeros(n, 100, device=dev).index_add_(0, v, big[u]), atol=1e-4)
    assert torch.allclose(dgl.ops.copy_e_max(gr, w), torch.full((n, 4), float("-inf"), device=dev).scatter_reduce(
        0, v.view(-1, 1).expand(-1, 4), w, "amax"))
    assert torch.allclose(dgl.ops.copy_u_sum(gr, x), z)
    # direct C-ABI use: plan_valid on the second call
    indptr, indices, eids = gr._graph.relations[0].csc()
    csr = _capi.make_csr(indptr, indices, eids, n)
    out = torch.empty(n, 4, device=dev)
    need = max(_capi.spmm_csr_workspace_bytes("copy_rhs", "sum", csr, out.dtype, None, w, out),
               _capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, out.dtype, x, None, out))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _capi.spmm_csr("copy_rhs", "sum", csr, None, w, out, None, None, ws)
    out2 = torch.empty(n, 4, device=dev)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out2, None, None, ws, plan_valid=True)
    assert torch.allclose(out2, z, atol=1e-5)


@pytest.mark.parametrize("f", [1, 3, 8])
def test_mean_and_segment_reduce_take_the_same_kernels(dev, f):
    """`mean` (DGLA_MEAN: sum, then an IEEE division by the in-degree) and dgla_segment_reduce (rows = segments, no
    column ids, empty segments keep argument -1) are served by the same kernels: equal bits to the compositions."""
    from dgl_amd import _capi

    degs = _degree_cases()["hub-rows"] + _degree_cases()["random-with-empty-rows"][:500]
    indptr, indices, eids, nnz = _csr(dev, degs, torch.int64, 3, True)
    n = indptr.numel() - 1
    w = torch.randn(nnz, f, device=dev)
    csr = _capi.make_csr(indptr, indices, eids, n)
    s, m = torch.empty(n, f, device=dev), torch.empty(n, f, device=dev)
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_rhs", "sum", csr, s.dtype, None, w, s), dtype=torch.uint8, device=dev)
    before = _capi.narrow_reduce_calls()
    _capi.spmm_csr("copy_rhs", "sum", csr, None, w, s, None, None, ws)
    _capi.spmm_csr("copy_rhs", "sum", csr, None, w, m, None, None, ws, plan_valid=True, mean=True)
    assert _capi.narrow_reduce_calls() == before + 2
    deg = (indptr[1:] - indptr[:-1]).clamp(min=1).float().view(-1, 1)
    assert torch.equal(m, s / deg)
    # segment reduce over the same rows in position order
    seg = torch.full((n, f), 9.0, device=dev)
    _capi.segment_reduce("sum", w, indptr, seg)
    pos_csr = _capi.make_csr(indptr, indices, None, n)
    s2 = torch.empty(n, f, device=dev)
    _capi.spmm_csr("copy_rhs", "sum", pos_csr, None, w, s2, None, None, ws)
    assert _capi.narrow_reduce_calls() == before + 4 and torch.equal(seg, s2)
    smax = torch.full((n, f), 9.0, device=dev)
    amax = torch.full((n, f), 9, dtype=torch.int64, device=dev)
    _capi.segment_reduce("max", w, indptr, smax, amax)
    want, want_arg = _reference(indptr, None, w, "max")
    empty = (indptr[1:] == indptr[:-1]).view(-1, 1).expand(-1, f)
    assert torch.equal(smax, want) and torch.equal(amax[~empty], want_arg[~empty]) and bool((amax[empty] == -1).all())


def _ref_general(indptr, indices, eids, op, red, x, w):
    """out[r] = reduce over the positions of row r of op(x[col], w[eid]) (broadcast as torch does), + winners."""
    n, nnz = indptr.numel() - 1, indices.shape[0]
    dev = indices.device
    row = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
    eid = eids.long() if eids is not None else torch.arange(nnz, device=dev)
    col = indices.long()
    if op == "copy_lhs":
        msg = x[col]
    elif op == "copy_rhs":
        msg = w[eid]
    else:
        msg = {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div}[op](x[col], w[eid])
    msg = msg.reshape(nnz, -1)
    f = msg.shape[1]
    if red == "sum":
        return torch.zeros(n, f, dtype=torch.float64, device=dev).index_add(0, row, msg.double()), None, msg
    ident = float("-inf") if red == "max" else float("inf")
    idx = row.view(-1, 1).expand(-1, f)
    out = torch.full((n, f), ident, device=dev).scatter_reduce(0, idx, msg, "amax" if red == "max" else "amin", include_self=True)
    pos = torch.arange(nnz, device=dev).view(-1, 1).expand(-1, f)
    first = torch.full((n, f), nnz, dtype=torch.long, device=dev).scatter_reduce(
        0, idx, torch.where(msg == out[row], pos, torch.full_like(pos, nnz)), "amin", include_self=True)
    return out, first, msg


@pytest.mark.parametrize("case", ["random-with-empty-rows", "hub-rows", "boundaries-on-unit-boundaries", "fewer-than-a-unit"])
@pytest.mark.parametrize("red", ["sum", "max", "min"])
@pytest.mark.parametrize("op,ushape,eshape", [("copy_lhs", (5,), None), ("copy_lhs", (1,), None), ("mul", (8,), (8,)),
                                              ("add", (3,), (3,)), ("sub", (2, 4), (2, 1)), ("div", (4, 2), (4, 1)),
                                              ("mul", (7,), (1,)), ("mul", (2, 2), (2, 2))])
def test_narrow_copy_u_and_binary_operators(dev, case, red, op, ushape, eshape):
    from dgl_amd import _capi

    idtype = torch.int32 if red == "max" else torch.int64
    with_eids = op != "add"
    indptr, indices, eids, nnz = _csr(dev, _degree_cases()[case], idtype, 31, with_eids)
    n = indptr.numel() - 1
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((n,) + ushape, device=dev, generator=g)
    w = None if eshape is None else torch.rand((nnz,) + eshape, device=dev, generator=g) + 0.5
    csr = _capi.make_csr(indptr, indices, eids, n)
    f = x.reshape(n, -1).shape[1]
    out = torch.full((n,) + ushape, 7.0, device=dev)
    au = torch.full(out.shape, -5, dtype=idtype, device=dev) if red != "sum" else None
    ae = torch.full(out.shape, -5, dtype=idtype, device=dev) if (red != "sum" and w is not None) else None
    ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes(op, red, csr, out.dtype, x, w, out)), dtype=torch.uint8, device=dev)
    before = _capi.narrow_reduce_calls()
    _capi.spmm_csr(op, red, csr, x, w, out, au, ae, ws)
    assert _capi.narrow_reduce_calls() == before + 1
    want, first, msg = _ref_general(indptr, indices, eids, op, red, x, w)
    got = out.reshape(n, f)
    if red == "sum":
        row = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
        scale = torch.zeros(n, f, dtype=torch.float64, device=dev).index_add(0, row, msg.abs().double())
        assert bool(((got.double() - want).abs() <= 2e-6 * scale + 1e-30).all())
    else:
        assert torch.equal(got, want)
        has = first < nnz
        eid = eids.long() if eids is not None else torch.arange(nnz, device=dev)
        fc = first.clamp(max=max(nnz - 1, 0))
        assert torch.equal(au.reshape(n, f).long()[has], indices.long()[fc][has]) and bool((au.reshape(n, f)[~has] == 0).all())
        if ae is not None:
            assert torch.equal(ae.reshape(n, f).long()[has], eid[fc][has]) and bool((ae.reshape(n, f)[~has] == 0).all())


def test_narrow_operators_through_autograd(dev):
    """APPNP / SGC-like propagation of class logits: u_mul_e_sum with 7 columns and a scalar edge weight, forward and both
    gradients (the backward runs copy / SDDMM kernels of its own; the forward and dX = SpMM over the reverse graph are narrow)."""
    import dgl_amd as dgl
    from dgl_amd import _capi

    g0 = torch.Generator().manual_seed(9)
    n, e = 4000, 50000
    u, v = torch.randint(n, (e,), generator=g0).to(dev), torch.randint(n, (e,), generator=g0).to(dev)
    gr = dgl.graph((u, v), num_nodes=n)
    x = torch.randn(n, 7, device=dev, requires_grad=True)
    w = (torch.rand(e, 1, device=dev) + 0.5).requires_grad_(True)
    before = _capi.narrow_reduce_calls()
    y = dgl.ops.u_mul_e_sum(gr, x, w)
    want = torch.zeros(n, 7, device=dev, dtype=torch.float64).index_add(0, v, (x[u] * w).double())
    assert _capi.narrow_reduce_calls() > before and torch.allclose(y.double(), want, atol=1e-5)
    up = torch.randn_like(y)
    gx, gw = torch.autograd.grad((y * up).sum(), [x, w])
    wx, ww = torch.autograd.grad((want * up.double()).sum(), [x, w])
    assert torch.allclose(gx, wx, atol=1e-5) and torch.allclose(gw, ww, atol=1e-5)
    m = dgl.ops.copy_u_max(gr, x)
    wm = torch.full((n, 7), float("-inf"), device=dev).scatter_reduce(0, v.view(-1, 1).expand(-1, 7), x.detach()[u], "amax")
    assert torch.equal(m.detach(), wm)


def test_accumulating_calls_and_the_relation_loop_of_a_heterograph(dev):
    """DGLA_ACCUMULATE (out += result; rows without an edge untouched) — what the per-relation loop behind `update_all` on a
    heterograph uses for sums — on the narrow kernels, directly and through `multi`-relation `update_all`."""
    import dgl_amd as dgl
    import dgl_amd.function as fn
    from dgl_amd import _capi

    indptr, indices, eids, nnz = _csr(dev, _degree_cases()["hub-rows"] + [0, 3, 0, 9], torch.int64, 41, True)
    n = indptr.numel() - 1
    w = torch.randn(nnz, 3, device=dev)
    csr = _capi.make_csr(indptr, indices, eids, n)
    base = torch.randn(n, 3, device=dev)
    out = base.clone()
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_rhs", "sum", csr, out.dtype, None, w, out), dtype=torch.uint8, device=dev)
    before = _capi.narrow_reduce_calls()
    _capi.spmm_csr("copy_rhs", "sum", csr, None, w, out, None, None, ws, accumulate=True)
    assert _capi.narrow_reduce_calls() == before + 1
    plain = torch.empty(n, 3, device=dev)
    _capi.spmm_csr("copy_rhs", "sum", csr, None, w, plain, None, None, ws, plan_valid=True)
    assert torch.equal(out, base + plain)
    g0 = torch.Generator().manual_seed(1)
    na, nb, e = 300, 200, 4000
    pair = lambda s, d: (torch.randint(s, (e,), generator=g0).to(dev), torch.randint(d, (e,), generator=g0).to(dev))
    hg = dgl.heterograph({("a", "r1", "b"): pair(na, nb), ("a", "r2", "b"): pair(na, nb), ("b", "r3", "b"): pair(nb, nb)},
                         {"a": na, "b": nb})
    hg.nodes["a"].data["h"] = torch.randn(na, 5, device=dev)
    hg.nodes["b"].data["h"] = torch.randn(nb, 5, device=dev)
    hg.update_all(fn.copy_u("h", "m"), fn.sum("m", "y"))     # (relations sharing a destination type: ONE stacked merge launch,
    want = torch.zeros(nb, 5, device=dev, dtype=torch.float64)
    for c in hg.canonical_etypes:
        u, v = hg.edges(etype=c)
        want.index_add_(0, v, hg.nodes[c[0]].data["h"].double()[u])
    assert torch.allclose(hg.nodes["b"].data["y"].double(), want, atol=1e-5)   #  not this file's kernels — values all the same)
    hg2 = dgl.heterograph({("a", "r1", "b"): hg.edges(etype="r1"), ("b", "r3", "b"): hg.edges(etype="r3")}, {"a": na, "b": nb})
    hg2.nodes["a"].data["h"], hg2.nodes["b"].data["h"] = hg.nodes["a"].data["h"], hg.nodes["b"].data["h"]
    hg2.update_all(fn.copy_u("h", "m"), fn.max("m", "y"))     # max over relations: per-relation candidates + running compare
    wm = torch.full((nb, 5), float("-inf"), device=dev)
    for c in hg2.canonical_etypes:
        u, v = hg2.edges(etype=c)
        wm = wm.scatter_reduce(0, v.view(-1, 1).expand(-1, 5), hg2.nodes[c[0]].data["h"][u], "amax")
    wm = torch.where(torch.isinf(wm), torch.zeros_like(wm), wm)
    assert torch.equal(hg2.nodes["b"].data["y"], wm)
