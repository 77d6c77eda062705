"""Position-ordered hand-off of edge tensors (dgl_amd.edge_order, VERDICT r2 Next #4) on the GPU:
the GAT pipeline u_add_v -> leaky_relu -> edge_softmax -> u_mul_e_sum and its gradients with the
hand-off ON must equal the same computation with the hand-off OFF (the plain edge-id-ordered path,
which is the one pinned to the oracle everywhere else), on graphs built from an UNSORTED COO so
that the in-edge CSR carries DGL's usual edge-id map; everything a user can read is edge-id ordered."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _graph(dev, n=3000, e=40000, seed=0, idtype=torch.int32):
    import dgl_amd as dgl

    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = (torch.rand(e, generator=g) ** 2 * n).long().clamp_(max=n - 1)   # skewed in-degrees, some hubs
    dst[:3000] = 7                                                        # a hub row longer than a merge unit
    return dgl.graph((src.to(dev), dst.to(dev)), num_nodes=n, idtype=idtype, device=dev)


def _gat(g, el, er, ft, drop_mask=None):
    import dgl_amd as dgl
    import dgl_amd.function as fn

    with g.local_scope():
        g.srcdata.update({"ft": ft, "el": el})
        g.dstdata.update({"er": er})
        g.apply_edges(fn.u_add_v("el", "er", "e"))
        e = F.leaky_relu(g.edata.pop("e"), 0.2)
        a = dgl.edge_softmax(g, e)
        if drop_mask is not None:
            a = a * drop_mask                      # an edge-id-ordered tensor: forces the conversion
        g.edata["a"] = a
        g.update_all(fn.u_mul_e("ft", "a", "m"), fn.sum("m", "o"))
        return g.dstdata["o"], g.edata["a"]


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("heads,d", [(8, 8), (4, 16), (1, 32)])
def test_gat_pipeline_handoff_equals_plain_path(dev, idtype, heads, d):
    import dgl_amd as dgl
    from dgl_amd import edge_order as E

    g = _graph(dev, idtype=idtype)
    assert g._graph.relations[0].csc()[2] is not None          # the usual edge-id map is there
    n = g.num_nodes()
    torch.manual_seed(1)
    mk = lambda *s: torch.randn(*s, device=dev)
    base = [mk(n, heads, 1), mk(n, heads, 1), mk(n, heads, d)]
    up = mk(n, heads, d)
    res = {}
    for on in (False, True):
        dgl.set_edge_order_handoff(on)
        try:
            el, er, ft = (t.clone().requires_grad_(True) for t in base)
            out, a = _gat(g, el, er, ft)
            assert (type(a) is E.PosOrdered) == on
            (out * up).sum().backward()
            res[on] = (out.detach(), E.to_eid_order(a).detach(), el.grad, er.grad, ft.grad)
        finally:
            dgl.set_edge_order_handoff(False)
    for x, y, what in zip(res[True], res[False], ("out", "attention", "d el", "d er", "d ft")):
        torch.testing.assert_close(x, y, rtol=2e-5, atol=2e-6, msg=lambda m: what + ": " + m)
    # and the plain path itself against a dense evaluation of the same layer (independent check)
    src, dst = g.edges()
    e = F.leaky_relu(base[0][src.long()] + base[1][dst.long()], 0.2).double()
    ex = torch.exp(e - torch.zeros(n, heads, 1, device=dev, dtype=torch.float64).index_reduce_(
        0, dst.long(), e, "amax", include_self=False)[dst.long()])
    den = torch.zeros(n, heads, 1, device=dev, dtype=torch.float64).index_add_(0, dst.long(), ex)
    att = ex / den[dst.long()]
    want = torch.zeros(n, heads, d, device=dev, dtype=torch.float64).index_add_(0, dst.long(), att * base[2][src.long()].double())
    torch.testing.assert_close(res[True][1].double(), att, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(res[True][0].double(), want, rtol=1e-4, atol=1e-5)


def test_user_visible_values_are_edge_id_ordered(dev):
    """What user code reads — edata, indexing, arithmetic with its own tensors, .cpu() — is the
    reference's edge-id order whatever the storage order is."""
    import dgl_amd as dgl
    from dgl_amd import edge_order as E

    g = _graph(dev, n=500, e=6000, seed=3)
    n, e_cnt = g.num_nodes(), g.num_edges()
    torch.manual_seed(2)
    el, er = torch.randn(n, 4, 1, device=dev), torch.randn(n, 4, 1, device=dev)
    src, dst = g.edges()
    want = el[src.long()] + er[dst.long()]
    assert type(dgl.ops.u_add_v(g, el, er)) is torch.Tensor        # the default: plain tensors, as the reference
    with dgl.edge_order_handoff():
        got = dgl.ops.u_add_v(g, el, er)
    assert type(got) is E.PosOrdered and got.shape == want.shape
    assert not E.handoff_enabled()
    assert torch.equal(got.cpu(), want.cpu())
    assert torch.equal(got[torch.tensor([5, 0, e_cnt - 1], device=dev)], want[[5, 0, e_cnt - 1]])
    assert torch.equal(got + torch.zeros_like(want), want)
    assert torch.equal(torch.as_tensor(got.cpu().numpy()), want.cpu())
    mask = torch.rand(e_cnt, 1, 1, device=dev) > 0.5
    assert torch.equal(got * mask, want * mask)
    sm = dgl.edge_softmax(g, got)             # a tag that arrives is honoured outside the scope too
    sm_plain = dgl.edge_softmax(g, want)
    assert type(sm) is E.PosOrdered and type(sm_plain) is torch.Tensor
    torch.testing.assert_close(sm.eid_order(), sm_plain, rtol=1e-6, atol=1e-8)
    # a tagged tensor of ANOTHER graph with the same shape is not mistaken for this graph's layout
    g2 = _graph(dev, n=500, e=6000, seed=4)
    with dgl.edge_order_handoff():
        other = dgl.ops.u_add_v(g2, el, er)
    assert type(other) is E.PosOrdered
    out = dgl.ops.copy_e_sum(g, other)
    s2, d2 = g2.edges()
    # (fp64 reference: an fp32 index_add_ runs on atomics in a different order every time, and with
    # randn operands a near-zero sum of a dozen terms carries ~1e-6 of ABSOLUTE error either way)
    want2 = torch.zeros(n, 4, 1, device=dev, dtype=torch.float64).index_add_(
        0, dst.long(), (el[s2.long()] + er[d2.long()]).double())
    torch.testing.assert_close(out.double(), want2, rtol=1e-5, atol=1e-5)
    # max / min reducers need edge ids for arg_e: the tagged operand is converted, results unchanged
    mx = dgl.ops.copy_e_max(g, got)
    ref = torch.full((n, 4, 1), float("-inf"), device=dev).index_reduce_(0, dst.long(), want, "amax", include_self=True)
    # (rows nobody reaches keep the reducer's identity at this level, as in the reference's gspmm)
    torch.testing.assert_close(mx, ref)


@pytest.mark.parametrize("plain_in", [True, False])
def test_edge_softmax_standalone_and_with_mask(dev, plain_in):
    """edge_softmax on a plain edge-id-ordered score (gathered once on the way in, gradient
    scattered once on the way out) and a pipeline interrupted by an edge-id-ordered mask."""
    import dgl_amd as dgl

    g = _graph(dev, n=2000, e=30000, seed=5, idtype=torch.int64)
    n, e_cnt = g.num_nodes(), g.num_edges()
    torch.manual_seed(3)
    base = [torch.randn(n, 8, 1, device=dev), torch.randn(n, 8, 1, device=dev), torch.randn(n, 8, 4, device=dev)]
    score0 = torch.randn(e_cnt, 8, 1, device=dev)
    mask = (torch.rand(e_cnt, 1, 1, device=dev) > 0.3).float()
    up = torch.randn(n, 8, 4, device=dev)
    res = {}
    for on in (False, True):
        dgl.set_edge_order_handoff(on)
        try:
            if plain_in:
                s = score0.clone().requires_grad_(True)
                a = dgl.edge_softmax(g, s)
                ft = base[2].clone().requires_grad_(True)
                out = dgl.ops.u_mul_e_sum(g, ft, a * mask)
                (out * up).sum().backward()
                res[on] = (out.detach(), s.grad, ft.grad)
            else:
                el, er, ft = (t.clone().requires_grad_(True) for t in base)
                out, _ = _gat(g, el, er, ft, drop_mask=mask)
                (out * up).sum().backward()
                res[on] = (out.detach(), el.grad, er.grad, ft.grad)
        finally:
            dgl.set_edge_order_handoff(False)
    for x, y in zip(res[True], res[False]):
        torch.testing.assert_close(x, y, rtol=2e-5, atol=2e-6)


def test_u_dot_v_attention_and_1d_features(dev):
    """Transformer-style attention: u_dot_v scores -> softmax -> u_mul_e_sum; and 1-D edge tensors."""
    import dgl_amd as dgl

    g = _graph(dev, n=1500, e=20000, seed=6)
    n = g.num_nodes()
    torch.manual_seed(4)
    q, k, v = (torch.randn(n, 4, 16, device=dev) for _ in range(3))
    w1 = torch.rand(n, device=dev)
    res = {}
    for on in (False, True):
        dgl.set_edge_order_handoff(on)
        try:
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            s = dgl.ops.u_dot_v(g, kk, qq) / 4.0
            a = dgl.edge_softmax(g, s)
            out = dgl.ops.u_mul_e_sum(g, vv, a)
            out.square().sum().backward()
            x1 = w1.clone().requires_grad_(True)
            e1 = dgl.ops.u_add_v(g, x1, x1)              # 1-D
            o1 = dgl.ops.u_mul_e_sum(g, x1, torch.tanh(e1))
            o1.sum().backward()
            res[on] = (out.detach(), qq.grad, kk.grad, vv.grad, o1.detach(), x1.grad)
        finally:
            dgl.set_edge_order_handoff(False)
    for x, y in zip(res[True], res[False]):
        torch.testing.assert_close(x, y, rtol=5e-5, atol=5e-6)


def test_map_free_graphs_and_blocks_do_not_start_a_handoff(dev):
    import dgl_amd as dgl
    from dgl_amd import edge_order as E

    n = 300
    dst = torch.sort(torch.randint(0, n, (4000,)))[0]
    src = torch.randint(0, n, (4000,))
    g = dgl.graph((src.to(dev), dst.to(dev)), num_nodes=n, device=dev)   # edges already sorted by destination
    x = torch.randn(n, 2, device=dev)
    rel = g._graph.relations[0]
    out = dgl.ops.u_add_v(g, x, x)
    if rel.csc()[2] is None:
        assert type(out) is torch.Tensor
    assert not E.wants_handoff(rel) or rel.csc()[2] is not None


def test_explicit_gradient_and_inplace_sources_on_the_real_kernels(dev):
    """``edge_softmax(g, s).backward(grad)`` with a NON-uniform explicit gradient, and
    ``buf.copy_(attn)`` / ``frame[:] = attn`` with the hand-off on, equal the plain path."""
    import dgl_amd as dgl
    from dgl_amd import edge_order as E

    g = _graph(dev, n=2000, e=30000, seed=3)
    torch.manual_seed(5)
    s0 = torch.randn(g.num_edges(), 4, 1, device=dev)
    gy = torch.randn(g.num_edges(), 4, 1, device=dev)
    res = {}
    for on in (False, True):
        dgl.set_edge_order_handoff(on)
        try:
            s = s0.clone().requires_grad_(True)
            a = dgl.edge_softmax(g, F.leaky_relu(s, 0.2))
            assert (type(a) is E.PosOrdered) == on
            buf = torch.zeros_like(s0)
            buf.copy_(a.detach())
            frame = torch.zeros_like(s0)
            frame[:] = a.detach()
            a.backward(gy)
            (g2,) = torch.autograd.grad(dgl.edge_softmax(g, F.leaky_relu(s, 0.2)), s, grad_outputs=gy)
            res[on] = (E.to_eid_order(a).detach(), buf, frame, s.grad.clone(), g2)
        finally:
            dgl.set_edge_order_handoff(False)
    for x, y, what in zip(res[True], res[False], ("attention", "copy_", "setitem", "backward(grad)", "grad(grad_outputs)")):
        torch.testing.assert_close(x, y, rtol=2e-5, atol=2e-6, msg=lambda m: what + ": " + m)


def mapped_of(indptr, indices, eids, n):
    from dgl_amd import _capi
    return _capi.make_csr(indptr, indices, eids, n)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1,), (3,), (8,), (4, 4), (16,)])
def test_softmax_reads_by_edge_id_and_writes_by_position_in_one_pass(dev, idtype, dtype, shape):
    """DGLA_ESM_OUT_POSITION: the merge-path softmax over a CSC WITH an edge-id map, output in position order ==
    gather_rows through the map followed by the map-free softmax, bit for bit — rows cut by unit boundaries (a hub of
    60 k in-edges: the fix-up kernel's path), empty rows, every vector width of the kernel."""
    from dgl_amd import _capi

    n, e = 3000, 200000
    g0 = torch.Generator().manual_seed(12)
    src = torch.randint(0, n, (e,), generator=g0)
    dst = torch.randint(n // 8, n, (e,), generator=g0)
    dst[: 60000] = n - 3
    indptr, indices, eids = _capi.coo_to_csr(dst.to(dev).to(idtype), src.to(dev).to(idtype), None, n, n)
    assert eids is not None and not torch.equal(eids.long(), torch.arange(e, device=dev))
    score = (torch.randn((e,) + shape, device=dev) * 3).to(dtype)
    dim = 1
    for d in shape:
        dim *= d
    plain = _capi.make_csr(indptr, indices, None, n)
    mapped = _capi.make_csr(indptr, indices, eids, n)
    need = int(_capi.edge_softmax_workspace_bytes(plain, dtype, dim))
    if need == 0:   # no merge-path kernel for this shape (fp64 with 16 columns): the flag is refused
        with pytest.raises(Exception, match="DGLA_ESM_OUT_POSITION"):
            _capi.edge_softmax_forward(mapped_of(indptr, indices, eids, n), score, torch.empty_like(score), None,
                                       out_position=True)
        return
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    want = torch.empty_like(score)
    _capi.edge_softmax_forward(plain, _capi.gather_rows(score, eids), want, ws)
    got = torch.full_like(score, float("nan"))
    _capi.edge_softmax_forward(mapped, score, got, ws, plan_valid=True, out_position=True)
    assert torch.equal(got, want)
    # and the flag is refused where it cannot be honoured
    with pytest.raises(Exception, match="DGLA_ESM_OUT_POSITION"):
        _capi.edge_softmax_forward(mapped, score, got, None, out_position=True)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1,), (3,), (8,), (4, 4)])
def test_softmax_backward_forms_out_times_grad_inside_the_kernel(dev, idtype, dtype, shape):
    """DGLA_ESM_B_IS_GRAD: backward(out, g) with the flag == backward(out, out * g) without it, bit for bit (the
    product is rounded to the storage type exactly as torch's elementwise kernel leaves it) — map-free CSC (the
    transposing fast path), CSC with an edge-id map, a hub row through the fix-up kernel."""
    from dgl_amd import _capi

    n, e = 3000, 200000
    g0 = torch.Generator().manual_seed(13)
    src = torch.randint(0, n, (e,), generator=g0)
    dst = torch.randint(n // 8, n, (e,), generator=g0)
    dst[: 60000] = n - 3
    indptr, indices, eids = _capi.coo_to_csr(dst.to(dev).to(idtype), src.to(dev).to(idtype), None, n, n)
    dim = 1
    for d in shape:
        dim *= d
    for m in (None, eids):
        csr = _capi.make_csr(indptr, indices, m, n)
        need = int(_capi.edge_softmax_workspace_bytes(csr, dtype, dim))
        if need == 0:
            continue
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        score = (torch.randn((e,) + shape, device=dev) * 2).to(dtype)
        out = torch.empty_like(score)
        _capi.edge_softmax_forward(csr, score, out, ws)
        g = torch.randn((e,) + shape, device=dev).to(dtype)
        want, got = torch.empty_like(out), torch.full_like(out, float("nan"))
        _capi.edge_softmax_backward(csr, out, (out * g).contiguous(), want, ws, plan_valid=True)
        _capi.edge_softmax_backward(csr, out, g, got, ws, plan_valid=True, sds_is_grad=True)
        assert torch.equal(got, want)
    # no workspace = the lane-group kernel: the flag is refused
    score = torch.randn((e,) + shape, device=dev).to(dtype)
    with pytest.raises(Exception, match="DGLA_ESM_B_IS_GRAD"):
        _capi.edge_softmax_backward(_capi.make_csr(indptr, indices, None, n), score, score, torch.empty_like(score), None,
                                    sds_is_grad=True)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1,), (3,), (8,), (4, 4), (16,)])
@pytest.mark.parametrize("is_grad", [False, True])
def test_softmax_backward_with_the_saved_softmax_in_position_order(dev, idtype, dtype, shape, is_grad):
    """DGLA_ESM_OUT_POSITION in the BACKWARD pass (round 5): `out` read and `back` written by position, `sds` read
    by edge id through the map == the all-edge-id backward, bit for bit, permuted — with and without
    DGLA_ESM_B_IS_GRAD, a hub row through the fix-up kernel, empty rows, every vector width."""
    from dgl_amd import _capi

    n, e = 3000, 200000
    g0 = torch.Generator().manual_seed(14)
    src = torch.randint(0, n, (e,), generator=g0)
    dst = torch.randint(n // 8, n, (e,), generator=g0)
    dst[: 60000] = n - 3
    indptr, indices, eids = _capi.coo_to_csr(dst.to(dev).to(idtype), src.to(dev).to(idtype), None, n, n)
    dim = 1
    for d in shape:
        dim *= d
    mapped = _capi.make_csr(indptr, indices, eids, n)
    need = int(_capi.edge_softmax_workspace_bytes(mapped, dtype, dim))
    if need == 0:
        with pytest.raises(Exception, match="DGLA_ESM_OUT_POSITION"):
            z = torch.zeros((e,) + shape, device=dev, dtype=dtype)
            _capi.edge_softmax_backward(mapped, z, z, torch.empty_like(z), None, out_position=True)
        return
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    score = (torch.randn((e,) + shape, device=dev) * 2).to(dtype)
    out = torch.empty_like(score)
    _capi.edge_softmax_forward(mapped, score, out, ws)                       # edge-id order in and out
    out_pos = _capi.gather_rows(out, eids)                                    # the same softmax, by position
    g = torch.randn((e,) + shape, device=dev).to(dtype)
    sds = g if is_grad else (out * g).contiguous()
    want = torch.empty_like(out)
    _capi.edge_softmax_backward(mapped, out, sds, want, ws, plan_valid=True, sds_is_grad=is_grad)
    got = torch.full_like(out, float("nan"))
    _capi.edge_softmax_backward(mapped, out_pos, sds, got, ws, plan_valid=True, sds_is_grad=is_grad, out_position=True)
    assert torch.equal(got, _capi.gather_rows(want, eids))


@pytest.mark.parametrize("norm_by", ["dst", "src"])
@pytest.mark.parametrize("shape", [(), (8, 1), (3,)])
def test_plain_edge_softmax_keeps_position_order_to_itself(dev, shape, norm_by, monkeypatch):
    """The default (plain) ``dgl.edge_softmax`` behind an edge-id map: forward and gradient equal the dense evaluation
    and the map-through route it replaces, nothing but plain tensors leaves, the saved tensor is the position-ordered one."""
    import dgl_amd as dgl
    from dgl_amd import autograd, edge_order as E
    from dgl_amd.sparse_kernels import _edge_softmax_backward, _edge_softmax_forward

    monkeypatch.setattr(E, "PLAIN_SOFTMAX_POS_MIN_EDGES", 0)     # (the route is reserved for large graphs: force it here)
    g = _graph(dev, n=2500, e=50000, seed=8)
    rel = g._graph.relations[0] if norm_by == "dst" else g._graph.reverse().relations[0]
    assert rel.csc()[2] is not None
    torch.manual_seed(6)
    s0 = torch.randn((g.num_edges(),) + shape, device=dev) * 2
    up = torch.randn((g.num_edges(),) + shape, device=dev)
    s = s0.clone().requires_grad_(True)
    a = dgl.edge_softmax(g, s, norm_by=norm_by)
    assert type(a) is torch.Tensor and E.plain_softmax_route(rel, s0)
    a.backward(up)
    assert type(s.grad) is torch.Tensor
    gi = g._graph if norm_by == "dst" else g._graph.reverse()
    want = _edge_softmax_forward(gi, s0)                                      # reads and writes through the map
    assert torch.equal(a.detach(), want)
    want_back = _edge_softmax_backward(gi, want, want * up)
    assert torch.equal(s.grad, want_back)
    # dense evaluation
    src, dst = g.edges()
    key = (dst if norm_by == "dst" else src).long()
    sd = s0.double().reshape(g.num_edges(), -1)
    mx = torch.full((g.num_nodes(), sd.shape[1]), float("-inf"), dtype=torch.float64, device=dev).index_reduce_(
        0, key, sd, "amax", include_self=True)
    ex = torch.exp(sd - mx[key])
    den = torch.zeros_like(mx).index_add_(0, key, ex)
    torch.testing.assert_close(a.detach().double().reshape(sd.shape), ex / den[key], rtol=1e-5, atol=1e-7)
