"""GPU tests of the operator API (dgl_amd.ops / DGLGraph.update_all / apply_edges / edge_softmax)
through the registry layer (DGLFuncCall -> sparse._CAPI_DGLKernel*).

Structure follows the reference's tests: tests/python/common/ops/test_ops.py (forward AND
gradients of gspmm / gsddmm over broadcast shapes), test_edge_softmax.py, function/test_basics.py
(update_all known answers), tests/python/pytorch/test_ffi-stream.py (current-stream semantics)
and tests/python/pytorch/nn/test_nn.py:40-75 (GraphConv == dense A X W).  The reference checks
against its slow UDF path; here the checker is the CPU oracle for forward values and plain
PyTorch (gather / index_add_ / scatter_reduce with autograd) for gradients.
"""
import math

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

SPMM_SHAPES = [((1, 2, 1, 3, 1), (4, 1, 3, 1, 1)), ((5, 3, 1, 7), (1, 3, 7, 1)), ((1, 3, 1), (4, 1, 3)),
               ((3, 3), (1, 3)), ((1,), (3,)), ((3,), (1,)), ((1,), (1,)), ((), ())]
SDDMM_SHAPES = [((1, 2, 1, 3, 1), (4, 1, 3, 1, 1)), ((5, 3, 1, 7), (1, 3, 7, 7)), ((1, 3, 3), (4, 1, 3)),
                ((3,), (3,)), ((1,), (1,))]


def make_graph(kind, dev, idtype):
    import dgl_amd as dgl

    if kind == "homo":
        return dgl.rand_graph(30, 100, idtype=idtype, device=dev, seed=1)
    return dgl.rand_bipartite("_U", "_E", "_V", 30, 40, 300, idtype=idtype, device=dev, seed=2)


def graph_arrays(g):
    rel = g._graph.relations[0]
    csc = [t.cpu().numpy() for t in rel.csc()]
    src, dst = [t.cpu().numpy() for t in g.edges()]
    return csc, src, dst


def torch_spmm(op, reduce, src, dst, n_dst, u, e):
    """Plain PyTorch reference with autograd: messages by gather, reduction by index ops."""
    src, dst = src.long(), dst.long()
    if op == "copy_lhs":
        m = u[src]
    elif op == "copy_rhs":
        m = e
    else:
        a, b = u[src], e
        nd = max(a.dim(), b.dim())
        a = a.reshape((a.shape[0],) + (1,) * (nd - a.dim()) + tuple(a.shape[1:]))
        b = b.reshape((b.shape[0],) + (1,) * (nd - b.dim()) + tuple(b.shape[1:]))
        m = {"add": a + b, "sub": a - b, "mul": a * b, "div": a / b}[op]
    out_shape = (n_dst,) + tuple(m.shape[1:])
    if reduce in ("sum", "mean"):
        out = torch.zeros(out_shape, dtype=m.dtype, device=m.device).index_add_(0, dst, m)
        if reduce == "mean":
            deg = torch.bincount(dst, minlength=n_dst).clamp(min=1).to(m.dtype)
            out = out / deg.reshape((-1,) + (1,) * (m.dim() - 1))
        return out
    idx = dst.reshape((-1,) + (1,) * (m.dim() - 1)).expand_as(m)
    init = torch.zeros(out_shape, dtype=m.dtype, device=m.device)
    return init.scatter_reduce(0, idx, m, "amax" if reduce == "max" else "amin", include_self=False)


@pytest.mark.parametrize("kind", ["homo", "bipartite"])
@pytest.mark.parametrize("shp", SPMM_SHAPES)
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "copy_lhs", "copy_rhs"])
@pytest.mark.parametrize("reduce", ["sum", "min", "max", "mean"])
@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_gspmm_forward_backward(dev, kind, shp, op, reduce, idtype):
    import dgl_amd as dgl

    g = make_graph(kind, dev, idtype)
    csc, src, dst = graph_arrays(g)
    torch.manual_seed(12345)
    n_src, n_dst, n_e = g.num_src_nodes(), g.num_dst_nodes(), g.num_edges()
    u = (torch.rand((n_src,) + shp[0], device=dev, dtype=torch.float64) + 1).requires_grad_()
    e = (torch.rand((n_e,) + shp[1], device=dev, dtype=torch.float64) + 1).requires_grad_()
    uu = u if op != "copy_rhs" else None
    ee = e if op != "copy_lhs" else None
    out = dgl.ops.gspmm(g, op, reduce, uu, ee)
    # forward vs the oracle (mean = sum / clamp(deg, 1))
    o_op, o_e = (op, ee) if op not in ("sub", "div") else ("add" if op == "sub" else "mul",
                                                           -ee if op == "sub" else 1.0 / ee)
    ref, _, _ = oracle.spmm_csr(o_op, "sum" if reduce == "mean" else reduce, csc[0], csc[1], csc[2],
                                None if uu is None else uu.detach().cpu().numpy(),
                                None if o_e is None else o_e.detach().cpu().numpy())
    if reduce == "mean":
        deg = np.maximum(np.diff(csc[0]), 1).reshape((-1,) + (1,) * (ref.ndim - 1))
        ref = ref / deg
    if reduce in ("max", "min"):
        np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
    else:
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-12)
    # backward vs plain PyTorch autograd on the same (finite) outputs
    tref = torch_spmm(op, reduce, torch.as_tensor(src, device=dev), torch.as_tensor(dst, device=dev),
                      n_dst, u, e)
    finite = torch.isfinite(out.detach())
    w = torch.rand_like(tref)
    grads = torch.autograd.grad((torch.where(finite, out, torch.zeros_like(out)) * w).sum(),
                                [t for t in (uu, ee) if t is not None], allow_unused=True)
    grefs = torch.autograd.grad((tref * w).sum(), [t for t in (uu, ee) if t is not None],
                                allow_unused=True)
    for a, b in zip(grads, grefs):
        a = torch.zeros(1, device=dev) if a is None else a
        b = torch.zeros(1, device=dev) if b is None else b
        assert torch.allclose(a, b, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("kind", ["homo", "bipartite"])
@pytest.mark.parametrize("shp", SDDMM_SHAPES)
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "dot", "copy_lhs", "copy_rhs"])
@pytest.mark.parametrize("lt", ["u", "v", "e"])
@pytest.mark.parametrize("rt", ["u", "v", "e"])
def test_gsddmm_forward_backward(dev, kind, shp, op, lt, rt):
    import dgl_amd as dgl

    if op == "dot" and shp[0][-1:] != shp[1][-1:]:
        pytest.skip("dot needs equal last dims")
    g = make_graph(kind, dev, torch.int32)
    _, src, dst = graph_arrays(g)
    n = {"u": g.num_src_nodes(), "v": g.num_dst_nodes(), "e": g.num_edges()}
    torch.manual_seed(12345)
    lhs = (torch.rand((n[lt],) + shp[0], device=dev, dtype=torch.float64) + 1).requires_grad_()
    rhs = (torch.rand((n[rt],) + shp[1], device=dev, dtype=torch.float64) + 1).requires_grad_()
    ll = lhs if op != "copy_rhs" else None
    rr = rhs if op != "copy_lhs" else None
    out = dgl.ops.gsddmm(g, op, ll, rr, lt, rt)
    ts, td = torch.as_tensor(src, device=dev).long(), torch.as_tensor(dst, device=dev).long()
    sel = {"u": lambda x: x[ts], "v": lambda x: x[td], "e": lambda x: x}
    if op == "copy_lhs":
        ref = sel[lt](lhs)
    elif op == "copy_rhs":
        ref = sel[rt](rhs)
    else:
        a, b = sel[lt](lhs), sel[rt](rhs)
        nd = max(a.dim(), b.dim())
        a = a.reshape((a.shape[0],) + (1,) * (nd - a.dim()) + tuple(a.shape[1:]))
        b = b.reshape((b.shape[0],) + (1,) * (nd - b.dim()) + tuple(b.shape[1:]))
        ref = {"add": lambda: a + b, "sub": lambda: a - b, "mul": lambda: a * b, "div": lambda: a / b,
               "dot": lambda: (a * b).sum(-1, keepdim=True)}[op]()
    assert out.shape == ref.shape
    assert torch.allclose(out, ref, rtol=1e-12, atol=1e-12)
    w = torch.rand_like(ref)
    ins = [t for t in (ll, rr) if t is not None]
    ga = torch.autograd.grad((out * w).sum(), ins, allow_unused=True)
    gb = torch.autograd.grad((ref * w).sum(), ins, allow_unused=True)
    for a_, b_ in zip(ga, gb):
        assert torch.allclose(a_, b_, rtol=1e-8, atol=1e-10)


def test_aliases_exist_and_route(dev):
    import dgl_amd as dgl

    g = make_graph("homo", dev, torch.int64)
    x = torch.rand(30, 5, device=dev)
    w = torch.rand(100, 5, device=dev)
    assert torch.equal(dgl.ops.copy_u_sum(g, x), dgl.ops.gspmm(g, "copy_lhs", "sum", x, None))
    assert torch.equal(dgl.ops.u_mul_e_max(g, x, w), dgl.ops.gspmm(g, "mul", "max", x, w))
    assert torch.equal(dgl.ops.copy_e_min(g, w), dgl.ops.gspmm(g, "copy_rhs", "min", None, w))
    assert torch.equal(dgl.ops.u_dot_v(g, x, x), dgl.ops.gsddmm(g, "dot", x, x))
    assert torch.equal(dgl.ops.e_sub_v(g, w, x), dgl.ops.gsddmm(g, "sub", w, x, "e", "v"))
    assert torch.equal(dgl.ops.copy_u(g, x), x[g.edges()[0].long()])
    for name in ("u_add_e_sum", "u_sub_e_max", "u_div_e_mean", "copy_u_mean", "v_mul_e", "e_dot_u"):
        assert hasattr(dgl.ops, name)
    with pytest.raises(dgl.DGLError, match="data type"):
        dgl.ops.u_mul_e_sum(g, x, w.double())


# ---------------------------------------------------------------------------------------
# edge softmax
# ---------------------------------------------------------------------------------------
def test_edge_softmax_docstring_examples(dev):
    import dgl_amd as dgl

    g = dgl.graph((torch.tensor([0, 0, 0, 1, 1, 2]), torch.tensor([0, 1, 2, 1, 2, 2])), device=dev)
    e = torch.ones(6, 1, device=dev)
    a = dgl.edge_softmax(g, e)
    assert torch.allclose(a[:, 0].cpu(), torch.tensor([1, .5, 1 / 3, .5, 1 / 3, 1 / 3]), atol=1e-6)
    a = dgl.edge_softmax(g, e, norm_by="src")
    assert torch.allclose(a[:, 0].cpu(), torch.tensor([1 / 3, 1 / 3, 1 / 3, .5, .5, 1.]), atol=1e-6)
    a = dgl.edge_softmax(g, e[:4], torch.tensor([0, 1, 2, 3]))
    assert torch.allclose(a[:, 0].cpu(), torch.tensor([1., .5, 1., .5]), atol=1e-6)
    hg = dgl.heterograph({("user", "follows", "user"): ([0, 0, 1], [0, 1, 2]),
                          ("developer", "develops", "game"): ([0, 1], [0, 1])}, device=dev)
    res = dgl.edge_softmax(hg, {("user", "follows", "user"): torch.ones(3, 1, device=dev),
                                ("developer", "develops", "game"): torch.ones(2, 1, device=dev)},
                           norm_by="src")
    assert torch.allclose(res[("user", "follows", "user")][:, 0].cpu(), torch.tensor([.5, .5, 1.]))
    assert torch.allclose(res[("developer", "develops", "game")][:, 0].cpu(), torch.tensor([1., 1.]))


def test_edge_softmax_unidirectional(dev):
    import dgl_amd as dgl

    g = dgl.heterograph({("A", "AB", "B"): ([1, 2, 3] * 3, [0, 0, 0, 1, 1, 1, 2, 2, 2]),
                         ("B", "BB", "B"): ([0, 1, 2] * 3, [0, 0, 0, 1, 1, 1, 2, 2, 2])}, device=dev)
    res = dgl.edge_softmax(g, {"AB": torch.ones(9, device=dev) * 2, "BB": torch.ones(9, device=dev)})
    ab, bb = res[("A", "AB", "B")], res[("B", "BB", "B")]
    assert torch.allclose(ab, torch.full_like(ab, math.exp(2) / ((math.exp(2) + math.exp(1)) * 3)))
    assert torch.allclose(bb, torch.full_like(bb, math.exp(1) / ((math.exp(2) + math.exp(1)) * 3)))


@pytest.mark.parametrize("norm_by", ["src", "dst"])
@pytest.mark.parametrize("shp", [(1,), (8, 1), (3, 4)])
@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_edge_softmax_clique_vs_dense(dev, norm_by, shp, idtype):
    import dgl_amd as dgl

    n = 7
    src = torch.arange(n).repeat_interleave(n)
    dst = torch.arange(n).repeat(n)
    g = dgl.graph((src, dst), idtype=idtype, device=dev)
    torch.manual_seed(0)
    e1 = torch.rand((n * n,) + shp, device=dev, dtype=torch.float64, requires_grad=True)
    s1 = dgl.edge_softmax(g, e1, norm_by=norm_by)
    e2 = e1.detach().clone().requires_grad_()
    dense = e2.reshape((n, n) + shp)
    s2 = torch.softmax(dense, 1 if norm_by == "src" else 0).reshape((-1,) + shp)
    assert torch.allclose(s1, s2)
    w = torch.rand_like(s1)
    (s1 * w).sum().backward()
    (s2 * w).sum().backward()
    assert torch.allclose(e1.grad, e2.grad)


# ---------------------------------------------------------------------------------------
# update_all / apply_edges
# ---------------------------------------------------------------------------------------
def test_update_all_docstring_hetero(dev):
    import dgl_amd as dgl
    import dgl_amd.function as fn

    g = dgl.heterograph({("user", "follows", "user"): ([0, 1], [1, 1]),
                         ("game", "attracts", "user"): ([0], [1])}, device=dev)
    g.nodes["user"].data["h"] = torch.tensor([[1.], [2.]], device=dev)
    g.nodes["game"].data["h"] = torch.tensor([[1.]], device=dev)
    g.update_all(fn.copy_u("h", "m"), fn.sum("m", "h"))
    assert torch.equal(g.nodes["user"].data["h"].cpu(), torch.tensor([[0.], [4.]]))
    # per-relation call + cross reducer give the same
    g.nodes["user"].data["h"] = torch.tensor([[1.], [2.]], device=dev)
    g.multi_update_all({"follows": (fn.copy_u("h", "m"), fn.sum("m", "h")),
                        "attracts": (fn.copy_u("h", "m"), fn.sum("m", "h"))}, "sum")
    assert torch.equal(g.nodes["user"].data["h"].cpu(), torch.tensor([[0.], [4.]]))


def test_update_all_star_and_zero_degree(dev):
    import dgl_amd as dgl
    import dgl_amd.function as fn

    # function/test_basics.py:389-416 — nodes 1..4 -> 0
    g = dgl.graph((torch.tensor([1, 2, 3, 4]), torch.tensor([0, 0, 0, 0])), num_nodes=5, device=dev)
    h = torch.arange(10, dtype=torch.float32, device=dev).reshape(5, 2)
    for red, want0 in (("sum", h[1:].sum(0)), ("max", h[4]), ("min", h[1]), ("mean", h[1:].mean(0))):
        g.ndata["h"] = h
        g.update_all(fn.copy_u("h", "m"), getattr(fn, red)("m", "o"))
        assert torch.allclose(g.ndata["o"][0], want0)
        assert torch.equal(g.ndata["o"][1:], torch.zeros(4, 2, device=dev))  # inf replaced by 0
    # raw operator keeps the infinities (python/dgl/heterograph.py:5115-5122 vs ops)
    o = dgl.ops.copy_u_max(g, h)
    assert torch.isneginf(o[1:]).all()


def test_update_all_unfused_message(dev):
    """u_mul_v has no fused SpMM: messages are materialised by g-SDDMM, then copy_e + reduce
    (python/dgl/core.py:399-413)."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    g = make_graph("homo", dev, torch.int64)
    torch.manual_seed(3)
    g.ndata["x"] = torch.rand(30, 4, device=dev)
    g.edata["w"] = torch.rand(100, 4, device=dev)
    g.update_all(fn.u_mul_v("x", "x", "m"), fn.sum("m", "o1"))
    s, d = g.edges()
    ref = torch.zeros(30, 4, device=dev).index_add_(0, d.long(), g.ndata["x"][s.long()] * g.ndata["x"][d.long()])
    assert torch.allclose(g.ndata["o1"], ref, rtol=1e-5, atol=1e-6)
    g.apply_edges(fn.u_add_v("x", "x", "s"))
    assert torch.equal(g.edata["s"], g.ndata["x"][s.long()] + g.ndata["x"][d.long()])
    g.apply_edges(fn.e_dot_v("w", "x", "dp"))
    assert torch.allclose(g.edata["dp"], (g.edata["w"] * g.ndata["x"][d.long()]).sum(-1, keepdim=True))
    with g.local_scope():
        g.ndata["tmp"] = torch.ones(30, 1, device=dev)
    assert "tmp" not in g.ndata
    # a user-defined message function takes the reference's plain-tensor route (dgl_amd/udf.py) next to a built-in reduce
    g.update_all(lambda edges: {"m": edges.src["x"]}, fn.sum("m", "o"))
    g.update_all(fn.copy_u("x", "m"), fn.sum("m", "o_builtin"))
    assert torch.allclose(g.ndata["o"], g.ndata["o_builtin"], rtol=1e-5, atol=1e-6)
    with pytest.raises(dgl.DGLError, match="return a dict"):
        g.update_all(lambda edges: edges.src["x"], fn.sum("m", "o"))


def test_gat_layer_pipeline(dev):
    """The GATConv call pattern (nn/pytorch/conv/gatconv.py:332-346): u_add_v -> leaky_relu ->
    edge_softmax -> u_mul_e + sum, forward and backward against plain PyTorch."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    n, m, H, D = 200, 3000, 8, 16
    g = dgl.rand_graph(n, m, device=dev, seed=5)
    s, d = [t.long() for t in g.edges()]
    torch.manual_seed(1)
    ft = torch.randn(n, H, D, device=dev, requires_grad=True)
    al = torch.randn(1, H, D, device=dev, requires_grad=True)
    ar = torch.randn(1, H, D, device=dev, requires_grad=True)

    def ours():
        with g.local_scope():
            g.srcdata.update({"ft": ft, "el": (ft * al).sum(-1, keepdim=True)})
            g.dstdata.update({"er": (ft * ar).sum(-1, keepdim=True)})
            g.apply_edges(fn.u_add_v("el", "er", "e"))
            e = torch.nn.functional.leaky_relu(g.edata.pop("e"), 0.2)
            g.edata["a"] = dgl.edge_softmax(g, e)
            g.update_all(fn.u_mul_e("ft", "a", "m"), fn.sum("m", "ft"))
            return g.dstdata["ft"]

    def ref():
        el, er = (ft * al).sum(-1, keepdim=True), (ft * ar).sum(-1, keepdim=True)
        e = torch.nn.functional.leaky_relu(el[s] + er[d], 0.2)
        mx = torch.full((n, H, 1), -float("inf"), device=dev).scatter_reduce(
            0, d.view(-1, 1, 1).expand_as(e), e, "amax", include_self=True)
        ex = torch.exp(e - mx[d])
        den = torch.zeros(n, H, 1, device=dev).index_add_(0, d, ex)
        a = ex / den[d]
        return torch.zeros(n, H, D, device=dev).index_add_(0, d, ft[s] * a)

    o1, o2 = ours(), ref()
    assert torch.allclose(o1, o2, rtol=1e-4, atol=1e-5)
    w = torch.randn_like(o1)
    g1 = torch.autograd.grad((o1 * w).sum(), [ft, al, ar])
    g2 = torch.autograd.grad((o2 * w).sum(), [ft, al, ar])
    for a_, b_ in zip(g1, g2):
        assert torch.allclose(a_, b_, rtol=1e-3, atol=1e-4)


def test_graphconv_path_graph_equals_dense(dev):
    # tests/python/pytorch/nn/test_nn.py:40-75
    import dgl_amd as dgl
    import dgl_amd.function as fn

    g = dgl.graph((torch.tensor([0, 1]), torch.tensor([1, 2])), num_nodes=3, device=dev)
    A = torch.zeros(3, 3, device=dev)
    A[[1, 2], [0, 1]] = 1
    torch.manual_seed(0)
    X, W, b = torch.rand(3, 5, device=dev), torch.rand(5, 2, device=dev), torch.rand(2, device=dev)
    g.ndata["h"] = X @ W
    g.update_all(fn.copy_u("h", "m"), fn.sum("m", "h"))
    assert torch.allclose(g.ndata["h"] + b, A @ X @ W + b, atol=1e-6)


def test_kernels_run_on_current_stream(dev):
    # tests/python/pytorch/test_ffi-stream.py:45-64
    import dgl_amd as dgl

    g = dgl.rand_graph(5000, 100000, device=dev, seed=9)
    x = torch.rand(5000, 64, device=dev)
    ref = dgl.ops.copy_u_sum(g, x)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        xs = torch.rand(5000, 64, device=dev)  # produced on stream s, consumed by the kernel on s
        out = dgl.ops.copy_u_sum(g, xs)
        ref2 = torch.zeros_like(out).index_add_(0, g.edges()[1].long(), xs[g.edges()[0].long()])
    s.synchronize()
    assert torch.allclose(out, ref2, rtol=1e-5, atol=1e-5)
    assert ref.shape == out.shape


def test_formats_select_kernels(dev):
    """formats(['coo']) forces the COO (atomic) SpMM, formats(['csr']) the CSR SDDMM
    (src/array/kernel.cc:26-43,230-247); results agree with the default path."""
    import dgl_amd as dgl

    g = make_graph("bipartite", dev, torch.int32)
    torch.manual_seed(2)
    x = torch.rand(30, 6, device=dev)
    y = torch.rand(40, 6, device=dev)
    w = torch.rand(300, 6, device=dev)
    base = dgl.ops.u_mul_e_sum(g, x, w)
    gc = g.formats(["coo"])
    assert torch.allclose(dgl.ops.u_mul_e_sum(gc, x, w), base, rtol=1e-5, atol=1e-6)
    assert torch.equal(dgl.ops.u_mul_e_max(gc, x, w), dgl.ops.u_mul_e_max(g, x, w))
    gr = g.formats(["csr"])
    assert torch.equal(dgl.ops.u_add_v(gr, x, y), dgl.ops.u_add_v(g, x, y))
    with pytest.raises(dgl.DGLError, match="SpMM only supports CSC and COO"):
        dgl.ops.copy_u_sum(gr, x)


def test_half_through_api(dev):
    # test_ops.py:184-223
    import dgl_amd as dgl

    g = dgl.graph((torch.arange(1, 901), torch.zeros(900, dtype=torch.int64)), num_nodes=901, device=dev)
    torch.manual_seed(0)
    for dt, rtol, atol in ((torch.float16, 1e-3, 0.5), (torch.bfloat16, 4e-3, 2.0)):
        x = (torch.rand(901, 32, device=dev) + 1).to(dt).requires_grad_()
        w = (torch.rand(900, 32, device=dev) + 1).to(dt).requires_grad_()
        out = dgl.ops.u_mul_e_sum(g, x, w)
        ref = (x.float()[1:] * w.float()).sum(0)
        assert torch.allclose(out[0].float(), ref, rtol=rtol, atol=atol)
        out.float().sum().backward()
        assert torch.allclose(x.grad[1:].float(), w.float(), rtol=rtol, atol=atol)
        assert torch.allclose(w.grad.float(), x.float()[1:], rtol=rtol, atol=atol)


def test_hetero_update_all_max_tracks_relations(dev):
    import dgl_amd as dgl
    import dgl_amd.function as fn
    from dgl_amd.sparse_kernels import _gspmm_hetero

    g = dgl.heterograph({("a", "r1", "c"): ([0, 1, 2], [0, 0, 1]),
                         ("b", "r2", "c"): ([0, 1], [0, 2])}, device=dev)
    ha = torch.tensor([[1., 9.], [5., 2.], [3., 3.]], device=dev)
    hb = torch.tensor([[4., 10.], [7., 1.]], device=dev)
    g.nodes["a"].data["h"], g.nodes["b"].data["h"] = ha, hb
    g.update_all(fn.copy_u("h", "m"), fn.max("m", "o"))
    assert torch.equal(g.nodes["c"].data["o"].cpu(), torch.tensor([[5., 10.], [3., 3.], [7., 1.]]))
    u = [None] * 3
    u[g.get_ntype_id("a")], u[g.get_ntype_id("b")] = ha, hb
    out, (au, ae, ant, aet) = _gspmm_hetero(g._graph, "copy_lhs", "max", 3, tuple(u))
    c = g.get_ntype_id("c")
    assert torch.equal(au[c].cpu(), torch.tensor([[1, 0], [2, 2], [1, 1]], dtype=au[c].dtype))
    assert torch.equal(ant[c].cpu(), torch.tensor([[0, 1], [0, 0], [1, 1]], dtype=ant[c].dtype))


@pytest.mark.parametrize("tdtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("op", ["copy_lhs", "mul", "copy_rhs"])
def test_fused_mean_equals_sum_then_divide(dev, tdtype, op):
    """reduce 'mean' divides inside the kernel (DGLA_MEAN); values must be the bits the
    reference's composition produces — sum in the storage type, then / clamp(in_degree, 1)
    (python/dgl/ops/spmm.py:109-114) — including rows longer than a merge unit, isolated
    nodes, and the gradient."""
    import dgl_amd as dgl

    rng = np.random.default_rng(3)
    n_src, n_dst, e = 400, 300, 9000
    src = rng.integers(0, n_src, e)
    dst = np.minimum((rng.random(e) ** 3 * n_dst).astype(np.int64), n_dst - 1)   # hub rows + empties
    g = dgl.heterograph({("a", "r", "b"): (torch.from_numpy(src), torch.from_numpy(dst))},
                        {"a": n_src, "b": n_dst}, device=dev)
    torch.manual_seed(0)
    x = (torch.rand(n_src, 4, 8, device=dev) + 0.5).to(tdtype)
    w = (torch.rand(e, 4, 1, device=dev) + 0.5).to(tdtype)
    lhs = None if op == "copy_rhs" else x.clone().requires_grad_(True)
    rhs = None if op == "copy_lhs" else w.clone().requires_grad_(True)
    got = dgl.ops.gspmm(g, op, "mean", lhs, rhs)
    lhs2 = None if lhs is None else lhs.detach().clone().requires_grad_(True)
    rhs2 = None if rhs is None else rhs.detach().clone().requires_grad_(True)
    s = dgl.ops.gspmm(g, op, "sum", lhs2, rhs2)
    deg = g.in_degrees().to(tdtype).clamp(min=1).reshape(-1, 1, 1)
    want = s / deg
    assert got.dtype == want.dtype and got.shape == want.shape
    assert torch.equal(got.view(torch.int16 if got.element_size() == 2 else
                                (torch.int32 if got.element_size() == 4 else torch.int64)),
                       want.view(torch.int16 if got.element_size() == 2 else
                                 (torch.int32 if got.element_size() == 4 else torch.int64)))
    if tdtype in (torch.float32, torch.float64):
        wgt = torch.rand_like(got)
        ins = [t for t in (lhs, rhs) if t is not None]
        ins2 = [t for t in (lhs2, rhs2) if t is not None]
        g1 = torch.autograd.grad((got * wgt).sum(), ins)
        g2 = torch.autograd.grad((want * wgt).sum(), ins2)
        for a, b in zip(g1, g2):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_identity_edge_id_map_is_dropped_for_large_sorted_coo(dev, monkeypatch):
    """A COO already sorted by destination gives a CSC whose edge-id map is the identity: large
    graphs drop it at format build time (the kernels then run map-free); mini-batch-sized graphs
    skip the check (its synchronisation would cost more than the map)."""
    from dgl_amd.graph_index import Relation

    n, e = 500, 6000
    g = torch.Generator().manual_seed(3)
    dst = torch.sort(torch.randint(0, n, (e,), generator=g)).values.to(dev)
    src = torch.randint(0, n, (e,), generator=g).to(dev)
    small = Relation(n, n, row=src, col=dst, idtype=torch.int64, device=dev)
    assert small.csc()[2] is not None                      # below the threshold: map kept
    monkeypatch.setattr(Relation, "_IDENTITY_CHECK_MIN_EDGES", 1000)
    big = Relation(n, n, row=src, col=dst, idtype=torch.int64, device=dev)
    assert big.csc()[2] is None                            # identity recognised and dropped
    assert torch.equal(big.csc()[0], small.csc()[0]) and torch.equal(big.csc()[1], small.csc()[1])
    assert torch.equal(small.csc()[2], torch.arange(e, device=dev))
    shuffled = Relation(n, n, row=src.flip(0).contiguous(), col=dst.flip(0).contiguous(), idtype=torch.int64, device=dev)
    assert shuffled.csc()[2] is not None                   # not the identity: kept
