"""Built-in functions vs the SAME computation written as user-defined functions — the reference's own test oracle
(tests/python/common/ops/test_ops.py:87-181,226-299: ``dgl.rand_graph(30, 100)`` / ``rand_bipartite(30, 40, 300)``,
rtol = atol = 1e-4), forward and gradients.  The UDF side is plain torch (dgl_amd/udf.py: edge batches + degree
bucketing) and never touches ``oracle/``: a second, independent oracle for the kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = ATOL = 1e-4     # the reference's F.allclose defaults (tests/backend/pytorch/__init__.py:18)

udf_msg = {
    "add": lambda edges: {"m": edges.src["x"] + edges.data["w"]},
    "sub": lambda edges: {"m": edges.src["x"] - edges.data["w"]},
    "mul": lambda edges: {"m": edges.src["x"] * edges.data["w"]},
    "div": lambda edges: {"m": edges.src["x"] / edges.data["w"]},
    "copy_lhs": lambda edges: {"m": edges.src["x"]},
    "copy_rhs": lambda edges: {"m": edges.data["w"]},
}
udf_reduce = {
    "sum": lambda nodes: {"v": nodes.mailbox["m"].sum(1)},
    "min": lambda nodes: {"v": nodes.mailbox["m"].min(1)[0]},
    "max": lambda nodes: {"v": nodes.mailbox["m"].max(1)[0]},
}
spmm_shapes = [((1, 2, 1, 3, 1), (4, 1, 3, 1, 1)), ((3, 3), (1, 3)), ((1,), (3,)), ((3,), (1,)), ((1,), (1,)), ((), ())]
sddmm_shapes = [((1, 2, 1, 3, 1), (4, 1, 3, 1, 1)), ((5, 3, 1, 7), (1, 3, 7, 7)), ((1, 3, 3), (4, 1, 3)), ((3,), (3,)),
                ((1,), (1,))]


def _graphs(dev, idtype):
    import dgl_amd as dgl

    return [dgl.rand_graph(30, 100, seed=1).astype(idtype).to(dev),
            dgl.rand_bipartite("_U", "_E", "_V", 30, 40, 300, seed=2).astype(idtype).to(dev)]


def _close(a, b):
    return torch.allclose(a, b, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("reducer", ["sum", "min", "max"])
@pytest.mark.parametrize("msg", ["add", "sub", "mul", "div", "copy_lhs", "copy_rhs"])
def test_gspmm_equals_udf_message_passing(dev, idtype, dtype, msg, reducer):
    import dgl_amd as dgl

    rng = np.random.default_rng(12345)
    for g in _graphs(dev, idtype):
        for shp in spmm_shapes:
            hu = torch.from_numpy(rng.random((g.number_of_src_nodes(),) + shp[0]) + 1).to(dtype).to(dev)
            he = torch.from_numpy(rng.random((g.num_edges(),) + shp[1]) + 1).to(dtype).to(dev)
            u, e = hu.clone().requires_grad_(), he.clone().requires_grad_()
            v = dgl.ops.gspmm(g, msg, reducer, u, e)
            if reducer in ("max", "min"):
                v = torch.where(torch.isinf(v), torch.zeros_like(v), v)
            v.sum().backward()
            g.srcdata["x"], g.edata["w"] = hu.clone().requires_grad_(), he.clone().requires_grad_()
            g.update_all(udf_msg[msg], udf_reduce[reducer])
            v1 = g.dstdata["v"]
            assert _close(v, v1), (msg, reducer, shp)
            v1.sum().backward()
            for name, mine, theirs in (("u", u, g.srcdata["x"]), ("e", e, g.edata["w"])):
                if (name == "u" and msg == "copy_rhs") or (name == "e" and msg == "copy_lhs"):
                    continue
                if reducer in ("min", "max"):      # ties may pick another winner: the reference compares in L1
                    rate = (theirs.grad - mine.grad).abs().sum() / mine.grad.abs().sum()
                    assert rate.item() < 1e-2, (msg, reducer, shp, name, rate)
                else:
                    assert _close(theirs.grad, mine.grad), (msg, reducer, shp, name)
            for fr, k in ((g.srcdata, "x"), (g.edata, "w"), (g.dstdata, "v")):
                fr.pop(k)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("msg", ["add", "sub", "mul", "div", "dot", "copy_lhs", "copy_rhs"])
@pytest.mark.parametrize("lhs_target,rhs_target", [(a, b) for a in "uve" for b in "uve" if a != b])
def test_gsddmm_equals_udf_apply_edges(dev, idtype, msg, lhs_target, rhs_target):
    import dgl_amd as dgl

    rng = np.random.default_rng(4321)

    def binary(x, y):
        return {"add": lambda: x + y, "sub": lambda: x - y, "mul": lambda: x * y, "div": lambda: x / y,
                "dot": lambda: (x * y).sum(-1, keepdim=True), "copy_lhs": lambda: x, "copy_rhs": lambda: y}[msg]()

    def pick(edges, t):
        return {"u": edges.src, "v": edges.dst, "e": edges.data}[t]

    for g in _graphs(dev, idtype):
        sizes = {"u": g.number_of_src_nodes(), "e": g.num_edges(), "v": g.number_of_dst_nodes()}
        frames = {"u": g.srcdata, "e": g.edata, "v": g.dstdata}
        for shp in sddmm_shapes:
            fl = torch.from_numpy(rng.random((sizes[lhs_target],) + shp[0]) + 1).to(dev)
            fr = torch.from_numpy(rng.random((sizes[rhs_target],) + shp[1]) + 1).to(dev)
            lhs, rhs = fl.clone().requires_grad_(), fr.clone().requires_grad_()
            e = dgl.ops.gsddmm(g, msg, lhs, rhs, lhs_target=lhs_target, rhs_target=rhs_target)
            e.sum().backward()
            frames[lhs_target]["x"] = fl.clone().requires_grad_()
            frames[rhs_target]["y"] = fr.clone().requires_grad_()
            g.apply_edges(lambda edges: {"m": binary(pick(edges, lhs_target)["x"], pick(edges, rhs_target)["y"])})
            e1 = g.edata["m"]
            assert _close(e, e1), (msg, shp)
            e1.sum().backward()
            if msg != "copy_rhs":
                assert _close(frames[lhs_target]["x"].grad, lhs.grad), (msg, shp, "lhs")
            if msg != "copy_lhs":
                assert _close(frames[rhs_target]["y"].grad, rhs.grad), (msg, shp, "rhs")
            frames[lhs_target].pop("x")
            frames[rhs_target].pop("y")
            g.edata.pop("m")


def test_mixed_builtin_and_udf_pairs_and_apply_node_func(dev):
    """A built-in message with a UDF reduce, a UDF message with a built-in reduce, and ``apply_node_func`` after a
    fused pair (python/dgl/core.py:392-424): all equal the fully built-in result."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    g = dgl.rand_graph(200, 3000, seed=7).to(dev)
    x = torch.rand(200, 4, 5, device=dev) + 1
    w = torch.rand(3000, 4, 1, device=dev) + 1
    g.ndata["x"], g.edata["w"] = x, w
    g.update_all(fn.u_mul_e("x", "w", "m"), fn.sum("m", "ref"))
    g.update_all(fn.u_mul_e("x", "w", "m"), lambda nodes: {"a": nodes.mailbox["m"].sum(1)})
    g.update_all(lambda edges: {"m": edges.src["x"] * edges.data["w"]}, fn.sum("m", "b"))
    g.update_all(fn.u_mul_e("x", "w", "m"), fn.sum("m", "c"), lambda nodes: {"c": nodes.data["c"] * 2 + nodes.data["x"]})
    ref = g.ndata["ref"]
    assert _close(g.ndata["a"], ref) and _close(g.ndata["b"], ref)
    assert _close(g.ndata["c"], ref * 2 + x)
    # max with zero-degree rows: built-in reduce -> 0 (inf replaced), as the UDF route's zero initializer
    g.update_all(fn.copy_u("x", "m"), fn.max("m", "mx"))
    g.update_all(fn.copy_u("x", "m"), lambda nodes: {"mx2": nodes.mailbox["m"].max(1)[0]})
    assert _close(g.ndata["mx"], g.ndata["mx2"])


def test_pull_builtin_equals_pull_udf(dev):
    """``g.pull(nid, ...)`` on a node subset (test_heterograph-kernel.py:226-361 `partial=True`)."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    g = dgl.rand_graph(100, 1500, seed=3)
    g.add_edges(g.nodes(), g.nodes())                 # no zero-degree nodes, as the reference's test
    g = g.to(dev)
    nid = torch.arange(0, 100, 2, device=dev)
    hu = torch.rand(100, 5, 3, device=dev) - 0.5
    he = torch.rand(g.num_edges(), 5, 3, device=dev) + 1
    for red in ("sum", "max", "min", "mean"):
        g.ndata["u"], g.edata["e"] = hu.clone().requires_grad_(), he.clone().requires_grad_()
        g.pull(nid, fn.u_mul_e("u", "e", "m"), getattr(fn, red)("m", "r1"))
        r1 = g.ndata.pop("r1")
        r1.sum().backward()
        gu1, ge1 = g.ndata["u"].grad, g.edata["e"].grad
        g.ndata["u"], g.edata["e"] = hu.clone().requires_grad_(), he.clone().requires_grad_()
        op = {"sum": lambda m: m.sum(1), "max": lambda m: m.max(1)[0], "min": lambda m: m.min(1)[0],
              "mean": lambda m: m.mean(1)}[red]
        g.pull(nid, lambda edges: {"m": edges.src["u"] * edges.data["e"]}, lambda nodes: {"r2": op(nodes.mailbox["m"])})
        r2 = g.ndata.pop("r2")
        r2.sum().backward()
        assert _close(r1, r2), red
        assert bool((r1[1::2] == 0).all())            # rows outside nid keep the zero initializer
        assert _close(gu1, g.ndata["u"].grad) and _close(ge1, g.edata["e"].grad), red


def test_pull_udfs_see_the_original_ids(dev):
    """Inside ``pull`` the UDFs run on an extracted compute graph; ``nodes.nodes()`` and ``edges.edges()[2]`` must report
    the ORIGINAL node / edge ids (core.py:405-423 passes g.dstdata[NID] / g.edata[EID] as orig_nid / orig_eid)."""
    import dgl_amd as dgl

    g = dgl.rand_graph(60, 400, seed=5)
    g.add_edges(g.nodes(), g.nodes())
    g = g.to(dev)
    g.ndata["x"] = torch.rand(60, 3, device=dev)
    nid = torch.tensor([3, 17, 42, 59], device=dev)
    seen = {}

    def msg(edges):
        seen["eid"] = edges.edges()[2].clone()
        return {"m": edges.src["x"]}

    def red(nodes):
        seen.setdefault("nid", []).append(nodes.nodes().clone())
        return {"r": nodes.mailbox["m"].sum(1)}

    def app(nodes):
        seen["apply"] = nodes.nodes().clone()
        return {"r": nodes.data["r"] + 1}

    g.pull(nid, msg, red, app)
    _, _, want_eid = g.in_edges(nid, form="all")
    assert sorted(seen["eid"].tolist()) == sorted(want_eid.tolist())
    assert sorted(torch.cat(seen["nid"]).tolist()) == nid.tolist()
    assert sorted(seen["apply"].tolist()) == nid.tolist()
