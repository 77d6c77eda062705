"""Graph-level readout (dgl_amd/readout.py ≙ python/dgl/readout.py) over a batch of a few hundred graphs: the segment-reduce
/ segment-softmax kernels with one segment per batched graph, against per-graph torch evaluations, values and gradients.
(The reference's tests/python/common/test_readout.py runs unmodified through tools/ref_suite --suite mp.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(dev, k=300, seed=0):
    import dgl_amd as dgl

    g = torch.Generator().manual_seed(seed)
    graphs = []
    for i in range(k):
        n = int(torch.randint(1, 40, (1,), generator=g))
        e = int(torch.randint(0, 120, (1,), generator=g)) if i % 17 else 0
        graphs.append(dgl.graph((torch.randint(n, (e,), generator=g).to(dev), torch.randint(n, (e,), generator=g).to(dev)),
                                num_nodes=n))
    b = dgl.batch(graphs)
    b.ndata["h"] = torch.randn(b.num_nodes(), 7, device=dev, requires_grad=True)
    b.ndata["w"] = torch.rand(b.num_nodes(), 1, device=dev)
    b.edata["h"] = torch.randn(b.num_edges(), 3, device=dev, requires_grad=True)
    return b


@pytest.mark.parametrize("op", ["sum", "mean", "max", "min"])
def test_readout_matches_per_graph_torch(dev, op):
    import dgl_amd as dgl

    b = _batch(dev)
    for side, lens, feat in (("nodes", b.batch_num_nodes(), b.ndata["h"]), ("edges", b.batch_num_edges(), b.edata["h"])):
        got = getattr(dgl, "readout_" + side)(b, "h", op=op)
        chunks = torch.split(feat, lens.tolist())
        f = {"sum": lambda t: t.sum(0), "mean": lambda t: t.mean(0) if len(t) else t.sum(0),
             "max": lambda t: t.max(0)[0] if len(t) else t.sum(0), "min": lambda t: t.min(0)[0] if len(t) else t.sum(0)}[op]
        want = torch.stack([f(c) for c in chunks])
        assert got.shape == want.shape and torch.allclose(got, want, atol=1e-5)
        w = torch.randn_like(got)
        (g1,) = torch.autograd.grad((got * w).sum(), [feat], retain_graph=True)
        (g2,) = torch.autograd.grad((want * w).sum(), [feat])
        assert torch.allclose(g1, g2, atol=1e-5)
        if op != "min":
            assert torch.equal(getattr(dgl, "%s_%s" % (op, side))(b, "h"), got)
    wsum = dgl.sum_nodes(b, "h", "w")
    assert torch.allclose(wsum, torch.stack([c.sum(0) for c in torch.split(b.ndata["h"] * b.ndata["w"], b.batch_num_nodes().tolist())]), atol=1e-5)


def test_softmax_broadcast_and_topk(dev):
    import dgl_amd as dgl

    b = _batch(dev, 120, 1)
    lens = b.batch_num_nodes().tolist()
    sm = dgl.softmax_nodes(b, "h")
    want = torch.cat([torch.softmax(c, 0) for c in torch.split(b.ndata["h"], lens)])
    assert torch.allclose(sm, want, atol=1e-6)
    es = dgl.softmax_edges(b, "h")
    assert torch.allclose(es, torch.cat([torch.softmax(c, 0) for c in torch.split(b.edata["h"], b.batch_num_edges().tolist())]), atol=1e-6)
    gf = torch.randn(b.batch_size, 4, device=dev)
    assert torch.equal(dgl.broadcast_nodes(b, gf), torch.cat([gf[i:i + 1].expand(n, 4) for i, n in enumerate(lens)]))
    assert dgl.broadcast_edges(b, gf).shape == (b.num_edges(), 4)
    k = 5
    vals, idx = dgl.topk_nodes(b, "h", k, sortby=2)
    for i, c in enumerate(torch.split(b.ndata["h"].detach(), lens)):
        order = torch.argsort(c[:, 2], descending=True)[:k]
        assert torch.equal(vals[i, : len(order)], c[order]) and bool((vals[i, len(order):] == 0).all())
        assert torch.equal(idx[i, : len(order)], order)
    vals2, _ = dgl.topk_nodes(b, "h", k, descending=False)
    for i, c in enumerate(torch.split(b.ndata["h"].detach(), lens)):
        m = min(k, c.shape[0])
        assert torch.equal(vals2[i, :m], torch.sort(c, 0)[0][:m])
