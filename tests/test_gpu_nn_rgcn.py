"""The R-GCN side of ``dgl_amd.nn`` (BASELINE configs[4]): ``TypedLinear`` (python/dgl/nn/pytorch/linear.py), ``RelGraphConv``
(conv/relgraphconv.py), ``HeteroGraphConv`` / ``HeteroLinear`` / ``HeteroEmbedding`` (hetero.py) against plain torch
evaluations of the same formulas, values and gradients; shapes and cases follow tests/python/pytorch/nn/test_nn.py
(test_typed_linear :2041, test_rgcn :430, test_hetero_conv :1409)."""
import io

import pytest
import torch

pytestmark = pytest.mark.gpu


def _weights(lin):
    """(num_types, in, out) dense weights of a TypedLinear, differentiable."""
    if lin.regularizer == "bdd":
        w = lin.W.view(lin.num_types, lin.num_bases, lin.submat_in, lin.submat_out)
        return torch.stack([torch.block_diag(*w[t]) for t in range(lin.num_types)])
    return lin.get_weight()


@pytest.mark.parametrize("feat", [4, 32, 96])
@pytest.mark.parametrize("regularizer,num_bases", [(None, None), ("basis", 4), ("bdd", 4)])
def test_typed_linear_sorted_and_unsorted(dev, feat, regularizer, num_bases):
    import dgl_amd.nn as dglnn

    torch.manual_seed(feat)
    lin = dglnn.TypedLinear(feat, 2 * feat, 5, regularizer=regularizer, num_bases=num_bases).to(dev)
    n = 3000
    x = torch.randn(n, feat, device=dev, requires_grad=True)
    t = torch.randint(0, 5, (n,), device=dev)
    y = lin(x, t)
    want = torch.bmm(x.unsqueeze(1), _weights(lin)[t]).squeeze(1)
    assert y.shape == (n, 2 * feat) and torch.allclose(y, want, atol=1e-4, rtol=1e-4)
    ts, idx = torch.sort(t)
    ys = lin(x[idx], ts, sorted_by_type=True)
    assert torch.allclose(ys, y[idx], atol=1e-4, rtol=1e-4)
    w = torch.randn_like(y)
    params = [x] + list(lin.parameters())
    g1 = torch.autograd.grad((ys * w[idx]).sum(), params, retain_graph=True)
    g2 = torch.autograd.grad((want * w).sum(), params)
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, atol=2e-4, rtol=1e-3)
    assert "TypedLinear(in_size=%d" % feat in repr(lin)
    with pytest.raises(ValueError):
        dglnn.TypedLinear(6, 8, 3, "bdd", 4)
    with pytest.raises(ValueError):
        dglnn.TypedLinear(6, 8, 3, "basis")


@pytest.mark.parametrize("out_feat", [1, 8, 32])
@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_rel_graph_conv_matches_the_formula(dev, out_feat, idtype):
    import dgl_amd as dgl
    import dgl_amd.nn as dglnn

    torch.manual_seed(3)
    n, e, r, i = 100, 1000, 5, 10
    u, v = torch.randint(n, (e,), device=dev, dtype=idtype), torch.randint(n, (e,), device=dev, dtype=idtype)
    g = dgl.graph((u, v), num_nodes=n)
    et = torch.arange(e, device=dev) % r
    h = torch.randn(n, i, device=dev, requires_grad=True)
    norm = torch.rand(e, 1, device=dev)
    variants = [dglnn.RelGraphConv(i, out_feat, r), dglnn.RelGraphConv(i, out_feat, r, "basis", 2)]
    if out_feat % 2 == 0:
        variants.append(dglnn.RelGraphConv(i, out_feat, r, "bdd", 2))
    for conv in variants:
        conv = conv.to(dev)
        torch.save(conv, io.BytesIO())                               # pickles, like the reference's test
        for nrm in (None, norm):
            out = conv(g, h, et, nrm)
            m = torch.bmm(h[u.long()].unsqueeze(1), _weights(conv.linear_r)[et]).squeeze(1)
            if nrm is not None:
                m = m * nrm
            want = torch.zeros(n, out_feat, device=dev).index_add(0, v.long(), m) + conv.h_bias + h @ conv.loop_weight
            assert out.shape == (n, out_feat) and torch.allclose(out, want, atol=1e-4, rtol=1e-4)
            w = torch.randn_like(out)
            params = [h] + list(conv.parameters())
            g1 = torch.autograd.grad((out * w).sum(), params, retain_graph=True, allow_unused=True)
            g2 = torch.autograd.grad((want * w).sum(), params, allow_unused=True)
            for a, b in zip(g1, g2):
                assert (a is None and b is None) or torch.allclose(a, b, atol=2e-4, rtol=1e-3)
        # edges sorted by type: the segment_mm route gives the same layer
        st, perm = torch.sort(et)
        gs = dgl.graph((u[perm], v[perm]), num_nodes=n)
        assert torch.allclose(conv(gs, h, st, norm[perm], presorted=True), conv(g, h, et, norm), atol=1e-4, rtol=1e-4)
    plain = dglnn.RelGraphConv(i, out_feat, r, bias=False, self_loop=False, activation=torch.relu, layer_norm=True,
                               dropout=0.0).to(dev)
    m = torch.bmm(h[u.long()].unsqueeze(1), plain.linear_r.W[et]).squeeze(1)
    want = torch.relu(plain.layer_norm_weight(torch.zeros(n, out_feat, device=dev).index_add(0, v.long(), m)))
    assert torch.allclose(plain(g, h, et), want, atol=1e-4, rtol=1e-4)
    assert dglnn.RelGraphConv(i, 40, r, "basis").linear_r.num_bases == r        # default number of bases


def _myagg(alist, dsttype):
    rst = alist[0]
    for k in range(1, len(alist)):
        rst = rst + (k + 1) * alist[k]
    return rst


@pytest.mark.parametrize("agg", ["sum", "max", "min", "mean", "stack", _myagg])
@pytest.mark.parametrize("canonical_keys", [False, True])
def test_hetero_graph_conv(dev, agg, canonical_keys):
    import dgl_amd as dgl
    import dgl_amd.nn as dglnn

    t = lambda x: torch.tensor(x, device=dev)
    g = dgl.heterograph({("user", "follows", "user"): (t([0, 0, 2, 1]), t([1, 2, 1, 3])),
                         ("user", "plays", "game"): (t([0, 0, 0, 1, 2]), t([0, 2, 3, 0, 2])),
                         ("store", "sells", "game"): (t([0, 0, 1, 1]), t([0, 3, 1, 2]))})
    keys = {"follows": ("user", "follows", "user"), "plays": ("user", "plays", "game"), "sells": ("store", "sells", "game")}
    dims = {"follows": (2, 3), "plays": (2, 4), "sells": (3, 4)}
    torch.manual_seed(0)
    mods = {(keys[k] if canonical_keys else k): dglnn.GraphConv(*dims[k], allow_zero_in_degree=True) for k in dims}
    conv = dglnn.HeteroGraphConv(mods, agg).to(dev)
    torch.save(conv, io.BytesIO())
    uf, gf, sf = torch.randn(4, 2, device=dev), torch.randn(4, 4, device=dev), torch.randn(2, 3, device=dev)
    h = conv(g, {"user": uf, "game": gf, "store": sf})
    assert set(h) == {"user", "game"}
    parts = {k: conv._get_module(keys[k])(g[keys[k]], ({"user": uf, "store": sf}[keys[k][0]], {"user": uf, "game": gf}[keys[k][2]]))
             for k in dims}
    if agg == "stack":
        assert h["user"].shape == (4, 1, 3) and h["game"].shape == (4, 2, 4)
        # relations are visited in canonical-edge-type order: ("store", "sells", "game") before ("user", "plays", "game")
        assert torch.allclose(h["game"], torch.stack([parts["sells"], parts["plays"]], 1), atol=1e-6)
    else:
        assert h["user"].shape == (4, 3) and h["game"].shape == (4, 4)
        both = torch.stack([parts["plays"], parts["sells"]])
        want = {"sum": both.sum(0), "max": both.max(0)[0], "min": both.min(0)[0], "mean": both.mean(0)}.get(
            agg if isinstance(agg, str) else "", parts["sells"] + 2 * parts["plays"])
        assert torch.allclose(h["game"], want, atol=1e-6) and torch.allclose(h["user"], parts["follows"], atol=1e-6)
    # a block, with (source, destination) inputs; and a source type without input is skipped
    blk = dgl.to_block(g, {"user": t([0, 1, 2, 3]), "game": t([0, 1, 2, 3]), "store": t([]).long()})
    hb = conv(blk, ({"user": uf, "game": gf, "store": sf}, {"user": uf, "game": gf, "store": sf[0:0]}))
    assert set(hb) == {"user", "game"}
    for k in ("user", "game"):
        assert torch.allclose(hb[k], h[k], atol=1e-5)
    hb2 = conv(blk, {"user": uf, "game": gf, "store": sf})
    assert all(torch.allclose(hb2[k], h[k], atol=1e-5) for k in ("user", "game"))
    h3 = conv(g, {"user": uf, "game": gf})
    assert set(h3) == {"user", "game"} and (h3["game"].shape == ((4, 1, 4) if agg == "stack" else (4, 4)))
    with pytest.raises(dgl.DGLError):
        dglnn.HeteroGraphConv(mods, "median")


def test_hetero_linear_and_embedding(dev):
    import dgl_amd.nn as dglnn

    lin = dglnn.HeteroLinear({"user": 1, ("user", "follows", "user"): 2}, 3).to(dev)
    out = lin({"user": torch.randn(2, 1, device=dev), ("user", "follows", "user"): torch.randn(3, 2, device=dev)})
    assert out["user"].shape == (2, 3) and out[("user", "follows", "user")].shape == (3, 3)
    emb = dglnn.HeteroEmbedding({"user": 2, ("user", "follows", "user"): 3}, 4).to(dev)
    assert set(emb.weight) == {"user", ("user", "follows", "user")} and emb.weight["user"].shape == (2, 4)
    e = emb({"user": torch.tensor([0, 1, 1], device=dev), ("user", "follows", "user"): torch.tensor([2], device=dev)})
    assert e["user"].shape == (3, 4) and e[("user", "follows", "user")].shape == (1, 4)
