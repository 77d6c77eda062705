"""dgl_amd.nn — GraphConv / SAGEConv / GATConv as callers of the hot path (BASELINE configs[0], [3], [2]) against dense
torch evaluations of the formulas in python/dgl/nn/pytorch/conv/{graphconv,sageconv,gatconv}.py, forward and
gradients; GATConv's attention block inside the opt-in hand-off scope ≡ the plain path, nothing tagged leaves it."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _graph(dev, n=600, e=9000, seed=0, self_loops=True):
    import dgl_amd as dgl

    g0 = torch.Generator().manual_seed(seed)
    src, dst = torch.randint(0, n, (e,), generator=g0), torch.randint(0, n, (e,), generator=g0)
    if self_loops:
        src, dst = torch.cat([src, torch.arange(n)]), torch.cat([dst, torch.arange(n)])
    perm = torch.randperm(src.numel(), generator=g0)          # unsorted COO: the CSC carries DGL's usual edge-id map
    src, dst = src[perm].to(dev), dst[perm].to(dev)
    return dgl.graph((src, dst), num_nodes=n, idtype=torch.int32, device=dev), src.long(), dst.long()


def _dense_gat(layer, x, src, dst, n, edge_weight=None):
    h, d = layer._num_heads, layer._out_feats
    f = layer.fc(x).view(n, h, d)
    el, er = (f * layer.attn_l).sum(-1), (f * layer.attn_r).sum(-1)
    e = F.leaky_relu(el[src] + er[dst], layer.leaky_relu.negative_slope)
    mx = torch.full((n, h), float("-inf"), device=x.device, dtype=e.dtype).index_reduce_(0, dst, e, "amax", include_self=True)
    ex = torch.exp(e - mx[dst])
    a = ex / torch.zeros(n, h, device=x.device, dtype=e.dtype).index_add_(0, dst, ex)[dst]
    aw = a if edge_weight is None else a * edge_weight.unsqueeze(-1)
    out = torch.zeros(n, h, d, device=x.device, dtype=e.dtype).index_add_(0, dst, aw.unsqueeze(-1) * f[src])
    if layer.res_fc is not None:
        out = out + layer.res_fc(x).view(n, -1, d)
    if layer.has_explicit_bias:
        out = out + layer.bias.view(1, h, d)
    return out, aw.unsqueeze(-1)


@pytest.mark.parametrize("handoff", [True, False])
@pytest.mark.parametrize("residual,with_weight", [(False, False), (True, True)])
def test_gatconv_matches_dense_and_keeps_tagged_tensors_inside(dev, handoff, residual, with_weight):
    import dgl_amd as dgl
    from dgl_amd import edge_order as E

    g, src, dst = _graph(dev)
    n = g.num_nodes()
    torch.manual_seed(1)
    layer = dgl.nn.GATConv(24, 8, num_heads=4, residual=residual).to(dev)
    layer.handoff = handoff
    x = torch.randn(n, 24, device=dev, requires_grad=True)
    ew = (torch.rand(g.num_edges(), device=dev) + 0.5) if with_weight else None
    up = torch.randn(n, 4, 8, device=dev)
    out, att = layer(g, x, edge_weight=ew, get_attention=True)
    assert type(out) is torch.Tensor and type(att) is torch.Tensor and not E.handoff_enabled()
    assert att.shape == (g.num_edges(), 4, 1)
    (out * up).sum().backward()
    got = [out.detach(), att.detach(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    x.grad = None
    layer.zero_grad()
    want_out, want_att = _dense_gat(layer, x, src, dst, n, ew)
    (want_out * up).sum().backward()
    want = [want_out.detach(), want_att.detach(), x.grad] + [p.grad for p in layer.parameters()]
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5)
    assert "a" not in g.edata and "ft" not in g.ndata                      # local_scope: the graph's frames are untouched


def test_gatconv_zero_in_degree_and_block_input(dev):
    import dgl_amd as dgl

    g, _, _ = _graph(dev, n=50, e=40, seed=3, self_loops=False)
    layer = dgl.nn.GATConv(6, 3, num_heads=2).to(dev)
    with pytest.raises(dgl.DGLError, match="0-in-degree"):
        layer(g, torch.randn(50, 6, device=dev))
    layer.set_allow_zero_in_degree(True)
    assert layer(g, torch.randn(50, 6, device=dev)).shape == (50, 2, 3)
    # a message-flow block: destination nodes are the first rows of the source features
    src = torch.tensor([0, 1, 2, 3, 4, 5, 2, 3], device=dev)
    dst = torch.tensor([0, 1, 2, 0, 1, 2, 0, 1], device=dev)
    blk = dgl.create_block((src, dst), num_src_nodes=6, num_dst_nodes=3, device=dev)
    assert layer(blk, torch.randn(6, 6, device=dev)).shape == (3, 2, 3)


@pytest.mark.parametrize("norm", ["both", "right", "left", "none"])
@pytest.mark.parametrize("fin,fout", [(40, 8), (8, 40)])
def test_graphconv_matches_dense(dev, norm, fin, fout):
    import dgl_amd as dgl

    g, src, dst = _graph(dev, seed=2)
    n = g.num_nodes()
    torch.manual_seed(2)
    layer = dgl.nn.GraphConv(fin, fout, norm=norm).to(dev)
    with torch.no_grad():
        layer.bias.uniform_(-1, 1)
    x = torch.randn(n, fin, device=dev, requires_grad=True)
    out = layer(g, x)
    out.square().sum().backward()
    gx = x.grad.clone()
    x.grad = None
    A = torch.zeros(n, n, device=dev).index_put_((dst, src), torch.ones(src.numel(), device=dev), accumulate=True)
    dout, din = A.sum(0).clamp(min=1), A.sum(1).clamp(min=1)
    left = {"both": dout.pow(-0.5), "left": 1 / dout}.get(norm, torch.ones_like(dout))
    right = {"both": din.pow(-0.5), "right": 1 / din}.get(norm, torch.ones_like(din))
    want = (right.unsqueeze(1) * (A @ (x * left.unsqueeze(1)))) @ layer.weight + layer.bias
    torch.testing.assert_close(out, want, rtol=2e-4, atol=2e-4)
    want.square().sum().backward()
    torch.testing.assert_close(gx, x.grad, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("agg", ["mean", "gcn", "pool"])
def test_sageconv_on_a_block_matches_dense(dev, agg):
    import dgl_amd as dgl

    n_src, n_dst, e = 400, 120, 3000
    g0 = torch.Generator().manual_seed(4)
    src = torch.randint(0, n_src, (e,), generator=g0).to(dev)
    dst = torch.randint(0, n_dst, (e,), generator=g0).to(dev)
    blk = dgl.create_block((src, dst), num_src_nodes=n_src, num_dst_nodes=n_dst, device=dev)
    torch.manual_seed(5)
    layer = dgl.nn.SAGEConv(16, 10, agg).to(dev)
    x = torch.randn(n_src, 16, device=dev)
    out = layer(blk, x)
    A = torch.zeros(n_dst, n_src, device=dev).index_put_((dst, src), torch.ones(e, device=dev), accumulate=True)
    deg = A.sum(1)
    if agg == "mean":
        want = layer.fc_self(x[:n_dst]) + layer.fc_neigh((A @ x) / deg.clamp(min=1).unsqueeze(1))
    elif agg == "gcn":
        want = layer.fc_neigh((A @ x + x[:n_dst]) / (deg + 1).unsqueeze(1)) + layer.bias
    else:
        p = F.relu(layer.fc_pool(x))
        mx = torch.full((n_dst, 16), float("-inf"), device=dev).index_reduce_(0, dst, p[src], "amax", include_self=True)
        mx = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
        want = layer.fc_self(x[:n_dst]) + layer.fc_neigh(mx)
    torch.testing.assert_close(out, want, rtol=2e-4, atol=2e-4)
