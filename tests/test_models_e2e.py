"""End-to-end model-level parity on the BASELINE.json config shapes (the callers of the hot
path, python/dgl/nn/pytorch/conv/{graphconv,sageconv}.py written out with the operator API):

* configs[0]: 2-layer GraphConv (norm='both') on a Cora-shaped graph (2 708 nodes, 10 556
  edges, 1 433 -> 16 -> 7), a few SGD steps; losses and weights against the same network on
  a dense normalised adjacency in plain PyTorch.
* GraphSAGE-mean layer on a sampled block (configs[3] shape in miniature): a rectangular
  block built with create_block, mean aggregation through copy_u + mean.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cora_like(dev, n=2708, e=10556, seed=0):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e // 2)
    dst = rng.integers(0, n, e // 2)
    # symmetric like the citation graph, plus self loops (GraphConv's usual preprocessing)
    s = np.concatenate([src, dst, np.arange(n)])
    d = np.concatenate([dst, src, np.arange(n)])
    return torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev), n


def test_two_layer_graphconv_cora_training_matches_dense(dev):
    import dgl_amd as dgl
    import dgl_amd.function as fn

    s, d, n = _cora_like(dev)
    g = dgl.graph((s, d), num_nodes=n, device=dev)
    f_in, hid, classes = 1433, 16, 7
    torch.manual_seed(0)
    x = (torch.rand(n, f_in, device=dev) < 0.02).float()          # sparse bag-of-words rows
    y = torch.randint(0, classes, (n,), device=dev)
    w_init = [torch.randn(f_in, hid, device=dev) * 0.05, torch.zeros(hid, device=dev),
              torch.randn(hid, classes, device=dev) * 0.3, torch.zeros(classes, device=dev)]

    out_deg = g.out_degrees().float().clamp(min=1)
    in_deg = g.in_degrees().float().clamp(min=1)

    def graphconv(h, w, b):
        # graphconv.py:405-457 with norm='both': D_out^-1/2 on the source side, aggregate,
        # D_in^-1/2 on the destination side; weight first when it shrinks the feature
        h = h * out_deg.pow(-0.5).unsqueeze(-1)
        if w.shape[0] > w.shape[1]:
            h = h @ w
        with g.local_scope():
            g.srcdata["h"] = h
            g.update_all(fn.copy_u("h", "m"), fn.sum("m", "h"))
            h = g.dstdata["h"]
        if w.shape[0] <= w.shape[1]:
            h = h @ w
        return h * in_deg.pow(-0.5).unsqueeze(-1) + b

    A = torch.zeros(n, n, device=dev)
    A.index_put_((d, s), torch.ones(s.numel(), device=dev), accumulate=True)
    A_hat = in_deg.pow(-0.5).unsqueeze(1) * A * out_deg.pow(-0.5).unsqueeze(0)

    def dense_conv(h, w, b):
        return A_hat @ (h @ w) + b

    def train(conv):
        ws = [w.clone().requires_grad_(True) for w in w_init]
        losses = []
        for _ in range(5):
            h = torch.relu(conv(x, ws[0], ws[1]))
            logits = conv(h, ws[2], ws[3])
            loss = torch.nn.functional.cross_entropy(logits, y)
            losses.append(float(loss.detach()))
            grads = torch.autograd.grad(loss, ws)
            with torch.no_grad():
                for w, gr in zip(ws, grads):
                    w -= 0.5 * gr
        return losses, ws

    l1, w1 = train(graphconv)
    l2, w2 = train(dense_conv)
    np.testing.assert_allclose(l1, l2, rtol=2e-5)
    assert l1[-1] < l1[0]
    for a, b in zip(w1, w2):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_sage_mean_on_block_matches_dense(dev):
    """SAGEConv 'mean' on a message-flow block (sageconv.py:215-267): h_dst' = W_self h_dst +
    W_neigh mean_{u in N(v)} h_u, rectangular graph from create_block; forward + gradients."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    rng = np.random.default_rng(3)
    n_src, n_dst, fan = 5000, 1024, 15
    dst = np.repeat(np.arange(n_dst), fan)
    src = rng.integers(0, n_src, n_dst * fan)
    # a few destination nodes without sampled neighbours
    keep = dst >= 8
    blk = dgl.create_block((torch.from_numpy(src[keep]), torch.from_numpy(dst[keep])),
                           num_src_nodes=n_src, num_dst_nodes=n_dst, device=dev)
    torch.manual_seed(1)
    h = torch.randn(n_src, 100, device=dev, requires_grad=True)
    w_self = torch.randn(100, 64, device=dev, requires_grad=True)
    w_neigh = torch.randn(100, 64, device=dev, requires_grad=True)

    def ours():
        with blk.local_scope():
            blk.srcdata["h"] = h
            blk.update_all(fn.copy_u("h", "m"), fn.mean("m", "neigh"))
            return h[:n_dst] @ w_self + blk.dstdata["neigh"] @ w_neigh

    def ref():
        s_t, d_t = torch.from_numpy(src[keep]).to(dev), torch.from_numpy(dst[keep]).to(dev)
        agg = torch.zeros(n_dst, 100, device=dev).index_add_(0, d_t, h[s_t])
        deg = torch.bincount(d_t, minlength=n_dst).clamp(min=1).unsqueeze(-1)
        return h[:n_dst] @ w_self + (agg / deg) @ w_neigh

    o1, o2 = ours(), ref()
    assert torch.allclose(o1, o2, rtol=1e-4, atol=1e-4)
    assert "neigh" not in blk.dstdata  # local scope left no trace
    wgt = torch.randn_like(o1)
    g1 = torch.autograd.grad((o1 * wgt).sum(), [h, w_self, w_neigh])
    g2 = torch.autograd.grad((o2 * wgt).sum(), [h, w_self, w_neigh])
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3)


def _random_graph(dev, n=600, e=9000, seed=0):
    import dgl_amd as dgl

    g0 = torch.Generator().manual_seed(seed)
    src, dst = torch.randint(0, n, (e,), generator=g0), torch.randint(0, n, (e,), generator=g0)
    src, dst = torch.cat([src, torch.arange(n)]), torch.cat([dst, torch.arange(n)])     # + self loops: no empty row
    perm = torch.randperm(src.numel(), generator=g0)          # unsorted COO: the CSC carries DGL's usual edge-id map
    src, dst = src[perm].to(dev), dst[perm].to(dev)
    return dgl.graph((src, dst), num_nodes=n, idtype=torch.int32, device=dev), src.long(), dst.long()


def _dense_gat_attention(ft, el, er, src, dst, n, slope):
    e = torch.nn.functional.leaky_relu(el[src] + er[dst], slope)                # (E, H, 1)
    mx = torch.full((n,) + e.shape[1:], float("-inf"), device=e.device, dtype=e.dtype).index_reduce_(0, dst, e, "amax")
    ex = torch.exp(e - mx[dst])
    a = ex / torch.zeros_like(mx).index_add_(0, dst, ex)[dst]
    return torch.zeros((n,) + ft.shape[1:], device=e.device, dtype=e.dtype).index_add_(0, dst, a * ft[src])


@pytest.mark.parametrize("route", ["composed", "composed_handoff", "default"])
@pytest.mark.parametrize("heads,d", [(4, 8), (8, 32), (1, 5)])
def test_gat_attention_block_matches_dense(dev, route, heads, d):
    """configs[2]'s attention block (gatconv.py:330-347) as dgl_amd.nn.gat_attention: the composed operators (plain, and
    inside the opt-in hand-off scope) and the default route (the fused kernel where it applies) against a dense torch
    evaluation, forward and gradients; nothing is left in the graph's frames."""
    import dgl_amd as dgl

    g, src, dst = _random_graph(dev)
    n = g.num_nodes()
    torch.manual_seed(heads * 100 + d)
    ft = torch.randn(n, heads, d, device=dev, requires_grad=True)
    el = torch.randn(n, heads, 1, device=dev, requires_grad=True)
    er = torch.randn(n, heads, 1, device=dev, requires_grad=True)
    up = torch.randn(n, heads, d, device=dev)
    kw = {"composed": dict(fused=False), "composed_handoff": dict(fused=False, handoff=True), "default": {}}[route]
    out = dgl.nn.gat_attention(g, ft, el, er, 0.2, **kw)
    assert type(out) is torch.Tensor and out.shape == (n, heads, d)
    got = [out.detach()] + list(torch.autograd.grad((out * up).sum(), [ft, el, er]))
    want_out = _dense_gat_attention(ft, el, er, src, dst, n, 0.2)
    want = [want_out.detach()] + list(torch.autograd.grad((want_out * up).sum(), [ft, el, er]))
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5)
    assert "a" not in g.edata and "ft" not in g.ndata and "el" not in g.ndata


def test_sage_pool_aggregator_matches_dense(dev):
    """SAGEConv 'pool' (sageconv.py:253-259): max over neighbours of relu(W_pool h_u); rows without neighbours give 0."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    n_src, n_dst, e = 400, 120, 3000
    g0 = torch.Generator().manual_seed(4)
    src, dst = torch.randint(0, n_src, (e,), generator=g0).to(dev), torch.randint(8, n_dst, (e,), generator=g0).to(dev)
    blk = dgl.create_block((src, dst), num_src_nodes=n_src, num_dst_nodes=n_dst, device=dev)
    torch.manual_seed(5)
    x = torch.randn(n_src, 16, device=dev, requires_grad=True)
    w = torch.randn(16, 16, device=dev, requires_grad=True)
    with blk.local_scope():
        blk.srcdata["h"] = torch.relu(x @ w)
        blk.update_all(fn.copy_u("h", "m"), fn.max("m", "neigh"))
        out = blk.dstdata["neigh"]
    p = torch.relu(x @ w)
    mx = torch.full((n_dst, 16), float("-inf"), device=dev).index_reduce_(0, dst, p[src], "amax", include_self=True)
    want = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    torch.testing.assert_close(out, want, rtol=1e-6, atol=0)
    assert bool((out[:8] == 0).all())
    up = torch.randn_like(out)
    for a, b in zip(torch.autograd.grad((out * up).sum(), [x, w]), torch.autograd.grad((want * up).sum(), [x, w])):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("presorted", [False, True])
def test_rgcn_message_passing_matches_the_formula(dev, presorted):
    """configs[4]'s layer (relgraphconv.py:140-215): h_v' = sum_r sum_{u in N_r(v)} norm_uv W_r h_u, written with the
    operator API — per-edge typed transform through gather_mm (edge types in any order) or segment_mm (edges sorted by
    type), then copy_e + sum — against a bmm + index_add evaluation; values and gradients."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    torch.manual_seed(3)
    n, e, r, fi, fo = 300, 4000, 5, 24, 16
    u, v = torch.randint(n, (e,), device=dev), torch.randint(n, (e,), device=dev)
    et = torch.randint(0, r, (e,), device=dev)
    if presorted:
        et, perm = torch.sort(et)
        u, v = u[perm], v[perm]
    g = dgl.graph((u, v), num_nodes=n)
    h = torch.randn(n, fi, device=dev, requires_grad=True)
    w = torch.randn(r, fi, fo, device=dev, requires_grad=True)
    norm = torch.rand(e, 1, device=dev)
    hu = h[u]
    m = dgl.segment_mm(hu, w, torch.bincount(et, minlength=r)) if presorted else dgl.gather_mm(hu, w, idx_b=et)
    with g.local_scope():
        g.edata["m"] = m * norm
        g.update_all(fn.copy_e("m", "m"), fn.sum("m", "h"))
        out = g.dstdata["h"]
    want = torch.zeros(n, fo, device=dev).index_add(0, v, torch.bmm(h[u].unsqueeze(1), w[et]).squeeze(1) * norm)
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-4)
    up = torch.randn_like(out)
    for a, b in zip(torch.autograd.grad((out * up).sum(), [h, w]), torch.autograd.grad((want * up).sum(), [h, w])):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4)


def test_gat_attention_on_a_block_fused_equals_composed(dev):
    """A message-flow block (more source than destination nodes): the fused operator and the composed operators give the
    same block output and the same gradients (ft / el live on the sources, er on the destinations)."""
    import dgl_amd as dgl

    n_src, n_dst, e, h, d = 5000, 1200, 40_000, 4, 16
    g0 = torch.Generator().manual_seed(8)
    src, dst = torch.randint(0, n_src, (e,), generator=g0).to(dev), torch.randint(5, n_dst, (e,), generator=g0).to(dev)
    blk = dgl.create_block((src, dst), num_src_nodes=n_src, num_dst_nodes=n_dst, device=dev)
    torch.manual_seed(2)
    ps = [torch.randn(n_src, h, d, device=dev, requires_grad=True), torch.randn(n_src, h, 1, device=dev, requires_grad=True),
          torch.randn(n_dst, h, 1, device=dev, requires_grad=True)]
    up = torch.randn(n_dst, h, d, device=dev)
    assert dgl.ops.gat_attention_applies(blk, *ps)
    outs = []
    for kw in (dict(fused=False), dict(fused=True)):
        o = dgl.nn.gat_attention(blk, ps[0], ps[1], ps[2], 0.2, **kw)
        outs.append([o.detach()] + list(torch.autograd.grad((o * up).sum(), ps)))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5)
    assert bool((outs[1][0][:5] == 0).all())            # destinations without in-edges
