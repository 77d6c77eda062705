"""Uniform neighbour sampling + block construction.

Random picks cannot be bit-compared with the reference's Philox streams; what the reference's
own tests check are the properties below (tests/python/common/sampling/test_sampling.py:
_test_sample_neighbors — picked edges are edges of the graph, per-node counts, no repetition
without replacement; tests/python/common/transforms/test_to_block.py — dst nodes first, induced
ids map back) — checked here exactly (integer work), plus reproducibility from the seed, a
chi-square test of uniformity, and end-to-end equality of a 2-layer mean-SAGE forward on the
sampled blocks with a dense evaluation over the same sampled edges."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(dev, idtype, n=3000, e=60000, seed=0):
    import dgl_amd as dgl

    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e)
    dst = np.minimum((rng.random(e) ** 3 * n).astype(np.int64), n - 1)   # skewed in-degrees, some 0
    g = dgl.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n, idtype=idtype, device=dev)
    return g, src, dst


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("fanout,replace", [(5, False), (15, False), (1, False), (128, False), (10, True), (-1, False)])
def test_sample_neighbors_properties(dev, idtype, fanout, replace):
    from dgl_amd import _capi
    from dgl_amd.sampling import _csc_of

    g, src, dst = _graph(dev, idtype)
    rel, csr, (ip, ix, ei) = _csc_of(g)
    seeds = torch.from_numpy(np.random.default_rng(1).permutation(3000)[:700]).to(dev).to(idtype)
    indptr, s, e = _capi.sample_neighbors(csr, seeds, fanout, replace, 42)
    indptr_h, s_h, e_h = indptr.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy()
    indeg = np.bincount(dst, minlength=3000)
    for i, v in enumerate(seeds.cpu().numpy()):
        lo, hi = indptr_h[i], indptr_h[i + 1]
        want = indeg[v] if fanout < 0 else (min(indeg[v], fanout) if not replace else (fanout if indeg[v] else 0))
        assert hi - lo == want
        # every pick is an in-edge of v, and (src, eid) are consistent with the COO
        assert (dst[e_h[lo:hi]] == v).all() and (src[e_h[lo:hi]] == s_h[lo:hi]).all()
        if not replace:
            assert len(set(e_h[lo:hi].tolist())) == hi - lo
    assert indptr_h[-1] <= len(s_h)
    # reproducible from the seed, different under another seed
    indptr2, s2, e2 = _capi.sample_neighbors(csr, seeds, fanout, replace, 42)
    n_e = int(indptr_h[-1])
    assert torch.equal(indptr, indptr2) and torch.equal(e[:n_e], e2[:n_e])
    if 0 < fanout < 20:
        _, _, e3 = _capi.sample_neighbors(csr, seeds, fanout, replace, 43)
        assert not torch.equal(e[:n_e], e3[:n_e])
    if fanout < 0:   # the whole neighbourhood in CSC order: bit-exact
        ip_h, ei_h = ip.cpu().numpy(), ei.cpu().numpy()
        for i, v in enumerate(seeds.cpu().numpy()[:50]):
            np.testing.assert_array_equal(e_h[indptr_h[i]:indptr_h[i + 1]], ei_h[ip_h[v]:ip_h[v + 1]])


def test_sampling_is_uniform(dev):
    """One node with 40 in-neighbours, fanout 4, 6 000 independent draws: every neighbour is
    picked ~600 times (chi-square with 39 dof, p ~ 1e-6 bound 90)."""
    import dgl_amd as dgl
    from dgl_amd import _capi
    from dgl_amd.sampling import _csc_of

    n = 6000
    src = torch.arange(1, 41).repeat(n)                   # the same 40 sources into every node
    dst = torch.repeat_interleave(torch.arange(n), 40)
    g = dgl.graph((src, dst), num_nodes=n, device=dev)
    _, csr, _ = _csc_of(g)
    seeds = torch.arange(n, device=dev)
    indptr, s, _ = _capi.sample_neighbors(csr, seeds, 4, False, 7)
    counts = torch.bincount(s[: int(indptr[-1])], minlength=41)[1:].double().cpu().numpy()
    assert counts.sum() == n * 4
    chi2 = ((counts - 600.0) ** 2 / 600.0).sum()
    assert chi2 < 90, chi2
    # with replacement
    indptr, s, _ = _capi.sample_neighbors(csr, seeds, 4, True, 8)
    counts = torch.bincount(s[: int(indptr[-1])], minlength=41)[1:].double().cpu().numpy()
    assert ((counts - 600.0) ** 2 / 600.0).sum() < 90


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_to_block_renumbering(dev, idtype):
    from dgl_amd import _capi

    rng = np.random.default_rng(3)
    n = 5000
    seeds = torch.from_numpy(rng.permutation(n)[:300]).to(dev).to(idtype)
    src = torch.from_numpy(rng.integers(0, n, 4000)).to(dev).to(idtype)
    node_map = torch.full((n,), -1, dtype=torch.int32, device=dev)
    local, src_nodes, k = _capi.to_block(seeds, src, node_map)
    assert torch.equal(src_nodes[:300], seeds)                        # destination nodes come first
    assert torch.equal(src_nodes[local.long()], src)                  # ids map back
    assert k == len(set(seeds.tolist()) | set(src.tolist()))          # every node once
    rest = src_nodes[300:]
    assert bool((rest[1:] > rest[:-1]).all())                         # others by ascending id
    assert int((node_map != -1).sum()) == 0                           # scratch is clean again
    # no sampled edges at all
    local, src_nodes, k = _capi.to_block(seeds, src[:0], node_map)
    assert k == 300 and local.numel() == 0


def test_two_layer_sage_on_sampled_blocks_matches_dense(dev):
    import dgl_amd as dgl
    import dgl_amd.function as fn

    g, src, dst = _graph(dev, torch.int64, n=2000, e=50000, seed=5)
    torch.manual_seed(0)
    feat = torch.randn(2000, 32, device=dev)
    w = [torch.randn(32, 16, device=dev), torch.randn(32, 16, device=dev),
         torch.randn(16, 8, device=dev), torch.randn(16, 8, device=dev)]
    sampler = dgl.NeighborSampler([15, 10], seed=3)
    seeds = torch.arange(0, 2000, 7, device=dev)
    input_nodes, output_nodes, blocks = sampler.sample_blocks(g, seeds)
    assert torch.equal(output_nodes, seeds) and len(blocks) == 2
    assert torch.equal(blocks[0].srcdata[dgl.NID], input_nodes)
    assert torch.equal(blocks[1].dstdata[dgl.NID], seeds)
    assert torch.equal(blocks[0].dstdata[dgl.NID], blocks[1].srcdata[dgl.NID])

    def layer(blk, h, ws, wn):
        with blk.local_scope():
            blk.srcdata["h"] = h
            blk.update_all(fn.copy_u("h", "m"), fn.mean("m", "n"))
            return h[: blk.num_dst_nodes()] @ ws + blk.dstdata["n"] @ wn

    h = feat[input_nodes]
    h = torch.relu(layer(blocks[0], h, w[0], w[1]))
    out = layer(blocks[1], h, w[2], w[3])

    # dense evaluation over exactly the sampled edges (original ids from EID)
    def dense_layer(blk, h_full_by_node, ws, wn):
        eid = blk.edata[dgl.EID].long()
        s_t = torch.from_numpy(src).to(dev)[eid]
        d_t = torch.from_numpy(dst).to(dev)[eid]
        dn = blk.dstdata[dgl.NID].long()
        agg = torch.zeros(2000, h_full_by_node.shape[1], device=dev).index_add_(0, d_t, h_full_by_node[s_t])
        deg = torch.bincount(d_t, minlength=2000).clamp(min=1).unsqueeze(-1)
        res = torch.zeros(2000, ws.shape[1], device=dev)
        res[dn] = h_full_by_node[dn] @ ws + (agg / deg)[dn] @ wn
        return res

    h1 = torch.relu(dense_layer(blocks[0], feat, w[0], w[1]))
    want = dense_layer(blocks[1], h1, w[2], w[3])[seeds]
    assert torch.allclose(out, want, rtol=1e-4, atol=1e-4)
    # a second call draws a different sample, the same sampler state reproduces the first
    _, _, b2 = sampler.sample_blocks(g, seeds)
    assert not torch.equal(b2[1].edata[dgl.EID], blocks[1].edata[dgl.EID])
    _, _, b3 = dgl.NeighborSampler([15, 10], seed=3).sample_blocks(g, seeds)
    assert torch.equal(b3[1].edata[dgl.EID], blocks[1].edata[dgl.EID])


def test_sample_neighbors_out_direction_and_to_block(dev):
    """edge_dir='out' samples outbound edges through the out-edge CSR; dgl.to_block on the sampled
    frontier gives the same block structure NeighborSampler builds (dst nodes first, ids map back)."""
    import dgl_amd as dgl

    g, src, dst = _graph(dev, torch.int64, n=1500, e=30000, seed=9)
    seeds = torch.arange(0, 1500, 5, device=dev)
    fo = dgl.sampling.sample_neighbors(g, seeds, 4, edge_dir="out", seed=5)
    s, d = fo.edges()
    eid = fo.edata[dgl.EID].long()
    assert torch.equal(torch.from_numpy(src).to(dev)[eid], s) and torch.equal(torch.from_numpy(dst).to(dev)[eid], d)
    outdeg = np.bincount(src, minlength=1500)
    cnt = torch.bincount(s, minlength=1500).cpu().numpy()
    assert (cnt[seeds.cpu().numpy()] == np.minimum(outdeg[seeds.cpu().numpy()], 4)).all() and cnt.sum() == len(s)
    # inbound frontier -> block
    fi = dgl.sampling.sample_neighbors(g, seeds, 6, seed=2)
    blk = dgl.to_block(fi, seeds)
    assert blk.num_dst_nodes() == seeds.numel() and torch.equal(blk.srcdata[dgl.NID][: seeds.numel()], seeds)
    bs, bd = blk.edges()
    fs, fd = fi.edges()
    # (to_block's edge ids are ids IN THE FRONTIER, as in the reference; the sampler maps them on to the original graph)
    got = torch.stack([blk.srcdata[dgl.NID][bs.long()], blk.dstdata[dgl.NID][bd.long()], fi.edata[dgl.EID][blk.edata[dgl.EID].long()]])
    assert torch.equal(fs[blk.edata[dgl.EID].long()], blk.srcdata[dgl.NID][bs.long()])
    want = torch.stack([fs, fd, fi.edata[dgl.EID]])
    key = lambda t: t[:, torch.argsort(t[2])]
    assert torch.equal(key(got), key(want))
    with pytest.raises(ValueError, match="do not end in dst_nodes"):
        dgl.to_block(fi, seeds[:10])


# ---- weighted sampling (dgla_sample_neighbors_weighted ≙ src/array/cuda/rowwise_sampling_prob.cu) ----
@pytest.mark.gpu
@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
@pytest.mark.parametrize("pdt", [torch.float32, torch.float64])
@pytest.mark.parametrize("replace", [False, True])
def test_weighted_sampling_structure(dev, idt, pdt, replace):
    from dgl_amd import _capi

    rng = np.random.default_rng(4)
    n, e, fanout = 300, 9000, 7
    src = torch.from_numpy(rng.integers(0, n, e)).to(dev)
    dst = torch.from_numpy(np.concatenate([rng.integers(0, n - 3, e - 700), np.full(700, n - 1)])).to(dev)
    indptr, indices, eids = _capi.coo_to_csr(dst.to(idt), src.to(idt), None, n)
    csr = _capi.make_csr(indptr, indices, eids, n)
    prob = torch.from_numpy(rng.random(e)).to(dev).to(pdt)
    prob[torch.from_numpy(rng.random(e) < 0.3).to(dev)] = 0            # 30 % of the edges can never be picked
    prob[(dst == 5)] = 0                                             # a row without any positive edge
    seeds = torch.arange(n, device=dev, dtype=idt)
    ip, s_out, e_out = _capi.sample_neighbors_weighted(csr, prob, seeds, fanout, replace, rng_seed=11)
    ip2, s2, e2 = _capi.sample_neighbors_weighted(csr, prob, seeds, fanout, replace, rng_seed=11)
    cnt = int(ip[-1])
    assert torch.equal(ip, ip2) and torch.equal(e_out[:cnt], e2[:cnt])   # reproducible from the seed
    ip3, _, e3 = _capi.sample_neighbors_weighted(csr, prob, seeds, fanout, replace, rng_seed=12)
    assert not torch.equal(e_out[:cnt], e3[:int(ip3[-1])]) or cnt == 0
    ip_h, e_h, s_h = ip.cpu().numpy(), e_out.cpu().numpy(), s_out.cpu().numpy()
    p_h, src_h, dst_h = prob.cpu().numpy(), src.cpu().numpy(), dst.cpu().numpy()
    for r in range(n):
        picked = e_h[ip_h[r]: ip_h[r + 1]]
        pos = int(((dst_h == r) & (p_h > 0)).sum())
        assert len(picked) == ((fanout if pos else 0) if replace else min(fanout, pos)), r
        assert np.all(dst_h[picked] == r) and np.all(p_h[picked] > 0)
        assert np.array_equal(src_h[picked], s_h[ip_h[r]: ip_h[r + 1]])
        if not replace:
            assert len(set(picked.tolist())) == len(picked)


@pytest.mark.gpu
@pytest.mark.parametrize("replace", [False, True])
def test_weighted_sampling_follows_the_weights(dev, replace):
    """fanout = 1: P(edge) = w / sum(w), with or without replacement.  4000 independent rows with
    the same five weights; the empirical frequencies must match to ~3 sigma."""
    from dgl_amd import _capi

    w = np.array([0.5, 0.0, 2.0, 1.0, 4.5])
    rows = 4000
    indptr = torch.arange(0, 5 * rows + 1, 5, dtype=torch.int64, device=dev)
    indices = torch.arange(5, device=dev).repeat(rows)
    csr = _capi.make_csr(indptr, indices, None, 5)
    prob = torch.from_numpy(np.tile(w, rows)).to(dev)
    seeds = torch.arange(rows, device=dev)
    ip, s_out, _ = _capi.sample_neighbors_weighted(csr, prob, seeds, 1, replace, rng_seed=99)
    assert int(ip[-1]) == rows
    freq = np.bincount(s_out[:rows].cpu().numpy(), minlength=5) / rows
    want = w / w.sum()
    sigma = np.sqrt(want * (1 - want) / rows)
    assert np.all(np.abs(freq - want) <= 4 * sigma + 1e-12), (freq, want)
    # without replacement, fanout 2: the pair {4, 2} is the most likely, and weight-0 never shows
    ip, s_out, _ = _capi.sample_neighbors_weighted(csr, prob, seeds, 2, False, rng_seed=5)
    pairs = s_out[: 2 * rows].cpu().numpy().reshape(rows, 2)
    assert not np.any(pairs == 1)
    first = np.bincount(pairs[:, 0], minlength=5) / rows     # the first pick is the smallest key: ~ w / sum(w)
    assert np.all(np.abs(first - want) <= 4 * sigma + 1e-12)


@pytest.mark.gpu
def test_sample_neighbors_api_with_prob(dev):
    import dgl_amd as dgl

    g = dgl.graph((torch.tensor([0, 1, 2, 3, 4, 5]), torch.tensor([6, 6, 6, 6, 6, 6])), num_nodes=7, device=dev)
    g.edata["w"] = torch.tensor([0., 0., 1., 0., 2., 0.], device=dev)
    sub = dgl.sampling.sample_neighbors(g, torch.tensor([6], device=dev), 4, prob="w", seed=3)
    assert sorted(sub.edata[dgl.EID].tolist()) == [2, 4]       # only the positive-weight edges exist
    sampler = dgl.NeighborSampler([3], prob="w")
    _, _, blocks = sampler.sample_blocks(g, torch.tensor([6], device=dev))
    assert sorted(blocks[0].edata[dgl.EID].tolist()) == [2, 4]
    # a probability tensor that is not one value per edge of g is rejected before any kernel reads it
    for bad in (torch.ones(3, device=dev), torch.ones(6, 2, device=dev), torch.ones(7, device=dev)):
        with pytest.raises(dgl.DGLError, match="one value per edge"):
            dgl.sampling.sample_neighbors(g, torch.tensor([6], device=dev), 2, prob=bad, seed=3)


@pytest.mark.gpu
def test_block_backward_takes_the_coo_sum_and_agrees_with_the_csc_route(dev, monkeypatch):
    """Sampled blocks are transient: the backward pass of copy_u + mean on a block runs the sum
    over the reversed block through the COO kernel (no CSC is sorted for one use); under
    USE_DETERMINISTIC_ALG it builds the CSC like the reference.  Both gradients agree, and both
    equal the gradient of a dense evaluation."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    g, src, dst = _graph(dev, torch.int64, n=3000, e=60000, seed=9)
    sampler = dgl.NeighborSampler([10], seed=4)
    seeds = torch.arange(0, 3000, 5, device=dev)
    inp, _, (blk,) = sampler.sample_blocks(g, seeds)
    torch.manual_seed(1)
    h0 = torch.randn(inp.shape[0], 24, device=dev)
    up = torch.randn(seeds.shape[0], 24, device=dev)

    def grad_of(block):
        h = h0.clone().requires_grad_(True)
        with block.local_scope():
            block.srcdata["h"] = h
            block.update_all(fn.copy_u("h", "m"), fn.mean("m", "n"))
            (block.dstdata["n"] * up).sum().backward()
        return h.grad

    rel = blk._graph.relations[0]
    assert rel.transient and rel.reverse().transient
    g_coo = grad_of(blk)
    assert "coo" in rel.reverse()._set and "csc" not in rel.reverse()._set     # no CSC was built
    # the reference's route: a fresh block (same sample), CSC of the reversed block
    monkeypatch.setenv("USE_DETERMINISTIC_ALG", "1")
    _, _, (blk2,) = dgl.NeighborSampler([10], seed=4).sample_blocks(g, seeds)
    g_csc = grad_of(blk2)
    assert "csc" in blk2._graph.relations[0].reverse()._set
    assert torch.allclose(g_coo, g_csc, rtol=1e-5, atol=1e-6)
    # dense check: d/dh of sum_v up[v] . mean_{u -> v} h[u]
    ip, ix, _ = rel.csc()
    deg = (ip[1:] - ip[:-1]).clamp(min=1).to(h0.dtype)
    rows = torch.repeat_interleave(torch.arange(seeds.shape[0], device=dev), (ip[1:] - ip[:-1]).long())
    want = torch.zeros_like(h0).index_add_(0, ix.long(), (up / deg.unsqueeze(-1))[rows])
    assert torch.allclose(g_coo, want, rtol=1e-5, atol=1e-6)
