"""Neighbour sampling and message-flow blocks on graphs with several relations (R-GCN mini-batches): the per-type forms
of ``sample_neighbors`` (python/dgl/sampling/neighbor.py:300-395), ``to_block`` (transforms/functional.py) and
``NeighborSampler.sample_blocks`` (dataloading/neighbor_sampler.py:150-175), against a host re-evaluation."""
import pytest
import torch

pytestmark = pytest.mark.gpu

NID = EID = "_ID"


def _graph(dev, seed=0):
    import dgl_amd as dgl

    g = torch.Generator().manual_seed(seed)
    n = {"user": 300, "item": 500, "tag": 40}

    def pairs(s, d, m):
        return torch.randint(n[s], (m,), generator=g).to(dev), torch.randint(n[d], (m,), generator=g).to(dev)

    data = {("user", "buys", "item"): pairs("user", "item", 4000), ("item", "bought_by", "user"): pairs("item", "user", 3000),
            ("user", "follows", "user"): pairs("user", "user", 2500), ("tag", "labels", "item"): pairs("tag", "item", 900),
            ("item", "has", "tag"): (torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.int64, device=dev))}
    return dgl.heterograph(data, n), n


def _edge_keys(g, etype):
    u, v = g.edges(etype=etype)
    return u.long(), v.long()


@pytest.mark.parametrize("edge_dir", ["in", "out"])
@pytest.mark.parametrize("replace", [False, True])
def test_sample_neighbors_per_relation(dev, edge_dir, replace):
    import dgl_amd as dgl

    g, n = _graph(dev)
    seeds = {"user": torch.tensor([3, 7, 7, 299], device=dev), "item": torch.arange(0, 500, 7, device=dev)}
    fan = {"buys": 3, "bought_by": 2, "follows": 0, "labels": -1, "has": 4}
    torch.manual_seed(1)
    f = dgl.sampling.sample_neighbors(g, seeds, fan, edge_dir=edge_dir, replace=replace)
    assert f.ntypes == g.ntypes and f.canonical_etypes == g.canonical_etypes
    assert all(f.num_nodes(t) == n[t] for t in n)
    for c in g.canonical_etypes:
        s_t, e, d_t = c
        u, v = _edge_keys(f, c)
        eid = f.edges[c].data[EID].long()
        gu, gv = _edge_keys(g, c)
        assert torch.equal(gu[eid], u) and torch.equal(gv[eid], v)          # every pick is that edge of g
        side, own = (d_t, v) if edge_dir == "in" else (s_t, u)
        if side not in seeds or fan[e] == 0 or g.num_edges(c) == 0:
            assert f.num_edges(c) == 0
            continue
        sd = seeds[side].long()
        assert bool(torch.isin(own, sd).all())
        deg = torch.bincount(gv if edge_dir == "in" else gu, minlength=n[side])
        got = torch.bincount(own, minlength=n[side])
        mult = torch.bincount(sd, minlength=n[side])                         # a seed given twice is sampled twice
        k = fan[e]
        if k < 0:
            want = deg * mult
        elif replace:
            want = torch.where(deg > 0, torch.full_like(deg, k), deg) * mult
        else:
            want = torch.minimum(deg, torch.full_like(deg, k)) * mult
        assert torch.equal(got, want), c
        if not replace:
            per_seed_unique = torch.unique(torch.stack([own, eid]), dim=1).shape[1]
            assert per_seed_unique >= int((torch.minimum(deg, torch.full_like(deg, k if k >= 0 else 10 ** 9)) * (mult > 0))[sd.unique()].sum())
    # an int fanout means every relation; a tensor of seeds needs a single node type
    f2 = dgl.sampling.sample_neighbors(g, {"item": torch.tensor([1, 2], device=dev)}, 2, edge_dir=edge_dir)
    assert f2.num_edges(("user", "follows", "user")) == 0
    with pytest.raises(dgl.DGLError):
        dgl.sampling.sample_neighbors(g, torch.tensor([1], device=dev), 2)
    with pytest.raises(dgl.DGLError):
        dgl.sampling.sample_neighbors(g, seeds, {"buys": 1})


def test_to_block_keeps_one_node_set_per_type(dev):
    import dgl_amd as dgl

    g, n = _graph(dev, 2)
    seeds = {"user": torch.tensor([5, 9, 200], device=dev), "item": torch.tensor([499, 0, 17, 18], device=dev)}
    torch.manual_seed(3)
    f = dgl.sampling.sample_neighbors(g, seeds, {"buys": 4, "bought_by": 3, "follows": 2, "labels": 2, "has": 1})
    blk = dgl.to_block(f, seeds)
    assert blk.is_block and blk.srctypes == g.ntypes and blk.dsttypes == g.ntypes
    for t in g.ntypes:
        want_dst = seeds.get(t, torch.empty(0, dtype=torch.int64, device=dev))
        assert torch.equal(blk.dstnodes[t].data[NID], want_dst) and blk.num_dst_nodes(t) == want_dst.shape[0]
        src = blk.srcnodes[t].data[NID]
        assert torch.equal(src[: want_dst.shape[0]], want_dst) and blk.num_src_nodes(t) == src.shape[0]
        assert torch.unique(src).shape[0] == src.shape[0]
        # exactly the destination nodes plus the sources of the relations leaving this type
        used = [_edge_keys(f, c)[0] for c in g.canonical_etypes if c[0] == t]
        want_set = torch.unique(torch.cat([want_dst.long()] + used))
        assert torch.equal(torch.sort(src.long())[0], want_set)
    for c in g.canonical_etypes:
        bu, bv = _edge_keys(blk, c)
        u = blk.srcnodes[c[0]].data[NID].long()[bu]
        v = blk.dstnodes[c[2]].data[NID].long()[bv]
        ind = blk.edges[c].data[EID].long()                  # ids in the FRONTIER (the reference's convention)
        fu, fv = _edge_keys(f, c)
        assert blk.num_edges(c) == f.num_edges(c) and torch.equal(fu[ind], u) and torch.equal(fv[ind], v)
        assert torch.equal(torch.sort(ind)[0], torch.arange(f.num_edges(c), device=dev))
        eid = f.edges[c].data[EID].long()[ind]               # ... and through the frontier, of the graph
        gu, gv = _edge_keys(g, c)
        assert torch.equal(gu[eid], u) and torch.equal(gv[eid], v)
    with pytest.raises(ValueError):
        dgl.to_block(f, {"user": seeds["user"], "item": seeds["item"][:2]})     # item-bound edges outside the given items
    only_users = dgl.to_block(f, {"user": seeds["user"]})  # NO item destination at all: those relations come out empty
    assert only_users.num_edges(("user", "buys", "item")) == 0 and only_users.num_dst_nodes("item") == 0
    assert only_users.num_edges(("user", "follows", "user")) == f.num_edges(("user", "follows", "user"))


def test_two_layer_blocks_carry_an_rgcn_step(dev):
    """blocks[0].dst == blocks[1].src per type, and a per-relation ``copy_u`` / sum + cross-type sum on a block equals the
    same aggregation written with index_add over the block's edges."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    g, n = _graph(dev, 4)
    sampler = dgl.dataloading.NeighborSampler([{"buys": 3, "bought_by": 3, "follows": 2, "labels": 1, "has": 1}, 4], seed=7) \
        if hasattr(dgl, "dataloading") else dgl.sampling.NeighborSampler(
            [{"buys": 3, "bought_by": 3, "follows": 2, "labels": 1, "has": 1}, 4], seed=7)
    out_seeds = {"item": torch.tensor([4, 8, 15, 16, 23, 42], device=dev), "user": torch.tensor([1, 2], device=dev)}
    input_nodes, output_nodes, blocks = sampler.sample_blocks(g, out_seeds)
    assert len(blocks) == 2 and all(torch.equal(output_nodes[t], out_seeds[t]) for t in out_seeds)
    for t in g.ntypes:
        assert torch.equal(blocks[0].dstnodes[t].data[NID], blocks[1].srcnodes[t].data[NID])
        assert torch.equal(input_nodes[t], blocks[0].srcnodes[t].data[NID])
    feat = {t: torch.randn(n[t], 16, device=dev) for t in g.ntypes}
    blk = blocks[1]
    for t in g.ntypes:
        blk.srcnodes[t].data["h"] = feat[t][blk.srcnodes[t].data[NID].long()]
    blk.multi_update_all({c: (fn.copy_u("h", "m"), fn.sum("m", "y")) for c in g.canonical_etypes if blk.num_edges(c) > 0}, "sum")
    for t in g.ntypes:
        want = torch.zeros(blk.num_dst_nodes(t), 16, device=dev, dtype=torch.float64)
        touched = False
        for c in g.canonical_etypes:
            if c[2] != t or blk.num_edges(c) == 0:
                continue
            bu, bv = _edge_keys(blk, c)
            want.index_add_(0, bv, blk.srcnodes[c[0]].data["h"].double()[bu])
            touched = True
        if touched:
            got = blk.dstnodes[t].data["y"]
            assert torch.allclose(got.double(), want, atol=1e-5)
    # a second call draws other neighbours; a fresh sampler with the same seed repeats the first
    again = dgl.sampling.NeighborSampler(sampler.fanouts, seed=7).sample_blocks(g, out_seeds)[2]
    assert all(torch.equal(again[1].edges[c].data[EID], blocks[1].edges[c].data[EID]) for c in g.canonical_etypes)
