"""Static-shape (padded) neighbour sampling and block building — dgla_sample_neighbors_padded /
dgla_to_block_padded / NeighborSampler.sample_blocks_padded — and a whole mini-batch step captured
in one hipGraph.  Real rows must be exactly what the unpadded calls give for the same draw counter
(reference semantics: python/dgl/dataloading/neighbor_sampler.py sample_blocks,
src/array/cuda/rowwise_sampling.cu, src/graph/transform/cuda/cuda_to_block.cu)."""
import numpy as np
import pytest
import torch

from tests.graphgen import synth_csr

pytestmark = pytest.mark.gpu
GOLD = 0x9E3779B97F4A7C15


@pytest.fixture()
def dev():
    return torch.device("cuda:0")


def _graph(dev, idtype, n=5000, e=60000):
    from dgl_amd.graph_index import GraphIndex, Relation
    from dgl_amd.heterograph import DGLGraph

    gs = synth_csr(n, n, e, "L", seed=3, device=dev, idtype=idtype)
    rel = Relation(n, n, csc=(gs["indptr"], gs["indices"], None), idtype=idtype, device=dev)
    return DGLGraph(GraphIndex([n], [(0, 0)], [rel]), ["_N"], [("_N", "_E", "_N")]), gs


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("replace", [False, True])
@pytest.mark.parametrize("fanout,n_slots,n_valid", [(5, 64, 64), (10, 100, 37), (15, 257, 1)])
def test_padded_sampler_equals_the_unpadded_one_on_real_rows(dev, idtype, replace, fanout, n_slots, n_valid):
    from dgl_amd import _capi

    n = 5000
    g, gs = _graph(dev, idtype)
    csr = _capi.make_csr(gs["indptr"], gs["indices"], None, n)
    gen = torch.Generator(device=dev).manual_seed(n_slots)
    seeds = torch.randperm(n, device=dev, generator=gen)[:n_slots].to(idtype)
    seeds[n_valid:] = 0                                   # padding slots hold some valid node id
    nv = torch.tensor([n_valid], dtype=torch.int64, device=dev)
    ctr = torch.tensor([7], dtype=torch.int64, device=dev)
    indptr, src, eids = _capi.sample_neighbors_padded(csr, seeds, nv, fanout, replace, 12345, ctr)
    ref_ptr, ref_src, ref_eids = _capi.sample_neighbors(csr, seeds[:n_valid].contiguous(), fanout, replace,
                                                        (12345 + 7 * GOLD) & 0xFFFFFFFFFFFFFFFF)
    torch.cuda.synchronize()
    total = int(ref_ptr[-1])
    cap = n_slots * fanout
    K = _capi.SINK_ROWS
    assert indptr.shape[0] == n_slots + 1 + K and src.shape[0] == cap
    assert torch.equal(indptr[: n_valid + 1], ref_ptr)
    assert bool((indptr[n_valid: n_slots + 1] == total).all())      # padding slots pick nothing
    sink = indptr[n_slots:].long()
    assert int(sink[-1]) == cap and bool((sink[1:] >= sink[:-1]).all())   # the sink rows share the rest
    assert int((sink[1:] - sink[:-1]).max()) <= (cap - total + K - 1) // K + 1
    assert torch.equal(src[:total], ref_src[:total]) and torch.equal(eids[:total], ref_eids[:total])
    assert bool(torch.isin(src[total:], seeds[:n_valid]).all())       # padding edges point at real seeds


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("replace", [False, True])
def test_padded_weighted_sampler_equals_the_unpadded_one_on_real_rows(dev, idtype, replace):
    from dgl_amd import _capi

    n, n_slots, n_valid, fanout = 5000, 90, 41, 6
    g, gs = _graph(dev, idtype)
    csr = _capi.make_csr(gs["indptr"], gs["indices"], None, n)
    gen = torch.Generator(device=dev).manual_seed(3)
    prob = torch.rand(gs["indices"].shape[0], device=dev, generator=gen)
    prob[::7] = 0                                           # edges that may never be picked
    seeds = torch.randperm(n, device=dev, generator=gen)[:n_slots].to(idtype)
    seeds[n_valid:] = 0
    nv = torch.tensor([n_valid], dtype=torch.int64, device=dev)
    ctr = torch.tensor([3], dtype=torch.int64, device=dev)
    indptr, src, eids = _capi.sample_neighbors_padded(csr, seeds, nv, fanout, replace, 77, ctr, prob=prob)
    ref_ptr, ref_src, ref_eids = _capi.sample_neighbors_weighted(csr, prob, seeds[:n_valid].contiguous(), fanout,
                                                                 replace, (77 + 3 * GOLD) & 0xFFFFFFFFFFFFFFFF)
    torch.cuda.synchronize()
    total = int(ref_ptr[-1])
    assert torch.equal(indptr[: n_valid + 1], ref_ptr)
    assert bool((indptr[n_valid: n_slots + 1] == total).all()) and int(indptr[-1]) == n_slots * fanout
    assert torch.equal(src[:total], ref_src[:total]) and torch.equal(eids[:total], ref_eids[:total])
    assert bool((prob[eids[:total].long()] > 0).all())
    assert bool(torch.isin(src[total:], seeds[:n_valid]).all())


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_padded_to_block_renumbers_like_the_unpadded_one(dev, idtype):
    from dgl_amd import _capi
    from dgl_amd.sampling import _node_map

    n, n_slots, n_valid, fanout = 5000, 120, 77, 10
    g, gs = _graph(dev, idtype)
    csr = _capi.make_csr(gs["indptr"], gs["indices"], None, n)
    gen = torch.Generator(device=dev).manual_seed(5)
    seeds = torch.randperm(n, device=dev, generator=gen)[:n_slots].to(idtype)
    seeds[n_valid:] = seeds[3]                            # padding = copies of a REAL seed: must not hijack its id
    nv = torch.tensor([n_valid], dtype=torch.int64, device=dev)
    indptr, src, _ = _capi.sample_neighbors_padded(csr, seeds, nv, fanout, False, 99, None)
    node_map = _node_map(g, dev)
    local, src_nodes, num_src = _capi.to_block_padded(seeds, nv, src, node_map, fill=0, num_nodes=n)
    torch.cuda.synchronize()
    k = int(num_src)
    assert bool((node_map == -1).all())                   # scratch restored
    assert torch.equal(src_nodes[:n_slots], seeds)        # destination slots first, in order
    assert bool((src_nodes[k:] == 0).all())
    # every edge resolves to a slot holding its node, and never to a padding slot
    assert torch.equal(src_nodes[local.long()], src)
    assert bool(((local < n_valid) | (local >= n_slots)).all())
    # the nodes of the block = valid seeds + the other sampled nodes, each once, new ones ascending
    total = int(indptr[n_slots])
    new = src_nodes[n_slots:k]
    want_new = torch.unique(src[:total])
    want_new = want_new[~torch.isin(want_new, seeds[:n_valid])]
    assert torch.equal(new, want_new)


def _sage_mean_reference(feat, indptr, src_global, n_rows):
    ip = indptr.long().cpu().numpy()
    sg = src_global.long().cpu().numpy()
    f = feat.double().cpu().numpy()
    out = np.zeros((n_rows, f.shape[1]))
    for r in range(n_rows):
        if ip[r + 1] > ip[r]:
            out[r] = f[sg[ip[r]:ip[r + 1]]].mean(0)
    return out


def test_padded_blocks_aggregate_like_a_dense_reference_and_replay_in_a_hipgraph(dev):
    """Two padded layers, copy_u + mean through update_all: real rows equal a numpy mean over the
    sampled neighbours; the same computation captured in a hipGraph reproduces the eager result for
    the same draw counter and draws new neighbours when the counter moves on."""
    import dgl_amd as dgl
    import dgl_amd.function as fn
    from dgl_amd.sampling import EID, NID

    n, batch, f = 5000, 48, 12
    g, gs = _graph(dev, torch.int64)
    feat = torch.rand(n, f, device=dev)
    sampler = dgl.NeighborSampler([4, 3], seed=11)
    seeds = torch.zeros(batch, dtype=torch.int64, device=dev)
    out_buf = torch.zeros(batch, f, device=dev)
    picks_buf = torch.zeros(batch * 3, dtype=torch.int64, device=dev)

    def body():
        inp, n_inp, out_nodes, blocks = sampler.sample_blocks_padded(g, seeds)
        h = feat[inp.long()]
        with blocks[0].local_scope():
            blocks[0].srcdata["h"] = h
            blocks[0].update_all(fn.copy_u("h", "m"), fn.mean("m", "n"))
            h1 = blocks[0].dstdata["n"]
        h1 = h1[: blocks[1].num_src_nodes()]
        with blocks[1].local_scope():
            blocks[1].srcdata["h"] = h1
            blocks[1].update_all(fn.copy_u("h", "m"), fn.mean("m", "n"))
            out_buf.copy_(blocks[1].dstdata["n"][:batch])
        picks_buf.copy_(blocks[1].srcdata[NID][blocks[1]._graph.relations[0].csc()[1].long()])
        return blocks

    gen = torch.Generator(device=dev).manual_seed(1)
    seeds.copy_(torch.randperm(n, device=dev, generator=gen)[:batch])
    blocks = body()
    torch.cuda.synchronize()
    # dense check of both layers on the real rows
    b0, b1 = blocks
    ip0, loc0, _ = b0._graph.relations[0].csc()
    ip1, loc1, _ = b1._graph.relations[0].csc()
    nid0, nid1 = b0.srcdata[NID], b1.srcdata[NID]
    n1 = int(b1.num_src_valid)                       # real destination rows of the outer block
    h1_ref = _sage_mean_reference(feat, ip0[: n1 + 1], nid0[loc0.long()], n1)
    # rows of padding slots inside [batch, n1)? none: slots [0, batch) are seeds, [batch, n1) new nodes
    out_ref = np.zeros((batch, f))
    ip1h, loc1h = ip1.long().cpu().numpy(), loc1.long().cpu().numpy()
    for r in range(batch):
        if ip1h[r + 1] > ip1h[r]:
            out_ref[r] = h1_ref[loc1h[ip1h[r]:ip1h[r + 1]]].mean(0)
    np.testing.assert_allclose(out_buf.cpu().numpy(), out_ref, rtol=1e-5, atol=1e-6)
    eager_first = out_buf.clone()

    # capture, then replay at the same counter value: same draws, same numbers
    step = dgl.CapturedStep(lambda s: body() and None, {"s": seeds}, warmup=1)   # `s` IS the static seed buffer
    graph = step.graph
    sampler.counter.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_buf, eager_first)
    first_picks = picks_buf.clone()
    graph.replay()                                   # counter is now 1: other neighbours
    torch.cuda.synchronize()
    assert int(sampler.counter) == 2
    assert not torch.equal(picks_buf, first_picks)
    # a new seed batch through the same graph (CapturedStep copies it into the static buffer)
    sampler.counter.fill_(40)
    step(s=torch.randperm(n, device=dev, generator=gen)[:batch])
    torch.cuda.synchronize()
    replayed = out_buf.clone()
    sampler.counter.fill_(40)
    body()
    torch.cuda.synchronize()
    assert torch.equal(out_buf, replayed)


def test_padded_sampler_refuses_a_batch_too_small_for_its_sink_rows(dev):
    """slots x fanout < 64 would give a block more destination than source slots: refused with the
    minimum batch in the message; 64 pick slots exactly are fine."""
    import dgl_amd as dgl
    from dgl_amd._lib import DGLAMDError

    g, _ = _graph(dev, torch.int64)
    sampler = dgl.NeighborSampler([3], seed=2)
    with pytest.raises(DGLAMDError, match="at least 22 seed slots"):
        sampler.sample_blocks_padded(g, torch.arange(21, device=dev))
    inp, n_inp, out_nodes, blocks = sampler.sample_blocks_padded(g, torch.arange(22, device=dev))
    assert blocks[0].num_dst_nodes() <= blocks[0].num_src_nodes()
