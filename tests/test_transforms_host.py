"""Graph helpers around the hot path (dgl_amd/transforms.py) and ``to_block`` on CPU-resident graphs: torch index
arithmetic, checked on CPU against direct constructions.  (The reference's own layer / readout tests, which build their
cases with these, run unmodified on the GPU through tools/ref_suite.)"""
import pytest
import torch

import dgl_amd as dgl


def _g():
    g = dgl.graph((torch.tensor([0, 1, 2, 2]), torch.tensor([1, 2, 3, 0])), num_nodes=5)
    g.ndata["x"] = torch.arange(5.0)
    g.edata["w"] = torch.tensor([1.0, 2.0, 3.0, 4.0])
    return g


def test_add_and_remove_self_loops_and_edges():
    g = _g()
    h = dgl.add_self_loop(g)
    u, v = h.edges()
    assert h.num_edges() == 9 and u[4:].tolist() == [0, 1, 2, 3, 4] and v[4:].tolist() == [0, 1, 2, 3, 4]
    assert h.edata["w"].tolist() == [1, 2, 3, 4, 1, 1, 1, 1, 1] and torch.equal(h.ndata["x"], g.ndata["x"])
    assert dgl.add_self_loop(g, fill_data="sum").edata["w"][4:].tolist() == [4.0, 1.0, 2.0, 3.0, 0.0]
    g2 = _g()
    g2.edata["t"] = torch.tensor([5.0, 6.0, 7.0, 8.0])
    h2 = dgl.add_self_loop(g2, edge_feat_names=["w"], fill_data=2.0)     # "t" is not named: kept, zeros on the new edges
    assert h2.edata["w"][4:].tolist() == [2.0] * 5 and h2.edata["t"].tolist() == [5, 6, 7, 8, 0, 0, 0, 0, 0]
    back = dgl.remove_self_loop(h)
    assert torch.equal(back.edges()[0], g.edges()[0]) and torch.equal(back.edata["w"], g.edata["w"])
    r = dgl.remove_edges(g, [1, 3], store_ids=True)
    assert r.edges()[0].tolist() == [0, 2] and r.edata["w"].tolist() == [1.0, 3.0] and r.edata["_ID"].tolist() == [0, 2]
    assert g.num_edges() == 4                                     # the input is not modified
    bi = dgl.heterograph({("a", "r", "b"): (torch.tensor([0]), torch.tensor([1]))})
    with pytest.raises(dgl.DGLError):
        dgl.add_self_loop(bi)


def test_reorder_graph_by_given_permutation_and_by_ids():
    g = _g()
    perm = torch.tensor([3, 1, 0, 2])
    h = dgl.reorder_graph(g, edge_permute_algo="custom", permute_config={"edges_perm": perm})
    assert torch.equal(h.edges()[0], g.edges()[0][perm]) and torch.equal(h.edata["w"], g.edata["w"][perm])
    assert torch.equal(h.edata["_ID"], perm)
    d = dgl.reorder_graph(g, edge_permute_algo="dst", store_ids=False)
    assert d.edges()[1].tolist() == [0, 1, 2, 3] and "_ID" not in d.edata
    with pytest.raises(dgl.DGLError):
        dgl.reorder_graph(g, node_permute_algo="rcmk")


def test_batch_and_unbatch_round_trip():
    g1, g2 = _g(), dgl.graph((torch.tensor([1, 1]), torch.tensor([2, 0])), num_nodes=3)
    g2.ndata["x"] = torch.tensor([7.0, 8.0, 9.0])
    g2.edata["w"] = torch.tensor([5.0, 6.0])
    b = dgl.batch([g1, g2])
    assert b.batch_size == 2 and b.batch_num_nodes().tolist() == [5, 3] and b.batch_num_edges().tolist() == [4, 2]
    assert b.edges()[0].tolist() == [0, 1, 2, 2, 6, 6] and b.edges()[1].tolist() == [1, 2, 3, 0, 7, 5]
    assert b.ndata["x"].tolist() == [0, 1, 2, 3, 4, 7, 8, 9]
    parts = dgl.unbatch(b)
    for a, c in zip(parts, (g1, g2)):
        assert torch.equal(a.edges()[0], c.edges()[0]) and torch.equal(a.edges()[1], c.edges()[1])
        assert torch.equal(a.ndata["x"], c.ndata["x"]) and torch.equal(a.edata["w"], c.edata["w"])
    assert g1.batch_size == 1 and g1.batch_num_nodes().tolist() == [5]
    hb = dgl.batch([dgl.heterograph({("a", "r", "b"): (torch.tensor([0, 1]), torch.tensor([1, 0]))})] * 2)
    assert hb.num_nodes("a") == 4 and hb.edges(etype="r")[0].tolist() == [0, 1, 2, 3]
    with pytest.raises(dgl.DGLError):
        dgl.batch([])


def test_scipy_and_adjacency():
    sp = pytest.importorskip("scipy.sparse")
    m = sp.random(8, 8, density=0.3, random_state=1, format="coo")
    g = dgl.from_scipy(m, eweight_name="w")
    assert g.num_nodes() == 8 and g.num_edges() == m.nnz and torch.allclose(g.edata["w"].double(), torch.as_tensor(m.data))
    a = g.adj_external().to_dense()
    assert torch.equal(a != 0, torch.as_tensor(m.toarray()) != 0)
    assert torch.equal(g.adj_external(transpose=True).to_dense(), a.T)
    assert (g.adj_external(scipy_fmt="csr") != m.tocsr().astype(bool)).nnz == 0
    b = dgl.bipartite_from_scipy(sp.random(3, 5, density=0.5, random_state=2), "u", "e", "v")
    assert b.num_nodes("u") == 3 and b.num_nodes("v") == 5
    with pytest.raises(dgl.DGLError):
        dgl.from_scipy(sp.random(3, 5, density=0.5))


def test_to_block_on_a_cpu_graph_and_without_destination_nodes():
    g = dgl.graph((torch.tensor([2, 3, 4, 9]), torch.tensor([5, 6, 7, 5])), num_nodes=100)
    blk = dgl.to_block(g)                                          # destinations: the nodes with an inbound edge, ascending
    assert blk.is_block and blk.dstdata["_ID"].tolist() == [5, 6, 7]
    assert blk.srcdata["_ID"].tolist() == [5, 6, 7, 2, 3, 4, 9] and blk.num_src_nodes() == 7 and blk.num_dst_nodes() == 3
    u, v = blk.edges()
    back = sorted(zip(blk.srcdata["_ID"][u.long()].tolist(), blk.dstdata["_ID"][v.long()].tolist()))
    assert back == [(2, 5), (3, 6), (4, 7), (9, 5)]
    assert blk.to("cpu").is_block and blk.astype(torch.int32).is_block
    hg = dgl.heterograph({("user", "plays", "game"): (torch.tensor([0, 1, 2]), torch.tensor([1, 1, 0])),
                          ("store", "sells", "game"): (torch.tensor([0, 1, 1]), torch.tensor([0, 1, 2]))})
    hb = dgl.to_block(hg)
    assert hb.dstnodes["game"].data["_ID"].tolist() == [0, 1, 2] and hb.num_dst_nodes("user") == 0
    assert hb.srcnodes["user"].data["_ID"].tolist() == [0, 1, 2] and hb.srcnodes["game"].data["_ID"].tolist() == [0, 1, 2]


def test_reverse_edges_simple_and_bidirected():
    g = dgl.graph((torch.tensor([0, 1, 1, 2, 2]), torch.tensor([1, 2, 2, 2, 0])), num_nodes=4)
    g.edata["w"] = torch.arange(5.0)
    r = dgl.add_reverse_edges(g, copy_edata=True)
    assert r.edges()[0].tolist() == [0, 1, 1, 2, 2, 1, 2, 2, 0] and r.edges()[1].tolist() == [1, 2, 2, 2, 0, 0, 1, 1, 2]
    assert r.edata["w"].tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 4]              # the self-loop 2 -> 2 is not doubled
    s, back = dgl.to_simple(g, writeback_mapping=True)
    assert list(zip(*[t.tolist() for t in s.edges()])) == [(0, 1), (1, 2), (2, 0), (2, 2)]
    assert s.edata["count"].tolist() == [1, 2, 1, 1] and back.tolist() == [0, 1, 1, 3, 2]
    b = dgl.to_bidirected(g)
    assert list(zip(*[t.tolist() for t in b.edges()])) == [(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (2, 2)]


def test_subgraphs():
    g = _g()                                                        # edges 0->1, 1->2, 2->3, 2->0 on 5 nodes
    sg = dgl.node_subgraph(g, [2, 0, 3])
    assert sg.ndata["_ID"].tolist() == [2, 0, 3] and sg.ndata["x"].tolist() == [2.0, 0.0, 3.0]
    assert list(zip(*[t.tolist() for t in sg.edges()])) == [(0, 2), (0, 1)] and sg.edata["_ID"].tolist() == [2, 3]
    assert sg.edata["w"].tolist() == [3.0, 4.0]
    mask = torch.tensor([True, False, True, True, False])
    assert torch.equal(dgl.node_subgraph(g, mask).ndata["_ID"], torch.tensor([0, 2, 3]))
    eg = dgl.edge_subgraph(g, [3, 0])
    assert eg.ndata["_ID"].tolist() == [0, 1, 2] and list(zip(*[t.tolist() for t in eg.edges()])) == [(2, 0), (0, 1)]
    assert eg.edata["_ID"].tolist() == [3, 0]
    keep = dgl.edge_subgraph(g, [3, 0], relabel_nodes=False)
    assert keep.num_nodes() == 5 and list(zip(*[t.tolist() for t in keep.edges()])) == [(2, 0), (0, 1)]
    ins = dgl.in_subgraph(g, [0, 3])
    assert ins.num_nodes() == 5 and list(zip(*[t.tolist() for t in ins.edges()])) == [(2, 3), (2, 0)] and ins.edata["_ID"].tolist() == [2, 3]
    hg = dgl.heterograph({("a", "r", "b"): (torch.tensor([0, 1, 2]), torch.tensor([1, 1, 0])),
                          ("b", "s", "a"): (torch.tensor([0, 1]), torch.tensor([2, 0]))})
    hs = dgl.node_subgraph(hg, {"a": [0, 2], "b": [0, 1]})
    assert list(zip(*[t.tolist() for t in hs.edges(etype="r")])) == [(0, 1), (1, 0)]
    assert list(zip(*[t.tolist() for t in hs.edges(etype="s")])) == [(0, 1), (1, 0)]
    he = dgl.edge_subgraph(hg, {"r": [2], "s": torch.tensor([True, False])})
    assert he.num_nodes("a") == 1 and he.num_nodes("b") == 1 and he.nodes["a"].data["_ID"].tolist() == [2]


def test_to_heterogeneous_records_the_original_ids():
    """python/dgl/convert.py:878-887: hg.ndata[dgl.NID] / hg.edata[dgl.EID] map per-type nodes / edges back to the
    homogeneous graph; to_homogeneous(store_type=False, return_count=True) as the reference documents it."""
    import pytest

    import dgl_amd as dgl
    from dgl_amd.heterograph import EID, ETYPE, NID, NTYPE

    hg = dgl.heterograph({("a", "x", "b"): ([0, 1, 2], [1, 0, 1]), ("b", "y", "a"): ([0, 1], [2, 2])},
                         {"a": 3, "b": 2})
    g, ncount, ecount = dgl.to_homogeneous(hg, return_count=True)
    assert ncount == [3, 2] and ecount == [3, 2]
    g.ndata["h"] = torch.arange(5.0)
    g.edata["w"] = torch.arange(5.0) * 10
    back = dgl.to_heterogeneous(g, hg.ntypes, hg.etypes)
    for nt in hg.ntypes:
        nid = back.nodes[nt].data[NID]
        assert torch.equal(g.ndata["h"][nid.long()], back.nodes[nt].data["h"])
        assert torch.equal(g.ndata[NTYPE][nid.long()], torch.full_like(nid, hg.get_ntype_id(nt)))
    for cet in back.canonical_etypes:
        eid = back.edges[cet].data[EID]
        assert torch.equal(g.edata["w"][eid.long()], back.edges[cet].data["w"])
        u, v = back.edges(etype=cet)
        gu, gv = g.edges()
        assert torch.equal(back.nodes[cet[0]].data[NID][u.long()], gu[eid.long()])
        assert torch.equal(back.nodes[cet[2]].data[NID][v.long()], gv[eid.long()])
    plain = dgl.to_homogeneous(hg, store_type=False)
    assert NTYPE not in plain.ndata and ETYPE not in plain.edata and NID in plain.ndata
    with pytest.raises(dgl.DGLError):
        dgl.to_heterogeneous(g, hg.ntypes, hg.etypes, metagraph=object())
