"""BASELINE configs[2] at FULL size and configs[4] at 1/8 size, on SURVEY §8(d)'s inputs, under the
PLAIN bar (VERDICT r3 Next #1a, #1b) — driver-run (`pytest -m gpu`), not a builder-run script.

configs[2] — GATConv pieces on the ogbn-arxiv shape: N = 169 343 nodes, E = 1 166 243 directed
edges + their reverses + one self loop per node = 2 501 829 edges, COO in RANDOM order so the
in-edge CSR carries DGL's usual scattered edge-id map (src/graph/unit_graph.cc:1418-1450), H = 8
heads, D in {8, 32}, features ``U(0,1)+1`` (tests/python/common/ops/test_ops.py:124-129).
Checked against the oracle (= the reference's CPU kernels, src/array/cpu/{spmm,sddmm}.h):
  * ``u_add_v``, ``copy_u_max`` + arg_u, ``u_mul_e_max`` + arg_u / arg_e: BIT-EXACT;
  * ``u_dot_v``, edge softmax forward, ``u_mul_e_sum``: plain ``max |out-ref| / |ref| <= 1e-5``;
  * mixed-sign inputs as an EXTRA case under the condition-aware bound (a dot of mixed signs cancels,
    |err| <= 1e-5 sum |a||b| is what fp32 arithmetic in any order can promise).
configs[4] — R-GCN: 8 relations x 1.5 M edges on 1.25 M nodes, F = 256, bf16 storage, ONE stacked
launch, equal to the fp32 oracle's running sum (src/array/cpu/spmm.h:78-109: fp32 accumulator for
16-bit storage) rounded to bf16.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle  # the checker
from tests.graphgen import synth_csr
from tests.tolerance import max_rel_err

pytestmark = pytest.mark.gpu

C3_NODES, C3_EDGES, HEADS = 169_343, 1_166_243, 8


def _h(t):
    return None if t is None else t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def c3(dev):
    """The C3 graph through the library's own COO -> CSC conversion (stable in edge order, like the
    reference's), plus host copies for the oracle."""
    from dgl_amd import _capi

    base = synth_csr(C3_NODES, C3_NODES, C3_EDGES, "U", seed=20250825, device=dev)
    deg = (base["indptr"][1:] - base["indptr"][:-1]).long()
    d0 = torch.repeat_interleave(torch.arange(C3_NODES, device=dev), deg)
    s0 = base["indices"].long()
    loops = torch.arange(C3_NODES, device=dev)
    src = torch.cat([s0, d0, loops])
    dst = torch.cat([d0, s0, loops])
    e = src.numel()
    assert e == 2 * C3_EDGES + C3_NODES == 2_501_829
    g = torch.Generator(device=dev).manual_seed(77)
    order = torch.randperm(e, device=dev, generator=g)           # edge ids = a random order of the COO
    src, dst = src[order].to(torch.int32).contiguous(), dst[order].to(torch.int32).contiguous()
    indptr, indices, eids = _capi.coo_to_csr(dst, src, None, C3_NODES, C3_NODES)   # rows = dst
    assert eids is not None and not bool((eids[1:] > eids[:-1]).all())              # a scattered map
    return {"n": C3_NODES, "e": e, "src": src, "dst": dst, "indptr": indptr, "indices": indices, "eids": eids,
            "csr": _capi.make_csr(indptr, indices, eids, C3_NODES), "coo": _capi.make_coo(src, dst, None, C3_NODES, C3_NODES),
            "host": tuple(_h(t) for t in (indptr, indices, eids, src, dst))}


def _feat(dev, seed, *shape):
    torch.manual_seed(seed)
    return torch.rand(*shape, device=dev) + 1          # U(0,1)+1


def test_c3_u_add_v_bit_exact(dev, c3):
    from dgl_amd import _capi

    el, er = _feat(dev, 1, c3["n"], HEADS, 1), _feat(dev, 2, c3["n"], HEADS, 1)
    out = torch.empty(c3["e"], HEADS, 1, device=dev)
    _capi.sddmm_coo("add", c3["coo"], el, er, out, 0, 2)
    ip, ix, ei, src, dst = c3["host"]
    ref = oracle.sddmm_coo("add", src, dst, None, _h(el), _h(er), "u", "v")
    np.testing.assert_array_equal(_h(out).reshape(ref.shape), ref)


@pytest.mark.parametrize("d", [8, 32])
def test_c3_u_dot_v_plain_bar(dev, c3, d):
    from dgl_amd import _capi

    ft = _feat(dev, 3 + d, c3["n"], HEADS, d)
    out = torch.empty(c3["e"], HEADS, 1, device=dev)
    _capi.sddmm_coo("dot", c3["coo"], ft, ft, out, 0, 2)
    ip, ix, ei, src, dst = c3["host"]
    ref = oracle.sddmm_coo("dot", src, dst, None, _h(ft), _h(ft), "u", "v")
    err = max_rel_err(_h(out).reshape(ref.shape), ref)
    assert err <= 1e-5, "u_dot_v D=%d: plain max rel err vs the oracle %.3g" % (d, err)


@pytest.mark.parametrize("d", [8, 32])
def test_c3_u_dot_v_mixed_signs_condition_aware(dev, c3, d):
    """EXTRA case (not the bar): mixed-sign rows cancel inside the dot product, so the relative error
    of a near-zero result is unbounded for ANY summation order; what holds is |err| <= 1e-5 sum|a||b|."""
    from dgl_amd import _capi

    torch.manual_seed(40 + d)
    ft = torch.rand(c3["n"], HEADS, d, device=dev) - 0.3
    out = torch.empty(c3["e"], HEADS, 1, device=dev)
    _capi.sddmm_coo("dot", c3["coo"], ft, ft, out, 0, 2)
    ip, ix, ei, src, dst = c3["host"]
    ref = oracle.sddmm_coo("dot", src, dst, None, _h(ft), _h(ft), "u", "v")
    mag = oracle.sddmm_coo("dot", src, dst, None, np.abs(_h(ft)), np.abs(_h(ft)), "u", "v")
    got = _h(out).reshape(ref.shape)
    assert float(np.max(np.abs(got - ref) / np.maximum(mag, 1e-30))) <= 1e-5


def _scores(dev, c3):
    """GATConv's score: leaky_relu(el[src] + er[dst]) in edge-id order."""
    el, er = _feat(dev, 1, c3["n"], HEADS, 1), _feat(dev, 2, c3["n"], HEADS, 1)
    return F.leaky_relu(el[c3["src"].long()] + er[c3["dst"].long()] - 3.0, 0.2).contiguous()   # both signs occur


def test_c3_edge_softmax_forward_plain_bar(dev, c3):
    from dgl_amd import _capi

    score = _scores(dev, c3)
    a = torch.empty_like(score)
    ws = torch.empty(_capi.edge_softmax_workspace_bytes(c3["csr"], score.dtype, HEADS), dtype=torch.uint8, device=dev)
    _capi.edge_softmax_forward(c3["csr"], score, a, ws)
    ip, ix, ei, _, _ = c3["host"]
    ref = oracle.edge_softmax_fwd(ip, ei, _h(score).reshape(c3["e"], HEADS))
    err = max_rel_err(_h(a).reshape(ref.shape), ref)
    assert err <= 1e-5, "edge softmax forward: plain max rel err vs the oracle %.3g" % err
    # and through the public operator default (plain) and with the opt-in position-ordered hand-off
    import dgl_amd as dgl
    from dgl_amd import edge_order as E

    g = dgl.graph((c3["src"], c3["dst"]), num_nodes=c3["n"], idtype=torch.int32, device=dev)
    got = dgl.edge_softmax(g, score)
    assert type(got) is torch.Tensor                              # default: plain edge-id order
    err = max_rel_err(_h(got).reshape(ref.shape), ref)
    assert err <= 1e-5, "dgl.edge_softmax: plain max rel err vs the oracle %.3g" % err
    with dgl.edge_order_handoff():                                # opt-in hand-off: same values
        got = dgl.edge_softmax(g, score)
    assert type(got) is E.PosOrdered
    err = max_rel_err(_h(E.to_eid_order(got)).reshape(ref.shape), ref)
    assert err <= 1e-5, "dgl.edge_softmax (hand-off): plain max rel err vs the oracle %.3g" % err


@pytest.mark.parametrize("d", [8, 32])
def test_c3_u_mul_e_sum_plain_bar(dev, c3, d):
    from dgl_amd import _capi

    score = _scores(dev, c3)
    ip, ix, ei, _, _ = c3["host"]
    a_host = oracle.edge_softmax_fwd(ip, ei, _h(score).reshape(c3["e"], HEADS)).reshape(c3["e"], HEADS, 1)
    a = torch.from_numpy(a_host).to(dev)                       # the SAME attention on both sides
    ft = _feat(dev, 9 + d, c3["n"], HEADS, d)
    o = torch.empty(c3["n"], HEADS, d, device=dev)
    ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes("mul", "sum", c3["csr"], o.dtype, ft, a, o)),
                     dtype=torch.uint8, device=dev)
    _capi.spmm_csr("mul", "sum", c3["csr"], ft, a, o, None, None, ws)
    ref, _, _ = oracle.spmm_csr("mul", "sum", ip, ix, ei, _h(ft), a_host)
    err = max_rel_err(_h(o).reshape(ref.shape), ref)
    assert err <= 1e-5, "u_mul_e_sum D=%d: plain max rel err vs the oracle %.3g" % (d, err)


@pytest.mark.parametrize("op", ["copy_lhs", "mul"])
@pytest.mark.parametrize("reduce", ["max", "min"])
def test_c3_max_min_and_args_bit_exact(dev, c3, op, reduce):
    from dgl_amd import _capi

    d = 8
    torch.manual_seed(17)
    ft = torch.round(torch.rand(c3["n"], HEADS, d, device=dev) * 64) / 16 + 1      # a coarse grid: many exact ties
    w = (torch.round(torch.rand(c3["e"], HEADS, 1, device=dev) * 8) / 4 + 0.5) if op == "mul" else None
    o = torch.empty(c3["n"], HEADS, d, device=dev)
    au = torch.empty(c3["n"], HEADS, d, dtype=torch.int32, device=dev)
    ae = torch.empty_like(au)
    ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes(op, reduce, c3["csr"], o.dtype, ft, w, o)),
                     dtype=torch.uint8, device=dev)
    _capi.spmm_csr(op, reduce, c3["csr"], ft, w, o, au, ae, ws)
    ip, ix, ei, _, _ = c3["host"]
    ref, ru, re_ = oracle.spmm_csr(op, reduce, ip, ix, ei, _h(ft), _h(w))
    np.testing.assert_array_equal(_h(o).reshape(ref.shape), ref)
    np.testing.assert_array_equal(_h(au).reshape(ru.shape), ru)
    if re_ is not None:
        np.testing.assert_array_equal(_h(ae).reshape(re_.shape), re_)


def test_c3_gat_layer_end_to_end_plain_bar(dev, c3):
    """The whole GATConv message-passing block through the public API (apply_edges -> leaky_relu ->
    edge_softmax -> update_all) against the oracle run operator by operator."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    d = 8
    el, er, ft = _feat(dev, 1, c3["n"], HEADS, 1), _feat(dev, 2, c3["n"], HEADS, 1), _feat(dev, 5, c3["n"], HEADS, d)
    g = dgl.graph((c3["src"], c3["dst"]), num_nodes=c3["n"], idtype=torch.int32, device=dev)
    with g.local_scope():
        g.srcdata.update({"ft": ft, "el": el})
        g.dstdata.update({"er": er})
        g.apply_edges(fn.u_add_v("el", "er", "e"))
        g.edata["a"] = dgl.edge_softmax(g, F.leaky_relu(g.edata.pop("e") - 3.0, 0.2))
        g.update_all(fn.u_mul_e("ft", "a", "m"), fn.sum("m", "o"))
        out = g.dstdata["o"]
    ip, ix, ei, src, dst = c3["host"]
    e_ref = oracle.sddmm_coo("add", src, dst, None, _h(el), _h(er), "u", "v").reshape(c3["e"], HEADS)
    s_ref = _h(F.leaky_relu(torch.from_numpy(e_ref) - 3.0, 0.2))
    a_ref = oracle.edge_softmax_fwd(ip, ei, s_ref)
    ref, _, _ = oracle.spmm_csr("mul", "sum", ip, ix, ei, _h(ft), a_ref.reshape(c3["e"], HEADS, 1))
    err = max_rel_err(_h(out).reshape(ref.shape), ref)
    assert err <= 1e-5, "GAT block: plain max rel err vs the oracle %.3g" % err


def test_c5_stacked_bf16_launch_equals_rounded_fp32_oracle(dev):
    """configs[4] at 1/8 size: 8 relations x 1.5 M edges, 1.25 M nodes, F = 256, bf16, ONE stacked launch."""
    from dgl_amd import _capi
    from dgl_amd.graph_index import stack_csc

    n, e, f, r = 1_250_000, 1_500_000, 256, 8
    torch.manual_seed(3)
    x = (torch.rand(n, f, device=dev) + 1).to(torch.bfloat16)
    gs = [synth_csr(n, n, e, "U", seed=100 + k, device=dev) for k in range(r)]
    indptr, indices, eids, relid = stack_csc([(g["indptr"], g["indices"], None) for g in gs], n, torch.int32)
    scsr = _capi.make_csr(indptr, indices, eids, n)
    out = torch.empty(n, f, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", scsr, x, None, out), dtype=torch.uint8, device=dev)
    _capi.spmm_csr_stacked("copy_lhs", scsr, relid, [x] * r, None, out, ws)
    torch.cuda.synchronize()
    xf = x.float().cpu().numpy()
    acc = np.zeros((n, f), dtype=np.float32)
    for g in gs:   # the reference's loop: every relation adds into the same running fp32 output
        oracle.spmm_csr("copy_lhs", "sum", _h(g["indptr"]), _h(g["indices"]), None, xf, None, out=acc)
    want = torch.from_numpy(acc).to(torch.bfloat16)
    got = out.cpu()
    same = (want.view(torch.int16) == got.view(torch.int16))
    # a running fp32 sum that lands within an fp32 rounding of a bf16 tie may round the other way:
    # allow one bf16 ulp on those, and count them
    off = (~same).sum().item()
    ulp = (want.view(torch.int16).int() - got.view(torch.int16).int()).abs().max().item()
    assert ulp <= 1 and off <= 1e-4 * same.numel(), (off, ulp)
    assert max_rel_err(got.float().numpy(), acc) <= 2.0 ** -8


def test_c5_full_size_stacked_bf16_launch_equals_rounded_fp32_oracle(dev):
    """configs[4] AT FULL SIZE (VERDICT r4 Next #3): 8 relations x 12.5 M edges, 10 M nodes, F = 256, bf16, ONE
    stacked launch, against the reference's own CPU kernel looped over the relations in fp32 and rounded to bf16
    (the rule of the 1/8-size test above).  5 GB of features, 100 M edges; the host side takes most of the time."""
    from dgl_amd import _capi
    from dgl_amd.graph_index import stack_csc

    n, e, f, r = 10_000_000, 12_500_000, 256, 8
    torch.manual_seed(3)
    x = torch.rand(n, f, device=dev).add_(1).to(torch.bfloat16)
    gs = [synth_csr(n, n, e, "U", seed=100 + k, device=dev, sort_cols=False) for k in range(r)]
    indptr, indices, eids, relid = stack_csc([(g["indptr"], g["indices"], None) for g in gs], n, torch.int32)
    scsr = _capi.make_csr(indptr, indices, eids, n)
    out = torch.empty(n, f, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", scsr, x, None, out), dtype=torch.uint8, device=dev)
    _capi.spmm_csr_stacked("copy_lhs", scsr, relid, [x] * r, None, out, ws)
    torch.cuda.synchronize()
    xf = x.cpu().float().numpy()
    acc = np.zeros((n, f), dtype=np.float32)
    for g in gs:   # the reference's loop: every relation adds into the same running fp32 output
        oracle.spmm_csr("copy_lhs", "sum", _h(g["indptr"]), _h(g["indices"]), None, xf, None, out=acc)
    del xf
    want = torch.from_numpy(acc).to(torch.bfloat16)
    got = out.cpu()
    wi, gi = want.view(torch.int16), got.view(torch.int16)
    diff = wi != gi
    off = int(diff.sum())
    # a running fp32 sum that lands within an fp32 rounding of a bf16 tie may round the other way: one bf16 ulp
    ulp = int((wi[diff].int() - gi[diff].int()).abs().max()) if off else 0
    assert ulp <= 1 and off <= 1e-4 * wi.numel(), (off, ulp)


def test_papers100m_shaped_sampling_step_block_structure_is_exact(dev):
    """configs[3]'s own graph size (VERDICT r4 Next #3): ``NeighborSampler([15, 10]).sample_blocks`` on a synthetic
    in-edge CSR with ogbn-papers100M's 111 059 956 nodes / 1 615 685 872 edges.  Picks come from our own counter-based
    stream (not the reference's Philox), so what is pinned is everything that is NOT random, re-evaluated on the host
    from the picks the GPU reports: every pick is an in-edge of its seed, min(degree, fanout) distinct picks per seed,
    and the block built from them — destination nodes first, then new sources by ascending id, local ids, indptr —
    is bit-exact (dgl.to_block, python/dgl/transforms/functional.py; neighbor_sampler.py sample_blocks)."""
    import dgl_amd as dgl
    from dgl_amd.graph_index import GraphIndex, Relation
    from dgl_amd.heterograph import DGLGraph

    n, e = 111_059_956, 1_615_685_872
    gen = torch.Generator(device=dev).manual_seed(20250824)
    raw = torch.empty(n, device=dev).log_normal_(1.9, 1.2, generator=gen).clamp_(max=20000.0)
    raw = raw.double()
    deg = torch.floor(raw * (e / float(raw.sum()))).long()
    short = int(e - int(deg.sum()))
    assert 0 <= short < n
    deg[:short] += 1                                           # hand the rounding remainder to the first rows
    assert int(deg.sum()) == e
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=indptr[1:])
    del raw
    indices = torch.randint(0, n, (e,), device=dev, generator=gen)          # columns need no order inside a row
    rel = Relation(n, n, csc=(indptr, indices, None), idtype=torch.int64, device=dev)
    g = DGLGraph(GraphIndex([n], [(0, 0)], [rel]), ["_N"], [("_N", "_E", "_N")])
    seeds = torch.randint(0, n, (4096,), device=dev, generator=gen).unique()
    fanouts = [15, 10]
    inp, outp, blocks = dgl.NeighborSampler(fanouts, seed=7).sample_blocks(g, seeds)
    assert torch.equal(outp, seeds) and len(blocks) == 2
    dst = seeds
    for blk, fanout in zip(reversed(blocks), reversed(fanouts)):
        dst_h = dst.cpu().numpy()
        assert np.array_equal(_h(blk.dstdata[dgl.NID]), dst_h)
        bptr, local, _ = blk._graph.relations[0].csc()
        bptr_h, local_h = _h(bptr).astype(np.int64), _h(local).astype(np.int64)
        eid = blk.edata[dgl.EID]                                             # map-free graph: edge id == CSC position
        eid_h = _h(eid)
        lo, hi = _h(indptr[dst.long()]), _h(indptr[dst.long() + 1])
        want_cnt = np.minimum(hi - lo, fanout)
        assert np.array_equal(np.diff(bptr_h), want_cnt) and bptr_h[0] == 0 and bptr_h[-1] == eid_h.shape[0]
        row_of = np.repeat(np.arange(dst_h.shape[0]), want_cnt)
        assert ((eid_h >= lo[row_of]) & (eid_h < hi[row_of])).all()          # every pick is an in-edge of its seed
        key = row_of.astype(np.int64) * (1 << 32) + (eid_h - lo[row_of])
        assert np.unique(key).shape[0] == key.shape[0]                        # distinct picks per seed (no replacement)
        src_pick = _h(indices[eid.long()])
        # host to_block: destination nodes first, then the other sources by ascending id
        extra = np.setdiff1d(np.unique(src_pick), dst_h)
        src_nodes = np.concatenate([dst_h, extra])
        assert np.array_equal(_h(blk.srcdata[dgl.NID]), src_nodes)
        order = np.argsort(src_nodes, kind="stable")
        want_local = order[np.searchsorted(src_nodes[order], src_pick)]
        assert np.array_equal(local_h, want_local)
        assert blk.num_src_nodes() == src_nodes.shape[0] and blk.num_dst_nodes() == dst_h.shape[0]
        dst = blk.srcdata[dgl.NID]
    assert torch.equal(inp, dst)
