"""Pins the CPU oracle.  The reference stores no golden vectors for this path and cannot be
built or imported here, so the oracle is checked against
 (a) the closed-form / known-answer cases of the reference's own tests and docstrings, and
 (b) independent implementations: scipy.sparse, torch scatter_reduce, dense matmul.
"""
import math

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import oracle
from tests.graphgen import coo_to_csc, coo_to_csr


def test_docstring_copy_u_sum_two_relations():
    # python/dgl/heterograph.py:5092-5104: follows ([0,1]->[1,1]) + attracts ([0]->[1]),
    # user h = [[1],[2]], game h = [[1]]  ->  user h = [[0],[4]] (sum accumulated over relations)
    out = np.zeros((2, 1), np.float32)
    for src, dst, h in (([0, 1], [1, 1], [[1.], [2.]]), ([0], [1], [[1.]])):
        indptr, indices, eids = coo_to_csc(np.array(src), np.array(dst), 2)
        o, _, _ = oracle.spmm_csr("copy_lhs", "sum", indptr, indices, eids,
                                  np.array(h, np.float32), None)
        out += o
    np.testing.assert_array_equal(out, [[0.], [4.]])


def test_docstring_edge_softmax():
    # python/dgl/ops/edge_softmax.py:72-107
    src, dst = np.array([0, 0, 0, 1, 1, 2]), np.array([0, 1, 2, 1, 2, 2])
    indptr, _, eids = coo_to_csc(src, dst, 3)
    e = np.ones((6, 1), np.float32)
    got = oracle.edge_softmax_fwd(indptr, eids, e)
    np.testing.assert_allclose(got[:, 0], [1, .5, 1 / 3, .5, 1 / 3, 1 / 3], rtol=1e-6)
    # norm_by='src' == the same on the reversed graph
    indptr, _, eids = coo_to_csc(dst, src, 3)
    got = oracle.edge_softmax_fwd(indptr, eids, e)
    np.testing.assert_allclose(got[:, 0], [1 / 3, 1 / 3, 1 / 3, .5, .5, 1], rtol=1e-6)
    # first four edges only (eids subset)
    indptr, _, eids = coo_to_csc(src[:4], dst[:4], 3)
    got = oracle.edge_softmax_fwd(indptr, eids, e[:4])
    np.testing.assert_allclose(got[:, 0], [1, .5, 1, .5], rtol=1e-6)


def test_edge_softmax_unidirectional_closed_form():
    # tests/python/common/ops/test_edge_softmax.py:86-114: two relations into the same dst
    # nodes, scores 2 and 1 -> e^2 / ((e^2 + e) * 3) and e / ((e^2 + e) * 3).  With one
    # relation at a time the per-relation softmax is 1/3; the cross-relation form is
    # reproduced by concatenating the relations' edges, which is what the reference does.
    src = np.array([1, 2, 3] * 3 + [0, 1, 2] * 3)
    dst = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2] * 2)
    score = np.concatenate([np.full(9, 2.0), np.full(9, 1.0)]).astype(np.float64)[:, None]
    indptr, _, eids = coo_to_csc(src, dst, 3)
    got = oracle.edge_softmax_fwd(indptr, eids, score)[:, 0]
    np.testing.assert_allclose(got[:9], math.exp(2) / ((math.exp(2) + math.exp(1)) * 3))
    np.testing.assert_allclose(got[9:], math.exp(1) / ((math.exp(2) + math.exp(1)) * 3))


def test_edge_softmax_clique_vs_dense():
    # test_edge_softmax.py:25-58: clique graph == dense softmax over the source axis
    n = 6
    src, dst = np.repeat(np.arange(n), n), np.tile(np.arange(n), n)
    rng = np.random.default_rng(0)
    e = rng.random((n * n, 3, 1))
    indptr, _, eids = coo_to_csc(src, dst, n)
    got = oracle.edge_softmax_fwd(indptr, eids, e)
    dense = torch.softmax(torch.from_numpy(e.reshape(n, n, 3, 1)), 0).numpy().reshape(n * n, 3, 1)
    np.testing.assert_allclose(got, dense, rtol=1e-12)
    # backward vs autograd
    g = rng.random(e.shape)
    t = torch.from_numpy(e.reshape(n, n, 3, 1)).requires_grad_()
    torch.softmax(t, 0).backward(torch.from_numpy(g.reshape(n, n, 3, 1)))
    back = oracle.edge_softmax_bwd(indptr, eids, got, got * g)
    np.testing.assert_allclose(back, t.grad.numpy().reshape(e.shape), rtol=1e-9, atol=1e-12)


def test_star_graph_known_answer():
    # tests/python/common/function/test_basics.py:389-416: nodes 1..4 -> 0; reduce(sum) at
    # node 0 = sum of h[1..4]; zero-degree rows: 0 for sum, -inf/+inf with arg 0 for max/min
    src, dst = np.array([1, 2, 3, 4]), np.zeros(4, dtype=np.int64)
    indptr, indices, eids = coo_to_csc(src, dst, 5)
    h = np.arange(10, dtype=np.float32).reshape(5, 2)
    o, _, _ = oracle.spmm_csr("copy_lhs", "sum", indptr, indices, eids, h, None)
    np.testing.assert_array_equal(o[0], h[1:].sum(0))
    np.testing.assert_array_equal(o[1:], 0)
    o, au, _ = oracle.spmm_csr("copy_lhs", "max", indptr, indices, eids, h, None)
    np.testing.assert_array_equal(o[0], h[4])
    np.testing.assert_array_equal(au[0], [4, 4])
    assert np.all(np.isneginf(o[1:])) and np.all(au[1:] == 0)
    o, _, _ = oracle.spmm_csr("copy_lhs", "min", indptr, indices, eids, h, None)
    assert np.all(np.isposinf(o[1:]))


def test_multigraph_max():
    # test_basics.py:721-751: parallel edges, max over messages
    src, dst = np.array([0, 0, 0, 1]), np.array([1, 1, 1, 0])
    indptr, indices, eids = coo_to_csc(src, dst, 2)
    w = np.array([[1.], [5.], [3.], [2.]], np.float32)
    o, _, ae = oracle.spmm_csr("copy_rhs", "max", indptr, indices, eids, None, w)
    np.testing.assert_array_equal(o, [[2.], [5.]])
    np.testing.assert_array_equal(ae, [[3], [1]])


def test_graphconv_path_graph_dense():
    # tests/python/pytorch/nn/test_nn.py:40-75: GraphConv(norm='none') on a 3-node path graph
    # equals dense A @ X (then W, b)
    src, dst = np.array([0, 1]), np.array([1, 2])
    A = np.zeros((3, 3))
    A[dst, src] = 1
    X = np.random.default_rng(0).random((3, 5))
    indptr, indices, eids = coo_to_csc(src, dst, 3)
    o, _, _ = oracle.spmm_csr("copy_lhs", "sum", indptr, indices, eids, X, None)
    np.testing.assert_allclose(o, A @ X)


@pytest.mark.parametrize("idtype", [np.int32, np.int64])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_vs_scipy_and_torch(idtype, dtype):
    rng = np.random.default_rng(1)
    ns, nd, e = 40, 30, 300
    src, dst = rng.integers(0, ns, e), rng.integers(0, nd, e)
    indptr, indices, eids = coo_to_csc(src, dst, nd, idtype)
    X = (rng.random((ns, 7)) + 1).astype(dtype)
    W = (rng.random((e, 1)) + 1).astype(dtype)
    A = sp.csr_matrix((W[:, 0], (dst, src)), shape=(nd, ns))
    o, _, _ = oracle.spmm_csr("mul", "sum", indptr, indices, eids, X, W)
    np.testing.assert_allclose(o, A @ X, rtol=1e-5 if dtype == np.float32 else 1e-12)
    # max via torch.scatter_reduce(amax, include_self=False)
    msg = torch.from_numpy(X[src] + W)
    ref = torch.full((nd, 7), -np.inf, dtype=msg.dtype).scatter_reduce(
        0, torch.from_numpy(dst)[:, None].expand(-1, 7), msg, "amax", include_self=True)
    o, au, ae = oracle.spmm_csr("add", "max", indptr, indices, eids, X, W)
    np.testing.assert_array_equal(o, ref.numpy())
    # args point at an edge that attains the max, and at the FIRST such edge in CSR order
    for r in range(nd):
        for k in range(7):
            if indptr[r] == indptr[r + 1]:
                assert au[r, k] == 0 and ae[r, k] == 0
                continue
            pos = [j for j in range(indptr[r], indptr[r + 1])
                   if X[indices[j], k] + W[eids[j], 0] == o[r, k]]
            assert indices[pos[0]] == au[r, k] and eids[pos[0]] == ae[r, k]
    # COO flavour agrees with CSR
    o2, au2, ae2 = oracle.spmm_coo("add", "max", src.astype(idtype), dst.astype(idtype), None, nd, X, W)
    np.testing.assert_array_equal(o2, o)
    # SDDMM: u_dot_v and e_mul_v against numpy
    Y = (rng.random((nd, 7)) + 1).astype(dtype)
    d = oracle.sddmm_coo("dot", src.astype(idtype), dst.astype(idtype), None, X, Y)
    np.testing.assert_allclose(d[:, 0], (X[src] * Y[dst]).sum(1), rtol=1e-5)
    ip, ind, ed = coo_to_csr(src, dst, ns, idtype)
    d2 = oracle.sddmm_csr("dot", ip, ind, ed, X, Y)
    np.testing.assert_array_equal(d, d2)
    m = oracle.sddmm_coo("mul", src.astype(idtype), dst.astype(idtype), None, W, Y, "e", "v")
    np.testing.assert_array_equal(m, W * Y[dst])


def test_bcast_tables_match_numpy_broadcasting():
    # src/bcast.cc:36-90 — offsets must index the operands the way numpy broadcasting does
    rng = np.random.default_rng(2)
    for ls, rs in [((1, 2, 1, 3, 1), (4, 1, 3, 1, 1)), ((5, 3, 1, 7), (1, 3, 7, 1)),
                   ((1, 3, 1), (4, 1, 3)), ((3, 3), (1, 3)), ((1,), (3,)), ((3,), (1,))]:
        bc = oracle.BcastOff("mul", (1,) + ls, (1,) + rs)
        a, b = rng.random(ls), rng.random(rs)
        full = (a * b).reshape(-1)
        got = a.reshape(-1)[bc.lhs_offset] * b.reshape(-1)[bc.rhs_offset]
        np.testing.assert_array_equal(got, full)
        assert bc.out_len == full.size
    assert oracle.infer_broadcast_shape("dot", (5, 3, 1, 7), (1, 3, 7, 7)) == (5, 3, 7, 1)
