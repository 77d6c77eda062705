"""Fused multi-relation SpMM (one launch for all relations sharing a destination type):
the stacking helper on CPU, the C seam and the hetero update_all route on GPU."""
import numpy as np
import pytest
import torch

import oracle
from tests.graphgen import coo_to_csc


def _relations(rng, n_srcs, n_dst, n_edges, idtype=np.int32):
    rels = []
    for ns, ne in zip(n_srcs, n_edges):
        src = rng.integers(0, ns, ne)
        dst = np.minimum((rng.random(ne) ** 2 * n_dst).astype(np.int64), n_dst - 1)
        rels.append((ns, src, dst) + coo_to_csc(src, dst, n_dst, idtype))
    return rels


def test_stack_csc_layout():
    from dgl_amd.graph_index import stack_csc

    rng = np.random.default_rng(0)
    n_dst = 37
    rels = _relations(rng, [20, 5, 11], n_dst, [150, 0, 60])
    t = torch.from_numpy
    indptr, indices, eids, rel = stack_csc([(t(ip), t(ix), t(ei)) for _, _, _, ip, ix, ei in rels],
                                           n_dst, torch.int32)
    indptr, indices, eids, rel = (x.numpy() for x in (indptr, indices, eids, rel))
    assert indptr[-1] == 210 and indptr.dtype == np.int32
    for r in range(n_dst):
        want_i, want_e, want_k = [], [], []
        for k, (_, _, _, ip, ix, ei) in enumerate(rels):
            want_i += list(ix[ip[r]:ip[r + 1]])
            want_e += list(ei[ip[r]:ip[r + 1]])
            want_k += [k] * (ip[r + 1] - ip[r])
        sl = slice(indptr[r], indptr[r + 1])
        assert list(indices[sl]) == want_i and list(eids[sl]) == want_e and list(rel[sl]) == want_k


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["copy_lhs", "mul", "copy_rhs"])
@pytest.mark.parametrize("shape", [((12,), (12,)), ((4, 8), (4, 1)), ((5,), (5,)), ((100,), (100,))])
@pytest.mark.parametrize("tdtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("idtype", [np.int32, np.int64])
def test_stacked_seam_equals_sum_of_relations(dev, op, shape, tdtype, idtype):
    from dgl_amd import _capi
    from dgl_amd.graph_index import stack_csc

    rng = np.random.default_rng(7)
    n_dst = 300
    rels = _relations(rng, [200, 50, 1, 120], n_dst, [4000, 700, 30, 2500], idtype)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cscs = [(t(ip), t(ix), t(ei)) for _, _, _, ip, ix, ei in rels]
    indptr, indices, eids, rel = stack_csc(cscs, n_dst, cscs[0][0].dtype)
    us = [(torch.rand((ns,) + shape[0], device=dev, dtype=torch.float64) + 1).to(tdtype)
          for ns, *_ in rels]
    es = [(torch.rand((len(src),) + shape[1], device=dev, dtype=torch.float64) + 1).to(tdtype)
          for _, src, *_ in rels]
    use_u, use_e = op != "copy_rhs", op != "copy_lhs"
    want = 0
    for (ns, src, dst, ip, ix, ei), u, e in zip(rels, us, es):
        f64 = lambda x: x.double().cpu().numpy()
        r, _, _ = oracle.spmm_csr(op, "sum", ip, ix, ei, f64(u) if use_u else None,
                                  f64(e) if use_e else None)
        want = want + r
    csr = _capi.make_csr(indptr, indices, eids, max(ns for ns, *_ in rels))
    out = torch.full(want.shape, 3.0, dtype=tdtype, device=dev)
    ws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes(op, csr, us[0] if use_u else None,
                                                            es[0] if use_e else None, out),
                     dtype=torch.uint8, device=dev)
    tabs = _capi.spmm_csr_stacked(op, csr, rel, us if use_u else None, es if use_e else None, out, ws)
    tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.bfloat16: 1.2e-2}[tdtype]
    np.testing.assert_allclose(out.double().cpu().numpy(), want, rtol=tol, atol=tol)
    # accumulate on top + cached plan
    base = out.clone()
    _capi.spmm_csr_stacked(op, csr, rel, us if use_u else None, es if use_e else None, out, ws,
                           u_table=tabs[0], e_table=tabs[1], accumulate=True, plan_valid=True)
    np.testing.assert_allclose(out.double().cpu().numpy(), 2 * base.double().cpu().numpy(),
                               rtol=max(tol, 1e-6), atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("msg", ["copy_u", "u_mul_e", "copy_e"])
def test_hetero_update_all_uses_one_launch_and_matches_loop(dev, msg, monkeypatch):
    """Three relations into 'user' (+ one into 'item'): the fused route must give what the
    reference's per-relation accumulate loop gives."""
    import dgl_amd as dgl
    import dgl_amd.function as fn
    from dgl_amd import sparse_kernels

    rng = np.random.default_rng(11)
    nn = {"user": 120, "item": 80, "tag": 15}
    ed = {("user", "follows", "user"): 900, ("item", "bought_by", "user"): 1500,
          ("tag", "marks", "user"): 200, ("user", "buys", "item"): 700}
    data = {k: (torch.from_numpy(rng.integers(0, nn[k[0]], n)), torch.from_numpy(rng.integers(0, nn[k[2]], n)))
            for k, n in ed.items()}
    g = dgl.heterograph(data, num_nodes_dict=nn, idtype=torch.int32, device=dev)
    for nt, n in nn.items():
        g.nodes[nt].data["h"] = torch.rand(n, 4, 8, device=dev) + 1
    for k, n in ed.items():
        g.edges[k].data["w"] = torch.rand(n, 4, 1, device=dev) + 1
    mfunc = {"copy_u": fn.copy_u("h", "m"), "u_mul_e": fn.u_mul_e("h", "w", "m"),
             "copy_e": fn.copy_e("w", "m")}[msg]

    calls = []
    real_call = sparse_kernels._call

    def spy(name, *a):
        calls.append(name)
        return real_call(name, *a)

    real_hetero = sparse_kernels._call_hetero

    def spy_hetero(name, *a):
        calls.append(name)
        return real_hetero(name, *a)

    monkeypatch.setattr(sparse_kernels, "_call", spy)
    monkeypatch.setattr(sparse_kernels, "_call_hetero", spy_hetero)
    g.update_all(mfunc, fn.sum("m", "o"))
    fused = {nt: g.nodes[nt].data["o"].clone() for nt in ("user", "item")}
    assert calls.count("sparse._CAPI_DGLKernelSpMMStacked") == 1       # user: 3 relations, 1 launch
    assert calls.count("sparse._CAPI_DGLKernelSpMMHetero") == 1        # item: the reference's loop
    assert calls.count("sparse._CAPI_DGLKernelSpMM") == 0
    monkeypatch.setattr(sparse_kernels, "_FUSED_OPS", ())
    g.update_all(mfunc, fn.sum("m", "o"))
    for nt in fused:
        np.testing.assert_allclose(fused[nt].cpu().numpy(), g.nodes[nt].data["o"].cpu().numpy(),
                                   rtol=1e-5, atol=1e-6)
    # gradients flow through the fused route as well
    monkeypatch.setattr(sparse_kernels, "_FUSED_OPS", ("copy_lhs", "copy_rhs", "mul"))
    if msg != "copy_e":
        h = g.nodes["item"].data["h"].clone().requires_grad_(True)
        g.nodes["item"].data["h"] = h
        g.update_all(mfunc, fn.sum("m", "o"))
        g.nodes["user"].data["o"].sum().backward()
        assert h.grad is not None and float(h.grad.abs().sum()) > 0
