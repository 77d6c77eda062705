"""Heterograph g-SpMM (several relations reducing into one node type) pinned to the reference:
oracle.spmm_csr_hetero restates SpMMCsrHetero<kDGLCPU> / SpMMCmpCsrHetero
(src/array/cpu/spmm.cc:45-150, spmm.h:341-408) and equals the reference build bit for bit —
sums (whose rounding depends on adding every relation into ONE running output), max / min
values, arg_u / arg_e and the node- / edge-type trackers, ties and empty relations included.
On the GPU both routes of dgl_amd (the reference-style loop behind
sparse._CAPI_DGLKernelSpMMHetero and the fused stacked launch) are held to the committed
reference outputs: integers and max / min bit-exact, sums to 1e-5."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import ref
from tests.hetero_cases import META, NUM_NODES, all_cases, run_case

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_hetero_outputs.npz")
CASES = all_cases(full=False)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libdglref.so not built")
def test_oracle_equals_reference_build_bit_exact():
    bad = []
    for threads in (1, 4):
        ref.set_num_threads(threads)
        for c in all_cases(full=True):
            got, want = run_case(oracle, c), run_case(ref, c)
            if set(got) != set(want):
                bad.append((c["name"], "keys"))
                continue
            for k in want:
                if not (got[k].dtype == want[k].dtype and got[k].shape == want[k].shape and
                        np.array_equal(got[k], want[k])):
                    bad.append((c["name"], k))
    ref.set_num_threads(os.cpu_count() or 1)
    assert not bad, bad[:10]


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_golden(c):
    gold = np.load(GOLDEN)
    got = run_case(oracle, c)
    want = {k[len(c["name"]) + 5:]: gold[k] for k in gold.files if k.startswith(c["name"] + "/out/")}
    assert set(got) == set(want) and want
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_gpu_matches_reference_outputs(dev, c, fused, monkeypatch):
    from dgl_amd import sparse_kernels
    from dgl_amd.graph_index import GraphIndex, Relation

    if not fused:
        monkeypatch.setattr(sparse_kernels, "_FUSED_OPS", ())
    gold = np.load(GOLDEN)
    want = {k[len(c["name"]) + 5:]: gold[k] for k in gold.files if k.startswith(c["name"] + "/out/")}
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    idt = torch.int32 if c["rels"][0]["indptr"].dtype == np.int32 else torch.int64
    rels = [Relation(NUM_NODES[r["src"]], NUM_NODES[r["dst"]], csc=(t(r["indptr"]), t(r["indices"]), t(r["eids"])),
                     idtype=idt, device=dev) for r in c["rels"]]
    gidx = GraphIndex(NUM_NODES, META, rels)
    use_u, use_e = c["op"] != "copy_rhs", c["op"] != "copy_lhs"
    u = tuple(t(f) for f in c["ufeats"]) if use_u else tuple([None] * len(NUM_NODES))
    e = tuple(t(f) for f in c["efeats"]) if use_e else tuple([None] * len(META))
    outs, (au, ae, aut, aet) = sparse_kernels._gspmm_hetero(gidx, c["op"], c["reduce"], len(u), u + e)
    got = {}
    for key, lst in (("out", outs), ("arg_u", au), ("arg_e", ae), ("arg_u_ntype", aut), ("arg_e_etype", aet)):
        for nt, a in enumerate(lst):
            if a is not None:
                got["%s/%d" % (key, nt)] = a.cpu().numpy()
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k, w in want.items():
        assert got[k].dtype == w.dtype and got[k].shape == w.shape, k
        if c["reduce"] == "sum":
            np.testing.assert_allclose(got[k], w, rtol=1e-5 if w.dtype == np.float32 else 1e-12, err_msg=k)
        else:
            np.testing.assert_array_equal(got[k], w, err_msg=k)
