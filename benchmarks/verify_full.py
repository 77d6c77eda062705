#!/usr/bin/env python
"""Full-size parity of BASELINE configs[2] and configs[4] against the CPU oracle (VERDICT r2
Missing #7 / Next #1a).  The oracle here is the CHECKER (oracle/ = the reference's CPU kernels
restated in C and pinned to the reference build); the thing checked is the HIP path through the
C ABI.  One JSON line per check; exit code 1 if any check fails.

configs[2] (GATConv pieces, ogbn-arxiv shape + reverse + self loops: N = 169 343, E = 2 501 829,
H = 8, fp32, int32 ids, WITH DGL's usual random edge-id map):
    g-SDDMM u_add_v (bit-exact: one add per element), u_dot_v D = 8 / 32 (1e-5),
    edge softmax forward / backward (1e-5 of the reference's src/array/cpu/spmm.h:484-570 loop),
    g-SpMM u_mul_e_sum (H, D) x (H, 1) (1e-5; fp32-sum rule of tests/tolerance.py, plain figure beside it)
configs[4] (R-GCN, 8 relations x 12.5 M edges on 10 M nodes, F = 256, bf16):
    ONE stacked launch (dgla_spmm_csr_stacked) against the fp32 sum of the same bf16 values
    (reference: fp32 accumulator for 16-bit storage, src/array/cpu/spmm.h:78-109), 16-bit
    tolerance 2^-8; and the stacked max + type trackers (8 x 2.5 M edges into 2 M nodes,
    F = 64, fp32 and bf16): values, winning source node / edge id, node / edge type BIT-EXACT
    against oracle.spmm_csr_hetero (= SpMMCmpCsrHetero, spmm.h:341-408).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402  (the checker)
from dgl_amd import _capi  # noqa: E402
from tests.graphgen import synth_csr  # noqa: E402
from tests.tolerance import max_rel_err  # noqa: E402

FAILED = []
NT = min(64, os.cpu_count() or 1)


def emit(cfg, name, ok, **kw):
    r = {"config": cfg, "check": name, "ok": bool(ok)}
    r.update(kw)
    print(json.dumps(r), flush=True)
    if not ok:
        FAILED.append(name)


def h(t):
    return None if t is None else t.detach().cpu().numpy()


def c3(dev, scale):
    n, e, hh = 169_343 // scale, 2_501_829 // scale, 8
    g = synth_csr(n, n, e, "U", device=dev, with_eids=True)
    deg = (g["indptr"][1:] - g["indptr"][:-1]).long()
    col = torch.repeat_interleave(torch.arange(n, device=dev), deg).to(g["indices"].dtype)  # dst
    row = g["indices"]                                                                       # src
    coo = _capi.make_coo(row, col, g["eids"], n, n)
    csr = _capi.make_csr(g["indptr"], g["indices"], g["eids"], n)
    ip, ix, ei = h(g["indptr"]), h(g["indices"]), h(g["eids"])
    torch.manual_seed(2)
    el = torch.rand(n, hh, 1, device=dev) - 0.5
    er = torch.rand(n, hh, 1, device=dev) - 0.5
    out = torch.empty(e, hh, 1, device=dev)
    # ---- g-SDDMM u_add_v ------------------------------------------------------------------
    _capi.sddmm_coo("add", coo, el, er, out, 0, 2)
    ref = oracle.sddmm_coo("add", h(row), h(col), ei, h(el), h(er), "u", "v", nthreads=NT)
    got = h(out).reshape(ref.shape)
    emit("C3", "sddmm u_add_v (H=8), eid map", np.array_equal(got, ref), edges=e,
         bit_exact=bool(np.array_equal(got, ref)), max_rel_err_vs_oracle=max_rel_err(got, ref))
    score = torch.nn.functional.leaky_relu(out, 0.2)
    # ---- g-SDDMM u_dot_v --------------------------------------------------------------------
    for d in (8, 32):
        ft = torch.rand(n, hh, d, device=dev) - 0.3
        _capi.sddmm_coo("dot", coo, ft, ft, out, 0, 2)
        ref = oracle.sddmm_coo("dot", h(row), h(col), ei, h(ft), h(ft), "u", "v", nthreads=NT)
        got = h(out).reshape(ref.shape)
        # condition-aware: |err| <= 1e-5 * sum |a||b| (a dot of mixed signs can cancel)
        mag = oracle.sddmm_coo("dot", h(row), h(col), ei, np.abs(h(ft)), np.abs(h(ft)), "u", "v", nthreads=NT)
        worst = float(np.max(np.abs(got - ref) / np.maximum(mag, 1e-30)))
        emit("C3", "sddmm u_dot_v (H=8, D=%d), eid map" % d, worst <= 1e-5, edges=e,
             max_err_over_sum_abs_products=worst, max_rel_err_vs_oracle=max_rel_err(got, ref), tol=1e-5)
    # ---- edge softmax forward / backward -----------------------------------------------------
    a = torch.empty_like(score)
    ws = torch.empty(_capi.edge_softmax_workspace_bytes(csr, score.dtype, hh), dtype=torch.uint8, device=dev)
    _capi.edge_softmax_forward(csr, score, a, ws)
    ref_a = oracle.edge_softmax_fwd(ip, ei, h(score).reshape(e, hh), nthreads=NT)
    got = h(a).reshape(ref_a.shape)
    err = max_rel_err(got, ref_a)
    emit("C3", "edge_softmax forward (H=8), eid map", err <= 1e-5, edges=e, max_rel_err_vs_oracle=err, tol=1e-5)
    gy = torch.rand(e, hh, 1, device=dev) - 0.5
    sds = a * gy
    back = torch.empty_like(a)
    _capi.edge_softmax_backward(csr, a, sds, back, ws, plan_valid=True)
    ref_b = oracle.edge_softmax_bwd(ip, ei, h(a).reshape(e, hh), h(sds).reshape(e, hh), nthreads=NT)
    got = h(back).reshape(ref_b.shape)
    # backward = sds - out * sum_row(sds): a difference; bound the error by the magnitudes entering
    # it, |sds| + |out| * sum_row |sds|  (sum_row |sds| = |sds| - bwd(ones, |sds|))
    asds, aa = np.abs(h(sds).reshape(e, hh)), np.abs(h(a).reshape(e, hh))
    row_sum = asds - oracle.edge_softmax_bwd(ip, ei, np.ones_like(asds), asds, nthreads=NT)
    mag = asds + aa * row_sum
    worst = float(np.max(np.abs(got - ref_b) / np.maximum(mag, 1e-30)))
    emit("C3", "edge_softmax backward (H=8), eid map", worst <= 1e-5, edges=e,
         max_err_over_term_magnitudes=worst, max_rel_err_vs_oracle=max_rel_err(got, ref_b), tol=1e-5)
    # ---- g-SpMM u_mul_e_sum (H, D) x (H, 1) ---------------------------------------------------
    for d in (8, 32):
        ft = torch.rand(n, hh, d, device=dev) + 0.5
        o = torch.empty(n, hh, d, device=dev)
        wss = torch.empty(max(1, _capi.spmm_csr_workspace_bytes("mul", "sum", csr, o.dtype, ft, a, o)),
                          dtype=torch.uint8, device=dev)
        _capi.spmm_csr("mul", "sum", csr, ft, a, o, None, None, wss)
        ref, _, _ = oracle.spmm_csr("mul", "sum", ip, ix, ei, h(ft), h(a), nthreads=NT)
        exact, _, _ = oracle.spmm_csr("mul", "sum", ip, ix, ei, h(ft).astype(np.float64), h(a).astype(np.float64),
                                      nthreads=NT)
        got = h(o).reshape(ref.shape)
        e_ref, e_ex, r_ex = max_rel_err(got, ref), max_rel_err(got, exact), max_rel_err(ref, exact)
        emit("C3", "u_mul_e_sum (H=8, D=%d) x (H, 1), eid map" % d, e_ex <= 1e-5 and (e_ref <= 1e-5 or e_ex <= r_ex),
             edges=e, max_rel_err_vs_oracle=e_ref, max_rel_err_vs_exact_fp64=e_ex,
             oracle_max_rel_err_vs_exact_fp64=r_ex, tol=1e-5)


def c5(dev, scale):
    from dgl_amd.graph_index import stack_csc

    n, e, f, r = 10_000_000 // scale, 12_500_000 // scale, 256, 8
    torch.manual_seed(3)
    x = (torch.rand(n, f, device=dev) + 1).to(torch.bfloat16)
    gs = [synth_csr(n, n, e, "U", seed=100 + k, device=dev) for k in range(r)]
    indptr, indices, eids, relid = stack_csc([(g["indptr"], g["indices"], None) for g in gs], n, torch.int32)
    scsr = _capi.make_csr(indptr, indices, eids, n)
    out = torch.empty(n, f, device=dev, dtype=torch.bfloat16)
    sws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", scsr, x, None, out), dtype=torch.uint8,
                      device=dev)
    _capi.spmm_csr_stacked("copy_lhs", scsr, relid, [x] * r, None, out, sws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    xf = x.float().cpu().numpy()
    acc = np.zeros((n, f), dtype=np.float32)
    for g in gs:  # the reference's loop: every relation adds into the same running fp32 output
        oracle.spmm_csr("copy_lhs", "sum", h(g["indptr"]), h(g["indices"]), None, xf, None, nthreads=NT, out=acc)
    got = out.float().cpu().numpy()
    err = max_rel_err(got, acc)
    emit("C5", "hetero copy_u_sum, 8 relations in ONE stacked launch, bf16 F=256", err <= 2.0 ** -8,
         edges=r * e, max_rel_err_vs_fp32_oracle=err, tol=2.0 ** -8,
         bit_equal_to_rounded_fp32_oracle_fraction=float(np.mean(
             torch.from_numpy(acc).to(torch.bfloat16).float().numpy() == got)),
         oracle_seconds=time.perf_counter() - t0)


def c5max(dev, scale):
    from dgl_amd import sparse_kernels
    from dgl_amd.graph_index import GraphIndex, Relation

    n, e, f, r = 2_000_000 // scale, 2_500_000 // scale, 64, 8
    gs = [synth_csr(n, n, e, "U", seed=200 + k, device=dev, with_eids=True) for k in range(r)]
    rels = [Relation(n, n, csc=(g["indptr"], g["indices"], g["eids"]), idtype=torch.int32, device=dev) for g in gs]
    gidx = GraphIndex([n], [(0, 0)] * r, rels)
    orels = [{"src": 0, "dst": 0, "indptr": h(g["indptr"]), "indices": h(g["indices"]), "eids": h(g["eids"])}
             for g in gs]
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(4)
        x = (torch.round(torch.rand(n, f, device=dev) * 64) / 8 + 1).to(dt)  # 1/8 grid: many exact ties
        for reduce in ("max", "min"):
            res = sparse_kernels._gspmm_hetero(gidx, "copy_lhs", reduce, 1, (x,) + tuple([None] * r))
            o, (au, ae, aut, aet) = res[0][0], [t[0] if t is not None else None for t in res[1]]
            ro, rau, rae, raut, raet = oracle.spmm_csr_hetero("copy_lhs", reduce, orels, [n],
                                                              [x.float().cpu().numpy()], [None] * r)
            ok_v = np.array_equal(o.float().cpu().numpy(), ro[0])
            ok_u = np.array_equal(h(au), rau[0])
            ok_t = np.array_equal(h(aut), raut[0])
            emit("C5", "hetero copy_u_%s + arg_u + node-type tracker, 8 relations stacked, %s F=64" % (reduce, dt),
                 ok_v and ok_u and ok_t, edges=r * e, values_bit_exact=bool(ok_v), arg_u_bit_exact=bool(ok_u),
                 arg_u_ntype_bit_exact=bool(ok_t))
    # an edge operand as well (u_mul_e, scalar weights): winning EDGE ids and edge-type tracker
    w = tuple((torch.round(torch.rand(e, 1, device=dev) * 8) / 4 + 0.5) for _ in range(r))
    x = torch.round(torch.rand(n, 16, device=dev) * 16) / 4 + 1
    res = sparse_kernels._gspmm_hetero(gidx, "mul", "max", 1, (x,) + w)
    o, (au, ae, aut, aet) = res[0][0], [t[0] if t is not None else None for t in res[1]]
    ro, rau, rae, raut, raet = oracle.spmm_csr_hetero("mul", "max", orels, [n], [h(x)], [h(t) for t in w])
    oks = [np.array_equal(h(o), ro[0]), np.array_equal(h(au), rau[0]), np.array_equal(h(ae), rae[0]),
           np.array_equal(h(aut), raut[0]), np.array_equal(h(aet), raet[0])]
    emit("C5", "hetero u_mul_e_max + arg_u / arg_e + both trackers, 8 relations stacked, fp32 F=16, eid maps",
         all(oks), edges=r * e, values_bit_exact=bool(oks[0]), arg_u_bit_exact=bool(oks[1]),
         arg_e_bit_exact=bool(oks[2]), arg_u_ntype_bit_exact=bool(oks[3]), arg_e_etype_bit_exact=bool(oks[4]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--scale", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, fn in (("C3", c3), ("C5", c5), ("C5MAX", c5max)):
        if args.only and name not in args.only.split(","):
            continue
        fn(dev, args.scale)
        torch.cuda.empty_cache()
    if FAILED:
        print("FAILED: %s" % FAILED, file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
