#!/usr/bin/env python
"""Per-kernel microbenchmarks of the hot path beyond the headline (bench.py): every kernel
family at the BASELINE.json config shapes, each with its algorithmic bytes and the achieved
fraction of the HBM peak.  Runs on one MI355X:  python benchmarks/bench_ops.py [--only C3]

Prints one JSON line per (config, op).  Algorithmic-byte models:
  SpMM   E*(F_l*s [+ W_row*s] + i [+ i if eid map]) + (N+1)*i + N*F_out*s
  SDDMM  E*(lhs_row + rhs_row + out_row)*s + 2*E*i            (COO: row + col)
  edge_softmax fwd  E*H*s read + E*H*s written + (N+1)*i (+ E*i eid map)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dgl_amd import _capi  # noqa: E402
from tests.graphgen import C2_EDGES, C2_FEAT, C2_NODES, synth_csr  # noqa: E402

PEAK = 8000.0
VERIFY = False


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        fn()
        ev[k + 1].record()
    torch.cuda.synchronize()
    ts = [ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]
    return float(np.median(ts)), float(np.min(ts))


CACHE_BYTES = 256 << 20  # Infinity Cache: gathered operands smaller than this are served on-die


def emit(cfg, op, e, ms, ms_min, nbytes, gathered_bytes=None, compulsory_bytes=None, **kw):
    """`nbytes` = the no-reuse gather model of SURVEY.md §8d.  That model is roofline evidence only
    when the gathered operand cannot live in the caches: pass `gathered_bytes` (size of the tensor
    rows are gathered from) and `compulsory_bytes` (every byte once) and the line says which
    figure to read — for a cache-resident operand the gather-model fraction exceeds 1 and means
    nothing about HBM (Weak #8)."""
    r = {"config": cfg, "op": op, "edges": e, "ms_median": round(ms, 4), "ms_min": round(ms_min, 4),
         "edges_per_s": e / (ms * 1e-3), "alg_bytes": nbytes,
         "achieved_GBps": nbytes / (ms * 1e-3) / 1e9, "frac_of_8TBps": nbytes / (ms * 1e-3) / 1e9 / PEAK}
    if gathered_bytes is not None:
        resident = gathered_bytes <= CACHE_BYTES
        r["gathered_operand_MB"] = round(gathered_bytes / 1e6, 1)
        r["hbm_roofline_evidence"] = not resident
        if compulsory_bytes is not None:
            r["compulsory_bytes"] = compulsory_bytes
            r["compulsory_frac_of_8TBps"] = compulsory_bytes / (ms * 1e-3) / 1e9 / PEAK
        if resident:
            r["note"] = ("gathered operand fits the 256 MiB Infinity Cache: frac_of_8TBps is the no-reuse "
                         "model and NOT an HBM figure; compulsory_frac_of_8TBps is the HBM-side bound")
    r.update(kw)
    print(json.dumps(r), flush=True)


def spmm_bytes(n_rows, e, f_l, f_out, s, i, w_row=0, eid=False):
    return e * (f_l * s + w_row * s + i + (i if eid else 0)) + (n_rows + 1) * i + n_rows * f_out * s


def run_spmm(cfg, name, g, op, red, u, w, fo, dev, eid=False):
    n = g["num_rows"]
    csr = _capi.make_csr(g["indptr"], g["indices"], g["eids"] if eid else None, g["num_cols"])
    out = torch.empty((n,) + tuple(fo), device=dev, dtype=(u if u is not None else w).dtype)
    idt = g["indptr"].dtype
    au = torch.empty(out.shape, dtype=idt, device=dev) if red != "sum" else None
    ae = torch.empty(out.shape, dtype=idt, device=dev) if red != "sum" else None
    ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes(op, red, csr, out.dtype, u, w, out)),
                     dtype=torch.uint8, device=dev)
    _capi.spmm_csr(op, red, csr, u, w, out, au, ae, ws)
    ms, mn = timeit(lambda: _capi.spmm_csr(op, red, csr, u, w, out, au, ae, ws, plan_valid=True))
    s = out.element_size()
    i = g["indptr"].element_size()
    f_l = 0 if u is None else int(np.prod(u.shape[1:]))
    w_row = 0 if w is None else int(np.prod(w.shape[1:]))
    f_out = int(np.prod(fo))
    nb = spmm_bytes(n, g["nnz"], f_l, f_out, s, i, w_row, eid and w is not None)
    if red != "sum":
        nb += n * f_out * i * ((u is not None) + (w is not None))
    extra = {}
    if VERIFY and out.dtype in (torch.float32, torch.float64):
        # full-size parity against the CPU oracle (tests do this at 1/16 scale; bench.py for copy_u+sum)
        import oracle

        h = lambda t: None if t is None else t.cpu().numpy()
        ref, ru, re_ = oracle.spmm_csr(op, red, h(g["indptr"]), h(g["indices"]), h(g["eids"]) if eid else None,
                                      h(u), h(w), nthreads=min(64, os.cpu_count() or 1))
        got = out.cpu().numpy().reshape(ref.shape)
        if red == "sum":
            extra["parity_max_rel_err_vs_oracle"] = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)))
        else:
            extra["parity_values_bit_exact"] = bool(np.array_equal(got, ref))
            if ru is not None:
                extra["parity_arg_u_bit_exact"] = bool(np.array_equal(au.cpu().numpy().reshape(ru.shape), ru))
            if re_ is not None:
                extra["parity_arg_e_bit_exact"] = bool(np.array_equal(ae.cpu().numpy().reshape(re_.shape), re_))
    # the operand rows are gathered from, and every byte of the problem touched once
    gathered = None if u is None else u.numel() * u.element_size()
    es = out.element_size()
    ib = 4 if idt == torch.int32 else 8
    compulsory = out.numel() * es + (0 if u is None else u.numel() * es) + (0 if w is None else w.numel() * es) + \
        g["nnz"] * ib * (2 if eid else 1) + (n + 1) * ib
    emit(cfg, name, g["nnz"], ms, mn, nb, gathered_bytes=gathered, compulsory_bytes=compulsory,
         dtype=str(out.dtype), idtype=str(idt), **extra)
    return out


def coo_of(g, dev):
    deg = (g["indptr"][1:] - g["indptr"][:-1]).long()
    dst = torch.repeat_interleave(torch.arange(g["num_rows"], device=dev), deg).to(g["indices"].dtype)
    return g["indices"], dst  # row = src, col = dst


def c2(dev, args):
    n, e, f = C2_NODES // args.scale, C2_EDGES // args.scale, C2_FEAT
    g = synth_csr(n, n, e, "U", device=dev, with_eids=True)
    torch.manual_seed(1)
    x = torch.rand(n, f, device=dev) + 1
    w1 = torch.rand(e, 1, device=dev) + 1
    run_spmm("C2", "copy_u_sum", g, "copy_lhs", "sum", x, None, (f,), dev)
    run_spmm("C2", "copy_u_max(+arg_u)", g, "copy_lhs", "max", x, None, (f,), dev)
    run_spmm("C2", "u_mul_e_sum(scalar e, eid map)", g, "mul", "sum", x, w1, (f,), dev, eid=True)
    run_spmm("C2", "u_mul_e_sum(scalar e, no map)", g, "mul", "sum", x, w1, (f,), dev, eid=False)
    # API level ("kernel-only and API-level"): the same reduction through the operator
    # API — output allocation, shape inference, FFI packing, registry call — and through
    # DGLGraph.update_all with the result stored in the node frame; plus one autograd step
    import dgl_amd as dgl
    import dgl_amd.function as fn
    from dgl_amd.graph_index import GraphIndex, Relation
    from dgl_amd.heterograph import DGLGraph

    rel = Relation(n, n, csc=(g["indptr"], g["indices"], None), idtype=g["indptr"].dtype, device=dev)
    dg = DGLGraph(GraphIndex([n], [(0, 0)], [rel]), ["_N"], [("_N", "_E", "_N")])
    ms, mn = timeit(lambda: dgl.ops.copy_u_sum(dg, x))
    emit("C2", "copy_u_sum through dgl.ops.copy_u_sum (API level)", e, ms, mn, spmm_bytes(n, e, f, f, 4, 4))
    dg.ndata["h"] = x

    def ua():
        dg.update_all(fn.copy_u("h", "m"), fn.sum("m", "o"))

    ms, mn = timeit(ua)
    emit("C2", "copy_u_sum through DGLGraph.update_all (API level)", e, ms, mn, spmm_bytes(n, e, f, f, 4, 4))
    # scalar edge weights behind DGL's usual edge-id map, through the operator API: the graph keeps
    # a position-ordered copy of a narrow edge operand BY CONTENT (hash compared on the device), so
    # weights that repeat from call to call (normalisation weights) run map-free
    rel_m = Relation(n, n, csc=(g["indptr"], g["indices"], g["eids"]), idtype=g["indptr"].dtype, device=dev)
    dgm = DGLGraph(GraphIndex([n], [(0, 0)], [rel_m]), ["_N"], [("_N", "_E", "_N")])
    ms, mn = timeit(lambda: dgl.ops.u_mul_e_sum(dgm, x, w1))
    emit("C2", "u_mul_e_sum(scalar e, eid map) through dgl.ops, default (plain map path)",
         e, ms, mn, spmm_bytes(n, e, f, f, 4, 4) + e * 4)
    dgl.static_features(w1)
    ms, mn = timeit(lambda: dgl.ops.u_mul_e_sum(dgm, x, w1))
    dgl.release_static(w1)
    emit("C2", "u_mul_e_sum(scalar e, eid map) through dgl.ops, weights announced static_features",
         e, ms, mn, spmm_bytes(n, e, f, f, 4, 4) + e * 4)
    dgl.set_auto_edge_operand(0)  # opt-in from here to the end of this block
    ms, mn = timeit(lambda: dgl.ops.u_mul_e_sum(dgm, x, w1))
    emit("C2", "u_mul_e_sum(scalar e, eid map) through dgl.ops, same weights every call (opt-in content-keyed copy)",
         e, ms, mn, spmm_bytes(n, e, f, f, 4, 4) + e * 4)
    ws_ = [w1 + k for k in range(4)]
    k_ = [0]

    def changing():
        k_[0] += 1
        return dgl.ops.u_mul_e_sum(dgm, x, ws_[k_[0] % 4])

    ms, mn = timeit(changing)
    emit("C2", "u_mul_e_sum(scalar e, eid map) through dgl.ops, different weights every call (hash + re-gather)",
         e, ms, mn, spmm_bytes(n, e, f, f, 4, 4) + e * 12)
    dgl.set_auto_edge_operand(None)
    del dgm, rel_m, ws_
    xg = x.clone().requires_grad_(True)

    def fwd_bwd():
        xg.grad = None
        dgl.ops.copy_u_sum(dg, xg).sum().backward()

    fwd_bwd()  # builds the reverse graph's CSC once (not timed, like the forward CSC)
    ms, mn = timeit(fwd_bwd, reps=5)
    emit("C2", "copy_u_sum forward + backward (autograd: SpMM on the reverse graph)", 2 * e, ms, mn,
         2 * spmm_bytes(n, e, f, f, 4, 4))
    # max reducer forward + backward (VERDICT r3 Next #7): the backward is ONE dgla_spmm_cmp_backward launch reading
    # the winners in the graph's idtype (the reference: two .long() casts, a gather, a scatter_add_)
    up = torch.rand(n, f, device=dev)

    def fwd_bwd_max():
        xg.grad = None
        o = dgl.ops.copy_u_max(dg, xg)
        o.backward(up)

    fwd_bwd_max()  # (builds the out-edge CSR and the position map of the gather backward once, like the forward CSC)
    ms, mn = timeit(fwd_bwd_max, reps=5)
    words = (f + 31) // 32
    # backward = winner bits (arg_u, the CSC's column ids -> 4 B x words per edge) + a g-SpMM over the reverse graph that
    # reads a dZ row and the bit words per edge
    bwd_bytes = n * f * 8 + e * 4 + e * 4 * words + spmm_bytes(n, e, f, f, 4, 4) + e * (4 * words + 4)
    emit("C2", "copy_u_max forward + backward (backward = winner bits + masked g-SpMM over the reverse graph, no atomics)",
         e, ms, mn, spmm_bytes(n, e, f, f, 4, 4) + n * f * 4 + bwd_bytes)
    g1 = xg.grad.clone()
    fwd_bwd_max()
    print(json.dumps({"config": "C2", "op": "copy_u_max backward (gather path): bits equal across two runs",
                      "deterministic": bool(torch.equal(g1, xg.grad))}), flush=True)
    os.environ["DGLA_CMP_BACKWARD"] = "atomic"
    fwd_bwd_max()
    ms, mn = timeit(fwd_bwd_max, reps=5)
    emit("C2", "copy_u_max forward + backward, DGLA_CMP_BACKWARD=atomic (one dgla_spmm_cmp_backward launch, float atomics)",
         e, ms, mn, spmm_bytes(n, e, f, f, 4, 4) + n * f * 4 + n * f * 16)
    g2 = xg.grad.clone()
    fwd_bwd_max()
    print(json.dumps({"config": "C2", "op": "copy_u_max backward (atomic path): bits equal across two runs",
                      "deterministic": bool(torch.equal(g2, xg.grad)),
                      "max_abs_diff_gather_vs_atomic": float((g1 - g2).abs().max())}), flush=True)
    del os.environ["DGLA_CMP_BACKWARD"]
    from dgl_amd import _capi as _c
    argu = torch.randint(0, n, (n, f), device=dev, dtype=torch.int32)
    dx = torch.zeros(n, f, device=dev)
    ms, mn = timeit(lambda: _c.spmm_cmp_backward(up, argu, dx, atomic=True), reps=5)
    emit("C2", "dgla_spmm_cmp_backward alone (N x F gradient, random winners, atomics)", e, ms, mn, n * f * 12)
    ms, mn = timeit(lambda: dx.scatter_add_(0, argu.long(), up), reps=5)
    emit("C2", "reference composition of the same step: argu.long() + scatter_add_", e, ms, mn, n * f * 12)
    del dg, rel, xg, up, argu, dx
    xh = x.to(torch.bfloat16)
    # F=100 bf16 rows are 200 B: 8-byte aligned only -> exercises the narrow access path
    run_spmm("C2", "copy_u_sum bf16", g, "copy_lhs", "sum", xh, None, (f,), dev)
    # SDDMM at products scale: u_dot_v over F=100 (D=100: 25 of 32 lanes per edge), u_add_v
    row, col = coo_of(g, dev)
    coo = _capi.make_coo(row, col, None, n, n)
    oe = torch.empty(e, 1, device=dev)
    ms, mn = timeit(lambda: _capi.sddmm_coo("dot", coo, x, x, oe, 0, 2), reps=5)
    emit("C2", "sddmm u_dot_v (D=100)", e, ms, mn, e * (2 * f * 4 + 4 + 8), gathered_bytes=n * f * 4,
         compulsory_bytes=2 * n * f * 4 + e * (4 + 8))
    del oe
    if args.big:
        of = torch.empty(e, f, device=dev)
        ms, mn = timeit(lambda: _capi.sddmm_coo("add", coo, x, x, of, 0, 2), reps=3)
        emit("C2", "sddmm u_add_v (F=100)", e, ms, mn, e * (3 * f * 4 + 8))
        del of
    del x, xh, w1, row, col
    # gather rate vs working-set size of X (same E, fewer distinct columns): where does the
    # Infinity Cache (256 MB) / L2 (8 x 4 MB) start to serve the gathers?
    for ncols in (() if args.no_sweep else (n // 4, n // 16, n // 64, n // 256)):
        gg = synth_csr(n, max(ncols, 64), e, "U", device=dev)
        xx = torch.rand(max(ncols, 64), f, device=dev) + 1
        run_spmm("C2", "copy_u_sum, X=%d rows (%.0f MB)" % (xx.shape[0], xx.numel() * 4 / 1e6),
                 gg, "copy_lhs", "sum", xx, None, (f,), dev)
        del gg, xx


def c3(dev, args):
    # ogbn-arxiv-shaped + reverse edges + self loops ~ 2.5 M edges, 8 heads
    n, e, h = 169_343, 2_501_829, 8
    g = synth_csr(n, n, e, "U", device=dev, with_eids=True)
    row, col = coo_of(g, dev)
    coo = _capi.make_coo(row, col, g["eids"], n, n)
    s, i = 4, 4
    torch.manual_seed(2)
    for d in (8, 32, 64):
        el = torch.rand(n, h, 1, device=dev)
        er = torch.rand(n, h, 1, device=dev)
        ft = torch.rand(n, h, d, device=dev)
        out = torch.empty(e, h, 1, device=dev)
        if d == 8:
            ms, mn = timeit(lambda: _capi.sddmm_coo("add", coo, el, er, out, 0, 2))
            emit("C3", "sddmm u_add_v (H=8)", e, ms, mn, e * (3 * h * s + 3 * i))
        ms, mn = timeit(lambda: _capi.sddmm_coo("dot", coo, ft, ft, out, 0, 2))
        emit("C3", "sddmm u_dot_v (H=8,D=%d)" % d, e, ms, mn, e * (2 * h * d * s + h * s + 3 * i),
             gathered_bytes=n * h * d * s, compulsory_bytes=2 * n * h * d * s + e * (h * s + 3 * i))
        if d == 8:
            csr = _capi.make_csr(g["indptr"], g["indices"], g["eids"], n)
            a = torch.empty_like(out)
            ms, mn = timeit(lambda: _capi.edge_softmax_forward(csr, out, a))
            emit("C3", "edge_softmax fwd (H=8) lane-group kernel", e, ms, mn, e * (2 * h * s + i) + (n + 1) * i)
            sds = a * out
            back = torch.empty_like(out)
            ms, mn = timeit(lambda: _capi.edge_softmax_backward(csr, a, sds, back))
            emit("C3", "edge_softmax bwd (H=8) lane-group kernel", e, ms, mn, e * (3 * h * s + i) + (n + 1) * i)
            ws = torch.empty(_capi.edge_softmax_workspace_bytes(csr, out.dtype, h), dtype=torch.uint8, device=dev)
            _capi.edge_softmax_forward(csr, out, a, ws)
            ms, mn = timeit(lambda: _capi.edge_softmax_forward(csr, out, a, ws, plan_valid=True))
            emit("C3", "edge_softmax fwd (H=8) merge-path", e, ms, mn, e * (2 * h * s + i) + (n + 1) * i)
            ms, mn = timeit(lambda: _capi.edge_softmax_backward(csr, a, sds, back, ws, plan_valid=True))
            emit("C3", "edge_softmax bwd (H=8) merge-path", e, ms, mn, e * (3 * h * s + i) + (n + 1) * i)
            # identity edge-id map (COO already sorted by destination): sequential score rows
            csr0 = _capi.make_csr(g["indptr"], g["indices"], None, n)
            ms, mn = timeit(lambda: _capi.edge_softmax_forward(csr0, out, a, ws, plan_valid=True))
            emit("C3", "edge_softmax fwd (H=8) merge-path, eid = position", e, ms, mn, e * (2 * h * s) + (n + 1) * i)
        a = torch.rand(e, h, 1, device=dev)
        run_spmm("C3", "u_mul_e_sum (H=8,D=%d)x(H,1)" % d, g, "mul", "sum", ft, a, (h, d), dev, eid=True)


def c5(dev, args):
    # R-GCN: 8 relations x 12.5 M edges on 10 M nodes, F=256, bf16; per-relation SpMM accumulating
    n, e, f, r = 10_000_000 // args.scale, 12_500_000 // args.scale, 256, 8
    torch.manual_seed(3)
    x = (torch.rand(n, f, device=dev) + 1).to(torch.bfloat16)
    out = torch.zeros(n, f, device=dev, dtype=torch.bfloat16)
    rel = []
    for k in range(r):
        g = synth_csr(n, n, e, "U", seed=100 + k, device=dev)
        csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
        ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out),
                         dtype=torch.uint8, device=dev)
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, accumulate=True)
        rel.append((g, csr, ws))

    def step():
        out.zero_()
        for g, csr, ws in rel:
            _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, accumulate=True,
                           plan_valid=True)

    ms, mn = timeit(step, reps=5, warm=2)
    s, i = 2, 4
    nb = r * (e * (f * s + i) + (n + 1) * i + 2 * n * f * s) + n * f * s
    emit("C5", "hetero copy_u_sum, 8 relations accumulate, bf16 F=256", r * e, ms, mn, nb)
    # the same reduction as ONE launch over the row-wise stacked CSR (dgla_spmm_csr_stacked)
    from dgl_amd.graph_index import stack_csc
    indptr, indices, eids, relid = stack_csc([(g["indptr"], g["indices"], None) for g, _, _ in rel], n,
                                             torch.int32)
    scsr = _capi.make_csr(indptr, indices, eids, n)
    xs = [x] * r
    sws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", scsr, x, None, out),
                      dtype=torch.uint8, device=dev)
    tabs = _capi.spmm_csr_stacked("copy_lhs", scsr, relid, xs, None, out, sws)
    ms, mn = timeit(lambda: _capi.spmm_csr_stacked("copy_lhs", scsr, relid, xs, None, out, sws,
                                                   u_table=tabs[0], plan_valid=True), reps=5, warm=2)
    nb1 = r * e * (f * s + i + 1) + (n + 1) * i + n * f * s
    emit("C5", "hetero copy_u_sum, 8 relations in ONE stacked launch, bf16 F=256", r * e, ms, mn, nb1)
    del indptr, indices, eids, relid, sws
    g, csr, ws = rel[0]
    o1 = torch.empty_like(out)
    ms, mn = timeit(lambda: _capi.spmm_csr("copy_lhs", "sum", csr, x, None, o1, None, None, ws, plan_valid=True), reps=5)
    emit("C5", "one relation copy_u_sum bf16 F=256 (write)", e, ms, mn, e * (f * s + i) + (n + 1) * i + n * f * s)


def c5max(dev, args):
    """Hetero max with type trackers: 8 relations x 2.5 M edges into one node type of 2 M nodes,
    F = 64 fp32, through the operator layer (`_gspmm_hetero`): the reference's running compare
    relation by relation (2 launches + scratch per relation) against ONE stacked launch."""
    from dgl_amd import sparse_kernels
    from dgl_amd.graph_index import GraphIndex, Relation

    n, e, f, r = 2_000_000 // args.scale, 2_500_000 // args.scale, 64, 8
    torch.manual_seed(4)
    x = torch.rand(n, f, device=dev) + 1
    rels = []
    for k in range(r):
        g = synth_csr(n, n, e, "U", seed=200 + k, device=dev)
        rels.append(Relation(n, n, csc=(g["indptr"], g["indices"], None), idtype=torch.int32, device=dev))
    gidx = GraphIndex([n], [(0, 0)] * r, rels)
    u = (x,)
    e_t = tuple([None] * r)
    s, i = 4, 4
    for fused, tag in ((False, "running compare relation by relation"), (True, "ONE stacked launch + arg pass")):
        sparse_kernels.FUSE_HETERO = fused
        ms, mn = timeit(lambda: sparse_kernels._gspmm_hetero(gidx, "copy_lhs", "max", 1, u + e_t), reps=5, warm=2)
        # compulsory: every edge's source row + index (+ relation byte when stacked), out + 2 id arrays written
        nb = r * e * (f * s + i + (1 if fused else 0)) + (n + 1) * i + n * f * (s + 2 * i)
        emit("C5", "hetero copy_u_max + trackers, 8 relations, fp32 F=64: %s" % tag, r * e, ms, mn, nb)
    sparse_kernels.FUSE_HETERO = True


def seg(dev, args):
    """Segment reduce / scatter add at readout-like shapes: F = 100 fp32.
    (a) 62 M rows in 2.4 M segments of C2's degree sequence (= copy_e SpMM of C2),
    (b) the same rows in 64 huge segments (graph readout over a batch of 64 graphs),
    (c) scatter add of 62 M rows into 2.4 M rows is skipped unless --big (24.7 GB in)
    Algorithmic bytes: rows*F*s read + segments*F*s written + (segments+1)*i."""
    from tests.graphgen import lognormal_degrees

    rows = C2_EDGES // args.scale // 4      # 15.5 M rows x 400 B = 6.2 GB
    f = 100
    torch.manual_seed(3)
    feat = torch.rand(rows, f, device=dev)
    for label, nseg in (("C2-like degree sequence", C2_NODES // args.scale // 4), ("64 equal segments", 64)):
        if nseg == 64:
            lens = np.full(64, rows // 64, dtype=np.int64)
            lens[-1] += rows - lens.sum()
        else:
            lens = lognormal_degrees(nseg, rows)
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])).to(dev)
        for red in ("sum", "max"):
            out = torch.empty(nseg, f, device=dev)
            arg = torch.empty(nseg, f, dtype=torch.int64, device=dev) if red == "max" else None
            ws = torch.empty(max(1, _capi.segment_reduce_workspace_bytes(red, feat, off, out)),
                             dtype=torch.uint8, device=dev)
            _capi.segment_reduce(red, feat, off, out, arg, ws)
            ms, mn = timeit(lambda: _capi.segment_reduce(red, feat, off, out, arg, ws, plan_valid=True))
            nb = rows * f * 4 + nseg * f * 4 + (nseg + 1) * 8 + (nseg * f * 8 if red == "max" else 0)
            emit("SEG", "segment_reduce %s, %d rows -> %d segments (%s), F=100 fp32" % (red, rows, nseg, label),
                 rows, ms, mn, nb)
    # scatter add: rows -> nseg random targets
    nseg = C2_NODES // args.scale // 4
    idx = torch.randint(0, nseg, (rows,), device=dev)
    acc = torch.zeros(nseg, f, device=dev)
    ms, mn = timeit(lambda: _capi.scatter_add(feat, idx, acc), reps=5)
    emit("SEG", "scatter_add %d rows -> %d rows (random idx; sorted, atomic-free path), F=100" % (rows, nseg), rows, ms, mn,
         rows * (f * 4 * 2 + 8))


def mm(dev, args):
    """segment_mm at the R-GCN shape of config 5: 8 relations, D1 = D2 = 256,
    10 M rows split evenly, bf16 / fp16 / fp32; forward, A-gradient (b_trans) and weight
    gradient.  Reports TFLOP/s (2 M K N) next to the algorithmic bytes (A + C once, weights
    once): the transform is memory-bound above ~650 TFLOP/s in bf16."""
    rows, r, k, n = 10_000_000 // args.scale, 8, 256, 256
    seglen = torch.full((r,), rows // r, dtype=torch.int64)
    seglen[-1] += rows - int(seglen.sum())
    sl = seglen.to(dev)
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        torch.manual_seed(0)
        a = (torch.rand(rows, k, device=dev) - 0.5).to(dt)
        b = (torch.rand(r, k, n, device=dev) - 0.5).to(dt)
        c = torch.empty(rows, n, device=dev, dtype=dt)
        s = a.element_size()
        flops = 2.0 * rows * k * n
        nb = rows * (k + n) * s + r * k * n * s
        for name, fn in (
                ("segment_mm fwd", lambda: _capi.segment_mm(a, b, c, sl)),
                ("segment_mm dA (b_trans)", lambda: _capi.segment_mm(c, b, a, sl, b_trans=True)),
                ("segment_mm dB", lambda: _capi.segment_mm_backward_b(a, c, b, sl))):
            ms, mn = timeit(fn, reps=5, warm=2)
            emit("MM", "%s, %d rows x %d x %d, %d relations, %s" % (name, rows, k, n, r, str(dt)), rows, ms, mn,
                 nb, tflops=flops / (ms * 1e-3) / 1e12)
        # what the reference does: one GEMM per relation through the vendor library (hipBLASLt via torch)
        off = [0] + torch.cumsum(seglen, 0).tolist()
        ms, mn = timeit(lambda: [torch.mm(a[off[i]:off[i + 1]], b[i], out=c[off[i]:off[i + 1]]) for i in range(r)],
                        reps=5, warm=2)
        emit("MM", "per-relation torch.mm loop (vendor GEMM; the reference's structure), %s" % str(dt), rows, ms,
             mn, nb, tflops=flops / (ms * 1e-3) / 1e12)
        del a, b, c
    # gather_mm on unsorted rows (HGT-style typed linear): the permutation is read inside the
    # kernels vs the reference's structure (two index_select copies around a segment_mm)
    rows = 10_000_000 // args.scale
    a = (torch.rand(rows, k, device=dev) - 0.5).to(torch.bfloat16)
    b = (torch.rand(r, k, n, device=dev) - 0.5).to(torch.bfloat16)
    idx = torch.randint(0, r, (rows,), device=dev)
    perm = torch.sort(idx, stable=True)[1].contiguous()
    sl = torch.bincount(idx, minlength=r)
    c = torch.empty(rows, n, device=dev, dtype=torch.bfloat16)
    flops, nb = 2.0 * rows * k * n, rows * (k + n) * 2 + r * k * n * 2
    ms, mn = timeit(lambda: _capi.segment_mm(a, b, c, sl, row_index=perm), reps=5, warm=2)
    emit("MM", "gather_mm core, 10 M unsorted rows: segment_mm reading through the permutation, bf16", rows, ms, mn,
         nb, tflops=flops / (ms * 1e-3) / 1e12)
    rev = torch.empty_like(perm)
    rev[perm] = torch.arange(rows, device=dev)
    cs = torch.empty_like(c)

    def ref_structure():
        _capi.segment_mm(torch.index_select(a, 0, perm), b, cs, sl)
        return torch.index_select(cs, 0, rev)

    ms, mn = timeit(ref_structure, reps=5, warm=2)
    emit("MM", "gather_mm core, reference structure: index_select + segment_mm + index_select, bf16", rows, ms, mn,
         nb, tflops=flops / (ms * 1e-3) / 1e12)
    del a, b, c, cs, idx, perm, rev
    # many small relations (the case the grouped launch exists for): 512 relations, 1 M rows
    rows, r = 1_000_000 // args.scale, 512
    lens = np.random.default_rng(0).multinomial(rows, np.ones(r) / r)
    seglen = torch.from_numpy(lens)
    sl = seglen.to(dev)
    a = (torch.rand(rows, k, device=dev) - 0.5).to(torch.bfloat16)
    b = (torch.rand(r, k, n, device=dev) - 0.5).to(torch.bfloat16)
    c = torch.empty(rows, n, device=dev, dtype=torch.bfloat16)
    flops, nb = 2.0 * rows * k * n, rows * (k + n) * 2 + r * k * n * 2
    ms, mn = timeit(lambda: _capi.segment_mm(a, b, c, sl), reps=5, warm=2)
    emit("MM", "segment_mm fwd, %d rows, 512 relations (~%d rows each), bf16" % (rows, rows // r), rows, ms, mn, nb,
         tflops=flops / (ms * 1e-3) / 1e12)
    off = [0] + torch.cumsum(seglen, 0).tolist()
    ms, mn = timeit(lambda: [torch.mm(a[off[i]:off[i + 1]], b[i], out=c[off[i]:off[i + 1]]) for i in range(r)],
                    reps=5, warm=2)
    emit("MM", "per-relation torch.mm loop, 512 relations, bf16", rows, ms, mn, nb, tflops=flops / (ms * 1e-3) / 1e12)


def sample(dev, args):
    """Mini-batch path of config 4 on the C2-shaped graph: uniform neighbour
    sampling, block construction and the block SpMM, for the usual 1 024-seed batch with
    fanouts (15, 10) and for a 256 k-seed layer.  Algorithmic bytes of the sampler: per seed
    2 indptr entries + per pick one column id, one edge id (read) and two ids written."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    n, e = C2_NODES // args.scale, C2_EDGES // args.scale
    gs = synth_csr(n, n, e, "U", device=dev, idtype=torch.int64)
    csr = _capi.make_csr(gs["indptr"], gs["indices"], None, n)
    torch.manual_seed(0)
    i = 8
    for n_seeds, fanout in ((1024, 10), (262144, 15)):
        seeds = torch.randperm(n, device=dev)[:n_seeds].contiguous()
        ms, mn = timeit(lambda: _capi.sample_neighbors(csr, seeds, fanout, False, 1), reps=10, warm=3)
        indptr, src, eids = _capi.sample_neighbors(csr, seeds, fanout, False, 1)
        picks = int(indptr[-1])
        emit("SAMPLE", "sample_neighbors %d seeds, fanout %d (%d picks; includes output allocation)" % (n_seeds, fanout, picks),
             picks, ms, mn, n_seeds * 3 * i + picks * 4 * i)
        node_map = torch.full((n,), -1, dtype=torch.int32, device=dev)
        ms, mn = timeit(lambda: _capi.to_block(seeds, src[:picks], node_map), reps=10, warm=3)
        emit("SAMPLE", "to_block %d seeds + %d sampled sources (sort + renumber + 1 read-back)" % (n_seeds, picks), picks,
             ms, mn, picks * 4 * i)
    # whole 2-layer sample_blocks + SAGE-mean forward on the blocks, F = 100
    rel_g = dgl.graph((gs["indices"], torch.repeat_interleave(torch.arange(n, device=dev), (gs["indptr"][1:] - gs["indptr"][:-1]))),
                      num_nodes=n, device=dev)
    rel_g._graph.relations[0]._csc = (gs["indptr"], gs["indices"], None)
    sampler = dgl.NeighborSampler([15, 10], seed=1)
    feat = torch.rand(n, 100, device=dev)
    seeds = torch.randperm(n, device=dev)[:1024].contiguous()

    def step():
        inp, out, blocks = sampler.sample_blocks(rel_g, seeds)
        h = feat[inp]
        for blk in blocks:
            with blk.local_scope():
                blk.srcdata["h"] = h
                blk.update_all(fn.copy_u("h", "m"), fn.mean("m", "n"))
                h = blk.dstdata["n"]
        return h

    ms, mn = timeit(step, reps=10, warm=3)
    inp, _, blocks = sampler.sample_blocks(rel_g, seeds)
    tot = sum(b.num_edges() for b in blocks)
    emit("SAMPLE", "1024-seed batch end to end: sample_blocks(15,10) + feature gather (%d rows) + 2 x copy_u/mean, F=100"
         % inp.shape[0], tot, ms, mn, inp.shape[0] * 400 * 2 + tot * 408)


def gat_graph(dev, args):
    """Config 3 as a whole layer (GATConv forward: u_add_v -> leaky_relu -> edge softmax ->
    u_mul_e + sum, H = 8, D = 8 / 32) on the ogbn-arxiv-shaped graph: eager launches through the
    Python operator API vs ONE hipGraph replay of the same launches (tests/test_hip_graph.py)."""
    import dgl_amd as dgl
    import dgl_amd.function as fn

    n, e = 169_343, 2_501_829
    g = dgl.rand_graph(n, e, device=dev, seed=1, idtype=torch.int32)
    for d in (8, 32):
        ft = torch.randn(n, 8, d, device=dev)
        el = torch.randn(n, 8, 1, device=dev)
        er = torch.randn(n, 8, 1, device=dev)

        def layer():
            with g.local_scope():
                g.srcdata.update({"ft": ft, "el": el})
                g.dstdata.update({"er": er})
                g.apply_edges(fn.u_add_v("el", "er", "e"))
                sc = torch.nn.functional.leaky_relu(g.edata.pop("e"), 0.2)
                g.edata["a"] = dgl.edge_softmax(g, sc)
                g.update_all(fn.u_mul_e("ft", "a", "m"), fn.sum("m", "o"))
                return g.dstdata["o"]

        layer()
        ms, mn = timeit(layer, reps=20, warm=5)
        nb = e * (8 * 4 * 6 + 8 * d * 4 + 3 * 4) + n * 8 * d * 4
        emit("GAT", "GATConv forward H=8 D=%d, eager (5 launches + torch glue)" % d, e, ms, mn, nb)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            layer()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            layer()
        ms, mn = timeit(graph.replay, reps=20, warm=5)
        emit("GAT", "GATConv forward H=8 D=%d, one hipGraph replay" % d, e, ms, mn, nb)
        # the same block written out once more inside the opt-in hand-off scope — the opt-in scope around the
        # attention block (nothing tagged leaves the scope: the layer's output is a node tensor)
        def layer_scoped():
            with dgl.edge_order_handoff():
                return layer()

        layer_scoped()
        ms, mn = timeit(layer_scoped, reps=20, warm=5)
        emit("GAT", "GATConv forward H=8 D=%d, eager, attention block inside dgl_amd.edge_order_handoff()" % d,
             e, ms, mn, nb)
    # the whole layer written out with the operator API: fc + attention block + bias, forward and forward + backward
    # (in_feats 128 -> 8 heads x 8); the block through dgl_amd.nn.gat_attention — composed operators vs the default route
    lin = torch.nn.Linear(128, 64, bias=False).to(dev)
    al, ar = torch.randn(1, 8, 8, device=dev, requires_grad=True), torch.randn(1, 8, 8, device=dev, requires_grad=True)
    x = torch.randn(n, 128, device=dev, requires_grad=True)
    nb = e * (8 * 4 * 6 + 8 * 8 * 4 + 3 * 4) + n * 8 * 8 * 4
    for route, kw in (("composed", dict(fused=False)), ("composed + hand-off", dict(fused=False, handoff=True)), ("default", {})):
        def fwd():
            f = lin(x).view(n, 8, 8)
            return dgl.nn.gat_attention(g, f, (f * al).sum(-1, keepdim=True), (f * ar).sum(-1, keepdim=True), 0.2, **kw)

        ms, mn = timeit(fwd, reps=20, warm=5)
        emit("GAT", "GAT layer (128 -> 8 heads x 8) forward, attention block %s" % route, e, ms, mn, nb)

        def train():
            x.grad = al.grad = ar.grad = None
            lin.zero_grad()
            fwd().square().sum().backward()

        ms, mn = timeit(train, reps=10, warm=3)
        emit("GAT", "GAT layer (128 -> 8 heads x 8) forward + backward, attention block %s" % route, e, ms, mn, 3 * nb)


def edge_order(dev, args):
    """Position-ordered hand-off of edge tensors (dgl_amd.edge_order; VERDICT r2 Next #4) through the
    operator API on graphs that carry DGL's usual random edge-id map: edge softmax alone (plain
    edge-id-ordered scores in; forward, and forward + backward), and the GATConv forward layer, with
    the hand-off off / on; the map-free kernel time is the yardstick."""
    import dgl_amd as dgl
    import dgl_amd.function as fn
    from dgl_amd.graph_index import GraphIndex, Relation
    from dgl_amd.heterograph import DGLGraph

    h, s, i = 8, 4, 4
    sizes = [(169_343, 2_501_829, "C3")]
    if args.scale == 1:
        sizes.append((C2_NODES, C2_EDGES, "C2-size"))
    for n, e, tag in sizes:
        g = synth_csr(n, n, e, "U", device=dev, with_eids=True)
        rel = Relation(n, n, csc=(g["indptr"], g["indices"], g["eids"]), idtype=g["indptr"].dtype, device=dev)
        dg = DGLGraph(GraphIndex([n], [(0, 0)], [rel]), ["_N"], [("_N", "_E", "_N")])
        rel0 = Relation(n, n, csc=(g["indptr"], g["indices"], None), idtype=g["indptr"].dtype, device=dev)
        dg0 = DGLGraph(GraphIndex([n], [(0, 0)], [rel0]), ["_N"], [("_N", "_E", "_N")])
        x = torch.rand(e, h, 1, device=dev)
        nb_f = e * (2 * h * s + i) + (n + 1) * i
        nb_fb = e * (5 * h * s + 2 * i) + 2 * (n + 1) * i      # fwd: read + write; bwd: out, grad in, grad out
        ms0, mn0 = timeit(lambda: dgl.edge_softmax(dg0, x))
        emit(tag, "edge_softmax fwd via API, map-free graph (yardstick)", e, ms0, mn0, nb_f - e * i)
        for on in (False, True):
            dgl.set_edge_order_handoff(on)
            ms, mn = timeit(lambda: dgl.edge_softmax(dg, x))
            emit(tag, "edge_softmax fwd via API, random edge-id map, hand-off %s" % ("ON (gather in, position-ordered out)" if on else "off"),
                 e, ms, mn, nb_f, ratio_to_map_free=ms / ms0)
            xg = x.clone().requires_grad_(True)
            up = torch.rand(e, h, 1, device=dev)

            def fb():
                xg.grad = None
                (dgl.edge_softmax(dg, xg) * 2.0).backward(up)   # (* 2.0 keeps the layout: `up` plays a position-ordered gradient)

            ms, mn = timeit(fb)
            emit(tag, "edge_softmax fwd + bwd via API, random edge-id map, hand-off %s" % ("ON" if on else "off"),
                 e, ms, mn, nb_fb)
            del xg, up
        dgl.set_edge_order_handoff(False)
        # the GAT forward layer (H = 8) and one training step of it
        for d in (8, 32):
            ft = torch.randn(n, h, d, device=dev)
            el = torch.randn(n, h, 1, device=dev)
            er = torch.randn(n, h, 1, device=dev)

            def layer(graph, ft=ft, el=el, er=er):
                with graph.local_scope():
                    graph.srcdata.update({"ft": ft, "el": el})
                    graph.dstdata.update({"er": er})
                    graph.apply_edges(fn.u_add_v("el", "er", "e"))
                    sc = torch.nn.functional.leaky_relu(graph.edata.pop("e"), 0.2)
                    graph.edata["a"] = dgl.edge_softmax(graph, sc)
                    graph.update_all(fn.u_mul_e("ft", "a", "m"), fn.sum("m", "o"))
                    return graph.dstdata["o"]

            nb = e * (h * 4 * 6 + h * d * 4 + 3 * 4) + n * h * d * 4
            ms0, mn0 = timeit(lambda: layer(dg0), reps=10, warm=3)
            emit(tag, "GATConv forward H=8 D=%d via API, map-free graph (yardstick)" % d, e, ms0, mn0, nb)
            for on in (False, True):
                dgl.set_edge_order_handoff(on)
                ms, mn = timeit(lambda: layer(dg), reps=10, warm=3)
                emit(tag, "GATConv forward H=8 D=%d via API, random edge-id map, hand-off %s" % (d, "ON" if on else "off"),
                     e, ms, mn, nb, ratio_to_map_free=ms / ms0)
                ps = [t.clone().requires_grad_(True) for t in (ft, el, er)]

                def train():
                    for p in ps:
                        p.grad = None
                    layer(dg, *ps).square().sum().backward()

                ms, mn = timeit(train, reps=10, warm=3)
                emit(tag, "GATConv forward + backward H=8 D=%d via API, random edge-id map, hand-off %s" % (d, "ON" if on else "off"),
                     e, ms, mn, 3 * nb)
            dgl.set_edge_order_handoff(False)
            del ft, el, er
        del g, dg, dg0, rel, rel0, x


def fmt(dev, args):
    """COO -> CSC of the C2 graph: dgla_coo_to_csr (radix sort of (dst, position)
    + one fused gather / indptr kernel) vs the same result from torch primitives (stable argsort,
    bincount, cumsum, two gathers).  Algorithmic bytes: read row + col, write indices + eids +
    indptr (the sort's own passes are overhead on top)."""
    n, e = C2_NODES // args.scale, C2_EDGES // args.scale
    for idt in (torch.int32, torch.int64):
        g = synth_csr(n, n, e, "U", device=dev, idtype=idt)
        src = g["indices"]
        dst = torch.repeat_interleave(torch.arange(n, device=dev), (g["indptr"][1:] - g["indptr"][:-1]).long()).to(idt)
        perm = torch.randperm(e, device=dev)
        src, dst = src[perm].contiguous(), dst[perm].contiguous()   # an unsorted COO
        del g, perm
        i = 4 if idt == torch.int32 else 8
        nb = e * i * 4 + (n + 1) * i
        ms, mn = timeit(lambda: _capi.coo_to_csr(dst, src, None, n, n), reps=5, warm=2)
        emit("FMT", "COO -> CSC, %d edges, %s ids: dgla_coo_to_csr" % (e, str(idt)), e, ms, mn, nb)

        def torch_path():
            order = torch.argsort(dst, stable=True)
            counts = torch.bincount(dst.long(), minlength=n)
            indptr = torch.zeros(n + 1, dtype=idt, device=dev)
            indptr[1:] = torch.cumsum(counts, 0).to(idt)
            return indptr, src[order], order.to(idt)

        ms, mn = timeit(torch_path, reps=5, warm=2)
        emit("FMT", "COO -> CSC, %d edges, %s ids: torch argsort + bincount + cumsum + gathers" % (e, str(idt)), e, ms,
             mn, nb)
        a, b = _capi.coo_to_csr(dst, src, None, n), torch_path()
        assert all(torch.equal(x, y) for x, y in zip(a, b))
        del src, dst


def sparse_front(dev, args):
    """The dgl.sparse front end on the C2-shaped matrix (2.45 M x 2.45 M, 61.9 M nonzeros): the same kernels as the
    operator API.  A matrix made from CSR keeps its values in CSR order (no map); one made from COO indices — how
    dgl.sparse users usually make them — reads its values through the value-index permutation, like an edge-id map."""
    import dgl_amd.sparse as dglsp

    n, e, f = C2_NODES // args.scale, C2_EDGES // args.scale, C2_FEAT
    g = synth_csr(n, n, e, "U", device=dev, with_eids=True)
    torch.manual_seed(2)
    x = torch.rand(n, f, device=dev) + 1
    val = torch.rand(e, device=dev) + 1
    a_csr = dglsp.from_csr(g["indptr"].long(), g["indices"].long(), val, (n, n))
    row = torch.repeat_interleave(torch.arange(n, device=dev), (g["indptr"][1:] - g["indptr"][:-1]).long())
    perm = torch.randperm(e, device=dev)
    a_coo = dglsp.from_coo(row[perm], g["indices"].long()[perm], val[perm], (n, n))
    a_coo.csr()                                     # (built once, like a graph's formats)
    nb = spmm_bytes(n, e, f, f, 4, 8, 1)
    for name, a, extra in (("made from CSR (values in CSR order)", a_csr, 0), ("made from COO indices (value-index map)", a_coo, e * 8)):
        ms, mn = timeit(lambda: dglsp.spmm(a, x))
        emit("SP", "dgl.sparse.spmm, int64 indices, F=100, matrix " + name, e, ms, mn, nb + extra)
    x1, x2 = torch.rand(n, 16, device=dev), torch.rand(16, n, device=dev)
    ms, mn = timeit(lambda: dglsp.sddmm(a_csr, x1, x2))
    emit("SP", "dgl.sparse.sddmm (K=16), matrix made from CSR", e, ms, mn, e * (2 * 16 * 4 + 8 + 8))
    ms, mn = timeit(lambda: a_csr.softmax(1))
    emit("SP", "dgl.sparse softmax over rows, matrix made from CSR", e, ms, mn, e * (4 + 4 + 8))
    ms, mn = timeit(lambda: a_coo.softmax(1))
    emit("SP", "dgl.sparse softmax over rows, matrix made from COO indices", e, ms, mn, e * (4 + 4 + 8 + 8))
    ms, mn = timeit(lambda: a_csr.smax(1))
    emit("SP", "dgl.sparse smax along rows (g-SpMM copy_e max), matrix made from CSR", e, ms, mn, e * (4 + 8) + n * 12)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="also ops whose output is E x F (25 GB)")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--verify", action="store_true", help="check every fp32 SpMM result against the CPU oracle at full size")
    args = ap.parse_args()
    global VERIFY
    VERIFY = args.verify
    dev = torch.device("cuda:0")
    for name, fn in (("C2", c2), ("C3", c3), ("C5", c5), ("C5MAX", c5max), ("SEG", seg), ("MM", mm), ("SAMPLE", sample), ("GAT", gat_graph), ("EO", edge_order), ("FMT", fmt), ("SP", sparse_front)):
        if args.only and name not in args.only.split(","):
            continue
        fn(dev, args)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
