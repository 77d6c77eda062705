"""The rest of the hot path on bench.py's clock (VERDICT r5 Next #1a): the operators `north_star` names besides
copy_u+sum, each timed with the same HIP-event protocol as the headline's variants (median of `reps` calls after
warm-up, events on the launch stream), with its ALGORITHMIC bytes per SURVEY.md §8(d) and the fraction of the 8 TB/s peak.

  u_mul_e_sum        C2 (N = 2.45 M, E = 61.9 M, F = 100, scalar e, fp32), map-free CSC and behind a random edge-id map
                     (the default for every graph built from COO), default path (nothing announced static)
  sddmm_u_dot_v      C3 (H = 8, D = 32) and C2-size (D = 100)                       [benchmarks/kernel/bench_gsddmm_u_dot_v.py]
  edge_softmax       fwd and fwd + bwd, H = 8, C3 and C2-size, map-free and mapped  [bench_edgesoftmax.py]
  gat_attention      configs[2]'s block (u_add_v -> leaky_relu -> edge_softmax -> u_mul_e_sum), fwd and fwd + bwd at C3 and
                     C2-size (fwd), mapped graph: the composed operators and the default route (the fused kernel)
  rgcn_stacked_bf16  configs[4]'s one-GPU piece: 8 relations x 12.5 M edges, N = 10 M, F = 256, bf16, ONE stacked launch
  coo_to_csc, copy_u_max fwd + bwd, segment_mm fwd / dB (bf16, fp32)   the §8 "next" rows f2, f3 and the max / min gradient

Byte models: SpMM  E*(F_l*s + w*s + i [+ i map]) + (N+1)*i + N*F_out*s;  SDDMM dot  E*(2*H*D*s + 2*i) + E*H*s;
softmax fwd  E*(2*H*s [+ i map]) + (N+1)*i, fwd + bwd  E*(5*H*s [+ 2*i map]) + 2*(N+1)*i;
GAT fwd  E*(H*D*s + H*s + i) + N*(H*D*s + 2*H*s) + (N+1)*i  (ft row + el gathered per edge; er, out per node — the
compulsory gather of the fused form; the composed form moves 6 more (E, H) tensors), fwd + bwd 3x that.
An operand that fits the 256 MiB Infinity Cache is marked `"hbm_roofline_evidence": false` (the gather model then says
nothing about HBM).  Everything here is builder-side plumbing around the C ABI / operator API — no oracle, no CPU path."""
import numpy as np
import torch

PEAK = 8000.0
CACHE_BYTES = 256 << 20
C3_NODES, C3_EDGES = 169_343, 2_501_829      # ogbn-arxiv + reverse edges + self loops


def _time(fn, reps=10, warm=3):
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


def _line(ms_mn, nbytes, edges, gathered=None, **kw):
    ms, mn = ms_mn
    d = {"ms": ms, "ms_min": mn, "edges": int(edges), "edges_per_s": edges / (ms * 1e-3), "alg_bytes": int(nbytes),
         "achieved_GBps": nbytes / (ms * 1e-3) / 1e9, "roofline_frac": nbytes / (ms * 1e-3) / 1e9 / PEAK}
    if gathered is not None:
        d["hbm_roofline_evidence"] = bool(gathered > CACHE_BYTES)
    d.update(kw)
    return d


def _dgl_graph(g, with_map, dev):
    from dgl_amd.graph_index import GraphIndex, Relation
    from dgl_amd.heterograph import DGLGraph

    n = g["num_rows"]
    rel = Relation(n, g["num_cols"], csc=(g["indptr"], g["indices"], g["eids"] if with_map else None),
                   idtype=g["indptr"].dtype, device=dev)
    return DGLGraph(GraphIndex([n], [(0, 0)], [rel]), ["_N"], [("_N", "_E", "_N")])


def op_variants(dev, c2_graph=None, scale=1, reps=10):
    """`c2_graph`: bench.py's own C2 CSR (variant U, int32, no edge ids) — re-used, a random edge-id map is added here."""
    import dgl_amd as dgl
    from dgl_amd import _capi
    from dgl_amd.graph_index import stack_csc
    from tests.graphgen import C2_EDGES, C2_FEAT, C2_NODES, synth_csr

    res = {}
    s, i, h = 4, 4, 8
    # ---------------- C2-size -----------------------------------------------------------------------------------
    n, e, f = C2_NODES // scale, C2_EDGES // scale, C2_FEAT
    g = dict(c2_graph) if c2_graph is not None else synth_csr(n, n, e, "U", seed=20250824, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    g["eids"] = torch.randperm(e, device=dev, generator=gen).to(torch.int32)
    torch.manual_seed(1)
    x = torch.rand(n, f, device=dev) + 1
    w1 = torch.rand(e, 1, device=dev) + 1
    out = torch.empty(n, f, device=dev)
    for tag, eid in (("map_free", False), ("eid_map", True)):
        csr = _capi.make_csr(g["indptr"], g["indices"], g["eids"] if eid else None, n)
        ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes("mul", "sum", csr, x.dtype, x, w1, out)), dtype=torch.uint8, device=dev)
        _capi.spmm_csr("mul", "sum", csr, x, w1, out, None, None, ws)
        t = _time(lambda: _capi.spmm_csr("mul", "sum", csr, x, w1, out, None, None, ws, plan_valid=True), reps)
        nb = e * (f * s + s + i + (i if eid else 0)) + (n + 1) * i + n * f * s
        res["u_mul_e_sum_C2_" + tag] = _line(t, nb, e, gathered=n * f * s, op="dgla_spmm_csr mul/sum, scalar e, fp32, int32 ids, default path")
        del csr, ws
    # SDDMM u_dot_v over the C2 graph as COO (the reference's preferred SDDMM format), D = 100
    deg = (g["indptr"][1:] - g["indptr"][:-1]).long()
    dst = torch.repeat_interleave(torch.arange(n, device=dev), deg).to(torch.int32)
    coo = _capi.make_coo(g["indices"], dst, None, n, n)
    oe = torch.empty(e, 1, device=dev)
    t = _time(lambda: _capi.sddmm_coo("dot", coo, x, x, oe, 0, 2), max(3, reps // 2))
    res["sddmm_u_dot_v_C2size_D100"] = _line(t, e * (2 * f * s + 2 * i) + e * s, e, gathered=n * f * s, op="dgla_sddmm_coo dot(u, v)")
    del coo, oe
    # COO -> CSC of the same graph given as an UNSORTED COO (reference: COOSort + cusparseXcoo2csr)
    perm = torch.randperm(e, device=dev, generator=gen)
    rsrc, rdst = g["indices"][perm].contiguous(), dst[perm].contiguous()
    del perm
    t = _time(lambda: _capi.coo_to_csr(rdst, rsrc, None, n, n), max(3, reps // 2), warm=2)
    res["coo_to_csc_C2_int32"] = _line(t, e * i * 4 + (n + 1) * i, e, op="dgla_coo_to_csr, 61.9 M unsorted edges -> indptr, indices, edge ids (own MSD sort)")
    del rsrc, rdst, dst, deg
    # copy_u + max, forward + backward through the operator API (autograd): forward with winners, backward = winner bits +
    # gated masked g-SpMM over the reverse graph (no atomics).  Bytes: forward E*(F*s + i) + (N+1)*i + 2*N*F*s (out, arg_u),
    # backward N*F*(s + i) (dZ, arg_u) + E*i (column ids) + E*W*s (bit words written) + E*(F*s + W*s + i) + (N+1)*i + N*F*s
    dgm = _dgl_graph(g, False, dev)
    xg = x.clone().requires_grad_(True)
    up = torch.rand(n, f, device=dev)

    def fb_max():
        xg.grad = None
        dgl.ops.copy_u_max(dgm, xg).backward(up)

    fb_max()   # (builds the out-edge CSR and the position map once, like the forward CSC)
    t = _time(fb_max, max(3, reps // 2), warm=2)
    words = (f + 31) // 32
    nb = (e * (f * s + i) + (n + 1) * i + 2 * n * f * s) + n * f * (s + i) + e * i + e * words * s + \
         e * (f * s + words * s + i) + (n + 1) * i + n * f * s
    res["copy_u_max_fwd_bwd_C2"] = _line(t, nb, e, gathered=n * f * s, op="dgl.ops.copy_u_max(...).backward(), fp32, int32 ids")
    del dgm, xg, up, x, w1, out
    torch.cuda.empty_cache()
    # edge softmax through the operator API (what a caller pays: allocation + FFI + kernels), H = 8
    for tag, with_map in (("map_free", False), ("eid_map", True)):
        dg = _dgl_graph(g, with_map, dev)
        sc = torch.rand(e, h, 1, device=dev)
        t = _time(lambda: dgl.edge_softmax(dg, sc), reps)
        res["edge_softmax_fwd_C2size_" + tag] = _line(t, e * (2 * h * s + (i if with_map else 0)) + (n + 1) * i, e, gathered=e * h * s)
        sg = sc.clone().requires_grad_(True)
        up = torch.rand(e, h, 1, device=dev)

        def fb():
            sg.grad = None
            dgl.edge_softmax(dg, sg).backward(up)

        t = _time(fb, max(3, reps // 2))
        res["edge_softmax_fwd_bwd_C2size_" + tag] = _line(t, e * (5 * h * s + (2 * i if with_map else 0)) + 2 * (n + 1) * i, e, gathered=e * h * s)
        del sc, sg, up
        if with_map:   # configs[2]'s block at C2 size, forward (VERDICT r5 Next #3: <= 3.5 ms at D = 8 with or without a map)
            for d in (8, 32):
                ft, el, er = (torch.randn(n, h, d, device=dev), torch.randn(n, h, 1, device=dev), torch.randn(n, h, 1, device=dev))
                nb = e * (h * d * s + h * s + i) + n * (h * d * s + 2 * h * s) + (n + 1) * i
                for route, kw in (("composed", dict(fused=False)), ("default", {})):
                    with torch.no_grad():
                        t = _time(lambda: dgl.nn.gat_attention(dg, ft, el, er, 0.2, **kw), max(3, reps // 2))
                    res["gat_attention_fwd_C2size_H8_D%d_eid_map_%s" % (d, route)] = _line(t, nb, e, gathered=n * h * d * s)
                del ft, el, er
        del dg
    del g
    torch.cuda.empty_cache()
    # ---------------- C3 ----------------------------------------------------------------------------------------
    n, e = C3_NODES, C3_EDGES
    g = synth_csr(n, n, e, "U", seed=3, device=dev, with_eids=True)
    deg = (g["indptr"][1:] - g["indptr"][:-1]).long()
    dst = torch.repeat_interleave(torch.arange(n, device=dev), deg).to(torch.int32)
    coo = _capi.make_coo(g["indices"], dst, g["eids"], n, n)
    d = 32
    torch.manual_seed(2)
    ft = torch.rand(n, h, d, device=dev)
    oe = torch.empty(e, h, 1, device=dev)
    t = _time(lambda: _capi.sddmm_coo("dot", coo, ft, ft, oe, 0, 2), 2 * reps)
    res["sddmm_u_dot_v_C3_H8_D32"] = _line(t, e * (2 * h * d * s + 2 * i) + e * h * s, e, gathered=n * h * d * s, op="dgla_sddmm_coo dot(u, v)")
    del coo, oe, dst, deg
    for tag, with_map in (("map_free", False), ("eid_map", True)):
        dg = _dgl_graph(g, with_map, dev)
        sc = torch.rand(e, h, 1, device=dev)
        t = _time(lambda: dgl.edge_softmax(dg, sc), 2 * reps)
        res["edge_softmax_fwd_C3_" + tag] = _line(t, e * (2 * h * s + (i if with_map else 0)) + (n + 1) * i, e, gathered=e * h * s)
        sg = sc.clone().requires_grad_(True)
        up = torch.rand(e, h, 1, device=dev)

        def fb3():
            sg.grad = None
            dgl.edge_softmax(dg, sg).backward(up)

        t = _time(fb3, 2 * reps)
        res["edge_softmax_fwd_bwd_C3_" + tag] = _line(t, e * (5 * h * s + (2 * i if with_map else 0)) + 2 * (n + 1) * i, e, gathered=e * h * s)
        del sc, sg, up
        if with_map:
            for d in (8, 32):
                ps = [torch.randn(n, h, d, device=dev, requires_grad=True), torch.randn(n, h, 1, device=dev, requires_grad=True),
                      torch.randn(n, h, 1, device=dev, requires_grad=True)]
                up = torch.randn(n, h, d, device=dev)
                nb = e * (h * d * s + h * s + i) + n * (h * d * s + 2 * h * s) + (n + 1) * i
                for route, kw in (("composed", dict(fused=False)), ("default", {})):
                    with torch.no_grad():
                        t = _time(lambda: dgl.nn.gat_attention(dg, ps[0], ps[1], ps[2], 0.2, **kw), 2 * reps)
                    res["gat_attention_fwd_C3_H8_D%d_eid_map_%s" % (d, route)] = _line(t, nb, e, gathered=n * h * d * s)

                    def train():
                        for p in ps:
                            p.grad = None
                        dgl.nn.gat_attention(dg, ps[0], ps[1], ps[2], 0.2, **kw).backward(up)

                    t = _time(train, 2 * reps)
                    res["gat_attention_fwd_bwd_C3_H8_D%d_eid_map_%s" % (d, route)] = _line(t, 3 * nb, e, gathered=n * h * d * s)
                del ps, up
        del dg
    del g, ft
    torch.cuda.empty_cache()
    # ---------------- configs[4], the one-GPU piece --------------------------------------------------------------
    n, e, f, r = 10_000_000 // scale, 12_500_000 // scale, 256, 8
    torch.manual_seed(3)
    x = (torch.rand(n, f, device=dev) + 1).to(torch.bfloat16)
    out = torch.zeros(n, f, device=dev, dtype=torch.bfloat16)
    rels = []
    for k in range(r):
        gk = synth_csr(n, n, e, "U", seed=100 + k, device=dev)
        rels.append((gk["indptr"], gk["indices"], None))
    indptr, indices, eids, relid = stack_csc(rels, n, torch.int32)
    del rels
    scsr = _capi.make_csr(indptr, indices, eids, n)
    xs = [x] * r
    sws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", scsr, x, None, out), dtype=torch.uint8, device=dev)
    tabs = _capi.spmm_csr_stacked("copy_lhs", scsr, relid, xs, None, out, sws)
    t = _time(lambda: _capi.spmm_csr_stacked("copy_lhs", scsr, relid, xs, None, out, sws, u_table=tabs[0], plan_valid=True),
              max(3, reps // 2), warm=2)
    res["rgcn_stacked_bf16_C5_one_gpu"] = _line(t, r * e * (f * 2 + i + 1) + (n + 1) * i + n * f * 2, r * e, gathered=n * f * 2,
                                               op="dgla_spmm_csr_stacked copy_lhs/sum, 8 relations x %d edges, N = %d, F = 256, bf16" % (e, n))
    del scsr, sws, tabs, xs, x, out, indptr, indices, eids, relid
    torch.cuda.empty_cache()
    # ---------------- configs[4]'s per-relation transform: segment_mm -------------------------------
    # 10 M rows x 256 x 256, 8 relations; bytes: A + C once, weights once (weight gradient: A + dC once, dB once)
    rows, k, m_, r = 10_000_000 // scale, 256, 256, 8
    seglen = torch.full((r,), rows // r, dtype=torch.int64, device=dev)
    seglen[-1] += rows - int(seglen.sum())
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
        torch.manual_seed(4)
        a = (torch.rand(rows, k, device=dev) - 0.5).to(dt)
        b = (torch.rand(r, k, m_, device=dev) - 0.5).to(dt)
        c = torch.empty(rows, m_, device=dev, dtype=dt)
        es = a.element_size()
        nb = rows * (k + m_) * es + r * k * m_ * es
        t = _time(lambda: _capi.segment_mm(a, b, c, seglen), max(3, reps // 2), warm=2)
        res["segment_mm_fwd_%s" % tag] = _line(t, nb, rows, tflops=2.0 * rows * k * m_ / (t[0] * 1e-3) / 1e12,
                                               op="dgla_segment_mm, MFMA" + (", fp32 as two scaled fp16 terms" if dt == torch.float32 else ""))
        t = _time(lambda: _capi.segment_mm_backward_b(a, c, b, seglen), max(3, reps // 2), warm=2)
        extra = {}
        if dt == torch.float32:
            fell_back, listed = _capi.segment_mm_backward_b_last_route()
            extra = {"redone_by_three_term_kernel": bool(fell_back), "listed_elements": listed}
        res["segment_mm_dB_%s" % tag] = _line(t, nb, rows, tflops=2.0 * rows * k * m_ / (t[0] * 1e-3) / 1e12,
                                              op="dgla_segment_mm_backward_b" + (", fp32 as two scaled fp16 terms" if dt == torch.float32 else ""), **extra)
        del a, b, c
    return res
