"""copy_e / copy_u reductions with NARROW features (F = 1 ... 32 fp32 columns) on the C2 graph: where the merge kernel stops
being memory-bound.  `DGLA_NARROW_REDUCE=0 python benchmarks/exp_narrow_features.py` times the merge kernel on every width,
without the variable widths <= 8 go to csrc/narrow_reduce.hip (profiles/r5/narrow_feature_reductions_{before,after}.jsonl)."""
import os, sys, torch, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "benchmarks"))
from bench_ops import synth_csr, timeit, C2_NODES, C2_EDGES
from dgl_amd import _capi
dev = torch.device("cuda:0")
n, e = C2_NODES, C2_EDGES
g = synth_csr(n, n, e, "U", device=dev, with_eids=True)
csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
for f in (1, 2, 4, 8, 16, 32):
    w = torch.rand(e, f, device=dev)
    for red in ("sum", "max"):
        out = torch.empty(n, f, device=dev)
        ae = torch.empty(out.shape, dtype=g["indptr"].dtype, device=dev) if red != "sum" else None
        ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes("copy_rhs", red, csr, out.dtype, None, w, out)), dtype=torch.uint8, device=dev)
        _capi.spmm_csr("copy_rhs", red, csr, None, w, out, None, ae, ws)
        ms, mn = timeit(lambda: _capi.spmm_csr("copy_rhs", red, csr, None, w, out, None, ae, ws, plan_valid=True))
        nb = e * f * 4 + n * f * 4 + (n * f * 4 if red != "sum" else 0) + n * 4
        print(json.dumps({"narrow_calls": _capi.narrow_reduce_calls(), "op": "copy_e_%s F=%d" % (red, f), "ms": round(ms, 4), "GBps": round(nb / ms / 1e6, 1), "frac": round(nb / ms / 1e6 / 8000, 3)}), flush=True)
    x = torch.rand(n, f, device=dev)
    out = torch.empty(n, f, device=dev)
    ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, out.dtype, x, None, out)), dtype=torch.uint8, device=dev)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws)
    ms, mn = timeit(lambda: _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True))
    print(json.dumps({"op": "copy_u_sum F=%d" % f, "ms": round(ms, 4)}), flush=True)
    w1 = torch.rand(e, 1, device=dev)      # a scalar weight per edge against F columns (APPNP / SGC-like propagation)
    ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes("mul", "sum", csr, out.dtype, x, w1, out)), dtype=torch.uint8, device=dev)
    _capi.spmm_csr("mul", "sum", csr, x, w1, out, None, None, ws)
    ms, mn = timeit(lambda: _capi.spmm_csr("mul", "sum", csr, x, w1, out, None, None, ws, plan_valid=True))
    print(json.dumps({"op": "u_mul_e_sum F=%d, scalar e" % f, "ms": round(ms, 4)}), flush=True)
