import sys, json, torch
sys.path.insert(0, "/root/repo")
from dgl_amd import _capi
from tests.graphgen import synth_csr
from benchmarks.bench_ops import timeit
dev = torch.device("cuda:0")
# one relation with ALL 100 M edges, N = 10 M, F = 256 bf16: the non-stacked kernel on the C5 volume
for n, e in ((10_000_000, 100_000_000), (2_500_000, 100_000_000), (10_000_000, 25_000_000)):
    g = synth_csr(n, n, e, "U", device=dev, idtype=torch.int32)
    x = torch.rand(n, 256, device=dev).to(torch.bfloat16)
    out = torch.empty(n, 256, device=dev, dtype=torch.bfloat16)
    csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out), dtype=torch.uint8, device=dev)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws)
    ms, mn = timeit(lambda: _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True), reps=5)
    lines = e * 4
    print(json.dumps({"N": n, "E": e, "ms": round(ms, 3), "G_lines_per_s": round(lines / ms / 1e6, 1),
                      "G_edges_per_s": round(e / ms / 1e6, 2), "X_GB": n * 512 / 1e9}), flush=True)
    del g, x, out, ws
