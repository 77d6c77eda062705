import sys, json, torch, numpy as np
sys.path.insert(0, '.')
from dgl_amd import _capi
from tests.graphgen import synth_csr, C2_NODES, C2_EDGES
dev = torch.device("cuda:0")
n, e = C2_NODES, C2_EDGES
for variant in ("U", "L"):
    g = synth_csr(n, n, e, variant, device=dev)
    csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
    for f in (100, 52, 48, 28, 24, 16, 12):
        x = torch.rand(n, f, device=dev) + 1
        out = torch.empty(n, f, device=dev)
        ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out), dtype=torch.uint8, device=dev)
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws)
        for _ in range(3): _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record()
        for k in range(10):
            _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True); ev[k + 1].record()
        torch.cuda.synchronize()
        ms = float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(10)]))
        print(json.dumps({"variant": variant, "feature_columns": f, "ms_per_call": round(ms, 4), "G_edges_per_s": round(e / ms / 1e6, 2)}), flush=True)
        del x, out, ws
