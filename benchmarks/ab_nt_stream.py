"""In-process interleaved A/B of DGLA_TUNE_NT_STREAM: non-temporal loads of an edge operand that is read in
position order (segment reduce, copy_e; the g-SpMM `mul` case is the control: the bit does not reach it)."""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgl_amd import _capi
from tests.graphgen import C2_EDGES, C2_NODES, synth_csr, lognormal_degrees

dev = torch.device("cuda:0")
cases = {}

def add(name, fn):
    fn(); cases[name] = fn

rows, f = C2_EDGES // 4, 100
torch.manual_seed(3)
feat = torch.rand(rows, f, device=dev)
keep = []
for label, nseg in (("612k", C2_NODES // 4), ("64", 64), ("2.4M", C2_NODES)):
    if nseg == 64:
        lens = np.full(64, rows // 64, dtype=np.int64); lens[-1] += rows - lens.sum()
    else:
        lens = lognormal_degrees(nseg, rows)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])).to(dev)
    for red in ("sum", "max"):
        out = torch.empty(nseg, f, device=dev)
        arg = torch.empty(nseg, f, dtype=torch.int64, device=dev) if red == "max" else None
        ws = torch.empty(max(1, _capi.segment_reduce_workspace_bytes(red, feat, off, out)), dtype=torch.uint8, device=dev)
        _capi.segment_reduce(red, feat, off, out, arg, ws)
        keep.append((off, out, arg, ws))
        add("segment_reduce %s -> %s segments" % (red, label),
            lambda red=red, off=off, out=out, arg=arg, ws=ws: _capi.segment_reduce(red, feat, off, out, arg, ws, plan_valid=True))

n, e = C2_NODES, C2_EDGES
g = synth_csr(n, n, e, "U", device=dev, with_eids=False)
csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
x = torch.rand(n, f, device=dev) + 1
w1 = torch.rand(e, 1, device=dev) + 1
out = torch.empty(n, f, device=dev)
ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes("mul", "sum", csr, out.dtype, x, w1, out)), dtype=torch.uint8, device=dev)
_capi.spmm_csr("mul", "sum", csr, x, w1, out, None, None, ws)
add("C2 u_mul_e_sum scalar e, no map", lambda: _capi.spmm_csr("mul", "sum", csr, x, w1, out, None, None, ws, plan_valid=True))

def timeit(fn, reps=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]))

base = _capi.get_tuning() & ~_capi.TUNE_NT_STREAM
res = {k: {0: [], 1: []} for k in cases}
for rnd in range(6):
    for on in (0, 1):
        _capi.set_tuning(base | (_capi.TUNE_NT_STREAM if on else 0))
        for k, fn in cases.items():
            res[k][on].append(timeit(fn))
for k, v in res.items():
    print(json.dumps({"case": k, "ms_default_loads": round(float(np.median(v[0])), 4), "ms_nt_stream": round(float(np.median(v[1])), 4),
                      "rounds_default": [round(t, 3) for t in v[0]], "rounds_nt": [round(t, 3) for t in v[1]]}), flush=True)
