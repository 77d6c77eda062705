"""The atomic max / min backward (dgla_spmm_cmp_backward) at shrinking output footprints: is it HBM or the L2 atomic units?
(profiles/r4/cmp_backward_atomic_rate.jsonl)"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dgl_amd import _capi as _c
dev = torch.device("cuda:0")
n = 2449029
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
for f in (100, 50, 25, 12, 4):
    up = torch.rand(n, f, device=dev)
    # winners: random among 25 "neighbours" of each row (locality like a real graph's: neighbours random over all rows)
    nb = torch.randint(0, n, (n, 25), device=dev, dtype=torch.int32)
    pick = torch.randint(0, 25, (n, f), device=dev)
    argu = torch.gather(nb, 1, pick).contiguous()
    dx = torch.zeros(n, f, device=dev)
    ms = timeit(lambda: _c.spmm_cmp_backward(up, argu, dx, atomic=True))
    print(json.dumps({"F": f, "dX_MB": n * f * 4 / 1e6, "ms": ms, "ms_scaled_to_F100": ms * 100 / f, "G_atomics_per_s": n * f / ms / 1e6}), flush=True)
