"""edge_softmax forward / backward and u_add_v SDDMM at growing edge counts (H = 8): does the
fraction of the HBM peak recover once the launch is large enough?"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd import _capi  # noqa: E402
from tests.graphgen import synth_csr  # noqa: E402

dev = torch.device("cuda:0")


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
    return float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]))


h, s, i = 8, 4, 4
SIZES = ((169_343, 2_501_829), (600_000, 15_000_000), (2_449_029, 61_859_140))
if len(sys.argv) > 1:  # e.g. "2" = only the largest; "2 nomap" = only without an edge-id map
    SIZES = (SIZES[int(sys.argv[1])],)
MAPS = (False,) if "nomap" in sys.argv else ((True,) if "map" in sys.argv else (True, False))
for n, e in SIZES:
    for with_eids in MAPS:
        g = synth_csr(n, n, e, "U", device=dev, with_eids=with_eids)
        csr = _capi.make_csr(g["indptr"], g["indices"], g["eids"] if with_eids else None, n)
        x = torch.rand(e, h, 1, device=dev)
        a = torch.empty_like(x)
        ws = torch.empty(_capi.edge_softmax_workspace_bytes(csr, x.dtype, h), dtype=torch.uint8, device=dev)
        _capi.edge_softmax_forward(csr, x, a, ws)
        ms = timeit(lambda: _capi.edge_softmax_forward(csr, x, a, ws, plan_valid=True))
        nb = e * (2 * h * s + (i if with_eids else 0)) + (n + 1) * i
        print(json.dumps({"op": "edge_softmax fwd", "edges": e, "eid_map": with_eids, "ms": round(ms, 4),
                          "GBps": round(nb / ms / 1e6, 1), "frac": round(nb / ms / 1e6 / 8000, 3)}), flush=True)
        sds = a * x
        back = torch.empty_like(x)
        ms = timeit(lambda: _capi.edge_softmax_backward(csr, a, sds, back, ws, plan_valid=True))
        nb = e * (3 * h * s + (i if with_eids else 0)) + (n + 1) * i
        print(json.dumps({"op": "edge_softmax bwd", "edges": e, "eid_map": with_eids, "ms": round(ms, 4),
                          "GBps": round(nb / ms / 1e6, 1), "frac": round(nb / ms / 1e6 / 8000, 3)}), flush=True)
        del x, a, sds, back, ws
