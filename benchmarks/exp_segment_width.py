"""segment_reduce sum over sequential rows at several row widths: is the merge kernel bound by
lanes per row (25 of 32 lanes busy at F = 100) or by the memory system?"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd import _capi  # noqa: E402
from tests.graphgen import lognormal_degrees  # noqa: E402

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


total_bytes = 6_000_000_000
for dt, es in ((torch.float32, 4), (torch.bfloat16, 2)):
    for f in (32, 64, 100, 128, 256, 512):
        rows = total_bytes // (f * es)
        nseg = rows // 25
        feat = torch.rand(rows, f, device=dev).to(dt)
        lens = lognormal_degrees(nseg, rows)
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])).to(dev)
        out = torch.empty(nseg, f, device=dev, dtype=dt)
        ws = torch.empty(max(1, _capi.segment_reduce_workspace_bytes("sum", feat, off, out)), dtype=torch.uint8, device=dev)
        _capi.segment_reduce("sum", feat, off, out, None, ws)
        ms = timeit(lambda: _capi.segment_reduce("sum", feat, off, out, None, ws, plan_valid=True))
        nb = rows * f * es + nseg * f * es + (nseg + 1) * 8
        print(json.dumps({"dtype": str(dt), "F": f, "rows": rows, "ms": round(ms, 4), "GBps": round(nb / ms / 1e6, 1),
                          "rows_per_s_G": round(rows / ms / 1e6, 2)}), flush=True)
        del feat, out, ws, off
