"""fp32 segment_mm forward, weights-stationary path: the two-scaled-fp16-term kernel (default, round 5) against the
three-bf16-term kernel (DGLA_TUNE_MM_X3) in ONE process on one box, at the R-GCN shape (10 M rows, 8 relations) and
narrower ones.  Appends to gpurun_out/r5/segment_mm_h2_vs_x3.jsonl."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd import _capi  # noqa: E402


def timeit(fn, reps=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    dev = torch.device("cuda:0")
    rows, r = 10_000_000, 8
    base = _capi.get_tuning()
    out = []
    for k, n, dist in ((256, 256, "uniform"), (256, 256, "normal"), (128, 128, "uniform"), (256, 64, "uniform"), (64, 256, "uniform"),
                       (200, 72, "uniform")):
        sl = torch.full((r,), rows // r, dtype=torch.int64, device=dev)
        torch.manual_seed(0)
        a = (torch.rand(rows, k, device=dev) - 0.5) if dist == "uniform" else torch.randn(rows, k, device=dev)
        b = torch.rand(r, k, n, device=dev) - 0.5
        c = torch.empty(rows, n, device=dev)
        nb = rows * (k + n) * 4 + r * k * n * 4
        res = {}
        for label, tune in (("H2", base), ("X3", base | _capi.TUNE_MM_X3)):
            _capi.set_tuning(tune)
            ms, mn = timeit(lambda: _capi.segment_mm(a, b, c, sl))
            res[label] = (ms, mn, c.clone() if k * n <= 128 * 128 else None)
            row = {"op": "segment_mm fwd fp32 %d x %d x %d, %d relations, A ~ %s, %s" % (rows, k, n, r, dist, label),
                   "ms_median": round(ms, 3), "ms_min": round(mn, 3), "frac_of_8TBps": round(nb / (ms * 1e-3) / 8e12, 4),
                   "tflops": round(2.0 * rows * k * n / (ms * 1e-3) / 1e12, 1)}
            print(json.dumps(row), flush=True)
            out.append(row)
        _capi.set_tuning(base)
        del a, b, c
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r5", "segment_mm_h2_vs_x3.jsonl"), "a") as fh:
        for row in out:
            fh.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
