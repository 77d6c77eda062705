"""segment_mm forward, 10 M rows x K = 256, 8 relations: the weights-stationary kernel (DGLA_MM_WS=1) against
the tiled kernel it replaces (DGLA_MM_WS=0), bf16 and fp32 (3 x bf16 split), N = 256 and N = 64.

  python benchmarks/bench_mm_ws.py            -> one JSON line per (dtype, N, kernel): median / min ms, TB/s of the
                                                 algorithmic bytes (A read once + C written once)
  python benchmarks/bench_mm_ws.py --profile  -> three launches of each weights-stationary case and nothing else
                                                 (the command tools/profile_mm_ws.sh wraps in rocprofv3)
Reference structure this replaces: src/array/cuda/gather_mm.cu:201-291 (a host loop of cuBLAS GEMMs).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgl_amd import _capi  # noqa: E402

ROWS, REL, K = 10_000_000, 8, 256


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    seglen = torch.full((REL,), ROWS // REL, dtype=torch.int64, device=dev)
    for dt in (torch.bfloat16, torch.float32):
        a = (torch.rand(ROWS, K, device=dev) - 0.5).to(dt)
        for n in (256, 64):
            b = (torch.rand(REL, K, n, device=dev) - 0.5).to(dt)
            c = torch.empty(ROWS, n, device=dev, dtype=dt)
            nbytes = (ROWS * K + ROWS * n) * a.element_size()
            for mode in (("1",) if args.profile else ("0", "1")):
                os.environ["DGLA_MM_WS"] = mode
                if args.profile:
                    for _ in range(3):
                        _capi.segment_mm(a, b, c, seglen)
                    torch.cuda.synchronize()
                    continue
                med, mn = timeit(lambda: _capi.segment_mm(a, b, c, seglen))
                print(json.dumps({"what": "segment_mm fwd %d x %d x %d, %d relations" % (ROWS, K, n, REL), "dtype": str(dt),
                                  "kernel": "weights-stationary" if mode == "1" else "tiled (round 3)", "ms_median": round(med, 4),
                                  "ms_min": round(mn, 4), "algorithmic_GB": nbytes / 1e9, "TBps_at_min": round(nbytes / mn / 1e9, 3),
                                  "frac_of_8TBps": round(nbytes / mn / 1e9 / 8, 3),
                                  "useful_TFLOPs_at_min": round(2.0 * ROWS * K * n / mn / 1e9, 1)}), flush=True)
            del b, c
        del a
    os.environ.pop("DGLA_MM_WS", None)


if __name__ == "__main__":
    main()
