#!/usr/bin/env python
"""Measured HBM peak on this box: sweep of the stream-copy kernel variants
(dgla_stream_copy_variant) + torch's own D2D copy, 4 GiB buffers.  One JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd import _capi  # noqa: E402


def t_ms(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        fn()
        ev[k + 1].record()
    torch.cuda.synchronize()
    return min(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))


def main():
    dev = torch.device("cuda:0")
    nbytes = 4 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    src.random_(0, 255)
    res = {}
    for mode, mname in ((0, "copy_nt"), (1, "copy_plain"), (2, "read_only")):
        for bsel in range(4):
            for usel in range(2):
                v = mode | (bsel << 2) | (usel << 4)
                ms = t_ms(lambda: _capi.stream_copy_variant(dst, src, v))
                moved = nbytes if mode == 2 else 2 * nbytes
                res["%s_bpc%d_u%d" % (mname, 4 << bsel, 8 if usel else 4)] = round(moved / ms / 1e6, 1)
        for usel in range(2):  # one tile per workgroup (no grid-stride loop)
            v = mode | (usel << 4) | 32
            ms = t_ms(lambda: _capi.stream_copy_variant(dst, src, v))
            moved = nbytes if mode == 2 else 2 * nbytes
            res["%s_onetile_u%d" % (mname, 8 if usel else 4)] = round(moved / ms / 1e6, 1)
    ms = t_ms(lambda: dst.copy_(src))
    res["torch_copy_"] = round(2 * nbytes / ms / 1e6, 1)
    print(json.dumps({"unit": "GB/s", "bytes": nbytes, "results": res,
                      "best_copy": max(v for k, v in res.items() if k.startswith(("copy", "torch"))),
                      "best_read": max(v for k, v in res.items() if k.startswith("read"))}))


if __name__ == "__main__":
    main()
