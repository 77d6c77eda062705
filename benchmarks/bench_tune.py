#!/usr/bin/env python
"""A/B of the CSR SpMM tuning bits (dgla_set_tuning) and of the row width on the headline
graph: prints one JSON line per (variant, F, flags) with whole-step and merge-kernel times.

    python benchmarks/bench_tune.py [--scale S] [--reps R]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.graphgen import C2_EDGES, C2_NODES, synth_csr  # noqa: E402


def probe_counters(ws, num_rows, nnz):
    """{local, sampled} of the locality probe stored behind the merge plan (spmm_csr.hip.h)."""
    waves = (num_rows + nnz + 511) // 512
    off = (8 * (waves + 1) + 255) // 256 * 256
    return [int(v) for v in ws[off:off + 8].view(torch.int32).tolist()]


def run(dev, g, x, flags, reps, split_valid=False):
    from dgl_amd import _capi

    default = _capi.get_tuning()
    _capi.set_tuning(flags)
    n = g["num_rows"]
    out = torch.empty(n, x.shape[1], device=dev, dtype=x.dtype)
    csr = _capi.make_csr(g["indptr"], g["indices"], None, g["num_cols"])
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out),
                     dtype=torch.uint8, device=dev)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws)
    for _ in range(2):
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True)
    torch.cuda.synchronize()
    run.probe = probe_counters(ws, n, g["nnz"])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(reps)]
    for a, b in kev:
        a.record()
        b.record()
    torch.cuda.synchronize()
    ev[0].record()
    for k in range(reps):
        _capi.set_profile_events(*kev[k])
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True,
                       split_valid=split_valid)
        ev[k + 1].record()
    torch.cuda.synchronize()
    _capi.set_profile_events(None, None)
    _capi.set_tuning(default)
    step = [ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]
    kern = [a.elapsed_time(b) for a, b in kev]
    return float(np.median(step)), float(np.median(kern)), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--flags", default="0,1,2,4,6,8,9,10,14,15,73,105")
    ap.add_argument("--split-valid", action="store_true",
                    help="also time every split-row setting with the copy kept from the previous call")
    ap.add_argument("--variants", default="U,L")
    ap.add_argument("--feats", default="100,96,128,64,104")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, e = C2_NODES // args.scale, C2_EDGES // args.scale
    flags = [int(v) for v in args.flags.split(",")]
    for variant in args.variants.split(","):
        g = synth_csr(n, n, e, variant, device=dev)
        for f in [int(v) for v in args.feats.split(",")]:
            torch.manual_seed(12345)
            x = torch.rand(n, f, device=dev) + 1
            base = None
            runs = [(fl, False) for fl in (flags if f == 100 else [0, 15])]
            if args.split_valid and f == 100:
                runs += [(fl, True) for fl in flags if fl & 8]
            for fl, sv in runs:
                step, kern, out = run(dev, g, x, fl, args.reps, split_valid=sv)
                if base is None:
                    base = out.clone()
                same = bool(torch.equal(out, base))
                b_alg = e * (f * 4 + 4) + (n + 1) * 4 + n * f * 4
                print(json.dumps({"variant": variant, "F": f, "flags": fl, "split_valid": sv,
                                  "probe_local_sampled": run.probe, "step_ms": round(step, 4),
                                  "merge_kernel_ms": round(kern, 4),
                                  "G_edges_per_s": round(e / step / 1e6, 3),
                                  "alg_GBps_kernel": round(b_alg / kern / 1e6, 1),
                                  "bit_identical_to_flags0": same}), flush=True)
            del x, base
        del g


if __name__ == "__main__":
    main()
