"""configs[4]'s one-GPU piece under the microscope (VERDICT r4 Next #5): WHERE does the stacked 8-relation bf16
launch (10.8 ms = 0.66 of 8 TB/s in round 4) lose against the request model's 8.3 ms?

Timing A/B (one run, HIP events):
  shared    8 relations gather from ONE 5-GB table (what bench_ops measures; R-GCN's layer input is one node type)
  distinct  8 relations, 8 tables (41 GB): the address-translation hypothesis of docs/DESIGN_detail_r1_r5.md §3.2 predicts a slowdown
  single    the SAME 100 M edges as one relation through the plain merge-path kernel (no relation byte, no
            pointer table in LDS): what stacking itself costs; plus the XCD unit order toggled, and fp32 rows of the
            same BYTE length (F = 128): is it the bf16 arithmetic or the memory system?
PMC mode (--mode X --pmc): runs only the timed launch of that mode a few times, for `rocprofv3 --pmc ...`.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd import _capi  # noqa: E402
from dgl_amd.graph_index import stack_csc  # noqa: E402
from tests.graphgen import synth_csr  # noqa: E402


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="all")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r5", "stacked_tables_ab.jsonl"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, e, f, r = 10_000_000 // args.scale, 12_500_000 // args.scale, 256, 8
    torch.manual_seed(3)
    rows = []

    def emit(name, ms, mn, nbytes, **kw):
        row = dict(op=name, ms_median=round(ms, 4), ms_min=round(mn, 4), edges=r * e, alg_bytes=nbytes,
                   frac_of_8TBps=round(nbytes / (ms * 1e-3) / 8e12, 4), **kw)
        rows.append(row)
        print(json.dumps(row), flush=True)

    s, i = 2, 4
    modes = ["shared", "distinct", "single"] if args.mode == "all" else [args.mode]
    if "shared" in modes or "distinct" in modes:
        gs = [synth_csr(n, n, e, "U", seed=100 + k, device=dev, sort_cols=False) for k in range(r)]
        indptr, indices, eids, relid = stack_csc([(g["indptr"], g["indices"], None) for g in gs], n, torch.int32)
        del gs
        scsr = _capi.make_csr(indptr, indices, eids, n)
        out = torch.empty(n, f, device=dev, dtype=torch.bfloat16)
        nb1 = r * e * (f * s + i + 1) + (n + 1) * i + n * f * s
        for mode in [m for m in modes if m != "single"]:
            x0 = (torch.rand(n, f, device=dev) + 1).to(torch.bfloat16)
            xs = [x0] * r if mode == "shared" else [x0] + [x0.clone() for _ in range(r - 1)]
            sws = torch.empty(_capi.spmm_csr_stacked_workspace_bytes("copy_lhs", scsr, x0, None, out), dtype=torch.uint8,
                              device=dev)
            tabs = _capi.spmm_csr_stacked("copy_lhs", scsr, relid, xs, None, out, sws)
            run = lambda: _capi.spmm_csr_stacked("copy_lhs", scsr, relid, xs, None, out, sws, u_table=tabs[0],
                                                 plan_valid=True)
            if args.pmc:
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
            else:
                ms, mn = timeit(run)
                emit("stacked 8-relation copy_u_sum bf16 F=256, %s" % ("ONE shared 5-GB table" if mode == "shared"
                                                                       else "8 distinct tables (41 GB)"), ms, mn, nb1,
                     tables_GB=round((1 if mode == "shared" else r) * n * f * s / 1e9, 1))
            del xs, x0, sws, tabs
        del indptr, indices, eids, relid, scsr, out
    if "single" in modes:
        g = synth_csr(n, n, r * e, "U", seed=99, device=dev, sort_cols=False)
        csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
        x = (torch.rand(n, f, device=dev) + 1).to(torch.bfloat16)
        out = torch.empty(n, f, device=dev, dtype=torch.bfloat16)
        ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out), dtype=torch.uint8,
                         device=dev)
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws)
        run = lambda: _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True)
        if args.pmc:
            for _ in range(3):
                run()
            torch.cuda.synchronize()
        else:
            ms, mn = timeit(run)
            emit("ONE relation with the same 100 M edges, copy_u_sum bf16 F=256 (plain merge-path kernel)", ms, mn,
                 r * e * (f * s + i) + (n + 1) * i + n * f * s, tables_GB=round(n * f * s / 1e9, 1))
            base = _capi.get_tuning()
            _capi.set_tuning(base ^ 1)      # the XCD-contiguous unit order, the one SpMM knob that is not shape-gated
            ws_x = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out),
                               dtype=torch.uint8, device=dev)
            _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws_x)
            ms, mn = timeit(lambda: _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws_x, plan_valid=True))
            emit("ONE relation, the same 100 M edges, bf16 F=256, tuning flags %d instead of %d (XCD order toggled)"
                 % (base ^ 1, base), ms, mn, r * e * (f * s + i) + (n + 1) * i + n * f * s)
            _capi.set_tuning(base)
            del ws_x
            # the same gather volume in fp32 rows of the same BYTE length (F = 128 fp32 = 512 B): is it the bf16
            # arithmetic (unpack + fp32 accumulate + round) or the memory system?
            x32 = torch.rand(n, f // 2, device=dev) + 1
            o32 = torch.empty(n, f // 2, device=dev)
            ws2 = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x32.dtype, x32, None, o32),
                              dtype=torch.uint8, device=dev)
            _capi.spmm_csr("copy_lhs", "sum", csr, x32, None, o32, None, None, ws2)
            ms, mn = timeit(lambda: _capi.spmm_csr("copy_lhs", "sum", csr, x32, None, o32, None, None, ws2, plan_valid=True))
            emit("ONE relation, the same 100 M edges, copy_u_sum fp32 F=128 (the same 512-byte rows)", ms, mn,
                 r * e * (f * s + i) + (n + 1) * i + n * f * s, tables_GB=round(n * f * s / 1e9, 1))
    if not args.pmc:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "a") as fh:
            for row in rows:
                fh.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
