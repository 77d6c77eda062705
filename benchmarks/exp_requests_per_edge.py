#!/usr/bin/env python
"""Requests per edge at the L2 <-> fabric interface, by layout (run under
`rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace`): launches the C2
g-SpMM copy_u + sum a few times for (dtype, variant, tuning flags) and prints the launch order so
that the counter rows of the merge kernel can be matched to it.

    python benchmarks/exp_requests_per_edge.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd import _capi  # noqa: E402
from tests.graphgen import C2_EDGES, C2_NODES, synth_csr  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, e, f = C2_NODES, C2_EDGES, 100
    default = _capi.get_tuning()
    order = []
    for variant in ("U", "L"):
        g = synth_csr(n, n, e, variant, device=dev)
        csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
        for dt in (torch.float32, torch.bfloat16):
            torch.manual_seed(1)
            x = (torch.rand(n, f, device=dev) + 1).to(dt)
            out = torch.empty(n, f, device=dev, dtype=dt)
            for flags in (default & ~8, default):
                _capi.set_tuning(flags)
                ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, dt, x, None, out),
                                 dtype=torch.uint8, device=dev)
                _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws)
                _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True)
                torch.cuda.synchronize()
                order.append({"variant": variant, "dtype": str(dt), "split_layouts": bool(flags & 8), "launches": 2})
                del ws
            del x, out
        del g
    _capi.set_tuning(default)
    print(json.dumps({"edges": e, "merge_kernel_launch_order": order}))


if __name__ == "__main__":
    main()
