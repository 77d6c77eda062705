"""Tiny driver for PMC passes over the grouped MFMA GEMM (bf16 forward + weight gradient at the
R-GCN shape): run under rocprofv3 --pmc ... (see profiles/r1/README.md)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgl_amd import _capi  # noqa: E402

dev = torch.device("cuda:0")
rows, r, k, n = 10_000_000, 8, 256, 256
seglen = torch.full((r,), rows // r, dtype=torch.int64, device=dev)
a = (torch.rand(rows, k, device=dev) - 0.5).to(torch.bfloat16)
b = (torch.rand(r, k, n, device=dev) - 0.5).to(torch.bfloat16)
c = torch.empty(rows, n, device=dev, dtype=torch.bfloat16)
db = torch.empty_like(b)
for _ in range(3):
    _capi.segment_mm(a, b, c, seglen)
    _capi.segment_mm_backward_b(a, c, db, seglen)
torch.cuda.synchronize()
