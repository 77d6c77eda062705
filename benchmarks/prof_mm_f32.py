"""PMC driver: fp32 segment_mm forward (LDS-direct kernel) and the vendor GEMM at the R-GCN shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgl_amd import _capi  # noqa: E402

dev = torch.device("cuda:0")
rows, r, k, n = 10_000_000, 8, 256, 256
seglen = torch.full((r,), rows // r, dtype=torch.int64, device=dev)
a = (torch.rand(rows, k, device=dev) - 0.5)
b = (torch.rand(r, k, n, device=dev) - 0.5)
c = torch.empty(rows, n, device=dev)
for _ in range(3):
    _capi.segment_mm(a, b, c, seglen)
    for i in range(r):
        torch.mm(a[i * (rows // r):(i + 1) * (rows // r)], b[i], out=c[i * (rows // r):(i + 1) * (rows // r)])
torch.cuda.synchronize()
