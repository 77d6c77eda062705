"""PMC / trace driver: fp32 segment_mm weight gradient at the R-GCN shape (10 M x 256 x 256, 8 relations)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgl_amd import _capi  # noqa: E402

dev = torch.device("cuda:0")
rows, r, k, n = 10_000_000, 8, 256, 256
seglen = torch.full((r,), rows // r, dtype=torch.int64, device=dev)
a = (torch.rand(rows, k, device=dev) - 0.5)
c = (torch.rand(rows, n, device=dev) - 0.5)
b = torch.empty(r, k, n, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    _capi.segment_mm_backward_b(a, c, b, seglen)
torch.cuda.synchronize()
print(_capi.segment_mm_backward_b_last_route())
