import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from dgl_amd import _capi
from tests.test_gpu_softmax_kernels import _graph
dev = torch.device("cuda:0")
for kind, dim in (("tiny", 1), ("ones", 1), ("hub", 8), ("mix", 4)):
    rng = np.random.default_rng(hash((kind, dim)) % 2 ** 31)
    indptr = _graph(kind, rng)
    e, n = int(indptr[-1]), indptr.size - 1
    score = (rng.standard_normal((e, dim)) * 3).astype(np.float32)
    ip = torch.from_numpy(indptr.astype(np.int32)).to(dev)
    csr = _capi.make_csr(ip, torch.zeros(e, dtype=torch.int32, device=dev), None, n)
    x = torch.from_numpy(score).to(dev)
    ws = torch.empty(_capi.edge_softmax_workspace_bytes(csr, x.dtype, dim), dtype=torch.uint8, device=dev)
    out = torch.full_like(x, float("nan"))
    _capi.edge_softmax_forward(csr, x, out, ws)
    ref = oracle.edge_softmax_fwd(indptr.astype(np.int32), None, score.astype(np.float64))
    got = out.cpu().numpy()
    bad = np.argwhere(~(np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-7))
    print(kind, dim, "edges", e, "bad", len(bad))
    rows = np.repeat(np.arange(n), np.diff(indptr))
    for (ei, h) in bad[:12]:
        r = rows[ei]
        d = (r + ei)  # merge position of the edge
        print("  edge", ei, "h", h, "row", r, "deg", indptr[r + 1] - indptr[r], "pos in row", ei - indptr[r],
              "unit", d // 256, "merge pos in unit", d % 256, "got", got[ei, h], "ref", ref[ei, h])
