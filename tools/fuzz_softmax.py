"""Randomised differential run of the merge-path edge softmax (forward + backward) against the
oracle: random degree mixes (empty rows, unit rows, rows around the slack and the unit capacity,
hubs), widths 1..16, with / without an edge-id map, int32 / int64 ids."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import oracle  # noqa: E402
from dgl_amd import _capi  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    n = int(rng.integers(1, 4000))
    kind = rng.integers(0, 5)
    if kind == 0:
        deg = rng.integers(0, 4, size=n)
    elif kind == 1:
        deg = rng.integers(40, 90, size=n)
    elif kind == 2:
        deg = np.minimum(rng.lognormal(2.0, 1.5, size=n), 6000).astype(np.int64)
    elif kind == 3:
        deg = rng.integers(0, 8, size=n)
        for _ in range(int(rng.integers(1, 4))):
            deg[int(rng.integers(0, n))] = int(rng.integers(900, 5000))
    else:
        deg = rng.integers(900, 1100, size=max(n // 40, 1))
    indptr = np.zeros(deg.size + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    e, nr = int(indptr[-1]), deg.size
    if e == 0:
        continue
    dim = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16]))
    idt = np.int32 if rng.random() < 0.5 else np.int64
    eids = rng.permutation(e).astype(idt) if rng.random() < 0.5 else None
    score = (rng.standard_normal((e, dim)) * rng.choice([0.5, 3.0, 20.0])).astype(np.float32)
    if "-v" in sys.argv:
        print("case", case, "kind", int(kind), "rows", nr, "edges", e, "dim", dim, "eids", eids is not None, idt.__name__, flush=True)
    tdt = torch.int32 if idt == np.int32 else torch.int64
    ip = torch.from_numpy(indptr.astype(idt)).to(dev)
    csr = _capi.make_csr(ip, torch.zeros(e, dtype=tdt, device=dev), None if eids is None else torch.from_numpy(eids).to(dev), nr)
    x = torch.from_numpy(score).to(dev)
    ws = torch.empty(_capi.edge_softmax_workspace_bytes(csr, x.dtype, dim), dtype=torch.uint8, device=dev)
    out = torch.full_like(x, float("nan"))
    _capi.edge_softmax_forward(csr, x, out, ws)
    exact = oracle.edge_softmax_fwd(indptr.astype(idt), eids, score.astype(np.float64))
    got = out.cpu().numpy()
    err = np.abs(got - exact)
    ok_f = np.isfinite(got).all() and (err <= 1e-5 * np.abs(exact) + 1e-7).all()
    sds = rng.standard_normal((e, dim)).astype(np.float32) * got
    back = torch.full_like(x, float("nan"))
    _capi.edge_softmax_backward(csr, out, torch.from_numpy(sds).to(dev), back, ws, plan_valid=True)
    ref_b = oracle.edge_softmax_bwd(indptr.astype(idt), eids, got.astype(np.float64), sds.astype(np.float64))
    torch.cuda.synchronize()
    gb = back.cpu().numpy()
    ok_b = np.isfinite(gb).all() and (np.abs(gb - ref_b) <= 1e-4 * np.abs(ref_b) + 3e-6).all()
    if not (ok_f and ok_b):
        bad += 1
        print("BAD case", case, "kind", kind, "rows", nr, "edges", e, "dim", dim, "eids", eids is not None, idt.__name__,
              "fwd", ok_f, float(err.max()), "bwd", ok_b, flush=True)
print("cases done, bad =", bad)
