#!/usr/bin/env python
"""A/B of the two 16-bit segment_mm forward kernels (register-staged vs LDS-direct,
DGLA_TUNE_GLDS): bit-equality over ragged shapes, then timing at the R-GCN shape."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgl_amd import _capi  # noqa: E402

GLDS = 16
dev = torch.device("cuda:0")
base = _capi.get_tuning() & ~GLDS


def run(a, b, seglen, n, flags, b_trans=False, row_index=None):
    _capi.set_tuning(flags)
    c = torch.full((a.shape[0], n), float("nan"), dtype=a.dtype, device=dev)
    _capi.segment_mm(a, b, c, seglen, b_trans=b_trans, row_index=row_index)
    torch.cuda.synchronize()
    return c


bad = 0
g = torch.Generator(device="cpu").manual_seed(5)
def same(c0, c1, dt, a=None, b=None, sl=None):
    if dt != torch.float32:
        return torch.equal(c0.view(torch.int16), c1.view(torch.int16))
    # fp32: the LDS-direct kernel contracts k in a permuted order -> compare both with fp64
    return torch.allclose(c0, c1, rtol=1e-4, atol=1e-4 * float(c0.abs().max()))


for dt in (torch.bfloat16, torch.float16, torch.float32):
    for k in (24, 32, 64, 100, 256, 544):
        for n in (136, 256, 264, 512, 776):
            for seg in ([1], [127, 129, 0, 5], [1000, 3, 0, 0, 2049], [300] * 9):
                for indexed in (False, True):
                    m = sum(seg)
                    r = len(seg)
                    a = torch.randn((m, k), generator=g).to(dt).to(dev)
                    b = torch.randn((r, k, n), generator=g).to(dt).to(dev)
                    sl = torch.tensor(seg, dtype=torch.int64)
                    ri = torch.randperm(m, generator=g).to(dev) if indexed else None
                    c0 = run(a, b, sl, n, base, row_index=ri)
                    c1 = run(a, b, sl, n, base | GLDS, row_index=ri)
                    ok = same(c0, c1, dt)
                    bt = b.transpose(1, 2).contiguous()
                    c2 = run(a, bt, sl, n, base | GLDS, b_trans=True, row_index=ri)
                    ok = ok and same(c0, c2, dt)
                    if not ok:
                        bad += 1
                        d = (c0.float() - c1.float()).abs()
                        print("MISMATCH", dt, k, n, seg, indexed, "max", float(torch.nan_to_num(d, nan=1e9).max()),
                              "rows", torch.nonzero(torch.nan_to_num(d, nan=1e9).amax(1) > 0).flatten()[:8].tolist(), flush=True)
# big ragged case (many tiles per relation, every XCD slot in use)
for dt in (torch.bfloat16, torch.float16, torch.float32):
    for k, n in ((64, 256), (256, 264), (96, 520)):
        seg = [100_000, 1, 0, 255, 257, 150_003, 77_777, 12]
        m, r = sum(seg), len(seg)
        a = torch.randn((m, k), generator=g).to(dt).to(dev)
        b = torch.randn((r, k, n), generator=g).to(dt).to(dev)
        sl = torch.tensor(seg, dtype=torch.int64)
        for ri in (None, torch.randperm(m, generator=g).to(dev)):
            c0 = run(a, b, sl, n, base, row_index=ri)
            c1 = run(a, b, sl, n, base | GLDS, row_index=ri)
            if not same(c0, c1, dt):
                bad += 1
                print("MISMATCH-large", dt, k, n, ri is not None, flush=True)
print(json.dumps({"check": "glds vs register-staged, bit equality", "mismatches": bad}), flush=True)


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    return float(np.median(ts)), float(np.min(ts))


for (rows, k, n, r) in ((10_000_000, 256, 256, 8), (10_000_000, 128, 256, 8), (10_000_000, 104, 256, 8), (4_000_000, 256, 512, 8), (2_000_000, 1024, 1024, 4)):
    for dt in (torch.bfloat16, torch.float32):
        a = torch.randn((rows, k), device=dev, dtype=dt)
        b = torch.randn((r, k, n), device=dev, dtype=dt) * 0.05
        c = torch.empty((rows, n), device=dev, dtype=dt)
        sl = torch.full((r,), rows // r, dtype=torch.int64)
        for flags, name in ((base, "reg"), (base | GLDS, "glds")):
            _capi.set_tuning(flags)
            ms, mn = timeit(lambda: _capi.segment_mm(a, b, c, sl))
            print(json.dumps({"shape": [rows, k, n, r], "dtype": str(dt), "kernel": name, "ms": round(ms, 4), "ms_min": round(mn, 4),
                              "tflops": 2.0 * rows * k * n / (ms * 1e-3) / 1e12}), flush=True)
        del a, b, c
_capi.set_tuning(base)

# ---- weight gradient: LDS-direct (transposing LDS reads) vs register-staged --------------------
bad = 0
for dt in (torch.bfloat16, torch.float16):
    for d1, d2 in ((8, 8), (64, 128), (256, 256), (136, 72), (520, 264)):
        for seg in ([1], [127, 129, 0, 5], [5000, 3, 0, 0, 2049], [40000, 17]):
            m, r = sum(seg), len(seg)
            a = torch.randn((m, d1), generator=g).to(dt).to(dev)
            dc = torch.randn((m, d2), generator=g).to(dt).to(dev)
            sl = torch.tensor(seg, dtype=torch.int64)
            outs = []
            for flags in (base, base | GLDS):
                _capi.set_tuning(flags)
                db = torch.full((r, d1, d2), float("nan"), dtype=dt, device=dev)
                _capi.segment_mm_backward_b(a, dc, db, sl)
                torch.cuda.synchronize()
                outs.append(db.float())
            off, want = 0, torch.zeros((r, d1, d2), device=dev)
            for i, n_ in enumerate(seg):
                want[i] = a[off:off + n_].float().T @ dc[off:off + n_].float()
                off += n_
            tol = (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * want.abs().clamp_min(1.0) + 1e-3 * max(seg) ** 0.5
            for name, o in zip(("reg", "glds"), outs):
                if not bool(((o - want).abs() <= tol).all()):
                    bad += 1
                    print("MISMATCH dB", name, dt, d1, d2, seg, float((o - want).abs().max()), flush=True)
print(json.dumps({"check": "weight gradient vs fp32 torch, both kernels", "mismatches": bad}), flush=True)
for (rows, d1, d2, r) in ((10_000_000, 256, 256, 8), (10_000_000, 128, 256, 8), (4_000_000, 512, 512, 8)):
    dt = torch.bfloat16
    a = torch.randn((rows, d1), device=dev, dtype=dt)
    dc = torch.randn((rows, d2), device=dev, dtype=dt)
    db = torch.empty((r, d1, d2), device=dev, dtype=dt)
    sl = torch.full((r,), rows // r, dtype=torch.int64)
    for flags, name in ((base, "reg"), (base | GLDS, "glds")):
        _capi.set_tuning(flags)
        ms, mn = timeit(lambda: _capi.segment_mm_backward_b(a, dc, db, sl))
        print(json.dumps({"dB shape": [rows, d1, d2, r], "kernel": name, "ms": round(ms, 4), "ms_min": round(mn, 4),
                          "tflops": 2.0 * rows * d1 * d2 / (ms * 1e-3) / 1e12}), flush=True)
    del a, dc, db
_capi.set_tuning(base)

# ---- fp32 weight gradient ------------------------------------------------------------------
bad = 0
for d1, d2 in ((4, 4), (64, 128), (256, 256), (132, 76), (520, 264)):
    for seg in ([1], [127, 129, 0, 5], [5000, 3, 0, 0, 2049], [40000, 17]):
        m, r = sum(seg), len(seg)
        a = torch.randn((m, d1), generator=g).to(dev)
        dc = torch.randn((m, d2), generator=g).to(dev)
        sl = torch.tensor(seg, dtype=torch.int64)
        off, want = 0, torch.zeros((r, d1, d2), device=dev, dtype=torch.float64)
        for i, n_ in enumerate(seg):
            want[i] = a[off:off + n_].double().T @ dc[off:off + n_].double()
            off += n_
        for flags in (base, base | GLDS):
            _capi.set_tuning(flags)
            db = torch.full((r, d1, d2), float("nan"), device=dev)
            _capi.segment_mm_backward_b(a, dc, db, sl)
            torch.cuda.synchronize()
            if not bool(((db.double() - want).abs() <= 1e-5 * want.abs() + 1e-4 * max(seg) ** 0.5).all()):
                bad += 1
                print("MISMATCH dB fp32", flags, d1, d2, seg, float((db.double() - want).abs().max()), flush=True)
print(json.dumps({"check": "fp32 weight gradient vs fp64 torch, both kernels", "mismatches": bad}), flush=True)
for (rows, d1, d2, r) in ((10_000_000, 256, 256, 8), (10_000_000, 128, 256, 8), (4_000_000, 512, 512, 8)):
    a = torch.randn((rows, d1), device=dev)
    dc = torch.randn((rows, d2), device=dev)
    db = torch.empty((r, d1, d2), device=dev)
    sl = torch.full((r,), rows // r, dtype=torch.int64)
    for flags, name in ((base, "reg"), (base | GLDS, "glds")):
        _capi.set_tuning(flags)
        ms, mn = timeit(lambda: _capi.segment_mm_backward_b(a, dc, db, sl))
        print(json.dumps({"dB fp32 shape": [rows, d1, d2, r], "kernel": name, "ms": round(ms, 4), "ms_min": round(mn, 4),
                          "tflops": 2.0 * rows * d1 * d2 / (ms * 1e-3) / 1e12}), flush=True)
    del a, dc, db
_capi.set_tuning(base)
