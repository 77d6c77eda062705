"""Seeded random stress of segment_mm / gather_mm (forward, transposed weights, row-indexed,
weight gradient) around the kernels' internal boundaries: 128-row tiles, 64-byte K slabs and
their tails, 16-byte pieces (LDS-direct kernels) vs odd widths (register-staged kernels),
128 / 256-wide output tiles, empty relations, ring depths (K, rows per slab).  Checked against
the exact fp64 product with the condition-aware bound of tests/test_mm.py."""
import numpy as np
import pytest
import torch

from tests.test_mm import _close, _exact_segment_mm

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16, torch.float64]


def _dims(rng):
    pick = rng.integers(0, 4)
    if pick == 0:    # whole 16-byte pieces: LDS-direct kernels, with and without K tails
        return int(rng.choice([8, 16, 24, 32, 40, 64, 72, 104, 128, 200, 256, 520])), \
               int(rng.choice([8, 16, 64, 72, 128, 136, 256, 264, 512]))
    if pick == 1:    # fp32-only pieces (multiples of 4)
        return int(rng.choice([4, 12, 20, 36, 100, 132, 260])), int(rng.choice([4, 12, 36, 100, 132, 260]))
    if pick == 2:    # odd widths: register-staged kernels
        return int(rng.integers(1, 300)), int(rng.integers(1, 300))
    return int(rng.integers(1, 40)), int(rng.integers(1, 40))


def _seglen(rng):
    r = int(rng.integers(1, 12))
    kind = rng.integers(0, 4)
    if kind == 0:
        seg = rng.integers(0, 300, r)
    elif kind == 1:   # around tile / slab sizes
        seg = rng.choice([0, 1, 31, 32, 33, 127, 128, 129, 255, 256, 257, 2047, 2048, 2049], r)
    elif kind == 2:   # one long relation (several split-K slabs, deep ring)
        seg = rng.integers(0, 50, r)
        seg[rng.integers(0, r)] = int(rng.integers(3000, 9000))
    else:
        seg = rng.integers(0, 5, r)
    if seg.sum() == 0:
        seg[0] = 1
    return [int(v) for v in seg]


@pytest.mark.parametrize("chunk", range(8))
def test_segment_mm_random_cases(dev, chunk):
    from dgl_amd import _capi

    rng = np.random.default_rng(9000 + chunk)
    for _ in range(24):
        dtype = DTYPES[int(rng.integers(0, len(DTYPES)))]
        d1, d2 = _dims(rng)
        seg = _seglen(rng)
        m, r = sum(seg), len(seg)
        g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
        a = (torch.rand(m, d1, generator=g) - 0.3).to(dtype).to(dev)
        b = (torch.rand(r, d1, d2, generator=g) - 0.6).to(dtype).to(dev)
        sl = torch.tensor(seg, dtype=torch.int64 if rng.integers(0, 2) else torch.int32)
        if rng.integers(0, 2):
            sl = sl.to(dev)
        tag = (str(dtype), d1, d2, seg)
        # forward, contiguous and through a row permutation (the core of gather_mm)
        want, mag = _exact_segment_mm(a, b, seg)
        c = torch.full((m, d2), 7.0, dtype=dtype, device=dev)
        _capi.segment_mm(a, b, c, sl)
        _close(c, want, mag, dtype, d1, ("fwd",) + tag)
        perm = torch.randperm(m, generator=g).to(dev)
        ap = torch.empty_like(a)
        ap[perm] = a                                  # logical row i lives at physical row perm[i]
        cp = torch.full((m, d2), 7.0, dtype=dtype, device=dev)
        _capi.segment_mm(ap, b, cp, sl, row_index=perm)
        _close(cp[perm], want, mag, dtype, d1, ("fwd indexed",) + tag)
        # transposed weights: dA = dC . B^T
        want_t, mag_t = _exact_segment_mm(c, b, seg, b_trans=True)
        da = torch.full((m, d1), 7.0, dtype=dtype, device=dev)
        _capi.segment_mm(c, b, da, sl, b_trans=True)
        _close(da, want_t, mag_t, dtype, d2, ("b_trans",) + tag)
        # weight gradient, contiguous and row-indexed
        dc = (torch.rand(m, d2, generator=g) - 0.6).to(dtype).to(dev)
        db = torch.full((r, d1, d2), 7.0, dtype=dtype, device=dev)
        _capi.segment_mm_backward_b(a, dc, db, sl)
        dcp = torch.empty_like(dc)
        dcp[perm] = dc
        dbp = torch.full((r, d1, d2), 7.0, dtype=dtype, device=dev)
        _capi.segment_mm_backward_b(ap, dcp, dbp, sl, row_index=perm)
        a64, c64 = a.double().cpu().numpy(), dc.double().cpu().numpy()
        off = 0
        for i, n_ in enumerate(seg):
            w = a64[off:off + n_].T @ c64[off:off + n_]
            mg = np.abs(a64[off:off + n_]).T @ np.abs(c64[off:off + n_])
            _close(db[i], w, mg, dtype, n_, ("dB rel %d" % i,) + tag)
            _close(dbp[i], w, mg, dtype, n_, ("dB indexed rel %d" % i,) + tag)
            off += n_
