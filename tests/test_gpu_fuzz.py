"""Seeded random stress of the CSR g-SpMM / SDDMM / segment reduce kernels against the oracle:
graph sizes chosen around the kernels' internal boundaries (512-item merge units, 64-lane
groups, 16-byte pieces), random degree skew, every op / reducer / id width / dtype.  Integer
outputs and max / min values bit-exact; sums within 1e-5 of the exact (fp64) result."""
import numpy as np
import pytest
import torch

import oracle
from tests.graphgen import coo_to_csc

pytestmark = pytest.mark.gpu

OPS = ["add", "sub", "mul", "div", "copy_lhs", "copy_rhs"]


def _case(rng):
    kind = rng.integers(0, 5)
    if kind == 0:      # around one merge unit
        n_dst = int(rng.integers(1, 40))
        e = int(512 - n_dst + rng.integers(-3, 4))
    elif kind == 1:    # many empty rows + one hub
        n_dst = int(rng.integers(300, 1500))
        e = int(rng.integers(600, 4000))
    elif kind == 2:    # exact multiples of the unit size
        n_dst = int(rng.choice([64, 128, 256]))
        e = int(rng.choice([512, 1024, 2048])) - n_dst
    elif kind == 3:    # tiny
        n_dst = int(rng.integers(1, 6))
        e = int(rng.integers(0, 12))
    else:
        n_dst = int(rng.integers(1, 3000))
        e = int(rng.integers(0, 9000))
    e = max(e, 0)
    n_src = int(rng.integers(1, 2000))
    src = rng.integers(0, n_src, e)
    skew = rng.choice([1.0, 2.0, 6.0])
    dst = np.minimum((rng.random(e) ** skew * n_dst).astype(np.int64), n_dst - 1)
    if kind == 1 and e:
        dst[: e // 2] = rng.integers(0, n_dst)   # hub
    feat = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 25, 33, 64, 65, 100, 130]))
    return n_src, n_dst, src, dst, feat


@pytest.mark.parametrize("chunk", range(12))
def test_spmm_random_cases(dev, chunk):
    from dgl_amd import _capi

    rng = np.random.default_rng(1000 + chunk)
    for it in range(25):
        n_src, n_dst, src, dst, f = _case(rng)
        op = OPS[rng.integers(0, 6)]
        red = ["sum", "max", "min"][rng.integers(0, 3)]
        idt = [np.int32, np.int64][rng.integers(0, 2)]
        dt = [np.float32, np.float64][rng.integers(0, 2)]
        use_eids = bool(rng.integers(0, 2))
        indptr, indices, eids = coo_to_csc(src, dst, n_dst, idt)
        u = (rng.random((n_src, f)) + 0.5).astype(dt) if op != "copy_rhs" else None
        escalar = bool(rng.integers(0, 2)) and op not in ("copy_lhs", "copy_rhs") and f > 1
        w = (rng.random((len(src), 1 if escalar else f)) + 0.5).astype(dt) if op != "copy_lhs" else None
        if not use_eids:
            w = None if w is None else w[eids]
            eids = None
        ref, ru, re_ = oracle.spmm_csr(op, red, indptr, indices, eids, u, w)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep = (t(indptr), t(indices), t(eids))
        csr = _capi.make_csr(keep[0], keep[1], keep[2], n_src)
        tu, tw = t(u), t(w)
        out = torch.full(ref.shape, 9.0, dtype=(tu if tu is not None else tw).dtype, device=dev)
        tid = torch.int32 if idt == np.int32 else torch.int64
        au = torch.full(ref.shape, -7, dtype=tid, device=dev) if red != "sum" and u is not None else None
        ae = torch.full(ref.shape, -7, dtype=tid, device=dev) if red != "sum" and w is not None else None
        nb = _capi.spmm_csr_workspace_bytes(op, red, csr, out.dtype, tu, tw, out)
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        if n_dst == 0 or f == 0:
            continue
        _capi.spmm_csr(op, red, csr, tu, tw, out, au, ae, ws)
        tag = (chunk, it, op, red, n_dst, len(src), f, idt.__name__, dt.__name__)
        got = out.cpu().numpy()
        if red == "sum":
            f64 = lambda a: None if a is None else a.astype(np.float64)
            exact = oracle.spmm_csr(op, red, indptr, indices, eids, f64(u), f64(w))[0]
            np.testing.assert_allclose(got, exact, rtol=1e-5 if dt == np.float32 else 1e-12,
                                       atol=1e-6 if dt == np.float32 else 1e-12, err_msg=str(tag))
        else:
            np.testing.assert_array_equal(got, ref, err_msg=str(tag))
            if ru is not None:
                np.testing.assert_array_equal(au.cpu().numpy(), ru, err_msg=str(tag))
            if re_ is not None:
                np.testing.assert_array_equal(ae.cpu().numpy(), re_, err_msg=str(tag))


@pytest.mark.parametrize("chunk", range(4))
def test_sddmm_and_segment_random_cases(dev, chunk):
    from dgl_amd import _capi

    rng = np.random.default_rng(5000 + chunk)
    for it in range(25):
        n_src, n_dst, src, dst, f = _case(rng)
        if len(src) == 0:
            continue
        idt = [np.int32, np.int64][rng.integers(0, 2)]
        op = (OPS + ["dot"])[rng.integers(0, 7)]
        lt, rt = "uev"[rng.integers(0, 3)], "uev"[rng.integers(0, 3)]
        cnt = {"u": n_src, "e": len(src), "v": n_dst}
        lhs = (rng.random((cnt[lt], f)) + 0.5).astype(np.float32) if op != "copy_rhs" else None
        rhs = (rng.random((cnt[rt], f)) + 0.5).astype(np.float32) if op != "copy_lhs" else None
        row, col = src.astype(idt), dst.astype(idt)
        want = oracle.sddmm_coo(op, row, col, None, lhs, rhs, lt, rt)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep = (t(row), t(col))
        coo = _capi.make_coo(keep[0], keep[1], None, n_src, n_dst)
        out = torch.full(want.shape if want.ndim > 1 else (want.shape[0], 1), 3.0, device=dev)
        _capi.sddmm_coo(op, coo, t(lhs), t(rhs), out, _capi.TARGETS[lt], _capi.TARGETS[rt])
        got = out.cpu().numpy().reshape(want.shape)
        if op == "dot":
            np.testing.assert_allclose(got, want, rtol=1e-5, err_msg=str((chunk, it, op, lt, rt, f)))
        else:
            np.testing.assert_array_equal(got, want, err_msg=str((chunk, it, op, lt, rt, f)))
        # segment reduce over the same degree sequence
        seglen = np.bincount(dst, minlength=n_dst)
        off = np.concatenate([[0], np.cumsum(seglen)]).astype(idt)
        feat = rng.standard_normal((len(src), f)).astype(np.float32)
        red = ["sum", "max", "min"][rng.integers(0, 3)]
        w_out, w_arg = oracle.segment_reduce(red, feat, off)
        o = torch.full(w_out.shape, 5.0, device=dev)
        a = torch.full(w_out.shape, 9, dtype=torch.int32 if idt == np.int32 else torch.int64, device=dev) \
            if red != "sum" else None
        _capi.segment_reduce(red, t(feat), t(off), o, a)
        if red == "sum":
            mag, _ = oracle.segment_reduce("sum", np.abs(feat).astype(np.float64), off)
            err = np.abs(o.cpu().numpy().astype(np.float64) - w_out)
            assert (err <= (1e-5 + 2 * seglen.max() * 2.0 ** -24) * mag + 1e-30).all()
        else:
            np.testing.assert_array_equal(o.cpu().numpy(), w_out)
            np.testing.assert_array_equal(a.cpu().numpy(), w_arg)
