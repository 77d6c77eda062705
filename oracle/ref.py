"""Front end of ``oracle/_ref/libdglref.so`` — the REFERENCE's own CPU g-SpMM / g-SDDMM
kernels compiled from /root/reference (recipe: oracle/Makefile, driver: ref_driver.cc).

TEST INFRASTRUCTURE ONLY, same rule as the rest of ``oracle/``.  Function signatures match
``oracle/__init__.py`` so a test can run both and demand bit equality:

    oracle.spmm_csr(...)      our plain-C restatement
    oracle.ref.spmm_csr(...)  dgl::aten::SpMMCsr<kDGLCPU, IdType, DType> itself

``available()`` is False when the library was never built (e.g. a checkout without
/root/reference); callers skip, they never fall back silently.
"""
import ctypes
import os

import numpy as np

from . import infer_broadcast_shape

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libdglref.so")
_lib = None

TARGETS = {"u": 0, "e": 1, "v": 2}


class _Feat(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("ndim", ctypes.c_int32),
                ("shape", ctypes.POINTER(ctypes.c_int64))]


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libdglref.so is not built (make -C oracle ref needs "
                               "the reference checkout at /root/reference)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ref_last_error.restype = ctypes.c_char_p
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("reference raised: " + lib().ref_last_error().decode("utf-8", "replace"))


try:  # numpy has no bfloat16; bf16 buffers travel as uint16 with dtype code 3
    import ml_dtypes  # noqa: F401
except Exception:  # pragma: no cover
    ml_dtypes = None


def _dcode(a, bf16=False):
    if bf16:
        assert a.dtype == np.uint16
        return 3
    return {np.dtype(np.float32): 0, np.dtype(np.float64): 1}[a.dtype]


def _feat(a):
    """(struct, keepalive) for a C-contiguous array or None."""
    if a is None:
        return None, None
    shp = (ctypes.c_int64 * a.ndim)(*a.shape)
    return _Feat(a.ctypes.data_as(ctypes.c_void_p), a.ndim, shp), (a, shp)


def _fp(f):
    return None if f is None else ctypes.byref(f)


def _prep(x):
    if x is None:
        return None
    x = np.ascontiguousarray(x)
    return x.reshape(x.shape[0], 1) if x.ndim == 1 else x


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i64(v):
    return ctypes.c_int64(int(v))


def _shapes(op, u, e):
    use_l, use_r = op != "copy_rhs", op != "copy_lhs"
    lshape = u.shape if use_l else e.shape
    rshape = e.shape if use_r else u.shape
    return use_l, use_r, infer_broadcast_shape(op, lshape[1:], rshape[1:])


def set_num_threads(n):
    """The reference's parallel_for runs on the OpenMP default team
    (include/dgl/runtime/parallel_for.h:46-113)."""
    omp = ctypes.CDLL("libgomp.so.1")
    omp.omp_set_num_threads(int(n))


def calc_bcast_off(op, lhs_shape, rhs_shape):
    """dgl::CalcBcastOff (src/bcast.cc:36-90) on dummy arrays of the given full shapes.
    Returns dict(use_bcast, lhs_len, rhs_len, out_len, reduce_size, lhs_offset, rhs_offset)."""
    def fake(shape):
        shp = (ctypes.c_int64 * len(shape))(*shape)
        return _Feat(ctypes.c_void_p(1), len(shape), shp), shp  # data non-NULL, never read

    lf, k1 = fake(tuple(lhs_shape))
    rf, k2 = fake(tuple(rhs_shape))
    lens = (ctypes.c_int64 * 4)()
    cap = 1 << 16
    lo = np.zeros(cap, dtype=np.int64)
    ro = np.zeros(cap, dtype=np.int64)
    use = lib().ref_calc_bcast_off(op.encode(), ctypes.byref(lf), ctypes.byref(rf), lens,
                                   _ptr(lo), _ptr(ro), _i64(cap))
    if use < 0:
        _check(-1)
    n = lens[2] if use else 0
    return {"use_bcast": bool(use), "lhs_len": lens[0], "rhs_len": lens[1], "out_len": lens[2],
            "reduce_size": lens[3], "lhs_offset": lo[:n].copy() if use else None,
            "rhs_offset": ro[:n].copy() if use else None}


def spmm_csr(op, reduce, indptr, indices, eids, ufeat, efeat, num_cols=None, bf16=False, out=None):
    """`out` (optional): a pre-ZEROED array to write into, so a timing loop measures the
    kernel alone (the reference's Python caller allocates it, _sparse_ops.py:227)."""
    indptr = np.ascontiguousarray(indptr)
    idt = indptr.dtype
    indices = np.ascontiguousarray(indices, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    squeeze = (ufeat is None or ufeat.ndim == 1) and (efeat is None or efeat.ndim == 1)
    u, e = _prep(ufeat), _prep(efeat)
    use_l, use_r, fshape = _shapes(op, u, e)
    fdt = (u if u is not None else e).dtype
    n_rows = indptr.shape[0] - 1
    if num_cols is None:
        num_cols = u.shape[0] if use_l and u is not None else (int(indices.max()) + 1 if indices.size else 0)
    if out is None:
        out = np.zeros((n_rows,) + fshape, dtype=fdt)
    assert out.shape == (n_rows,) + fshape and out.dtype == fdt and out.flags.c_contiguous
    argu = arge = None
    if reduce != "sum":
        argu = np.zeros(out.shape, dtype=idt)
        arge = np.zeros(out.shape, dtype=idt)
    uf, k1 = _feat(u if use_l else None)
    ef, k2 = _feat(e if use_r else None)
    of, k3 = _feat(out)
    _check(lib().ref_spmm_csr(op.encode(), reduce.encode(), idt.itemsize * 8, _dcode(out, bf16),
                              _i64(n_rows), _i64(num_cols), _i64(indices.shape[0]),
                              _ptr(indptr), _ptr(indices), _ptr(eids), _fp(uf), _fp(ef), _fp(of),
                              _ptr(argu), _ptr(arge)))
    if reduce != "sum":
        if not use_l:
            argu = None
        if not use_r:
            arge = None
    if squeeze:
        out = out.reshape(-1)
        argu = None if argu is None else argu.reshape(-1)
        arge = None if arge is None else arge.reshape(-1)
    return out, argu, arge


def spmm_coo(op, reduce, row, col, eids, num_dst, ufeat, efeat, num_src=None, bf16=False):
    row = np.ascontiguousarray(row)
    idt = row.dtype
    col = np.ascontiguousarray(col, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    squeeze = (ufeat is None or ufeat.ndim == 1) and (efeat is None or efeat.ndim == 1)
    u, e = _prep(ufeat), _prep(efeat)
    use_l, use_r, fshape = _shapes(op, u, e)
    fdt = (u if u is not None else e).dtype
    if num_src is None:
        num_src = u.shape[0] if use_l and u is not None else (int(row.max()) + 1 if row.size else 0)
    out = np.zeros((num_dst,) + fshape, dtype=fdt)
    argu = arge = None
    if reduce != "sum":
        argu = np.zeros(out.shape, dtype=idt)
        arge = np.zeros(out.shape, dtype=idt)
    uf, k1 = _feat(u if use_l else None)
    ef, k2 = _feat(e if use_r else None)
    of, k3 = _feat(out)
    _check(lib().ref_spmm_coo(op.encode(), reduce.encode(), idt.itemsize * 8, _dcode(out, bf16),
                              _i64(num_src), _i64(num_dst), _i64(row.shape[0]), _ptr(row),
                              _ptr(col), _ptr(eids), _fp(uf), _fp(ef), _fp(of), _ptr(argu),
                              _ptr(arge)))
    if reduce != "sum":
        if not use_l:
            argu = None
        if not use_r:
            arge = None
    if squeeze:
        out = out.reshape(-1)
        argu = None if argu is None else argu.reshape(-1)
        arge = None if arge is None else arge.reshape(-1)
    return out, argu, arge


def _sddmm(fn, op, graph_args, nnz, lhs, rhs, lhs_target, rhs_target, idt, bf16):
    squeeze = (lhs is None or lhs.ndim == 1) and (rhs is None or rhs.ndim == 1)
    l, r = _prep(lhs), _prep(rhs)
    use_l, use_r, fshape = _shapes(op, l, r)
    fdt = (l if l is not None else r).dtype
    out = np.zeros((nnz,) + fshape, dtype=fdt)
    lf, k1 = _feat(l if use_l else None)
    rf, k2 = _feat(r if use_r else None)
    of, k3 = _feat(out)
    _check(fn(op.encode(), idt.itemsize * 8, _dcode(out, bf16), *graph_args, _fp(lf), _fp(rf),
              _fp(of), TARGETS[lhs_target], TARGETS[rhs_target]))
    return out.reshape(-1) if squeeze else out


def sddmm_coo(op, row, col, eids, lhs, rhs, lhs_target="u", rhs_target="v", num_src=None,
              num_dst=None, bf16=False):
    row = np.ascontiguousarray(row)
    idt = row.dtype
    col = np.ascontiguousarray(col, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    ns = num_src if num_src is not None else (int(row.max()) + 1 if row.size else 0)
    nd = num_dst if num_dst is not None else (int(col.max()) + 1 if col.size else 0)
    args = (_i64(ns), _i64(nd), _i64(row.shape[0]), _ptr(row), _ptr(col), _ptr(eids))
    return _sddmm(lib().ref_sddmm_coo, op, args, row.shape[0], lhs, rhs, lhs_target, rhs_target,
                  idt, bf16)


def sddmm_csr(op, indptr, indices, eids, lhs, rhs, lhs_target="u", rhs_target="v",
              num_cols=None, bf16=False):
    indptr = np.ascontiguousarray(indptr)
    idt = indptr.dtype
    indices = np.ascontiguousarray(indices, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    nc = num_cols if num_cols is not None else (int(indices.max()) + 1 if indices.size else 0)
    args = (_i64(indptr.shape[0] - 1), _i64(nc), _i64(indices.shape[0]), _ptr(indptr),
            _ptr(indices), _ptr(eids))
    return _sddmm(lib().ref_sddmm_csr, op, args, indices.shape[0], lhs, rhs, lhs_target,
                  rhs_target, idt, bf16)


def edge_softmax_fwd(indptr, eids, score, indices=None, bf16=False):
    """`indices` is not read by the kernel (spmm.h:484-522); zeros stand in when omitted."""
    indptr = np.ascontiguousarray(indptr)
    idt = indptr.dtype
    indices = np.zeros(int(indptr[-1]), dtype=idt) if indices is None else \
        np.ascontiguousarray(indices, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    s = _prep(score)
    out = np.zeros_like(s)
    sf, k1 = _feat(s)
    of, k2 = _feat(out)
    nc = int(indices.max()) + 1 if indices.size else 0
    _check(lib().ref_edge_softmax_forward(idt.itemsize * 8, _dcode(s, bf16),
                                          _i64(indptr.shape[0] - 1), _i64(nc),
                                          _i64(indices.shape[0]), _ptr(indptr), _ptr(indices),
                                          _ptr(eids), _fp(sf), _fp(of)))
    return out.reshape(np.shape(score))


def edge_softmax_bwd(indptr, eids, out, sds, indices=None, bf16=False):
    indptr = np.ascontiguousarray(indptr)
    idt = indptr.dtype
    indices = np.zeros(int(indptr[-1]), dtype=idt) if indices is None else \
        np.ascontiguousarray(indices, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    o, s = _prep(out), _prep(np.ascontiguousarray(sds, dtype=np.asarray(out).dtype))
    back = np.zeros_like(o)
    f1, k1 = _feat(o)
    f2, k2 = _feat(s)
    f3, k3 = _feat(back)
    nc = int(indices.max()) + 1 if indices.size else 0
    _check(lib().ref_edge_softmax_backward(idt.itemsize * 8, _dcode(o, bf16),
                                           _i64(indptr.shape[0] - 1), _i64(nc),
                                           _i64(indices.shape[0]), _ptr(indptr), _ptr(indices),
                                           _ptr(eids), _fp(f1), _fp(f2), _fp(f3)))
    return back.reshape(np.shape(out))


# --------------------------------------------------------------------------- #
# segment reduce family (src/array/cpu/segment_reduce.cc)
# --------------------------------------------------------------------------- #
def segment_reduce(reduce, feat, offsets):
    offsets = np.ascontiguousarray(offsets)
    idt = offsets.dtype
    f = np.ascontiguousarray(feat)
    n = offsets.shape[0] - 1
    out = np.zeros((n,) + f.shape[1:], dtype=f.dtype)
    arg = None if reduce == "sum" else np.zeros(out.shape, dtype=idt)
    ff, k1 = _feat(f)
    of, k2 = _feat(out)
    _check(lib().ref_segment_reduce(reduce.encode(), idt.itemsize * 8, _dcode(out), _i64(n),
                                    _ptr(offsets), _fp(ff), _fp(of), _ptr(arg)))
    return out, arg


def scatter_add(feat, idx, out):
    idx = np.ascontiguousarray(idx)
    f = np.ascontiguousarray(feat, dtype=out.dtype)
    assert out.flags.c_contiguous
    ff, k1 = _feat(f)
    of, k2 = _feat(out)
    _check(lib().ref_scatter_add(idx.dtype.itemsize * 8, _dcode(out), _i64(f.shape[0]), _ptr(idx),
                                 _fp(ff), _fp(of)))
    return out


def backward_segment_cmp(feat, arg, out):
    arg = np.ascontiguousarray(arg)
    f = np.ascontiguousarray(feat, dtype=out.dtype)
    assert out.flags.c_contiguous
    ff, k1 = _feat(f)
    of, k2 = _feat(out)
    _check(lib().ref_backward_segment_cmp(arg.dtype.itemsize * 8, _dcode(out), _fp(ff), _ptr(arg),
                                          _fp(of)))
    return out


def coo_to_csr(row, col, data, num_rows, num_cols=None):
    """aten::impl::COOToCSR<kDGLCPU> (src/array/cpu/spmat_op_impl_coo.cc:747-764): returns
    ``(indptr, indices, data)`` with ``data`` = edge id of every CSR position."""
    row = np.ascontiguousarray(row)
    idt = row.dtype
    col = np.ascontiguousarray(col, dtype=idt)
    data = None if data is None else np.ascontiguousarray(data, dtype=idt)
    if num_cols is None:
        num_cols = int(col.max()) + 1 if col.size else 0
    nnz = row.shape[0]
    indptr = np.zeros(num_rows + 1, dtype=idt)
    indices = np.zeros(nnz, dtype=idt)
    out = np.zeros(nnz, dtype=idt)
    _check(lib().ref_coo_to_csr(idt.itemsize * 8, _i64(num_rows), _i64(num_cols), _i64(nnz), _ptr(row),
                                _ptr(col), _ptr(data), _ptr(indptr), _ptr(indices), _ptr(out)))
    return indptr, indices, out


# --------------------------------------------------------------------------- #
# heterograph SpMM (src/array/cpu/spmm.cc:45-150 SpMMCsrHetero)
# --------------------------------------------------------------------------- #
def spmm_csr_hetero(op, reduce, rels, num_nodes, ufeats, efeats):
    """``rels``: list of dicts(indptr, indices, eids, src, dst) — one in-edge CSR per relation with
    its (src, dst) node-type ids; ``ufeats`` per node type, ``efeats`` per relation (entries may
    be None where unused).  Returns ``(outs, arg_u, arg_e, arg_u_ntype, arg_e_etype)`` per node
    type (None where the type receives nothing / the array is not produced)."""
    use_u, use_e = op != "copy_rhs", op != "copy_lhs"
    n_et, n_nt = len(rels), len(num_nodes)
    idt = np.asarray(rels[0]["indptr"]).dtype
    any_feat = next(f for f in (list(ufeats) if use_u else []) + (list(efeats) if use_e else []) if f is not None)
    fdt = any_feat.dtype
    u = [None if (not use_u or f is None) else _prep(f) for f in (ufeats if use_u else [None] * n_nt)]
    e = [None if (not use_e or f is None) else _prep(f) for f in (efeats if use_e else [None] * n_et)]
    u0 = u[rels[0]["src"]] if use_u else None
    e0 = e[0] if use_e else None
    _, _, fshape = _shapes(op, u0, e0)
    touched = {r["dst"] for r in rels}
    outs = [np.zeros((num_nodes[nt],) + fshape, dtype=fdt) if nt in touched else None for nt in range(n_nt)]
    cmp = reduce != "sum"
    mk = lambda on: [np.zeros(o.shape, dtype=idt) if (o is not None and cmp and on) else None for o in outs]
    au, ae, aut, aet = mk(use_u), mk(use_e), mk(use_u), mk(use_e)
    I64 = ctypes.c_int64 * n_et
    VP = ctypes.c_void_p * n_et
    VPN = ctypes.c_void_p * n_nt
    ips = [np.ascontiguousarray(r["indptr"], dtype=idt) for r in rels]
    ixs = [np.ascontiguousarray(r["indices"], dtype=idt) for r in rels]
    eis = [None if r["eids"] is None else np.ascontiguousarray(r["eids"], dtype=idt) for r in rels]
    vp = lambda a: None if a is None else a.ctypes.data
    feats_n, keep = (_Feat * n_nt)(), []
    outs_n = (_Feat * n_nt)()
    for nt in range(n_nt):
        for arr, slot in ((u[nt], feats_n), (outs[nt], outs_n)):
            if arr is None:
                slot[nt] = _Feat(None, 0, None)
            else:
                f, k = _feat(arr)
                slot[nt] = f
                keep.append(k)
    feats_e = (_Feat * n_et)()
    for et in range(n_et):
        if e[et] is None:
            feats_e[et] = _Feat(None, 0, None)
        else:
            f, k = _feat(e[et])
            feats_e[et] = f
            keep.append(k)
    I32 = ctypes.c_int32 * n_et
    _check(lib().ref_spmm_csr_hetero(
        op.encode(), reduce.encode(), idt.itemsize * 8, _dcode(any_feat if any_feat.ndim > 1 else _prep(any_feat)),
        n_et, n_nt, I64(*[len(p) - 1 for p in ips]), I64(*[num_nodes[r["src"]] for r in rels]),
        I64(*[len(x) for x in ixs]), VP(*[vp(p) for p in ips]), VP(*[vp(x) for x in ixs]),
        VP(*[vp(x) for x in eis]), I32(*[r["src"] for r in rels]), I32(*[r["dst"] for r in rels]),
        feats_n, feats_e, outs_n, VPN(*[vp(a) for a in au]), VPN(*[vp(a) for a in ae]),
        VPN(*[vp(a) for a in aut]), VPN(*[vp(a) for a in aet])))
    return outs, au, ae, aut, aet
