"""CPU oracle for the g-SpMM / g-SDDMM hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product package ``dgl_amd`` never does (a test greps for it).

The arithmetic lives in ``oracle.c`` / ``kernels_impl.inc`` (plain C + OpenMP restating
``src/array/cpu/{spmm.h,spmm.cc,sddmm.h,spmm_binary_ops.h}`` of the reference); this
module is the numpy front end plus the broadcast bookkeeping of ``src/bcast.cc``.

Parity status: PINNED to the reference build.  ``oracle.ref`` wraps
``oracle/_ref/libdglref.so`` — the reference's own CPU kernels compiled from
/root/reference by ``oracle/Makefile`` — and ``tests/test_oracle_vs_reference.py`` demands
bit equality with this restatement over an exhaustive sweep; the reference build's outputs
are committed as ``tests/golden/reference_cpu_outputs.npz``.  See the oracle.c header for
what remains unpinned (libxsmm / cuSPARSE summation order).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "copy_lhs": 4, "copy_rhs": 5, "dot": 6}
TARGETS = {"u": 0, "e": 1, "v": 2}


def build(force=False):
    """Compile oracle.c with gcc (``make -C oracle``)."""
    src_m = max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("oracle.c", "kernels_impl.inc")
    )
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    """The oracle library; DGLA_ORACLE_LIB names another build of the same sources (the
    AddressSanitizer / UBSan build of tests/test_oracle_sanitized.py)."""
    global _lib
    if _lib is None:
        alt = os.environ.get("DGLA_ORACLE_LIB")
        if alt:
            _lib = ctypes.CDLL(alt)
        else:
            build()
            _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


# --------------------------------------------------------------------------- #
# Broadcast bookkeeping
# --------------------------------------------------------------------------- #
class BcastOff:
    """Restates ``struct BcastOff`` (include/dgl/bcast.h:20-53) as computed by
    ``CalcBcastOff`` (src/bcast.cc:36-90).  ``lhs_offset``/``rhs_offset`` are None when
    ``use_bcast`` is False."""

    def __init__(self, op, lhs_shape, rhs_shape):
        # shapes include the leading (node/edge) dimension; ndim >= 2 each
        # (src/array/check.h:46-50).  A missing operand is given the other's shape by the
        # callers below, mirroring that copy_lhs / copy_rhs ignore it (bcast.cc:18-20).
        lhs_shape, rhs_shape = tuple(lhs_shape), tuple(rhs_shape)
        self.lhs_len = int(np.prod(lhs_shape[1:], dtype=np.int64))
        self.rhs_len = int(np.prod(rhs_shape[1:], dtype=np.int64))
        self.reduce_size = 1
        use = False
        if op not in ("copy_lhs", "copy_rhs"):
            use = len(lhs_shape) != len(rhs_shape) or lhs_shape[1:] != rhs_shape[1:]
        self.use_bcast = use
        self.lhs_offset = self.rhs_offset = None
        if use:
            lnd, rnd = len(lhs_shape), len(rhs_shape)
            max_ndim = max(lnd, rnd) - 1
            out_len, j = 1, 0
            if op == "dot":
                self.reduce_size = lhs_shape[-1]
                j = 1  # the reduce axis takes no part in the offsets (bcast.cc:50-54)
            stride_l = stride_r = 1
            lo, ro = [0], [0]
            while j < max_ndim:  # axes from back to front (bcast.cc:58-83)
                dl = 1 if lnd - 1 - j < 1 else lhs_shape[lnd - 1 - j]
                dr = 1 if rnd - 1 - j < 1 else rhs_shape[rnd - 1 - j]
                for i in range(1, max(dl, dr)):
                    for k in range(out_len):
                        lo.append(lo[k] + i * (1 if i < dl else 0) * stride_l)
                        ro.append(ro[k] + i * (1 if i < dr else 0) * stride_r)
                out_len *= max(dl, dr)
                stride_l *= dl
                stride_r *= dr
                j += 1
            self.out_len = out_len
            self.lhs_offset = np.asarray(lo, dtype=np.int64)
            self.rhs_offset = np.asarray(ro, dtype=np.int64)
        else:
            self.out_len = self.rhs_len if op == "copy_rhs" else self.lhs_len
            if op == "dot":
                self.reduce_size = lhs_shape[-1]
                self.out_len //= self.reduce_size


def infer_broadcast_shape(op, shp1, shp2):
    """Output feature shape (without the leading dim).  Restates
    python/dgl/_sparse_ops.py:10-60: numpy-style right-aligned broadcasting; for ``dot``
    the last axis is reduced to size 1."""
    shp1, shp2 = tuple(shp1), tuple(shp2)
    if op == "copy_lhs":
        return shp1
    if op == "copy_rhs":
        return shp2
    pad = max(len(shp1), len(shp2))
    a = (1,) * (pad - len(shp1)) + shp1
    b = (1,) * (pad - len(shp2)) + shp2
    for d1, d2 in zip(a, b):
        if d1 != d2 and d1 != 1 and d2 != 1:
            raise ValueError("Feature shapes {} and {} are not valid for broadcasting.".format(shp1, shp2))
    out = tuple(max(d1, d2) for d1, d2 in zip(a, b))
    return out[:-1] + (1,) if op == "dot" else out


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _sfx(dtype, idtype):
    d = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.dtype(dtype)]
    i = {np.dtype(np.int32): "i32", np.dtype(np.int64): "i64"}[np.dtype(idtype)]
    return d + "_" + i


def _prep_feat(x, dtype):
    if x is None:
        return None
    x = np.ascontiguousarray(x, dtype=dtype)
    return x.reshape(x.shape[0], 1) if x.ndim == 1 else x


def _nthreads(n):
    # default: at most 8 threads.  Nearly every caller is a parity test on a graph of a few
    # hundred edges, where waking a 64-256 thread team costs far more than the kernel; the
    # timed CPU baseline (bench.py) passes its thread count explicitly.
    return int(n) if n else max(1, min((os.cpu_count() or 2) // 2, 8))


def _i64(v):
    return ctypes.c_int64(int(v))


# --------------------------------------------------------------------------- #
# g-SpMM
# --------------------------------------------------------------------------- #
def spmm_csr(op, reduce, indptr, indices, eids, ufeat, efeat, nthreads=None, out=None):
    """``out[r] = reduce_{j in row r} op(ufeat[indices[j]], efeat[eid(j)])``.

    Returns ``(out, arg_u, arg_e)`` (args are None for sum / for the unused side), with
    shapes and dtypes as python/dgl/_sparse_ops.py:156-265 produces them.
    """
    assert reduce in ("sum", "max", "min")
    indptr = np.ascontiguousarray(indptr)
    idt = indptr.dtype
    indices = np.ascontiguousarray(indices, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    fdt = (ufeat if ufeat is not None else efeat).dtype
    squeeze = (ufeat is None or ufeat.ndim == 1) and (efeat is None or efeat.ndim == 1)
    u, e = _prep_feat(ufeat, fdt), _prep_feat(efeat, fdt)
    use_l, use_r = op != "copy_rhs", op != "copy_lhs"
    lshape = u.shape if use_l else e.shape
    rshape = e.shape if use_r else u.shape
    bc = BcastOff(op, lshape, rshape)
    n_rows = indptr.shape[0] - 1
    oshape = (n_rows,) + infer_broadcast_shape(op, lshape[1:], rshape[1:])
    if out is None:
        out = np.zeros(oshape, dtype=fdt)
    else:
        # accumulate into the caller's buffer: `out[k] += ...` continues from what it holds, as the
        # reference's kernel does with its pre-zeroed / partially summed output (spmm.h:60-70)
        assert reduce == "sum" and out.shape == oshape and out.dtype == fdt and out.flags.c_contiguous
    assert int(np.prod(oshape[1:])) == bc.out_len
    L = lib()
    sfx = _sfx(fdt, idt)
    lo, ro = _ptr(bc.lhs_offset), _ptr(bc.rhs_offset)
    if reduce == "sum":
        getattr(L, "oracle_spmm_sum_csr_" + sfx)(
            OPS[op], _i64(n_rows), _ptr(indptr), _ptr(indices), _ptr(eids),
            _ptr(u), _ptr(e), _ptr(out), _i64(bc.out_len), _i64(bc.lhs_len),
            _i64(bc.rhs_len), lo, ro, _nthreads(nthreads))
        argu = arge = None
    else:
        argu = np.zeros(oshape, dtype=idt)
        arge = np.zeros(oshape, dtype=idt)
        getattr(L, "oracle_spmm_cmp_csr_" + sfx)(
            OPS[op], 1 if reduce == "max" else 0, _i64(n_rows), _ptr(indptr),
            _ptr(indices), _ptr(eids), _ptr(u), _ptr(e), _ptr(out), _ptr(argu),
            _ptr(arge), _i64(bc.out_len), _i64(bc.lhs_len), _i64(bc.rhs_len), lo, ro,
            _nthreads(nthreads))
        if not use_l:
            argu = None
        if not use_r:
            arge = None
    if squeeze:
        out = out.reshape(-1)
        argu = None if argu is None else argu.reshape(-1)
        arge = None if arge is None else arge.reshape(-1)
    return out, argu, arge


def spmm_coo(op, reduce, row, col, eids, num_dst, ufeat, efeat):
    """COO flavour (row = source ids, col = destination ids)."""
    row = np.ascontiguousarray(row)
    idt = row.dtype
    col = np.ascontiguousarray(col, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    fdt = (ufeat if ufeat is not None else efeat).dtype
    squeeze = (ufeat is None or ufeat.ndim == 1) and (efeat is None or efeat.ndim == 1)
    u, e = _prep_feat(ufeat, fdt), _prep_feat(efeat, fdt)
    use_l, use_r = op != "copy_rhs", op != "copy_lhs"
    lshape = u.shape if use_l else e.shape
    rshape = e.shape if use_r else u.shape
    bc = BcastOff(op, lshape, rshape)
    oshape = (num_dst,) + infer_broadcast_shape(op, lshape[1:], rshape[1:])
    out = np.zeros(oshape, dtype=fdt)
    L = lib()
    sfx = _sfx(fdt, idt)
    lo, ro = _ptr(bc.lhs_offset), _ptr(bc.rhs_offset)
    nnz = row.shape[0]
    if reduce == "sum":
        getattr(L, "oracle_spmm_sum_coo_" + sfx)(
            OPS[op], _i64(nnz), _i64(num_dst), _ptr(row), _ptr(col), _ptr(eids),
            _ptr(u), _ptr(e), _ptr(out), _i64(bc.out_len), _i64(bc.lhs_len),
            _i64(bc.rhs_len), lo, ro)
        argu = arge = None
    else:
        argu = np.zeros(oshape, dtype=idt)
        arge = np.zeros(oshape, dtype=idt)
        getattr(L, "oracle_spmm_cmp_coo_" + sfx)(
            OPS[op], 1 if reduce == "max" else 0, _i64(nnz), _i64(num_dst), _ptr(row),
            _ptr(col), _ptr(eids), _ptr(u), _ptr(e), _ptr(out), _ptr(argu), _ptr(arge),
            _i64(bc.out_len), _i64(bc.lhs_len), _i64(bc.rhs_len), lo, ro)
        if not use_l:
            argu = None
        if not use_r:
            arge = None
    if squeeze:
        out = out.reshape(-1)
        argu = None if argu is None else argu.reshape(-1)
        arge = None if arge is None else arge.reshape(-1)
    return out, argu, arge


def copy_u_sum_csr(indptr, indices, x, nthreads=None, out=None):
    """Vectorisable copy_u+sum (same order as ``spmm_csr('copy_lhs','sum')``); the
    function ``bench.py`` times as the CPU baseline."""
    idt = indptr.dtype
    n_rows = indptr.shape[0] - 1
    dim = int(np.prod(x.shape[1:]))
    if out is None:
        out = np.zeros((n_rows,) + x.shape[1:], dtype=x.dtype)
    else:
        out[...] = 0
    getattr(lib(), "oracle_spmm_copy_u_sum_csr_" + _sfx(x.dtype, idt))(
        _i64(n_rows), _ptr(indptr), _ptr(indices), _ptr(x), _ptr(out), _i64(dim),
        _nthreads(nthreads))
    return out


def llc_bytes():
    """Last-level cache size the reference's tiling uses: sysconf(_SC_LEVEL3_CACHE_SIZE), else
    its compile-time default of 32 MiB... (spmm_blocking_libxsmm.h:44-52)."""
    try:
        v = os.sysconf("SC_LEVEL3_CACHE_SIZE")
        if v and v > 0:
            return int(v)
    except (ValueError, OSError):
        pass
    return 32 << 20


def copy_u_sum_csr_blocked(indptr, indices, x, nthreads=None, out=None, llc=None, num_cols=None):
    """copy_u+sum organised like the reference's libxsmm path (K-blocked, dynamic M blocks,
    per-call re-tiling; src/array/cpu/spmm_blocking_libxsmm.h:432-557) with a plain vectorised
    row add in place of the JIT kernel.  Column ids must ascend inside a row (the reference
    relies on it too).  Returns ``(out, info)``, info = M/K block sizes and counts."""
    idt = indptr.dtype
    n_rows = indptr.shape[0] - 1
    dim = int(np.prod(x.shape[1:]))
    if out is None:
        out = np.zeros((n_rows,) + x.shape[1:], dtype=x.dtype)
    else:
        out[...] = 0
    info = (ctypes.c_int64 * 4)()
    fn = getattr(lib(), "oracle_spmm_copy_u_sum_csr_blocked_" + _sfx(x.dtype, idt))
    fn.restype = ctypes.c_int
    rc = fn(_i64(n_rows), _i64(x.shape[0] if num_cols is None else num_cols), _ptr(indptr), _ptr(indices),
            _ptr(x), _ptr(out), _i64(dim), _nthreads(nthreads), _i64(llc or llc_bytes()), info)
    if rc != 0:
        raise MemoryError("oracle_spmm_copy_u_sum_csr_blocked: scratch allocation failed")
    return out, {"M_block": int(info[0]), "K_block": int(info[1]), "num_M_blocks": int(info[2]),
                 "num_K_blocks": int(info[3])}


# --------------------------------------------------------------------------- #
# g-SDDMM
# --------------------------------------------------------------------------- #
def _sddmm_common(op, lhs, rhs):
    fdt = (lhs if lhs is not None else rhs).dtype
    squeeze = (lhs is None or lhs.ndim == 1) and (rhs is None or rhs.ndim == 1)
    l, r = _prep_feat(lhs, fdt), _prep_feat(rhs, fdt)
    use_l, use_r = op != "copy_rhs", op != "copy_lhs"
    lshape = l.shape if use_l else r.shape
    rshape = r.shape if use_r else l.shape
    bc = BcastOff(op, lshape, rshape)
    feat_shape = infer_broadcast_shape(op, lshape[1:], rshape[1:])
    return fdt, squeeze, l, r, bc, feat_shape


def sddmm_coo(op, row, col, eids, lhs, rhs, lhs_target="u", rhs_target="v", nthreads=None):
    """``out[eid] = op(lhs[sel(lhs_target)], rhs[sel(rhs_target)])`` over COO edges
    (row = source ids, col = destination ids)."""
    row = np.ascontiguousarray(row)
    idt = row.dtype
    col = np.ascontiguousarray(col, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    fdt, squeeze, l, r, bc, feat_shape = _sddmm_common(op, lhs, rhs)
    nnz = row.shape[0]
    out = np.zeros((nnz,) + feat_shape, dtype=fdt)
    getattr(lib(), "oracle_sddmm_coo_" + _sfx(fdt, idt))(
        OPS[op], _i64(nnz), _ptr(row), _ptr(col), _ptr(eids), _ptr(l), _ptr(r),
        _ptr(out), _i64(bc.out_len), _i64(bc.lhs_len), _i64(bc.rhs_len),
        _i64(bc.reduce_size), _ptr(bc.lhs_offset), _ptr(bc.rhs_offset),
        TARGETS[lhs_target], TARGETS[rhs_target], _nthreads(nthreads))
    return out.reshape(-1) if squeeze else out


def sddmm_csr(op, indptr, indices, eids, lhs, rhs, lhs_target="u", rhs_target="v", nthreads=None):
    """CSR flavour; rows are SOURCE nodes here (see kernels_impl.inc note)."""
    indptr = np.ascontiguousarray(indptr)
    idt = indptr.dtype
    indices = np.ascontiguousarray(indices, dtype=idt)
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    fdt, squeeze, l, r, bc, feat_shape = _sddmm_common(op, lhs, rhs)
    nnz = indices.shape[0]
    out = np.zeros((nnz,) + feat_shape, dtype=fdt)
    getattr(lib(), "oracle_sddmm_csr_" + _sfx(fdt, idt))(
        OPS[op], _i64(indptr.shape[0] - 1), _ptr(indptr), _ptr(indices), _ptr(eids),
        _ptr(l), _ptr(r), _ptr(out), _i64(bc.out_len), _i64(bc.lhs_len),
        _i64(bc.rhs_len), _i64(bc.reduce_size), _ptr(bc.lhs_offset),
        _ptr(bc.rhs_offset), TARGETS[lhs_target], TARGETS[rhs_target],
        _nthreads(nthreads))
    return out.reshape(-1) if squeeze else out


# --------------------------------------------------------------------------- #
# edge softmax
# --------------------------------------------------------------------------- #
def edge_softmax_fwd(indptr, eids, score, nthreads=None):
    indptr = np.ascontiguousarray(indptr)
    idt = indptr.dtype
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    s = np.ascontiguousarray(score)
    dim = int(np.prod(s.shape[1:])) if s.ndim > 1 else 1
    out = np.zeros_like(s)
    getattr(lib(), "oracle_edge_softmax_fwd_" + _sfx(s.dtype, idt))(
        _i64(indptr.shape[0] - 1), _ptr(indptr), _ptr(eids), _ptr(s), _ptr(out),
        _i64(dim), _nthreads(nthreads))
    return out


def edge_softmax_bwd(indptr, eids, out, sds, nthreads=None):
    indptr = np.ascontiguousarray(indptr)
    idt = indptr.dtype
    eids = None if eids is None else np.ascontiguousarray(eids, dtype=idt)
    o = np.ascontiguousarray(out)
    s = np.ascontiguousarray(sds, dtype=o.dtype)
    dim = int(np.prod(o.shape[1:])) if o.ndim > 1 else 1
    back = np.zeros_like(o)
    getattr(lib(), "oracle_edge_softmax_bwd_" + _sfx(o.dtype, idt))(
        _i64(indptr.shape[0] - 1), _ptr(indptr), _ptr(eids), _ptr(o), _ptr(s),
        _ptr(back), _i64(dim), _nthreads(nthreads))
    return back


# --------------------------------------------------------------------------- #
# segment reduce / scatter add (src/array/cpu/segment_reduce.h)
# --------------------------------------------------------------------------- #
def segment_reduce(reduce, feat, offsets, nthreads=None):
    """Returns ``(out, arg)``; ``arg`` is None for sum.  Shapes as
    python/dgl/_sparse_ops.py:641-676 (_segment_reduce) produces them."""
    assert reduce in ("sum", "max", "min")
    offsets = np.ascontiguousarray(offsets)
    idt = offsets.dtype
    f = np.ascontiguousarray(feat)
    n = offsets.shape[0] - 1
    dim = int(np.prod(f.shape[1:])) if f.ndim > 1 else 1
    out = np.zeros((n,) + f.shape[1:], dtype=f.dtype)
    arg = None if reduce == "sum" else np.zeros(out.shape, dtype=idt)
    getattr(lib(), "oracle_segment_reduce_" + _sfx(f.dtype, idt))(
        {"sum": -1, "max": 1, "min": 0}[reduce], _i64(n), _ptr(offsets), _ptr(f), _ptr(out),
        _ptr(arg), _i64(dim), _nthreads(nthreads))
    return out, arg


def scatter_add(feat, idx, out):
    """In place: ``out[idx[i]] += feat[i]`` for i ascending."""
    idx = np.ascontiguousarray(idx)
    f = np.ascontiguousarray(feat, dtype=out.dtype)
    assert out.flags.c_contiguous
    dim = int(np.prod(out.shape[1:])) if out.ndim > 1 else 1
    getattr(lib(), "oracle_scatter_add_" + _sfx(out.dtype, idx.dtype))(
        _i64(f.shape[0]), _ptr(idx), _ptr(f), _ptr(out), _i64(dim))
    return out


def backward_segment_cmp(feat, arg, out):
    """In place: ``out[arg[i, k], k] = feat[i, k]`` where ``arg >= 0``."""
    arg = np.ascontiguousarray(arg)
    f = np.ascontiguousarray(feat, dtype=out.dtype)
    assert out.flags.c_contiguous
    dim = int(np.prod(out.shape[1:])) if out.ndim > 1 else 1
    getattr(lib(), "oracle_bwd_segment_cmp_" + _sfx(out.dtype, arg.dtype))(
        _i64(f.shape[0]), _ptr(arg), _ptr(f), _ptr(out), _i64(dim))
    return out


# --------------------------------------------------------------------------- #
# heterograph SpMM: the per-relation loop of SpMMCsrHetero (src/array/cpu/spmm.cc:45-150)
# --------------------------------------------------------------------------- #
def spmm_csr_hetero(op, reduce, rels, num_nodes, ufeats, efeats):
    """Restates SpMMCsrHetero<kDGLCPU>: sum — every relation adds into the (zeroed) output of
    its destination type (spmm.cc:55-69); max / min — outputs start at -inf / +inf, the
    node- / edge-type trackers at -1 (spmm.cc:72-98), then the relations run IN ORDER through
    SpMMCmpCsrHetero (spmm.h:341-408), whose compare is strict and seeded from the current
    output: an earlier relation keeps ties.  Same signature / return as oracle.ref.spmm_csr_hetero."""
    use_u, use_e = op != "copy_rhs", op != "copy_lhs"
    n_nt = len(num_nodes)
    outs = [None] * n_nt
    au, ae, aut, aet = ([None] * n_nt for _ in range(4))
    for et, r in enumerate(rels):
        s, d = r["src"], r["dst"]
        u = ufeats[s] if use_u else None
        e = efeats[et] if use_e else None
        if reduce == "sum":
            # every relation's edges are added into the SAME running output, in relation order
            # (spmm.cc:58-68): the rounding differs from summing per-relation totals
            if outs[d] is None:
                probe = spmm_csr(op, reduce, r["indptr"][:1].repeat(len(r["indptr"])), r["indices"][:0],
                                 None if r["eids"] is None else r["eids"][:0], u,
                                 None if e is None else e[:0])[0]
                outs[d] = np.zeros(probe.shape if probe.ndim > 1 else (probe.shape[0], 1), dtype=probe.dtype)
            spmm_csr(op, reduce, r["indptr"], r["indices"], r["eids"],
                     None if u is None else (u if u.ndim > 1 else u.reshape(-1, 1)),
                     None if e is None else (e if e.ndim > 1 else e.reshape(-1, 1)), out=outs[d])
            continue
        o, cu, ce = spmm_csr(op, reduce, r["indptr"], r["indices"], r["eids"], u, e)
        if o.ndim == 1:
            o = o.reshape(-1, 1)
            cu = None if cu is None else cu.reshape(-1, 1)
            ce = None if ce is None else ce.reshape(-1, 1)
        if outs[d] is None:
            outs[d] = np.full(o.shape, -np.inf if reduce == "max" else np.inf, dtype=o.dtype)
            idt = np.asarray(r["indptr"]).dtype
            if use_u:
                au[d] = np.zeros(o.shape, dtype=idt)
                aut[d] = np.full(o.shape, -1, dtype=idt)
            if use_e:
                ae[d] = np.zeros(o.shape, dtype=idt)
                aet[d] = np.full(o.shape, -1, dtype=idt)
        better = (outs[d] < o) if reduce == "max" else (outs[d] > o)
        outs[d] = np.where(better, o, outs[d])
        if use_u:
            au[d] = np.where(better, cu, au[d])
            aut[d] = np.where(better, s, aut[d]).astype(au[d].dtype)
        if use_e:
            ae[d] = np.where(better, ce, ae[d])
            aet[d] = np.where(better, et, aet[d]).astype(ae[d].dtype)
    return outs, au, ae, aut, aet
