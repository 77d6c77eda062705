"""Torch-tensor front end of the graph-free C seam (include/dgl_amd.h, `dgla_*`).

Everything here is plumbing: it turns torch tensors into (pointer, shape) pairs, picks the
current HIP stream (reference: kernels are queued on the current PyTorch stream,
tests/python/pytorch/test_ffi-stream.py:45-64) and calls the shared library.  No arithmetic
happens in Python.
"""
import ctypes

import torch

from . import _lib
from ._lib import COO, CSR, LIB, Tensor, check_call

_DTYPES = {torch.float32: 0, torch.float64: 1, torch.float16: 2, torch.bfloat16: 3}
TARGETS = {"u": 0, "e": 1, "v": 2}


def _require_gpu(t):
    if not t.is_cuda:
        raise _lib.DGLAMDError(
            "dgl_amd kernels run on a ROCm GPU; got a tensor on %s (no CPU fallback)" % t.device)


def _tensor(t, keep):
    """dgla_tensor for a contiguous torch tensor (1-D is viewed as (n, 1), the reference's
    unsqueeze in python/dgl/_sparse_ops.py:208-217)."""
    if t is None:
        return Tensor(None, 0, None)
    if type(t) is not torch.Tensor:
        from .edge_order import reject_tagged
        reject_tagged(t)
    _require_gpu(t)
    if not t.is_contiguous():
        raise _lib.DGLAMDError("feature tensors must be contiguous")
    shape = tuple(t.shape) if t.dim() > 1 else (t.shape[0], 1)
    arr = (ctypes.c_int64 * len(shape))(*shape)
    keep.append(arr)
    return Tensor(t.data_ptr(), len(shape), arr)


def _idbits(t):
    if t.dtype == torch.int32:
        return 32
    if t.dtype == torch.int64:
        return 64
    raise _lib.DGLAMDError("index tensors must be int32 or int64")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def make_csr(indptr, indices, eids, num_cols):
    _require_gpu(indptr)
    c = CSR(indptr.shape[0] - 1, int(num_cols), indices.shape[0], _idbits(indptr),
            indptr.data_ptr(), _ptr(indices), _ptr(eids))
    c._keep = (indptr, indices, eids)  # the struct borrows the pointers: keep the tensors alive with it
    return c


def make_coo(row, col, eids, num_src, num_dst):
    _require_gpu(row)
    c = COO(int(num_src), int(num_dst), row.shape[0], _idbits(row), _ptr(row), _ptr(col), _ptr(eids))
    c._keep = (row, col, eids)
    return c


def spmm_csr_workspace_bytes(op, reduce, csr, dtype, ufeat, efeat, out):
    keep = []
    tu, te, to = _tensor(ufeat, keep), _tensor(efeat, keep), _tensor(out, keep)
    return LIB.dgla_spmm_csr_workspace_bytes(op.encode(), reduce.encode(), ctypes.byref(csr),
                                             _DTYPES[dtype], ctypes.byref(tu), ctypes.byref(te),
                                             ctypes.byref(to))


def spmm_csr(op, reduce, csr, ufeat, efeat, out, arg_u=None, arg_e=None, workspace=None,
             accumulate=False, plan_valid=False, mean=False, split_keep=False, split_valid=False,
             prepare_only=False):
    """out = g-SpMM over `csr` (rows = destination nodes).  `workspace` is a uint8 tensor of
    at least spmm_csr_workspace_bytes(); it also caches the merge plan between calls.
    `split_keep`: `ufeat` is static (the caller will pass the same unchanged tensor again), so
    the split-row copy is made whatever the locality probe says; `split_valid`: the workspace
    still holds that copy of this very `ufeat` — only for callers that own the tensor;
    `prepare_only` (with `split_keep`): plan + side copy only, for the PRODUCER of `ufeat`."""
    keep = []
    tu, te, to = _tensor(ufeat, keep), _tensor(efeat, keep), _tensor(out, keep)
    flags = (_lib.DGLA_ACCUMULATE if accumulate else 0) | (_lib.DGLA_PLAN_VALID if plan_valid else 0) | \
        (_lib.DGLA_MEAN if mean else 0) | (_lib.DGLA_SPLIT_VALID if split_valid else 0) | \
        (_lib.DGLA_SPLIT_KEEP if split_keep else 0) | (_lib.DGLA_PREPARE_ONLY if prepare_only else 0)
    check_call(LIB.dgla_spmm_csr(
        op.encode(), reduce.encode(), ctypes.byref(csr), _DTYPES[out.dtype], ctypes.byref(tu),
        ctypes.byref(te), ctypes.byref(to), _ptr(arg_u), _ptr(arg_e), _ptr(workspace),
        0 if workspace is None else workspace.numel() * workspace.element_size(), flags,
        _stream(out)))


def pointer_table(tensors, device):
    """Device array of the tensors' data pointers (int64), as dgla_spmm_csr_stacked wants."""
    return torch.tensor([0 if t is None else t.data_ptr() for t in tensors], dtype=torch.int64,
                        device=device)


def spmm_csr_stacked_workspace_bytes(op, csr, ufeat0, efeat0, out):
    keep = []
    tu, te, to = _tensor(ufeat0, keep), _tensor(efeat0, keep), _tensor(out, keep)
    return LIB.dgla_spmm_csr_stacked_workspace_bytes(op.encode(), ctypes.byref(csr),
                                                     _DTYPES[out.dtype], ctypes.byref(tu),
                                                     ctypes.byref(te), ctypes.byref(to))


def spmm_csr_stacked(op, csr, rel, ufeats, efeats, out, workspace, u_table=None, e_table=None,
                     accumulate=False, plan_valid=False):
    """Fused sum over several relations (SpMMCsrHetero in one launch).  `csr` is the stacked
    matrix, `rel` the uint8 relation id per stacked edge, `ufeats` / `efeats` lists with one
    tensor per relation (or None when the operator does not read that side)."""
    keep = []
    u0 = None if ufeats is None else ufeats[0]
    e0 = None if efeats is None else efeats[0]
    tu, te, to = _tensor(u0, keep), _tensor(e0, keep), _tensor(out, keep)
    if u_table is None and ufeats is not None:
        u_table = pointer_table(ufeats, out.device)
    if e_table is None and efeats is not None:
        e_table = pointer_table(efeats, out.device)
    n_rel = len(ufeats if ufeats is not None else efeats)
    flags = (_lib.DGLA_ACCUMULATE if accumulate else 0) | (_lib.DGLA_PLAN_VALID if plan_valid else 0)
    check_call(LIB.dgla_spmm_csr_stacked(
        op.encode(), ctypes.byref(csr), rel.data_ptr(), n_rel, _DTYPES[out.dtype], ctypes.byref(tu),
        ctypes.byref(te), _ptr(u_table), _ptr(e_table), ctypes.byref(to), _ptr(workspace),
        0 if workspace is None else workspace.numel() * workspace.element_size(), flags,
        _stream(out)))
    return u_table, e_table


def spmm_coo(op, reduce, coo, ufeat, efeat, out, arg_u=None, arg_e=None):
    keep = []
    tu, te, to = _tensor(ufeat, keep), _tensor(efeat, keep), _tensor(out, keep)
    check_call(LIB.dgla_spmm_coo(op.encode(), reduce.encode(), ctypes.byref(coo),
                                 _DTYPES[out.dtype], ctypes.byref(tu), ctypes.byref(te),
                                 ctypes.byref(to), _ptr(arg_u), _ptr(arg_e), _stream(out)))


def sddmm_coo(op, coo, lhs, rhs, out, lhs_target, rhs_target):
    keep = []
    tl, tr, to = _tensor(lhs, keep), _tensor(rhs, keep), _tensor(out, keep)
    check_call(LIB.dgla_sddmm_coo(op.encode(), ctypes.byref(coo), _DTYPES[out.dtype],
                                  ctypes.byref(tl), ctypes.byref(tr), ctypes.byref(to),
                                  lhs_target, rhs_target, _stream(out)))


def sddmm_csr(op, csr, lhs, rhs, out, lhs_target, rhs_target):
    keep = []
    tl, tr, to = _tensor(lhs, keep), _tensor(rhs, keep), _tensor(out, keep)
    check_call(LIB.dgla_sddmm_csr(op.encode(), ctypes.byref(csr), _DTYPES[out.dtype],
                                  ctypes.byref(tl), ctypes.byref(tr), ctypes.byref(to),
                                  lhs_target, rhs_target, _stream(out)))


def edge_softmax_workspace_bytes(csr, dtype, dim):
    return LIB.dgla_edge_softmax_workspace_bytes(ctypes.byref(csr), _DTYPES[dtype], int(dim))


def _ws_args(workspace, plan_valid):
    return (_ptr(workspace), 0 if workspace is None else workspace.numel() * workspace.element_size(),
            _lib.DGLA_PLAN_VALID if plan_valid else 0)


def edge_softmax_forward(csr, score, out, workspace=None, plan_valid=False, out_position=False):
    """`workspace` (uint8 tensor of edge_softmax_workspace_bytes) selects the degree-balanced
    merge-path kernels; None runs the scratch-free lane-group kernel.  `out_position` (merge-path only):
    scores are read through the CSR's edge-id map, `out` is written in position order
    (DGLA_ESM_OUT_POSITION)."""
    keep = []
    ts, to = _tensor(score, keep), _tensor(out, keep)
    wp, wn, fl = _ws_args(workspace, plan_valid)
    if out_position:
        fl |= _lib.DGLA_ESM_OUT_POSITION
    check_call(LIB.dgla_edge_softmax_forward(ctypes.byref(csr), _DTYPES[out.dtype],
                                             ctypes.byref(ts), ctypes.byref(to), wp, wn, fl, _stream(out)))


def edge_softmax_backward(csr, out, sds, back, workspace=None, plan_valid=False, sds_is_grad=False,
                          out_position=False):
    """`sds_is_grad` (merge-path kernels only): `sds` is the upstream gradient itself, the product with `out` is
    formed inside the kernel (DGLA_ESM_B_IS_GRAD).  `out_position` (merge-path only): `out` is read and `back` written
    in the CSR's position order, `sds` is read through the edge-id map (DGLA_ESM_OUT_POSITION)."""
    keep = []
    to, ts, tb = _tensor(out, keep), _tensor(sds, keep), _tensor(back, keep)
    wp, wn, fl = _ws_args(workspace, plan_valid)
    if sds_is_grad:
        fl |= _lib.DGLA_ESM_B_IS_GRAD
    if out_position:
        fl |= _lib.DGLA_ESM_OUT_POSITION
    check_call(LIB.dgla_edge_softmax_backward(ctypes.byref(csr), _DTYPES[out.dtype],
                                              ctypes.byref(to), ctypes.byref(ts),
                                              ctypes.byref(tb), wp, wn, fl, _stream(back)))


def gat_attention_workspace_bytes(csc, heads, dim):
    return int(LIB.dgla_gat_attention_workspace_bytes(ctypes.byref(csc), int(heads), int(dim)))


def gat_attention_forward(csc, ft, el, er, slope, out, mz, workspace):
    """out[v] = sum_u softmax_v(leaky_relu(el[u] + er[v])) ft[u] per head in one pass (dgla_gat_attention_forward);
    `mz` (N_dst, H, 2) fp32 receives each row's softmax maximum and normaliser for the backward."""
    keep = []
    tf, tl, tr, to = (_tensor(t, keep) for t in (ft, el, er, out))
    _require_gpu(mz)
    check_call(LIB.dgla_gat_attention_forward(ctypes.byref(csc), _DTYPES[ft.dtype], ctypes.byref(tf), ctypes.byref(tl),
                                              ctypes.byref(tr), float(slope), ctypes.byref(to), mz.data_ptr(),
                                              _ptr(workspace), 0 if workspace is None else workspace.numel(),
                                              _stream(out)))


def gat_attention_backward(csc, csr, ft, el, er, out, mz, dout, slope, d_ft, d_el, d_er, workspace):
    """Gradients of dgla_gat_attention_forward; `csr` = the out-edge CSR (rows = source nodes) of the same graph."""
    keep = []
    ts = [_tensor(t, keep) for t in (ft, el, er, out, dout, d_ft, d_el, d_er)]
    check_call(LIB.dgla_gat_attention_backward(ctypes.byref(csc), ctypes.byref(csr), _DTYPES[ft.dtype],
                                               ctypes.byref(ts[0]), ctypes.byref(ts[1]), ctypes.byref(ts[2]),
                                               ctypes.byref(ts[3]), mz.data_ptr(), ctypes.byref(ts[4]), float(slope),
                                               ctypes.byref(ts[5]), ctypes.byref(ts[6]), ctypes.byref(ts[7]),
                                               _ptr(workspace), 0 if workspace is None else workspace.numel(),
                                               _stream(d_ft)))


def stream_copy(dst, src):
    _require_gpu(dst)
    check_call(LIB.dgla_stream_copy(dst.data_ptr(), src.data_ptr(),
                                    src.numel() * src.element_size(), _stream(dst)))


def stream_copy_variant(dst, src, variant):
    _require_gpu(dst)
    check_call(LIB.dgla_stream_copy_variant(dst.data_ptr(), src.data_ptr(),
                                            src.numel() * src.element_size(), int(variant),
                                            _stream(dst)))


def set_profile_events(before, after):
    """Record two torch.cuda.Event objects around the merge kernel of the next spmm_csr calls
    of this thread (None, None disables).  The events must have been recorded once already so
    that their HIP handles exist."""
    if before is None or after is None:
        LIB.dgla_spmm_set_profile_events(None, None)
    else:
        LIB.dgla_spmm_set_profile_events(before.cuda_event, after.cuda_event)


(TUNE_XCD, TUNE_SPLIT, TUNE_GLDS, TUNE_SPLIT_FORCE, TUNE_MM_F32, TUNE_MM_X3, TUNE_NO_GATE, TUNE_NO_STAGE_W) = (1, 8, 16, 64, 128, 2048, 4096, 8192)  # include/dgl_amd.h DGLA_TUNE_*


def set_tuning(flags):
    """Process-wide tuning bits (include/dgl_amd.h: DGLA_TUNE_*).  The SpMM bits XCD, SPLIT*
    never change result bits (SPLIT changes the workspace layout: plans do not survive a change of
    it).  The matrix-multiply bits do change bits: DGLA_TUNE_GLDS contracts fp32 k in a
    permuted order and DGLA_TUNE_MM_F32 selects the exact fp32 MFMA instead of the default
    split-operand kernels (same fp32-level error bound, different low-order bits); DGLA_TUNE_MM_X3 keeps the
    weights-stationary fp32 forward on three bf16 terms instead of two scaled fp16 terms."""
    check_call(LIB.dgla_set_tuning(int(flags)))
    from . import sparse_kernels
    sparse_kernels._tuning_epoch[0] = int(flags)  # scratch sizes remembered per relation depend on the bits


def get_tuning():
    return int(LIB.dgla_get_tuning())


def narrow_reduce_calls():
    """dgla_spmm_csr calls served by the narrow-feature copy_e kernels so far (dgla_narrow_reduce_calls)."""
    return int(LIB.dgla_narrow_reduce_calls())


def segment_reduce(reduce, feat, offsets, out, arg=None, workspace=None, plan_valid=False):
    """out[i] = reduce(feat[offsets[i]:offsets[i+1]]) (dgla_segment_reduce).  `workspace` may
    be None (stream-ordered scratch inside the call) or a uint8 tensor of
    segment_reduce_workspace_bytes() that also caches the merge plan of `offsets`."""
    keep = []
    tf, to = _tensor(feat, keep), _tensor(out, keep)
    _require_gpu(offsets)
    check_call(LIB.dgla_segment_reduce(
        reduce.encode(), _idbits(offsets), _DTYPES[out.dtype], ctypes.byref(tf),
        offsets.data_ptr(), offsets.shape[0] - 1, ctypes.byref(to), _ptr(arg), _ptr(workspace),
        0 if workspace is None else workspace.numel() * workspace.element_size(),
        _lib.DGLA_PLAN_VALID if plan_valid else 0, _stream(out)))


def segment_reduce_workspace_bytes(reduce, feat, offsets, out):
    keep = []
    tf, to = _tensor(feat, keep), _tensor(out, keep)
    return LIB.dgla_segment_reduce_workspace_bytes(reduce.encode(), _idbits(offsets),
                                                   _DTYPES[out.dtype], ctypes.byref(tf),
                                                   offsets.shape[0] - 1, ctypes.byref(to))


def scatter_add(feat, idx, out):
    """out[idx[i]] += feat[i] (dgla_scatter_add; `out` is not zeroed)."""
    keep = []
    tf, to = _tensor(feat, keep), _tensor(out, keep)
    _require_gpu(idx)
    check_call(LIB.dgla_scatter_add(_idbits(idx), _DTYPES[out.dtype], ctypes.byref(tf),
                                    idx.data_ptr(), ctypes.byref(to), _stream(out)))


def backward_segment_cmp(feat, arg, out):
    """out[arg[i, k], k] = feat[i, k] where arg >= 0 (dgla_backward_segment_cmp)."""
    keep = []
    tf, to = _tensor(feat, keep), _tensor(out, keep)
    _require_gpu(arg)
    check_call(LIB.dgla_backward_segment_cmp(_idbits(arg), _DTYPES[out.dtype], ctypes.byref(tf),
                                             arg.data_ptr(), ctypes.byref(to), _stream(out)))


def spmm_cmp_backward(dz, arg, out, other=None, arg_other=None, other_group=1, atomic=True):
    """out[arg[i, k], k] (+)= dz[i, k] * other[arg_other[i, k], (k // other_group) % row_len(other)]
    where arg >= 0 (dgla_spmm_cmp_backward; `out` is not zeroed)."""
    keep = []
    tz, to = _tensor(dz, keep), _tensor(out, keep)
    _require_gpu(arg)
    tother = _tensor(other, keep) if other is not None else None
    check_call(LIB.dgla_spmm_cmp_backward(_idbits(arg), _DTYPES[out.dtype], ctypes.byref(tz), arg.data_ptr(),
                                          ctypes.byref(tother) if tother is not None else None,
                                          arg_other.data_ptr() if arg_other is not None else None, int(other_group),
                                          ctypes.byref(to), 1 if atomic else 0, _stream(out)))
    return out


def spmm_cmp_mask_words(dtype, feat_len):
    return int(LIB.dgla_spmm_cmp_mask_words(_DTYPES[dtype], int(feat_len)))


def spmm_cmp_mask_bytes(dtype, num_rows, nnz, feat_len):
    """Size of the `mask` buffer of :func:`spmm_cmp_mask` (edge words + the pass's partial sums)."""
    return int(LIB.dgla_spmm_cmp_mask_bytes(_DTYPES[dtype], int(num_rows), int(nnz), int(feat_len)))


CMP_MASK_DEFER, CMP_MASK_FINISH = 0x100, 0x200   # include/dgl_amd.h DGLA_CMP_MASK_*


def spmm_cmp_mask(csr, arg, dz, mask, dx, by_edge=False, mode=0):
    """Winner bits of a max / min g-SpMM over the FORWARD matrix `csr` (dgla_spmm_cmp_mask): for every edge in
    position order, one bit per output column; elements no edge claims go to ``dx[arg]`` here (``dx`` zeroed by
    the caller).  ``mode=CMP_MASK_DEFER``: ``dx`` is not touched (the masked g-SpMM then stores instead of accumulating);
    the same call with ``mode=CMP_MASK_FINISH`` behind the g-SpMM adds what was kept aside."""
    keep = []
    tz, tx = _tensor(dz, keep), _tensor(dx, keep)
    _require_gpu(arg)
    check_call(LIB.dgla_spmm_cmp_mask(ctypes.byref(csr), _DTYPES[dz.dtype], arg.data_ptr(), (1 if by_edge else 0) | int(mode),
                                      ctypes.byref(tz), _ptr(mask), ctypes.byref(tx), _stream(dx)))


def spmm_csr_masked_workspace_bytes(csr, ufeat, out):
    keep = []
    tu, to = _tensor(ufeat, keep), _tensor(out, keep)
    return LIB.dgla_spmm_csr_masked_workspace_bytes(ctypes.byref(csr), _DTYPES[out.dtype], ctypes.byref(tu),
                                                    ctypes.byref(to))


def spmm_csr_masked(csr, ufeat, mask, out, workspace=None, accumulate=False, plan_valid=False):
    """out (+)= sum over the edges of every row of `csr` of ufeat[col] gated per (edge, column) by `mask`, which is
    indexed through the CSR's edge-id map (dgla_spmm_csr_masked)."""
    keep = []
    tu, to = _tensor(ufeat, keep), _tensor(out, keep)
    flags = (_lib.DGLA_ACCUMULATE if accumulate else 0) | (_lib.DGLA_PLAN_VALID if plan_valid else 0)
    check_call(LIB.dgla_spmm_csr_masked(ctypes.byref(csr), _DTYPES[out.dtype], ctypes.byref(tu), _ptr(mask),
                                        ctypes.byref(to), _ptr(workspace),
                                        0 if workspace is None else workspace.numel() * workspace.element_size(), flags,
                                        _stream(out)))


def _mm_check(*ts):
    for t in ts:
        if t is None:
            continue
        _require_gpu(t)
        if not t.is_contiguous():
            raise _lib.DGLAMDError("matrices handed to segment_mm / gather_mm must be contiguous")


def segment_mm(a, b, c, seglen, b_trans=False, row_index=None):
    """c[rows of r] = a[rows of r] @ b[r]  (or @ b[r].T when b_trans); `seglen` may be a CPU or
    GPU int32/int64 tensor (dgla_segment_mm).  With `row_index` (int64, GPU) the logical row r
    lives at physical row row_index[r] of both a and c (dgla_segment_mm_indexed)."""
    _mm_check(a, b, c, row_index)
    if row_index is not None and row_index.dtype != torch.int64:
        raise _lib.DGLAMDError("row_index must be int64")
    k = a.shape[1]
    n = c.shape[1]
    check_call(LIB.dgla_segment_mm_indexed(
        _idbits(seglen), _DTYPES[a.dtype], a.data_ptr(), b.data_ptr(), c.data_ptr(), seglen.data_ptr(),
        0 if seglen.is_cuda else 1, _ptr(row_index), a.shape[0], seglen.shape[0], k, n,
        1 if b_trans else 0, None, 0, _stream(a)))


def segment_mm_backward_b(a, dc, db, seglen, row_index=None):
    """db[r] = a[rows of r].T @ dc[rows of r] (dgla_segment_mm_backward_b[_indexed])."""
    _mm_check(a, dc, db, row_index)
    if row_index is not None and row_index.dtype != torch.int64:
        raise _lib.DGLAMDError("row_index must be int64")
    check_call(LIB.dgla_segment_mm_backward_b_indexed(
        _idbits(seglen), _DTYPES[a.dtype], a.data_ptr(), dc.data_ptr(), db.data_ptr(),
        seglen.data_ptr(), 0 if seglen.is_cuda else 1, _ptr(row_index), a.shape[0], seglen.shape[0],
        a.shape[1], dc.shape[1], None, 0, _stream(a)))


def segment_mm_backward_b_last_route():
    """(fell_back, listed_elements) of the most recent fp32 two-term weight-gradient launch
    (dgla_segment_mm_backward_b_last_route; synchronises)."""
    fb, n = ctypes.c_uint32(0), ctypes.c_uint32(0)
    check_call(LIB.dgla_segment_mm_backward_b_last_route(ctypes.byref(fb), ctypes.byref(n)))
    return int(fb.value), int(n.value)


def gather_mm(a, b, c, idx_a=None, idx_b=None, idx_c=None):
    """c[idx_c[i]] = a[idx_a[i]] @ b[idx_b[i]] (dgla_gather_mm); absent index = identity."""
    _mm_check(a, b, c, idx_a, idx_b, idx_c)
    ref = next(t for t in (idx_a, idx_b, idx_c) if t is not None)
    rows = ref.shape[0]
    check_call(LIB.dgla_gather_mm(_idbits(ref), _DTYPES[a.dtype], a.data_ptr(), b.data_ptr(),
                                  c.data_ptr(), _ptr(idx_a), _ptr(idx_b), _ptr(idx_c), rows,
                                  a.shape[1], b.shape[-1], _stream(a)))


def coo_to_csr(row, col, eids, num_rows, num_minor=0):
    """(indptr, indices, eids_out) of the CSR that compresses `row` (dgla_coo_to_csr): stable,
    so COO order is kept inside every row; `eids_out[i]` = edge id of CSR position i.  `num_minor`
    (every `col` id is below it) lets int64 graphs take the packed 32-bit sort."""
    _require_gpu(row)
    bits = _idbits(row)
    nnz = row.shape[0]
    indptr = torch.empty(num_rows + 1, dtype=row.dtype, device=row.device)
    indices = torch.empty(nnz, dtype=row.dtype, device=row.device)
    eids_out = torch.empty(nnz, dtype=row.dtype, device=row.device)
    check_call(LIB.dgla_coo_to_csr_bounded(bits, int(num_rows), int(num_minor), nnz, _ptr(row), _ptr(col),
                                           _ptr(eids), indptr.data_ptr(), _ptr(indices), _ptr(eids_out), None, 0,
                                           _stream(row)))
    return indptr, indices, eids_out


def sample_neighbors(csr, seeds, fanout, replace=False, rng_seed=0):
    """Uniform in-neighbour sampling over the in-edge CSR `csr` (dgla_sample_neighbors):
    returns ``(indptr, src, eids)`` — a CSR over the seeds with GLOBAL source / edge ids."""
    _require_gpu(seeds)
    rng_seed = int(rng_seed) & 0xFFFFFFFFFFFFFFFF
    n = seeds.shape[0]
    dev, dt = seeds.device, seeds.dtype
    indptr = torch.empty(n + 1, dtype=dt, device=dev)
    st = _stream(seeds)
    if fanout < 0:  # all neighbours: sizes first
        check_call(LIB.dgla_sample_neighbors(ctypes.byref(csr), seeds.data_ptr(), n, fanout, 0, rng_seed,
                                             indptr.data_ptr(), None, None, None, 0, st))
        cap = int(indptr[-1])
    else:
        cap = n * fanout
    src = torch.empty(cap, dtype=dt, device=dev)
    eids = torch.empty(cap, dtype=dt, device=dev)
    check_call(LIB.dgla_sample_neighbors(ctypes.byref(csr), seeds.data_ptr(), n, fanout, 1 if replace else 0,
                                         rng_seed, indptr.data_ptr(), _ptr(src), _ptr(eids), None, 0, st))
    return indptr, src, eids


def sample_neighbors_weighted(csr, prob, seeds, fanout, replace=False, rng_seed=0):
    """Weighted in-neighbour sampling (dgla_sample_neighbors_weighted): `prob` is a float32 /
    float64 tensor with one non-negative weight per EDGE ID.  Returns ``(indptr, src, eids)``;
    edges of weight 0 never appear, so rows may hold fewer than `fanout` entries."""
    _require_gpu(seeds)
    _require_gpu(prob)
    if prob.dtype not in (torch.float32, torch.float64) or prob.dim() != 1 or not prob.is_contiguous():
        raise _lib.DGLAMDError("prob must be a contiguous 1-D float32 / float64 tensor")
    rng_seed = int(rng_seed) & 0xFFFFFFFFFFFFFFFF
    n = seeds.shape[0]
    dev, dt = seeds.device, seeds.dtype
    indptr = torch.empty(n + 1, dtype=dt, device=dev)
    src = torch.empty(n * fanout, dtype=dt, device=dev)
    eids = torch.empty(n * fanout, dtype=dt, device=dev)
    check_call(LIB.dgla_sample_neighbors_weighted(
        ctypes.byref(csr), prob.data_ptr(), _DTYPES[prob.dtype], seeds.data_ptr(), n, int(fanout),
        1 if replace else 0, rng_seed, indptr.data_ptr(), _ptr(src), _ptr(eids), None, 0, _stream(seeds)))
    return indptr, src, eids


def to_block(seeds, src, node_map):
    """Block-local renumbering of `src` (dgla_to_block).  Returns ``(local_src, src_nodes,
    num_src)``; reading ``num_src`` back is the one host synchronisation of block building
    (the reference synchronises at the same place, cuda_to_block.cu).  The node count is known here
    (``node_map`` has one entry per node), so the sort inside looks at the used key bits only
    (dgla_to_block_padded with every seed valid: 3 radix passes instead of 8 for int64 ids)."""
    _require_gpu(src)
    n, nnz = seeds.shape[0], src.shape[0]
    dev, dt = seeds.device, seeds.dtype
    local = torch.empty(nnz, dtype=dt, device=dev)
    src_nodes = torch.empty(n + nnz, dtype=dt, device=dev)
    num = torch.empty(1, dtype=torch.int64, device=dev)
    if n > 0:
        ws = torch.empty(max(1, LIB.dgla_to_block_workspace_bytes(_idbits(seeds), nnz)), dtype=torch.uint8, device=dev)
        check_call(LIB.dgla_to_block_padded(_idbits(seeds), seeds.data_ptr(), n, None, _ptr(src), nnz,
                                            int(node_map.shape[0]), node_map.data_ptr(), _ptr(local),
                                            src_nodes.data_ptr(), num.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _stream(seeds)))
    else:
        check_call(LIB.dgla_to_block(_idbits(seeds), seeds.data_ptr(), n, _ptr(src), nnz, node_map.data_ptr(),
                                     _ptr(local), src_nodes.data_ptr(), num.data_ptr(), None, 0, _stream(seeds)))
    k = int(num.item())
    return local, src_nodes[:k], k


SINK_ROWS = 64  # sink rows of a padded block (dgla_sample_neighbors_padded)


def sample_neighbors_padded(csr, seeds, num_valid, fanout, replace=False, rng_seed=0, rng_counter=None,
                            sink_rows=SINK_ROWS, prob=None):
    """Static-shape form of :func:`sample_neighbors` (dgla_sample_neighbors_padded): nothing is read
    back.  ``seeds`` has ``n`` slots of which the first ``num_valid`` (int64 device tensor of one
    element, or None = all) are real.  Returns ``(indptr[n + 1 + sink_rows], src[n * fanout],
    eids[n * fanout])``: the real rows' picks first, then the edges of ``sink_rows`` extra SINK rows
    (rows ``n ..``) pointing at the real seeds in turn.  ``rng_counter`` (int64 device tensor) is added to the draw counter on the device,
    so a captured call samples afresh on every replay once the caller bumps it."""
    _require_gpu(seeds)
    n = seeds.shape[0]
    dev, dt = seeds.device, seeds.dtype
    cap = n * int(fanout)
    indptr = torch.empty(n + 1 + int(sink_rows), dtype=dt, device=dev)
    src = torch.empty(cap, dtype=dt, device=dev)
    eids = torch.empty(cap, dtype=dt, device=dev)
    ws = torch.empty(max(1, LIB.dgla_sample_neighbors_workspace_bytes(_idbits(seeds), n)), dtype=torch.uint8,
                     device=dev)
    if prob is not None:
        _require_gpu(prob)
        if prob.dtype not in (torch.float32, torch.float64) or prob.dim() != 1 or not prob.is_contiguous():
            raise _lib.DGLAMDError("prob must be a contiguous 1-D float32 / float64 tensor")
    check_call(LIB.dgla_sample_neighbors_padded(
        ctypes.byref(csr), _ptr(prob), _DTYPES[prob.dtype] if prob is not None else 0, seeds.data_ptr(), n,
        _ptr(num_valid), int(fanout), 1 if replace else 0,
        int(rng_seed) & 0xFFFFFFFFFFFFFFFF, _ptr(rng_counter), int(sink_rows), indptr.data_ptr(), src.data_ptr(),
        eids.data_ptr(),
        ws.data_ptr(), ws.numel(), _stream(seeds)))
    return indptr, src, eids


def to_block_padded(seeds, num_valid, src, node_map, fill=0, num_nodes=0):
    """Static-shape form of :func:`to_block` (dgla_to_block_padded).  Returns ``(local_src,
    src_nodes[n + nnz], num_src)`` with ``num_src`` an int64 DEVICE tensor; entries of ``src_nodes``
    past it hold ``fill`` (a valid node id).  Only the first ``num_valid`` seeds claim a local id."""
    _require_gpu(src)
    n, nnz = seeds.shape[0], src.shape[0]
    dev, dt = seeds.device, seeds.dtype
    local = torch.empty(nnz, dtype=dt, device=dev)
    src_nodes = torch.full((n + nnz,), int(fill), dtype=dt, device=dev)
    num = torch.empty(1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(1, LIB.dgla_to_block_workspace_bytes(_idbits(seeds), nnz)), dtype=torch.uint8, device=dev)
    check_call(LIB.dgla_to_block_padded(_idbits(seeds), seeds.data_ptr(), n, _ptr(num_valid), _ptr(src), nnz,
                                        int(num_nodes), node_map.data_ptr(), _ptr(local), src_nodes.data_ptr(), num.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _stream(seeds)))
    return local, src_nodes, num


def gather_rows(src, idx, out=None):
    """out[i] = src[idx[i]] along dim 0 (dgla_gather_rows): the pack kernel of the halo exchange."""
    _require_gpu(src)
    _require_gpu(idx)
    if not src.is_contiguous():
        raise _lib.DGLAMDError("gather_rows: src must be contiguous")
    idx = idx.contiguous()
    if out is None:
        out = torch.empty((idx.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    elif not out.is_contiguous() or out.shape[0] != idx.shape[0] or out.shape[1:] != src.shape[1:] \
            or out.dtype != src.dtype:
        raise _lib.DGLAMDError("gather_rows: out must be contiguous with one row per index")
    row_bytes = src.element_size()
    for d in src.shape[1:]:
        row_bytes *= int(d)
    check_call(LIB.dgla_gather_rows(_idbits(idx), src.data_ptr(), idx.data_ptr(), idx.shape[0],
                                    row_bytes, out.data_ptr(), _stream(out)))
    return out


def partition_map(mode, num_parts, part_range, idx, want_part=True, want_local=True):
    """(part, local) ids of the global ids `idx` under a remainder (mode 0) or range (mode 1)
    partition (dgla_partition_map)."""
    _require_gpu(idx)
    idx = idx.contiguous()
    part = torch.empty_like(idx) if want_part else None
    local = torch.empty_like(idx) if want_local else None
    check_call(LIB.dgla_partition_map(_idbits(idx), int(mode), int(num_parts), _ptr(part_range),
                                      idx.data_ptr(), idx.shape[0], _ptr(part), _ptr(local),
                                      _stream(idx)))
    return part, local


def partition_to_global(mode, num_parts, part_range, local_idx, part_id):
    _require_gpu(local_idx)
    local_idx = local_idx.contiguous()
    out = torch.empty_like(local_idx)
    check_call(LIB.dgla_partition_to_global(_idbits(local_idx), int(mode), int(num_parts),
                                            _ptr(part_range), local_idx.data_ptr(),
                                            local_idx.shape[0], int(part_id), out.data_ptr(),
                                            _stream(out)))
    return out


def scatter_rows(src, idx, out):
    """out[idx[i]] = src[i] along dim 0 (dgla_scatter_rows)."""
    _require_gpu(src)
    _require_gpu(idx)
    if not (src.is_contiguous() and out.is_contiguous() and idx.is_contiguous()):
        raise _lib.DGLAMDError("scatter_rows: tensors must be contiguous")
    row_bytes = src.element_size()
    for d in src.shape[1:]:
        row_bytes *= int(d)
    check_call(LIB.dgla_scatter_rows(_idbits(idx), src.data_ptr(), idx.data_ptr(), idx.shape[0],
                                     row_bytes, out.data_ptr(), _stream(out)))
    return out
