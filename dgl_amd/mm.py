"""``segment_mm`` / ``gather_mm``: per-relation dense transforms next to the g-SpMM
(R-GCN / HGT ``TypedLinear``, python/dgl/nn/pytorch/linear.py:208-210).

Mirror of python/dgl/ops/segment.py:106-136 and python/dgl/ops/gather_mm.py:8-62 (API),
python/dgl/backend/pytorch/sparse.py:968-1020 (``SEGMENTMM`` / ``GATHERMM`` autograd) and
python/dgl/_sparse_ops.py:436-480 (allocation + FFI call).  The arithmetic runs in
libdgl_amd.so (csrc/segment_mm.hip): all relations in ONE grouped MFMA launch instead of the
reference's host loop of cuBLAS calls.
"""
import torch

from . import _ffi
from ._lib import DGLAMDError


def _call(name, ref, *args):
    _ffi.use_current_stream(ref.device)
    return _ffi.get_global_func(name)(*args)


def _nd(t, host_ok=False):
    return None if t is None else _ffi.NDArray(t.contiguous(), host_ok=host_ok)


def _segment_mm(A, B, out, seglen_A, b_trans=False):
    """python/dgl/_sparse_ops.py:436-446."""
    if out.numel():
        _call("sparse._CAPI_DGLKernelSEGMENTMM", A, _nd(A), _nd(B), _nd(out), _nd(seglen_A, True),
              False, bool(b_trans))
    return out


def _segment_mm_backward_B(A, dC, dB, seglen):
    """python/dgl/_sparse_ops.py:449-454."""
    if dB.numel():
        _call("sparse._CAPI_DGLKernelSEGMENTMMBackwardB", A, _nd(A), _nd(dC), _nd(dB), _nd(seglen, True))
    return dB


def _gather_mm(A, B, out, idx_a=None, idx_b=None):
    """python/dgl/_sparse_ops.py:457-466."""
    if out.numel():
        _call("sparse._CAPI_DGLKernelGATHERMM", A, _nd(A), _nd(B), _nd(out), _nd(idx_a), _nd(idx_b))
    return out


class SEGMENTMM(torch.autograd.Function):
    """backend/pytorch/sparse.py:968-990."""

    @staticmethod
    def forward(ctx, A, B, seglen_A):
        if B.dim() != 3:
            raise ValueError("segment_mm expects B to be a 3D tensor.")
        C = torch.empty((A.shape[0], B.shape[2]), device=A.device, dtype=A.dtype)
        A, B = A.contiguous(), B.contiguous()
        _segment_mm(A, B, C, seglen_A)
        # rows beyond sum(seglen) belong to no relation: the library zero-fills them (the
        # reference's output starts as th.zeros); it also clamps device-side lengths to A's rows
        if not seglen_A.is_cuda:  # host lengths: checking costs nothing (a device tensor is
            total = int(seglen_A.sum())  # not read back, that would synchronise)
            if total > A.shape[0]:
                raise DGLAMDError("Segment index out of bound of A->shape[0].")  # gather_mm.cu:224
        ctx.backward_cache = A, B, seglen_A
        return C

    @staticmethod
    def backward(ctx, dZ):
        A, B, seglen_A = ctx.backward_cache
        dZ = dZ.contiguous()
        A_grad = B_grad = None
        if ctx.needs_input_grad[0]:  # A_grad = Out_grad . B^T
            A_grad = torch.empty(A.shape, device=A.device, dtype=A.dtype)  # tail zero-filled by the library
            _segment_mm(dZ, B, A_grad, seglen_A, b_trans=True)
        if ctx.needs_input_grad[1]:  # B_grad = A^T . Out_grad
            B_grad = torch.empty(B.shape, device=B.device, dtype=B.dtype)
            _segment_mm_backward_B(A, dZ, B_grad, seglen_A)
        return A_grad, B_grad, None


def _sort_by_relation(idx_b, num_rel):
    sorted_idx, perm = torch.sort(idx_b, stable=True)
    # ids >= num_rel would make bincount longer than the weight stack: drop those counts, the
    # rows sort to the end, belong to no relation and come out as zeros (the kernel never
    # indexes B past its last matrix)
    seglen = torch.bincount(sorted_idx, minlength=num_rel)[:num_rel].contiguous()
    return perm, seglen


class GATHERMM(torch.autograd.Function):
    """backend/pytorch/sparse.py:993-1020 for ``idx_a is None`` (the only form dgl.ops.gather_mm
    exposes).  Backward: A_grad[i] = dZ[i] . B[idx_b[i]]^T through the same kernel on the
    transposed weights; B_grad[r] = sum_{i: idx_b[i] = r} A[i]^T dZ[i] by sorting the rows by
    relation and running the grouped weight-gradient GEMM (the reference scatters outer
    products with atomics, gather_mm.cu GatherMMScatterKernel2)."""

    @staticmethod
    def forward(ctx, A, B, idx_b):
        if B.dim() != 3:
            raise ValueError("Expected dimension of B is 3. Got " + str(B.dim()))
        A, B = A.contiguous(), B.contiguous()
        C = torch.zeros((len(idx_b), B.shape[2]), device=A.device, dtype=A.dtype)
        _gather_mm(A, B, C, None, idx_b)
        ctx.backward_cache = A, B, idx_b
        return C

    @staticmethod
    def backward(ctx, dZ):
        A, B, idx_b = ctx.backward_cache
        dZ = dZ.contiguous()
        A_grad = B_grad = None
        if ctx.needs_input_grad[0]:
            A_grad = torch.zeros(A.shape, device=A.device, dtype=A.dtype)
            _gather_mm(dZ, B.transpose(1, 2).contiguous(), A_grad, None, idx_b)
        if ctx.needs_input_grad[1]:
            perm, seglen = _sort_by_relation(idx_b, B.shape[0])
            B_grad = torch.empty(B.shape, device=B.device, dtype=B.dtype)
            _segment_mm_backward_B(A.index_select(0, perm), dZ.index_select(0, perm), B_grad, seglen)
        return A_grad, B_grad, None


class GATHERMM_SORTED(torch.autograd.Function):
    """``c[i] = a[i] @ b[idx_b[i]]`` for large inputs: rows are visited grouped by relation through
    a permutation (``dgla_segment_mm_indexed``) instead of being copied into sorted order and back
    (the reference's two ``index_select`` passes, python/dgl/ops/gather_mm.py:44-60).  Forward,
    A-gradient and weight gradient all run on the grouped MFMA kernels."""

    @staticmethod
    def forward(ctx, A, B, idx_b):
        from . import _capi

        if B.dim() != 3:
            raise ValueError("Expected dimension of B is 3. Got " + str(B.dim()))
        A, B = A.contiguous(), B.contiguous()
        perm, seglen = _sort_by_relation(idx_b, B.shape[0])
        perm = perm.long().contiguous()
        C = torch.empty((A.shape[0], B.shape[2]), device=A.device, dtype=A.dtype)
        if C.numel():
            _capi.segment_mm(A, B, C, seglen, row_index=perm)
        ctx.backward_cache = A, B, perm, seglen
        return C

    @staticmethod
    def backward(ctx, dZ):
        from . import _capi

        A, B, perm, seglen = ctx.backward_cache
        dZ = dZ.contiguous()
        A_grad = B_grad = None
        if ctx.needs_input_grad[0]:
            A_grad = torch.empty(A.shape, device=A.device, dtype=A.dtype)
            if A_grad.numel():
                _capi.segment_mm(dZ, B, A_grad, seglen, b_trans=True, row_index=perm)
        if ctx.needs_input_grad[1]:
            B_grad = torch.empty(B.shape, device=B.device, dtype=B.dtype)
            if B_grad.numel():
                _capi.segment_mm_backward_b(A, dZ, B_grad, seglen, row_index=perm)
        return A_grad, B_grad, None


def segment_mm(a, b, seglen_a):
    """``a[0:s0] @ b[0], a[s0:s0+s1] @ b[1], ...`` stacked (python/dgl/ops/segment.py:106-136).
    ``a``: (N, D1), ``b``: (R, D1, D2), ``seglen_a``: (R,) integer tensor on the CPU (as in
    the reference) or on the GPU, summing to N."""
    if a.dim() != 2:
        raise DGLAMDError("segment_mm expects a 2-D left operand")
    return SEGMENTMM.apply(a, b, seglen_a)


def gather_mm(a, b, *, idx_b):
    """``c[i] = a[i] @ b[idx_b[i]]`` (python/dgl/ops/gather_mm.py:8-62).  Like the reference,
    large problems are grouped by relation and run on the segment_mm kernels; here the grouping
    is a permutation the kernels read through — no sorted copies of ``a`` and ``c`` — and stays on
    the device (no ``.cpu()`` synchronisation: seglen is consumed on the GPU)."""
    N, D1 = a.shape
    R, _, D2 = b.shape
    if N > 1000000 or D1 > 8 or D2 > 8:
        if len(idx_b) != N:
            raise DGLAMDError("gather_mm expects one relation index per row of a")
        return GATHERMM_SORTED.apply(a, b, idx_b)
    return GATHERMM.apply(a, b, idx_b)
