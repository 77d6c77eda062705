"""Segment aggregation operators: ``segment_reduce`` / ``segment_softmax`` / ``scatter_add``.

Mirror of python/dgl/ops/segment.py:9-114 (API), python/dgl/backend/pytorch/sparse.py:826-865
(``SegmentReduce`` / ``ScatterAdd`` autograd) and python/dgl/_sparse_ops.py:641-760
(``_segment_reduce`` / ``_scatter_add`` / ``_bwd_segment_cmp``: allocation + the FFI call).
The arithmetic runs in libdgl_amd.so behind ``sparse._CAPI_DGLKernelSegmentReduce`` /
``…ScatterAdd`` / ``…BwdSegmentCmp`` (csrc/segment.hip): segment reduce is the merge-path
g-SpMM kernel run over ``offsets`` as a CSR, so a few huge segments and millions of tiny ones
cost the same per element.
"""
import torch

from . import _ffi
from ._lib import DGLAMDError


def _call(name, ref, *args):
    _ffi.use_current_stream(ref.device)
    return _ffi.get_global_func(name)(*args)


def _nd(t):
    return None if t is None else _ffi.NDArray(t)


def _segment_reduce(op, feat, offsets):
    """``(out, arg)`` with ``out[i] = op(feat[offsets[i]:offsets[i+1]])``; ``arg`` is the
    arg-min/max row per element (-1 for an empty segment), None for sum
    (python/dgl/_sparse_ops.py:641-695).  No gradients here."""
    if op not in ("sum", "max", "min"):
        raise DGLAMDError("Unsupported reduce function " + str(op))
    n = offsets.shape[0] - 1
    feat = feat.contiguous()
    offsets = offsets.contiguous()
    out = torch.empty((n,) + tuple(feat.shape[1:]), dtype=feat.dtype, device=feat.device)
    arg = None
    if op in ("min", "max"):
        arg = torch.empty(out.shape, dtype=offsets.dtype, device=feat.device)
    if out.numel():
        _call("sparse._CAPI_DGLKernelSegmentReduce", feat, op, _nd(feat), _nd(offsets), _nd(out),
              _nd(arg))
    return out, arg


def _scatter_add(x, idx, m):
    """``out[idx[i]] += x[i]`` into a zero ``(m, ...)`` tensor (python/dgl/_sparse_ops.py:698-730)."""
    x = x.contiguous()
    out = torch.zeros((m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if x.numel():
        _call("sparse._CAPI_DGLKernelScatterAdd", x, _nd(x), _nd(idx.contiguous()), _nd(out))
    return out


def _bwd_segment_cmp(feat, arg, m):
    """``out[arg[i, k], k] = feat[i, k]`` into a zero ``(m, ...)`` tensor
    (python/dgl/_sparse_ops.py:733-760)."""
    feat = feat.contiguous()
    out = torch.zeros((m,) + tuple(feat.shape[1:]), dtype=feat.dtype, device=feat.device)
    if feat.numel() and out.numel():
        _call("sparse._CAPI_DGLKernelBwdSegmentCmp", feat, _nd(feat), _nd(arg.contiguous()), _nd(out))
    return out


class SegmentReduce(torch.autograd.Function):
    """backend/pytorch/sparse.py:826-853."""

    @staticmethod
    def forward(ctx, op, x, offsets):
        y, arg = _segment_reduce(op, x, offsets)
        ctx.save_for_backward(arg, offsets)
        ctx.op = op
        ctx.m = x.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        arg, offsets = ctx.saved_tensors
        m = ctx.m
        if ctx.op == "sum":
            # every row of segment i receives dy[i]: the segment id of row j is the number of
            # segment ends <= j (handles empty segments anywhere, cf. dmlc/dgl#2610)
            seg = torch.searchsorted(offsets[1:].contiguous(),
                                     torch.arange(m, device=offsets.device, dtype=offsets.dtype),
                                     right=True)
            dx = dy[seg]
        else:
            dx = _bwd_segment_cmp(dy.contiguous(), arg, m)
        return None, dx, None


class ScatterAdd(torch.autograd.Function):
    """backend/pytorch/sparse.py:856-865."""

    @staticmethod
    def forward(ctx, x, idx, m):
        y = _scatter_add(x, idx, m)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return dy[idx.long()], None, None


def scatter_add(x, idx, m):
    return ScatterAdd.apply(x, idx, m)


def segment_reduce(seglen, value, reducer="sum"):
    """Aggregates ``value`` along dim 0 by segments of lengths ``seglen`` (zero-length
    segments allowed); reducer in sum / max / min / mean (python/dgl/ops/segment.py:9-61).

    >>> dgl_amd.segment_reduce(torch.tensor([1, 0, 5, 4]), torch.ones(10, 3))
    tensor([[1., 1., 1.], [0., 0., 0.], [5., 5., 5.], [4., 4., 4.]])
    """
    offsets = torch.cumsum(torch.cat([seglen.new_zeros((1,)), seglen], 0), 0)
    if reducer == "mean":
        rst = SegmentReduce.apply("sum", value, offsets)
        z = torch.clamp(seglen, 1, max(len(value), 1)).to(rst.dtype)
        return rst / z.reshape((rst.shape[0],) + (1,) * (rst.dim() - 1))
    if reducer in ("min", "sum", "max"):
        rst = SegmentReduce.apply(reducer, value, offsets)
        if reducer in ("min", "max"):
            rst = torch.masked_fill(rst, torch.isinf(rst), 0)  # F.replace_inf_with_zero
        return rst
    raise DGLAMDError("reducer {} not recognized.".format(reducer))


def segment_softmax(seglen, value):
    """Softmax over each segment (python/dgl/ops/segment.py:64-103)."""
    value_max = segment_reduce(seglen, value, reducer="max")
    value = torch.exp(value - torch.repeat_interleave(value_max, seglen, dim=0))
    value_sum = segment_reduce(seglen, value, reducer="sum")
    return value / torch.repeat_interleave(value_sum, seglen, dim=0)
