"""Differentiable g-SpMM / g-SDDMM / edge softmax (torch.autograd.Function).

Mirror of python/dgl/backend/pytorch/sparse.py for the single-relation path: ``GSpMM``
(:162-248), ``GSDDMM`` (:443-503), ``EdgeSoftmax`` (:685-747) and the wrappers ``gspmm`` /
``gsddmm`` (:1023-1046).  The backward passes are expressed with the same two kernels on the
reversed graph, so forward parity of the kernels gives the gradients.  One deliberate
difference: edge softmax uses the fused forward / backward kernels on the GPU (the reference
runs max-SpMM, sub-SDDMM, exp, sum-SpMM, div-SDDMM there, sparse.py:709-713).
"""
import torch

from .sparse_kernels import (_edge_softmax_backward, _edge_softmax_forward, _gsddmm,
                             _gsddmm_hetero, _gspmm, _gspmm_hetero)


def _reduce_grad(grad, shape):
    """Sum `grad` over the axes that were broadcast in the forward pass so that it has the
    operand's `shape` (sparse.py:43-74)."""
    in_shape, g_shape = tuple(shape[1:]), tuple(grad.shape[1:])
    if in_shape == g_shape:
        return grad
    pad = (1,) * (len(g_shape) - len(in_shape)) + in_shape
    axes = tuple(i + 1 for i, (a, b) in enumerate(zip(g_shape, pad)) if a != b)
    if axes:
        grad = grad.sum(dim=axes, keepdim=True)
    return grad.reshape((-1,) + in_shape)


def _last_dim_is_reduced(u, e):
    # u: (N, ..., D) times e: (E, ..., 1): the gradient w.r.t. e contracts D -> use `dot`
    if u is None or e is None:
        return False
    return u.shape[1:-1] == e.shape[1:-1] and e.shape[-1] == 1 and u.shape[-1] > 1


def _keep_for_spmm_backward(op, reduce_op, need_dx, need_dy):
    """Which forward tensors the backward needs (sparse.py:92-128)."""
    keep_x = op != "copy_lhs" and need_dy and (reduce_op == "sum" or op == "mul")
    keep_y = op != "copy_rhs" and need_dx and (
        (reduce_op == "sum" and op in ("mul", "add")) or (reduce_op != "sum" and op == "mul"))
    keep_arg = (need_dx or need_dy) and reduce_op in ("max", "min")
    return keep_x, keep_y, keep_arg


class GSpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gidx, op, reduce_op, X, Y):
        out, (argX, argY) = _gspmm(gidx, op, reduce_op, X, Y)
        ref = X if X is not None else Y
        ctx.meta = (gidx, op, reduce_op, None if X is None else X.shape,
                    None if Y is None else Y.shape, ref.dtype, ref.device,
                    _last_dim_is_reduced(X, Y))
        need_dx = X is not None and X.requires_grad
        need_dy = Y is not None and Y.requires_grad
        kx, ky, ka = _keep_for_spmm_backward(op, reduce_op, need_dx, need_dy)
        ctx.save_for_backward(X if kx else None, Y if ky else None,
                              argX if ka else None, argY if ka else None)
        return out

    @staticmethod
    def backward(ctx, dZ):
        gidx, op, reduce_op, x_shape, y_shape, dtype, device, reduce_last = ctx.meta
        X, Y, argX, argY = ctx.saved_tensors
        dZ = dZ.contiguous()
        dX = dY = None
        if op != "copy_rhs" and ctx.needs_input_grad[3]:
            if reduce_op == "sum":
                rev = gidx.reverse()
                if op == "mul":
                    dX = gspmm(rev, "mul", "sum", dZ, Y)
                else:  # add, copy_lhs: the message is linear in X with coefficient 1
                    dX = gspmm(rev, "copy_lhs", "sum", dZ, None)
            else:
                dX = torch.zeros((x_shape[0],) + tuple(dZ.shape[1:]), dtype=dtype, device=device)
                if op == "mul":
                    g = Y.expand(-1, *dZ.shape[1:]).gather(0, argY.long()) * dZ
                    dX.scatter_add_(0, argX.long(), g)
                else:
                    dX.scatter_add_(0, argX.long(), dZ)
            dX = _reduce_grad(dX, x_shape)
        if op != "copy_lhs" and ctx.needs_input_grad[4]:
            if reduce_op == "sum":
                if op == "mul":
                    dY = gsddmm(gidx, "dot" if reduce_last else "mul", X, dZ)
                else:  # add, copy_rhs
                    dY = gsddmm(gidx, "copy_rhs", X, dZ)
            else:
                dY = torch.zeros((y_shape[0],) + tuple(dZ.shape[1:]), dtype=dtype, device=device)
                if op == "mul":
                    g = X.expand(-1, *dZ.shape[1:]).gather(0, argX.long()) * dZ
                    dY.scatter_add_(0, argY.long(), g)
                else:
                    dY.scatter_add_(0, argY.long(), dZ)
            dY = _reduce_grad(dY, y_shape)
        return None, None, None, dX, dY


class GSpMMMean(torch.autograd.Function):
    """``mean`` reducer with the division fused into the SpMM kernel (the reference composes
    ``gspmm(.., 'sum', ..) / clamp(in_degrees, 1)``, python/dgl/ops/spmm.py:109-114, i.e. one
    more pass over the output; forward values are identical).  Backward: the incoming gradient
    is divided by the same degrees, then flows through the sum reducer's backward."""

    @staticmethod
    def forward(ctx, gidx, op, X, Y, in_deg):
        out, _ = _gspmm(gidx, op, "sum", X, Y, mean=True)
        ref = X if X is not None else Y
        ctx.meta = (gidx, op, None if X is None else X.shape, None if Y is None else Y.shape,
                    _last_dim_is_reduced(X, Y))
        need_dx = X is not None and X.requires_grad
        need_dy = Y is not None and Y.requires_grad
        kx, ky, _ = _keep_for_spmm_backward(op, "sum", need_dx, need_dy)
        ctx.save_for_backward(X if kx else None, Y if ky else None, in_deg)
        return out

    @staticmethod
    def backward(ctx, dZ):
        gidx, op, x_shape, y_shape, reduce_last = ctx.meta
        X, Y, in_deg = ctx.saved_tensors
        den = in_deg.to(dZ.dtype).clamp(min=1).reshape((dZ.shape[0],) + (1,) * (dZ.dim() - 1))
        dZ = (dZ / den).contiguous()
        dX = dY = None
        if op != "copy_rhs" and ctx.needs_input_grad[2]:
            rev = gidx.reverse()
            dX = gspmm(rev, "mul", "sum", dZ, Y) if op == "mul" else gspmm(rev, "copy_lhs", "sum", dZ, None)
            dX = _reduce_grad(dX, x_shape)
        if op != "copy_lhs" and ctx.needs_input_grad[3]:
            if op == "mul":
                dY = gsddmm(gidx, "dot" if reduce_last else "mul", X, dZ)
            else:
                dY = gsddmm(gidx, "copy_rhs", X, dZ)
            dY = _reduce_grad(dY, y_shape)
        return None, None, dX, dY, None


def gspmm_mean(gidx, op, lhs_data, rhs_data, in_deg):
    """Single-relation ``mean`` g-SpMM on a graph whose CSC format is allowed."""
    if op == "sub":
        op, rhs_data = "add", -rhs_data
    if op == "div":
        op, rhs_data = "mul", 1.0 / rhs_data
    lhs_data, rhs_data = _autocast(lhs_data, rhs_data)
    with torch.autocast("cuda", enabled=False):
        return GSpMMMean.apply(gidx, op, lhs_data, rhs_data, in_deg)


class GSDDMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gidx, op, X, Y, lhs_target, rhs_target):
        out = _gsddmm(gidx, op, X, Y, lhs_target, rhs_target)
        ctx.meta = (gidx, op, lhs_target, rhs_target, None if X is None else X.shape,
                    None if Y is None else Y.shape)
        need_dx = X is not None and X.requires_grad
        need_dy = Y is not None and Y.requires_grad
        prod = op in ("mul", "dot")
        ctx.save_for_backward(X if prod and need_dy else None, Y if prod and need_dx else None)
        return out

    @staticmethod
    def backward(ctx, dZ):
        gidx, op, lt, rt, x_shape, y_shape = ctx.meta
        X, Y = ctx.saved_tensors
        dZ = dZ.contiguous()

        def operand_grad(own_tgt, other_tgt, other, copy_op):
            # gradient of out = op(own[own_tgt], other[other_tgt]) w.r.t. `own` (sparse.py:459-503)
            if own_tgt in ("u", "v"):
                g = gidx if own_tgt == "v" else gidx.reverse()
                if op in ("add", copy_op):
                    return gspmm(g, "copy_rhs", "sum", None, dZ)
                if other_tgt == own_tgt:
                    return gspmm(g, "copy_rhs", "sum", None, dZ) * other
                if other_tgt == "e":
                    return gspmm(g, "copy_rhs", "sum", None, dZ * other)
                return gspmm(g, "mul", "sum", other, dZ)
            if op in ("add", copy_op):
                return dZ
            return gsddmm(gidx, "mul", dZ, other, "e", other_tgt)

        dX = dY = None
        if op != "copy_rhs" and ctx.needs_input_grad[2]:
            dX = _reduce_grad(operand_grad(lt, rt, Y, "copy_lhs"), x_shape)
        if op != "copy_lhs" and ctx.needs_input_grad[3]:
            dY = _reduce_grad(operand_grad(rt, lt, X, "copy_rhs"), y_shape)
        return None, None, dX, dY, None, None


class EdgeSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gidx, score, eids, norm_by):
        if eids is not None:
            gidx = gidx.edge_subgraph([eids], True)
        if norm_by == "src":
            gidx = gidx.reverse()
        out = _edge_softmax_forward(gidx, score, "copy_rhs")
        ctx.gidx = gidx
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        sds = out * grad_out
        return None, _edge_softmax_backward(ctx.gidx, out, sds), None, None


def _autocast(*tensors):
    # the reference casts operands to the autocast dtype and runs the Function with autocast
    # disabled (sparse.py:146-159,1030-1032)
    if not torch.is_autocast_enabled():
        return tensors
    dt = torch.get_autocast_gpu_dtype()
    return tuple(t.to(dt) if (t is not None and t.is_floating_point()) else t for t in tensors)


def gspmm(gidx, op, reduce_op, lhs_data, rhs_data):
    if op == "sub":
        op, rhs_data = "add", -rhs_data
    elif op == "div":
        op, rhs_data = "mul", 1.0 / rhs_data
    lhs_data, rhs_data = _autocast(lhs_data, rhs_data)
    with torch.autocast("cuda", enabled=False):
        return GSpMM.apply(gidx, op, reduce_op, lhs_data, rhs_data)


def gsddmm(gidx, op, lhs_data, rhs_data, lhs_target="u", rhs_target="v"):
    if op == "sub":
        op, rhs_data = "add", -rhs_data
    elif op == "div":
        op, rhs_data = "mul", 1.0 / rhs_data
    lhs_data, rhs_data = _autocast(lhs_data, rhs_data)
    with torch.autocast("cuda", enabled=False):
        return GSDDMM.apply(gidx, op, lhs_data, rhs_data, lhs_target, rhs_target)


def edge_softmax(gidx, logits, eids=None, norm_by="dst"):
    (logits,) = _autocast(logits)
    with torch.autocast("cuda", enabled=False):
        return EdgeSoftmax.apply(gidx, logits, eids, norm_by)


# ---- heterogeneous wrappers (forward only for max/min; sum is differentiable through the
# per-relation Functions above) --------------------------------------------------------
def gspmm_hetero(gidx, op, reduce_op, lhs_len, *lhs_and_rhs):
    lhs, rhs = list(lhs_and_rhs[:lhs_len]), list(lhs_and_rhs[lhs_len:])
    if op == "sub":
        op, rhs = "add", [None if t is None else -t for t in rhs]
    elif op == "div":
        op, rhs = "mul", [None if t is None else 1.0 / t for t in rhs]
    if reduce_op == "sum" and any(t is not None and t.requires_grad for t in lhs + rhs):
        # differentiable path: per-relation autograd Functions, summed per destination type
        outs = [None] * gidx.number_of_ntypes()
        for et in range(gidx.number_of_etypes()):
            s, d = gidx.metagraph.find_edge(et)
            u = lhs[s] if op != "copy_rhs" else None
            e = rhs[et] if op != "copy_lhs" else None
            if (op != "copy_rhs" and u is None) or (op != "copy_lhs" and e is None):
                continue
            o = gspmm(gidx.get_relation_graph(et), op, "sum", u, e)
            outs[d] = o if outs[d] is None else outs[d] + o
        return tuple(outs)
    out, _ = _gspmm_hetero(gidx, op, reduce_op, lhs_len, tuple(lhs) + tuple(rhs))
    return out


def gsddmm_hetero(gidx, op, lhs_len, lhs_target, rhs_target, *lhs_and_rhs):
    lhs, rhs = list(lhs_and_rhs[:lhs_len]), list(lhs_and_rhs[lhs_len:])
    if op == "sub":
        op, rhs = "add", [None if t is None else -t for t in rhs]
    elif op == "div":
        op, rhs = "mul", [None if t is None else 1.0 / t for t in rhs]
    if any(t is not None and t.requires_grad for t in lhs + rhs):
        outs = []
        for et in range(gidx.number_of_etypes()):
            s, d = gidx.metagraph.find_edge(et)
            pick = lambda tup, tgt: tup[{"u": s, "v": d, "e": et}[tgt]]
            l = pick(lhs, lhs_target) if op != "copy_rhs" else None
            r = pick(rhs, rhs_target) if op != "copy_lhs" else None
            if (op != "copy_rhs" and l is None) or (op != "copy_lhs" and r is None):
                outs.append(None)
            else:
                outs.append(gsddmm(gidx.get_relation_graph(et), op, l, r, lhs_target, rhs_target))
        return tuple(outs)
    return _gsddmm_hetero(gidx, op, lhs_len, lhs_target, rhs_target, tuple(lhs) + tuple(rhs))
