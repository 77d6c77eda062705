"""Differentiable g-SpMM / g-SDDMM / edge softmax (torch.autograd.Function).

Mirror of python/dgl/backend/pytorch/sparse.py for the single-relation path: ``GSpMM``
(:162-248), ``GSDDMM`` (:443-503), ``EdgeSoftmax`` (:685-747) and the wrappers ``gspmm`` /
``gsddmm`` (:1023-1046).  The backward passes are expressed with the same two kernels on the
reversed graph, so forward parity of the kernels gives the gradients.  One deliberate
difference: edge softmax uses the fused forward / backward kernels on the GPU (the reference
runs max-SpMM, sub-SDDMM, exp, sum-SpMM, div-SDDMM there, sparse.py:709-713).
"""
import torch

from . import edge_order as _eo
from ._lib import DGLAMDError
from .sparse_kernels import (_edge_softmax_backward, _edge_softmax_forward, _gsddmm,
                             _gsddmm_hetero, _gspmm, _gspmm_hetero, _update_grad_minmax_hetero)


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


def _bcast_group(other_shape, dz_shape):
    """(row length of `other`, group) such that element k of a dz row pairs with element
    (k // group) % len of the other operand's row, or None when the broadcast is neither
    "leading dims then ones" nor "ones then trailing dims"."""
    s = tuple(int(d) for d in dz_shape)
    t = tuple(int(d) for d in other_shape)
    t = (1,) * (len(s) - len(t)) + t
    if len(t) != len(s) or any(a != b and a != 1 for a, b in zip(t, s)):
        return None
    n = 1
    for d in t:
        n *= d
    a = 0
    while a < len(s) and t[a] == s[a]:
        a += 1
    if all(d == 1 for d in t[a:]):                       # leading dims of dz, then ones
        g = 1
        for d in s[a:]:
            g *= d
        return n, max(g, 1)
    b = len(s)
    while b > 0 and t[b - 1] == s[b - 1]:
        b -= 1
    if all(d == 1 for d in t[:b]):                       # ones, then trailing dims of dz
        return n, 1
    return None


def _cmp_backward(dZ, arg, rows, other, arg_other, atomic):
    """Gradient of a max / min g-SpMM operand: ``out[arg[i, k], k] += dZ[i, k] * other[arg_other[i, k], k']``
    in ONE launch reading the winners in the graph's idtype (dgla_spmm_cmp_backward); the reference is two
    ``.long()`` casts, a ``gather`` and a ``scatter_add_`` (python/dgl/backend/pytorch/sparse.py:217-244)."""
    from . import _capi

    out = torch.zeros((rows,) + tuple(dZ.shape[1:]), dtype=dZ.dtype, device=dZ.device)
    if dZ.numel() == 0 or rows == 0:
        return out
    if not dZ.is_cuda:
        raise _capi._lib.DGLAMDError("dgl_amd kernels run on a ROCm GPU (no CPU fallback)")
    grp = (0, 1)
    if other is not None:
        grp = _bcast_group(other.shape[1:], dZ.shape[1:])
        if grp is None:
            # a broadcast in the middle of the feature shape: compose (rare; still no host round trip)
            g = other.expand(-1, *dZ.shape[1:]).gather(0, arg_other.long()) * dZ
            return _capi.spmm_cmp_backward(g.contiguous(), arg.contiguous(), out, atomic=atomic)
        other = other.contiguous()
    return _capi.spmm_cmp_backward(dZ, arg.contiguous(), out, other,
                                   None if other is None else arg_other.contiguous(), grp[1], atomic=atomic)


def _cmp_backward_node_gather(gidx, dZ, arg_u, rows):
    """dX of a max / min g-SpMM whose message is X itself (copy_lhs, add) WITHOUT atomics: winner bits per
    (edge, column) over the forward CSC (dgla_spmm_cmp_mask), then one merge-path g-SpMM over the reverse matrix
    gated by them (dgla_spmm_csr_masked) — the reference's ``dX.scatter_add_(0, argX.long(), dZ)``
    (python/dgl/backend/pytorch/sparse.py:216-224) as a gather, same bits on every run.  Returns None when this
    relation should take the atomic kernel instead (a sampled block rebuilt every step: the CSR it needs would
    be built for one use; ``DGLA_CMP_BACKWARD=atomic``)."""
    import os
    from . import _capi

    rel = gidx.relations[0]
    if (rel.transient or not rel.allowed("csr") or not rel.allowed("csc") or rel.num_edges == 0 or
            os.environ.get("DGLA_CMP_BACKWARD", "") == "atomic" or not dZ.is_cuda):
        return None
    n_edges = rel.num_edges
    feat = 1
    for d in dZ.shape[1:]:
        feat *= d
    if feat == 0 or dZ.shape[0] == 0 or rows == 0:
        return None
    cache = rel.__dict__.setdefault("_cmp_gather", {})
    if "csr" not in cache:
        ip, idx, eids = rel.csc()        # forward: rows = destinations
        rip, ridx, reids = rel.csr()     # its transpose: rows = sources (built once per graph, like the sum's backward)
        pos = torch.arange(n_edges, dtype=rel.idtype, device=rel.device)
        if eids is None:
            inv = pos
        else:
            inv = torch.empty_like(pos)
            inv[eids.long()] = pos       # forward position of every edge id
        pmap = inv if reids is None else inv[reids.long()]
        cache["fwd"] = _capi.make_csr(ip, idx, eids, rel.num_src)
        cache["csr"] = _capi.make_csr(rip, ridx, pmap.contiguous(), rel.num_dst)
        cache["ws"] = {}
    dz2 = dZ.reshape(dZ.shape[0], feat)
    # two-step form: the bit kernel leaves dX alone, the gated g-SpMM STORES its rows (no zero fill, no read of dX), the finish
    # call adds the elements no edge claimed (dX[0]; and whatever a hand-made arg names)
    dX = torch.empty(rows, feat, dtype=dZ.dtype, device=dZ.device)
    mask = torch.empty(_capi.spmm_cmp_mask_bytes(dZ.dtype, dZ.shape[0], n_edges, feat), dtype=torch.uint8, device=dZ.device)
    arg2 = arg_u.reshape(arg_u.shape[0], feat).contiguous()
    _capi.spmm_cmp_mask(cache["fwd"], arg2, dz2, mask, dX, mode=_capi.CMP_MASK_DEFER)
    key = (dZ.dtype, feat)
    ws = cache["ws"].get(key)
    fresh = ws is None
    if fresh:
        ws = cache["ws"][key] = torch.empty(max(int(_capi.spmm_csr_masked_workspace_bytes(cache["csr"], dz2, dX)), 1),
                                            dtype=torch.uint8, device=dZ.device)
    _capi.spmm_csr_masked(cache["csr"], dz2, mask, dX, workspace=ws, accumulate=False, plan_valid=not fresh)
    _capi.spmm_cmp_mask(cache["fwd"], arg2, dz2, mask, dX, mode=_capi.CMP_MASK_FINISH)
    return dX.reshape((rows,) + tuple(dZ.shape[1:]))


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
                # a source node can win at many destinations: a gather over the reverse graph gated by winner bits
                # (deterministic) when the message is X itself, else the atomic sum (as the reference's scatter_add_)
                dX = _cmp_backward_node_gather(gidx, dZ, argX, x_shape[0]) if op != "mul" else None
                if dX is None:
                    dX = _cmp_backward(dZ, argX, x_shape[0], Y if op == "mul" else None, argY, atomic=True)
            dX = _reduce_grad(dX, x_shape)
        if op != "copy_lhs" and ctx.needs_input_grad[4]:
            if reduce_op == "sum":
                if op == "mul":
                    dY = gsddmm(gidx, "dot" if reduce_last else "mul", X, dZ, _handoff=False)
                else:  # add, copy_rhs
                    dY = gsddmm(gidx, "copy_rhs", X, dZ, _handoff=False)
            else:
                # an edge has ONE destination: every (edge, k) is written at most once — plain stores
                dY = _cmp_backward(dZ, argY, y_shape[0], X if op == "mul" else None, argX, atomic=False)
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
                dY = gsddmm(gidx, "dot" if reduce_last else "mul", X, dZ, _handoff=False)
            else:
                dY = gsddmm(gidx, "copy_rhs", X, dZ, _handoff=False)
            dY = _reduce_grad(dY, y_shape)
        return None, None, dX, dY, None


def gspmm_mean(gidx, op, lhs_data, rhs_data, in_deg):
    """Single-relation ``mean`` g-SpMM on a graph whose CSC format is allowed."""
    if op == "sub":
        op, rhs_data = "add", -rhs_data
    if op == "div":
        op, rhs_data = "mul", 1.0 / rhs_data
    lhs_data = None if lhs_data is None else _eo.plain(lhs_data)
    rhs_data = None if rhs_data is None else _eo.plain(rhs_data)
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
            return gsddmm(gidx, "mul", dZ, other, "e", other_tgt, _handoff=False)

        dX = dY = None
        if op != "copy_rhs" and ctx.needs_input_grad[2]:
            dX = _reduce_grad(operand_grad(lt, rt, Y, "copy_lhs"), x_shape)
        if op != "copy_lhs" and ctx.needs_input_grad[3]:
            dY = _reduce_grad(operand_grad(rt, lt, X, "copy_rhs"), y_shape)
        return None, None, dX, dY, None, None


class EdgeSoftmax(torch.autograd.Function):
    """Edge softmax, plain tensors in and out (sparse.py:685-747).  Behind DGL's usual edge-id map the softmax is
    KEPT in the CSC's position order between forward and backward — a tensor nobody outside this Function sees:
    the forward kernel reads the scores through the map and writes by position, the backward kernel streams that
    saved tensor and reads the caller's gradient through the map; what leaves is permuted back to edge-id order by
    ONE gather through the inverse map.  Per pass that is two scattered 32-byte READS per edge where reading and
    writing through the map is a scattered read plus a scattered WRITE (62 M edges x 8 heads: 4.6 -> ~3.3 ms forward,
    12.3 -> ~7 ms forward + backward; profiles/r5).  Same kernels, same arithmetic, same bits."""

    @staticmethod
    def forward(ctx, gidx, score, eids, norm_by):
        if eids is not None:
            gidx = gidx.edge_subgraph([eids], True)
        if norm_by == "src":
            gidx = gidx.reverse()
        ctx.gidx, ctx.pos = gidx, False
        rel = gidx.relations[0]
        if eids is None and _eo.plain_softmax_route(rel, score):
            s = (score.unsqueeze(-1) if score.dim() == 1 else score).contiguous()
            out_pos = _eo.softmax_gather_pos_forward(rel, s, rel.csc()[2])
            if out_pos is not None:
                from . import _capi
                ctx.pos, ctx.expand = True, score.dim() == 1
                ctx.save_for_backward(out_pos)
                out = _capi.gather_rows(out_pos, _eo.inverse_map(rel))
                return out.squeeze(-1) if ctx.expand else out
        out = _edge_softmax_forward(gidx, score, "copy_rhs")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        if ctx.pos:
            from . import _capi
            rel = ctx.gidx.relations[0]
            g = (grad_out.unsqueeze(-1) if ctx.expand else grad_out).to(out.dtype)
            back = _capi.gather_rows(_eo.softmax_pos_backward_from_eid_grad(rel, out, g), _eo.inverse_map(rel))
            return None, (back.squeeze(-1) if ctx.expand else back), None, None
        sds = out * grad_out
        return None, _edge_softmax_backward(ctx.gidx, out, sds), None, None


class GATAttention(torch.autograd.Function):
    """``out[v] = sum_{u->v} softmax_v(leaky_relu(el[u] + er[v])) ft[u]`` per head as one operator
    (csrc/gat_attention.hip).  Saved for the backward: the operands, ``out`` and the rows' softmax (max, normaliser) —
    2 floats per (node, head); the attention weights are recomputed, no (E, H) tensor is kept."""

    @staticmethod
    def forward(ctx, gidx, ft, el, er, slope):
        rel = gidx.relations[0]
        ft, el, er = ft.contiguous(), el.contiguous(), er.contiguous()
        n_dst, (h, d) = rel.num_dst, ft.shape[1:]
        out = torch.empty((n_dst, h, d), dtype=ft.dtype, device=ft.device)
        mz = torch.empty((n_dst, h, 2), dtype=torch.float32, device=ft.device)
        ws = _gat_workspace(rel, h, d)
        _call_unit("dgl_amd._CAPI_GATAttentionForward", rel, ("csc",), _nd(ft), _nd(el), _nd(er), float(slope), _nd(out),
                   _nd(mz), _nd(ws))
        ctx.meta = (gidx, float(slope))
        ctx.save_for_backward(ft, el, er, out, mz)
        return out

    @staticmethod
    def backward(ctx, dout):
        gidx, slope = ctx.meta
        ft, el, er, out, mz = ctx.saved_tensors
        rel = gidx.relations[0]
        dout = _eo.plain(dout).contiguous()
        d_ft, d_el, d_er = torch.empty_like(ft), torch.empty_like(el), torch.empty_like(er)
        ws = _gat_workspace(rel, ft.shape[1], ft.shape[2])
        _call_unit("dgl_amd._CAPI_GATAttentionBackward", rel, ("csc", "csr"), _nd(ft), _nd(el), _nd(er), _nd(out), _nd(mz),
                   _nd(dout), slope, _nd(d_ft), _nd(d_el), _nd(d_er), _nd(ws))
        return None, d_ft, d_el, d_er, None


def _nd(t):
    from . import _ffi
    return None if t is None else _ffi.NDArray(t)


def _call_unit(name, rel, fmts, *args):
    from . import _ffi
    _ffi.use_current_stream(rel.device)
    h = None
    for f in fmts:          # every format the call reads is registered on the one unit-graph handle
        h = rel.handle(f)
    return _ffi.get_global_func(name)(h, *args)


def _gat_workspace(rel, heads, dim):
    """Scratch of the one-pass GAT kernels (chunk table + partial rows + the backward's per-node record), kept on the
    relation and grown on demand."""
    key = (int(heads), int(dim))
    need = rel.__dict__.setdefault("_gat_need", {})
    nbytes = need.get(key)
    if nbytes is None:
        nbytes = need[key] = _call_unit("dgl_amd._CAPI_GATAttentionWorkspaceBytes", rel, ("csc",), key[0], key[1])
    ws = rel.__dict__.get("_gat_ws")
    if ws is None or ws.numel() < nbytes:
        ws = rel.__dict__["_gat_ws"] = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=rel.device)
    return ws


def gat_attention(gidx, ft, el, er, slope):
    ft, el, er = _eo.plain(ft), _eo.plain(el), _eo.plain(er)
    with torch.autocast("cuda", enabled=False):
        return GATAttention.apply(gidx, ft, el, er, slope)


def _autocast(*tensors):
    # the reference casts operands to the autocast dtype and runs the Function with autocast
    # disabled (sparse.py:146-159,1030-1032)
    if not torch.is_autocast_enabled():
        return tensors
    dt = torch.get_autocast_gpu_dtype()
    return tuple(t.to(dt) if (t is not None and t.is_floating_point()) else t for t in tensors)


def _edge_operand_route(gidx, op, reduce_op, rhs_data):
    """Can a position-ordered edge operand be used as it is?  Returns the relation it is ordered by
    (then dgl_amd.edge_order.PosGSpMM runs), else None (then the tensor is converted to edge-id order)."""
    rel = _eo.tag_of(rhs_data)
    if rel is None or reduce_op != "sum" or op == "copy_lhs" or gidx.number_of_etypes() != 1:
        return None
    q = gidx.relations[0]
    if rel is q:
        return rel if q.allowed("csc") else None
    if rel.reverse() is q:
        return rel if rel.allowed("csr") else None
    return None


def gspmm(gidx, op, reduce_op, lhs_data, rhs_data):
    if op == "sub":
        op, rhs_data = "add", -rhs_data
    elif op == "div":
        op, rhs_data = "mul", 1.0 / rhs_data
    lhs_data = _eo.plain(lhs_data) if lhs_data is not None else None
    rel = _edge_operand_route(gidx, op, reduce_op, rhs_data) if not torch.is_autocast_enabled() else None
    if rel is not None:
        return _eo.PosGSpMM.apply(gidx, op, lhs_data, rhs_data, rel)
    rhs_data = _eo.plain(rhs_data) if rhs_data is not None else None
    lhs_data, rhs_data = _autocast(lhs_data, rhs_data)
    with torch.autocast("cuda", enabled=False):
        return GSpMM.apply(gidx, op, reduce_op, lhs_data, rhs_data)


def gsddmm(gidx, op, lhs_data, rhs_data, lhs_target="u", rhs_target="v", _handoff=True):
    """``_handoff=False``: the result is the gradient of an EDGE-ID-ORDERED tensor (the backward passes
    of the plain Functions above): it must come out edge-id ordered and untagged."""
    if op == "sub":
        op, rhs_data = "add", -rhs_data
    elif op == "div":
        op, rhs_data = "mul", 1.0 / rhs_data
    # position-ordered hand-off (dgl_amd.edge_order): the result is produced in the CSC position order
    # of the relation when the graph's edge ids are not its CSC order anyway and no operand is an
    # edge-id-ordered edge tensor; 'e' operands tagged with this relation are read as they are
    if _handoff and gidx.number_of_etypes() == 1 and not torch.is_autocast_enabled():
        rel = gidx.relations[0]
        use = {"l": op != "copy_rhs", "r": op != "copy_lhs"}
        nat_l = use["l"] and lhs_target == "e" and _eo.tag_of(lhs_data) is rel
        nat_r = use["r"] and rhs_target == "e" and _eo.tag_of(rhs_data) is rel
        plain_e = (use["l"] and lhs_target == "e" and not nat_l) or (use["r"] and rhs_target == "e" and not nat_r)
        if not plain_e and (nat_l or nat_r or _eo.wants_handoff(rel)) and rel.allowed("csc") and \
                (lhs_data if use["l"] else rhs_data).is_cuda:
            lhs_data = lhs_data if (nat_l or lhs_data is None) else _eo.plain(lhs_data)
            rhs_data = rhs_data if (nat_r or rhs_data is None) else _eo.plain(rhs_data)
            return _eo.PosGSDDMM.apply(gidx, op, lhs_data, rhs_data, lhs_target, rhs_target)
    lhs_data = _eo.plain(lhs_data) if lhs_data is not None else None
    rhs_data = _eo.plain(rhs_data) if rhs_data is not None else None
    lhs_data, rhs_data = _autocast(lhs_data, rhs_data)
    with torch.autocast("cuda", enabled=False):
        return GSDDMM.apply(gidx, op, lhs_data, rhs_data, lhs_target, rhs_target)


def edge_softmax(gidx, logits, eids=None, norm_by="dst"):
    if eids is None and norm_by == "dst" and gidx.number_of_etypes() == 1 and logits.is_cuda and \
            not torch.is_autocast_enabled():
        rel = gidx.relations[0]
        tag = _eo.tag_of(logits)
        if rel.allowed("csc") and (tag is rel or (tag is None and _eo.wants_handoff(rel))):
            return _eo.PosEdgeSoftmax.apply(gidx, logits)
    logits = _eo.plain(logits)
    (logits,) = _autocast(logits)
    with torch.autocast("cuda", enabled=False):
        return EdgeSoftmax.apply(gidx, logits, eids, norm_by)


# ---- heterogeneous graphs (python/dgl/backend/pytorch/sparse.py:251-440,506-600,750-850) ------
def _zeros_like_out(gidx, nt, ref):
    return torch.zeros((gidx.num_nodes(nt),) + tuple(ref.shape[1:]), dtype=ref.dtype, device=ref.device)


class GSpMM_hetero(torch.autograd.Function):
    """g-SpMM over every relation of a heterograph, differentiable (sparse.py:251-440).  The
    forward is ``_gspmm_hetero`` — ONE stacked launch per destination type for sum reductions —
    whether or not gradients are needed; the backward runs the same machinery on the reversed
    graph (sum) or scatters through the recorded winners (max / min)."""

    @staticmethod
    def forward(ctx, gidx, op, reduce_op, X_len, *feats):
        out, (argX, argY, argX_nt, argY_et) = _gspmm_hetero(gidx, op, reduce_op, X_len, feats)
        X, Y = feats[:X_len], feats[X_len:]
        need = lambda ts: any(t is not None and t.requires_grad for t in ts)
        ctx.meta = (gidx, op, reduce_op, X_len, len(Y),
                    tuple(None if t is None else t.shape for t in X),
                    tuple(None if t is None else t.shape for t in Y))
        # first relation that has both operands decides whether the last axis is contracted
        reduce_last = False
        for et in range(gidx.number_of_etypes()):
            s, _ = gidx.metagraph.find_edge(et)
            if X_len and len(Y) and X[s] is not None and Y[et] is not None:
                reduce_last = _last_dim_is_reduced(X[s], Y[et])
                break
        ctx.reduce_last = reduce_last
        kx, ky, ka = _keep_for_spmm_backward(op, reduce_op, need(X), need(Y))
        n_nt = gidx.number_of_ntypes()
        pad = lambda ts: tuple(ts) if ts is not None else (None,) * n_nt
        ctx.n_saved = (len(X), len(Y), n_nt)
        ctx.save_for_backward(*(X if kx else (None,) * len(X)), *(Y if ky else (None,) * len(Y)),
                              *(pad(argX) if ka else (None,) * n_nt), *(pad(argX_nt) if ka else (None,) * n_nt),
                              *(pad(argY) if ka else (None,) * n_nt), *(pad(argY_et) if ka else (None,) * n_nt))
        ctx.mark_non_differentiable(*[o for o in out if o is not None and not o.is_floating_point()])
        return tuple(out)

    @staticmethod
    def backward(ctx, *dZ):
        gidx, op, reduce_op, X_len, Y_len, X_shape, Y_shape = ctx.meta
        nx, ny, n_nt = ctx.n_saved
        sv = ctx.saved_tensors
        X, Y = sv[:nx], sv[nx:nx + ny]
        argX, argX_nt, argY, argY_et = (sv[nx + ny + k * n_nt: nx + ny + (k + 1) * n_nt] for k in range(4))
        dZ = tuple(None if z is None else z.contiguous() for z in dZ)
        need_x = any(ctx.needs_input_grad[4:4 + X_len])
        need_y = any(ctx.needs_input_grad[4 + X_len:])
        dX, dY = (None,) * X_len, (None,) * Y_len
        if op != "copy_rhs" and need_x:
            g_rev = gidx.reverse()
            if reduce_op == "sum":
                # on the reversed graph the "source" features are the incoming gradients, indexed
                # by (forward) destination type
                if op == "mul":
                    dX = gspmm_hetero(g_rev, "mul", "sum", len(dZ), *(dZ + tuple(Y)))
                else:  # add, copy_lhs: coefficient 1
                    dX = gspmm_hetero(g_rev, "copy_lhs", "sum", len(dZ), *dZ)
            elif op in ("add", "copy_lhs"):
                shapes = [None if shp is None else torch.empty(shp[0], 0) for shp in X_shape]
                dX = _update_grad_minmax_hetero(g_rev, "copy_lhs", dZ, argX, argX_nt, shapes)
            else:
                raise NotImplementedError("gradient of u_mul_e with a max / min reducer on a graph with "
                                          "several relations (the reference has none either)")
            dX = tuple(None if (d is None or X_shape[i] is None) else _reduce_grad(d, X_shape[i])
                       for i, d in enumerate(dX))
        if op != "copy_lhs" and need_y:
            if reduce_op == "sum":
                if op == "mul":
                    dY = gsddmm_hetero(gidx, "dot" if ctx.reduce_last else "mul", X_len, "u", "v",
                                       *(tuple(X) + dZ))
                else:  # add, copy_rhs
                    dY = gsddmm_hetero(gidx, "copy_rhs", gidx.number_of_ntypes(), "u", "v",
                                       *((None,) * gidx.number_of_ntypes() + dZ))
            elif op in ("add", "copy_rhs"):
                shapes = [None if shp is None else torch.empty(shp[0], 0) for shp in Y_shape]
                dY = _update_grad_minmax_hetero(gidx.reverse(), "copy_rhs", dZ, argY, argY_et, shapes)
            else:
                raise NotImplementedError("gradient of u_mul_e with a max / min reducer on a graph with "
                                          "several relations (the reference has none either)")
            dY = tuple(None if (d is None or Y_shape[i] is None) else _reduce_grad(d, Y_shape[i])
                       for i, d in enumerate(dY))
        fix = lambda gs, n: tuple(gs[i] if i < len(gs) else None for i in range(n))
        return (None, None, None, None) + fix(dX, X_len) + fix(dY, Y_len)


class GSDDMM_hetero(torch.autograd.Function):
    """g-SDDMM over every relation, differentiable (sparse.py:506-600): one FFI call forward,
    per-target g-SpMM / g-SDDMM calls over all relations backward."""

    @staticmethod
    def forward(ctx, gidx, op, X_len, lhs_target, rhs_target, *feats):
        out = _gsddmm_hetero(gidx, op, X_len, lhs_target, rhs_target, feats)
        X, Y = feats[:X_len], feats[X_len:]
        ctx.meta = (gidx, op, X_len, len(Y), lhs_target, rhs_target,
                    tuple(None if t is None else t.shape for t in X),
                    tuple(None if t is None else t.shape for t in Y))
        need = lambda ts: any(t is not None and t.requires_grad for t in ts)
        prod = op in ("mul", "dot")
        ctx.save_for_backward(*(X if prod and need(Y) else (None,) * len(X)),
                              *(Y if prod and need(X) else (None,) * len(Y)))
        return tuple(out)

    @staticmethod
    def backward(ctx, *dZ):
        gidx, op, X_len, Y_len, lt, rt, X_shape, Y_shape = ctx.meta
        sv = ctx.saved_tensors
        X, Y = sv[:X_len], sv[X_len:]
        dZ = tuple(None if z is None else z.contiguous() for z in dZ)
        n_nt = gidx.number_of_ntypes()
        none_nt = (None,) * n_nt

        def operand_grad(own_tgt, other_tgt, other, copy_op):
            # d out / d own, summed over the edges (sparse.py:545-600 in hetero form)
            if own_tgt in ("u", "v"):
                g = gidx if own_tgt == "v" else gidx.reverse()
                if op in ("add", copy_op):
                    return gspmm_hetero(g, "copy_rhs", "sum", n_nt, *(none_nt + dZ))
                if other_tgt == "e":
                    prod = tuple(None if (z is None or o is None) else z * o for z, o in zip(dZ, other))
                    return gspmm_hetero(g, "copy_rhs", "sum", n_nt, *(none_nt + prod))
                if other_tgt == own_tgt:
                    summed = gspmm_hetero(g, "copy_rhs", "sum", n_nt, *(none_nt + dZ))
                    return tuple(None if (s_ is None or o is None) else s_ * o for s_, o in zip(summed, other))
                # the other operand sits on the opposite end of the edge: u_mul_e on `g`
                return gspmm_hetero(g, "mul", "sum", len(other), *(tuple(other) + dZ))
            if op in ("add", copy_op):
                return dZ
            return gsddmm_hetero(gidx, "mul", len(dZ), "e", other_tgt, *(dZ + tuple(other)))

        dX, dY = (None,) * X_len, (None,) * Y_len
        if op != "copy_rhs" and any(ctx.needs_input_grad[5:5 + X_len]):
            dX = operand_grad(lt, rt, Y, "copy_lhs")
            dX = tuple(None if (d is None or X_shape[i] is None) else _reduce_grad(d, X_shape[i])
                       for i, d in enumerate(dX[:X_len]))
        if op != "copy_lhs" and any(ctx.needs_input_grad[5 + X_len:]):
            dY = operand_grad(rt, lt, X, "copy_rhs")
            dY = tuple(None if (d is None or Y_shape[i] is None) else _reduce_grad(d, Y_shape[i])
                       for i, d in enumerate(dY[:Y_len]))
        fix = lambda gs, n: tuple(gs[i] if i < len(gs) else None for i in range(n))
        return (None, None, None, None, None) + fix(dX, X_len) + fix(dY, Y_len)


def _merged_softmax_graph(gidx, ets, norm_by):
    """One bipartite relation holding the edges of relations `ets` (which share the node type the
    softmax normalises over) one after the other, cached on the graph index: the fused
    single-relation softmax kernels then normalise over ALL of a node's edges whatever their
    type, which is what EdgeSoftmax_hetero computes with five hetero launches (sparse.py:750-800)."""
    from .graph_index import GraphIndex, Relation

    cache = gidx.__dict__.setdefault("_softmax_merged", {})
    key = (tuple(ets), norm_by)
    if key not in cache:
        rows, cols, offs = [], [], [0]
        for et in ets:
            r, c, old = gidx.relations[et].coo()
            if old is not None:
                raise DGLAMDError("edge_softmax needs COO in edge-id order")
            rows.append(r if norm_by == "dst" else c)
            cols.append(c if norm_by == "dst" else r)
            offs.append(offs[-1] + r.shape[0])
        s, d = gidx.metagraph.find_edge(ets[0])
        nt = d if norm_by == "dst" else s
        # source ids of different relations may collide; they are irrelevant for the softmax
        rel = Relation(int(max(int(x.max()) + 1 if x.numel() else 1 for x in rows)), gidx.num_nodes(nt),
                       torch.cat(rows), torch.cat(cols), idtype=gidx.dtype, device=gidx.ctx)
        cache[key] = (GraphIndex([rel.num_src, rel.num_dst], [(0, 1)], [rel]), offs)
    return cache[key]


class EdgeSoftmax_hetero(torch.autograd.Function):
    """Edge softmax over all relations (sparse.py:750-850): the normaliser of a node runs over
    its incoming (``norm_by='dst'``) or outgoing edges of EVERY type.  Forward and backward are
    the fused kernels on the merged relation of each node type."""

    @staticmethod
    def forward(ctx, gidx, eids, norm_by, *score):
        if eids is not None:
            raise DGLAMDError("eids is not supported on graphs with several relations")
        groups = {}
        for et in range(gidx.number_of_etypes()):
            s, d = gidx.metagraph.find_edge(et)
            if et < len(score) and score[et] is not None:
                groups.setdefault(d if norm_by == "dst" else s, []).append(et)
        outs = [None] * len(score)
        ctx.parts = []
        keep = []
        for nt, ets in groups.items():
            sub, offs = _merged_softmax_graph(gidx, ets, norm_by)
            res = _edge_softmax_forward(sub, torch.cat([score[et] for et in ets]), "copy_rhs")
            ctx.parts.append((sub, ets, offs))
            keep.append(res)
            for i, et in enumerate(ets):
                outs[et] = res[offs[i]:offs[i + 1]]
        ctx.save_for_backward(*keep)
        ctx.n_score = len(score)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_out):
        grads = [None] * ctx.n_score
        for (sub, ets, offs), out in zip(ctx.parts, ctx.saved_tensors):
            g = torch.cat([grad_out[et] if grad_out[et] is not None else
                           torch.zeros_like(out[offs[i]:offs[i + 1]]) for i, et in enumerate(ets)])
            back = _edge_softmax_backward(sub, out, out * g)
            for i, et in enumerate(ets):
                grads[et] = back[offs[i]:offs[i + 1]]
        return (None, None, None) + tuple(grads)


def _neg(ts):
    return [None if t is None else -t for t in ts]


def _inv(ts):
    return [None if t is None else 1.0 / t for t in ts]


def gspmm_hetero(gidx, op, reduce_op, lhs_len, *lhs_and_rhs):
    lhs, rhs = list(lhs_and_rhs[:lhs_len]), list(lhs_and_rhs[lhs_len:])
    if op == "sub":
        op, rhs = "add", _neg(rhs)
    elif op == "div":
        op, rhs = "mul", _inv(rhs)
    feats = _autocast(*[None if t is None else _eo.plain(t) for t in (lhs + rhs)])
    with torch.autocast("cuda", enabled=False):
        return GSpMM_hetero.apply(gidx, op, reduce_op, lhs_len, *feats)


def gsddmm_hetero(gidx, op, lhs_len, lhs_target, rhs_target, *lhs_and_rhs):
    lhs, rhs = list(lhs_and_rhs[:lhs_len]), list(lhs_and_rhs[lhs_len:])
    if op == "sub":
        op, rhs = "add", _neg(rhs)
    elif op == "div":
        op, rhs = "mul", _inv(rhs)
    feats = _autocast(*[None if t is None else _eo.plain(t) for t in (lhs + rhs)])
    with torch.autocast("cuda", enabled=False):
        return GSDDMM_hetero.apply(gidx, op, lhs_len, lhs_target, rhs_target, *feats)


def edge_softmax_hetero(gidx, eids=None, norm_by="dst", *score):
    score = _autocast(*[None if t is None else _eo.plain(t) for t in score])
    with torch.autocast("cuda", enabled=False):
        return EdgeSoftmax_hetero.apply(gidx, eids, norm_by, *score)
