"""Edge tensors kept in CSC POSITION order between the operators of one graph (VERDICT r2 Next #4).

The reference's GAT path is ``apply_edges(u_add_v)`` -> ``leaky_relu`` -> ``edge_softmax`` ->
``update_all(u_mul_e, sum)`` (python/dgl/nn/pytorch/conv/gatconv.py:337-346; five launches for the
softmax alone, python/dgl/backend/pytorch/sparse.py:709-713).  Every one of those passes indexes an
``(E, H)`` tensor through the edge-id map of the in-edge CSR: with DGL's usual edge ids (the order
of the COO the graph was built from) every edge's 32-byte score row is its own access to a 128-byte
line, in both directions (docs/DESIGN_detail_r1_r5.md §3.4: 0.20 of HBM instead of 0.36-0.66 map-free).

Nothing forces the tensor that travels BETWEEN those operators to be in edge-id order: only what
the user reads has to be.  So g-SDDMM / edge softmax hand out a :class:`PosOrdered` tensor — a
``torch.Tensor`` subclass whose storage is in the position order of ``rel``'s in-edge CSR — and
g-SpMM / g-SDDMM / edge softmax take such a tensor without any map (on ``rel`` itself) or through
one composed map (on ``rel.reverse()``, for the backward passes).  Everything else sees an ordinary
edge-id-ordered tensor:

  * order-preserving element-wise functions (``leaky_relu``, ``dropout``, ``exp``, ``* scalar``,
    ``view`` / ``sum`` that keep axis 0, ...) run on the storage as it is and keep the tag;
  * ANY other torch function first converts to edge-id order (a differentiable scatter through the
    map, one pass — what the kernel would have paid anyway), so printing, indexing, ``.cpu()``,
    arithmetic with other tensors, ``edata`` reads by user code all give the reference's values;
  * gradients need no tag: autograd hands a gradient over in the layout of the tensor it belongs
    to, and the Functions below know which layout their inputs / outputs had.

DEFAULT: OFF (VERDICT r4 Weak #1).  ``__torch_function__`` is a whitelist over an open-ended torch
surface and a handful of C entry points never dispatch to it (``torch.autograd.backward``, the legacy
``torch.utils.dlpack.to_dlpack``); a library whose first bar is "results identical to the reference"
must not put that between the user and the values by default.  So every operator returns PLAIN
edge-id-ordered tensors unless the caller opts in, for a region of code it controls::

    with dgl_amd.edge_order_handoff():          # e.g. around a GAT layer's forward
        e = dgl.ops.u_add_v(g, el, er); a = dgl.ops.edge_softmax(g, F.leaky_relu(e)); ...

(also usable as a decorator; ``dgl_amd.set_edge_order_handoff(True)`` / ``DGLA_EDGE_ORDER_HANDOFF=1``
opt in process-wide).  Opting in installs two process-wide shims in front of the C entry points that
bypass ``__torch_function__`` (``torch.autograd.backward``: explicit gradients are the caller's, in
edge-id order; ``torch.utils.dlpack.to_dlpack``: exports edge-id order) — programs that never opt in
never see them.  The scope decides only whether an operator STARTS a hand-off; a tagged tensor that
outlives the scope keeps behaving like its edge-id-ordered values everywhere (tests/test_edge_order_sweep.py
sweeps every overridable torch function for values, gradients and hook payloads).  A graph whose CSC
needs no map (edges already sorted by destination) never uses the mechanism.
Kernels: the same C-ABI seam as everything else (dgla_spmm_csr / dgla_sddmm_coo /
dgla_edge_softmax_* with an explicit CSR / COO whose map is dropped or composed) — no new device code.
"""
import contextlib
import numbers
import os
import threading

import torch
import torch.nn.functional as F

from . import _capi
from ._lib import DGLAMDError

_ENABLED = [os.environ.get("DGLA_EDGE_ORDER_HANDOFF", "0") in ("1", "true", "yes", "on")]
_SCOPE = threading.local()   # .depth > 0 inside `with edge_order_handoff():` (per thread, like grad mode)
MIN_EDGES = 0  # graphs with fewer edges never start a hand-off (tags that arrive are still honoured)


def set_edge_order_handoff(on):
    """Opt the whole process in to / out of the position-ordered hand-off of edge tensors (default OFF:
    every operator returns plain edge-id-ordered tensors, as the reference does)."""
    _ENABLED[0] = bool(on)
    if on:
        _install_global_shims()


def handoff_enabled():
    return _ENABLED[0] or getattr(_SCOPE, "depth", 0) > 0


class edge_order_handoff(contextlib.ContextDecorator):
    """``with dgl_amd.edge_order_handoff():`` — inside the block (this thread) g-SDDMM / edge softmax hand their
    ``(E, ...)`` results over in the CSC position order of the graph (no edge-id map on the way to the next
    operator); outside, operators return plain tensors.  ``edge_order_handoff(False)`` switches a process-wide
    opt-in off for the block."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            _install_global_shims()
            _SCOPE.depth = getattr(_SCOPE, "depth", 0) + 1
        else:
            self._saved = (_ENABLED[0], getattr(_SCOPE, "depth", 0))
            _ENABLED[0], _SCOPE.depth = False, 0
        return self

    def __exit__(self, *exc):
        if self.on:
            _SCOPE.depth -= 1
        else:
            _ENABLED[0], _SCOPE.depth = self._saved
        return False


def _no_tf():
    return torch._C.DisableTorchFunctionSubclass()


def raw(t):
    """The storage of a (possibly tagged) tensor as a plain torch.Tensor, autograd history kept."""
    if type(t) is PosOrdered:
        with _no_tf():
            return t.as_subclass(torch.Tensor)
    return t


class _Tag:
    """The layout tag of ONE storage.  Every alias of the storage (views, ``detach()``, ``.data``)
    holds the same object, so re-laying the storage into edge-id order (``rel = None``) is seen by
    all of them at once."""
    __slots__ = ("rel",)

    def __init__(self, rel):
        self.rel = rel


def tag_of(t):
    return t._dgla_rel if type(t) is PosOrdered else None


def _same_storage(a, b):
    try:
        return a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.numel() > 0
    except RuntimeError:
        return False


def wrap(t, rel, alias_of=None):
    """Tag the plain tensor ``t`` as position-ordered on ``rel``.  ``alias_of``: a tagged tensor ``t``
    was derived from — when both share one storage they share one tag object."""
    _install_global_shims()
    with _no_tf():
        r = t.as_subclass(PosOrdered)
        if alias_of is not None and alias_of._dgla_tag is not None and alias_of._dgla_tag.rel is rel \
                and _same_storage(r, alias_of):
            r._dgla_tag = alias_of._dgla_tag
        else:
            r._dgla_tag = _Tag(rel)
    return r


# ---------------------------------------------------------------------------------------------
# conversion to edge-id order (differentiable)
# ---------------------------------------------------------------------------------------------
def inverse_map(rel):
    """``inv[eid] = CSC position`` of every edge (the inverse of the in-edge CSR's edge-id map), built once per graph.
    A permutation applied as a GATHER through it — ``out_eid[e] = x_pos[inv[e]]`` — costs one scattered 32-byte READ
    per edge at the fabric's request rate; the same permutation as a scatter through the map costs a scattered
    WRITE per edge, more than twice that on this part (62 M edges x 8 heads: 1.2 vs 2.6 ms)."""
    inv = rel.__dict__.get("_inv_csc_map")
    if inv is None:
        m = rel.csc()[2]
        inv = torch.empty_like(m)
        inv[m.long()] = torch.arange(m.numel(), device=m.device, dtype=m.dtype)
        rel.__dict__["_inv_csc_map"] = inv
    return inv


class _ToEid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rel):
        ctx.rel = rel
        m = rel.csc()[2]
        if m is None:
            return x.clone()
        return _capi.gather_rows(x.contiguous(), inverse_map(rel))  # out[e] = x[inv[e]]  (== out[eid(p)] = x[p])

    @staticmethod
    def backward(ctx, g):
        m = ctx.rel.csc()[2]
        if m is None:
            return g, None
        return _capi.gather_rows(g.contiguous(), m), None  # g_pos[p] = g[eid(p)]


class _ToPos(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rel):
        ctx.rel = rel
        m = rel.csc()[2]
        return x.clone() if m is None else _capi.gather_rows(x.contiguous(), m)

    @staticmethod
    def backward(ctx, g):
        m = ctx.rel.csc()[2]
        if m is None:
            return g, None
        return _capi.gather_rows(g.contiguous(), inverse_map(ctx.rel)), None


def to_eid_order(t):
    """A plain tensor in edge-id order with the values of ``t`` (no-op for untagged tensors)."""
    rel = tag_of(t)
    if rel is None:
        return raw(t)
    x = raw(t)
    if x.dim() == 0 or x.shape[0] != rel.num_edges:
        raise DGLAMDError("position-ordered edge tensor with a foreign leading dimension")
    return _ToEid.apply(x, rel)


plain = to_eid_order  # what every entry point that does not understand tags applies to its tensors


# ---------------------------------------------------------------------------------------------
# the tensor subclass
# ---------------------------------------------------------------------------------------------
_META_METHODS = {
    "size", "dim", "ndimension", "numel", "nelement", "element_size", "data_ptr", "is_contiguous", "stride",
    "storage_offset", "is_floating_point", "is_complex", "is_signed", "get_device", "is_pinned", "is_shared",
    "untyped_storage", "__len__", "__hash__", "retain_grad", "is_set_to",
    "_is_view", "is_inference", "is_neg", "is_conj", "has_names", "requires_grad_",
    "register_post_accumulate_grad_hook",   # the hook receives the (tagged) tensor itself
}
_META_PROPS = {
    "shape", "dtype", "device", "requires_grad", "grad_fn", "is_cuda", "is_leaf", "ndim", "layout", "names",
    "is_sparse", "is_quantized", "is_meta", "_version", "is_cpu", "is_mkldnn", "is_xpu", "itemsize", "nbytes",
    "output_nr", "_base", "retains_grad", "is_nested", "is_mps", "is_xla", "is_ipu", "is_maia", "is_mtia",
    "is_sparse_csr", "is_vulkan", "_grad_fn", "volatile", "name", "_backward_hooks", "_post_accumulate_grad_hooks",
}
# element-wise on every row independently: the row order of the storage is irrelevant
_UNARY_KEEP = {
    F.leaky_relu, F.relu, F.elu, F.gelu, F.selu, F.silu, F.softplus, F.dropout, F.sigmoid, F.tanh, F.hardtanh,
    F.relu6, F.celu, F.logsigmoid, F.feature_alpha_dropout, F.alpha_dropout,
    torch.exp, torch.log, torch.sigmoid, torch.tanh, torch.neg, torch.abs, torch.relu, torch.sqrt, torch.rsqrt,
    torch.square, torch.reciprocal, torch.clone, torch.detach, torch.clamp, torch.clamp_min, torch.clamp_max,
    torch.nan_to_num, torch.sign, torch.expm1, torch.log1p, torch.zeros_like, torch.ones_like, torch.empty_like,
    torch.Tensor.exp, torch.Tensor.log, torch.Tensor.sigmoid, torch.Tensor.tanh, torch.Tensor.neg,
    torch.Tensor.__neg__, torch.Tensor.abs, torch.Tensor.__abs__, torch.Tensor.relu, torch.Tensor.sqrt,
    torch.Tensor.rsqrt, torch.Tensor.square, torch.Tensor.reciprocal, torch.Tensor.clone, torch.Tensor.detach,
    torch.Tensor.contiguous, torch.Tensor.clamp, torch.Tensor.clamp_min, torch.Tensor.clamp_max,
    torch.Tensor.float, torch.Tensor.double, torch.Tensor.half, torch.Tensor.bfloat16, torch.Tensor.nan_to_num,
    torch.Tensor.sign, torch.Tensor.expm1, torch.Tensor.log1p, torch.Tensor.__pos__,
}
_BINARY_KEEP = {
    torch.mul, torch.add, torch.sub, torch.div, torch.true_divide, torch.pow, torch.maximum, torch.minimum,
    torch.Tensor.mul, torch.Tensor.add, torch.Tensor.sub, torch.Tensor.div, torch.Tensor.true_divide,
    torch.Tensor.pow, torch.Tensor.maximum, torch.Tensor.minimum,
    torch.Tensor.__mul__, torch.Tensor.__rmul__, torch.Tensor.__add__, torch.Tensor.__radd__,
    torch.Tensor.__sub__, torch.Tensor.__rsub__, torch.Tensor.__truediv__, torch.Tensor.__rtruediv__,
    torch.Tensor.__pow__,
}
# keep the tag when the result still has the E rows on axis 0 (checked on the result)
_SHAPE_KEEP = {
    torch.Tensor.view, torch.Tensor.reshape, torch.reshape, torch.Tensor.unsqueeze, torch.unsqueeze,
    torch.Tensor.squeeze, torch.squeeze, torch.Tensor.flatten, torch.flatten, torch.Tensor.view_as,
    torch.Tensor.reshape_as, torch.Tensor.expand, torch.Tensor.expand_as,
}
_REDUCE_KEEP = {torch.sum, torch.Tensor.sum, torch.mean, torch.Tensor.mean, torch.amax, torch.Tensor.amax,
                torch.amin, torch.Tensor.amin}


def _is_prop_get(func):
    return getattr(func, "__name__", "") == "__get__" and hasattr(func, "__self__")


def _row_broadcastable(other, me):
    """May `other` combine element-wise with the E-row tensor `me` without caring about row order?"""
    if isinstance(other, numbers.Number):
        return True
    if not isinstance(other, torch.Tensor):
        return False
    if type(other) is PosOrdered:
        return tag_of(other) is tag_of(me) and other.dim() >= 1 and other.shape[0] == me.shape[0]
    if other.dim() == 0:
        return True
    if other.dim() < me.dim():
        return True   # broadcast adds leading axes: never touches axis 0 of `me`
    return other.dim() == me.dim() and other.shape[0] == 1 and me.shape[0] != 1


def _reduce_keeps_rows(args, kwargs, me):
    dim = kwargs.get("dim", args[1] if len(args) > 1 else None)
    if dim is None or isinstance(dim, torch.dtype):
        return False
    dims = dim if isinstance(dim, (tuple, list)) else (dim,)
    nd = me.dim()
    return all(isinstance(d, int) and (d % nd) != 0 for d in dims)


_INPLACE_DUNDER = {"__iadd__": "add", "__isub__": "sub", "__imul__": "mul", "__itruediv__": "div",
                   "__ipow__": "pow"}
_UNARY_KEEP_NAMES = {f.__name__ for f in _UNARY_KEEP} - {"clone", "detach", "contiguous", "float", "double", "half",
                                                         "bfloat16", "zeros_like", "ones_like", "empty_like",
                                                         "__neg__", "__abs__", "__pos__", "dropout",
                                                         "feature_alpha_dropout", "alpha_dropout"}
_BINARY_KEEP_NAMES = {"mul", "add", "sub", "div", "true_divide", "pow"}


def _other_tensors(args, kwargs, me):
    return [a for a in _flatten((args, kwargs)) if isinstance(a, torch.Tensor) and a is not me]


def _grads_to_pos(outputs, grads):
    """Explicit gradients (``Tensor.backward(gradient)``, ``autograd.backward(grad_tensors=)``,
    ``autograd.grad(grad_outputs=)``) are written by the caller in EDGE-ID order; the producer of a
    tagged output reads its gradient in the storage layout.  Convert each gradient that belongs to
    a tagged output (a gradient already tagged with the same relation is taken as it is)."""
    if grads is None:
        return None
    single = isinstance(grads, torch.Tensor)
    outs = (outputs,) if isinstance(outputs, torch.Tensor) else tuple(outputs)
    gl = (grads,) if single else tuple(grads)
    res = []
    for o, g in zip(outs, gl):
        rel = tag_of(o)
        if g is None or rel is None:
            res.append(raw(g) if g is not None and tag_of(g) is None else
                       (to_eid_order(g) if g is not None else None))
        elif tag_of(g) is rel:
            res.append(raw(g))
        else:
            gp = to_eid_order(g)
            if gp.dim() == 0 or gp.shape[0] != rel.num_edges:
                gp = gp.expand(raw(o).shape)
            res.append(_ToPos.apply(gp, rel))
    res.extend(raw(g) if isinstance(g, torch.Tensor) else g for g in gl[len(outs):])
    return res[0] if single else tuple(res)


_BACKWARD_HOOKED = [False]


def _hook_in_eid_order(tag, hook):
    """``tagged.register_hook(hook)``: autograd hands the gradient over in the layout of the tensor it belongs to
    (position order).  The user's hook is written against the reference's tensor: it gets the gradient in EDGE-ID
    order (a plain tensor), and a gradient it returns is taken in edge-id order and put back into the layout."""
    def in_eid_order(g):
        rel = tag.rel
        if rel is None or g is None:          # the storage was re-laid into edge-id order since
            return hook(g)
        r = hook(_ToEid.apply(raw(g), rel))
        if r is None:
            return None
        if tag_of(r) is rel:
            return raw(r)
        r = to_eid_order(r)
        return _ToPos.apply(r if r.shape == g.shape else r.expand(g.shape), rel)

    in_eid_order.__wrapped__ = hook
    return in_eid_order


def _install_global_shims():
    """Two C entry points never dispatch to ``__torch_function__``: ``torch.autograd.backward`` (``autograd.grad`` and
    ``Tensor.backward`` do) and the legacy ``torch.utils.dlpack.to_dlpack`` (``torch.from_dlpack`` /
    ``Tensor.__dlpack__`` do).  A shim is put in front of each the first time the hand-off is opted in to: explicit
    ``grad_tensors`` are converted to the layout of the tagged output they belong to, and a tagged tensor is exported
    in edge-id order (what the reference's own backend hands to DLPack, python/dgl/backend/pytorch/tensor.py:432-435).
    Programs that never opt in never see either."""
    if _BACKWARD_HOOKED[0]:
        return
    _BACKWARD_HOOKED[0] = True
    import torch.utils.dlpack as _dlpack

    inner_dl = _dlpack.to_dlpack

    def to_dlpack(tensor, *args, **kwargs):
        return inner_dl(to_eid_order(tensor) if tag_of(tensor) is not None else raw(tensor), *args, **kwargs)

    to_dlpack.__doc__ = inner_dl.__doc__
    to_dlpack.__wrapped__ = inner_dl
    _dlpack.to_dlpack = to_dlpack
    if getattr(torch, "to_dlpack", None) is inner_dl:
        torch.to_dlpack = to_dlpack
    inner = torch.autograd.backward

    def backward(tensors, grad_tensors=None, *args, **kwargs):
        if grad_tensors is not None and any(tag_of(t) is not None for t in _flatten((tensors,))):
            grad_tensors = _grads_to_pos(tensors, grad_tensors)
        return inner(tensors, grad_tensors, *args, **kwargs)

    backward.__doc__ = inner.__doc__
    backward.__wrapped__ = inner
    torch.autograd.backward = backward


class PosOrdered(torch.Tensor):
    """An ``(E, ...)`` edge tensor whose rows are stored in the position order of the in-edge CSR
    of ``_dgla_rel`` instead of edge-id order.  Behaves like the edge-id-ordered tensor under every
    torch function (see the module docstring)."""

    _dgla_tag = None

    @property
    def _dgla_rel(self):
        tag = self._dgla_tag
        return None if tag is None else tag.rel

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        tagged = [a for a in _flatten((args, kwargs)) if type(a) is PosOrdered and a._dgla_rel is not None]
        if not tagged:
            with _no_tf():
                return func(*args, **kwargs)
        me = tagged[0]
        name = getattr(func, "__name__", "")
        # -- metadata: no values involved -----------------------------------------------------
        if _is_prop_get(func):
            pname = getattr(func.__self__, "__name__", "")
            if pname in _META_PROPS:
                with _no_tf():
                    return func(*args, **kwargs)
            if pname in ("data", "grad", "_grad"):  # tensors with the storage layout of `me`
                with _no_tf():
                    r = func(*args, **kwargs)
                return wrap(r, me._dgla_rel, alias_of=me) if isinstance(r, torch.Tensor) else r
        elif name in _META_METHODS:
            with _no_tf():
                return func(*args, **kwargs)
        elif name == "register_hook" and func is torch.Tensor.register_hook:
            hook = kwargs["hook"] if "hook" in kwargs else args[1]
            with _no_tf():
                return func(args[0], _hook_in_eid_order(args[0]._dgla_tag, hook))
        elif func is torch.autograd.grad:
            # gradients arrive in the layout of the tensor they belong to: tag those of tagged inputs;
            # explicit grad_outputs are the caller's (edge-id order): convert those of tagged outputs
            args, kwargs = list(args), dict(kwargs)
            outputs = kwargs["outputs"] if "outputs" in kwargs else args[0]
            if "grad_outputs" in kwargs:
                kwargs["grad_outputs"] = _grads_to_pos(outputs, kwargs["grad_outputs"])
            elif len(args) > 2:
                args[2] = _grads_to_pos(outputs, args[2])
            with _no_tf():
                res = func(*args, **kwargs)
            inputs = kwargs.get("inputs", args[1] if len(args) > 1 else None)
            inputs = (inputs,) if isinstance(inputs, torch.Tensor) else tuple(inputs)
            return tuple(wrap(r, i._dgla_rel) if (r is not None and type(i) is PosOrdered and i._dgla_rel is not None)
                         else r for r, i in zip(res, inputs))
        elif name == "backward":
            # Tensor.backward(self, gradient) ends in torch.autograd.backward, which converts the
            # explicit gradient (_install_backward_hook)
            with _no_tf():
                return func(*args, **kwargs)
        # -- order-preserving functions: run on the storage, keep the tag -----------------------
        keep = False
        if name == "type" and func is torch.Tensor.type:
            # t.type() is metadata; t.type(dtype) is a cast like .float(): same rows, keep the tag
            if len(args) == 1 and not kwargs:
                with _no_tf():
                    return func(*args, **kwargs)
            keep = len(tagged) == 1 and args[0] is me
        elif func in _UNARY_KEEP:
            # bounds / fill values given as tensors (torch.clamp(t, min=per_edge)) must not care about
            # the row order either
            keep = len(tagged) == 1 and args and args[0] is me and "out" not in kwargs \
                and all(_row_broadcastable(o, me) for o in _other_tensors(args, kwargs, me))
        elif func in _BINARY_KEEP:
            others = [a for a in args[:2] if a is not me]
            keep = len(args) >= 2 and all(_row_broadcastable(o, me) for o in others) and "out" not in kwargs \
                and all(k in ("alpha", "rounding_mode") for k in kwargs)
        elif func in _REDUCE_KEEP:
            keep = len(tagged) == 1 and args and args[0] is me and _reduce_keeps_rows(args, kwargs, me)
        if keep:
            with _no_tf():
                res = func(*_untag(args), **_untag(kwargs))
            if isinstance(res, torch.Tensor) and res.dim() >= 1 and res.shape[0] == me.shape[0]:
                return me if res is raw(me) else wrap(res, me._dgla_rel, alias_of=me)
            # (requires_grad_ and friends return their argument)
            return res
        if func in _SHAPE_KEEP and len(tagged) == 1 and args and args[0] is me:
            with _no_tf():
                res = func(*_untag(args), **_untag(kwargs))
            if isinstance(res, torch.Tensor) and res.dim() >= 1 and res.shape[0] == me.shape[0] \
                    and me.dim() >= 1 and (me.is_contiguous() or res.dim() == me.dim()):
                return wrap(res, me._dgla_rel, alias_of=me)
            # axis 0 changed: fall through and redo it on the edge-id-ordered values
        # -- in-place functions ------------------------------------------------------------------------
        out_targets = [o for o in _flatten(kwargs.get("out", ())) if isinstance(o, torch.Tensor)]
        inplace = (name.endswith("_") and not name.endswith("__")) or name == "__setitem__" \
            or name in _INPLACE_DUNDER or bool(out_targets)
        if inplace:
            # the tensors WRITTEN: args[0] of an in-place method, or the out= targets; everything else
            # is only read and goes in as an edge-id-ordered copy (its own storage is left alone)
            if out_targets:
                written = out_targets
            elif args:
                written = [args[0]] if isinstance(args[0], torch.Tensor) else []
            else:   # python-level in-place functions dispatch with keywords only (torch.nn.init.constant_(tensor=..))
                written = [v for k, v in kwargs.items() if k in ("tensor", "input", "self") and isinstance(v, torch.Tensor)][:1]
            tgt = written[0] if len(written) == 1 and not out_targets and tag_of(written[0]) is not None else None
            if tgt is not None and name == "detach_":
                # the tagged object is an alias (a view) of its storage: torch refuses detach_ on views
                if raw(tgt).requires_grad:
                    raise DGLAMDError("detach_() on a position-ordered edge tensor inside an autograd graph; use detach()")
                return tgt
            if tgt is not None and args:
                base = _INPLACE_DUNDER.get(name, name[:-1] if name.endswith("_") else name)
                rest = _other_tensors(args[1:], kwargs, tgt)
                if (base in _UNARY_KEEP_NAMES or base in _BINARY_KEEP_NAMES) \
                        and all(_row_broadcastable(o, tgt) for o in rest):
                    # order-preserving: run on the storage as it is, the tag (of every alias) stays.  The function
                    # is applied to the user's OWN tensor object (not to an alias of it), so hooks registered on it
                    # before / after behave as they do on a plain tensor
                    with _no_tf():
                        if tgt.requires_grad and tgt._backward_hooks:
                            # the tagged object is a VIEW of its producer's output: torch re-bases a view's grad_fn
                            # on an in-place write and hooks registered before it would silently stop firing
                            raise DGLAMDError("in-place modification of a position-ordered edge tensor that has "
                                              "gradient hooks registered; use the out-of-place form")
                        func(tgt, *_untag(args[1:]), **_untag(kwargs))
                    return tgt
                if base == "copy" and len(args) >= 2 and isinstance(args[1], torch.Tensor) and not tgt.requires_grad:
                    # t.copy_(src): src's rows into position order, the tag stays
                    src, rel = args[1], tag_of(tgt)
                    if tag_of(src) is rel or _row_broadcastable(src, tgt):
                        s_pos = raw(src)
                    else:
                        s_eid = to_eid_order(src)
                        s_pos = _ToPos.apply(s_eid.expand(raw(tgt).shape) if s_eid.shape != raw(tgt).shape else s_eid, rel)
                    with _no_tf():
                        func(raw(tgt), s_pos, *args[2:], **kwargs)
                    return tgt
            for t in written:
                if tag_of(t) is not None:
                    _untag_in_place(t)
            conv = {id(t): to_eid_order(t) for t in tagged if tag_of(t) is not None}
            with _no_tf():
                return func(*_map(args, conv), **_map(kwargs, conv))
        # -- everything else sees edge-id order -----------------------------------------------------
        conv = {id(t): to_eid_order(t) for t in tagged}
        with _no_tf():
            return func(*_map(args, conv), **_map(kwargs, conv))

    def eid_order(self):
        """This tensor's values as a plain edge-id-ordered tensor."""
        return to_eid_order(self)

    def __reduce_ex__(self, proto):  # pickling / torch.save store the reference's layout
        return to_eid_order(self).__reduce_ex__(proto)


def _flatten(x):
    if isinstance(x, (list, tuple)):
        for y in x:
            yield from _flatten(y)
    elif isinstance(x, dict):
        for y in x.values():
            yield from _flatten(y)
    else:
        yield x


def _map(x, conv):
    if isinstance(x, tuple):
        return tuple(_map(y, conv) for y in x)
    if isinstance(x, list):
        return [_map(y, conv) for y in x]
    if isinstance(x, dict):
        return {k: _map(v, conv) for k, v in x.items()}
    if type(x) is PosOrdered:
        if id(x) in conv:
            return conv[id(x)]
        # no layout tag (any more): the object itself goes in — in-place metadata functions (t_(), unsqueeze_())
        # must act on the tensor the user holds, not on an alias of it
        return x if x._dgla_rel is None else raw(x)
    return x


def _untag(x):
    if isinstance(x, tuple):
        return tuple(_untag(y) for y in x)
    if isinstance(x, list):
        return [_untag(y) for y in x]
    if isinstance(x, dict):
        return {k: _untag(v) for k, v in x.items()}
    return raw(x)


def _untag_in_place(t):
    """Re-lay the storage of `t` into edge-id order and drop the tag, so that an in-place function
    modifies the tensor the user thinks it has.  Refused for tensors inside an autograd graph: the
    producer's backward expects its output layout."""
    rel = t._dgla_rel
    if rel is None:
        return
    x = raw(t)
    if x.requires_grad and torch.is_grad_enabled():
        raise DGLAMDError("in-place modification of a position-ordered edge tensor that is part of an autograd "
                          "graph; take t.eid_order() first or call dgl_amd.set_edge_order_handoff(False)")
    with torch.no_grad():
        x.copy_(to_eid_order(t).detach())
    # one tag object per storage: every alias (views, detach(), .data) is edge-id ordered from here on
    t._dgla_tag.rel = None


def reject_tagged(t):
    """Guard of the kernel-facing layers (dgl_amd._ffi.NDArray, dgl_amd._capi._tensor): a tagged
    tensor must never reach a kernel that does not know its layout."""
    if type(t) is PosOrdered and t._dgla_rel is not None:
        raise DGLAMDError("internal error: a position-ordered edge tensor reached a kernel entry that is not "
                          "layout-aware (dgl_amd.edge_order); please report, and set DGLA_EDGE_ORDER_HANDOFF=0")


# ---------------------------------------------------------------------------------------------
# kernels in position space (graph-free C-ABI seam with explicit CSR / COO)
# ---------------------------------------------------------------------------------------------
_TARGET = {"u": 0, "e": 1, "v": 2}


def wants_handoff(rel):
    """Should an operator on `rel` START a hand-off (produce a tagged tensor)?"""
    if not handoff_enabled() or rel.transient or not rel.allowed("csc") or rel.num_edges < max(MIN_EDGES, 1):
        return False
    if not rel.device.type == "cuda":
        return False
    return rel.csc()[2] is not None   # map-free graphs have nothing to gain


def _ctx(rel):
    c = rel.__dict__.get("_pos_ctx")
    if c is None:
        indptr, indices, _ = rel.csc()
        deg = (indptr[1:] - indptr[:-1]).long()
        dst = torch.repeat_interleave(torch.arange(rel.num_dst, device=rel.device, dtype=rel.idtype), deg,
                                      output_size=int(indices.shape[0]))
        c = rel.__dict__["_pos_ctx"] = {
            "csr": _capi.make_csr(indptr, indices, None, rel.num_src),
            "coo": _capi.make_coo(indices, dst, None, rel.num_src, rel.num_dst),  # edges listed in CSC position order
            "ws": {}, "esm": {}, "rev": None}
    return c


def _rev_csr(rel):
    """In-edge CSR of ``rel.reverse()`` (= out-edge CSR of ``rel``) whose edge map leads to the
    CSC POSITIONS of ``rel`` instead of edge ids: position p' of the reverse graph reads row
    ``inv_csc[eid(p')]`` of a tensor laid out in rel's CSC order."""
    c = _ctx(rel)
    if c["rev"] is None:
        ip, ix, m_csr = rel.csr()
        m_csc = rel.csc()[2]
        if m_csc is None:
            comp = m_csr
        else:
            inv = torch.empty_like(m_csc)
            inv[m_csc.long()] = torch.arange(m_csc.numel(), device=m_csc.device, dtype=m_csc.dtype)
            comp = inv if m_csr is None else inv[m_csr.long()].contiguous()
        c["rev"] = {"csr": _capi.make_csr(ip, ix, comp, rel.num_dst), "ws": {}}
    return c["rev"]


def spmm_pos(rel, on_reverse, op, u, e, v_rows):
    """sum-reducing g-SpMM on ``rel`` (or ``rel.reverse()``) whose edge operand ``e`` (raw) is in
    rel's CSC position order."""
    from .sparse_kernels import infer_broadcast_shape

    holder = _rev_csr(rel) if on_reverse else _ctx(rel)
    use_u, use_e = op != "copy_rhs", op != "copy_lhs"
    ref = u if use_u else e
    u_shp = tuple(u.shape) if use_u else (0,)
    e_shp = tuple(e.shape) if use_e else (0,)
    out = torch.empty((v_rows,) + infer_broadcast_shape(op, u_shp[1:], e_shp[1:]), dtype=ref.dtype, device=ref.device)
    if rel.num_edges == 0 or out.numel() == 0:
        return out.zero_()
    uu = u.contiguous() if use_u else None
    ee = e.contiguous() if use_e else None
    key = (op, ref.dtype, u_shp[1:], e_shp[1:], _capi.get_tuning())
    ent = holder["ws"].get(key)
    if ent is None:
        nbytes = _capi.spmm_csr_workspace_bytes(op, "sum", holder["csr"], ref.dtype, uu, ee, out)
        ent = holder["ws"][key] = [torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=ref.device), False]
    _capi.spmm_csr(op, "sum", holder["csr"], uu, ee, out, None, None, ent[0], plan_valid=ent[1])
    ent[1] = True
    return out


def sddmm_pos(rel, op, lhs, rhs, lhs_target, rhs_target):
    """g-SDDMM over rel's edges LISTED IN CSC POSITION ORDER: 'e' operands and the output are
    position-ordered raw tensors."""
    from .sparse_kernels import infer_broadcast_shape

    c = _ctx(rel)
    use_l, use_r = op != "copy_rhs", op != "copy_lhs"
    ref = lhs if use_l else rhs
    l_shp = tuple(lhs.shape) if use_l else (0,)
    r_shp = tuple(rhs.shape) if use_r else (0,)
    out = torch.empty((rel.num_edges,) + infer_broadcast_shape(op, l_shp[1:], r_shp[1:]), dtype=ref.dtype,
                      device=ref.device)
    if rel.num_edges > 0 and out.numel() > 0:
        _capi.sddmm_coo(op, c["coo"], lhs.contiguous() if use_l else None, rhs.contiguous() if use_r else None, out,
                        _TARGET[lhs_target], _TARGET[rhs_target])
    return out


def _esm_ws(rel, t):
    c = _ctx(rel)
    dim = 1
    for d in t.shape[1:]:
        dim *= int(d)
    key = (dim, t.dtype)
    ent = c["esm"].get(key)
    if ent is None:
        nbytes = _capi.edge_softmax_workspace_bytes(c["csr"], t.dtype, dim)
        ent = c["esm"][key] = [torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=t.device), False]
    return c["csr"], ent


def softmax_pos_forward(rel, score):
    out = torch.empty_like(score)
    if rel.num_edges and score.numel():
        csr, ent = _esm_ws(rel, score)
        _capi.edge_softmax_forward(csr, score, out, ent[0], plan_valid=ent[1])
        ent[1] = True
    return out


def softmax_gather_pos_forward(rel, score, edge_map):
    """Edge softmax of EDGE-ID-ordered scores, result in position order, in ONE pass: the merge-path kernel reads
    through the CSC's edge-id map and writes by position (DGLA_ESM_OUT_POSITION) — instead of a gather pass in front
    of the map-free softmax.  None when the shape has no merge-path kernel (long feature rows): the caller
    falls back to gather + softmax."""
    dim = 1
    for d in score.shape[1:]:
        dim *= int(d)
    if not rel.num_edges or not score.numel():
        return None
    c = _ctx(rel)
    if not int(_capi.edge_softmax_workspace_bytes(c["csr"], score.dtype, dim)):
        return None          # no merge-path kernel for this (feature length, dtype)
    if "csr_map" not in c:
        indptr, indices, _ = rel.csc()
        c["csr_map"] = _capi.make_csr(indptr, indices, edge_map, rel.num_src)
    _, ent = _esm_ws(rel, score)          # the plan depends on indptr only: shared with the map-free calls
    out = torch.empty_like(score)
    _capi.edge_softmax_forward(c["csr_map"], score, out, ent[0], plan_valid=ent[1], out_position=True)
    ent[1] = True
    return out


# Below this many edges the plain edge softmax reads AND writes through the map in one kernel: the extra gather pass of
# the position route is a launch of its own, and at GAT-on-arxiv size (2.5 M edges) launches are what a step is made
# of (forward 0.110 ms in one kernel against 0.163 ms as kernel + gather; at 62 M edges 4.58 against 3.8 ms).
PLAIN_SOFTMAX_POS_MIN_EDGES = 1 << 23


def plain_softmax_route(rel, score):
    """May a PLAIN edge softmax on ``rel`` keep its result in position order internally?  (A CSC with an edge-id
    map, built once and kept: not a sampled block; a feature length the merge-path kernels take; enough edges for
    the scattered write it saves to outweigh the gather launch it adds.)"""
    if rel.transient or not rel.allowed("csc") or not score.is_cuda or rel.num_edges < max(PLAIN_SOFTMAX_POS_MIN_EDGES, 1) \
            or score.numel() == 0:
        return False
    if rel.csc()[2] is None:
        return False
    dim = 1
    for d in score.shape[1:]:
        dim *= int(d)
    return bool(int(_capi.edge_softmax_workspace_bytes(_ctx(rel)["csr"], score.dtype, dim)))


def softmax_pos_backward_from_eid_grad(rel, out_pos, grad_eid):
    """Backward of the edge softmax with the saved softmax in POSITION order and the caller's gradient in EDGE-ID
    order: the kernel streams ``out_pos``, reads ``grad_eid`` through the map, forms ``out * g`` itself and writes the
    result by position (DGLA_ESM_OUT_POSITION | DGLA_ESM_B_IS_GRAD) — one scattered read per edge in the pass."""
    c = _ctx(rel)
    if "csr_map" not in c:
        indptr, indices, m = rel.csc()
        c["csr_map"] = _capi.make_csr(indptr, indices, m, rel.num_src)
    _, ent = _esm_ws(rel, out_pos)
    back = torch.empty_like(out_pos)
    _capi.edge_softmax_backward(c["csr_map"], out_pos, grad_eid.contiguous(), back, ent[0], plan_valid=ent[1],
                                sds_is_grad=True, out_position=True)
    ent[1] = True
    return back


def softmax_pos_backward(rel, out, sds):
    back = torch.empty_like(out)
    if rel.num_edges and out.numel():
        csr, ent = _esm_ws(rel, out)
        _capi.edge_softmax_backward(csr, out, sds, back, ent[0], plan_valid=ent[1])
        ent[1] = True
    return back


def softmax_pos_backward_from_grad(rel, out, grad):
    """grad_score = out * grad - out * sum_row(out * grad) from the UPSTREAM gradient: the merge-path kernel forms
    the product itself (DGLA_ESM_B_IS_GRAD; same bits as multiplying first, one elementwise pass less); shapes
    without a merge-path kernel multiply first."""
    if not rel.num_edges or not out.numel():
        return torch.empty_like(out)
    dim = 1
    for d in out.shape[1:]:
        dim *= int(d)
    c = _ctx(rel)
    if not int(_capi.edge_softmax_workspace_bytes(c["csr"], out.dtype, dim)):
        return softmax_pos_backward(rel, out, (out * grad).contiguous())
    back = torch.empty_like(out)
    csr, ent = _esm_ws(rel, out)
    _capi.edge_softmax_backward(csr, out, grad.contiguous(), back, ent[0], plan_valid=ent[1], sds_is_grad=True)
    ent[1] = True
    return back


# ---------------------------------------------------------------------------------------------
# differentiable operators in position space
# ---------------------------------------------------------------------------------------------
def _unsq(t):
    return (t.unsqueeze(-1), True) if t is not None and t.dim() == 1 else (t, False)


class PosGSDDMM(torch.autograd.Function):
    """g-SDDMM on a single relation whose result (and any 'e'-target operand) is position-ordered.
    Same gradient routing as autograd.GSDDMM (sparse.py:459-503)."""

    @staticmethod
    def forward(ctx, gidx, op, X, Y, lhs_target, rhs_target):
        rel = gidx.relations[0]
        x, ex = _unsq(raw(X) if X is not None else None)
        y, ey = _unsq(raw(Y) if Y is not None else None)
        out = sddmm_pos(rel, op, x, y, lhs_target, rhs_target)
        if (ex or x is None) and (ey or y is None):
            out = out.squeeze(-1)
        ctx.meta = (gidx, op, lhs_target, rhs_target, None if X is None else X.shape, None if Y is None else Y.shape)
        need_dx = X is not None and X.requires_grad
        need_dy = Y is not None and Y.requires_grad
        prod = op in ("mul", "dot")
        ctx.save_for_backward(raw(X) if prod and need_dy else None, raw(Y) if prod and need_dx else None)
        return wrap(out, rel)

    @staticmethod
    def backward(ctx, dZ):
        from .autograd import _reduce_grad, gsddmm, gspmm

        gidx, op, lt, rt, x_shape, y_shape = ctx.meta
        rel = gidx.relations[0]
        X, Y = ctx.saved_tensors
        dZ = raw(dZ).contiguous()
        dZt = wrap(dZ, rel)
        tag_e = lambda t, tgt: wrap(t, rel) if (tgt == "e" and t is not None) else t

        def operand_grad(own_tgt, other_tgt, other, copy_op):
            if own_tgt in ("u", "v"):
                g = gidx if own_tgt == "v" else gidx.reverse()
                if op in ("add", copy_op):
                    return gspmm(g, "copy_rhs", "sum", None, dZt)
                if other_tgt == own_tgt:
                    return gspmm(g, "copy_rhs", "sum", None, dZt) * other
                if other_tgt == "e":
                    return gspmm(g, "copy_rhs", "sum", None, wrap(dZ * other, rel))
                return gspmm(g, "mul", "sum", other, dZt)
            if op in ("add", copy_op):
                return dZ
            return raw(gsddmm(gidx, "mul", dZt, tag_e(other, other_tgt), "e", other_tgt))

        dX = dY = None
        if op != "copy_rhs" and ctx.needs_input_grad[2]:
            dX = _reduce_grad(raw(operand_grad(lt, rt, Y, "copy_lhs")), x_shape)
        if op != "copy_lhs" and ctx.needs_input_grad[3]:
            dY = _reduce_grad(raw(operand_grad(rt, lt, X, "copy_rhs")), y_shape)
        return None, None, dX, dY, None, None


class PosGSpMM(torch.autograd.Function):
    """sum-reducing g-SpMM on ``Q = gidx.relations[0]`` whose edge operand ``Y`` is stored in the CSC
    position order of ``rel`` (``rel is Q`` or ``rel is Q.reverse()``).  Gradient routing of
    autograd.GSpMM (sparse.py:162-248); the gradient w.r.t. ``Y`` comes out position-ordered."""

    @staticmethod
    def forward(ctx, gidx, op, X, Y, rel):
        from .autograd import _last_dim_is_reduced

        Q = gidx.relations[0]
        on_rev = Q is not rel
        x, ex = _unsq(X)
        y, ey = _unsq(raw(Y))
        _, d = gidx.metagraph.find_edge(0)
        out = spmm_pos(rel, on_rev, op, x, y, gidx.num_nodes(d))
        if (ex or x is None) and ey:
            out = out.squeeze(-1)
        ctx.meta = (gidx, op, rel, on_rev, None if X is None else X.shape, Y.shape, _last_dim_is_reduced(X, raw(Y)))
        need_dx = X is not None and X.requires_grad
        need_dy = Y.requires_grad
        ctx.save_for_backward(X if (op == "mul" and need_dy) else None,
                              raw(Y) if (op == "mul" and need_dx) else None)
        return out

    @staticmethod
    def backward(ctx, dZ):
        from .autograd import _reduce_grad, gspmm

        gidx, op, rel, on_rev, x_shape, y_shape, reduce_last = ctx.meta
        X, Y = ctx.saved_tensors
        dZ = dZ.contiguous()
        dX = dY = None
        if op != "copy_rhs" and ctx.needs_input_grad[2]:
            rev = gidx.reverse()
            if op == "mul":
                dX = gspmm(rev, "mul", "sum", dZ, wrap(Y, rel))
            else:
                dX = gspmm(rev, "copy_lhs", "sum", dZ, None)
            dX = _reduce_grad(dX, x_shape)
        if op != "copy_lhs" and ctx.needs_input_grad[3]:
            # edge (s, d) of rel is edge (d, s) of rel.reverse(): X sits on Q's source side
            xt, zt = ("v", "u") if on_rev else ("u", "v")
            dz, _ = _unsq(dZ)
            if op == "mul":
                xx, _ = _unsq(X)
                dY = sddmm_pos(rel, "dot" if reduce_last else "mul", xx, dz, xt, zt)
            else:
                dY = sddmm_pos(rel, "copy_rhs", None, dz, xt, zt)
            dY = _reduce_grad(dY, y_shape)
        return None, None, dX, dY, None


class PosEdgeSoftmax(torch.autograd.Function):
    """Edge softmax over incoming edges; the result is position-ordered.  ``score`` is either a
    tagged tensor of the same relation or a plain edge-id-ordered one (gathered once on the way in;
    its gradient scattered once on the way out)."""

    @staticmethod
    def forward(ctx, gidx, score):
        rel = gidx.relations[0]
        tagged = tag_of(score) is rel
        s = raw(score)
        s, expand = _unsq(s)
        s = s.contiguous()
        out = None
        if not tagged:
            m = rel.csc()[2]
            if m is not None:
                out = softmax_gather_pos_forward(rel, s, m)
                if out is None:
                    s = _capi.gather_rows(s, m)
        if out is None:
            out = softmax_pos_forward(rel, s)
        ctx.rel, ctx.tagged, ctx.expand = rel, tagged, expand
        ctx.save_for_backward(out)
        return wrap(out.squeeze(-1) if expand else out, rel)

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        rel = ctx.rel
        g = raw(grad_out)
        if ctx.expand:
            g = g.unsqueeze(-1)
        back = softmax_pos_backward_from_grad(rel, out, g.to(out.dtype))
        if not ctx.tagged:
            if rel.csc()[2] is not None:          # the score came in edge-id order: so does its gradient
                back = _capi.gather_rows(back, inverse_map(rel))
        return None, (back.squeeze(-1) if ctx.expand else back)
