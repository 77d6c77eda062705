"""Shape inference, output allocation and the FFI call sites of the hot path.

Host-side mirror of python/dgl/_sparse_ops.py (``_gspmm`` :156-265, ``_gsddmm`` :482-565,
hetero variants :268-433,568-638, ``_edge_softmax_forward/backward`` :720-800): same
argument meaning, dtype rules, 1-D feature handling, return values and error messages —
the arithmetic happens in libdgl_amd.so behind ``sparse._CAPI_DGLKernel*``.
"""
import itertools
import os
import weakref

import torch

from . import _ffi
from ._lib import LIB as _LIB, DGLAMDError

_TARGET = {"u": 0, "e": 1, "v": 2, 0: 0, 1: 1, 2: 2}

# ---- static source features --------------------------------------------------------------
# When a feature row is not a whole number of 128-byte cache lines (F = 100 fp32) the CSR SpMM
# gathers faster from a line-aligned "split-row" copy of the features (csrc/spmm_csr.hip.h); by
# default that copy is re-made on every call, because the library cannot know whether a tensor
# changed between two calls.  A caller that KNOWS its features are static (input features of
# full-graph training / inference) says so once, and the copy is made once per graph.
_static = {}             # id(tensor) -> (weakref to the tensor, token)
_tokens = itertools.count(1)


def _fingerprint(t):
    # what must stay the same for a kept copy to still describe `t`: torch's in-place version counter
    # (bumped by every in-place op, optimizer steps included), storage address, shape and strides
    return (t._version, t.data_ptr(), tuple(t.shape), tuple(t.stride()))


def static_features(t):
    """Declare ``t`` (a source-node OR edge feature tensor) unchanging for as long as it lives.
    g-SpMM calls that read it as the node operand keep its split-row copy between calls instead
    of re-making it; sum-reducing calls that read it as the EDGE operand on a graph whose CSC has
    an edge-id map keep a copy in CSC position order and run map-free (GCN-style normalisation
    weights: one random 128-byte line per 4-byte weight otherwise).  The promise is CHECKED at
    every use (one host compare of ``t._version``, ``data_ptr()``, shape, strides): an in-place
    write into ``t`` afterwards withdraws it — that call and later ones re-read ``t`` on the plain
    path, like the reference does on every call (python/dgl/_sparse_ops.py:156-265) — until
    ``static_features(t)`` is called again.  ``release_static(t)`` takes the promise back.
    Returns ``t``."""
    key = id(t)
    _static[key] = (weakref.ref(t, lambda _r, k=key: _static.pop(k, None)), next(_tokens), _fingerprint(t))
    return t


def release_static(t):
    _static.pop(id(t), None)


def set_auto_edge_operand(min_edges):
    """OPT-IN: on graphs with at least ``min_edges`` edges keep a CSC-position-ordered copy of narrow
    (<= 16 bytes per edge) edge operands of sum-reducing g-SpMM **by content hash** (no announcement
    needed; csrc/ffi_registry.hip ``auto_edge_operand``).  ``None`` / a negative value switches it
    off, which is the default: the stale-copy test is a 128-bit non-cryptographic hash compared on
    the device — a probabilistic shortcut — and the copy keeps 16 bytes per edge of device memory
    for the life of the graph.  ``static_features(w)`` gives the same speed without a hash."""
    from . import _ffi

    _ffi.get_global_func("dgl_amd._CAPI_SetAutoEdgeOperandMinEdges")(-1 if min_edges is None else int(min_edges))


def _static_token(t):
    ent = _static.get(id(t))
    if ent is None or ent[0]() is not t:
        return 0
    if ent[2] != _fingerprint(t):       # written in place (or re-pointed) since the announcement: the kept copy is stale
        _static.pop(id(t), None)
        return 0
    return ent[1]


def infer_broadcast_shape(op, shp1, shp2):
    """Feature shape of ``op(lhs, rhs)`` (python/dgl/_sparse_ops.py:10-60)."""
    shp1, shp2 = tuple(shp1), tuple(shp2)
    if op == "copy_lhs":
        return shp1
    if op == "copy_rhs":
        return shp2
    n = max(len(shp1), len(shp2))
    a, b = (1,) * (n - len(shp1)) + shp1, (1,) * (n - len(shp2)) + shp2
    for x, y in zip(a, b):
        if x != y and x != 1 and y != 1:
            raise DGLAMDError("Feature shapes {} and {} are not valid for broadcasting.".format(shp1, shp2))
    out = tuple(max(x, y) for x, y in zip(a, b))
    return out[:-1] + (1,) if op == "dot" else out


def _nd(t):
    return None if t is None else _ffi.NDArray(t, unsqueeze=True)


def _check_pair(u, e, use_u, use_e, what):
    if use_u and use_e and u.dtype != e.dtype:
        raise DGLAMDError(
            "The node features' data type {} doesn't match edge features' data type {}, "
            "please convert them to the same type.".format(u.dtype, e.dtype) if what == "spmm" else
            "The left operand's data type {} doesn't match the right operand's data type {}, "
            "please convert them to the same type.".format(u.dtype, e.dtype))


def _capi_tuning():
    # the library's tuning bits read at call time (a host read of one word): raw
    # LIB.dgla_set_tuning calls change the split-row bit without going through
    # dgl_amd._capi.set_tuning, and a scratch size cached under the old bits could be too small
    return _LIB.dgla_get_tuning()


_tuning_epoch = [0]  # kept for callers of dgl_amd._capi.set_tuning; no longer the cache key


def _call(name, rel, fmt, *args):
    dev = rel.device
    _ffi.use_current_stream(dev)
    return _ffi.get_global_func(name)(rel.handle(fmt), *args)


def _call_hetero(name, gidx, fmts, *args):
    """A registry function that takes the heterograph handle (the reference's HeteroGraphRef)."""
    _ffi.use_current_stream(gidx.ctx)
    return _ffi.get_global_func(name)(gidx.hetero_handle(fmts), *args)


def _spmm_format(rel, reduce_op="max", dtype=None):
    # aten::SpMM: SelectFormat(0, CSC_CODE) — CSC (built on demand) unless the graph is
    # restricted to COO (src/array/kernel.cc:26-43).  One departure: a TRANSIENT relation (a
    # sampled mini-batch block or its reverse, rebuilt every step) that has no CSC yet takes a
    # fp32 / fp64 SUM through the COO kernel (edge-parallel hardware atomics, like the reference's
    # own COO path) instead of sorting its edges into a CSC that is used once: the conversion
    # was half of a mini-batch GraphSAGE backward pass.  USE_DETERMINISTIC_ALG keeps the CSC route.
    if rel.allowed("csc"):
        if (rel.transient and reduce_op == "sum" and not rel.has("csc") and rel.allowed("coo") and
                dtype in (torch.float32, torch.float64) and "USE_DETERMINISTIC_ALG" not in os.environ):
            return "coo"
        return "csc"
    if rel.allowed("coo"):
        return "coo"
    raise DGLAMDError("SpMM only supports CSC and COO formats")


def _sddmm_format(rel):
    # aten::SDDMM: SelectFormat(0, COO_CODE) — COO preferred, else CSR (kernel.cc:230-247)
    if rel.allowed("coo"):
        return "coo"
    if rel.allowed("csr"):
        return "csr"
    raise DGLAMDError("SDDMM only supports CSR and COO formats")


def _gspmm(gidx, op, reduce_op, u, e, accumulate_into=None, mean=False):
    """out[v] = reduce_{(u,e,v)} op(u_feat, e_feat).  Returns ``(out, (arg_u, arg_e))``.
    ``mean=True`` (with reduce_op 'sum' on a CSC-capable graph) divides every row by
    max(in-degree, 1) inside the kernel."""
    if gidx.number_of_etypes() != 1:
        raise DGLAMDError("We only support gspmm on graph with one edge type")
    use_u, use_e = op != "copy_rhs", op != "copy_lhs"
    _check_pair(u, e, use_u, use_e, "spmm")
    rel = gidx.relations[0]
    expand_u = expand_e = False
    if use_u:
        if u.dim() == 1:
            u, expand_u = u.unsqueeze(-1), True
    if use_e:
        if e.dim() == 1:
            e, expand_e = e.unsqueeze(-1), True
    ref = u if use_u else e
    dtype, dev = ref.dtype, ref.device
    u_shp = tuple(u.shape) if use_u else (0,)
    e_shp = tuple(e.shape) if use_e else (0,)
    s, d = gidx.metagraph.find_edge(0)
    v_shp = (gidx.num_nodes(d),) + infer_broadcast_shape(op, u_shp[1:], e_shp[1:])
    use_cmp = reduce_op in ("max", "min")
    n_edges = gidx.num_edges(0)
    if accumulate_into is not None:
        v = accumulate_into
    elif n_edges == 0:
        v = torch.zeros(v_shp, dtype=dtype, device=dev)   # _sparse_ops.py:238: kernel is not called
    else:
        v = torch.empty(v_shp, dtype=dtype, device=dev)   # every row is written by the kernel
    arg_u = arg_e = None
    if use_cmp:
        mk = torch.zeros if n_edges == 0 else torch.empty
        if use_u:
            arg_u = mk(v_shp, dtype=rel.idtype, device=dev)
        if use_e:
            arg_e = mk(v_shp, dtype=rel.idtype, device=dev)
    if n_edges > 0 and v.numel() > 0:
        fmt = _spmm_format(rel, reduce_op if accumulate_into is None and not mean else "max", dtype)
        uu = u.contiguous() if use_u else None
        ee = e.contiguous() if use_e else None
        args = (op, reduce_op, _nd(uu), _nd(ee), _nd(v), _nd(arg_u), _nd(arg_e))
        if fmt == "csc":
            # scratch size of this (operator, shapes) on this relation: asked once, remembered on
            # the relation (a second FFI round trip per call was a quarter of the host time of a
            # small-graph operator); the tuning bits are part of the key because the split-row
            # layout lives in the same scratch
            key = (op, reduce_op, dtype, u_shp[1:], e_shp[1:], _capi_tuning())
            need = rel.__dict__.setdefault("_ws_need", {})
            nbytes = need.get(key)
            if nbytes is None:
                nbytes = need[key] = _call("sparse._CAPI_DGLKernelSpMMWorkspaceBytes", rel, fmt, *args)
            rel.ensure_workspace(nbytes)
            tok = _static_token(uu) if (use_u and _static) else 0
            tok_e = _static_token(ee) if (use_e and _static and reduce_op == "sum") else 0
            if tok or tok_e:  # one-shot announcement consumed by the SpMM call below
                _call("dgl_amd._CAPI_UnitGraphStaticOperand", rel, fmt, tok, tok_e)
        name = "sparse._CAPI_DGLKernelSpMM" if accumulate_into is None else \
            "sparse._CAPI_DGLKernelSpMMAccumulate"
        if mean:
            assert reduce_op == "sum" and accumulate_into is None and fmt == "csc"
            name = "sparse._CAPI_DGLKernelSpMMMean"
        _call(name, rel, fmt, *args)
    # 1-D inputs give 1-D outputs (_sparse_ops.py:258-264)
    if (expand_u or not use_u) and (expand_e or not use_e):
        v = v.squeeze(-1)
        arg_u = None if arg_u is None else arg_u.squeeze(-1)
        arg_e = None if arg_e is None else arg_e.squeeze(-1)
    return v, (arg_u, arg_e)


def _gsddmm(gidx, op, lhs, rhs, lhs_target="u", rhs_target="v"):
    """out[e] = op(lhs[target], rhs[target]) for every edge."""
    if gidx.number_of_etypes() != 1:
        raise DGLAMDError("We only support gsddmm on graph with one edge type")
    use_l, use_r = op != "copy_rhs", op != "copy_lhs"
    _check_pair(lhs, rhs, use_l, use_r, "sddmm")
    rel = gidx.relations[0]
    expand_l = expand_r = False
    if use_l and lhs.dim() == 1:
        lhs, expand_l = lhs.unsqueeze(-1), True
    if use_r and rhs.dim() == 1:
        rhs, expand_r = rhs.unsqueeze(-1), True
    ref = lhs if use_l else rhs
    l_shp = tuple(lhs.shape) if use_l else (0,)
    r_shp = tuple(rhs.shape) if use_r else (0,)
    n_edges = gidx.num_edges(0)
    out_shp = (n_edges,) + infer_broadcast_shape(op, l_shp[1:], r_shp[1:])
    out = torch.empty(out_shp, dtype=ref.dtype, device=ref.device)
    if n_edges > 0 and out.numel() > 0:
        fmt = _sddmm_format(rel)
        _call("sparse._CAPI_DGLKernelSDDMM", rel, fmt, op,
              _nd(lhs.contiguous() if use_l else None), _nd(rhs.contiguous() if use_r else None),
              _nd(out), _TARGET[lhs_target], _TARGET[rhs_target])
    if (expand_l or not use_l) and (expand_r or not use_r):
        out = out.squeeze(-1)
    return out


def _softmax_scratch(rel, t):
    """Give the graph handle the scratch the degree-balanced softmax kernels want (the
    reference would take it from the tensoradapter's workspace pool)."""
    dim = 1
    for d in t.shape[1:]:
        dim *= int(d)
    bits = 64 if t.dtype == torch.float64 else 32
    need = rel.__dict__.setdefault("_esm_need", {})
    nbytes = need.get((dim, bits))
    if nbytes is None:  # asked once per (width, accumulator) and relation
        nbytes = need[(dim, bits)] = _call("sparse._CAPI_DGLKernelEdge_softmaxWorkspaceBytes", rel, "csc", dim, bits)
    rel.ensure_softmax_workspace(nbytes)


def _edge_softmax_forward(gidx, e, op="copy_rhs"):
    """Fused softmax of edge scores over the incoming edges of each destination node
    (python/dgl/_sparse_ops.py:720-758; CPU-only in the reference)."""
    if gidx.number_of_etypes() != 1:
        raise DGLAMDError("We only support edge_softmax on graph with one edge type")
    rel = gidx.relations[0]
    expand = e.dim() == 1
    if expand:
        e = e.unsqueeze(-1)
    e = e.contiguous()
    out = torch.empty_like(e)
    if gidx.num_edges(0) > 0 and e.numel() > 0:
        _softmax_scratch(rel, e)
        _call("sparse._CAPI_DGLKernelEdge_softmax_forward", rel, "csc", op, None, _nd(e), _nd(out))
    return out.squeeze(-1) if expand else out


def _edge_softmax_backward(gidx, out, sds):
    """grad_score = sds - out * sum_dst(sds)   (python/dgl/_sparse_ops.py:761-800)."""
    rel = gidx.relations[0]
    expand = out.dim() == 1
    if expand:
        out, sds = out.unsqueeze(-1), sds.unsqueeze(-1)
    out, sds = out.contiguous(), sds.contiguous()
    back = torch.empty_like(out)
    if gidx.num_edges(0) > 0 and out.numel() > 0:
        _softmax_scratch(rel, out)
        _call("sparse._CAPI_DGLKernelEdge_softmax_backward", rel, "csc", "copy_rhs", _nd(out),
              _nd(sds), _nd(back), None)
    return back.squeeze(-1) if expand else back


# ---------------------------------------------------------------------------------------
# Heterogeneous graphs: one kernel launch per relation on the same stream, results of the
# relations that share a destination node type reduced into one buffer — the structure of
# SpMMCsrHetero (src/array/cuda/spmm_hetero.cu:26-200).
# ---------------------------------------------------------------------------------------
def _gspmm_hetero(gidx, op, reduce_op, u_len, u_and_e_tuple):
    """Returns ``(out_per_dst_ntype, (arg_u, arg_e, arg_u_ntype, arg_e_etype))`` like
    python/dgl/_sparse_ops.py:268-433.  ``u`` is indexed by source node type, ``e`` by edge
    type, outputs by destination node type.

    Destination types whose relations can be summed by one stacked launch take that route
    (``_fused_hetero``: sum, and max / min with their type trackers); everything else goes through ``sparse._CAPI_DGLKernelSpMMHetero``
    with the reference's argument lists: the per-relation loop, the accumulate-into-``V``
    contract for sum and the strict running compare with node / edge type tracking for
    max / min all live on the C++ side (csrc/ffi_registry.hip ≙ spmm_hetero.cu:26-200)."""
    u_tuple, e_tuple = u_and_e_tuple[:u_len], u_and_e_tuple[u_len:]
    use_u, use_e = op != "copy_rhs", op != "copy_lhs"
    n_nt, n_et = gidx.number_of_ntypes(), gidx.number_of_etypes()
    outs = [None] * n_nt
    use_cmp = reduce_op in ("max", "min")
    list_u, list_e, list_v = [None] * n_nt, [None] * n_et, [None] * n_nt
    arg_u, arg_e = [None] * n_nt, [None] * n_nt
    arg_u_nt, arg_e_et = [None] * n_nt, [None] * n_nt
    fused = _fused_hetero(gidx, op, reduce_op, u_tuple, e_tuple, outs,
                          (arg_u, arg_e, arg_u_nt, arg_e_et))

    squeeze = [False] * n_nt
    feat_shape = {}
    fmts = ["coo"] * n_et
    todo = False
    for et in range(n_et):
        s, d = gidx.metagraph.find_edge(et)
        rel = gidx.relations[et]
        fmts[et] = "csc" if rel.allowed("csc") else ("coo" if rel.allowed("coo") else "csr")
        if d in fused:
            continue
        u = u_tuple[s] if use_u else None
        e = e_tuple[et] if use_e else None
        if (use_u and u is None) or (use_e and e is None):
            continue
        _check_pair(u, e, use_u, use_e, "spmm")
        expand_u = expand_e = False
        if use_u and u.dim() == 1:
            u, expand_u = u.unsqueeze(-1), True
        if use_e and e.dim() == 1:
            e, expand_e = e.unsqueeze(-1), True
        o_feat = infer_broadcast_shape(op, tuple(u.shape[1:]) if use_u else (),
                                       tuple(e.shape[1:]) if use_e else ())
        if d in feat_shape and feat_shape[d] != o_feat:
            # src/array/kernel.cc:194-199: relations reducing into one node type must agree
            raise DGLAMDError("The feature shape of relation {} does not match others.".format(et))
        feat_shape[d] = o_feat
        ref = u if use_u else e
        if use_u:
            list_u[s] = u.contiguous()
        if use_e:
            list_e[et] = e.contiguous()
        if list_v[d] is None:
            v_shp = (gidx.num_nodes(d),) + o_feat
            list_v[d] = torch.zeros(v_shp, dtype=ref.dtype, device=ref.device)  # _sparse_ops.py:351
            squeeze[d] = (expand_u or not use_u) and (expand_e or not use_e)
            if use_cmp:
                if use_u:
                    arg_u[d] = torch.zeros(v_shp, dtype=rel.idtype, device=ref.device)
                    arg_u_nt[d] = torch.zeros(v_shp, dtype=rel.idtype, device=ref.device)
                if use_e:
                    arg_e[d] = torch.zeros(v_shp, dtype=rel.idtype, device=ref.device)
                    arg_e_et[d] = torch.zeros(v_shp, dtype=rel.idtype, device=ref.device)
        if gidx.num_edges(et) > 0 and list_v[d].numel() > 0:
            todo = True
            if fmts[et] == "csc":  # attach scratch to the relation so that its merge plan is cached
                args = (op, reduce_op, _nd(list_u[s] if use_u else None),
                        _nd(list_e[et] if use_e else None), _nd(list_v[d]), _nd(arg_u[d]), _nd(arg_e[d]))
                rel.ensure_workspace(_call("sparse._CAPI_DGLKernelSpMMWorkspaceBytes", rel, "csc", *args))
    if todo:
        if fmts and any(f == "csr" for f in fmts):
            raise DGLAMDError("SpMM only supports CSC and COO formats")
        nd = lambda ts: [_nd(t) for t in ts]
        _call_hetero("sparse._CAPI_DGLKernelSpMMHetero", gidx, fmts, op, reduce_op, nd(list_u),
                     nd(list_e), nd(list_v), nd(arg_u), nd(arg_e), nd(arg_u_nt), nd(arg_e_et))
    for d in range(n_nt):
        if d in fused or list_v[d] is None:
            continue
        if squeeze[d]:  # 1-D inputs give 1-D outputs (_sparse_ops.py:424-430)
            list_v[d] = list_v[d].squeeze(-1)
            for lst in (arg_u, arg_e, arg_u_nt, arg_e_et):
                lst[d] = None if lst[d] is None else lst[d].squeeze(-1)
        outs[d] = list_v[d]
    return tuple(outs), (arg_u, arg_e, arg_u_nt, arg_e_et)


_FUSED_OPS = ("copy_lhs", "copy_rhs", "mul")


FUSE_HETERO = True  # tests switch the stacked launches off to compare them with the sequential path


def _fused_hetero(gidx, op, reduce_op, u_tuple, e_tuple, outs, arg_lists):
    """Destination node types whose relations are reduced by ONE stacked launch
    (sparse._CAPI_DGLKernelSpMMStacked for sum, ...StackedCmp for max / min with the node / edge
    type trackers) instead of the reference's per-relation loop.  Fills ``outs[d]`` (and, for
    max / min, the four ``arg_lists``) for those types and returns their set; everything it does
    not take (other operators, mixed shapes, COO-only graphs, a single relation) is left to the
    sequential path."""
    done = set()
    if not FUSE_HETERO or reduce_op not in ("sum", "max", "min") or op not in _FUSED_OPS:
        return done
    use_cmp = reduce_op != "sum"
    use_u, use_e = op != "copy_rhs", op != "copy_lhs"
    by_dst = {}
    for et in range(gidx.number_of_etypes()):
        s, d = gidx.metagraph.find_edge(et)
        u = u_tuple[s] if use_u else None
        e = e_tuple[et] if use_e else None
        if (use_u and u is None) or (use_e and e is None) or gidx.num_edges(et) == 0:
            continue
        by_dst.setdefault(d, []).append((et, u, e))
    for d, items in by_dst.items():
        if len(items) < 2 or len(items) > 255:
            continue
        us = [u for _, u, _ in items]
        es = [e for _, _, e in items]
        ref = us[0] if use_u else es[0]
        same = lambda ts: all(t.shape[1:] == ts[0].shape[1:] and t.dtype == ts[0].dtype and
                              t.dim() >= 2 for t in ts)
        if (use_u and not same(us)) or (use_e and not same(es)):
            continue
        if use_u and use_e and us[0].dtype != es[0].dtype:
            continue
        if not all(gidx.relations[et].allowed("csc") for et, _, _ in items):
            continue
        # only the broadcast forms the stacked kernels implement: equal shapes or (..,H,D)x(..,H,1)
        if use_u and use_e:
            a, b = tuple(us[0].shape[1:]), tuple(es[0].shape[1:])
            if a != b and not (len(a) == len(b) and a[:-1] == b[:-1] and b[-1] == 1):
                continue
        stk, rel = gidx.stacked([et for et, _, _ in items])
        o_feat = infer_broadcast_shape(op, tuple(us[0].shape[1:]) if use_u else (),
                                       tuple(es[0].shape[1:]) if use_e else ())
        v = torch.empty((gidx.num_nodes(d),) + o_feat, dtype=ref.dtype, device=ref.device)
        us_c = [u.contiguous() for u in us] if use_u else None
        es_c = [e.contiguous() for e in es] if use_e else None
        table = lambda ts: None if ts is None else _nd(torch.tensor(
            [t.data_ptr() for t in ts], dtype=torch.int64, device=ref.device))
        if use_cmp:
            ids = lambda: torch.empty(v.shape, dtype=stk.idtype, device=ref.device)
            au, ant = (ids(), ids()) if use_u else (None, None)
            ae, aet = (ids(), ids()) if use_e else (None, None)
            types = torch.tensor([[gidx.metagraph.find_edge(et)[0] for et, _, _ in items],
                                  [et for et, _, _ in items]], dtype=torch.int32)
            args = (op, reduce_op, _nd(us_c[0]) if use_u else None, _nd(es_c[0]) if use_e else None,
                    table(us_c), table(es_c), _nd(v), _ffi.NDArray(rel), _ffi.NDArray(types, host_ok=True),
                    _nd(au), _nd(ae), _nd(ant), _nd(aet))
            nbytes = _call("sparse._CAPI_DGLKernelSpMMStackedCmpWorkspaceBytes", stk, "csc", *args)
            stk.ensure_workspace(nbytes)
            _call("sparse._CAPI_DGLKernelSpMMStackedCmp", stk, "csc", *args)
            arg_lists[0][d], arg_lists[1][d], arg_lists[2][d], arg_lists[3][d] = au, ae, ant, aet
        else:
            args = (op, _nd(us_c[0]) if use_u else None, _nd(es_c[0]) if use_e else None,
                    table(us_c), table(es_c), _nd(v), _ffi.NDArray(rel))
            nbytes = _call("sparse._CAPI_DGLKernelSpMMStackedWorkspaceBytes", stk, "csc", *args)
            stk.ensure_workspace(nbytes)
            _call("sparse._CAPI_DGLKernelSpMMStacked", stk, "csc", *args)
        outs[d] = v
        done.add(d)
    return done


def _gsddmm_hetero(gidx, op, lhs_len, lhs_target, rhs_target, lhs_and_rhs_tuple):
    """Per-relation SDDMM (python/dgl/_sparse_ops.py:568-638) through
    ``sparse._CAPI_DGLKernelSDDMMHetero``: operands indexed by node type for 'u' / 'v' targets
    and by edge type for 'e'; output indexed by edge type."""
    lhs_tuple, rhs_tuple = lhs_and_rhs_tuple[:lhs_len], lhs_and_rhs_tuple[lhs_len:]
    use_l, use_r = op != "copy_rhs", op != "copy_lhs"
    n_et = gidx.number_of_etypes()
    outs = [None] * n_et
    squeeze = [False] * n_et
    prep = {}

    def operand(tup, idx, side):
        t = tup[idx]
        if t is None:
            return None, False
        key = (side, idx)
        if key not in prep:
            ex = t.dim() == 1
            prep[key] = ((t.unsqueeze(-1) if ex else t).contiguous(), ex)
        return prep[key]

    fmts, todo = [], False
    for et in range(n_et):
        s, d = gidx.metagraph.find_edge(et)
        rel = gidx.relations[et]
        fmts.append("coo" if rel.allowed("coo") else ("csr" if rel.allowed("csr") else "csc"))
        pick = lambda tgt: {"u": s, "v": d, "e": et}[tgt]
        l, ex_l = operand(lhs_tuple, pick(lhs_target), "l") if use_l else (None, False)
        r, ex_r = operand(rhs_tuple, pick(rhs_target), "r") if use_r else (None, False)
        if (use_l and l is None) or (use_r and r is None):
            continue
        _check_pair(l, r, use_l, use_r, "sddmm")
        ref = l if use_l else r
        o_feat = infer_broadcast_shape(op, tuple(l.shape[1:]) if use_l else (),
                                       tuple(r.shape[1:]) if use_r else ())
        outs[et] = torch.empty((gidx.num_edges(et),) + o_feat, dtype=ref.dtype, device=ref.device)
        squeeze[et] = (ex_l or not use_l) and (ex_r or not use_r)
        if fmts[-1] == "csc":
            raise DGLAMDError("SDDMM only supports CSR and COO formats")
        todo = todo or outs[et].numel() > 0
    if todo:
        n_l = len(lhs_tuple)
        n_r = len(rhs_tuple)
        lst = lambda side, n: [_nd(prep[(side, i)][0]) if (side, i) in prep else None for i in range(n)]
        _call_hetero("sparse._CAPI_DGLKernelSDDMMHetero", gidx, fmts, op, lst("l", n_l), lst("r", n_r),
                     [_nd(o) for o in outs], _TARGET[lhs_target], _TARGET[rhs_target])
    return tuple(None if o is None else (o.squeeze(-1) if sq else o) for o, sq in zip(outs, squeeze))


def _update_grad_minmax_hetero(gidx, op, list_x, list_idx, list_idx_etype, list_dX):
    """Gradient of a max / min reduction over several relations w.r.t. the copied operand
    (python/dgl/_sparse_ops.py:723-770).  ``gidx`` is the REVERSED graph; ``list_x`` the incoming
    gradients per (forward) destination node type, ``list_idx`` / ``list_idx_etype`` the winning
    node (copy_lhs) or edge (copy_rhs) ids and their node / edge types recorded by the forward
    pass; ``list_dX`` fixes the output row counts (per source node type for copy_lhs, per edge
    type for copy_rhs).  Returns the gradients in that indexing."""
    use_u, use_e = op != "copy_rhs", op != "copy_lhs"
    list_out = [None] * len(list_dX)
    for etid in range(gidx.number_of_etypes()):
        src_id, dst_id = gidx.metagraph.find_edge(etid)  # reversed: src_id = forward destination type
        x = list_x[src_id]
        if x is None:
            continue
        if use_u and list_dX[dst_id] is not None and list_out[dst_id] is None:
            list_out[dst_id] = torch.zeros((len(list_dX[dst_id]),) + tuple(x.shape[1:]), dtype=x.dtype,
                                           device=x.device)
        if use_e and list_dX[etid] is not None and list_out[etid] is None:
            list_out[etid] = torch.zeros((len(list_dX[etid]),) + tuple(x.shape[1:]), dtype=x.dtype,
                                         device=x.device)
    nd = lambda ts: [None if t is None else _nd(t.contiguous()) for t in ts]
    _call_hetero("sparse._CAPI_DGLKernelUpdateGradMinMaxHetero", gidx,
                 ["coo" if r.allowed("coo") else r.formats[0] for r in gidx.relations], op, nd(list_x),
                 nd(list_idx), nd(list_idx_etype), nd(list_out))
    return tuple(list_out)
