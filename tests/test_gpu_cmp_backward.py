"""dgla_spmm_cmp_backward (VERDICT r3 Next #7): the max / min g-SpMM backward in ONE launch, winners read
in the graph's idtype, against the reference's composition (python/dgl/backend/pytorch/sparse.py:217-244:
``dX.scatter_add_(0, argX.long(), Y.expand(..).gather(0, argY.long()) * dZ)``)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(dz, arg, rows, other, arg_other):
    out = torch.zeros((rows,) + tuple(dz.shape[1:]), dtype=torch.float64, device=dz.device)
    g = dz.double()
    ok = arg >= 0
    if other is not None:
        g = other.double().expand(-1, *dz.shape[1:]).gather(0, arg_other.long().clamp(min=0)) * g
    g = torch.where(ok, g, torch.zeros_like(g))
    out.scatter_add_(0, arg.long().clamp(min=0), g)
    return out


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,oshape", [((7,), None), ((7,), (7,)), ((4, 8), (4, 1)), ((4, 8), (1, 8)), ((4, 8), (1,)),
                                          ((100,), (100,)), ((3, 5, 2), (3, 5, 2))])
def test_cmp_backward_equals_the_reference_composition(dev, idtype, dtype, shape, oshape):
    from dgl_amd import _capi
    from dgl_amd.autograd import _bcast_group

    n, rows, orows = 3000, 500, 800
    g = torch.Generator(device=dev).manual_seed(hash((shape, oshape)) % 1000)
    dz = torch.randn((n,) + shape, device=dev, generator=g).to(dtype)
    arg = torch.randint(-1, rows, (n,) + shape, device=dev, generator=g).to(idtype)     # -1: nothing won
    other = arg_other = None
    grp = (0, 1)
    if oshape is not None:
        oshape = (1,) * (len(shape) - len(oshape)) + oshape
        other = torch.randn((orows,) + oshape, device=dev, generator=g).to(dtype)
        arg_other = torch.randint(0, orows, (n,) + shape, device=dev, generator=g).to(idtype)
        grp = _bcast_group(oshape, shape)
        assert grp is not None
    out = torch.zeros((rows,) + shape, dtype=dtype, device=dev)
    _capi.spmm_cmp_backward(dz, arg, out, other, arg_other, grp[1], atomic=True)
    want = _reference(dz, arg, rows, other, arg_other)
    tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.bfloat16: 6e-2, torch.float16: 1e-2}[dtype]
    torch.testing.assert_close(out.double(), want, rtol=tol, atol=tol * 4)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_edge_operand_path_is_deterministic_and_exact(dev, idtype):
    """atomic = False: targets are written at most once -> plain stores, the bits of every run are equal
    and equal to dz * other computed element-wise."""
    from dgl_amd import _capi

    n, e, f = 20000, 300000, 16
    g = torch.Generator(device=dev).manual_seed(4)
    dz = torch.randn(n, f, device=dev, generator=g)
    perm = torch.stack([torch.randperm(e, device=dev, generator=g)[:n] for _ in range(f)], dim=1)   # injective per column
    arg = perm.to(idtype).contiguous()
    x = torch.randn(5000, f, device=dev, generator=g)
    argx = torch.randint(0, 5000, (n, f), device=dev, generator=g).to(idtype)
    outs = []
    for _ in range(3):
        out = torch.zeros(e, f, device=dev)
        _capi.spmm_cmp_backward(dz, arg, out, x, argx, 1, atomic=False)
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = torch.zeros(e, f, device=dev)
    want.scatter_(0, arg.long(), dz * x.gather(0, argx.long()))
    assert torch.equal(outs[0], want)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_edge_operand_row0_is_shared_with_empty_destinations(dev, idtype):
    """the forward records arg_e = 0 for every destination WITHOUT in-edges, so edge 0 is named
    by its real winner AND by every empty row.  The reference adds (scatter_add_, sparse.py:217-244): row 0 of the
    edge gradient = the winner's dz + the empty rows' dz, on every run (no store race)."""
    from dgl_amd import _capi

    n, e, f = 4096, 9000, 8
    g = torch.Generator(device=dev).manual_seed(11)
    dz = torch.randn(n, f, device=dev, generator=g)
    arg = (torch.arange(n, device=dev).unsqueeze(1) * 2 + 1).expand(n, f).clone()     # injective, never 0
    arg[7] = 0                                                                        # edge 0 wins row 7, every column
    empty = torch.arange(n, device=dev) % 5 == 3                                      # ~ 800 empty rows: arg 0
    arg[empty] = 0
    arg = arg.to(idtype).contiguous()
    want = torch.zeros(e, f, device=dev, dtype=torch.float64).scatter_add_(0, arg.long(), dz.double())
    outs = []
    for _ in range(3):
        out = torch.zeros(e, f, device=dev)
        _capi.spmm_cmp_backward(dz, arg, out, None, None, 1, atomic=False)
        outs.append(out)
    for out in outs:
        assert torch.equal(out[1:], want[1:].float())                                 # single writers: exact
        torch.testing.assert_close(out[0].double(), want[0], rtol=1e-5, atol=1e-4)    # the shared row: the sum
    # what update_all produces (empty rows' dz = 0 after inf -> 0): row 0 is exactly the true winner's gradient
    dz0 = torch.where(empty.unsqueeze(1), torch.zeros_like(dz), dz)
    out = torch.zeros(e, f, device=dev)
    _capi.spmm_cmp_backward(dz0, arg, out, None, None, 1, atomic=False)
    assert torch.equal(out[0], dz[7])


def test_autograd_edge_gradient_with_isolated_destinations_and_edge0_winning(dev):
    """End to end: u_mul_e_max on a graph whose rows 0..49 have no in-edges and where edge id 0 wins; dW[0] must be
    the true winner's gradient on every run (it used to race with the empty rows' zero stores)."""
    import dgl_amd as dgl

    n, e = 200, 3000
    g0 = torch.Generator().manual_seed(5)
    src = torch.randint(0, n, (e,), generator=g0)
    dst = torch.randint(50, n, (e,), generator=g0)
    g = dgl.graph((src.to(dev), dst.to(dev)), num_nodes=n, idtype=torch.int32, device=dev)
    x = (torch.rand(n, 6, generator=g0) + 1).to(dev)
    w0 = (torch.rand(e, 6, generator=g0) + 1)
    w0[0] = 100.0                                                                       # edge 0 wins all its columns
    up = torch.randn(n, 6, generator=g0).to(dev)
    grads = []
    for _ in range(4):
        w = w0.clone().to(dev).requires_grad_()
        out = dgl.ops.u_mul_e_max(g, x, w)
        out = torch.where(torch.isinf(out), torch.zeros_like(out), out)
        (out * up).sum().backward()
        grads.append(w.grad)
    want0 = up[dst[0]] * x[src[0]]
    for gr in grads:
        torch.testing.assert_close(gr[0], want0, rtol=1e-6, atol=1e-6)
        assert torch.equal(gr, grads[0])


def test_autograd_max_min_backward_runs_on_the_library_kernel(dev, monkeypatch):
    """dgl.ops.copy_u_max / u_mul_e_min gradients == a dense torch evaluation, and the library entry is the
    one that ran."""
    import dgl_amd as dgl
    from dgl_amd import _capi

    calls = []
    real = _capi.spmm_cmp_backward
    monkeypatch.setattr(_capi, "spmm_cmp_backward", lambda *a, **k: (calls.append(k.get("atomic")), real(*a, **k))[1])
    n, e = 300, 4000
    g0 = torch.Generator().manual_seed(9)
    src, dst = torch.randint(0, n, (e,), generator=g0), torch.randint(0, n, (e,), generator=g0)
    g = dgl.graph((src.to(dev), dst.to(dev)), num_nodes=n, idtype=torch.int32, device=dev)
    x = torch.randn(n, 4, 3, device=dev, requires_grad=True)
    w = torch.randn(e, 4, 1, device=dev, requires_grad=True)
    up = torch.randn(n, 4, 3, device=dev)
    out = dgl.ops.u_mul_e_max(g, x, w)
    out = torch.where(torch.isinf(out), torch.zeros_like(out), out)
    (out * up).sum().backward()
    assert calls == [True, False]
    # dense evaluation
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    msg = xd[src.to(dev)] * wd                                    # (e, 4, 3)
    m = torch.full((n, 4, 3), float("-inf"), dtype=torch.float64, device=dev).index_reduce_(
        0, dst.to(dev), msg, "amax", include_self=True)
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    (m * up.double()).sum().backward()
    torch.testing.assert_close(x.grad.double(), xd.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(w.grad.double(), wd.grad, rtol=1e-5, atol=1e-6)


# ---- the node operand's gradient as a gather over the reverse graph (dgla_spmm_cmp_mask + dgla_spmm_csr_masked) -------
def _graph(dev, n, e, idtype, seed, multi=True, hub=True):
    """Random multigraph with parallel edges, a hub of high in-degree, nodes without in-edges and a shuffled
    edge order (so both CSC and CSR carry an edge-id map)."""
    import dgl_amd as dgl

    g0 = torch.Generator().manual_seed(seed)
    src, dst = torch.randint(0, n, (e,), generator=g0), torch.randint(n // 10, n, (e,), generator=g0)   # rows < n/10: no in-edges
    if hub:
        dst[: e // 8] = n - 1
    if multi:
        src[e // 2: e // 2 + e // 16] = src[: e // 16]
        dst[e // 2: e // 2 + e // 16] = dst[: e // 16]
    return dgl.graph((src.to(dev), dst.to(dev)), num_nodes=n, idtype=idtype, device=dev), src.to(dev), dst.to(dev)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1,), (7,), (32,), (65,), (100,), (128,), (130,), (257,), (4, 8), (2, 3, 5)])
@pytest.mark.parametrize("red", ["max", "min"])
def test_gather_backward_equals_the_scatter_of_the_reference(dev, monkeypatch, idtype, dtype, shape, red):
    """copy_u_max / copy_u_min: dX through the winner-bit gather == ``zeros.scatter_add_(0, arg_u, dZ)`` of the
    reference (python/dgl/backend/pytorch/sparse.py:216-224), -inf rows and in-degree-0 rows (arg = 0: the
    reference sends their dZ to node 0) included; the two kernels of the gather path are the ones that ran."""
    import dgl_amd as dgl
    from dgl_amd import _capi, autograd

    ran = []
    r1, r2 = _capi.spmm_cmp_mask, _capi.spmm_csr_masked
    monkeypatch.setattr(_capi, "spmm_cmp_mask", lambda *a, **k: (ran.append("mask"), r1(*a, **k))[1])
    monkeypatch.setattr(_capi, "spmm_csr_masked", lambda *a, **k: (ran.append("spmm"), r2(*a, **k))[1])
    n, e = 700, 9000
    g, src, dst = _graph(dev, n, e, idtype, seed=len(shape) * 31 + shape[0])
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((n,) + shape, device=dev, generator=gen).to(dtype)
    x[5] = float("-inf") if red == "max" else float("inf")      # a source that never wins
    x[n // 2] = x[n // 3]                                        # ties between different sources
    x.requires_grad_()
    up = torch.randn((n,) + shape, device=dev, generator=gen).to(dtype)
    out = getattr(dgl.ops, "copy_u_" + red)(g, x)
    out.backward(up)
    assert ran == ["mask", "spmm", "mask"]      # bits (dX untouched) -> gated g-SpMM (stores) -> finish (unclaimed elements)
    got = x.grad.clone()
    # the reference composition on the winners the forward recorded (taken from a second forward: deterministic)
    gidx = g._graph
    _, (arg_u, _) = dgl.sparse_kernels._gspmm(gidx, "copy_lhs", red, x.detach(), None)
    want = torch.zeros((n,) + shape, dtype=torch.float64, device=dev)
    want.scatter_add_(0, arg_u.long(), up.double())
    tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.bfloat16: 6e-2, torch.float16: 1e-2}[dtype]
    torch.testing.assert_close(got.double(), want, rtol=tol, atol=tol * 4)
    # same bits on every run, and the atomic kernel agrees
    x.grad = None
    getattr(dgl.ops, "copy_u_" + red)(g, x).backward(up)
    assert torch.equal(got, x.grad)
    monkeypatch.setenv("DGLA_CMP_BACKWARD", "atomic")
    x.grad = None
    getattr(dgl.ops, "copy_u_" + red)(g, x).backward(up)
    assert ran == ["mask", "spmm", "mask"] * 2
    # the atomic kernel adds in the tensor's own type, in whatever order the atomics land (as the reference's
    # scatter_add_ does): node 0 collects the gradient of every destination without an in-edge — dozens of 16-bit
    # additions — so its row is held to the bound of that many roundings, the others to the plain tolerance
    torch.testing.assert_close(x.grad.double()[1:], want[1:], rtol=tol, atol=tol * 4)
    addends = int((arg_u.reshape(n, -1)[:, 0] == 0).sum())
    wide = tol * 4 * max(1.0, addends ** 0.5) if dtype in (torch.bfloat16, torch.float16) else tol * 4
    torch.testing.assert_close(x.grad.double()[:1], want[:1], rtol=tol, atol=wide)


def test_gather_backward_u_add_e_max(dev):
    """add: dX is the same scatter of dZ (the message is linear in X with coefficient 1); dY keeps its plain path."""
    import dgl_amd as dgl

    n, e = 400, 6000
    g, src, dst = _graph(dev, n, e, torch.int32, seed=11)
    gen = torch.Generator(device=dev).manual_seed(8)
    x = torch.randn(n, 6, device=dev, generator=gen, requires_grad=True)
    w = torch.randn(e, 6, device=dev, generator=gen, requires_grad=True)
    up = torch.randn(n, 6, device=dev, generator=gen)
    out = dgl.ops.u_add_e_max(g, x, w)
    out = torch.where(torch.isinf(out), torch.zeros_like(out), out)
    (out * up).sum().backward()
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    m = torch.full((n, 6), float("-inf"), dtype=torch.float64, device=dev).index_reduce_(
        0, dst, xd[src] + wd, "amax", include_self=True)
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    (m * up.double()).sum().backward()
    torch.testing.assert_close(x.grad.double(), xd.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(w.grad.double(), wd.grad, rtol=1e-5, atol=1e-6)


def test_mask_kernel_by_edge_and_long_rows(dev):
    """dgla_spmm_cmp_mask with by_edge (arg_e) on rows longer than one 64-edge batch: exactly the named edge's bit
    is set, every other bit of every word is clear."""
    from dgl_amd import _capi

    rows, f = 50, 70
    gen = torch.Generator(device=dev).manual_seed(2)
    deg = torch.randint(0, 400, (rows,), device=dev, generator=gen)
    deg[3] = 0
    indptr = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    indptr[1:] = deg.cumsum(0)
    e = int(indptr[-1])
    indices = torch.randint(0, 30, (e,), device=dev, generator=gen)
    eids = torch.randperm(e, device=dev, generator=gen)
    # the winner of (row, k): a random edge of the row (by id); empty rows: arg 0
    pick = (torch.rand(rows, f, device=dev, generator=gen) * deg[:, None].clamp(min=1)).long()
    pos = (indptr[:-1, None] + pick).clamp(max=max(e - 1, 0))
    arg_e = torch.where(deg[:, None] > 0, eids[pos], torch.zeros_like(pos))
    dz = torch.randn(rows, f, device=dev, generator=gen)
    dx = torch.zeros(30, f, device=dev)
    words = _capi.spmm_cmp_mask_words(torch.float32, f)
    nbytes = _capi.spmm_cmp_mask_bytes(torch.float32, rows, e, f)
    assert nbytes >= e * words * 4
    mask = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device=dev)
    csr = _capi.make_csr(indptr, indices, eids, 30)
    _capi.spmm_cmp_mask(csr, arg_e, dz, mask, dx, by_edge=True)
    bits = mask[: e * words * 4].view(torch.int32).reshape(e, words).cpu().numpy().astype("uint32")
    import numpy as np
    want = np.zeros((e, words), dtype="uint32")
    posc, degc = pos.cpu().numpy(), deg.cpu().numpy()
    for r in range(rows):
        if degc[r] == 0:
            continue
        for k in range(f):
            want[posc[r, k], k // 32] |= np.uint32(1) << np.uint32(k % 32)
    assert (bits == want).all()
    # nothing claimed in the empty row only: its dz went to dx[0]
    assert torch.equal(dx[0], dz[3]) and float(dx[1:].abs().max()) == 0.0


def test_mask_kernel_two_step_form_equals_the_one_step_form(dev):
    """DGLA_CMP_MASK_DEFER / _FINISH around a storing g-SpMM == the one-step form with a zeroed dX and an accumulating g-SpMM,
    bit for bit — including elements no edge claims whose (hand-made) arg names a NON-ZERO row, which the finish call finds by
    scanning again (the library's own forward never produces those)."""
    from dgl_amd import _capi

    rows, ncols, f = 300, 200, 100
    gen = torch.Generator(device=dev).manual_seed(12)
    deg = torch.randint(0, 90, (rows,), device=dev, generator=gen)
    deg[5] = 0
    indptr = torch.zeros(rows + 1, dtype=torch.int32, device=dev)
    indptr[1:] = deg.cumsum(0).to(torch.int32)
    e = int(indptr[-1])
    indices = torch.randint(0, ncols, (e,), device=dev, generator=gen).to(torch.int32)
    # winners: a random edge's source; some elements name a source that is NOT among the row's edges (unclaimed, target != 0)
    pick = (torch.rand(rows, f, device=dev, generator=gen) * deg[:, None].clamp(min=1)).long()
    pos = (indptr[:-1, None].long() + pick).clamp(max=max(e - 1, 0))
    arg_u = torch.where(deg[:, None] > 0, indices[pos].long(), torch.zeros_like(pos)).to(torch.int32)
    for rare in (False, True):
        arg = arg_u.clone()
        if rare:
            arg[7, 3] = ncols - 1 if int((indices[indptr[7]:indptr[8]] == ncols - 1).sum()) == 0 else arg[7, 3]
            arg[5, 9] = 17                              # the empty row naming a non-zero target
        dz = torch.randn(rows, f, device=dev, generator=gen)
        fwd = _capi.make_csr(indptr, indices, None, ncols)
        # reverse matrix with the position map, as autograd builds it
        dst = torch.repeat_interleave(torch.arange(rows, device=dev), deg).to(torch.int32)
        order = torch.argsort(indices.long() * rows + dst.long(), stable=True)
        rip = torch.zeros(ncols + 1, dtype=torch.int32, device=dev)
        rip[1:] = torch.bincount(indices.long(), minlength=ncols).cumsum(0).to(torch.int32)
        rev = _capi.make_csr(rip, dst[order].contiguous(), order.to(torch.int32).contiguous(), rows)
        nbytes = _capi.spmm_cmp_mask_bytes(torch.float32, rows, e, f)
        outs = []
        for two_step in (False, True):
            mask = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            dx = torch.zeros(ncols, f, device=dev) if not two_step else torch.full((ncols, f), float("nan"), device=dev)
            ws = torch.empty(max(int(_capi.spmm_csr_masked_workspace_bytes(rev, dz, dx)), 1), dtype=torch.uint8, device=dev)
            if two_step:
                _capi.spmm_cmp_mask(fwd, arg, dz, mask, dx, mode=_capi.CMP_MASK_DEFER)
                _capi.spmm_csr_masked(rev, dz, mask, dx, workspace=ws, accumulate=False)
                _capi.spmm_cmp_mask(fwd, arg, dz, mask, dx, mode=_capi.CMP_MASK_FINISH)
            else:
                _capi.spmm_cmp_mask(fwd, arg, dz, mask, dx)
                _capi.spmm_csr_masked(rev, dz, mask, dx, workspace=ws, accumulate=True)
            outs.append(dx)
        want = torch.zeros(ncols, f, dtype=torch.float64, device=dev).scatter_add_(0, arg.long(), dz.double())
        torch.testing.assert_close(outs[1].double(), want, rtol=1e-5, atol=1e-5)
        if rare:   # (the two forms add the rare elements by atomics at different moments: equal to rounding only)
            torch.testing.assert_close(outs[0], outs[1], rtol=1e-6, atol=1e-6)
        else:
            assert torch.equal(outs[0], outs[1])


def test_gather_backward_replays_inside_a_hipgraph(dev):
    """copy_u_max forward + backward captured in ONE hipGraph (no allocation, no read-back inside the two backward
    kernels): replays give the eager gradient for new inputs."""
    import dgl_amd as dgl

    n, e = 500, 6000
    g, src, dst = _graph(dev, n, e, torch.int32, seed=21)
    x = torch.randn(n, 20, device=dev, requires_grad=True)
    up = torch.randn(n, 20, device=dev)
    for _ in range(2):   # warm-up: formats, position map, workspaces
        x.grad = None
        dgl.ops.copy_u_max(g, x).backward(up)
    torch.cuda.synchronize()
    sx, sup = torch.randn(n, 20, device=dev, requires_grad=True), torch.randn(n, 20, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        for _ in range(2):
            sx.grad = None
            dgl.ops.copy_u_max(g, sx).backward(sup)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    sx.grad = None
    with torch.cuda.graph(graph):
        dgl.ops.copy_u_max(g, sx).backward(sup)
    for seed in (1, 2):
        gen = torch.Generator(device=dev).manual_seed(seed)
        with torch.no_grad():
            sx.copy_(torch.randn(n, 20, device=dev, generator=gen))
            sup.copy_(torch.randn(n, 20, device=dev, generator=gen))
        graph.replay()
        torch.cuda.synchronize()
        got = sx.grad.clone()
        ex = sx.detach().clone().requires_grad_()
        dgl.ops.copy_u_max(g, ex).backward(sup)
        assert torch.equal(got, ex.grad)


@pytest.mark.parametrize("feat", [4, 36, 100, 256])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gated_row_gathers_change_no_bit(dev, feat, dtype):
    """The masked g-SpMM fetches a row piece only when one of its columns is wanted (round 6); DGLA_TUNE_NO_GATE gathers
    every piece as before.  Same bits either way, hub row and rows without in-edges included."""
    import dgl_amd as dgl
    from dgl_amd import _capi

    n, e = 3000, 60000
    g, src, dst = _graph(dev, n, e, torch.int32, seed=feat)
    gen = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(n, feat, device=dev, generator=gen).to(dtype).requires_grad_()
    up = torch.randn(n, feat, device=dev, generator=gen).to(dtype)
    default = _capi.get_tuning()
    got = []
    try:
        for flags in (default, default | _capi.TUNE_NO_GATE):
            _capi.set_tuning(flags)
            x.grad = None
            dgl.ops.copy_u_max(g, x).backward(up)
            got.append(x.grad.clone())
    finally:
        _capi.set_tuning(default)
    assert torch.equal(got[0], got[1])
