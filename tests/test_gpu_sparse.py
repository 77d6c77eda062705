"""``dgl_amd.sparse`` on the GPU: the routes that END IN THE HIP KERNELS — products (g-SpMM / g-SDDMM), softmax (fused
edge-softmax), reductions along a dimension (g-SpMM ``copy_e``), sampling (csrc/sampling.hip), format conversion
(csrc/coo2csr.hip) — at sizes the reference's own 5 x 5 cases (run unmodified by tools/ref_suite: 1249 / 1249) do not
reach, values and gradients against dense / index arithmetic in float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(dev, shape, nnz, seed, val_shape=(), fmt="coo", dtype=torch.float32):
    import dgl_amd.sparse as dglsp

    g = torch.Generator().manual_seed(seed)
    lin = torch.randperm(shape[0] * shape[1], generator=g)[:nnz]
    row, col = (lin // shape[1]).to(dev), (lin % shape[1]).to(dev)
    val = torch.randn((nnz,) + val_shape, generator=g).to(device=dev, dtype=dtype).requires_grad_(True)
    A = dglsp.from_coo(row, col, val, shape)
    if fmt == "csr":
        A.csr()
    elif fmt == "csc":
        A.csc()
    return A, val, row.long(), col.long()


def _close(a, b, tol=2e-5):
    scale = float(b.abs().max()) + 1e-30
    return float((a.double() - b.double()).abs().max()) <= tol * scale


@pytest.mark.parametrize("fmt", ["coo", "csr", "csc"])
def test_products_at_size_match_index_arithmetic_with_gradients(dev, fmt):
    import dgl_amd.sparse as dglsp

    L, M, nnz, F = 3001, 2003, 60000, 40
    A, val, row, col = _rand(dev, (L, M), nnz, 1, fmt=fmt)
    X = torch.randn(M, F, device=dev, requires_grad=True)
    out = A @ X
    want = torch.zeros(L, F, device=dev, dtype=torch.float64).index_add(0, row, val.double().unsqueeze(1) * X.double()[col])
    assert out.shape == (L, F) and _close(out, want)
    w = torch.randn_like(out)
    gx, gv = torch.autograd.grad((out * w).sum(), [X, val])
    wx, wv = torch.autograd.grad((want * w.double()).sum(), [X, val])
    assert _close(gx, wx) and _close(gv, wv)
    # sddmm: (X1 @ X2) sampled at the nonzeros, times the values
    X1 = torch.randn(L, 24, device=dev, requires_grad=True)
    X2 = torch.randn(24, M, device=dev, requires_grad=True)
    C = dglsp.sddmm(A, X1, X2)
    wantc = (X1.double()[row] * X2.double().t()[col]).sum(1) * val.double()
    assert C.shape == A.shape and _close(C.val, wantc)
    w = torch.randn_like(C.val)
    g1 = torch.autograd.grad((C.val * w).sum(), [X1, X2, val])
    g2 = torch.autograd.grad((wantc * w.double()).sum(), [X1, X2, val])
    assert all(_close(a, b) for a, b in zip(g1, g2))


def test_batched_products(dev):
    import dgl_amd.sparse as dglsp

    L, M, nnz, K = 700, 900, 9000, 3
    A, val, row, col = _rand(dev, (L, M), nnz, 2, val_shape=(K,))
    X = torch.randn(M, 5, K, device=dev, requires_grad=True)
    out = dglsp.bspmm(A, X)
    want = torch.zeros(L, 5, K, device=dev, dtype=torch.float64).index_add(0, row, val.double().unsqueeze(1) * X.double()[col])
    assert _close(out, want) and _close(A @ X, want)
    X1 = torch.randn(L, 6, K, device=dev, requires_grad=True)
    X2 = torch.randn(6, M, K, device=dev, requires_grad=True)
    C = dglsp.bsddmm(A, X1, X2)
    wantc = torch.einsum("emk,emk->ek", X1.double()[row], X2.double().permute(1, 0, 2)[col]) * val.double()
    assert C.val.shape == (nnz, K) and _close(C.val, wantc)
    w = torch.randn_like(C.val)
    g1 = torch.autograd.grad((C.val * w).sum(), [X1, X2, val])
    g2 = torch.autograd.grad((wantc * w.double()).sum(), [X1, X2, val])
    assert all(_close(a, b) for a, b in zip(g1, g2))
    A1, v1, r1, c1 = _rand(dev, (L, M), nnz, 3)                       # scalar values, batched operands
    C1 = dglsp.bsddmm(A1, X1, X2)
    want1 = torch.einsum("emk,emk->ek", X1.double()[r1], X2.double().permute(1, 0, 2)[c1]) * v1.double().unsqueeze(1)
    assert _close(C1.val, want1)


@pytest.mark.parametrize("dim", [0, 1])
@pytest.mark.parametrize("val_shape", [(), (4,)])
def test_softmax_and_reductions_along_a_dimension(dev, dim, val_shape):
    import dgl_amd.sparse as dglsp

    L, M, nnz = 1200, 800, 30000
    A, val, row, col = _rand(dev, (L, M), nnz, 4 + dim, val_shape=val_shape)
    seg, n = (row, L) if dim == 1 else (col, M)          # dim = 1: over the nonzeros of a row
    vd = val.double()
    idx = seg.view((-1,) + (1,) * len(val_shape)).expand_as(vd)
    zeros = torch.zeros((n,) + val_shape, device=dev, dtype=torch.float64)
    mx = zeros.scatter_reduce(0, idx, vd, "amax", include_self=False)
    ex = torch.exp(vd - mx[seg])
    want = ex / zeros.index_add(0, seg, ex)[seg]
    S = A.softmax(dim)
    assert S.shape == A.shape and S.val.shape == val.shape and _close(S.val, want)
    w = torch.randn_like(S.val)
    (g1,) = torch.autograd.grad((S.val * w).sum(), [val])
    (g2,) = torch.autograd.grad((want * w.double()).sum(), [val])
    assert _close(g1, g2)
    for name, op in (("sum", "sum"), ("smax", "amax"), ("smin", "amin"), ("smean", "mean")):
        got = getattr(A, name)(dim)
        wantr = zeros.scatter_reduce(0, idx, vd, op, include_self=False)
        assert got.shape == wantr.shape and _close(got, wantr), name
        wr = torch.randn_like(got)
        (g1,) = torch.autograd.grad((got * wr).sum(), [val])
        (g2,) = torch.autograd.grad((wantr * wr.double()).sum(), [val])
        assert _close(g1, g2), name
        assert _close(dglsp.reduce(A, dim, name), wantr)
    assert _close(A.sum(), vd.sum(0)) and _close(A.smax(), vd.amax(0))
    # a row whose true maximum is -inf keeps it (only rows WITHOUT a nonzero give 0), as scatter_reduce does
    r0 = int(row[0])
    v2 = val.detach().clone()
    v2[row == r0] = float("-inf")
    B = dglsp.val_like(A, v2)
    got = B.smax(1)
    assert bool(torch.isinf(got[r0]).all()) and float(got[r0].reshape(-1)[0]) < 0


@pytest.mark.parametrize("dim", [0, 1])
@pytest.mark.parametrize("replace", [False, True])
@pytest.mark.parametrize("bias", [False, True])
def test_sample_picks_nonzeros_of_the_asked_rows(dev, dim, replace, bias):
    import dgl_amd.sparse as dglsp

    L, M, nnz, fan = 400, 300, 6000, 5
    A, val, row, col = _rand(dev, (L, M), nnz, 6)
    A = dglsp.val_like(A, val.detach().abs() + 0.1)
    ids = torch.tensor([7, 7, 0, 399 if dim == 0 else 299, 13], device=dev)
    S = A.sample(dim, fan, ids, replace, bias)
    assert S.shape == ((ids.shape[0], M) if dim == 0 else (L, ids.shape[0]))
    dense = A.to_dense()
    s_major, s_minor = (S.row, S.col) if dim == 0 else (S.col, S.row)
    orig = dense[ids[s_major.long()], s_minor.long()] if dim == 0 else dense[s_minor.long(), ids[s_major.long()]]
    assert torch.equal(orig, S.val) and bool((S.val > 0).all())        # every pick is a nonzero of the asked row, with its value
    deg = (dense != 0).sum(1 - dim)[ids]
    got = torch.bincount(s_major.long(), minlength=ids.shape[0])
    want = torch.where(deg > 0, torch.full_like(deg, fan), deg) if replace else torch.minimum(deg, torch.full_like(deg, fan))
    assert torch.equal(got, want)
    if not replace:
        assert not S.has_duplicate()
    full = A.sample(dim, fan)                                           # ids = None: every row / column
    assert full.shape == A.shape


def test_formats_built_by_the_coo2csr_kernel_and_device_moves(dev):
    import dgl_amd.sparse as dglsp

    A, val, row, col = _rand(dev, (5000, 4000), 80000, 7, dtype=torch.float64)
    dense_idx = row * 4000 + col
    for get, major, minor, n in ((A.csr, row, col, 5000), (A.csc, col, row, 4000)):
        indptr, mi, vidx = get()
        mj = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
        v = val if vidx is None else val[vidx.long()]
        key = (mj * 4000 + mi.long()) if get == A.csr else (mi.long() * 4000 + mj)
        assert bool((mj[1:] >= mj[:-1]).all())
        order = torch.argsort(key)
        assert torch.equal(key[order], torch.sort(dense_idx)[0]) and torch.equal(v[order], val[torch.argsort(dense_idx)])
    B = A.to(device="cpu")
    assert B.device.type == "cpu" and torch.equal(B.to_dense(), A.to_dense().cpu())
    C = B.cuda()
    assert C.device.type == "cuda" and torch.equal(C.to_dense(), A.to_dense())
    assert A.to(device="cuda") is A
