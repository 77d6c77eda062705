"""``dgl_amd.sparse``: the routes that are torch ops on the nonzeros (formats, element-wise / broadcast / unary operators,
coalesce, select, compact, sparse x sparse, ``sprod``), on CPU, against dense torch — the comparisons of the reference's
tests/python/pytorch/sparse/{test_sparse_matrix,test_elementwise_op,test_elementwise_op_sp,test_broadcast,test_matmul,
test_matrix_op,test_reduction}.py, which themselves run unmodified on the GPU through tools/ref_suite (1249 / 1249,
profiles/r5/reference_sparse_suite_on_mi355x.jsonl).  The kernel routes must REFUSE a CPU matrix, not fall back."""
import pytest
import torch

import dgl_amd.sparse as dglsp
from dgl_amd import DGLError


def _rand(shape, nnz, seed, fmt="coo", val_shape=(), dup=False):
    g = torch.Generator().manual_seed(seed)
    if dup:
        row, col = torch.randint(shape[0], (nnz,), generator=g), torch.randint(shape[1], (nnz,), generator=g)
    else:
        lin = torch.randperm(shape[0] * shape[1], generator=g)[:nnz]
        row, col = lin // shape[1], lin % shape[1]
    val = torch.randn((nnz,) + val_shape, generator=g, dtype=torch.float64).requires_grad_(True)
    if fmt == "coo":
        return dglsp.from_coo(row, col, val, shape), val
    major, minor, n = (row, col, shape[0]) if fmt == "csr" else (col, row, shape[1])
    order = torch.argsort(major, stable=True)
    indptr = torch.zeros(n + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(torch.bincount(major, minlength=n), 0)
    with torch.no_grad():
        v2 = val[order].clone()
    v2.requires_grad_(True)
    make = dglsp.from_csr if fmt == "csr" else dglsp.from_csc
    return make(indptr, minor[order], v2, shape), v2


def _dense(A):
    row, col = A.coo()
    out = torch.zeros(A.shape + tuple(A.val.shape[1:]), dtype=A.val.dtype)
    return out.index_put((row, col), A.val, accumulate=True)


@pytest.mark.parametrize("fmt", ["coo", "csr", "csc"])
@pytest.mark.parametrize("val_shape", [(), (3,)])
def test_formats_keep_the_value_order_of_the_creating_format(fmt, val_shape):
    A, val = _rand((7, 5), 16, 1, fmt, val_shape)
    assert A.val is val and A.nnz == 16 and A.shape == (7, 5)
    dense = _dense(A)
    for get, major_is_row in ((A.csr, True), (A.csc, False)):
        indptr, minor, vidx = get()
        v = A.val if vidx is None else A.val[vidx]
        n = A.shape[0] if major_is_row else A.shape[1]
        major = torch.repeat_interleave(torch.arange(n), indptr[1:] - indptr[:-1])
        idx = (major, minor) if major_is_row else (minor, major)
        assert torch.equal(torch.zeros_like(dense).index_put(idx, v, accumulate=True), dense)
        assert bool((major[1:] >= major[:-1]).all())
    if fmt != "coo":
        assert (A.csr() if fmt == "csr" else A.csc())[2] is None
    assert torch.equal(_dense(A.t()), dense.transpose(0, 1))
    assert torch.equal(A.t().t().to_dense(), dense) and torch.equal(A.to_dense(), dense)
    B = dglsp.val_like(A, torch.ones(16, dtype=torch.float64))
    assert B._rel is A._rel and torch.equal(B.to_dense() != 0, dense.reshape(7, 5, -1)[..., 0] != 0)


def test_spmatrix_and_torch_sparse_round_trips_share_storage():
    idx = torch.tensor([[0, 0, 1, 2], [1, 3, 3, 4]])
    val = torch.randn(4)
    A = dglsp.spmatrix(idx, val, (5, 6))
    assert A.indices().data_ptr() == idx.data_ptr() and A.val.data_ptr() == val.data_ptr()
    t = dglsp.to_torch_sparse_coo(A)
    assert t._indices().data_ptr() == idx.data_ptr() and t.shape == (5, 6)
    B = dglsp.from_torch_sparse(t)
    assert B.indices().data_ptr() == idx.data_ptr()
    for conv, lay in ((dglsp.to_torch_sparse_csr, torch.sparse_csr), (dglsp.to_torch_sparse_csc, torch.sparse_csc)):
        ts = conv(A)
        assert ts.layout == lay and torch.equal(ts.to_dense(), A.to_dense())
        assert torch.equal(dglsp.from_torch_sparse(ts).to_dense(), A.to_dense())


def test_print_matches_the_reference_format():
    A = dglsp.from_coo(torch.tensor([1, 1, 3]), torch.tensor([2, 1, 3]), torch.tensor([1.0, 1.0, 2.0]))
    assert str(A) == ("SparseMatrix(indices=tensor([[1, 1, 3],\n                             [2, 1, 3]]),\n"
                      "             values=tensor([1., 1., 2.]),\n             shape=(4, 4), nnz=3)")


def test_coalesce_duplicates_and_diag():
    A = dglsp.from_coo(torch.tensor([1, 0, 0, 0, 1]), torch.tensor([1, 1, 1, 2, 2]), torch.arange(5), (4, 4))
    assert A.has_duplicate()
    C = A.coalesce()
    assert C.row.tolist() == [0, 0, 1, 1] and C.col.tolist() == [1, 2, 1, 2] and C.val.tolist() == [3, 3, 0, 4]
    assert not C.has_duplicate()
    D = dglsp.diag(torch.arange(1.0, 4.0), (3, 5))
    assert D.is_diag() and D.nnz == 3 and D.csr()[0].tolist() == [0, 1, 2, 3] and D.csc()[0].tolist() == [0, 1, 2, 3, 3, 3]
    assert torch.equal(dglsp.identity((3, 3)).to_dense(), torch.eye(3))
    assert torch.equal(D.t().to_dense(), D.to_dense().T)
    Dsq = dglsp.diag(torch.tensor([2.0, 4.0]))
    assert torch.equal(Dsq.inv().val, torch.tensor([0.5, 0.25])) and torch.equal((-Dsq).val, -Dsq.val)
    with pytest.raises(DGLError):
        A.inv()


@pytest.mark.parametrize("fa", ["coo", "csr", "csc"])
@pytest.mark.parametrize("val_shape", [(), (2,)])
def test_elementwise_operators_match_dense(fa, val_shape):
    A, va = _rand((6, 8), 20, 3, fa, val_shape)
    B, vb = _rand((6, 8), 17, 4, "coo", val_shape)
    da, db = _dense(A), _dense(B)
    assert torch.allclose((A + B).to_dense(), da + db) and torch.allclose((A - B).to_dense(), da - db)
    assert torch.allclose(dglsp.add(A, B).to_dense(), da + db) and not (A + B).has_duplicate()
    assert torch.allclose((A * B).to_dense(), da * db)
    assert torch.allclose((A * 2.5).val, va * 2.5) and torch.allclose((2 * A).val, 2 * va)
    assert torch.allclose((A / torch.tensor(4.0)).val, va / 4) and torch.allclose((A ** 2).val, va ** 2)
    Bs = dglsp.val_like(A, torch.rand(va.shape, dtype=torch.float64) + 1)
    q = A / Bs
    assert torch.allclose(q.val, va / Bs.val)
    perm = torch.randperm(A.nnz)                       # same sparsity, another nonzero order: result in A's order
    r, c = A.coo()
    Bp = dglsp.from_coo(r[perm], c[perm], Bs.val[perm], A.shape)
    assert torch.allclose((A / Bp).val, va / Bs.val)
    for bad in (lambda: A + 1, lambda: 1 + A, lambda: A - 2.0, lambda: A ** B, lambda: 2 ** A):
        with pytest.raises(TypeError):
            bad()
    with pytest.raises(DGLError):
        A / B                                          # different sparsities
    # gradients of the sparse x sparse product flow to both value tensors
    w = torch.randn_like((A * B).val)
    ga, gb = torch.autograd.grad(((A * B).val * w).sum(), [va, vb])
    wa = torch.autograd.grad(((da * db))[(A * B).row, (A * B).col].mul(w).sum(), [va, vb])
    assert torch.allclose(ga, wa[0]) and torch.allclose(gb, wa[1])


def test_broadcast_operators():
    A, va = _rand((5, 7), 15, 5)
    r, c = A.coo()
    v_col, v_row = torch.randn(7, dtype=torch.float64), torch.randn(5, 1, dtype=torch.float64)
    assert torch.allclose(dglsp.sp_add_v(A, v_col).val, va + v_col[c])
    assert torch.allclose(dglsp.sp_mul_v(A, v_col.view(1, -1)).val, va * v_col[c])
    assert torch.allclose(dglsp.sp_sub_v(A, v_row).val, va - v_row.view(-1)[r])
    assert torch.allclose(dglsp.sp_div_v(A, v_row).val, va / v_row.view(-1)[r])
    A2, v2 = _rand((5, 7), 15, 6, val_shape=(3,))
    assert torch.allclose(dglsp.sp_broadcast_v(A2, v_col, "mul").val, v2 * v_col[A2.col].view(-1, 1))
    with pytest.raises(DGLError):
        dglsp.sp_add_v(A, torch.randn(6, dtype=torch.float64))


@pytest.mark.parametrize("fmt", ["coo", "csr", "csc"])
def test_select_compact_and_reductions(fmt):
    A, va = _rand((9, 6), 22, 7, fmt)
    d = _dense(A)
    idx = torch.tensor([4, 4, 0, 8])
    assert torch.equal(A.index_select(0, idx).to_dense(), d[idx]) and torch.equal(A.index_select(1, idx[:3]).to_dense(), d[:, idx[:3]])
    assert torch.equal(A.range_select(0, slice(2, 7)).to_dense(), d[2:7]) and torch.equal(A.range_select(1, slice(1, 3)).to_dense(), d[:, 1:3])
    for dim in (0, 1):
        lead = torch.tensor([5, 2, 5])
        C, ids = A.compact(dim, lead)
        nz = (d != 0).any(dim=1 - dim)
        rest = [i for i in range(A.shape[dim]) if bool(nz[i]) and i not in (5, 2)]
        assert ids.tolist() == [5, 2] + rest
        assert torch.equal(C.to_dense(), d.index_select(dim, ids))
        C2, ids2 = A.compact(dim)
        assert ids2.tolist() == [i for i in range(A.shape[dim]) if bool(nz[i])]
    # reductions over torch's scatter_reduce on CPU (the kernels take them on a GPU matrix: tests/test_gpu_sparse.py)
    for dim in (0, 1, None):
        mask = d != 0
        assert torch.allclose(A.sum(dim), d.sum(dim) if dim is not None else d.sum())
        big = torch.where(mask, d, torch.full_like(d, float("-inf")))
        want = big.amax(dim) if dim is not None else big.max()
        assert torch.allclose(A.smax(dim), torch.where(torch.isinf(want), torch.zeros_like(want), want))
        pr = torch.where(mask, d, torch.ones_like(d))
        wantp = pr.prod(dim) if dim is not None else va.prod()
        anyz = mask.any(dim) if dim is not None else torch.tensor(True)
        assert torch.allclose(A.sprod(dim), torch.where(anyz, wantp, torch.zeros_like(wantp)))
    assert torch.allclose(dglsp.reduce(A, 1, "smean"), d.sum(1) / (d != 0).sum(1).clamp(min=1))


@pytest.mark.parametrize("fa,fb", [("coo", "csr"), ("csc", "coo"), ("csr", "csc")])
def test_sparse_times_sparse_matches_dense_with_gradients(fa, fb):
    A, va = _rand((8, 11), 30, 8, fa)
    B, vb = _rand((11, 6), 25, 9, fb)
    C = dglsp.spspmm(A, B)
    want = _dense(A) @ _dense(B)
    assert C.shape == (8, 6) and not C.has_duplicate() and torch.allclose(C.to_dense(), want)
    assert torch.equal(C._keys(), torch.sort(C._keys())[0])            # coalesced: sorted indices
    assert torch.allclose((A @ B).to_dense(), want) and torch.allclose(dglsp.matmul(A, B).to_dense(), want)
    w = torch.randn_like(want)
    g1 = torch.autograd.grad((C.to_dense() * w).sum(), [va, vb])
    g2 = torch.autograd.grad(((_dense(A) @ _dense(B)) * w).sum(), [va, vb])
    assert torch.allclose(g1[0], g2[0]) and torch.allclose(g1[1], g2[1])
    D = dglsp.diag(torch.randn(11, dtype=torch.float64))
    assert torch.allclose((A @ D).to_dense(), _dense(A) @ D.to_dense()) and torch.allclose((D @ B).to_dense(), D.to_dense() @ _dense(B))
    Adup, _ = _rand((8, 11), 40, 10, dup=True)
    with pytest.raises(DGLError):
        dglsp.spspmm(Adup, B)
    with pytest.raises(DGLError):
        dglsp.matmul(torch.randn(3, 8), A)


def test_kernel_routes_refuse_a_cpu_matrix():
    A, _ = _rand((5, 7), 12, 11)
    A = A.float()
    X = torch.randn(7, 3)
    for call in (lambda: dglsp.spmm(A, X), lambda: A @ X, lambda: dglsp.sddmm(A, torch.randn(5, 2), torch.randn(2, 7)),
                 lambda: A.softmax(), lambda: A.sample(0, 2), lambda: dglsp.bspmm(dglsp.val_like(A, torch.randn(12, 2)), torch.randn(7, 3, 2))):
        with pytest.raises(DGLError, match="GPU"):
            call()


def test_to_dtype_and_device_are_noops_when_nothing_changes():
    A, _ = _rand((4, 4), 6, 12)
    assert A.to(dtype=torch.float64) is A and A.to(device="cpu") is A and A.double() is A
    B = A.float()
    assert B.dtype == torch.float32 and B._rel is A._rel and A.long().dtype == torch.int64
