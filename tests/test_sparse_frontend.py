"""dgl.sparse-style front end (spmm / bspmm / sddmm on a SparseMatrix) over the same kernels,
against dense torch evaluation, values and gradients (tests/python/pytorch/sparse/test_matmul.py,
test_sddmm.py of the reference do the same comparisons)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand_sparse(dev, L, M, nnz, seed, val_shape=()):
    import dgl_amd.sparse as dglsp

    g = torch.Generator().manual_seed(seed)
    lin = torch.randperm(L * M, generator=g)[:nnz]            # distinct positions
    row, col = (lin // M).to(dev), (lin % M).to(dev)
    val = torch.randn((nnz,) + val_shape, generator=g).to(dev).requires_grad_(True)
    return dglsp.spmatrix(torch.stack([row, col]), val, (L, M)), val


@pytest.mark.parametrize("xshape", [(7,), (7, 1), (7, 33)])
def test_spmm_matches_dense_with_gradients(dev, xshape):
    import dgl_amd.sparse as dglsp

    A, val = _rand_sparse(dev, 5, 7, 20, 1)
    X = torch.randn(xshape, device=dev, requires_grad=True)
    out = dglsp.spmm(A, X)
    assert torch.equal(out, A @ X)
    dense = torch.zeros(5, 7, device=dev).index_put((A.row, A.col), val)
    want = dense @ X
    assert out.shape == want.shape and torch.allclose(out, want, atol=1e-5)
    w = torch.randn_like(out)
    g1 = torch.autograd.grad((out * w).sum(), [X, val], retain_graph=True)
    g2 = torch.autograd.grad((want * w).sum(), [X, val])
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, atol=1e-5)


def test_bspmm_and_sddmm_match_dense(dev):
    import dgl_amd.sparse as dglsp

    A, val = _rand_sparse(dev, 6, 9, 25, 2, val_shape=(3,))
    X = torch.randn(9, 4, 3, device=dev, requires_grad=True)
    out = dglsp.bspmm(A, X)
    dense = torch.zeros(6, 9, 3, device=dev).index_put((A.row, A.col), val)
    want = torch.einsum("lmk,mnk->lnk", dense, X)
    assert torch.allclose(out, want, atol=1e-5)
    # docstring example shape: python/dgl/sparse/sddmm.py:42-51
    idx = torch.tensor([[1, 1, 2], [2, 3, 3]], device=dev)
    v = torch.arange(1, 4, device=dev).float().requires_grad_(True)
    B = dglsp.spmatrix(idx, v, (3, 4))
    X1 = torch.randn(3, 5, device=dev, requires_grad=True)
    X2 = torch.randn(5, 4, device=dev, requires_grad=True)
    C = dglsp.sddmm(B, X1, X2)
    want = (X1 @ X2)[idx[0], idx[1]] * v
    assert C.shape == (3, 4) and torch.allclose(C.val, want, atol=1e-5)
    g1 = torch.autograd.grad(C.val.sum(), [X1, X2, v], retain_graph=True)
    g2 = torch.autograd.grad(want.sum(), [X1, X2, v])
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, atol=1e-5)
    # 1-D operands: outer product
    D = dglsp.sddmm(B, torch.arange(3, device=dev).float(), torch.arange(4, device=dev).float())
    assert torch.allclose(D.val, (idx[0] * idx[1]).float() * v)
    assert torch.allclose(B.to_dense()[idx[0], idx[1]], v)
