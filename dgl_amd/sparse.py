"""Minimal ``dgl.sparse`` front end over the same kernels ("the dgl.sparse
front-end can be pointed at the same kernels"): ``spmatrix`` / ``from_coo`` / ``from_csr``,
``spmm``, ``bspmm``, ``sddmm``.

Mirror of python/dgl/sparse/{sparse_matrix.py,matmul.py:12-90,sddmm.py:10-53}; in the reference
these end in dgl_sparse/src/matmul.cc:44-139, which calls aten::CSRSpMM / aten::COOSDDMM — the
graph-free seam this library replaces.  A (L x M) sparse matrix is the graph whose edge
(col -> row) carries the value: ``A @ X`` is ``u_mul_e`` + ``sum``, ``(X1 @ X2) * A`` is
``u_dot_v`` times the values.  Gradients w.r.t. the dense operands AND the values come from the
same autograd Functions the operator API uses.
"""
import torch

from . import autograd as _F
from ._lib import DGLAMDError
from .graph_index import GraphIndex, Relation


class SparseMatrix:
    """COO-backed sparse matrix with a value per nonzero (scalar, or a vector for ``bspmm``)."""

    def __init__(self, row, col, val, shape):
        self.row, self.col, self.val = row, col, val
        self.shape = (int(shape[0]), int(shape[1]))
        # edges run column -> row: source nodes are columns, destination nodes are rows
        rel = Relation(self.shape[1], self.shape[0], col, row, idtype=row.dtype, device=row.device)
        self._gidx = GraphIndex([self.shape[1], self.shape[0]], [(0, 1)], [rel])

    @property
    def nnz(self):
        return int(self.row.shape[0])

    @property
    def device(self):
        return self.row.device

    @property
    def dtype(self):
        return self.val.dtype

    def coo(self):
        return self.row, self.col

    def indices(self):
        return torch.stack([self.row, self.col])

    def csr(self):
        """(indptr, indices, value_indices) over the rows, like SparseMatrix.csr()."""
        return self._gidx.relations[0].csc()

    def to_dense(self):
        out = torch.zeros(self.shape + tuple(self.val.shape[1:]), dtype=self.val.dtype, device=self.device)
        return out.index_put_((self.row.long(), self.col.long()), self.val, accumulate=True)

    def __matmul__(self, X):
        return spmm(self, X)

    def __repr__(self):
        return "SparseMatrix(indices={}, values={}, shape={}, nnz={})".format(
            self.indices(), self.val, self.shape, self.nnz)


def spmatrix(indices, val=None, shape=None):
    """``dglsp.spmatrix(indices, val, shape)`` with ``indices`` of shape (2, nnz)."""
    return from_coo(indices[0], indices[1], val, shape)


def from_coo(row, col, val=None, shape=None):
    row, col = row.contiguous(), col.contiguous()
    if val is None:
        val = torch.ones(row.shape[0], device=row.device)
    if shape is None:
        shape = (int(row.max()) + 1 if row.numel() else 0, int(col.max()) + 1 if col.numel() else 0)
    return SparseMatrix(row, col, val, shape)


def from_csr(indptr, indices, val=None, shape=None):
    n = indptr.shape[0] - 1
    row = torch.repeat_interleave(torch.arange(n, device=indptr.device, dtype=indptr.dtype),
                                  (indptr[1:] - indptr[:-1]).long())
    return from_coo(row, indices, val, shape if shape is not None else
                    (n, int(indices.max()) + 1 if indices.numel() else 0))


def spmm(A, X):
    """``A @ X`` for a sparse (L, M) ``A`` with scalar values and dense ``X`` of shape (M, N) or (M,)."""
    if not isinstance(A, SparseMatrix):
        raise DGLAMDError("Expect arg1 to be a SparseMatrix object, got {}.".format(type(A)))
    if A.val.dim() != 1:
        raise DGLAMDError("spmm expects scalar values; use bspmm for vector values")
    if X.shape[0] != A.shape[1]:
        raise DGLAMDError("spmm: X has {} rows, the sparse matrix {} columns".format(X.shape[0], A.shape[1]))
    vec = X.dim() == 1
    x = X.unsqueeze(-1) if vec else X
    out = _F.gspmm(A._gidx, "mul", "sum", x, A.val.reshape((-1,) + (1,) * (x.dim() - 1)))
    return out.squeeze(-1) if vec else out


def bspmm(A, X):
    """Batched: values of length K per nonzero, ``X`` of shape (M, N, K) -> (L, N, K)."""
    if A.val.dim() != 2 or X.dim() != 3 or A.val.shape[1] != X.shape[2]:
        raise DGLAMDError("bspmm expects values of shape (nnz, K) and X of shape (M, N, K)")
    return _F.gspmm(A._gidx, "mul", "sum", X, A.val.unsqueeze(1))


def sddmm(A, X1, X2):
    """``(X1 @ X2) * A`` at the nonzeros of ``A``: X1 (L, K) or (L,), X2 (K, N) or (N,)."""
    if X1.dim() == 1:
        X1 = X1.unsqueeze(-1)
    if X2.dim() == 1:
        X2 = X2.unsqueeze(0)
    if X1.shape[0] != A.shape[0] or X2.shape[1] != A.shape[1] or X1.shape[1] != X2.shape[0]:
        raise DGLAMDError("sddmm: shapes {} @ {} do not match the sparse matrix {}".format(
            tuple(X1.shape), tuple(X2.shape), A.shape))
    # u = column (rows of X2^T), v = row (rows of X1): one dot product per nonzero
    dots = _F.gsddmm(A._gidx, "dot", X2.t().contiguous(), X1.contiguous(), "u", "v")
    return SparseMatrix(A.row, A.col, dots.squeeze(-1) * A.val, A.shape)
