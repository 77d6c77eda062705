"""``dgl.sparse`` front end over the same kernels ("the dgl.sparse front-end can be pointed at
the same kernels").

Mirror of python/dgl/sparse/{sparse_matrix,matmul,sddmm,softmax,reduction,elementwise_op,elementwise_op_sp,broadcast,
unary_op}.py.  In the reference these end in dgl_sparse/src/*.cc, where the arithmetic on the nonzeros is done by
``aten::CSRSpMM`` / ``aten::COOSDDMM`` (matmul.cc:44-139) and the rest by torch ops (reduction.cc:23-70 is
``scatter_reduce``, elemenwise_op.cc:20-34 is a torch sparse add + coalesce).  Here:

  ============================  =====================================================================================
  ``spmm`` / ``bspmm`` / ``@``  g-SpMM ``u_mul_e`` + ``sum`` over the graph whose edge (column -> row) carries the value
  ``sddmm`` / ``bsddmm``        g-SDDMM ``u_dot_v`` times the values
  ``softmax``                   the fused edge-softmax kernels (rows = destinations; ``dim=0``: ``norm_by='src'``)
  ``sum/smax/smin/smean``       g-SpMM ``copy_e`` with that reducer (``dim=0``: over the reverse graph)
  ``sample``                    the neighbour-sampling kernels over the matrix' CSR / CSC
  COO <-> CSR <-> CSC           the COO -> CSR kernel (graph_index.Relation, csrc/coo2csr.hip)
  everything else               torch ops on the nonzeros, on the caller's device: element-wise / broadcast / unary
                                operators, ``sprod``, reductions over everything, coalesce, select, compact, and the
                                sparse x sparse product (expand - sort - compress in torch: rows of B gathered per
                                nonzero of A, one stable sort of the (row, col) keys, ``index_add`` of the products)
  ============================  =====================================================================================

A matrix keeps its nonzero VALUES in the order of the format it was created from and derives the other formats lazily
with a value-index permutation, as the reference does (sparse_matrix.cc / sparse_format.cc): ``csr()`` returns
``(indptr, indices, value_indices)``.  ``.val`` is the very tensor the matrix was created with (a leaf stays a leaf).
Gradients w.r.t. the dense operands AND the values come from the autograd Functions of the operator API.  The kernel
routes need a GPU matrix; on a CPU matrix they raise (no CPU fall-back in this library), the torch routes work anywhere.
"""
import operator as _operator
from numbers import Number

import torch

from . import autograd as _F
from ._lib import DGLAMDError
from .graph_index import GraphIndex, Relation

__all__ = ["SparseMatrix", "spmatrix", "from_coo", "from_csr", "from_csc", "val_like", "diag", "identity",
           "from_torch_sparse", "to_torch_sparse_coo", "to_torch_sparse_csr", "to_torch_sparse_csc",
           "spmm", "bspmm", "spspmm", "matmul", "sddmm", "bsddmm", "softmax", "reduce", "sum", "smax", "smin", "smean",
           "sprod", "add", "sub", "mul", "div", "power", "sp_add", "sp_sub", "sp_mul", "sp_div", "sp_power", "neg", "inv",
           "sp_broadcast_v", "sp_add_v", "sp_sub_v", "sp_mul_v", "sp_div_v"]


def is_scalar(x):
    return isinstance(x, Number) or (torch.is_tensor(x) and x.dim() == 0)


def _need_gpu(A, what):
    if not A.device.type == "cuda":
        raise DGLAMDError("dgl_amd.sparse.%s runs on the HIP kernels: the matrix must live on a GPU (got %s)" % (what, A.device))


class SparseMatrix:
    """(L x M) sparse matrix with one value — a scalar or a vector — per nonzero (sparse_matrix.py:8-760).

    Internally the graph with an edge (column -> row) per nonzero: ``_rel`` holds the formats (its in-edge CSR is the
    matrix' CSR, its out-edge CSR the matrix' CSC) and builds them with the library's own COO -> CSR kernel."""

    def __init__(self, rel, val, shape, is_diag=False):
        self._rel, self._val = rel, val
        self.shape = (int(shape[0]), int(shape[1]))
        self._diag = bool(is_diag)
        self._graph = None
        self._dup = None
        self._ind2d = None   # the (2, nnz) index tensor the matrix was created from, if it was (kept: no copy)

    # ---- attributes ----------------------------------------------------------------------------------------------
    @property
    def val(self):
        return self._val

    @property
    def nnz(self):
        return self._rel.num_edges

    @property
    def dtype(self):
        return self._val.dtype

    @property
    def device(self):
        return self._val.device

    @property
    def row(self):
        return self.coo()[0]

    @property
    def col(self):
        return self.coo()[1]

    # ---- formats -------------------------------------------------------------------------------------------------
    def coo(self):
        """(row, col), in the order of the values."""
        col, row, vidx = self._rel.coo()
        if vidx is not None:   # derived from a format that is itself derived: back into value order
            inv = torch.empty_like(vidx)
            inv[vidx.long()] = torch.arange(vidx.shape[0], dtype=vidx.dtype, device=vidx.device)
            row, col = row[inv.long()], col[inv.long()]
            self._rel._coo = (col, row, None)
        return row, col

    def indices(self):
        if self._ind2d is None:
            self._ind2d = torch.stack(self.coo())
        return self._ind2d

    def csr(self):
        """(indptr, indices, value_indices): ``val[value_indices]`` is in CSR order (None: it already is)."""
        return self._rel.csc()

    def csc(self):
        return self._rel.csr()

    def to_dense(self):
        row, col = self.coo()
        out = torch.zeros(self.shape + tuple(self._val.shape[1:]), dtype=self._val.dtype, device=self.device)
        return out.index_put((row.long(), col.long()), self._val, accumulate=True)

    def _gidx(self):
        """The bipartite graph index the kernels take: node type 0 = columns (sources), 1 = rows (destinations)."""
        if self._graph is None:
            self._graph = GraphIndex([self.shape[1], self.shape[0]], [(0, 1)], [self._rel])
        return self._graph

    # ---- conversions ---------------------------------------------------------------------------------------------
    def t(self):
        return SparseMatrix(self._rel.reverse(), self._val, (self.shape[1], self.shape[0]), self._diag)

    @property
    def T(self):  # noqa: N802
        return self.t()

    def transpose(self):
        return self.t()

    def to(self, device=None, dtype=None):
        device = self.device if device is None else torch.device(device)
        dtype = self.dtype if dtype is None else dtype
        same_dev = device == self.device or (device.type == self.device.type and device.index is None)
        if same_dev and dtype == self.dtype:
            return self
        if same_dev:
            return val_like(self, self._val.to(dtype=dtype))
        row, col = self.coo()
        return from_coo(row.to(device), col.to(device), self._val.to(device=device, dtype=dtype), self.shape)

    def cuda(self):
        return self.to(device="cuda")

    def cpu(self):
        return self.to(device="cpu")

    def float(self):
        return self.to(dtype=torch.float)

    def double(self):
        return self.to(dtype=torch.double)

    def int(self):
        return self.to(dtype=torch.int)

    def long(self):
        return self.to(dtype=torch.long)

    # ---- structure -----------------------------------------------------------------------------------------------
    def _keys(self):
        row, col = self.coo()
        return row.long() * self.shape[1] + col.long()

    def coalesce(self):
        """Unique, lexicographically sorted indices; values of equal indices are added (sparse_matrix_coalesce.cc)."""
        keys = self._keys()
        uniq, inverse = torch.unique(keys, sorted=True, return_inverse=True)
        val = torch.zeros((uniq.shape[0],) + tuple(self._val.shape[1:]), dtype=self._val.dtype, device=self.device)
        val = val.index_add(0, inverse, self._val)
        idt = self._rel.idtype
        return from_coo((uniq // self.shape[1]).to(idt), (uniq % self.shape[1]).to(idt), val, self.shape)

    def has_duplicate(self):
        if self._dup is None:
            self._dup = bool(torch.unique(self._keys()).shape[0] != self.nnz)
        return self._dup

    def is_diag(self):
        return self._diag

    def index_select(self, dim, index):
        """Rows (``dim=0``) or columns (``dim=1``) ``index`` (duplicates allowed) as a new matrix
        (matrix_ops.cc IndexSelect; no autograd in the reference, values are gathered here so it has one)."""
        if dim not in (0, 1):
            raise DGLAMDError("The selection dimension should be 0 or 1.")
        indptr, minor, vidx = self.csr() if dim == 0 else self.csc()
        index = index.to(indptr.device)
        starts, counts = indptr[index.long()].long(), (indptr[index.long() + 1] - indptr[index.long()]).long()
        total = int(counts.sum())
        major = torch.repeat_interleave(torch.arange(index.shape[0], device=indptr.device), counts, output_size=total)
        offs = torch.cumsum(counts, 0) - counts
        pos = starts[major] + (torch.arange(total, device=indptr.device) - offs[major])
        take = pos if vidx is None else vidx[pos].long()
        idt = self._rel.idtype
        new_indptr = torch.zeros(index.shape[0] + 1, dtype=idt, device=indptr.device)
        new_indptr[1:] = torch.cumsum(counts, 0).to(idt)
        val = self._val[take]
        if dim == 0:
            return from_csr(new_indptr, minor[pos], val, (index.shape[0], self.shape[1]))
        return from_csc(new_indptr, minor[pos], val, (self.shape[0], index.shape[0]))

    def range_select(self, dim, index):
        if not isinstance(index, slice) or index.step not in (None, 1):
            raise DGLAMDError("range_select expects a slice with step 1")
        n = self.shape[dim]
        start, stop, _ = index.indices(n)
        return self.index_select(dim, torch.arange(start, max(start, stop), device=self.device))

    def sample(self, dim, fanout, ids=None, replace=False, bias=False):
        """``fanout`` nonzeros of every row (``dim=0``) / column (``dim=1``) in ``ids`` — all of them when it has fewer
        and ``replace`` is False — through the neighbour-sampling kernels (csrc/sampling.hip ≙ RowWiseSampling,
        src/array/cuda/rowwise_sampling.cu); ``bias=True`` weighs the picks with the values.  Row i of the result is
        row ``ids[i]`` of the matrix."""
        from . import _capi

        if dim not in (0, 1):
            raise DGLAMDError("The sampling dimension should be 0 or 1.")
        _need_gpu(self, "SparseMatrix.sample")
        indptr, minor, vidx = self.csr() if dim == 0 else self.csc()
        n_major, n_minor = (self.shape[0], self.shape[1]) if dim == 0 else (self.shape[1], self.shape[0])
        idt = self._rel.idtype
        if ids is None:
            ids = torch.arange(n_major, dtype=idt, device=self.device)
        ids = ids.to(device=self.device, dtype=idt).contiguous()
        csr = _capi.make_csr(indptr.contiguous(), minor.contiguous(), None if vidx is None else vidx.contiguous(), n_minor)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if bias:
            if self._val.dim() != 1:
                raise DGLAMDError("biased sampling needs scalar values")
            p = self._val.detach()
            p = p if p.dtype in (torch.float32, torch.float64) else p.float()
            out_ptr, nbr, picked = _capi.sample_neighbors_weighted(csr, p.contiguous(), ids, int(fanout), bool(replace), seed)
        else:
            out_ptr, nbr, picked = _capi.sample_neighbors(csr, ids, int(fanout), bool(replace), seed)
        n_e = int(out_ptr[-1])
        val = self._val[picked[:n_e].long()]
        if dim == 0:
            return from_csr(out_ptr.to(idt), nbr[:n_e].contiguous(), val, (ids.shape[0], self.shape[1]))
        return from_csc(out_ptr.to(idt), nbr[:n_e].contiguous(), val, (self.shape[0], ids.shape[0]))

    def compact(self, dim, leading_indices=None):
        """Drop the rows / columns without a nonzero and relabel; ``leading_indices`` come first, in the given order,
        whether they hold nonzeros or not (matrix_ops.cc Compact).  Returns (matrix, original ids of the new labels)."""
        if dim not in (0, 1):
            raise DGLAMDError("The compact dimension should be 0 or 1.")
        row, col = self.coo()
        ids = (row if dim == 0 else col).long()
        present = torch.zeros(self.shape[dim], dtype=torch.bool, device=self.device)
        present[ids] = True
        if leading_indices is not None and leading_indices.numel():
            lead = leading_indices.to(self.device).long()
            # first occurrences, in the given order
            first = torch.full((self.shape[dim],), lead.shape[0], dtype=torch.long, device=self.device)
            first = first.scatter_reduce(0, lead, torch.arange(lead.shape[0], device=self.device), "amin")
            lead = lead[first[lead] == torch.arange(lead.shape[0], device=self.device)]
            present[lead] = False
            order = torch.cat([lead, torch.nonzero(present).reshape(-1)])
        else:
            order = torch.nonzero(present).reshape(-1)
        relabel = torch.full((self.shape[dim],), -1, dtype=torch.long, device=self.device)
        relabel[order] = torch.arange(order.shape[0], device=self.device)
        idt = self._rel.idtype
        new = relabel[ids].to(idt)
        if dim == 0:
            out = from_coo(new, col, self._val, (order.shape[0], self.shape[1]))
        else:
            out = from_coo(row, new, self._val, (self.shape[0], order.shape[0]))
        return out, order.to(idt)

    # ---- operators (bound below) ---------------------------------------------------------------------------------
    def __repr__(self):
        return _sparse_matrix_str(self)


def _sparse_matrix_str(spmat):
    indices_str = str(torch.stack(spmat.coo()))
    values_str = str(spmat.val)
    meta_str = "shape={}, nnz={}".format(spmat.shape, spmat.nnz)
    if spmat.val.dim() > 1:
        meta_str += ", val_size={}".format(tuple(spmat.val.shape[1:]))
    prefix = "{}(".format(type(spmat).__name__)

    def indent(s, n):
        lines = s.split("\n")
        return "\n".join([lines[0]] + [" " * n + line for line in lines[1:]])

    body = ("indices=" + indent(indices_str, len("indices=")) + ",\n" + "values=" + indent(values_str, len("values=")) +
            ",\n" + meta_str + ")")
    return prefix + indent(body, len(prefix))


# ---- creation ----------------------------------------------------------------------------------------------------
def _default_val(val, nnz, device):
    return torch.ones(nnz, device=device) if val is None else val


def _check_val(val, nnz):
    if val.dim() not in (1, 2):
        raise DGLAMDError("The values of a SparseMatrix can only be scalars or vectors.")
    if val.shape[0] != nnz:
        raise DGLAMDError("Expect the number of values ({}) to equal the number of nonzeros ({}).".format(val.shape[0], nnz))


def spmatrix(indices, val=None, shape=None):
    """``dglsp.spmatrix(indices, val, shape)`` with ``indices`` of shape (2, nnz) (sparse_matrix.py:763-842)."""
    if indices.dim() != 2 or indices.shape[0] != 2:
        raise DGLAMDError("The indices should be of shape (2, nnz), got {}.".format(tuple(indices.shape)))
    out = from_coo(indices[0], indices[1], val, shape)
    if indices.is_contiguous():
        out._ind2d = indices
    return out


def from_coo(row, col, val=None, shape=None):
    """sparse_matrix.py:845-921.  The index tensors are copied if they are not contiguous; ``val`` is kept as it is."""
    if shape is None:
        shape = (int(row.max()) + 1 if row.numel() else 0, int(col.max()) + 1 if col.numel() else 0)
    val = _default_val(val, row.shape[0], row.device)
    _check_val(val, row.shape[0])
    row, col = row.contiguous(), col.contiguous()
    rel = Relation(shape[1], shape[0], col, row, idtype=row.dtype, device=row.device)
    return SparseMatrix(rel, val, shape)


def from_csr(indptr, indices, val=None, shape=None):
    """sparse_matrix.py:924-1013: the values are in CSR order."""
    if shape is None:
        shape = (indptr.shape[0] - 1, int(indices.max()) + 1 if indices.numel() else 0)
    if indptr.shape[0] - 1 != shape[0]:
        raise DGLAMDError("indptr has {} entries, the matrix {} rows".format(indptr.shape[0], shape[0]))
    val = _default_val(val, indices.shape[0], indices.device)
    _check_val(val, indices.shape[0])
    indptr, indices = indptr.contiguous(), indices.contiguous()
    rel = Relation(shape[1], shape[0], csc=(indptr, indices, None), idtype=indices.dtype, device=indices.device)
    return SparseMatrix(rel, val, shape)


def from_csc(indptr, indices, val=None, shape=None):
    """sparse_matrix.py:1016-1105: the values are in CSC order."""
    if shape is None:
        shape = (int(indices.max()) + 1 if indices.numel() else 0, indptr.shape[0] - 1)
    if indptr.shape[0] - 1 != shape[1]:
        raise DGLAMDError("indptr has {} entries, the matrix {} columns".format(indptr.shape[0], shape[1]))
    val = _default_val(val, indices.shape[0], indices.device)
    _check_val(val, indices.shape[0])
    indptr, indices = indptr.contiguous(), indices.contiguous()
    rel = Relation(shape[1], shape[0], csr=(indptr, indices, None), idtype=indices.dtype, device=indices.device)
    return SparseMatrix(rel, val, shape)


def val_like(mat, val):
    """Same sparsity (shared formats), new values (sparse_matrix.py:1108-1143)."""
    _check_val(val, mat.nnz)
    if val.device != mat.device:
        raise DGLAMDError("The values and the sparse matrix should be on the same device.")
    out = SparseMatrix(mat._rel, val, mat.shape, mat._diag)
    out._dup, out._ind2d, out._graph = mat._dup, mat._ind2d, mat._graph
    return out


def diag(val, shape=None):
    """sparse_matrix.py:1146-1206."""
    if val.dim() > 2:
        raise DGLAMDError("The values of a DiagMatrix can only be scalars or vectors.")
    n = val.shape[0]
    if shape is None:
        shape = (n, n)
    elif n != min(shape):
        raise DGLAMDError("Expect len(val) to be min(shape) for a diagonal matrix, got {} for len(val) and {} for shape."
                          .format(n, shape))
    idx = torch.arange(n, device=val.device)
    out = from_coo(idx, idx, val, shape)
    out._diag, out._dup = True, False
    return out


def identity(shape, d=None, dtype=None, device=None):
    """sparse_matrix.py:1209-1281."""
    n = min(shape)
    val = torch.ones((n,) if d is None else (n, d), dtype=dtype, device=device)
    return diag(val, shape)


def from_torch_sparse(t):
    """sparse_matrix.py:1284-1339."""
    if t.layout == torch.sparse_coo:
        return spmatrix(t._indices(), t._values(), tuple(t.shape[:2]))     # (as they are: no copy, no coalesce)
    if t.layout == torch.sparse_csr:
        return from_csr(t.crow_indices(), t.col_indices(), t.values(), tuple(t.shape[:2]))
    if t.layout == torch.sparse_csc:
        return from_csc(t.ccol_indices(), t.row_indices(), t.values(), tuple(t.shape[:2]))
    raise DGLAMDError("Cannot convert Pytorch sparse tensor with layout {} to DGL sparse.".format(t.layout))


def to_torch_sparse_coo(spmat):
    shape = spmat.shape + tuple(spmat.val.shape[1:])
    return torch.sparse_coo_tensor(spmat.indices(), spmat.val, shape)


def _fmt_vals(spmat, fmt):
    indptr, indices, vidx = fmt
    return indptr, indices, spmat.val if vidx is None else spmat.val[vidx.long()]


def to_torch_sparse_csr(spmat):
    shape = spmat.shape + tuple(spmat.val.shape[1:])
    indptr, indices, val = _fmt_vals(spmat, spmat.csr())
    return torch.sparse_csr_tensor(indptr, indices, val, shape)


def to_torch_sparse_csc(spmat):
    shape = spmat.shape + tuple(spmat.val.shape[1:])
    indptr, indices, val = _fmt_vals(spmat, spmat.csc())
    return torch.sparse_csc_tensor(indptr, indices, val, shape)


# ---- products (matmul.py, sddmm.py) ------------------------------------------------------------------------------
def _is_sp(x, name):
    if not isinstance(x, SparseMatrix):
        raise DGLAMDError("Expect {} to be a SparseMatrix object, got {}.".format(name, type(x)))


def spmm(A, X):
    """``A @ X`` for a sparse (L, M) ``A`` with scalar values and dense ``X`` of shape (M, N) or (M,)
    (matmul.py:12-47 -> matmul.cc:44-88 SpMMNoAutoGrad -> aten::CSRSpMM)."""
    _is_sp(A, "arg1")
    if not isinstance(X, torch.Tensor):
        raise DGLAMDError("Expect arg2 to be a torch.Tensor, got {}.".format(type(X)))
    if A.val.dim() == 2:
        return bspmm(A, X)
    if X.shape[0] != A.shape[1]:
        raise DGLAMDError("spmm: X has {} rows, the sparse matrix {} columns".format(X.shape[0], A.shape[1]))
    _need_gpu(A, "spmm")
    vec = X.dim() == 1
    x = (X.unsqueeze(-1) if vec else X).contiguous()
    val = A.val.contiguous()
    out = _F.gspmm(A._gidx(), "mul", "sum", x, val.reshape((-1,) + (1,) * (x.dim() - 1)))
    return out.squeeze(-1) if vec else out


def bspmm(A, X):
    """Batched: values of length K per nonzero, ``X`` of shape (M, N, K) -> (L, N, K) (matmul.py:50-85)."""
    _is_sp(A, "arg1")
    if A.val.dim() != 2 or X.dim() != 3 or A.val.shape[1] != X.shape[2] or X.shape[0] != A.shape[1]:
        raise DGLAMDError("bspmm expects values of shape (nnz, K) and X of shape (M, N, K)")
    _need_gpu(A, "bspmm")
    return _F.gspmm(A._gidx(), "mul", "sum", X.contiguous(), A.val.contiguous().unsqueeze(1))


def spspmm(A, B):
    """Sparse x sparse (matmul.py:88-129 -> spspmm.cc, cuSPARSE SpGEMM behind aten::CSRMM there).  Here: the rows of B
    are gathered per nonzero of A (expand), the products keyed by (row, col), one stable sort, equal keys added
    (compress).  Differentiable w.r.t. both value tensors; the result is coalesced."""
    _is_sp(A, "arg1")
    _is_sp(B, "arg2")
    if A.shape[1] != B.shape[0]:
        raise DGLAMDError("Expect A.shape[1] ({}) to equal B.shape[0] ({}).".format(A.shape[1], B.shape[0]))
    if A.val.dim() != 1 or B.val.dim() != 1:
        raise DGLAMDError("spspmm only supports scalar nonzero values")
    if A.has_duplicate() or B.has_duplicate():
        raise DGLAMDError("SpSpMM does not support sparse matrices with duplicate entries; call coalesce() first.")
    a_row, a_col = A.coo()
    b_ptr, b_col, b_vidx = B.csr()
    starts = b_ptr[a_col.long()].long()
    counts = (b_ptr[a_col.long() + 1] - b_ptr[a_col.long()]).long()
    total = int(counts.sum())
    dev = A.device
    src = torch.repeat_interleave(torch.arange(A.nnz, device=dev), counts, output_size=total)
    offs = torch.cumsum(counts, 0) - counts
    pos = starts[src] + (torch.arange(total, device=dev) - offs[src])
    b_take = pos if b_vidx is None else b_vidx[pos].long()
    prod = A.val[src] * B.val[b_take]
    keys = a_row[src].long() * B.shape[1] + b_col[pos].long()
    uniq, inverse = torch.unique(keys, sorted=True, return_inverse=True)
    val = torch.zeros(uniq.shape[0], dtype=prod.dtype, device=dev).index_add(0, inverse, prod)
    idt = A._rel.idtype
    return from_coo((uniq // B.shape[1]).to(idt), (uniq % B.shape[1]).to(idt), val, (A.shape[0], B.shape[1]))


def matmul(A, B):
    """``A @ B`` (matmul.py:132-224)."""
    if not isinstance(A, (torch.Tensor, SparseMatrix)):
        raise DGLAMDError("Expect arg1 to be a torch.Tensor or SparseMatrix, got {}.".format(type(A)))
    if not isinstance(B, (torch.Tensor, SparseMatrix)):
        raise DGLAMDError("Expect arg2 to be a torch Tensor or SparseMatrix object, got {}.".format(type(B)))
    if isinstance(A, torch.Tensor) and isinstance(B, torch.Tensor):
        return torch.matmul(A, B)
    if isinstance(A, torch.Tensor):
        raise DGLAMDError("Expect arg2 to be a torch Tensor if arg 1 is torch Tensor, got {}.".format(type(B)))
    if isinstance(B, torch.Tensor):
        return spmm(A, B)
    return spspmm(A, B)


def sddmm(A, X1, X2):
    """``(X1 @ X2) * A`` at the nonzeros of ``A`` (sddmm.py:10-54 -> sddmm.cc -> aten::COOSDDMM).  X1 (L, K) or (L,),
    X2 (K, N) or (N,); with 3-D operands the batched form of :func:`bsddmm`."""
    _is_sp(A, "arg1")
    if X1.dim() == 3 or X2.dim() == 3:
        return bsddmm(A, X1, X2)
    if X1.dim() == 1:
        X1 = X1.unsqueeze(-1)
    if X2.dim() == 1:
        X2 = X2.unsqueeze(0)
    if X1.shape[0] != A.shape[0] or X2.shape[1] != A.shape[1] or X1.shape[1] != X2.shape[0]:
        raise DGLAMDError("sddmm: shapes {} @ {} do not match the sparse matrix {}".format(
            tuple(X1.shape), tuple(X2.shape), A.shape))
    _need_gpu(A, "sddmm")
    # u = column (rows of X2^T), v = row (rows of X1): one dot product per nonzero
    dots = _F.gsddmm(A._gidx(), "dot", X2.t().contiguous(), X1.contiguous(), "u", "v", _handoff=False).squeeze(-1)
    return val_like(A, dots * A.val if A.val.dim() == 1 else dots.unsqueeze(-1) * A.val)


def bsddmm(A, X1, X2):
    """Batched: X1 (L, M, K), X2 (M, N, K); values of shape (nnz,) or (nnz, K); result values (nnz, K) (sddmm.py:57-110)."""
    _is_sp(A, "arg1")
    if X1.dim() != 3 or X2.dim() != 3 or X1.shape[0] != A.shape[0] or X2.shape[1] != A.shape[1] or \
            X1.shape[1] != X2.shape[0] or X1.shape[2] != X2.shape[2]:
        raise DGLAMDError("bsddmm expects X1 of shape (L, M, K) and X2 of shape (M, N, K) for a sparse (L, N) matrix")
    _need_gpu(A, "bsddmm")
    u = X2.permute(1, 2, 0).contiguous()      # (N, K, M): column j's vectors, one per batch entry
    v = X1.permute(0, 2, 1).contiguous()      # (L, K, M)
    dots = _F.gsddmm(A._gidx(), "dot", u, v, "u", "v", _handoff=False).squeeze(-1)     # (nnz, K)
    return val_like(A, dots * (A.val.unsqueeze(-1) if A.val.dim() == 1 else A.val))


# ---- softmax, reductions (softmax.py, reduction.py) --------------------------------------------------------------
def softmax(input, dim=1):  # noqa: A002  (the reference's argument name)
    """Softmax over the nonzeros of every row (``dim=1``) or column (``dim=0``) (softmax.py:11-72 -> softmax.cc:27-95,
    five passes there; the fused edge-softmax kernels here)."""
    _is_sp(input, "input")
    if dim not in (0, 1):
        raise DGLAMDError("The softmax dimension should be 0 or 1.")
    _need_gpu(input, "softmax")
    val = input.val.contiguous()
    scalar = val.dim() == 1
    score = _F.edge_softmax(input._gidx(), val.unsqueeze(-1) if scalar else val, None, "dst" if dim == 1 else "src")
    from . import edge_order as _eo
    score = _eo.plain(score)
    return val_like(input, score.squeeze(-1) if scalar else score)


_REDUCERS = {"sum": "sum", "smax": "max", "smin": "min", "smean": "mean"}


def reduce(input, dim=None, rtype="sum"):  # noqa: A002
    """Reduce the nonzeros along ``dim`` (reduction.py:11-82; reduction.cc:23-90 is torch ``scatter_reduce`` there).
    Rows / columns without a nonzero give 0.  ``dim=None`` reduces all nonzeros."""
    _is_sp(input, "input")
    if rtype not in ("sum", "smax", "smin", "smean", "sprod"):
        raise DGLAMDError("unknown reduce function {}".format(rtype))
    val = input.val
    if dim is None:
        if val.shape[0] == 0 and rtype in ("smax", "smin"):
            raise DGLAMDError("Cannot compute {} of a sparse matrix without nonzeros".format(rtype))
        if rtype == "sum":
            return val.sum(0)
        if rtype == "smax":
            return val.amax(0)
        if rtype == "smin":
            return val.amin(0)
        if rtype == "smean":
            return val.mean(0)
        return val.prod(0)
    if dim not in (0, 1):
        raise DGLAMDError("The reduction dimension should be 0, 1 or None.")
    n_out = input.shape[1] if dim == 0 else input.shape[0]
    if rtype == "sprod" or input.device.type != "cuda" or not val.is_floating_point():
        # not a kernel of this library (no product reducer; integer values): torch, like the reference
        row, col = input.coo()
        idx = (col if dim == 0 else row).long().view((-1,) + (1,) * (val.dim() - 1)).expand_as(val)
        out = torch.zeros((n_out,) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device)
        op = {"sum": "sum", "smax": "amax", "smin": "amin", "smean": "mean", "sprod": "prod"}[rtype]
        return out.scatter_reduce(0, idx, val, op, include_self=False)
    gidx = input._gidx() if dim == 1 else input._gidx().reverse()
    v = val.contiguous()
    scalar = v.dim() == 1
    out = _F.gspmm(gidx, "copy_rhs", _REDUCERS[rtype], None, v.unsqueeze(-1) if scalar else v) \
        if rtype != "smean" else _mean(gidx, v.unsqueeze(-1) if scalar else v)
    if rtype in ("smax", "smin"):   # the kernels leave the reducer's identity in a row / column without a nonzero:
        # zero exactly those (a row whose true maximum is +-inf keeps it, as scatter_reduce(include_self=False) does)
        empty = gidx.relations[0].in_degrees() == 0
        out = torch.where(empty.view((-1,) + (1,) * (out.dim() - 1)), torch.zeros_like(out), out)
    return out.squeeze(-1) if scalar else out


def _mean(gidx, v):
    deg = gidx.relations[0].in_degrees().to(v.dtype).clamp(min=1)
    s = _F.gspmm(gidx, "copy_rhs", "sum", None, v)
    return s / deg.view((-1,) + (1,) * (s.dim() - 1))


def sum(input, dim=None):  # noqa: A001, A002
    return reduce(input, dim, "sum")


def smax(input, dim=None):  # noqa: A002
    return reduce(input, dim, "smax")


def smin(input, dim=None):  # noqa: A002
    return reduce(input, dim, "smin")


def smean(input, dim=None):  # noqa: A002
    return reduce(input, dim, "smean")


def sprod(input, dim=None):  # noqa: A002
    return reduce(input, dim, "sprod")


# ---- element-wise, broadcast and unary operators (elementwise_op_sp.py, broadcast.py, unary_op.py) ---------------
def _same_shape(A, B, what):
    if A.shape != B.shape:
        raise DGLAMDError("Cannot {} sparse matrices of different shapes: {} and {}.".format(what, A.shape, B.shape))
    if A.val.shape[1:] != B.val.shape[1:]:
        raise DGLAMDError("Cannot {} sparse matrices with values of different shapes: {} and {}.".format(
            what, tuple(A.val.shape[1:]), tuple(B.val.shape[1:])))
    if A.device != B.device:
        raise DGLAMDError("Cannot {} sparse matrices on different devices: {} and {}.".format(what, A.device, B.device))


def _spsp_add(A, B):
    _same_shape(A, B, "add")
    if A.is_diag() and B.is_diag():
        return diag(A.val + B.val, A.shape)
    ar, ac = A.coo()
    br, bc = B.coo()
    cat = SparseMatrix(Relation(A.shape[1], A.shape[0], torch.cat([ac, bc.to(ac.dtype)]), torch.cat([ar, br.to(ar.dtype)]),
                                idtype=ar.dtype, device=A.device), torch.cat([A.val, B.val]), A.shape)
    return cat.coalesce()


def _spsp_mul(A, B):
    _same_shape(A, B, "multiply")
    if A.is_diag() and B.is_diag():
        return diag(A.val * B.val, A.shape)
    if A.has_duplicate() or B.has_duplicate():
        raise DGLAMDError("Only support SpSpMul on sparse matrices without duplicate values")
    ka, kb = A._keys(), B._keys()
    kb_sorted, perm = torch.sort(kb)
    pos = torch.searchsorted(kb_sorted, ka).clamp(max=max(kb.shape[0] - 1, 0))
    hit = (kb_sorted[pos] == ka) if kb.shape[0] else torch.zeros_like(ka, dtype=torch.bool)
    lhs = torch.nonzero(hit).reshape(-1)
    rhs = perm[pos[lhs]]
    row, col = A.coo()
    return from_coo(row[lhs], col[lhs], A.val[lhs] * B.val[rhs], A.shape)


def _spsp_div(A, B):
    _same_shape(A, B, "divide")
    if A.is_diag() and B.is_diag():
        return diag(A.val / B.val, A.shape)
    if A.has_duplicate() or B.has_duplicate():
        raise DGLAMDError("Only support SpSpDiv on sparse matrices without duplicate values")
    ka, kb = A._keys(), B._keys()
    sa, pa = torch.sort(ka)
    sb, pb = torch.sort(kb)
    if sa.shape != sb.shape or not torch.equal(sa, sb):
        raise DGLAMDError("Cannot divide two COO matrices with different sparsities.")
    inv = torch.empty_like(pa)
    inv[pa] = torch.arange(pa.shape[0], device=pa.device)
    return val_like(A, A.val / B.val[pb[inv]])       # in the order of the left operand


def sp_add(A, B):
    return _spsp_add(A, B) if isinstance(B, SparseMatrix) else NotImplemented


def sp_sub(A, B):
    return _spsp_add(A, neg(B)) if isinstance(B, SparseMatrix) else NotImplemented


def sp_mul(A, B):
    if is_scalar(B):
        return val_like(A, A.val * B)
    return _spsp_mul(A, B) if isinstance(B, SparseMatrix) else NotImplemented


def sp_div(A, B):
    if is_scalar(B):
        return val_like(A, A.val / B)
    return _spsp_div(A, B) if isinstance(B, SparseMatrix) else NotImplemented


def sp_power(A, scalar):
    return val_like(A, A.val ** scalar) if is_scalar(scalar) else NotImplemented


def _rsub(A, B):
    return NotImplemented


def add(A, B):
    return A + B


def sub(A, B):
    return A - B


def mul(A, B):
    return A * B


def div(A, B):
    return A / B


def power(A, scalar):
    return A ** scalar


def neg(A):
    return val_like(A, -A.val)


def inv(A):
    """Inverse of a square diagonal matrix with scalar values (unary_op.py:29-56)."""
    if not A.is_diag():
        raise DGLAMDError("Non-diagonal sparse matrix does not support inversion.")
    if A.shape[0] != A.shape[1]:
        raise DGLAMDError("Expect a square matrix, got shape {}".format(A.shape))
    if A.val.dim() != 1:
        raise DGLAMDError("inv only supports 1D nonzero val")
    return diag(1.0 / A.val, A.shape)


def sp_broadcast_v(A, v, op):
    """``op(A, v)`` on the nonzeros with ``v`` of shape (1, A.shape[1]) / (A.shape[1],) broadcast over the rows or
    (A.shape[0], 1) over the columns (broadcast.py:10-101)."""
    fn = getattr(_operator, op)
    if v.dim() == 1:
        v = v.view(1, -1)
    msg = "Dimension mismatch for broadcasting. Got A.shape = {} and v.shape = {}.".format(A.shape, tuple(v.shape))
    if not (v.dim() <= 2 and 1 in v.shape):
        raise DGLAMDError(msg)
    bdim = None
    for d, (d1, d2) in enumerate(zip(A.shape, v.shape)):
        if d2 not in (1, d1):
            raise DGLAMDError(msg)
        if d1 != d2:
            if bdim is not None:
                raise DGLAMDError(msg)
            bdim = d
    if bdim is None:
        bdim = 0 if A.shape[0] == 1 else 1
    v = v.reshape(-1)[(A.col if bdim == 0 else A.row).long()]
    if A.val.dim() > 1:
        v = v.view(-1, 1)
    return val_like(A, fn(A.val, v))


def sp_add_v(A, v):
    return sp_broadcast_v(A, v, "add")


def sp_sub_v(A, v):
    return sp_broadcast_v(A, v, "sub")


def sp_mul_v(A, v):
    return sp_broadcast_v(A, v, "mul")


def sp_div_v(A, v):
    return sp_broadcast_v(A, v, "truediv")


SparseMatrix.__add__ = sp_add
SparseMatrix.__sub__ = sp_sub
SparseMatrix.__mul__ = sp_mul
SparseMatrix.__rmul__ = sp_mul
SparseMatrix.__truediv__ = sp_div
SparseMatrix.__pow__ = sp_power
SparseMatrix.__neg__ = neg
SparseMatrix.__matmul__ = matmul
SparseMatrix.neg = neg
SparseMatrix.inv = inv
SparseMatrix.softmax = softmax
SparseMatrix.reduce = reduce
SparseMatrix.sum = sum
SparseMatrix.smax = smax
SparseMatrix.smin = smin
SparseMatrix.smean = smean
SparseMatrix.sprod = sprod
