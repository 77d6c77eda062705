"""Stand-in for ``dgl.backend`` (python/dgl/backend/pytorch/tensor.py) holding exactly the tensor helpers the
reference's test backend (tests/backend/__init__.py) and the suites listed in run.py call.  Test infrastructure."""
import numpy as np
import torch as th

backend_name = "pytorch"
float16, bfloat16, float32, float64 = th.float16, th.bfloat16, th.float32, th.float64
uint8, int8, int16, int32, int64, bool = th.uint8, th.int8, th.int16, th.int32, th.int64, th.bool
data_type_dict = {"float16": th.float16, "float32": th.float32, "float64": th.float64, "int32": th.int32,
                  "int64": th.int64, "bool": th.bool}


def cpu():
    return th.device("cpu")


def tensor(data, dtype=None):
    if isinstance(data, th.Tensor):
        return data.clone().to(dtype) if dtype is not None else data.clone()
    if isinstance(data, np.ndarray):
        t = th.from_numpy(data)
        return t.to(dtype) if dtype is not None else t
    return th.tensor(data, dtype=dtype)


def as_scalar(data):
    return data.item()


def shape(x):
    return x.shape


def dtype(x):
    return x.dtype


def ndim(x):
    return x.dim()


def context(x):
    return x.device


def astype(x, ty):
    return x.type(ty)


def asnumpy(x):
    return x.detach().cpu().numpy()


def zerocopy_to_numpy(x):
    return x.detach().cpu().numpy()


def zerocopy_from_numpy(x):
    return th.as_tensor(x)


def copy_to(x, ctx, **kwargs):
    return x.to(ctx)


def sum(x, dim, keepdims=False):
    return th.sum(x, dim=dim, keepdim=keepdims)


def mean(x, dim):
    return th.mean(x, dim=dim)


def max(x, dim):
    return th.max(x, dim=dim)[0]


def min(x, dim):
    return th.min(x, dim=dim)[0]


def reduce_sum(x):
    return x.sum()


def softmax(x, dim=-1):
    return th.softmax(x, dim=dim)


def cat(seq, dim):
    return th.cat(seq, dim=dim)


def stack(seq, dim):
    return th.stack(seq, dim=dim)


def reshape(x, shp):
    return x.view(shp)


def unsqueeze(x, dim):
    return th.unsqueeze(x, dim)


def squeeze(x, dim):
    return th.squeeze(x, dim)


def gather_row(data, row_index):
    return th.index_select(data, 0, row_index.long())


def repeat(x, repeats, dim):
    return th.repeat_interleave(x, repeats, dim)


def zeros(shp, dtype, ctx):
    return th.zeros(shp, dtype=dtype, device=ctx)


def zeros_like(x):
    return th.zeros_like(x)


def ones(shp, dtype, ctx):
    return th.ones(shp, dtype=dtype, device=ctx)


def randn(shp):
    return th.randn(*shp)


def full(shp, fill_value, dtype, ctx):
    return th.full(shp, fill_value, dtype=dtype, device=ctx)


def full_1d(length, fill_value, dtype, ctx):
    return th.full((length,), fill_value, dtype=dtype, device=ctx)


def arange(start, stop, dtype=th.int64, ctx=None):
    return th.arange(start, stop, dtype=dtype, device=ctx)


def unique(x, return_inverse=False, return_counts=False):
    return th.unique(x, return_inverse=return_inverse, return_counts=return_counts)


def boolean_mask(x, mask):
    return x[mask]


def nonzero_1d(x):
    return th.nonzero(x, as_tuple=False).squeeze(-1)


def sort_1d(x):
    return th.sort(x)


def equal(x, y):
    return x == y


def logical_not(x):
    return ~x


def count_nonzero(x):
    return int((x != 0).sum())


def abs(x):
    return x.abs()


def clone(x):
    return x.clone()


def replace_inf_with_zero(x):
    return th.masked_fill(x, th.isinf(x), 0)


# ---- autograd helpers (python/dgl/backend/pytorch/tensor.py attach_grad / backward / grad / record_grad) ----------
def attach_grad(x):
    if x.grad is not None:
        x.grad.zero_()
        return x
    return x.requires_grad_()


def backward(x, head_gradient=None):
    if head_gradient is not None and head_gradient.shape[0] == 1 and len(head_gradient.shape) == 1:
        head_gradient = th.tensor(head_gradient.item()).to(head_gradient.device)   # as the reference does
    x.backward(head_gradient)


def grad(x):
    return x.grad


def is_no_grad(x):
    return x.grad is None or (x.grad == 0).all()


def is_recording():
    return th.is_grad_enabled()


class record_grad(object):
    def __enter__(self):
        pass

    def __exit__(self, exc_type, exc_value, exc_traceback):
        pass


no_grad = th.no_grad
