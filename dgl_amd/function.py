"""Built-in message / reduce function descriptors (``dgl.function``).

Names and fields follow python/dgl/function/message.py and reducer.py: ``copy_u``, ``copy_e``,
``u_add_v`` ... ``e_dot_v`` for messages and ``sum`` / ``max`` / ``min`` / ``mean`` for reducers.
They carry no arithmetic; ``DGLGraph.update_all`` / ``apply_edges`` map them to the fused
operators in :mod:`dgl_amd.ops` (python/dgl/core.py:273-425).
"""
import sys

_TARGET_CODE = {"u": 0, "v": 1, "e": 2}  # index into [srcdata, dstdata, edata] (core.py:288)


class BuiltinFunction:
    name = None


class CopyMessageFunction(BuiltinFunction):
    def __init__(self, target, in_field, out_field):
        self.target = _TARGET_CODE[target]
        self.in_field, self.out_field = in_field, out_field
        self.name = "copy_" + target

    def __repr__(self):
        return "fn.{}('{}', '{}')".format(self.name, self.in_field, self.out_field)


class BinaryMessageFunction(BuiltinFunction):
    def __init__(self, binary_op, lhs, rhs, lhs_field, rhs_field, out_field):
        self.binary_op = binary_op
        self.lhs, self.rhs = _TARGET_CODE[lhs], _TARGET_CODE[rhs]
        self.lhs_field, self.rhs_field, self.out_field = lhs_field, rhs_field, out_field
        self.name = "{}_{}_{}".format(lhs, binary_op, rhs)

    def __repr__(self):
        return "fn.{}('{}', '{}', '{}')".format(self.name, self.lhs_field, self.rhs_field, self.out_field)


class SimpleReduceFunction(BuiltinFunction):
    def __init__(self, name, msg_field, out_field):
        self.name, self.msg_field, self.out_field = name, msg_field, out_field

    def __repr__(self):
        return "fn.{}('{}', '{}')".format(self.name, self.msg_field, self.out_field)


def copy_u(u, out):
    return CopyMessageFunction("u", u, out)


def copy_e(e, out):
    return CopyMessageFunction("e", e, out)


# deprecated spellings kept by the reference (function/message.py)
copy_src = copy_u
copy_edge = copy_e

_mod = sys.modules[__name__]
__all__ = ["copy_u", "copy_e", "copy_src", "copy_edge", "sum", "max", "min", "mean"]
for _l in "uve":
    for _r in "uve":
        if _l == _r:
            continue
        for _b in ("add", "sub", "mul", "div", "dot"):
            def _make(l=_l, r=_r, b=_b):
                def f(lhs_field, rhs_field, out):
                    return BinaryMessageFunction(b, l, r, lhs_field, rhs_field, out)
                f.__name__ = "{}_{}_{}".format(l, b, r)
                return f
            setattr(_mod, "{}_{}_{}".format(_l, _b, _r), _make())
            __all__.append("{}_{}_{}".format(_l, _b, _r))


def sum(msg, out):  # noqa: A001  (same public name as the reference)
    return SimpleReduceFunction("sum", msg, out)


def max(msg, out):  # noqa: A001
    return SimpleReduceFunction("max", msg, out)


def min(msg, out):  # noqa: A001
    return SimpleReduceFunction("min", msg, out)


def mean(msg, out):
    return SimpleReduceFunction("mean", msg, out)
