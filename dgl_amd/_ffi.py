"""Python binding of the registry layer of libdgl_amd.so (``DGLFuncGetGlobal`` /
``DGLFuncCall``), i.e. the same calling convention python/dgl/_ffi/_ctypes/function.py uses
against libdgl.so: arguments are packed into ``DGLValue[]`` + type codes, NDArrays travel as
``DGLArray*`` handles, errors come back as -1 + ``DGLGetLastError()``.

Tensors are NOT copied and no DLPack capsule is needed: a ``DGLArray`` struct is filled
straight from the torch tensor (pointer, shape, dtype, device_type = kDGLROCM = 10).
"""
import ctypes
import struct

import torch

from . import _lib
from ._lib import LIB

# type codes, include/dgl/runtime/c_runtime_api.h:66-91
kObjectInt, kObjectFloat, kHandle, kNull, kArrayHandle, kObjectHandle, kStr = 0, 2, 3, 4, 7, 8, 11
kDGLROCM = 10
kDGLCPU = 1


class DGLValue(ctypes.Union):
    _fields_ = [("v_int64", ctypes.c_int64), ("v_float64", ctypes.c_double),
                ("v_handle", ctypes.c_void_p), ("v_str", ctypes.c_char_p)]


class DGLContext(ctypes.Structure):
    _fields_ = [("device_type", ctypes.c_int32), ("device_id", ctypes.c_int32)]


class DGLDataType(ctypes.Structure):
    _fields_ = [("code", ctypes.c_uint8), ("bits", ctypes.c_uint8), ("lanes", ctypes.c_uint16)]


class DGLArray(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("ctx", DGLContext), ("ndim", ctypes.c_int32),
                ("dtype", DGLDataType), ("shape", ctypes.POINTER(ctypes.c_int64)),
                ("strides", ctypes.POINTER(ctypes.c_int64)), ("byte_offset", ctypes.c_uint64)]


_DT = {
    torch.float32: (2, 32), torch.float64: (2, 64), torch.float16: (2, 16),
    torch.bfloat16: (4, 16), torch.int32: (0, 32), torch.int64: (0, 64), torch.uint8: (1, 8),
}

LIB.DGLGetLastError.restype = ctypes.c_char_p
LIB.DGLFuncGetGlobal.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
# (args / type codes travel as raw arrays: DGLValue is an 8-byte union, written below as int64 words)
LIB.DGLFuncCall.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_int, ctypes.POINTER(DGLValue), ctypes.POINTER(ctypes.c_int)]
LIB.DGLSetStream.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
LIB.DGLObjectFree.argtypes = [ctypes.c_void_p]


class NDArray:
    """A borrowed view of a torch tensor as a ``DGLArray``.  ``unsqueeze`` views a 1-D
    feature as (n, 1) (python/dgl/_sparse_ops.py:208-217)."""

    __slots__ = ("tensor", "_shape", "arr")

    def __init__(self, t, unsqueeze=False, host_ok=False):
        # host_ok: small integer side inputs the reference itself keeps in host memory
        # (segment_mm's seglen, python/dgl/ops/gather_mm.py:57) travel as kDGLCPU arrays
        if type(t) is not torch.Tensor:
            from .edge_order import reject_tagged
            reject_tagged(t)
        if not t.is_cuda and not (host_ok and not t.dtype.is_floating_point):
            raise _lib.DGLAMDError("dgl_amd: tensors must live on a ROCm GPU (no CPU fallback)")
        if not t.is_contiguous():
            raise _lib.DGLAMDError("dgl_amd: tensors handed to the kernels must be contiguous")
        self.tensor = t
        shape = tuple(t.shape)
        if unsqueeze and len(shape) == 1:
            shape = (shape[0], 1)
        self._shape = (ctypes.c_int64 * max(len(shape), 1))(*shape)
        code, bits = _DT[t.dtype]
        ctx = DGLContext(kDGLROCM, t.device.index or 0) if t.is_cuda else DGLContext(kDGLCPU, 0)
        self.arr = DGLArray(t.data_ptr(), ctx, len(shape), DGLDataType(code, bits, 1), self._shape,
                            None, 0)


class _Boxed:
    """A List<Value> made for the duration of one call (convert_to_object,
    python/dgl/_ffi/object_generic.py:27-59): released with DGLObjectFree afterwards."""

    __slots__ = ("handle", "parts", "keep")

    def __init__(self, seq):
        self.keep = list(seq)
        self.parts = []
        for x in self.keep:
            if x is None:
                self.parts.append(None)
            else:
                self.parts.append(get_global_func("_Value")(x))
        self.handle = get_global_func("_List")(*self.parts).handle

    def free(self):
        LIB.DGLObjectFree(ctypes.c_void_p(self.handle))
        for p in self.parts:
            if p is not None:
                LIB.DGLObjectFree(ctypes.c_void_p(p.handle))
        self.parts, self.handle = [], None


_encoded = {}   # str -> (bytes kept alive, address of its buffer)
_arrays = {}    # arity -> (c_int64 array type, c_int array type)
_f2i = struct.Struct("d").pack, struct.Struct("q").unpack


def _pack(args):
    """DGLValue[] + type codes of one call.  The union is written as raw 8-byte words (attribute
    access on a ctypes union array builds a Python object per field: 19 of the 57 us one small
    operator call cost on the host)."""
    n = len(args)
    types = _arrays.get(n)
    if types is None:
        types = _arrays[n] = (ctypes.c_int64 * max(n, 1), ctypes.c_int * max(n, 1))
    values, codes = types[0](), types[1]()
    keep = []
    for i, a in enumerate(args):
        t = type(a)
        if t is NDArray:
            values[i] = ctypes.addressof(a.arr)  # (kept alive through `keep`)
            codes[i] = kArrayHandle
            keep.append(a)
        elif a is None:
            codes[i] = kNull
        elif t is str:
            ent = _encoded.get(a)
            if ent is None:  # operator / reducer names: a handful of distinct strings
                b = a.encode("utf-8")
                ent = (b, ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value)
                if len(_encoded) < 1024:
                    _encoded[a] = ent
            keep.append(ent[0])
            values[i] = ent[1]
            codes[i] = kStr
        elif t is int or t is bool:
            values[i] = int(a)
            codes[i] = kObjectInt
        elif t is ObjectHandle:
            values[i] = a.handle or 0
            codes[i] = kObjectHandle
        elif t is float:
            values[i] = _f2i[1](_f2i[0](a))[0]
            codes[i] = kObjectFloat
        elif isinstance(a, (list, tuple)):
            a = _Boxed(a)
            keep.append(a)
            values[i] = a.handle or 0
            codes[i] = kObjectHandle
        elif isinstance(a, NDArray):
            values[i] = ctypes.addressof(a.arr)
            codes[i] = kArrayHandle
            keep.append(a)
        elif isinstance(a, (bool, int)):
            values[i] = int(a)
            codes[i] = kObjectInt
        elif isinstance(a, float):
            values[i] = _f2i[1](_f2i[0](float(a)))[0]
            codes[i] = kObjectFloat
        elif isinstance(a, ObjectHandle):
            values[i] = a.handle or 0
            codes[i] = kObjectHandle
        else:
            raise TypeError("cannot pass %r through the FFI" % type(a))
    return values, codes, n, keep


class ObjectHandle:
    """Opaque object living on the C++ side (the unit-graph handle)."""

    __slots__ = ("handle",)

    def __init__(self, handle):
        self.handle = handle


class Function:
    """A registered global function (python/dgl/_ffi/function.py:Function)."""

    __slots__ = ("name", "handle")

    def __init__(self, name, handle):
        self.name, self.handle = name, handle

    def __call__(self, *args):
        values, codes, n, keep = _pack(args)
        ret, ret_code = DGLValue(), ctypes.c_int(kNull)
        rc = LIB.DGLFuncCall(self.handle, values, codes, n, ctypes.byref(ret), ctypes.byref(ret_code))
        for k in keep:
            if type(k) is _Boxed:
                k.free()
        del keep
        if rc != 0:
            raise _lib.DGLAMDError(LIB.DGLGetLastError().decode("utf-8", "replace"))
        if ret_code.value == kNull:
            return None
        if ret_code.value == kObjectInt:
            return ret.v_int64
        if ret_code.value == kObjectFloat:
            return ret.v_float64
        if ret_code.value in (kObjectHandle, kHandle):
            return ObjectHandle(ret.v_handle)
        raise _lib.DGLAMDError("unsupported FFI return type code %d" % ret_code.value)


_cache = {}


def get_global_func(name):
    f = _cache.get(name)
    if f is None:
        h = ctypes.c_void_p()
        LIB.DGLFuncGetGlobal(name.encode(), ctypes.byref(h))
        if not h.value:
            raise _lib.DGLAMDError("global function %s is not registered" % name)
        f = _cache[name] = Function(name, h.value)
    return f


def list_global_func_names():
    n = ctypes.c_int()
    arr = ctypes.POINTER(ctypes.c_char_p)()
    LIB.DGLFuncListGlobalNames(ctypes.byref(n), ctypes.byref(arr))
    return [arr[i].decode() for i in range(n.value)]


def use_current_stream(device):
    """Queue the following kernel calls of this thread on PyTorch's current stream of
    `device` (reference: tensoradapter CUDACurrentStream, src/runtime/cuda/cuda_device_api.cc:362-367)."""
    # an index-less torch.device("cuda") means the CURRENT device (rank r under
    # torch.cuda.set_device(r)), not device 0: DGLFuncCall makes this id current around the call
    idx = device.index if device.index is not None else torch.cuda.current_device()
    LIB.DGLSetStream(kDGLROCM, idx, torch.cuda.current_stream(idx).cuda_stream)


# ---- DLPack hand-over (python/dgl/_ffi/_ctypes/ndarray.py:28-45) ---------------------------------
LIB.DGLArrayFromDLPack.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
LIB.DGLArrayToDLPack.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
LIB.DGLArrayFree.argtypes = [ctypes.c_void_p]
ctypes.pythonapi.PyCapsule_GetPointer.restype = ctypes.c_void_p
ctypes.pythonapi.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
ctypes.pythonapi.PyCapsule_SetName.argtypes = [ctypes.py_object, ctypes.c_char_p]
ctypes.pythonapi.PyCapsule_New.restype = ctypes.py_object
ctypes.pythonapi.PyCapsule_New.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]


class OwnedArray:
    """An array made by ``DGLArrayFromDLPack`` from a ``torch.utils.dlpack.to_dlpack`` capsule:
    the library owns the DLManagedTensor and releases it in ``DGLArrayFree``.  Usable wherever
    an ``NDArray`` argument is (the handle points at a ``DGLArray``)."""

    def __init__(self, capsule):
        ptr = ctypes.pythonapi.PyCapsule_GetPointer(capsule, b"dltensor")
        h = ctypes.c_void_p()
        if LIB.DGLArrayFromDLPack(ptr, ctypes.byref(h)) != 0:
            raise _lib.DGLAMDError(LIB.DGLGetLastError().decode("utf-8", "replace"))
        ctypes.pythonapi.PyCapsule_SetName(capsule, b"used_dltensor")  # consumed, as the reference does
        self.handle = h.value
        self.arr = ctypes.cast(h, ctypes.POINTER(DGLArray)).contents

    def to_dlpack(self):
        out = ctypes.c_void_p()
        if LIB.DGLArrayToDLPack(self.handle, ctypes.byref(out), 0) != 0:
            raise _lib.DGLAMDError(LIB.DGLGetLastError().decode("utf-8", "replace"))
        return ctypes.pythonapi.PyCapsule_New(out, b"dltensor", None)

    def free(self):
        if self.handle:
            LIB.DGLArrayFree(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def from_dlpack(capsule):
    return OwnedArray(capsule)
