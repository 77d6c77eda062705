"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/dgl_amd.h
declares, the registry lists the reference's global names, and the product package never
touches the oracle."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "dgl_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b((?:dgla_|DGL)[A-Za-z_0-9]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported():
    from dgl_amd import _lib

    names = _declared_functions()
    assert "dgla_spmm_csr" in names and "DGLFuncCall" in names and len(names) >= 18
    for n in names:
        assert hasattr(_lib.LIB, n), "libdgl_amd.so does not export " + n


def test_registry_lists_reference_names():
    from dgl_amd import _lib

    n = ctypes.c_int()
    arr = ctypes.POINTER(ctypes.c_char_p)()
    assert _lib.LIB.DGLFuncListGlobalNames(ctypes.byref(n), ctypes.byref(arr)) == 0
    names = {arr[i].decode() for i in range(n.value)}
    for want in ("sparse._CAPI_DGLKernelSpMM", "sparse._CAPI_DGLKernelSDDMM",
                 "sparse._CAPI_DGLKernelSpMMHetero", "sparse._CAPI_DGLKernelSDDMMHetero",
                 "sparse._CAPI_DGLKernelEdge_softmax_forward",
                 "sparse._CAPI_DGLKernelEdge_softmax_backward",
                 "sparse._CAPI_DGLKernelSegmentReduce", "sparse._CAPI_DGLKernelScatterAdd",
                 "sparse._CAPI_DGLKernelBwdSegmentCmp", "sparse._CAPI_DGLKernelSEGMENTMM",
                 "sparse._CAPI_DGLKernelSEGMENTMMBackwardB", "sparse._CAPI_DGLKernelGATHERMM",
                 "sparse._CAPI_DGLKernelGATHERMMSCATTER", "_List", "_Value"):
        assert want in names
    h = ctypes.c_void_p()
    assert _lib.LIB.DGLFuncGetGlobal(b"sparse._CAPI_DGLKernelSpMM", ctypes.byref(h)) == 0 and h.value
    assert _lib.LIB.DGLFuncGetGlobal(b"no.such.function", ctypes.byref(h)) == 0 and not h.value


def test_errors_cross_the_abi_as_strings():
    from dgl_amd import _lib

    # a call with a NULL csr must fail with -1 and a message, not crash
    rc = _lib.LIB.dgla_spmm_csr(b"copy_lhs", b"sum", None, 0, None, None, None, None, None, None,
                                0, 0, None)
    assert rc == -1 and b"null" in _lib.LIB.dgla_last_error()
    _lib.LIB.DGLAPISetLastError(b"hello")
    _lib.LIB.DGLGetLastError.restype = ctypes.c_char_p
    assert _lib.LIB.DGLGetLastError() == b"hello"


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dgl_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_cpu_tensor_is_refused():
    import pytest
    import torch

    from dgl_amd import DGLAMDError, _capi

    with pytest.raises(DGLAMDError, match="no CPU fallback|ROCm GPU"):
        _capi.make_csr(torch.zeros(3, dtype=torch.int32), torch.zeros(2, dtype=torch.int32), None, 2)


def test_list_objects_and_handle_tags():
    """`_List` / `_Value` box Python lists the way the reference's FFI does; handles carry a type
    tag, so a wrong handle is an error message, not a crash."""
    import pytest

    from dgl_amd import _ffi, _lib

    v = _ffi.get_global_func("_Value")(7)
    lst = _ffi.get_global_func("_List")(v, None, 3)
    assert v.handle and lst.handle
    # a list where a heterograph handle is expected
    f = _ffi.get_global_func("sparse._CAPI_DGLKernelSDDMMHetero")
    with pytest.raises(_lib.DGLAMDError, match="heterograph handle"):
        f(lst, "add", [], [], [], 0, 2)
    # an int where a list is expected
    with pytest.raises(_lib.DGLAMDError, match="expected a list"):
        f(lst, "add", 1, [], [], 0, 2)
    for h in (lst, v):
        assert _lib.LIB.DGLObjectFree(ctypes.c_void_p(h.handle)) == 0
    bogus = (ctypes.c_uint32 * 4)(123, 0, 0, 0)
    assert _lib.LIB.DGLObjectFree(ctypes.cast(bogus, ctypes.c_void_p)) == -1


def test_dlpack_round_trip_shares_memory_and_releases_owner():
    """torch -> to_dlpack -> DGLArrayFromDLPack -> (fields) -> DGLArrayToDLPack -> torch: one
    buffer throughout; the producer's deleter runs exactly when the last holder lets go
    (src/runtime/dlpack_convert.cc:57-140; CPU tensor: no GPU needed)."""
    import gc
    import weakref

    import torch
    from torch.utils import dlpack

    from dgl_amd import _ffi

    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    alive = weakref.ref(t.untyped_storage())
    arr = _ffi.from_dlpack(dlpack.to_dlpack(t))
    a = arr.arr
    assert a.ndim == 2 and a.shape[0] == 3 and a.shape[1] == 4
    assert a.dtype.code == 2 and a.dtype.bits == 32 and a.ctx.device_type == 1   # kDLCPU
    assert a.data == t.data_ptr()
    back = dlpack.from_dlpack(arr.to_dlpack())
    assert back.data_ptr() == t.data_ptr() and torch.equal(back, t)
    back[0, 0] = 99.0
    assert float(t[0, 0]) == 99.0
    ptr = t.data_ptr()
    del t
    arr.free()            # the array lets go; `back` still holds the memory through its capsule
    gc.collect()
    assert back.data_ptr() == ptr and float(back[0, 0]) == 99.0 and alive() is not None
    del back
    gc.collect()
    assert alive() is None


def test_tuning_constants_match_header_and_library_default():
    """DGLA_TUNE_* in include/dgl_amd.h == the Python mirror; the library's default (XCD order +
    LDS-direct segment_mm kernels) is what dgla_get_tuning() reports in a fresh process, and
    dgla_set_tuning round-trips (no GPU needed: flags are host state)."""
    import os
    import re

    from dgl_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "dgl_amd.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (DGLA_TUNE_\w+) (\d+)u", text)}
    assert defs == {"DGLA_TUNE_XCD": 1, "DGLA_TUNE_SPLIT": 8, "DGLA_TUNE_GLDS": 16, "DGLA_TUNE_SPLIT_FORCE": 64,
                    "DGLA_TUNE_MM_F32": 128, "DGLA_TUNE_MM_X3": 2048, "DGLA_TUNE_NO_GATE": 4096, "DGLA_TUNE_NO_STAGE_W": 8192}
    for name, value in defs.items():
        assert getattr(_lib, name) == value
    default = int(_lib.LIB.dgla_get_tuning())
    # the library's compiled-in default (csrc/common.h) and the header's documented one agree
    common = open(os.path.join(root, "dgl_amd", "csrc", "common.h")).read()
    bits = re.search(r"constexpr uint32_t kDefaultTuning = ([^;]+);", common).group(1)
    assert default == eval(bits.replace("u", "")) == _lib.DEFAULT_TUNING
    try:
        assert _lib.LIB.dgla_set_tuning(9) == 0 and int(_lib.LIB.dgla_get_tuning()) == 9
    finally:
        _lib.LIB.dgla_set_tuning(default)
