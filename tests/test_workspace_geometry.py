"""Host-side geometry of dgla_spmm_csr_workspace_bytes (no GPU: the function is arithmetic over
shapes): what each layout adds to the scratch of a call, and the unit size of small graphs.
Mirrors the decisions of csrc/spmm_csr.hip.h::spmm_geometry / spmm_split_shape_ok / spmm_tail_slices."""
import ctypes

import pytest

from dgl_amd import _lib

LIB = _lib.LIB
F32, F64, F16, BF16 = 0, 1, 2, 3
ESIZE = {F32: 4, F64: 8, F16: 2, BF16: 2}


def _tensor(rows, cols, elem_bytes, keep):
    shape = (ctypes.c_int64 * 2)(rows, cols)
    keep.append(shape)
    # a fake, well-aligned address: the size functions never dereference it
    return _lib.Tensor(ctypes.c_void_p(1 << 20), 2, ctypes.cast(shape, ctypes.POINTER(ctypes.c_int64)))


def ws_bytes(n_rows, n_cols, nnz, feat, dtype=F32, op=b"copy_lhs", red=b"sum", idbits=32):
    keep = []
    csr = _lib.CSR(n_rows, n_cols, nnz, idbits, ctypes.c_void_p(1 << 21), ctypes.c_void_p(1 << 22), None)
    u = _tensor(n_cols, feat, ESIZE[dtype], keep)
    o = _tensor(n_rows, feat, ESIZE[dtype], keep)
    e = _lib.Tensor(None, 0, None)
    return int(LIB.dgla_spmm_csr_workspace_bytes(op, red, ctypes.byref(csr), dtype, ctypes.byref(u), ctypes.byref(e),
                                                 ctypes.byref(o)))


@pytest.fixture()
def tuning():
    default = int(LIB.dgla_get_tuning())
    yield default
    LIB.dgla_set_tuning(default)


def test_split_layouts_add_what_they_copy(tuning):
    n, e = 2_449_029, 61_859_140                      # the headline graph: every layout is eligible
    LIB.dgla_set_tuning(tuning & ~_lib.DGLA_TUNE_SPLIT)
    plain = ws_bytes(n, n, e, 100)
    LIB.dgla_set_tuning(tuning)
    edge = ws_bytes(n, n, e, 100)                      # edge layout: one 128-byte side line + 16-byte tail per row
    assert n * 144 <= edge - plain < n * 144 + (1 << 20)
    # rows under two whole lines (144 .. 240 bytes) are gathered in place (they touch 2 lines either way)
    LIB.dgla_set_tuning(tuning & ~_lib.DGLA_TUNE_SPLIT)
    p48 = ws_bytes(n, n, e, 48)
    LIB.dgla_set_tuning(tuning)
    assert ws_bytes(n, n, e, 48) == p48
    # retired bits are refused
    for bit in (2, 4, 32, 256, 512, 1024):
        assert LIB.dgla_set_tuning(tuning | bit) != 0 and int(LIB.dgla_get_tuning()) == tuning
    # whole lines (512-byte rows): nothing to split
    LIB.dgla_set_tuning(tuning & ~_lib.DGLA_TUNE_SPLIT)
    p128 = ws_bytes(n, n, e, 128)
    LIB.dgla_set_tuning(tuning)
    assert ws_bytes(n, n, e, 128) == p128
    # straddle layout: 200-byte bf16 rows get line-aligned 256-byte slots
    LIB.dgla_set_tuning(tuning & ~_lib.DGLA_TUNE_SPLIT)
    pb = ws_bytes(n, n, e, 100, BF16)
    LIB.dgla_set_tuning(tuning)
    sb = ws_bytes(n, n, e, 100, BF16)
    assert n * 256 <= sb - pb < n * 256 + (1 << 20)
    # a graph whose features fit the caches never splits
    assert ws_bytes(1000, 1000, 30_000, 100) < (8 << 20)


def test_small_graphs_run_shorter_units(tuning):
    """A graph with fewer than 4096 units of 512 items gets shorter units (down to 64): more units,
    hence more carry slots in the scratch — visible as a per-item scratch cost that grows as the
    graph shrinks, and settles at the 512-item figure for large graphs."""
    f = 64                                             # two lines per row: no split layout involved
    def per_item(n, e):
        return ws_bytes(n, n, e, f) / (n + e)
    big = per_item(4_000_000, 60_000_000)              # 125 k units of 512
    mid = per_item(100_000, 900_000)                   # 1 M items -> 128-item units
    small = per_item(11_000, 170_000)                  # a sampled block -> 64-item units
    assert big < mid < small
    assert 3.5 < mid / big < 4.5 and 7 < small / big < 9.5   # slots per item scale with 512 / unit size
