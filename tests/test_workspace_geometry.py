"""Host-side geometry of dgla_spmm_csr_workspace_bytes (no GPU: the function is arithmetic over
shapes): what each layout adds to the scratch of a call, and the unit size of small graphs.
Mirrors the decisions of csrc/spmm_csr.cuh::spmm_geometry / spmm_split_shape_ok / spmm_tail_slices."""
import ctypes
import os

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
    LIB.dgla_set_tuning(tuning | _lib.DGLA_TUNE_SPLIT_CLASSIC)
    classic = ws_bytes(n, n, e, 100)                   # classic: whole rows (384 + 16 bytes)
    assert n * 400 <= classic - plain < n * 400 + (1 << 20)
    LIB.dgla_set_tuning(tuning)
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


def test_tail_pass_structure_is_part_of_every_shape_on_an_eligible_graph(tuning):
    n, e = 2_449_029, 61_859_140
    old = {k: os.environ.pop(k, None) for k in ("DGLA_TAIL_MIN_COLS", "DGLA_TAIL_SLICE_KB")}
    try:
        LIB.dgla_set_tuning(tuning | _lib.DGLA_TUNE_TAIL_PASS)
        with_f100 = ws_bytes(n, n, e, 100)
        with_f128 = ws_bytes(n, n, e, 128)
        LIB.dgla_set_tuning(tuning)
        base_f100 = ws_bytes(n, n, e, 100)
        base_f128 = ws_bytes(n, n, e, 128)
        slices = -(-n * 16 // (2560 << 10))            # 16 slices of ~2.5 MB of tails
        structure = 4 * (slices * n + 1) + 4 * e        # virtual row pointers + column ids (+ a small plan)
        # long-lived structure in EVERY shape's workspace (it must sit at one place for all calls) ...
        assert with_f128 - base_f128 >= structure
        # ... and the partial sums only where the pass runs
        assert with_f100 - base_f100 >= structure + 16 * slices * n
        # int64 ids and small graphs never carry it
        LIB.dgla_set_tuning(tuning | _lib.DGLA_TUNE_TAIL_PASS)
        assert ws_bytes(n, n, e, 100, idbits=64) == _ws64(n, e, tuning)
        assert ws_bytes(200_000, 200_000, 2_000_000, 100) < base_f100
    finally:
        LIB.dgla_set_tuning(tuning)
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


def _ws64(n, e, tuning):
    LIB.dgla_set_tuning(tuning)
    v = ws_bytes(n, n, e, 100, idbits=64)
    LIB.dgla_set_tuning(tuning | _lib.DGLA_TUNE_TAIL_PASS)
    return v


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
