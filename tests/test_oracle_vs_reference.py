"""Pins the oracle (our plain-C restatement) to the REFERENCE ITSELF: oracle/_ref/libdglref.so
is dgl's own src/array/cpu/{spmm,sddmm}.cc + src/bcast.cc compiled where they lie
(oracle/Makefile).  Every case of the exhaustive sweep must agree BIT FOR BIT — values,
arg_u / arg_e, broadcast offset tables.  Skipped only when the reference build is absent.
"""
import os

import numpy as np
import pytest

import oracle
from oracle import ref
from tests.golden_cases import all_cases, run_case

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libdglref.so not built")


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)


def test_full_sweep_bit_exact():
    bad = []
    cases = all_cases(full=True)
    assert len(cases) > 2000
    for c in cases:
        serial = c["kind"] == "spmm_coo" and c["reduce"] == "sum"
        if serial:
            # SpMMSumCoo adds with `omp atomic` (spmm.h:196-205): its order is only defined
            # on one thread, which is the order the oracle restates
            ref.set_num_threads(1)
        got, want = run_case(oracle, c), run_case(ref, c)
        if serial:
            ref.set_num_threads(os.cpu_count() or 1)
        for k in want:
            if not _same(got[k], want[k]):
                bad.append((c["name"], k))
    assert not bad, bad[:20]


@pytest.mark.parametrize("op,ls,rs", [
    ("mul", (5, 4, 8), (9, 4, 1)), ("add", (5, 3, 1), (9, 1, 4)), ("dot", (5, 3, 1, 4), (9, 1, 5, 4)),
    ("dot", (5, 2, 16), (5, 2, 16)), ("copy_lhs", (5, 7), (9, 3)), ("copy_rhs", (5, 7), (9, 3)),
    ("sub", (1, 2, 1, 3, 1), (1, 4, 1, 3, 1, 1)), ("div", (2, 5, 3, 1, 7), (2, 1, 3, 7, 1)),
    ("mul", (4, 1), (4, 1)), ("add", (4, 3), (4, 1)),
])
def test_bcast_tables_match_reference(op, ls, rs):
    """oracle.BcastOff restates CalcBcastOff (src/bcast.cc:36-90)."""
    want = ref.calc_bcast_off(op, ls, rs)
    got = oracle.BcastOff(op, ls, rs)
    assert got.use_bcast == want["use_bcast"]
    for k in ("lhs_len", "rhs_len", "out_len", "reduce_size"):
        assert getattr(got, k) == want[k], k
    if want["use_bcast"]:
        np.testing.assert_array_equal(got.lhs_offset, want["lhs_offset"])
        np.testing.assert_array_equal(got.rhs_offset, want["rhs_offset"])


def test_reference_errors_surface():
    # SWITCH_OP default branch: LOG(FATAL) "Unsupported SpMM binary operator" (spmm_binary_ops.h)
    indptr = np.array([0, 1], np.int32)
    idx = np.array([0], np.int32)
    x = np.ones((1, 2), np.float32)
    with pytest.raises(RuntimeError, match="[Uu]nsupported"):
        ref.spmm_csr("pow", "sum", indptr, idx, None, x, x)


def test_c2_shaped_sum_bit_exact():
    """A 1/64-scale ogbn-products-shaped CSR (heavy-tailed rows) through both, copy_u+sum."""
    from tests.graphgen import C2_EDGES, C2_NODES, synth_csr

    n, e = C2_NODES // 64, C2_EDGES // 64
    g = synth_csr(n, n, e, "U")
    rng = np.random.default_rng(5)
    x = (rng.random((n, 100)) + 1).astype(np.float32)
    ip, ix = g["indptr"].numpy(), g["indices"].numpy()
    want = ref.spmm_csr("copy_lhs", "sum", ip, ix, None, x, None)[0]
    np.testing.assert_array_equal(oracle.spmm_csr("copy_lhs", "sum", ip, ix, None, x, None)[0], want)
    np.testing.assert_array_equal(oracle.copy_u_sum_csr(ip, ix, x), want)
