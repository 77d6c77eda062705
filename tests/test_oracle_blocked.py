"""The libxsmm-style blocked CPU baseline (oracle.copy_u_sum_csr_blocked ≙ SpMMRedopCsrOpt,
src/array/cpu/spmm_blocking_libxsmm.h:432-557 without the JIT row kernel) against the naive
kernel it is an organisation of: same bits for every tiling, including many K blocks, M blocks of
one row, empty rows and a hub row."""
import numpy as np
import pytest
import torch

import oracle
from tests.graphgen import synth_csr


@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
@pytest.mark.parametrize("fdt", [np.float32, np.float64])
@pytest.mark.parametrize("llc", [None, 1 << 16, 1 << 10, 64])
def test_blocked_equals_naive_bit_for_bit(idt, fdt, llc):
    n, e, f = 3000, 70_000, 20
    g = synth_csr(n, n, e, "L", seed=3, idtype=idt)
    x = (np.random.default_rng(1).random((n, f)) + 1).astype(fdt)
    ip, ix = g["indptr"].numpy(), g["indices"].numpy()
    naive = oracle.copy_u_sum_csr(ip, ix, x, 4)
    for threads in (1, 3, 8):
        out, info = oracle.copy_u_sum_csr_blocked(ip, ix, x, threads, llc=llc)
        assert np.array_equal(out, naive), (threads, info)
        assert info["num_M_blocks"] * info["M_block"] >= n
        if llc == 64:
            assert info["num_K_blocks"] > 100       # the re-tiling path really ran
    full, _, _ = oracle.spmm_csr("copy_lhs", "sum", ip, ix, None, x, None)
    assert np.array_equal(naive, full)


def test_blocked_tile_sizes_follow_the_reference_formula():
    # spmm_blocking_libxsmm.h:464-475
    n, e, f, threads, llc = 5000, 120_000, 100, 4, 1 << 20
    g = synth_csr(n, n, e, "U", seed=2, idtype=torch.int32)
    x = np.ones((n, f), dtype=np.float32)
    _, info = oracle.copy_u_sum_csr_blocked(g["indptr"].numpy(), g["indices"].numpy(), x, threads, llc=llc)
    nnz_prob = (e / n) / n
    assert info["K_block"] == min(n, int(llc / (f * 4 * nnz_prob * 500)))
    assert info["M_block"] == n // (threads * 20)
    assert info["num_K_blocks"] == -(-n // info["K_block"]) and info["num_M_blocks"] == -(-n // info["M_block"])
