"""Maximum-size edge case: a CSR with MORE THAN 2^31 edges (int64 ids), the regime of
ogbn-papers100M-scale graphs that 288 GB of HBM makes a single-GPU problem.  Every 32-bit
offset in the merge plan, the LDS staging, the carry workspace or the fix-up would show up
here.  Checked exactly: small-integer features make every fp32 partial sum exact, so the result
is order-independent and compared bit for bit with an independent torch evaluation."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_spmm_with_more_than_2_31_edges(dev):
    from dgl_amd import _capi

    free, _ = torch.cuda.mem_get_info(dev)
    if free < 80 << 30:
        pytest.skip("needs ~60 GB of free HBM")
    n = 1 << 20
    deg = 2049                       # n * deg = 2^31 + 2^20 edges
    e = n * deg
    assert e > 2 ** 31
    gen = torch.Generator(device=dev).manual_seed(1)
    indices = torch.randint(0, n, (e,), device=dev, dtype=torch.int64, generator=gen)
    indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * deg
    # make the rows uneven around the 2^31 boundary: move the row ends (keeps indptr monotone)
    shift = torch.randint(-deg // 2, deg // 2, (n - 1,), device=dev, generator=gen)
    indptr[1:-1] += shift
    x = torch.randint(0, 4, (n, 1), device=dev, generator=gen).float()   # sums < 2^24: exact in fp32
    csr = _capi.make_csr(indptr, indices, None, n)
    for red in ("sum", "max"):
        out = torch.empty(n, 1, device=dev)
        arg = torch.empty(n, 1, dtype=torch.int64, device=dev) if red == "max" else None
        ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", red, csr, x.dtype, x, None, out),
                         dtype=torch.uint8, device=dev)
        _capi.spmm_csr("copy_lhs", red, csr, x, None, out, arg, None, ws)
        torch.cuda.synchronize()
        del ws
        # independent evaluation in slabs of rows (keeps the temporary small)
        step = 1 << 16
        for r0 in range(0, n, step):
            r1 = min(n, r0 + step)
            lo, hi = int(indptr[r0]), int(indptr[r1])
            vals = x[indices[lo:hi], 0]
            seg = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), (indptr[r0 + 1:r1 + 1] - indptr[r0:r1]))
            if red == "sum":
                want = torch.zeros(r1 - r0, device=dev).index_add_(0, seg, vals)
                assert torch.equal(out[r0:r1, 0], want), (red, r0)
            else:
                want = torch.full((r1 - r0,), -float("inf"), device=dev).scatter_reduce(0, seg, vals, "amax")
                assert torch.equal(out[r0:r1, 0], want), (red, r0)
                # arg: the FIRST position attaining the maximum, as a global column id
                pos = torch.arange(lo, hi, device=dev)
                is_max = vals == want[seg]
                first = torch.full((r1 - r0,), hi, device=dev, dtype=torch.int64).scatter_reduce(
                    0, seg[is_max], pos[is_max], "amin")
                assert torch.equal(arg[r0:r1, 0], indices[first]), (red, r0)
            del vals, seg, want
