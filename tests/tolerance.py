"""The fp32-sum acceptance rule of the parity tests.

north_star: reductions "within 1e-5 rel fp32" of the reference.  The reference's CPU kernel adds
a row's messages SEQUENTIALLY in fp32 (src/array/cpu/spmm.h:60-70), which on long rows is itself
off the exact sum by more than 1e-5 (measured: 2.3e-5 on a 2 362-edge row of the test graph,
tools/diag_sharded.py; the blocked partial sums of the HIP kernels stay within 3e-6 there).  So an
element passes when it is within 1e-5 of the reference's fp32 value OR at least as close to the
exact (fp64) sum as the reference's own value is — and it must ALWAYS be within 1e-5 of the exact
sum, flat.  The escape only exists for long rows: where the caller gives the rows' lengths, every
element of a row with fewer than SHORT_ROW edges must meet the PLAIN bar, 1e-5 of the reference's
value (VERDICT r3 Next #1d).  SHORT_ROW = 500, not the 1000 the verdict suggested: measured on
test_gpu_sharded's graph (40 k rows, U(0,1)+1 features, round 4), the REFERENCE's own sequential fp32
sum is off the exact sum by up to 1.4e-6 on rows under 100 edges, 3.7e-6 (100-250), 5.7e-6 (250-500),
7.9e-6 (500-750), 1.14e-5 (750-1000), 1.3e-5 (1000-2000): a result that is exact to 4e-7 misses "1e-5 of
the reference" on 19 of 4 M elements, all on rows of 750-1000 edges."""
import numpy as np


def max_rel_err(out, ref, floor=1e-30):
    """Plain max |out - ref| / max(|ref|, floor) — north_star's own wording of the bar, reported
    next to every verdict of the rule below so that the re-definition never hides a number."""
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if out.size == 0:
        return 0.0
    return float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), floor)))


LAST = {}  # plain figures of the most recent check (read by benchmarks / printed by failures)


SHORT_ROW = 500


def assert_fp32_sum(out, ref, exact, rtol=1e-5, atol=1e-6, row_len=None):
    """row_len: number of edges reduced into each row of `out` (shape (out.shape[0],)), or None."""
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    exact = np.asarray(exact, dtype=np.float64)
    LAST.update(max_rel_err_vs_reference=max_rel_err(out, ref), max_rel_err_vs_exact=max_rel_err(out, exact),
                reference_max_rel_err_vs_exact=max_rel_err(ref, exact))
    plain = ("plain max rel err: vs reference %.3g, vs exact fp64 sum %.3g (the reference itself is %.3g off "
             "the exact sum)" % (LAST["max_rel_err_vs_reference"], LAST["max_rel_err_vs_exact"],
                                 LAST["reference_max_rel_err_vs_exact"]))
    np.testing.assert_allclose(out, exact, rtol=rtol, atol=atol, err_msg=plain)
    near_ref = np.abs(out - ref) <= rtol * np.abs(ref) + atol
    closer = np.abs(out - exact) <= np.abs(ref - exact)
    if row_len is not None:
        short = (np.asarray(row_len).reshape((-1,) + (1,) * (out.ndim - 1)) < SHORT_ROW)
        short = np.broadcast_to(short, out.shape)
        plain_bad = short & ~near_ref
        assert not plain_bad.any(), "%d elements of rows with < %d edges are not within %g of the reference; %s" % (
            int(plain_bad.sum()), SHORT_ROW, rtol, plain)
        LAST["short_row_elements_under_plain_bar"] = int(short.sum())
    bad = ~(near_ref | closer)
    assert not bad.any(), "%d elements neither within %g of the reference nor closer to the exact sum than it; %s" % (
        int(bad.sum()), rtol, plain)
