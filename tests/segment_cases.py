"""Seeded cases of the segment-reduce family shared by the golden generator
(tests/golden/make_golden_segment.py), the oracle-vs-reference sweep and the GPU parity
tests.  Shapes follow the reference's tests/python/common/ops/test_ops.py:226-262
(test_segment_reduce: random segment lengths incl. zeros, feature sizes 1..; max/min/sum/mean)."""
import numpy as np

REDUCES = ["sum", "max", "min"]


def _seglens(kind, rng):
    if kind == "docstring":      # python/dgl/ops/segment.py:33-41
        return np.array([1, 0, 5, 4])
    if kind == "random":         # test_ops.py:236 style: lengths in [0, 10)
        return rng.integers(0, 10, 40)
    if kind == "empty_ends":     # empty segments first / last / in runs (dmlc/dgl#2610)
        return np.array([0, 0, 3, 0, 0, 0, 7, 1, 0, 0])
    if kind == "all_empty":
        return np.zeros(6, dtype=np.int64)
    if kind == "long":           # segments far longer than one 512-item merge unit
        return np.array([1500, 0, 2, 3000, 1, 700])
    if kind == "many_tiny":      # thousands of 0/1/2-length segments
        return rng.integers(0, 3, 3000)
    raise ValueError(kind)


def segment_case(name, kind, reduce, fshape, dtype=np.float32, idtype=np.int64, ties=False, seed=11):
    rng = np.random.default_rng(seed)
    seglen = _seglens(kind, rng).astype(idtype)
    n = int(seglen.sum())
    feat = rng.standard_normal((n,) + tuple(fshape))
    if ties:
        feat = np.round(feat * 2) / 2
    offsets = np.zeros(len(seglen) + 1, dtype=idtype)
    np.cumsum(seglen, out=offsets[1:])
    return {"kind": "segment_reduce", "name": name, "reduce": reduce, "offsets": offsets,
            "feat": feat.astype(dtype)}


def scatter_case(name, n, m, fshape, dtype=np.float32, idtype=np.int64, seed=12):
    rng = np.random.default_rng(seed)
    return {"kind": "scatter_add", "name": name, "m": m,
            "idx": rng.integers(0, m, n).astype(idtype),
            # multiples of 1/8 below 2^10: every partial sum is exact, so the result does not
            # depend on the order of the (atomic) additions and can be compared bit for bit
            "feat": (rng.integers(-64, 64, (n,) + tuple(fshape)) / 8.0).astype(dtype)}


def all_cases(full=False):
    cases = []
    kinds = ["docstring", "random", "empty_ends", "all_empty", "long", "many_tiny"]
    shapes = [("f1", ()), ("f7", (7,)), ("f100", (100,)), ("f4x8", (4, 8)), ("f130", (130,))]
    for red in REDUCES:
        for kind in kinds:
            for sname, shp in (shapes if full else [shapes[1]]):
                for idt in ((np.int32, np.int64) if full else (np.int64,)):
                    for dt in ((np.float32, np.float64) if full else (np.float32,)):
                        cases.append(segment_case(
                            "segred-%s-%s-%s-%s-%s" % (red, kind, sname, np.dtype(idt).name, np.dtype(dt).name),
                            kind, red, shp, dt, idt))
    for red in ("max", "min"):
        cases.append(segment_case("segred-%s-ties" % red, "random", red, (5,), ties=True))
    if not full:
        cases.append(segment_case("segred-sum-long-f20-int32", "long", "sum", (20,), np.float32, np.int32))
        cases.append(segment_case("segred-max-long-f2x4-f64", "long", "max", (2, 4), np.float64, np.int64))
        cases.append(segment_case("segred-min-many_tiny-f1-int32", "many_tiny", "min", (), np.float32, np.int32))
    for sname, shp in [("f1", ()), ("f8", (8,)), ("f100", (100,)), ("f3x5", (3, 5))]:
        for idt in ((np.int32, np.int64) if full else (np.int64,)):
            for dt in ((np.float32, np.float64) if full else (np.float32,)):
                cases.append(scatter_case("scatter-%s-%s-%s" % (sname, np.dtype(idt).name, np.dtype(dt).name),
                                          500, 37, shp, dt, idt))
    names = [c["name"] for c in cases]
    assert len(set(names)) == len(names)
    return cases


def run_case(backend, c):
    """Evaluate with `oracle` or `oracle.ref`.  Segment cases also run the backward of
    max/min (BackwardSegmentCmp) on the forward's own arg."""
    if c["kind"] == "segment_reduce":
        out, arg = backend.segment_reduce(c["reduce"], c["feat"], c["offsets"])
        res = {"out": out, "arg": arg}
        if arg is not None:
            dy = (np.arange(out.size, dtype=np.float64).reshape(out.shape) / 7 + 1).astype(out.dtype)
            back = np.zeros(c["feat"].shape, dtype=out.dtype)
            if back.size:
                backend.backward_segment_cmp(dy, arg, back)
            res["back"] = back
        return res
    if c["kind"] == "scatter_add":
        out = np.zeros((c["m"],) + c["feat"].shape[1:], dtype=c["feat"].dtype)
        backend.scatter_add(c["feat"], c["idx"], out)
        return {"out": out}
    raise ValueError(c["kind"])
