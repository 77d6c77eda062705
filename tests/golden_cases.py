"""Seeded small cases shared by the golden-fixture generator (tests/golden/make_golden.py),
the oracle-vs-reference sweep and the GPU-vs-golden parity tests.

A case is a dict of plain numpy inputs plus the operator description; `run_case(backend, c)`
evaluates it with ``oracle`` (our C restatement) or ``oracle.ref`` (the reference's own CPU
kernels built from /root/reference).  Graph shapes follow the reference's operator tests
(tests/python/common/ops/test_ops.py:87-181): rand_graph(30, 100) and
rand_bipartite(30, 40, 300), plus the degenerate graphs its kernels must survive.
"""
import numpy as np

from tests.graphgen import coo_to_csc, coo_to_csr

SPMM_OPS = ["add", "sub", "mul", "div", "copy_lhs", "copy_rhs"]
SDDMM_OPS = SPMM_OPS + ["dot"]
REDUCES = ["sum", "max", "min"]


def _graph(kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "homo":          # dgl.rand_graph(30, 100)
        return 30, 30, rng.integers(0, 30, 100), rng.integers(0, 30, 100)
    if kind == "bipartite":     # dgl.rand_bipartite(30, 40, 300)
        return 30, 40, rng.integers(0, 30, 300), rng.integers(0, 40, 300)
    if kind == "star":          # test_ops.py:193-216 shape: many leaves into node 0, rest isolated
        n = 700
        return n, n, np.arange(1, n), np.zeros(n - 1, dtype=np.int64)
    if kind == "multi":         # multigraph with duplicate edges and isolated nodes
        src = np.array([0, 0, 0, 1, 1, 2, 5, 5, 5, 5])
        dst = np.array([1, 1, 1, 2, 2, 0, 1, 1, 7, 7])
        return 9, 9, src, dst
    if kind == "ties":          # equal candidates: first CSR position must win max/min
        src = np.array([3, 1, 2, 0, 3, 1])
        dst = np.array([0, 0, 0, 0, 1, 1])
        return 4, 3, src, dst
    raise ValueError(kind)


def _feat(rng, n, shape, dtype, ties=False):
    x = rng.random((n,) + tuple(shape)) + 1
    if ties:
        x = np.round(x * 2) / 2  # few distinct values -> many exact ties
    return x.astype(dtype)


def spmm_case(name, kind, op, reduce, ushape, eshape, dtype=np.float32, idtype=np.int32,
              fmt="csr", use_eids=True, seed=1):
    n_src, n_dst, src, dst = _graph(kind, seed)
    rng = np.random.default_rng(seed + 1000)
    c = {"kind": "spmm_" + fmt, "name": name, "op": op, "reduce": reduce, "n_src": n_src,
         "n_dst": n_dst}
    c["ufeat"] = _feat(rng, n_src, ushape, dtype, kind == "ties") if op != "copy_rhs" else None
    c["efeat"] = _feat(rng, len(src), eshape, dtype, kind == "ties") if op != "copy_lhs" else None
    if fmt == "csr":
        c["indptr"], c["indices"], eids = coo_to_csc(src, dst, n_dst, idtype)
        c["eids"] = eids if use_eids else None
    else:
        c["row"], c["col"] = src.astype(idtype), dst.astype(idtype)
        c["eids"] = rng.permutation(len(src)).astype(idtype) if use_eids else None
    return c


def sddmm_case(name, kind, op, lshape, rshape, lt, rt, dtype=np.float32, idtype=np.int32,
               fmt="coo", seed=2):
    n_src, n_dst, src, dst = _graph(kind, seed)
    rng = np.random.default_rng(seed + 2000)
    count = {"u": n_src, "e": len(src), "v": n_dst}
    c = {"kind": "sddmm_" + fmt, "name": name, "op": op, "lhs_target": lt, "rhs_target": rt,
         "n_src": n_src, "n_dst": n_dst}
    c["lhs"] = _feat(rng, count[lt], lshape, dtype) if op != "copy_rhs" else None
    c["rhs"] = _feat(rng, count[rt], rshape, dtype) if op != "copy_lhs" else None
    if fmt == "coo":
        c["row"], c["col"] = src.astype(idtype), dst.astype(idtype)
        c["eids"] = None
    else:  # out-edge CSR: rows = source nodes, with the edge-id map
        c["indptr"], c["indices"], c["eids"] = coo_to_csr(src, dst, n_src, idtype)
    return c


def softmax_case(name, kind, shape, dtype=np.float32, idtype=np.int32, seed=3):
    n_src, n_dst, src, dst = _graph(kind, seed)
    rng = np.random.default_rng(seed + 3000)
    indptr, indices, eids = coo_to_csc(src, dst, n_dst, idtype)
    score = (rng.standard_normal((len(src),) + tuple(shape)) * 3).astype(dtype)
    grad = rng.standard_normal((len(src),) + tuple(shape)).astype(dtype)
    return {"kind": "edge_softmax", "name": name, "indptr": indptr, "indices": indices,
            "eids": eids, "score": score, "grad": grad}


def all_cases(full=False):
    """`full=False`: the subset committed as golden fixtures.  `full=True`: the exhaustive
    sweep run in memory by tests/test_oracle_vs_reference.py."""
    cases = []
    # ---- SpMM on CSR -------------------------------------------------------------------
    shapes = [("plain", (7,), (7,)), ("gat", (4, 8), (4, 1)), ("bcast", (3, 1), (1, 4)),
              ("scalar", (), ())]
    kinds = ["homo", "bipartite", "star", "multi", "ties"]
    for op in SPMM_OPS:
        for red in REDUCES:
            for sname, us, es in (shapes if full else shapes[:1]):
                for kind in (kinds if full else ["bipartite"]):
                    for idt in ((np.int32, np.int64) if full else (np.int32,)):
                        for dt in ((np.float32, np.float64) if full else (np.float32,)):
                            cases.append(spmm_case(
                                "spmm_csr-%s-%s-%s-%s-%s-%s" % (op, red, sname, kind, np.dtype(idt).name, np.dtype(dt).name),
                                kind, op, red, us, es, dt, idt))
    if not full:
        for op, red, sname, us, es, kind, idt, dt in [
                ("mul", "sum", "gat", (4, 8), (4, 1), "homo", np.int32, np.float32),
                ("mul", "max", "gat", (4, 8), (4, 1), "homo", np.int64, np.float32),
                ("add", "min", "bcast", (3, 1), (1, 4), "bipartite", np.int32, np.float64),
                ("div", "sum", "bcast", (3, 1), (1, 4), "multi", np.int64, np.float64),
                ("copy_lhs", "sum", "scalar", (), (), "star", np.int32, np.float32),
                ("copy_lhs", "max", "plain", (7,), (7,), "ties", np.int32, np.float32),
                ("copy_rhs", "min", "plain", (7,), (7,), "ties", np.int64, np.float32),
                ("mul", "max", "plain", (7,), (7,), "ties", np.int32, np.float32),
                ("copy_lhs", "min", "plain", (5,), (5,), "multi", np.int32, np.float32),
                ("copy_rhs", "sum", "plain", (5,), (5,), "star", np.int64, np.float64)]:
            cases.append(spmm_case(
                "spmm_csr-%s-%s-%s-%s-%s-%s" % (op, red, sname, kind, np.dtype(idt).name, np.dtype(dt).name),
                kind, op, red, us, es, dt, idt))
    # positions as edge ids (csr.data == null, spmm.h:55 `has_idx`)
    cases.append(spmm_case("spmm_csr-mul-sum-noeid", "homo", "mul", "sum", (6,), (6,), use_eids=False))
    cases.append(spmm_case("spmm_csr-copy_rhs-max-noeid", "homo", "copy_rhs", "max", (6,), (6,), use_eids=False))
    # ---- SpMM on COO -------------------------------------------------------------------
    for op, red in ([(o, r) for o in SPMM_OPS for r in REDUCES] if full else
                    [("copy_lhs", "sum"), ("mul", "sum"), ("add", "max"), ("copy_rhs", "min"),
                     ("mul", "max"), ("copy_lhs", "min")]):
        for idt in ((np.int32, np.int64) if full else (np.int32,)):
            cases.append(spmm_case("spmm_coo-%s-%s-%s" % (op, red, np.dtype(idt).name), "bipartite",
                                   op, red, (5,), (5,), np.float32, idt, fmt="coo"))
    # ---- SDDMM -------------------------------------------------------------------------
    tpairs = [("u", "v"), ("u", "e"), ("e", "v"), ("v", "u"), ("e", "e"), ("v", "v"), ("u", "u"),
              ("e", "u"), ("v", "e")]
    for op in SDDMM_OPS:
        for lt, rt in (tpairs if full else tpairs[:3]):
            for fmt in ("coo", "csr"):
                for sname, ls, rs in ([("plain", (8,), (8,)), ("heads", (4, 8), (4, 8)),
                                       ("bcast", (3, 1, 4), (1, 5, 4))] if full else
                                      [("heads", (4, 8), (4, 8))]):
                    if not full and fmt == "csr" and (lt, rt) != ("u", "v"):
                        continue
                    for idt in ((np.int32, np.int64) if full else (np.int32,)):
                        cases.append(sddmm_case(
                            "sddmm_%s-%s-%s%s-%s-%s" % (fmt, op, lt, rt, sname, np.dtype(idt).name),
                            "bipartite", op, ls, rs, lt, rt, np.float32, idt, fmt))
    if not full:
        cases.append(sddmm_case("sddmm_coo-dot-uv-bcast-int64-f64", "homo", "dot", (3, 1, 4), (1, 5, 4),
                                "u", "v", np.float64, np.int64, "coo"))
        cases.append(sddmm_case("sddmm_coo-add-uv-bcast", "homo", "add", (3, 1), (1, 4), "u", "v"))
        cases.append(sddmm_case("sddmm_coo-dot-uv-d32", "multi", "dot", (2, 32), (2, 32), "u", "v"))
    # ---- edge softmax ------------------------------------------------------------------
    for kind in ["homo", "bipartite", "star", "multi"]:
        for dt in ((np.float32, np.float64) if (full or kind == "homo") else (np.float32,)):
            cases.append(softmax_case("edge_softmax-%s-%s" % (kind, np.dtype(dt).name), kind, (4, 1), dt))
    names = [c["name"] for c in cases]
    assert len(set(names)) == len(names), "duplicate case names"
    return cases


def run_case(backend, c):
    """Evaluate a case with `oracle` or `oracle.ref`; returns a dict of output arrays."""
    k = c["kind"]
    if k == "spmm_csr":
        out, au, ae = backend.spmm_csr(c["op"], c["reduce"], c["indptr"], c["indices"], c["eids"],
                                       c["ufeat"], c["efeat"])
        return {"out": out, "arg_u": au, "arg_e": ae}
    if k == "spmm_coo":
        out, au, ae = backend.spmm_coo(c["op"], c["reduce"], c["row"], c["col"], c["eids"],
                                       c["n_dst"], c["ufeat"], c["efeat"])
        return {"out": out, "arg_u": au, "arg_e": ae}
    if k == "sddmm_coo":
        return {"out": backend.sddmm_coo(c["op"], c["row"], c["col"], c["eids"], c["lhs"], c["rhs"],
                                         c["lhs_target"], c["rhs_target"])}
    if k == "sddmm_csr":
        return {"out": backend.sddmm_csr(c["op"], c["indptr"], c["indices"], c["eids"], c["lhs"],
                                         c["rhs"], c["lhs_target"], c["rhs_target"])}
    if k == "edge_softmax":
        out = backend.edge_softmax_fwd(c["indptr"], c["eids"], c["score"])
        sds = (out * c["grad"]).astype(out.dtype)  # sparse.py:736 `sds = out * grad_out`
        return {"out": out, "back": backend.edge_softmax_bwd(c["indptr"], c["eids"], out, sds)}
    raise ValueError(k)
