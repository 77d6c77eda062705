"""Seeded heterograph g-SpMM cases shared by the golden generator, the oracle-vs-reference test
and the GPU test.  Metagraph of tests/python/common/test_heterograph-kernel.py's user / game /
developer example shape: several relations reduce into 'user', one into 'game', one node type
receives nothing; ties and empty rows on purpose."""
import numpy as np

from tests.graphgen import coo_to_csc

NUM_NODES = [40, 25, 9]                        # user, game, developer
META = [(0, 0), (1, 0), (2, 0), (0, 1), (2, 1)]   # follows, played-by, sponsors, plays, develops
NUM_EDGES = [300, 180, 20, 150, 0]             # the last relation has no edges


def hetero_case(op, reduce, fshape, eshape, dtype=np.float32, idtype=np.int32, ties=False, seed=7):
    rng = np.random.default_rng(seed)
    rels = []
    for (s, d), ne in zip(META, NUM_EDGES):
        src = rng.integers(0, NUM_NODES[s], ne)
        dst = rng.integers(0, max(NUM_NODES[d] - 3, 1), ne)     # the last 3 nodes of every type get nothing
        indptr, indices, eids = coo_to_csc(src, dst, NUM_NODES[d], idtype)
        rels.append({"indptr": indptr, "indices": indices, "eids": eids, "src": s, "dst": d})
    q = (lambda a: np.round(a * 2) / 2) if ties else (lambda a: a)
    ufeats = [q(rng.random((n,) + fshape) + 1).astype(dtype) for n in NUM_NODES]
    efeats = [q(rng.random((ne,) + eshape) + 1).astype(dtype) for ne in NUM_EDGES]
    name = "hetero-%s-%s-%s-%s-%s%s" % (op, reduce, "x".join(map(str, fshape)) or "s", np.dtype(idtype).name,
                                        np.dtype(dtype).name, "-ties" if ties else "")
    return {"name": name, "op": op, "reduce": reduce, "rels": rels, "ufeats": ufeats, "efeats": efeats}


def all_cases(full=False):
    cases = []
    for op in ("copy_lhs", "copy_rhs", "mul", "add"):
        for red in ("sum", "max", "min"):
            for idt in ((np.int32, np.int64) if full else (np.int32,)):
                for dt in ((np.float32, np.float64) if full else (np.float32,)):
                    cases.append(hetero_case(op, red, (6,), (6,), dt, idt))
                    if red != "sum":
                        cases.append(hetero_case(op, red, (3,), (3,), dt, idt, ties=True))
    cases.append(hetero_case("mul", "max", (4, 8), (4, 1), np.float32, np.int64))
    cases.append(hetero_case("copy_lhs", "sum", (5,), (5,), np.float64, np.int64))
    names = [c["name"] for c in cases]
    assert len(set(names)) == len(names)
    return cases


def run_case(backend, c):
    outs, au, ae, aut, aet = backend.spmm_csr_hetero(c["op"], c["reduce"], c["rels"], NUM_NODES, c["ufeats"], c["efeats"])
    res = {}
    for key, lst in (("out", outs), ("arg_u", au), ("arg_e", ae), ("arg_u_ntype", aut), ("arg_e_etype", aet)):
        for nt, a in enumerate(lst):
            if a is not None:
                res["%s/%d" % (key, nt)] = a
    return res
