"""Seeded COO -> CSR cases shared by tests/golden/make_golden_coo.py and tests/test_coo2csr.py."""
import numpy as np

SHAPES = [(1, 1, 0), (5, 7, 0), (1, 1, 1), (30, 40, 300), (1000, 10, 20000), (3, 50000, 60000),
          (70000, 70000, 1), (1 << 14, 333, 1 << 16), (100, 100, 5000)]


def coo_case(n, m, e, idtype, with_eids):
    rng = np.random.default_rng(e + n)
    row = rng.integers(0, n, e).astype(idtype)
    if e > 10 and n > 8:
        row[rng.integers(0, e, e // 3)] = n - 1      # a hub row and empty rows in the same graph
        row[row == 2] = 3
    col = rng.integers(0, m, e).astype(idtype)
    eids = rng.permutation(e).astype(idtype) if with_eids else None
    name = "coo-%d-%d-%d-%s-%s" % (n, m, e, np.dtype(idtype).name, "eid" if with_eids else "pos")
    return {"name": name, "num_rows": n, "num_cols": m, "row": row, "col": col, "eids": eids}


def all_cases():
    return [coo_case(n, m, e, idt, we) for (n, m, e) in SHAPES for idt in (np.int32, np.int64)
            for we in (False, True)]


def stable_definition(row, col, eids, n):
    """Rows compressed, COO order kept inside a row, data = original edge id."""
    order = np.argsort(row, kind="stable")
    indptr = np.zeros(n + 1, dtype=row.dtype)
    np.add.at(indptr, row + 1, 1)
    return (np.cumsum(indptr).astype(row.dtype), col[order],
            (order if eids is None else eids[order]).astype(row.dtype))
