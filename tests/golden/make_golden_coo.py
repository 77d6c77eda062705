#!/usr/bin/env python
"""Generates tests/golden/reference_coo2csr_outputs.npz from the REFERENCE's own
aten::impl::COOToCSR<kDGLCPU> (src/array/cpu/spmat_op_impl_coo.cc, compiled in place by
oracle/Makefile):  make -C oracle ref && python tests/golden/make_golden_coo.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402
from tests.coo_cases import all_cases  # noqa: E402


def main():
    assert ref.available()
    blob = {}
    for c in all_cases():
        if c["row"].size > 5000:
            continue  # keep the fixture small; the big shapes are checked against the definition
        ip, ix, ei = ref.coo_to_csr(c["row"], c["col"], c["eids"], c["num_rows"], c["num_cols"])
        for k, v in (("indptr", ip), ("indices", ix), ("eids", ei)):
            blob["%s/out/%s" % (c["name"], k)] = v
    path = os.path.join(HERE, "reference_coo2csr_outputs.npz")
    np.savez_compressed(path, **blob)
    print("wrote %s: %d arrays, %.1f KiB" % (path, len(blob), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
