#!/usr/bin/env python
"""Generates tests/golden/reference_hetero_outputs.npz from the REFERENCE's own
SpMMCsrHetero<kDGLCPU> (src/array/cpu/spmm.cc:45-150, compiled in place by oracle/Makefile):
    make -C oracle ref && python tests/golden/make_golden_hetero.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402
from tests.hetero_cases import all_cases, run_case  # noqa: E402


def main():
    assert ref.available()
    ref.set_num_threads(1)
    blob = {}
    for c in all_cases(full=False):
        for k, v in run_case(ref, c).items():
            blob["%s/out/%s" % (c["name"], k)] = v
    path = os.path.join(HERE, "reference_hetero_outputs.npz")
    np.savez_compressed(path, **blob)
    print("wrote %s: %d arrays, %.1f KiB" % (path, len(blob), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
