#!/usr/bin/env python
"""Generates tests/golden/reference_segment_outputs.npz from the REFERENCE's own CPU kernels
(dgl::aten::{SegmentReduce,ScatterAdd,BackwardSegmentCmp}<kDGLCPU>, compiled from
/root/reference/src/array/cpu/segment_reduce.cc by oracle/Makefile).

    make -C oracle ref && python tests/golden/make_golden_segment.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402
from tests.segment_cases import all_cases, run_case  # noqa: E402


def main():
    assert ref.available(), "build oracle/_ref first: make -C oracle ref"
    ref.set_num_threads(1)
    blob, names = {}, []
    for c in all_cases(full=False):
        out = run_case(ref, c)
        names.append(c["name"])
        for k, v in c.items():
            if isinstance(v, np.ndarray):
                blob["%s/in/%s" % (c["name"], k)] = v
        for k, v in out.items():
            if v is not None:
                blob["%s/out/%s" % (c["name"], k)] = v
    path = os.path.join(HERE, "reference_segment_outputs.npz")
    np.savez_compressed(path, **blob)
    print("wrote %s: %d cases, %.1f KiB" % (path, len(names), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
