"""The oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md §5:
the reference runs its C++ tests under sanitizers in CI; the checker of this repository gets the
same treatment).  oracle/Makefile builds _build/liboracle_asan.so from the same sources; a child
interpreter preloads the sanitizer runtime, points the oracle package at that build and drives
every kernel family over degenerate and ordinary inputs.  Any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import numpy as np, sys
sys.path.insert(0, %r)
import oracle
rng = np.random.default_rng(0)
def graph(n_dst, n_src, e, idt):
    dst = np.sort(rng.integers(0, n_dst, e)) if e else np.zeros(0, np.int64)
    indptr = np.zeros(n_dst + 1, dtype=idt); np.add.at(indptr, dst + 1, 1); indptr = np.cumsum(indptr).astype(idt)
    src = rng.integers(0, n_src, e).astype(idt)
    for r in range(n_dst):
        src[indptr[r]:indptr[r + 1]].sort()
    return indptr, src, rng.permutation(e).astype(idt)
for idt in (np.int32, np.int64):
    for fdt in (np.float32, np.float64):
        for (nd, ns, e) in ((7, 5, 0), (1, 1, 1), (40, 30, 500), (3, 200, 900)):
            ip, ix, ei = graph(nd, ns, e, idt)
            x = rng.random((ns, 6)).astype(fdt); w = rng.random((e, 6)).astype(fdt); w1 = rng.random((e, 1)).astype(fdt)
            for op in ("add", "mul", "copy_lhs", "copy_rhs", "div", "sub"):
                for red in ("sum", "max", "min"):
                    oracle.spmm_csr(op, red, ip, ix, ei, x, w)
            oracle.spmm_csr("mul", "sum", ip, ix, None, x.reshape(ns, 3, 2), w1.reshape(e, 1, 1) * np.ones((1, 3, 1), fdt))
            oracle.copy_u_sum_csr(ip, ix, x, 3)
            for llc in (None, 64):
                oracle.copy_u_sum_csr_blocked(ip, ix, x, 3, llc=llc, num_cols=ns)
            row = np.repeat(np.arange(nd), np.diff(ip)).astype(idt)
            oracle.spmm_coo("mul", "sum", ix, row, ei, nd, x, w)
            oracle.spmm_coo("copy_lhs", "max", ix, row, None, nd, x, None)
            y = rng.random((nd, 6)).astype(fdt)
            for op in ("add", "mul", "dot", "copy_lhs"):
                oracle.sddmm_coo(op, ix, row, ei, x, y)
            sc = rng.standard_normal((e, 4)).astype(fdt)
            out = oracle.edge_softmax_fwd(ip, ei, sc)
            oracle.edge_softmax_bwd(ip, ei, out, out * sc)
            off = np.array([0, 0, e // 2, e], dtype=idt)
            for red in ("sum", "max", "min"):
                oracle.segment_reduce(red, w, off)
print("sanitized run complete")
"""


@pytest.mark.timeout(900)
def test_oracle_under_asan_and_ubsan():
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_build/liboracle_asan.so"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.exists(asan):
        pytest.skip("no AddressSanitizer runtime in this image")
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(asan), DGLA_ORACLE_LIB=lib, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=850)
    report = p.stdout + p.stderr
    assert p.returncode == 0, report[-4000:]
    assert "sanitized run complete" in p.stdout
    assert "AddressSanitizer" not in report and "runtime error" not in report, report[-4000:]
