#!/usr/bin/env python3
"""Static audit of the gfx950 code objects under build/csrc (no GPU needed).

Two defects were found by reading disassembly in round 3, both invisible to the tests and both of the
kind a source change re-introduces silently:
  * FLAT memory instructions in a hot kernel (a pointer read from LDS is a generic pointer): the
    compiler cannot count their completion and drains every prefetch with `s_waitcnt vmcnt(0)`;
  * a run-time choice of load flavour between the prefetch loads of the merge kernel: same effect.
This tool extracts the device code of every object (`llvm-objdump --offloading`), disassembles it and
reports per kernel: flat / scratch instruction counts, global loads, the histogram of `vmcnt(N)` waits.
`python tools/isa_audit.py [--json] [object ...]`; tests/test_isa_audit.py asserts the invariants."""
import argparse
import collections
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_LABEL = re.compile(r"^[0-9a-f]+ <(.*)>:$")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")


def disassemble(obj):
    """Disassembly text of the gfx950 code object bundled in `obj` (None when there is none)."""
    tmp = tempfile.mkdtemp(prefix="isa_audit_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=False)
        cos = glob.glob(local + ".*gfx950*")
        if not cos:
            return None
        return subprocess.run([OBJDUMP, "-d", cos[0]], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              check=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def audit_text(text):
    """{kernel symbol: Counter} with keys flat, scratch, global_load, lds_dma, mfma and vmcnt<N>."""
    stats, cur = {}, None
    for line in text.splitlines():
        m = _LABEL.match(line)
        if m:
            cur = stats.setdefault(m.group(1), collections.Counter())
            continue
        if cur is None:
            continue
        body = line.split("//")[0]
        if "flat_" in body and re.search(r"\bflat_(load|store|atomic)", body):
            cur["flat"] += 1
        elif "scratch_" in body:
            cur["scratch"] += 1
        elif "global_load_lds" in body:
            cur["lds_dma"] += 1
        elif re.search(r"\b(global|buffer)_load", body):
            cur["global_load"] += 1
        if "v_mfma" in body:
            cur["mfma"] += 1
        w = _VMCNT.search(body)
        if w:
            cur["vmcnt%s" % w.group(1)] += 1
    return stats


def max_counted_wait(c):
    waits = [int(k[5:]) for k in c if k.startswith("vmcnt")]
    return max(waits) if waits else -1


def audit(objects):
    out = {}
    for obj in objects:
        text = disassemble(obj)
        if text is None:
            continue
        out[os.path.basename(obj)] = audit_text(text)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("objects", nargs="*", default=sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "*.o"))))
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    res = audit(args.objects)
    if args.json:
        print(json.dumps({o: {k: dict(c) for k, c in ks.items()} for o, ks in res.items()}))
        return 0
    for o, ks in res.items():
        flat = {k: c["flat"] for k, c in ks.items() if c["flat"]}
        scratch = {k: c["scratch"] for k, c in ks.items() if c["scratch"] and "rocprim" not in k}
        merge = {k: c for k, c in ks.items() if "spmm_csr_merge_kernel" in k}
        drained = [k for k, c in merge.items() if max_counted_wait(c) < 3]
        print("%-18s kernels %4d | flat ops in %3d | scratch in %2d (own kernels) | merge kernels %3d, drained prefetch in %d"
              % (o, len(ks), len(flat), len(scratch), len(merge), len(drained)))
        for k in list(flat)[:4] + list(scratch)[:4] + drained[:4]:
            print("     ", k[:140])
    return 0


if __name__ == "__main__":
    sys.exit(main())
