"""Narrow-feature reductions over rows cut into MANY units (hubs; a few huge segments): the fix-up of csrc/narrow_reduce.hip.
`DGLA_NARROW_REDUCE=0 python benchmarks/exp_narrow_hubs.py` times the merge kernel on the same inputs
(profiles/r5/narrow_hub_rows.jsonl)."""
import os, sys, torch, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "benchmarks"))
from bench_ops import timeit
from dgl_amd import _capi
dev = torch.device("cuda:0")
rows = 15464785
cases = {
    "64 equal segments (241 637 rows each: 944 units per segment)": torch.full((64,), rows // 64, dtype=torch.int64),
    "power-law-like: 600 K segments of 1 ... 40 rows + 2 000 hubs of 1 500 ... 20 000 rows": None,
}
g = torch.Generator().manual_seed(1)
small = torch.randint(1, 41, (600000,), generator=g)
hubs = torch.randint(1500, 20001, (2000,), generator=g)
mix = torch.cat([small, hubs])[torch.randperm(602000, generator=g)]
cases["power-law-like: 600 K segments of 1 ... 40 rows + 2 000 hubs of 1 500 ... 20 000 rows"] = mix
for name, seglen in cases.items():
    n, m = seglen.numel(), int(seglen.sum())
    offsets = torch.zeros(n + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(seglen, 0)
    offsets = offsets.to(dev)
    for f in (1, 4, 8):
        x = torch.rand(m, f, device=dev)
        for red in ("sum", "max"):
            out = torch.empty(n, f, device=dev)
            arg = torch.empty(n, f, dtype=torch.int64, device=dev) if red != "sum" else None
            ws = torch.empty(max(1, _capi.segment_reduce_workspace_bytes(red, x, offsets, out)), dtype=torch.uint8, device=dev)
            _capi.segment_reduce(red, x, offsets, out, arg, ws)
            ms, mn = timeit(lambda: _capi.segment_reduce(red, x, offsets, out, arg, ws, plan_valid=True))
            print(json.dumps({"narrow_calls": _capi.narrow_reduce_calls(), "segments": name, "rows": m, "op": "segment_reduce %s F=%d" % (red, f),
                              "ms": round(ms, 4)}), flush=True)
