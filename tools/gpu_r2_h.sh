#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2h
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log
python - <<'PY' > $OUT/static_e.jsonl 2>$OUT/static_e.err
import json, sys, numpy as np, torch
sys.path.insert(0, '.')
import dgl_amd as dgl
from dgl_amd import ops
from tests.graphgen import synth_csr, C2_NODES, C2_EDGES
dev = torch.device("cuda:0")
n, e, f = C2_NODES, C2_EDGES, 100
gg = synth_csr(n, n, e, "U", device=dev, idtype=torch.int32)
dst = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), (gg["indptr"][1:] - gg["indptr"][:-1]).long())
perm = torch.randperm(e, device=dev)
g = dgl.graph((gg["indices"][perm].contiguous(), dst[perm].contiguous()), num_nodes=n)
del gg, dst, perm
x = torch.rand(n, f, device=dev) + 1
w = torch.rand(e, 1, device=dev) + 0.5
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        fn(); ev[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]))
ms_plain = timeit(lambda: ops.u_mul_e_sum(g, x, w))
dgl.static_features(w)
ms_static_w = timeit(lambda: ops.u_mul_e_sum(g, x, w))
dgl.static_features(x)
ms_static_both = timeit(lambda: ops.u_mul_e_sum(g, x, w))
print(json.dumps({"op": "u_mul_e_sum via dgl.ops, C2, random edge ids (edge-id map)", "plain_ms": ms_plain,
                  "static_edge_weights_ms": ms_static_w, "static_weights_and_features_ms": ms_static_both}))
PY
cat $OUT/static_e.jsonl; tail -3 $OUT/static_e.err
