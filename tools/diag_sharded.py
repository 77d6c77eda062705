import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.graphgen import synth_csr
from dgl_amd import _capi
from dgl_amd.parallel import ShardedSpMM, SimulatedExchange, partition_assignment, shard_from_partition
dev = torch.device("cuda:0")
n, e, f, k = 40_000, 900_000, 100, 2
g = synth_csr(n, n, e, "L", seed=5, device=dev)
torch.manual_seed(8)
x = torch.rand(n, f, device=dev) + 1
out_full = torch.empty(n, f, device=dev)
csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out_full), dtype=torch.uint8, device=dev)
_capi.spmm_csr("copy_lhs", "sum", csr, x, None, out_full, None, None, ws)
deg = (g["indptr"][1:] - g["indptr"][:-1]).long()
rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
exact = torch.zeros(n, f, dtype=torch.float64, device=dev).index_add_(0, rows, x.double()[g["indices"].long()])
part, stats = partition_assignment(g["indptr"], g["indices"], k, seed=3)
shards = [shard_from_partition(g["indptr"], g["indices"], part, k, r) for r in range(k)]
ex = SimulatedExchange(shards)
xs = [x[s["rows"]].contiguous() for s in shards]
for r in range(k): ex.bind(r, xs[r])
got = torch.empty_like(out_full)
for r, s in enumerate(shards):
    op = ShardedSpMM(s, (f,), x.dtype, dev, exchange=ex, rank=r)
    o = torch.full((s["n_local"], f), float("nan"), device=dev)
    op.step(xs[r], o)
    got[s["rows"]] = o
def rel(a): return ((a.double() - exact).abs() / exact.abs().clamp_min(1e-30))
for name, a in (("single", out_full), ("sharded", got)):
    r = rel(a); m = r.max(dim=1).values; w = int(m.argmax())
    print(name, "max rel err vs exact", float(m.max()), "row", w, "deg", int(deg[w]), "value", float(exact[w, 0]), "n rows > 1e-5:", int((m > 1e-5).sum()))
d = ((got - out_full).abs() / out_full.abs()).max(dim=1).values
w = int(d.argmax())
print("sharded vs single: row", w, "deg", int(deg[w]), "rel", float(d[w]), "single err", float(rel(out_full)[w].max()), "sharded err", float(rel(got)[w].max()))
# the same for a sequential fp32 sum (the reference's order)
import oracle
ref, _, _ = oracle.spmm_csr("copy_lhs", "sum", g["indptr"].cpu().numpy(), g["indices"].cpu().numpy(), None, x.cpu().numpy(), None)
r = rel(torch.from_numpy(ref).to(dev)); m = r.max(dim=1).values; w = int(m.argmax())
print("reference order max rel err vs exact", float(m.max()), "row", w, "deg", int(deg[w]), "rows > 1e-5:", int((m > 1e-5).sum()))
