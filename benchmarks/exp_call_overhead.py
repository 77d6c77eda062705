import sys, time, torch
sys.path.insert(0, '.')
import dgl_amd as dgl
from dgl_amd import _capi
dev = torch.device("cuda:0")
g = dgl.rand_graph(2708, 10556, device=dev, seed=1, idtype=torch.int32)
x = torch.rand(2708, 16, device=dev)
import dgl_amd.function as fn
def api():
    with g.local_scope():
        g.ndata["h"] = x
        g.update_all(fn.copy_u("h", "m"), fn.sum("m", "o"))
        return g.ndata["o"]
def ops(): return dgl.ops.copy_u_sum(g, x)
for name, f in (("update_all", api), ("ops.copy_u_sum", ops)):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): f()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "us per call: %.1f" % (dt / 2000 * 1e6))
