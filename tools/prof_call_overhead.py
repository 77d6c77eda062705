"""cProfile of the host side of one small-graph operator call (Cora-sized graph)."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
import dgl_amd as dgl  # noqa: E402
import dgl_amd.function as fn  # noqa: E402

dev = torch.device("cuda:0")
g = dgl.rand_graph(2708, 10556, device=dev, seed=1, idtype=torch.int32)
x = torch.rand(2708, 16, device=dev)


def api():
    with g.local_scope():
        g.ndata["h"] = x
        g.update_all(fn.copy_u("h", "m"), fn.sum("m", "o"))
        return g.ndata["o"]


which = api if "update_all" in sys.argv else (lambda: dgl.ops.copy_u_sum(g, x))
for _ in range(50):
    which()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    which()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
