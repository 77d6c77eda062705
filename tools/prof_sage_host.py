"""Host-side profile (cProfile) of the mini-batch GraphSAGE step of benchmarks/bench_sage.py:
which Python / ctypes frames the 2 ms step is spent in.  Prints the top frames by own time."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
import dgl_amd as dgl  # noqa: E402
import dgl_amd.function as fn  # noqa: E402
from dgl_amd.graph_index import GraphIndex, Relation  # noqa: E402
from dgl_amd.heterograph import DGLGraph  # noqa: E402
from tests.graphgen import C2_EDGES, C2_NODES, synth_csr  # noqa: E402

dev = torch.device("cuda:0")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, e, f, classes, hidden, batch = C2_NODES // scale, C2_EDGES // scale, 100, 47, 256, 1024
gs = synth_csr(n, n, e, "L", seed=20250824, device=dev, idtype=torch.int64)
rel = Relation(n, n, csc=(gs["indptr"], gs["indices"], None), idtype=torch.int64, device=dev)
g = DGLGraph(GraphIndex([n], [(0, 0)], [rel]), ["_N"], [("_N", "_E", "_N")])
feat = torch.rand(n, f, device=dev)
labels = torch.randint(0, classes, (n,), device=dev)
params = [torch.randn(f, hidden, device=dev) * 0.05, torch.randn(f, hidden, device=dev) * 0.05,
          torch.randn(hidden, classes, device=dev) * 0.05, torch.randn(hidden, classes, device=dev) * 0.05]
for p in params:
    p.requires_grad_(True)
sampler = dgl.NeighborSampler([15, 10], seed=1)
gen = torch.Generator(device=dev).manual_seed(100)


def sage(blk, h, ws, wn):
    with blk.local_scope():
        blk.srcdata["h"] = h
        blk.update_all(fn.copy_u("h", "m"), fn.mean("m", "n"))
        return h[: blk.num_dst_nodes()] @ ws + blk.dstdata["n"] @ wn


def step():
    seeds = torch.randint(0, n, (batch,), device=dev, generator=gen).unique()
    inp, out, blocks = sampler.sample_blocks(g, seeds)
    h = feat[inp]
    h = torch.relu(sage(blocks[0], h, params[0], params[1]))
    logits = sage(blocks[1], h, params[2], params[3])
    loss = torch.nn.functional.cross_entropy(logits, labels[out.long()])
    grads = torch.autograd.grad(loss, params)
    with torch.no_grad():
        for p, gr in zip(params, grads):
            p -= 0.1 * gr
    return loss


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
