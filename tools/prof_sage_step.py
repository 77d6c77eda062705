"""Where a mini-batch GraphSAGE step (benchmarks/bench_sage.py) spends its time: phases timed with
a synchronisation after each (so the sum exceeds the pipelined step)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import dgl_amd as dgl  # noqa: E402
import dgl_amd.function as fn  # noqa: E402
from dgl_amd.graph_index import GraphIndex, Relation  # noqa: E402
from dgl_amd.heterograph import DGLGraph  # noqa: E402
from tests.graphgen import C2_EDGES, C2_NODES, synth_csr  # noqa: E402

dev = torch.device("cuda:0")
n, e, f, classes, hidden, batch = C2_NODES, C2_EDGES, 100, 47, 256, 1024
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


acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


for it in range(60):
    if it == 10:
        acc.clear()
    t = time.perf_counter()
    seeds = torch.randint(0, n, (batch,), device=dev, generator=gen).unique()
    t = tick("seeds.unique", t)
    inp, out, blocks = sampler.sample_blocks(g, seeds)
    t = tick("sample_blocks (2 x sample + to_block)", t)
    h = feat[inp]
    t = tick("feature gather", t)
    h = torch.relu(sage(blocks[0], h, params[0], params[1]))
    logits = sage(blocks[1], h, params[2], params[3])
    loss = torch.nn.functional.cross_entropy(logits, labels[out.long()])
    t = tick("forward", t)
    grads = torch.autograd.grad(loss, params)
    t = tick("backward", t)
    with torch.no_grad():
        for p, gr in zip(params, grads):
            p -= 0.1 * gr
    t = tick("sgd", t)
for k, v in acc.items():
    print("%-42s %.3f ms" % (k, v / 50 * 1e3))
print("%-42s %.3f ms" % ("sum (synchronised after every phase)", sum(acc.values()) / 50 * 1e3))

# cost of the reverse-format build the backward pass needs (CSC of the reversed block = CSR of the block)
import dgl_amd.graph_index as gi  # noqa: E402
tot = 0.0
for it in range(20):
    seeds = torch.randint(0, n, (batch,), device=dev, generator=gen).unique()
    inp, out, blocks = sampler.sample_blocks(g, seeds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in blocks:
        b._graph.relations[0].reverse().csc()
    torch.cuda.synchronize()
    tot += time.perf_counter() - t0
print("%-42s %.3f ms (both blocks; %d + %d edges)" % ("reverse CSC build", tot / 20 * 1e3, blocks[0].num_edges(), blocks[1].num_edges()))
