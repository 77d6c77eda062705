"""HIP-graph capture of the hot path: a GAT layer (u_add_v SDDMM -> fused edge softmax ->
u_mul_e SpMM) and a GraphConv aggregation are captured into a torch.cuda.CUDAGraph (= hipGraph
on ROCm) after one eager warm-up and replayed on new inputs.  This pins down what DESIGN.md
claims for the library: launches go to the CURRENT stream, no hidden synchronisation, no
synchronous copies, scratch cached per graph — otherwise capture fails.  Results must equal
the eager run bit for bit (same kernels, same order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gat_layer_and_graphconv_replay_from_a_hip_graph(dev):
    import dgl_amd as dgl
    import dgl_amd.function as fn

    n, m, H, D = 3000, 40000, 8, 16
    g = dgl.rand_graph(n, m, device=dev, seed=11)
    ft = torch.randn(n, H, D, device=dev)
    el = torch.randn(n, H, 1, device=dev)
    er = torch.randn(n, H, 1, device=dev)

    def layer():
        with g.local_scope():
            g.srcdata.update({"ft": ft, "el": el})
            g.dstdata.update({"er": er})
            g.apply_edges(fn.u_add_v("el", "er", "e"))
            e = torch.nn.functional.leaky_relu(g.edata.pop("e"), 0.2)
            g.edata["a"] = dgl.edge_softmax(g, e)
            g.update_all(fn.u_mul_e("ft", "a", "m"), fn.sum("m", "o"))
            g.update_all(fn.copy_u("ft", "m"), fn.max("m", "mx"))
            return g.dstdata["o"].clone(), g.dstdata["mx"].clone()

    eager = layer()                      # warm-up: builds CSC / COO, attaches scratch, caches plans
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        layer()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = layer()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(captured, eager):
        assert torch.equal(a, b)
    # new inputs in place: the replay reads the same buffers
    ft.copy_(torch.randn_like(ft))
    el.copy_(torch.randn_like(el))
    want = layer()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(captured, want):
        assert torch.equal(a, b)
