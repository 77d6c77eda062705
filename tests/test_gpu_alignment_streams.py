"""(1) Operands whose base pointers are NOT 16-byte aligned (views into larger buffers): the
kernels pick their access width from the actual pointers, so every such call must still be
exact.  (2) Independent graphs on independent HIP streams at the same time: no hidden shared
state between calls (per-graph scratch, thread-local launch stream)."""
import numpy as np
import pytest
import torch

import oracle
from tests.graphgen import coo_to_csc

pytestmark = pytest.mark.gpu


def _view(dev, arr, off_elems):
    """A contiguous tensor equal to `arr` whose storage starts `off_elems` elements into a
    larger buffer (base pointer misaligned by off_elems * itemsize bytes)."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    buf = torch.empty(t.numel() + 16, dtype=t.dtype, device=dev)
    v = buf[off_elems: off_elems + t.numel()].view(t.shape)
    v.copy_(t)
    assert v.is_contiguous() and v.data_ptr() == buf.data_ptr() + off_elems * t.element_size()
    return v


@pytest.mark.parametrize("feat", [100, 64, 25, 8])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("offs", [(1, 0, 0), (0, 1, 0), (0, 0, 3), (3, 2, 1), (2, 2, 2)])
def test_spmm_with_misaligned_operands(dev, feat, dtype, offs):
    from dgl_amd import _capi

    rng = np.random.default_rng(feat)
    n_src, n_dst, e = 700, 500, 9000
    src, dst = rng.integers(0, n_src, e), rng.integers(0, n_dst, e)
    indptr, indices, eids = coo_to_csc(src, dst, n_dst, np.int32)
    u = (rng.random((n_src, feat)) + 0.5).astype(dtype)
    w = (rng.random((e, feat)) + 0.5).astype(dtype)
    t = lambda a: torch.from_numpy(a).to(dev)
    keep = (t(indptr), t(indices), t(eids))
    csr = _capi.make_csr(keep[0], keep[1], keep[2], n_src)
    tu, tw = _view(dev, u, offs[0]), _view(dev, w, offs[1])
    for op, red in (("mul", "sum"), ("copy_lhs", "max"), ("add", "min")):
        ref, ru, re_ = oracle.spmm_csr(op, red, indptr, indices, eids, u, w if op != "copy_lhs" else None)
        out = _view(dev, np.full(ref.shape, 7.0, dtype), offs[2])
        au = torch.empty(ref.shape, dtype=torch.int32, device=dev) if red != "sum" else None
        ae = torch.empty(ref.shape, dtype=torch.int32, device=dev) if red != "sum" and op != "copy_lhs" else None
        te = tw if op != "copy_lhs" else None
        ws = torch.empty(max(1, _capi.spmm_csr_workspace_bytes(op, red, csr, out.dtype, tu, te, out)),
                         dtype=torch.uint8, device=dev)
        _capi.spmm_csr(op, red, csr, tu, te, out, au, ae, ws)
        if red == "sum":
            np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5 if dtype == np.float32 else 1e-12)
        else:
            np.testing.assert_array_equal(out.cpu().numpy(), ref)
            np.testing.assert_array_equal(au.cpu().numpy(), ru)


@pytest.mark.parametrize("off", [0, 1, 3, 8])
@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
def test_segment_mm_with_misaligned_operands(dev, off, tdtype):
    from dgl_amd import _capi

    torch.manual_seed(off)
    seglen = torch.tensor([130, 0, 77, 300])
    a0 = (torch.rand(507, 64) - 0.5).to(tdtype)
    b0 = (torch.rand(4, 64, 48) - 0.5).to(tdtype)
    a = _view(dev, a0.view(torch.int16 if tdtype == torch.bfloat16 else torch.float32).numpy(), off).view(tdtype)
    b = _view(dev, b0.view(torch.int16 if tdtype == torch.bfloat16 else torch.float32).numpy(), off).view(tdtype)
    cbuf = torch.empty(507 * 48 + 16, dtype=tdtype, device=dev)
    c = cbuf[off: off + 507 * 48].view(507, 48)
    _capi.segment_mm(a, b, c, seglen)
    o = 0
    for i, m in enumerate(seglen.tolist()):
        want = a0[o:o + m].float() @ b0[i].float()
        tol = dict(rtol=1e-5, atol=1e-5) if tdtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
        assert torch.allclose(c[o:o + m].float().cpu(), want, **tol), (i, off)
        o += m


def test_two_graphs_on_two_streams(dev):
    import dgl_amd as dgl
    import dgl_amd.function as fn

    gs, xs, wants = [], [], []
    for k in range(2):
        g = dgl.rand_graph(4000 + 500 * k, 120000, device=dev, seed=20 + k)
        x = torch.rand(g.num_nodes(), 100, device=dev) + k
        g.ndata["h"] = x
        g.update_all(fn.copy_u("h", "m"), fn.sum("m", "o"))     # eager reference, default stream
        wants.append(g.ndata["o"].clone())
        gs.append(g)
        xs.append(x)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs = [None, None]
    for rep in range(20):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                gs[k].update_all(fn.copy_u("h", "m"), fn.max("m", "mx"))
                gs[k].update_all(fn.copy_u("h", "m"), fn.sum("m", "o"))
                outs[k] = gs[k].ndata["o"]
    for s in streams:
        s.synchronize()
    for k in range(2):
        assert torch.equal(outs[k], wants[k])
