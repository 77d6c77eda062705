"""One hipGraph per step: record a function of STATIC-shape device tensors once, replay it.

MI355X-first counterpart of what the reference leaves to the host: its mini-batch loop pays a
Python / FFI / launch round trip per kernel (≈ 100 per GraphSAGE step) and a read-back per
sampling layer (python/dgl/dataloading/neighbor_sampler.py).  With the padded sampling calls
(`NeighborSampler.sample_blocks_padded`) every tensor of a step has a static shape, so the step is
recorded once and replayed with one launch.  PyTorch owns streams and the capture itself
(`torch.cuda.CUDAGraph` = hipGraph on ROCm); this class only packages the protocol the library's
operators need: warm-up on a side stream (plans, scratch buffers and format conversions are
built lazily and must exist before recording), static input buffers, and no allocation or
synchronisation inside the recorded region (the padded entry points guarantee that for the
library; `torch.cuda.graph` gives tensors created inside the region stable addresses)."""
import torch


class CapturedStep:
    """``step = CapturedStep(fn, inputs)``; ``outputs = step(name=tensor, ...)``.

    ``fn(**inputs)`` must read its batch-dependent data ONLY from the tensors in ``inputs`` (a dict
    of device tensors whose shapes never change) and may update module state in place (parameters,
    optimiser state, the sampler's draw counter).  Whatever it returns (a tensor, or a tuple / list /
    dict of tensors) is kept as static output storage: a later call overwrites it.  Calling the
    object copies the given tensors into the static inputs and replays the graph."""

    def __init__(self, fn, inputs, warmup=3):
        self.inputs = dict(inputs)
        for k, t in self.inputs.items():
            if not (torch.is_tensor(t) and t.is_cuda):
                raise ValueError("CapturedStep: input %r must be a device tensor" % k)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                fn(**self.inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn(**self.inputs)

    def __call__(self, **new_inputs):
        for k, t in new_inputs.items():
            self.inputs[k].copy_(t)
        self.graph.replay()
        return self.outputs
