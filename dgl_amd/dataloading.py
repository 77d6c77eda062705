"""``dgl.dataloading`` names of the block samplers (python/dgl/dataloading/neighbor_sampler.py): the samplers
themselves live in :mod:`dgl_amd.sampling`; the data-loader machinery around them (worker processes, prefetching,
DDP wrappers) is the reference's control plane and is not rebuilt here."""
from .sampling import NeighborSampler

MultiLayerNeighborSampler = NeighborSampler      # the reference keeps both names (neighbor_sampler.py:204)


class MultiLayerFullNeighborSampler(NeighborSampler):
    """Every in-edge of every seed, ``num_layers`` times (neighbor_sampler.py:207-240)."""

    def __init__(self, num_layers, **kwargs):
        super().__init__([-1] * int(num_layers), **kwargs)


__all__ = ["NeighborSampler", "MultiLayerNeighborSampler", "MultiLayerFullNeighborSampler"]
