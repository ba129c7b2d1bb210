"""Test-side transports for the product's collective logic (trep_amd/rccl.py::RowCollective).

GlooTransport implements the two transport primitives on torch.distributed's gloo backend (CPU); the pad / trim /
concatenate / reduce logic under test is RowCollective's own, exactly what the RCCL Communicator runs on the GPU box.
torch lives here, in tests/, and nowhere under trep_amd/."""
import numpy as np

from trep_amd import rccl


class GlooTransport(rccl.RowCollective):
    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self._group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def _all_reduce_host(self, v, op):
        import torch
        dist = self._dist
        t = torch.from_numpy(v)          # shares memory: in place
        dist.all_reduce(t, op={rccl.SUM: dist.ReduceOp.SUM, rccl.MAX: dist.ReduceOp.MAX, rccl.MIN: dist.ReduceOp.MIN}[op], group=self._group)

    def _all_gather_block(self, block):
        import torch
        out = torch.empty((self.world * block.shape[0],) + tuple(block.shape[1:]), dtype=torch.float64)
        self._dist.all_gather_into_tensor(out, torch.from_numpy(np.ascontiguousarray(block)), group=self._group)
        return out.numpy().reshape((self.world,) + tuple(block.shape))
