"""Multi-GPU support of the hot path: one process per GPU (torch.distributed, backend "nccl" == RCCL
over xGMI on ROCm), scenes sharded by point-cloud tile, no collective on the data path.

The reference is single-GPU (trainer.py:59-61), so there is nothing to mirror: the
pooled feature of a point depends only on its own views and images seen by several tiles are
replicated, hence the only exchange is the data-parallel sum of parameter gradients.
"""
import torch
import torch.distributed as dist


def tile_partition(xyz, n_tiles):
    """Split a cloud into ``n_tiles`` spatial tiles of (almost) equal point count: slabs along the
    longest horizontal axis.  Returns a list of LongTensors of point indices (original order kept
    inside a tile, which the mapping build's tie-breaks depend on)."""
    assert xyz.dim() == 2 and xyz.shape[1] == 3 and n_tiles >= 1
    extent = xyz[:, :2].max(0).values - xyz[:, :2].min(0).values
    axis = int(torch.argmax(extent))
    order = torch.sort(xyz[:, axis], stable=True).indices
    bounds = torch.linspace(0, xyz.shape[0], n_tiles + 1).round().long()
    return [torch.sort(order[int(bounds[i]):int(bounds[i + 1])]).values for i in range(n_tiles)]


class GradientBucket:
    """Flat fp32 bucket over a parameter list: one all-reduce (sum) per step instead of one per
    tensor; the ring all-reduce over xGMI is per-link bound, so few large messages beat many small
    ones.  ``reduce()`` may run on a side stream to overlap with the tail of the backward."""

    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.sizes = [p.numel() for p in self.params]
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)

    def reduce(self, average=True):
        """All-reduce the gradients of the bucket's parameters in place."""
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        off = 0
        for p, n in zip(self.params, self.sizes):
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if average:
            self.flat.div_(world)
        off = 0
        for p, n in zip(self.params, self.sizes):
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone().to(p.dtype)
            else:
                p.grad.copy_(g)
            off += n
