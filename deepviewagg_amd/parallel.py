"""Multi-GPU support of the hot path: one process per GPU (torch.distributed, backend "nccl" == RCCL
over xGMI on ROCm), scenes sharded by point-cloud tile, no collective on the data path.

The reference is single-GPU (trainer.py:59-61), so there is nothing to mirror: the
pooled feature of a point depends only on its own views and images seen by several tiles are
replicated, hence the only exchange is the data-parallel sum of parameter gradients (SURVEY.md 8(e):
~28 M parameters = 112 MB fp32 for the full model; the pooling modules themselves own a few kB).

``GradientBucket`` flattens a parameter list into fixed-size fp32 buckets.  ``start()`` enqueues, per bucket,
copy-in -> all-reduce -> (average) on a SIDE stream that waits for the producing stream, so the ring transfers
over xGMI (per-link bound: few large messages) run under whatever the main stream computes next -- in bench.py
the backward of the view pooling; ``finish()`` makes the main stream wait and writes the reduced gradients back.
``reduce()`` = ``start()`` + ``finish()``.
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


def shard_mapping(csr_idx, tile_points):
    """CSR pointers + view selection of the points ``tile_points`` (sorted LongTensor) of a scene mapping:
    returns (csr_tile [n + 1], view_index [V_tile]) -- the views of a point travel with the point."""
    sizes = (csr_idx[1:] - csr_idx[:-1])[tile_points]
    csr_tile = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
    starts = csr_idx[:-1][tile_points]
    view_index = torch.repeat_interleave(starts - csr_tile[:-1], sizes) + torch.arange(
        int(csr_tile[-1]), device=csr_idx.device)
    return csr_tile, view_index


class GradientBucket:
    """Bucketed data-parallel gradient all-reduce (sum or mean) of a parameter list.

    Contract between ``start()`` and ``finish()``: the gradients of the bucket's parameters belong to the side
    stream.  ``start()`` reads ``p.grad`` there (and, for a single contiguous fp32 gradient, all-reduces ``p.grad``
    itself in place), so NOTHING on the main stream may write or read these gradients until ``finish()`` has
    returned: no gradient accumulation into them (a second backward), no ``zero_grad``, no clipping, no optimizer
    step.  Call ``start()`` only after the backward that produces the bucket's gradients has been enqueued (autograd
    hooks that fire later would race with the collective); ``finish()`` makes the main stream wait for the side
    stream before it hands the gradients back.  The timing events of the last ``start()`` / ``finish()`` pair are
    kept (``timings()``): duration of the collective on the side stream and the part of it the main stream had to
    wait for.

    BatchNorm: the recompute chain's batch statistics (fp64 sums per layer) stay per process, like the reference's
    plain ``BatchNorm1d``: with dp > 1 every rank normalises with the statistics of its own
    tile; only the parameter gradients are summed."""

    def __init__(self, params, process_group=None, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.sizes = [p.numel() for p in self.params]
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        per = max(1, bucket_bytes // 4)
        n = self.flat.numel()
        self.chunks = [(o, min(o + per, n)) for o in range(0, n, per)] or [(0, 0)]
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._handles = None
        self._average = True
        self._inplace = False
        self._scale = False
        self._buf = self.flat
        self._ev = None        # (collective start, collective end, main-stream wait start, wait end) of the last pair

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def start(self, average=True):
        """Copy the current gradients into the flat buffer and enqueue the bucket all-reduces; on a HIP device
        they run on the bucket's own stream (after everything already enqueued on the current stream)."""
        self._average = average
        if not (dist.is_available() and dist.is_initialized()):
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            ctx = torch.cuda.stream(self.stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        world = dist.get_world_size(self.group)
        # mean inside the collective where the backend has it (RCCL): no separate scaling pass over the bucket
        avg_op = average and dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM
        ev = None
        if self.stream is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with ctx:
            if ev is not None:
                ev[0].record()
            # a single contiguous fp32 gradient is reduced where it lies (what DDP's gradient-as-bucket-view gives
            # every parameter): no flatten / unflatten copies
            g0 = self.params[0].grad if len(self.params) == 1 else None
            self._inplace = g0 is not None and g0.is_contiguous() and g0.dtype == torch.float32
            if self._inplace:
                buf = g0.view(-1)
            else:
                buf = self.flat
                # one multi-tensor copy instead of one launch per parameter
                srcs = [p.grad.reshape(-1).to(torch.float32) if p.grad is not None else
                        torch.zeros(n, dtype=torch.float32, device=buf.device)
                        for p, n in zip(self.params, self.sizes)]
                if srcs:
                    torch._foreach_copy_(list(buf.split(self.sizes)), srcs)
            self._buf = buf
            self._handles = []
            for a, b in self.chunks:
                if b > a:
                    self._handles.append(dist.all_reduce(buf[a:b], op=op, group=self.group, async_op=True))
            self._scale = average and not avg_op and world > 1
            if self.stream is not None:
                for hnd in self._handles:
                    hnd.wait()               # orders the collective inside the side stream, does not block the host
                if self._scale:
                    buf.div_(world)
                ev[1].record()
        self._ev = ev

    def timings(self):
        """(collective_ms, exposed_ms) of the last start() / finish() pair on a HIP device (synchronises), else None:
        the time the bucket's copy-in + all-reduce took on the side stream, and how long the main stream was blocked
        in finish() waiting for it (0 when the collective was fully hidden under the work enqueued in between)."""
        if self._ev is None or self.stream is None:
            return None
        torch.cuda.synchronize(self.flat.device)
        e = self._ev
        return e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])

    def finish(self):
        """Wait for the all-reduces and write the reduced gradients back into ``p.grad``."""
        if not (dist.is_available() and dist.is_initialized()) or self._handles is None:
            return
        if self.stream is not None:
            main = torch.cuda.current_stream(self.flat.device)
            if self._ev is not None:
                self._ev[2].record(main)
            main.wait_stream(self.stream)
            if self._ev is not None:
                self._ev[3].record(main)
        else:
            for hnd in self._handles:
                hnd.wait()
            if self._scale:
                self._buf.div_(dist.get_world_size(self.group))
        self._handles = None
        if self._inplace:
            return
        views = [v.view_as(p) for v, p in zip(self.flat.split(self.sizes), self.params)]
        dst, src = [], []
        for p, g in zip(self.params, views):
            if p.grad is None:
                p.grad = g.clone().to(p.dtype)
            else:
                dst.append(p.grad)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)

    def reduce(self, average=True):
        """All-reduce the gradients of the bucket's parameters in place."""
        self.start(average)
        self.finish()
