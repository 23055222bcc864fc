"""Data parallelism for the flow stage: one process per GPU, gradients all-reduced with RCCL.

Replaces the reference's single-process ``torch.nn.DataParallel`` (train.py:36-37: per-step
replicate / scatter / gather on GPU 0).  Samples are independent and every loss is a per-sample
value that is batch-meaned (train.py:147-150), so averaging the per-rank gradients of equal
per-rank batches reproduces DataParallel's global-batch mean exactly.

All 98 parameter gradients live in ONE flat fp32 buffer (5,134,324 elements = 20.5 MB): backward
accumulates straight into views of it, and a step needs one (chunked, async) all-reduce over xGMI
instead of 98 small ones; ``zero_grad`` is one memset.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'     # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class FlatGradients:
    """Owns a flat gradient buffer for ``params`` and averages it across ranks.

    chunks: the buffer is reduced in ``chunks`` async pieces so the ring starts on the first bytes
    while later ones are still being queued; xGMI is point-to-point, so a 20 MB payload is already
    per-link bound (~0.25 ms on a ring) -- there is nothing to gain from finer buckets.
    """

    def __init__(self, params, chunks=4, group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dtype = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dtype)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self.chunks = max(1, int(chunks))
        self.group = group

    def zero(self):
        self.flat.zero_()

    def check_views(self):
        """Guard against something (e.g. ``zero_grad(set_to_none=True)``) having detached a view."""
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + off * self.flat.element_size():
                raise RuntimeError('a parameter gradient no longer aliases the flat buffer; '
                                   'use FlatGradients.zero() instead of optimizer.zero_grad()')
            off += p.numel()

    def all_reduce_mean(self):
        """Average the flat gradient over all ranks (no-op for a single process)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        works = []
        n = self.numel
        step = (n + self.chunks - 1) // self.chunks
        for s in range(0, n, step):
            works.append(dist.all_reduce(self.flat[s:s + step], op=dist.ReduceOp.SUM, group=self.group,
                                         async_op=True))
        for w in works:
            w.wait()
        self.flat.mul_(1.0 / world)


def broadcast_parameters(module, src=0, group=None):
    """Rank ``src``'s weights to everyone (DataParallel's per-step replicate, done once)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def shard_batch(batch, rank, world):
    """Rank r's contiguous slice of a global batch (DataParallel's scatter along dim 0)."""
    B = batch.shape[0]
    if B % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (B, world))
    per = B // world
    return batch[rank * per:(rank + 1) * per]
