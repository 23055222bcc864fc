"""Data parallelism for the flow stage: one process per GPU, gradients all-reduced with RCCL.

Replaces the reference's single-process ``torch.nn.DataParallel`` (train.py:36-37: per-step
replicate / scatter / gather on GPU 0).  Samples are independent and every loss is a per-sample
value that is batch-meaned (train.py:147-150), so averaging the per-rank gradients of equal
per-rank batches reproduces DataParallel's global-batch mean exactly.

All 98 parameter gradients travel in ONE flat fp32 buffer (5,134,324 elements = 20.5 MB): a step needs a few large
all-reduces over xGMI instead of 98 small ones.  Backward assigns the gradients as in a single-process run; a piece of the
buffer is filled with one multi-tensor copy the moment its last gradient exists.  The buffer is cut into ``chunks`` pieces on parameter
boundaries; a piece is handed to RCCL (async, on RCCL's own stream) the moment backward has produced
its last gradient, so all but the final piece (the first pyramid layers, 2 % of the bytes) travel
while backward is still computing.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, device_index=None, force=False):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world).

    ``device_index``: the GPU this rank drives (default: LOCAL_RANK); it is made current and handed to
    ``init_process_group(device_id=...)`` so RCCL binds its communicator to that device up front instead of guessing
    at the first collective.  ``force``: create the process group even for a single process (world size 1) -- RCCL
    accepts that, and it lets one GPU rehearse the whole hook -> async all-reduce -> wait path.
    """
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'     # 'nccl' is RCCL on ROCm
        kw = {}
        if backend == 'nccl':
            idx = local_rank if device_index is None else int(device_index)
            torch.cuda.set_device(idx)
            kw['device_id'] = torch.device('cuda', idx)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


class FlatGradients:
    """Owns a flat gradient buffer for ``params`` and averages it across ranks.

    chunks: number of pieces of roughly equal size (cut on parameter boundaries).  xGMI is point-to-point,
    so a 5 MB piece is already per-link bound on a ring; finer buckets only add launch latency.
    overlap: launch each piece from a post-accumulate-grad hook while backward is still running
    (``all_reduce_mean`` then only launches what is left and waits).  Without it everything is launched
    after backward (used when the step is replayed as a hipGraph: collectives stay outside the graph).
    pack: how the gradients get into the buffer.  False: ``p.grad`` ARE views of the pre-zeroed buffer and backward accumulates
    into them (98 read-add-write kernels + a 20 MB fill per step).  True (what FlowTrainer uses): ``zero()`` drops the
    gradients, backward ASSIGNS them as in a single-process run, and a piece is copied into the buffer with one
    multi-tensor copy the moment its last gradient exists (then ``p.grad`` is re-pointed at the buffer, so the optimizer
    reads the reduced values) -- the data-parallel step then costs what the single-process step costs plus the exchange.
    """

    def __init__(self, params, chunks=4, group=None, overlap=False, single_rank_collectives=False, pack=False):
        self.single_rank_collectives = bool(single_rank_collectives)
        self.pack = bool(pack)
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dtype = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dtype)
        off, offsets, self.views = 0, [], []
        for p in self.params:
            n = p.numel()
            # a view with the PARAMETER's strides (channels_last convolution weights are dense but not row-major): the
            # fused optimizer walks parameter, gradient and moments by memory offset, so their layouts must agree
            self.views.append(self.flat[off:off + n].as_strided(p.size(), p.stride()))
            if not self.pack:
                p.grad = self.views[-1]
            offsets.append(off)
            off += n
        self.group = group
        # piece boundaries on parameter boundaries, ~numel/chunks elements each
        chunks = max(1, min(int(chunks), len(self.params)))
        target, bounds, acc = self.numel / chunks, [0], 0
        for i, p in enumerate(self.params):
            acc += p.numel()
            if acc >= target * len(bounds) and len(bounds) < chunks and i + 1 < len(self.params):
                bounds.append(i + 1)
        bounds.append(len(self.params))
        self.pieces = []                                  # (first element, one past the last, number of params)
        self._piece_params = []                           # parameter index range of each piece
        for a, b in zip(bounds[:-1], bounds[1:]):
            end = offsets[b] if b < len(self.params) else self.numel
            self.pieces.append((offsets[a], end, b - a))
            self._piece_params.append((a, b))
        self.chunks = len(self.pieces)
        self._piece_of = {}
        for k, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
            for p in self.params[a:b]:
                self._piece_of[id(p)] = k
        self._left = [n for _, _, n in self.pieces]
        self._works = [None] * self.chunks
        self._packed = [False] * self.chunks
        self._sources = None                              # gradient tensors of a captured hipGraph (remember_sources)
        self.track_sources = False                        # first passes of a trainer: remember where the packed gradients came from
        self.seen_sources = set()                         # ... (addresses; ops.deferred_bias_grads.adopted(extra=...))
        self._seen_tensors = []                           # ... and the tensors, alive until zero(): the allocator cannot hand a seen address to a later gradient
        self.before_pack = None                           # called before gradients are READ (ops.deferred_bias_grads.flush on the GPU)
        self.launched_early = 0                           # pieces sent from a hook during the last backward
        self.overlap = False
        self._hooks = []
        if overlap:
            self.enable_overlap()

    # ---- collective plumbing
    def _active(self):
        """Collectives run when a process group exists and has more than one rank -- or exactly one rank when the
        owner asked for it (``single_rank_collectives``: the RCCL rehearsal on a one-GPU box)."""
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or self.single_rank_collectives

    def _averaging_op(self):
        """RCCL averages in the collective itself; gloo (the CPU tests) sums and the buffer is scaled afterwards."""
        return dist.ReduceOp.AVG if dist.get_backend(self.group) == 'nccl' else dist.ReduceOp.SUM

    def _pack_piece(self, k, from_graph=False):
        """Copy the (assigned) gradients of piece k into the buffer and re-point ``p.grad`` at it.  ``from_graph``: the
        step was a hipGraph replay, whose backward wrote the graph's own static tensors (``remember_sources``); an EAGER
        step -- also one that follows a capture, e.g. a differently shaped batch -- packs what backward assigned to
        ``p.grad``, never the graph's buffers (they hold the previous replay's gradients)."""
        if self._packed[k]:
            return
        if self.before_pack is not None:
            self.before_pack()
        a, b = self._piece_params[k]
        dst, src = [], []
        for i in range(a, b):
            g = self._sources[i] if from_graph else self.params[i].grad
            if g is None:
                self.views[i].zero_()                     # a parameter the loss did not reach
            elif g.data_ptr() != self.views[i].data_ptr():
                dst.append(self.views[i]); src.append(g)
                if self.track_sources:
                    self.seen_sources.add(g.data_ptr())
                    self._seen_tensors.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        for i in range(a, b):
            self.params[i].grad = self.views[i]
        self._packed[k] = True

    def _launch(self, k, from_graph=False):
        if self.pack:
            self._pack_piece(k, from_graph)
        a, b, _ = self.pieces[k]
        self._works[k] = dist.all_reduce(self.flat[a:b], op=self._averaging_op(), group=self.group, async_op=True)

    def enable_overlap(self):
        """Hooks that send a piece during backward.  pack mode: hooks on the first THREE parameters of each piece only --
        backward produces the gradients in reverse registration order (weight and bias of one convolution in either
        order), so the piece's last gradient is one of those; a hook sends the piece when every gradient of it exists
        (``zero()`` dropped them, autograd assigns each exactly once per backward) and otherwise leaves it to a later
        hook or to ``all_reduce_mean``.  (98 Python hook calls per backward cost ~3 % of a step that is close to
        launch-bound.)  View mode: a counting hook on every parameter."""
        if self.overlap:
            return
        self.overlap = True
        if self.pack:
            def make(k):
                a, b = self._piece_params[k]
                mine = self.params[a:b]

                def hook(_p):
                    if self._works[k] is None and self._active() and all(q.grad is not None for q in mine):
                        self._launch(k)
                        self.launched_early += 1
                return hook
            self._hooks = [self.params[i].register_post_accumulate_grad_hook(make(k)) for k in range(self.chunks)
                           for i in range(self._piece_params[k][0], min(self._piece_params[k][0] + 3, self._piece_params[k][1]))]
            return

        def hook(p):
            k = self._piece_of[id(p)]
            self._left[k] -= 1
            if self._left[k] == 0 and self._works[k] is None and self._active():
                self._launch(k)                           # every gradient of piece k is final: send it now
                self.launched_early += 1
        self._hooks = [p.register_post_accumulate_grad_hook(hook) for p in self.params]

    def pack_all(self):
        """Copy every piece's (assigned) gradients into the buffer and point ``p.grad`` at its views -- what the trainer puts at
        the END of a captured forward + backward graph: a replay then leaves the whole gradient in ``flat``, ready for ONE
        collective (``all_reduce_flat``), and the optimizer graph that follows reads the views."""
        if not self.pack:
            return
        for k in range(self.chunks):
            self._pack_piece(k)

    def all_reduce_flat(self):
        """Average the whole buffer over the ranks with a single collective (the replayed step: nothing to overlap with, so
        one 20.5 MB message instead of four pieces).  On RCCL the call is enqueued on the communicator's stream and the
        current stream waits for it -- the host does not."""
        if not self._active():
            return
        if self.flat.is_cuda and dist.get_backend(self.group) == 'gloo':
            # gloo blocks the host for its staged copy anyway, but its own wait on a stream with a hipGraph replay in flight stalls
            # for seconds at a time (two ranks sharing one GPU: 6 ms after a synchronise, up to 18 s without,
            # profiles/r4_multirank_step_mode.md); gloo with device tensors is the one-GPU rehearsal, never the product path
            torch.cuda.current_stream().synchronize()
        dist.all_reduce(self.flat, op=self._averaging_op(), group=self.group)
        if self._averaging_op() != dist.ReduceOp.AVG and dist.get_world_size(self.group) > 1:
            self.flat.mul_(1.0 / dist.get_world_size(self.group))

    def remember_sources(self):
        """After a hipGraph capture of forward + backward: the tensors backward assigned as gradients are the graph's own
        static buffers, rewritten by every replay; ``all_reduce_mean(from_graph=True)`` copies from them."""
        if self.pack:
            self._sources = [p.grad for p in self.params]

    def zero(self):
        if self.pack:
            # always dropped: an eager backward must ASSIGN fresh gradients (after an exchange ``p.grad`` are views of the
            # buffer, which still holds the last reduced values; accumulating into them would apply stale gradients).  A
            # hipGraph replay never comes through here (its backward writes the captured tensors, not ``p.grad``).
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()
        self._left = [n for _, _, n in self.pieces]
        self._works = [None] * self.chunks
        self._packed = [False] * self.chunks
        self.launched_early = 0
        self.seen_sources.clear()
        self._seen_tensors = []

    def vector(self):
        """The gradients as one vector in LOGICAL element order, parameter by parameter (a copy).  ``flat`` itself is
        in memory order, which differs for channels_last weights; compare runs with this."""
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])

    def check_views(self):
        """Guard against something (e.g. ``zero_grad(set_to_none=True)``) having detached a view.  (pack mode: valid after
        ``all_reduce_mean`` of a step that exchanged.)"""
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + off * self.flat.element_size():
                raise RuntimeError('a parameter gradient no longer aliases the flat buffer; '
                                   'use FlatGradients.zero() instead of optimizer.zero_grad()')
            off += p.numel()

    def all_reduce_mean(self, from_graph=False):
        """Average the flat gradient over all ranks (no-op for a single process).  Pieces already sent by the
        hooks are only waited for; the rest is launched here.  ``from_graph``: the gradients of this step are in the
        tensors of the captured hipGraph (``remember_sources``), not in ``p.grad``."""
        if from_graph and (not self.pack or self._sources is None):
            raise RuntimeError('all_reduce_mean(from_graph=True) needs pack mode and remember_sources() after the capture')
        if not self._active():
            if from_graph:                                # nothing to exchange: the optimizer reads the graph's tensors
                for p, g in zip(self.params, self._sources):
                    p.grad = g
            return
        world = dist.get_world_size(self.group)
        for k in range(self.chunks):
            if self._works[k] is None:
                self._launch(k, from_graph)
        for w in self._works:
            w.wait()
        self._works = [None] * self.chunks
        self._packed = [False] * self.chunks              # (a hipGraph replay has no zero() between two exchanges)
        if self._averaging_op() != dist.ReduceOp.AVG and world > 1:
            self.flat.mul_(1.0 / world)


class PlainGradients:
    """Single-process counterpart of FlatGradients with the same small interface.  Without an exchange step there is
    no reason to alias the gradients into one buffer: ``zero()`` drops them (``grad = None``), so backward ASSIGNS each
    parameter's gradient instead of launching a read-add-write per parameter into a pre-zeroed buffer (98 extra
    kernels and one 20 MB fill per step on the 832x256 step: ~0.5 ms of 27)."""

    overlap = False
    chunks = 0
    launched_early = 0

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        self.numel = sum(p.numel() for p in self.params)

    def zero(self):
        for p in self.params:
            p.grad = None

    def all_reduce_mean(self):
        return None

    def check_views(self):
        return None

    def vector(self):
        """The gradients as one vector in logical element order (a copy; diagnostics and tests)."""
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])

    @property
    def flat(self):
        return self.vector()


def broadcast_parameters(module, src=0, group=None, single_rank=False):
    """Rank ``src``'s weights to everyone (DataParallel's per-step replicate, done once)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not single_rank):
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
