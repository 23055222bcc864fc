"""The --mode flow optimisation step (reference train.py:33-39,137-152) as a reusable object.

    zero_grad -> loss_pack = model(inputs) -> loss = sum_k w_k * mean_B(loss_pack[k])
    -> backward -> [all-reduce mean over ranks] -> Adam(lr).step()

The model is any module with the ``Model_flow`` contract (forward([B,3,3H,W]) -> dict of four [B]
tensors); on a GPU that is ``unopticalflow_amd.Model_flow`` (HIP kernels).  The trainer itself
holds no kernel code, which lets the multi-process logic be exercised on CPU with gloo.
"""
import torch

from .core.config import generate_loss_weights_dict
from .parallel import FlatGradients, PlainGradients, broadcast_parameters


def _on_device(t):
    """ops.on_device (a tensor the kernels can take: a HIP tensor), asked lazily -- a CPU oracle model never imports the operator layer"""
    from . import ops
    return ops.on_device(t)

_PROCESS_GC_FROZEN = False


class FlowTrainer:
    def __init__(self, cfg, model, distributed=False, allreduce_chunks=4, fused_adam=None, use_graph=False,
                 single_rank_collectives=False, gc_freeze_after=None, own_adam=None, defer_bias_grads=None):
        self.cfg = cfg
        self.model = model
        self.loss_weights = generate_loss_weights_dict(cfg)
        params = [p for p in model.parameters() if p.requires_grad]      # train.py:39
        if distributed:
            broadcast_parameters(model, single_rank=single_rank_collectives)
        # gradients leave for RCCL piece by piece during backward, except under hipGraph replay (collectives stay
        # outside the captured graph)
        # several ranks: one flat buffer that the all-reduce pieces are cut from; one process: plain per-parameter gradients
        # (assigned by backward, nothing to pre-zero or accumulate into)
        if distributed:
            self.grads = FlatGradients(params, chunks=allreduce_chunks, overlap=not use_graph,
                                       single_rank_collectives=single_rank_collectives, pack=True)
            if _on_device(params[0]):
                # bias gradients are finished by one batched launch at the END of a backward pass (ops.deferred_bias_grads); a piece
                # packed from a hook in the middle of the pass must see finished values
                from . import ops
                self.grads.before_pack = ops.deferred_bias_grads.flush
            if self.grads.overlap and hasattr(model, 'weight_shadow_groups'):
                # bf16 option: one weight-cast node per all-reduce piece, so that a piece's gradients exist (and its hook fires)
                # when backward has passed ITS layers, not at the very end of backward (net_utils.WeightShadows)
                model.weight_shadow_groups = max(model.weight_shadow_groups, 2 * self.grads.chunks)
        else:
            self.grads = PlainGradients(params)
        self.distributed = distributed
        self._defer_bias_grads = _on_device(params[0]) if defer_bias_grads is None else bool(defer_bias_grads)
        self._defer_checks = 2
        kw = {}
        if fused_adam is None:
            fused_adam = _on_device(params[0])
        if own_adam is None:
            own_adam = fused_adam and _on_device(params[0])
        if own_adam:
            # torch.optim.Adam whose step is one HIP launch over all tensors (optim.FlowAdam; same state layout and state_dict)
            from .optim import FlowAdam
            self.optimizer = FlowAdam([{'params': params, 'lr': cfg.lr}])
        else:
            if fused_adam:
                kw['fused'] = True
            if use_graph and params[0].is_cuda:
                kw['capturable'] = True                # step counters live on the device
            self.optimizer = torch.optim.Adam([{'params': params, 'lr': cfg.lr}], **kw)
        self.iteration = 0
        # hipGraph replay of the step (forward + loss + backward [+ Adam]): ~3000 kernel launches per
        # step become one graph launch, so the host never starves the GPU.  Built lazily from the first
        # batch; inputs are copied into a static buffer.
        self.use_graph = use_graph
        self._graph = None
        # host jitter: a step builds ~10^4 short-lived Python objects (autograd nodes, tensors, ctypes arguments); the
        # cyclic collector's full (generation-2) pass over the ~10^6 long-lived objects of an imported torch takes
        # 100-150 ms on the bench box -- more than the host's 2-3 step launch lead, so the GPU idles (one 50-90 ms step every
        # ~17: 26.3-27.3 ms mean against a 24.9 ms median, profiles/r3_headline_*.json).  With ``gc_freeze_after=N`` everything
        # alive after N steps is moved to the permanent generation (gc.freeze): later collections only look at what the steps
        # themselves allocate.  It is PROCESS-GLOBAL, so it is opt-in (train.py and bench.py ask for it, a library user's
        # process is left alone), done at most once per process, and ``close()`` undoes it.
        self.gc_freeze_after = gc_freeze_after
        self.fused_total_loss = True
        self._gc_frozen = False
        self._steps_here = 0                           # steps THIS object ran (a resumed run starts at iteration >> 2)

    def total_loss(self, loss_pack):
        """train.py:147-150; on the GPU one launch each way (ops.weighted_mean_sum) instead of a mean, a multiply and an add per key"""
        terms = list(loss_pack.values())
        if self.fused_total_loss and 1 <= len(terms) <= 8 and all(_on_device(t) and t.dtype == torch.float32 and t.dim() == 1 and
                                                                   t.shape == terms[0].shape for t in terms):
            from . import ops
            return ops.weighted_mean_sum(terms, [self.loss_weights[k] for k in loss_pack])
        loss = None
        for key in loss_pack:
            term = self.loss_weights[key] * loss_pack[key].mean()
            loss = term if loss is None else loss + term
        return loss

    def _backward(self, loss):
        """``loss.backward()`` of a step whose gradients were dropped by ``self.grads.zero()``: on the GPU the bias-gradient
        reductions of the pass finish in one launch at its end (ops.deferred_bias_grads; p.grad is None, so autograd adopts the
        tensors).  The first passes check that it did."""
        if not self._defer_bias_grads:
            loss.backward()
            return
        from . import ops
        checking = self._defer_checks > 0 and not torch.cuda.is_current_stream_capturing()
        if hasattr(self.grads, 'track_sources'):
            self.grads.track_sources = checking        # pack mode: pieces sent from hooks re-point p.grad; remember what they copied from
        with ops.deferred_bias_grads:
            loss.backward()
        if checking:
            self._defer_checks -= 1
            # every finished bias gradient must be a tensor autograd adopted: still some p.grad, or (data-parallel pack mode, eager
            # with hooks or not) the source a piece was packed from during this pass -- otherwise a clone of the unwritten tensor
            # would be averaged over the ranks
            if not ops.deferred_bias_grads.adopted(self.grads.params, extra=getattr(self.grads, 'seen_sources', ())):
                raise RuntimeError('deferred bias gradients were not adopted by autograd (a gradient was accumulated or cloned); '
                                   'construct FlowTrainer(defer_bias_grads=False)')

    def _eager_fwd_bwd(self, inputs):
        self.grads.zero()
        loss_pack = self.model(inputs)
        loss = self.total_loss(loss_pack)
        self._backward(loss)
        return loss, loss_pack

    def _build_graph(self, inputs):
        """Capture the step once.  One process: forward + loss + backward + Adam in one graph.  Several ranks: graph A =
        forward + loss + backward + the copy of every gradient into the flat buffer, then ONE all-reduce of that buffer
        outside any graph, then graph B = Adam on the buffer's views -- per step the host enqueues an input copy, two graph
        launches and one collective, whatever the ~3000 kernel launches inside would have cost it.
        The three warm-up iterations PyTorch needs before a capture (MIOpen picks its solvers, Adam creates its state) run on
        the first batch too, but must not count as training: parameters and optimizer state are restored afterwards (Adam's
        step counters included; a state loaded from a checkpoint is put back, a fresh one zeroed), so the replayed trajectory
        is the eager one.  The warm-up exchanges nothing (its updates are discarded on every rank alike)."""
        self._static_in = inputs.clone()
        saved_model = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        # (by parameter and key, not by tensor identity: FlowAdam re-homes the step counters into its own vector at its first step)
        saved_opt = {(p, k): torch.as_tensor(v).detach().clone() for p, st in self.optimizer.state.items() for k, v in st.items()
                     if torch.is_tensor(v) or k == 'step'}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # warm-up off the capture
            for it in range(3):
                self._eager_fwd_bwd(self._static_in)
                if self.distributed:
                    # the capture's Adam graph reads the flat buffer's views (never None, zero where backward reached nothing); the
                    # warm-up must see the same gradients, or a parameter without one sends FlowAdam to torch's Adam here and to its
                    # own -- still stateless -- kernel inside the capture
                    self.grads.pack_all()
                self.optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for k, v in self.model.state_dict().items():
                v.copy_(saved_model[k])
            for p, st in self.optimizer.state.items():   # in place: the graphs keep these tensors
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if (p, k) in saved_opt:
                            v.copy_(saved_opt[(p, k)])
                        else:
                            v.zero_()
        if hasattr(self.optimizer, 'prepare'):
            self.optimizer.prepare()                   # state / step vector / device tables of ALL parameters exist before any capture
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            loss, pack = self._eager_fwd_bwd(self._static_in)
            if self.distributed:
                self.grads.pack_all()                  # the replay leaves the gradient in the flat buffer; p.grad -> its views
            else:
                self.optimizer.step()
            self._static_loss = loss.detach()
            self._static_pack = {k: v.detach() for k, v in pack.items()}
        self._graph_opt = None
        if self.distributed:
            self._graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_opt):
                self.optimizer.step()

    def _graph_step(self, inputs):
        if self._graph is None:
            self._build_graph(inputs)
        self._static_in.copy_(inputs)
        self._graph.replay()
        if self.distributed:
            self.grads.all_reduce_flat()
            self._graph_opt.replay()
        self.iteration += 1
        return self._static_loss, self._static_pack

    def _settle_host(self):
        global _PROCESS_GC_FROZEN
        if self.gc_freeze_after and not self._gc_frozen and self._steps_here >= self.gc_freeze_after:
            self._gc_frozen = True
            if not _PROCESS_GC_FROZEN:                 # once per process, whichever trainer gets there first
                import gc
                gc.collect()
                gc.freeze()
                _PROCESS_GC_FROZEN = True
        self._steps_here += 1

    def close(self):
        """Undo the process-global side effect of ``gc_freeze_after`` (the frozen objects become collectable again)."""
        global _PROCESS_GC_FROZEN
        if self._gc_frozen and _PROCESS_GC_FROZEN:
            import gc
            gc.unfreeze()
            _PROCESS_GC_FROZEN = False
        self._gc_frozen = False

    def step(self, inputs):
        """One optimisation step on this rank's shard.  Returns (loss, loss_pack) (detached)."""
        self.model.train()
        self._settle_host()
        if self.use_graph and (self._graph is None or inputs.shape == self._static_in.shape):
            return self._graph_step(inputs)          # (a ragged last batch of an epoch falls through to the eager step)
        self.grads.zero()
        loss_pack = self.model(inputs)
        loss = self.total_loss(loss_pack)
        self._backward(loss)
        if self.distributed:
            self.grads.all_reduce_mean()
        self.optimizer.step()
        self.iteration += 1
        return loss.detach(), {k: v.detach() for k, v in loss_pack.items()}

    # ---- checkpoint format of train.py:23-31 (keys unwrapped, so 1-GPU and N-GPU files interchange)
    def state(self):
        return {'iteration': self.iteration, 'model_state_dict': self.model.state_dict(),
                'optimizer_state_dict': self.optimizer.state_dict()}

    def save(self, path):
        torch.save(self.state(), path)

    def load(self, path, map_location=None):
        data = torch.load(path, map_location=map_location)
        sd = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in data['model_state_dict'].items()}
        self.model.load_state_dict(sd)
        self.optimizer.load_state_dict(data['optimizer_state_dict'])
        relayout_optimizer_state(self.optimizer)
        self.iteration = data['iteration']
        return self.iteration


def relayout_optimizer_state(optimizer):
    """Give every per-parameter state tensor (Adam's exp_avg / exp_avg_sq / max_exp_avg_sq) its parameter's strides.
    ``Optimizer.load_state_dict`` keeps the strides the moments were SAVED with; a checkpoint written by the reference,
    by an NCHW run (--channels_last 0, --miopen_find 0, another MIOpen build) or before the conv weights went
    channels_last then leaves row-major moments next to channels_last parameters and gradients, and the fused
    (multi-tensor) Adam walks all four by memory offset -- it requires one layout.  Values are unchanged."""
    for group in optimizer.param_groups:
        for p in group['params']:
            st = optimizer.state.get(p)
            if not st:
                continue
            for k, v in st.items():
                if torch.is_tensor(v) and v.dim() == p.dim() and v.shape == p.shape and v.stride() != p.stride():
                    st[k] = torch.empty_like(p, dtype=v.dtype).copy_(v)      # empty_like preserves p's (dense) strides
