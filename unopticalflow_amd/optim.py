"""Adam for the flow step: ``torch.optim.Adam`` (the reference's optimizer, train.py:39: default betas / eps, lr from the
command line) whose ``step`` is ONE HIP launch over all parameter tensors (csrc/optim.hip, ``unflow_adam_multi``).

Same class hierarchy, same ``state`` layout and ``state_dict`` as torch's capturable Adam (``step`` a float32 device scalar,
``exp_avg``, ``exp_avg_sq`` per parameter), so checkpoints interchange with ``torch.optim.Adam`` and with the reference's
``optimizer_state_dict`` (train.py:23-31).  The step counters are views of one device vector that the kernel advances, so the
optimizer can sit inside a captured hipGraph.  Anything the kernel does not take -- a CPU parameter, a parameter without a gradient,
a gradient whose layout differs from its parameter's, amsgrad / weight decay / maximize -- goes through ``torch.optim.Adam.step``.
"""
import ctypes

import torch

from . import _lib, ops


class _Slot(ctypes.Structure):
    _fields_ = [('p', ctypes.c_void_p), ('m', ctypes.c_void_p), ('v', ctypes.c_void_p), ('numel', ctypes.c_longlong)]


def _dense_like(a, b):
    return a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype and a.device == b.device


def _dense(t):
    """Every element of the storage span exactly once (row-major, channels_last or any permutation of them): the kernel walks
    numel() consecutive floats from data_ptr()."""
    n = 1
    for size, stride in sorted(zip(t.shape, t.stride()), key=lambda ss: ss[1]):
        if size == 1:
            continue
        if stride != n:
            return False
        n *= size
    return True


class FlowAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, capturable=True)
        self._tables = None                # (device slots, device chunk map, nchunks, key) of the current state tensors
        self._steps = None                 # one float per parameter; state[p]['step'] are its 0-dim views
        self.native_steps = 0              # steps taken by the HIP kernel (tests / diagnostics)

    # ---- state
    def _params(self):
        return [p for g in self.param_groups for p in g['params']]

    def _init_state(self, params):
        dev = params[0].device
        if self._steps is None or self._steps.numel() != len(params) or self._steps.device != dev:
            self._steps = torch.zeros(len(params), dtype=torch.float32, device=dev)
        for i, p in enumerate(params):
            st = self.state[p]
            if len(st) == 0:
                st['step'] = self._steps[i]
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif not torch.is_tensor(st['step']) or st['step'].data_ptr() != self._steps[i].data_ptr():
                # counters that came from a checkpoint (fresh tensors; plain ints in a torch-1.2 file of the reference) or from
                # torch's own step: tie them to the vector the kernel advances
                self._steps[i].copy_(torch.as_tensor(st['step'], dtype=torch.float32))
                st['step'] = self._steps[i]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = None                # new moment tensors: rebuild the device table at the next step

    def _build_tables(self, params):
        key = tuple((p.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()) for p in params)
        if self._tables is not None and self._tables[3] == key:
            return self._tables
        chunk = _lib.load().unflow_adam_chunk()
        slots = (_Slot * len(params))()
        cmap = []
        for i, p in enumerate(params):
            st = self.state[p]
            slots[i] = _Slot(p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel())
            cmap += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        dev = params[0].device
        raw = torch.frombuffer(bytearray(bytes(slots)), dtype=torch.uint8).to(dev)
        cm = torch.tensor(cmap, dtype=torch.int32).reshape(-1).to(dev)
        self._tables = (raw, cm, len(cmap), key)
        return self._tables

    def prepare(self):
        """Create everything a native step needs -- moments, the step vector, the device tables -- for ALL parameters now, whether
        or not they have a gradient yet.  A trainer calls it before it captures ``step()`` into a hipGraph: allocations, zero-fills
        and host-to-device copies made DURING a capture would become graph nodes (every replay would reset the moments and the
        step counters) or fail the capture outright."""
        params = self._params()
        if not params or not all(ops.on_device(p) and p.dtype == torch.float32 for p in params) or len(params) > 128:
            return False
        self._init_state(params)
        self._build_tables(params)
        return True

    def _capture_ready(self, params):
        if self._steps is None or self._steps.numel() != len(params) or self._tables is None:
            return False
        for i, p in enumerate(params):
            st = self.state.get(p)
            if not st or not torch.is_tensor(st.get('step')) or st['step'].data_ptr() != self._steps[i].data_ptr():
                return False
        return self._tables[3] == tuple((p.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()) for p in params)

    def _native_ok(self, params):
        if not params or len(params) > 128:
            return False
        g = self.param_groups
        if len(g) != 1 or g[0].get('amsgrad') or g[0].get('weight_decay') or g[0].get('maximize') or g[0].get('differentiable'):
            return False
        if torch.is_tensor(g[0]['lr']):
            return False
        dev = params[0].device
        for p in params:
            if not ops.on_device(p) or p.device != dev or p.dtype != torch.float32 or p.grad is None or p.grad.is_sparse:
                return False
            if not _dense_like(p.grad, p) or not _dense(p):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        params = self._params()
        capturing = bool(params) and params[0].is_cuda and torch.cuda.is_current_stream_capturing()
        if closure is not None or not self._native_ok(params):
            if capturing and self.native_steps > 0:
                # torch's Adam under capture next to native steps outside it would walk two sets of step counters (a model with a
                # parameter the loss never reaches takes torch's Adam in the warm-up AND in the capture: consistent, allowed)
                raise RuntimeError('FlowAdam.step() inside a hipGraph capture would fall back to torch.optim.Adam (a parameter without a '
                                   'gradient, or a layout the kernel does not take) after native steps: give the warm-up steps the '
                                   'gradient view of the capture')
            if self._steps is not None:
                self._tables = None
            return super().step(closure)
        if capturing and not self._capture_ready(params):
            raise RuntimeError('FlowAdam.step() inside a hipGraph capture would have to allocate its state (moments, step vector, device '
                               'tables): call FlowAdam.prepare() -- or take one step with the same gradients -- before the capture')
        self._init_state(params)
        for p in params:                   # the moments walk the parameter's memory: same dense layout (trainer.relayout_optimizer_state)
            st = self.state[p]
            if not (_dense_like(st['exp_avg'], p) and _dense_like(st['exp_avg_sq'], p)):
                return super().step(closure)
        slots, cmap, nchunks, _ = self._build_tables(params)
        grads = (ctypes.c_void_p * len(params))(*[p.grad.data_ptr() for p in params])
        g = self.param_groups[0]
        lib = _lib.load()
        with ops._on(params[0].device):
            rc = lib.unflow_adam_multi(ctypes.c_void_p(slots.data_ptr()), ctypes.c_void_p(cmap.data_ptr()), nchunks, grads, len(params),
                                       ctypes.c_void_p(self._steps.data_ptr()),
                                       ctypes.c_float(g['lr']), ctypes.c_float(g['betas'][0]), ctypes.c_float(g['betas'][1]),
                                       ctypes.c_float(g['eps']), ops._stream())
        _lib.check(rc, 'unflow_adam_multi')
        self.native_steps += 1
        return None
