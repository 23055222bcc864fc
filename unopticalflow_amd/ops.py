"""Autograd operators over libunflow_hip.so.

Each operator replaces one eager op chain of the reference (file:line in the docstrings) with
the hand-written gfx950 kernels of ``csrc/``.  Tensors must live on a HIP device: there is no
CPU path here (the CPU restatement used for parity lives in ``oracle/`` and is test-only).
PyTorch supplies device memory (caching allocator) and the current stream; the kernels are
called through the plain-C ABI of ``include/unflow_hip.h`` with raw pointers.
"""
import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ['corr', 'warp_flow', 'warp_flow_masked', 'warp_corr', 'occ_weight', 'occ_weight_stacked', 'masked_mean', 'ssim_loss',
           'ssim_map', 'smooth2_loss', 'consis_loss', 'multiscale_losses', 'multiscale_supported', 'warp_flow_masked_pyramid', 'bias_leaky_relu_', 'bias_leaky_relu_into', 'upsample_bilinear_scaled', 'loss_combine', 'weighted_mean_sum', 'flow_head', 'img_pyramid']


def on_device(t):
    """True for a tensor the kernels can take.  The model asks THIS (not ``t.is_cuda``) before it chooses a path that only exists as a
    HIP kernel -- channels_last epilogues, fused up-sampling, deferred sums, one launch over the scales -- so that tests/hostexec.py can
    route the same paths to the host-executed build of the kernel sources (it patches this function inside its ``with`` block)."""
    return t.is_cuda


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def _on(dev):
    """Device guard for a launch: nothing to do (and nothing to pay: torch.cuda.device() costs ~10 us of host time per
    entry, ~150 times per step) when the tensors already live on the current device, which is the one-process-per-GPU case."""
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def _dev(*tensors):
    """All operands must be fp32 tensors on one HIP device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('unopticalflow_amd ops run on an MI355X (HIP) device only; got a %s tensor. '
                               'There is no CPU fallback.' % t.device)
        if t.dtype != torch.float32:
            raise TypeError('unopticalflow_amd ops compute in fp32; got %s' % t.dtype)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError('operands on different devices: %s vs %s' % (dev, t.device))
    return dev


def _call(name, *args, nbytes=0, shape=None, label=None):
    """Call one C entry point.  ``nbytes`` = algorithmic HBM bytes of the launch (include/unflow_hip.h,
    DESIGN.md section 3); only used when bench.py's kernel timing is on (``label``: the row the launch is timed under, default the name)."""
    lib = _lib.load()
    if kernel_timer.names is not None and (kernel_timer.names is True or (label or name) in kernel_timer.names):
        slot = lib.unflow_timing_begin()
        if slot < 0:
            raise RuntimeError('unflow_timing_begin failed (%d)' % slot)
        try:
            _lib.check(getattr(lib, name)(*args), name)
        finally:
            lib.unflow_timing_end()
        kernel_timer.table.setdefault((label or name, shape), []).append((slot, nbytes))
        return
    _lib.check(getattr(lib, name)(*args), name)


class _KernelTimer:
    """Kernel-exact timing of C entry points (bench.py's roofline legs): the library launches the kernels of a timed call
    with a HIP event pair attached (``unflow_timing_begin`` / hipExtLaunchKernelGGL), so a row is first-kernel-begin to
    last-kernel-end on the stream the kernels run on -- what a rocprofv3 kernel trace reports, without the marker
    packets of a hipEventRecord bracket.  Disabled by default: no events, no overhead.  ``enable(names)`` times every
    call of the named entry points (``True``: all of them) until ``disable()``; ``rows()`` groups the launches by
    (entry point, shape).  Not under hipGraph capture."""

    def __init__(self):
        self.names = None
        self.table = {}

    def enable(self, names=True, reserve=0):
        """``reserve``: event pairs to create now, so that arming a slot inside the timed loop costs no hipEventCreate."""
        self.names = names if names is True else frozenset(names)
        self.table = {}
        if reserve:
            _lib.load().unflow_timing_reserve(int(reserve))

    def disable(self):
        self.names = None

    def rows(self):
        """-> list of dicts (entry point, shape, launches, mean us, total us, algorithmic bytes, GB/s), heaviest first;
        call after a device synchronize.  Frees the library's event pairs."""
        lib = _lib.load()
        rows = []
        us = ctypes.c_float()
        for (name, shape), v in self.table.items():
            tot, n, nb = 0.0, 0, 0
            for slot, nbytes in v:
                if lib.unflow_timing_elapsed_us(slot, ctypes.byref(us)) == 0:        # (a call that launched nothing has no time)
                    tot += us.value; n += 1; nb += nbytes
            if n == 0:
                continue
            rows.append({'entry': name, 'shape': list(shape) if shape else None, 'launches': n,
                         'avg_us': round(tot / n, 2), 'total_us': tot, 'total_bytes': nb,
                         'algorithmic_GBps': round(nb / (tot * 1e-6) / 1e9, 1) if tot > 0 and nb else None})
        rows.sort(key=lambda r: -r['total_us'])
        self.table = {}
        lib.unflow_timing_reset()
        return rows


kernel_timer = _KernelTimer()


class _DeferredBiasGrads:
    """The second stage of every bias-gradient reduction of a backward pass as ONE launch at the end of the pass.

    The conv() epilogues (net_utils.py:7-11) and the predict_flow heads (pwc_tf.py:93-94) reduce their bias gradient in two
    stages: per-workgroup partial sums inside the backward kernel, then a tiny launch that adds them in a fixed order -- 49
    such launches per backward pass of the flow network, ~5 us each (218 us of a 24 ms step, profiles/r3).  Inside
    ``with ops.deferred_bias_grads:`` the backward nodes stop after the first stage (C ABI: gbias == NULL), hand autograd the
    still unwritten ``gbias`` tensor and register the job; the autograd engine runs ``flush`` as a final callback of the pass
    (on the stream ``backward()`` was called on, before it returns), where one ``unflow_bias_grad_finalize_batch`` launch
    finishes all of them with the same summation order, i.e. the same bits.

    OPT-IN, and only for a caller that knows two things about its backward pass (FlowTrainer does, a library user calling
    ``loss.backward()`` gets the immediate second stage): (1) every bias gradient is ASSIGNED, not accumulated -- ``p.grad`` is
    None when the pass starts, so AccumulateGrad adopts the tensor it is handed instead of reading it; only the tensor's STORAGE
    is kept here (a second reference to the tensor itself would make AccumulateGrad clone it -- still unwritten); (2) anything
    that READS a bias gradient before the pass ends calls ``flush()`` first (the eager data-parallel step packs all-reduce
    pieces from hooks: ``FlatGradients.before_pack``).  ``adopted(params)`` checks (1) after a pass.
    """

    def __init__(self):
        self.enabled = False
        self.jobs = []                 # (partials tensor, gbias storage, gbias address, n, C, mode)
        self.queued = False
        self.last_addresses = ()       # gbias addresses of the last flush (adopted())
        self.last_storages = []        # ... and their storages, kept until the next pass: a freed address could be handed to a later gradient and pass adopted() by accident
        import threading
        self.lock = threading.Lock()
        self._depth = 0

    def __enter__(self):
        self._depth += 1
        self.enabled = True
        return self

    def __exit__(self, *exc):
        self._depth -= 1
        if self._depth == 0:
            self.enabled = False
            self.flush()               # (normally empty: the engine's final callback ran inside the pass)
        return False

    def add(self, partials, gbias, n, C, mode):
        with self.lock:
            self.jobs.append((partials, gbias.untyped_storage(), gbias.data_ptr(), gbias.device, int(n), int(C), int(mode)))
            if not self.queued:
                self.queued = True
                self.last_addresses = ()
                self.last_storages = []
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self):
        with self.lock:
            self.queued = False
        self.flush()

    def flush(self):
        with self.lock:
            jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        self.last_addresses = tuple(self.last_addresses) + tuple(j[2] for j in jobs)
        self.last_storages.extend(j[1] for j in jobs)          # (a few KB: bias gradients)
        by_dev = {}
        for j in jobs:
            by_dev.setdefault(j[3], []).append(j)
        for dev, js in by_dev.items():
            n = len(js)
            P = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in js])
            G = (ctypes.c_void_p * n)(*[j[2] for j in js])
            N_ = (ctypes.c_int * n)(*[j[4] for j in js])
            C_ = (ctypes.c_int * n)(*[j[5] for j in js])
            M_ = (ctypes.c_int * n)(*[j[6] for j in js])
            with _on(dev):
                _call('unflow_bias_grad_finalize_batch', P, G, N_, C_, M_, n, _stream(),
                      nbytes=4 * sum(j[4] * j[5] + j[5] for j in js), shape=(n,))

    def adopted(self, params, extra=()):
        """True when every bias gradient finished by the last pass IS some parameter's ``.grad`` (autograd adopted the tensors it
        was handed: nothing was cloned or accumulated before the deferred launch wrote them).  ``extra``: addresses of gradients
        that were already copied elsewhere and re-pointed (the data-parallel trainer packs all-reduce pieces during the pass:
        ``FlatGradients.seen_sources``, which keeps those tensors alive until the check).  Addresses are compared, so both sides hold their
        memory until then: the finished gradients' storages stay referenced here (``last_storages``) -- an address freed in between could be
        handed to a later gradient and match by accident."""
        have = {p.grad.data_ptr() for p in params if p.grad is not None}
        have.update(extra)
        return all(a in have for a in self.last_addresses)


deferred_bias_grads = _DeferredBiasGrads()


class _DeferredLossSums:
    """The second stage of the loss reductions of a forward pass as ONE launch.

    Masked mean, SSIM loss, smoothness and consistency (model_flow_paper.py:90-99, 137-195) each end in a per-sample reduction of two
    stages -- partial sums per workgroup, then one workgroup per sample adds them in a fixed order: 12 such second stages per train
    step (4 losses x 3 scales), ~4-10 us each for a few KB.  Inside ``with ops.deferred_loss_sums:`` the forward ops stop after the
    first stage (C ABI: loss == NULL) and register the job; ``flush()`` finishes all of them with one
    ``unflow_loss_finalize_batch`` launch -- same summation order, same divisions, same bits.  ``loss_combine`` and
    ``weighted_mean_sum`` (the only readers of the per-scale losses in Model_flow.forward) flush first, leaving the block flushes too.
    OPT-IN: a caller that reads a loss tensor inside the block by other means must call ``flush()`` itself."""

    def __init__(self):
        self.enabled = False
        self.jobs = []                 # (partials, loss, sums or None, nblk, B, kind, n0, n1, device): tensors kept alive until the flush
        self._depth = 0
        self.launches = 0

    def __enter__(self):
        self._depth += 1
        self.enabled = True
        return self

    def __exit__(self, *exc):
        self._depth -= 1
        if self._depth == 0:
            self.enabled = False
            self.flush()
        return False

    def add(self, partials, loss, sums, op, H, W, kind, n0, n1, aligned=True):
        B = loss.shape[0]
        nblk = _lib.load().unflow_loss_partial_blocks(op, H, W, B, 1 if aligned else 0)
        if nblk <= 0:
            raise RuntimeError('unflow_loss_partial_blocks(%d, %d, %d) failed' % (op, H, W))
        self.jobs.append((partials, loss, sums, nblk, B, kind, float(n0), float(n1), loss.device))

    def flush(self):
        jobs, self.jobs = self.jobs, []
        by_dev = {}
        for j in jobs:
            by_dev.setdefault(j[8], []).append(j)
        for dev, js in by_dev.items():
            n = len(js)
            P = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in js])
            L = (ctypes.c_void_p * n)(*[j[1].data_ptr() for j in js])
            S = (ctypes.c_void_p * n)(*[None if j[2] is None else j[2].data_ptr() for j in js])
            N_ = (ctypes.c_int * n)(*[j[3] for j in js])
            B_ = (ctypes.c_int * n)(*[j[4] for j in js])
            K_ = (ctypes.c_int * n)(*[j[5] for j in js])
            A_ = (ctypes.c_float * n)(*[j[6] for j in js])
            C_ = (ctypes.c_float * n)(*[j[7] for j in js])
            with _on(dev):
                _call('unflow_loss_finalize_batch', P, L, S, N_, B_, K_, A_, C_, n, _stream(),
                      nbytes=4 * sum(j[3] * j[4] * 2 + 3 * j[4] for j in js), shape=(n,))
            self.launches += 1


deferred_loss_sums = _DeferredLossSums()


def _f32(x):
    """x rounded to fp32 (the second-stage kernels take their divisors as C floats computed in fp32)."""
    import struct
    return struct.unpack('f', struct.pack('f', x))[0]


def _finish_bias_grad(partials, gbias, n, C, mode):
    """-> the gbias pointer to hand to a backward entry point: NULL (and the job registered) when the reduction is deferred."""
    if deferred_bias_grads.enabled and _in_backward():
        deferred_bias_grads.add(partials, gbias, n, C, mode)
        return ctypes.c_void_p(0)
    return _ptr(gbias)


def _in_backward():
    """True while the autograd engine is executing a backward pass on this thread (a final callback can be queued)."""
    try:
        return torch._C._current_graph_task_id() != -1
    except AttributeError:                              # (an older torch: no way to tell, do not defer)
        return False


def _partials(B, H, W, dev):
    n = _lib.load().unflow_partials_per_sample(H, W)
    return torch.empty(B * n, dtype=torch.float32, device=dev)


# ------------------------------------------------------------------------------------------
# cost volume
# ------------------------------------------------------------------------------------------
# arithmetic of the cost-volume backward, chosen PER CALL (include/unflow_hip.h UNFLOW_CORR_BWD_*; the library keeps no mode): 'auto' = per
# shape, the fastest kernel that has passed a complete GPU parity run; 'fp32' = the fp32 FMA kernels everywhere; 'mfma' = banded bf16 hi/lo
# split products on the matrix cores (fp32 accumulation; ~4e-6 of the largest gradient away from the fp32 sums, deterministic) wherever the
# shape is served; 'fp32_next' = the fp32 kernels incl. the round-6 ones no GPU has run yet (small maps: csrc/corr_small_rows.h)
CORR_BACKWARD_MODES = {'auto': 0, 'fp32': 1, 'mfma': 2, 'fp32_next': 3}


class _Corr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, d, arithmetic=0):
        _dev(f1, f2)
        f1, f2 = f1.contiguous(), f2.contiguous()
        B, C, H, W = f1.shape
        cv = torch.empty((B, (2 * d + 1) ** 2, H, W), dtype=f1.dtype, device=f1.device)
        with _on(f1.device):
            _call('unflow_corr_fwd', _ptr(f1), _ptr(f2), _ptr(cv), B, C, H, W, d, _stream(),      # algorithmic bytes: read
                  nbytes=4 * B * H * W * (2 * C + (2 * d + 1) ** 2), shape=(B, C, H, W))          # f1, f2 once, write cv once
        ctx.save_for_backward(f1, f2)
        ctx.d = d
        ctx.arithmetic = arithmetic
        return cv

    @staticmethod
    def backward(ctx, g):
        f1, f2 = ctx.saved_tensors
        B, C, H, W = f1.shape
        g = g.contiguous()
        gf1, gf2 = torch.empty_like(f1), torch.empty_like(f2)
        with _on(f1.device):
            if ctx.arithmetic == 0:
                _call('unflow_corr_bwd', _ptr(f1), _ptr(f2), _ptr(g), _ptr(gf1), _ptr(gf2), B, C, H, W, ctx.d,
                      _stream(), nbytes=4 * B * H * W * (4 * C + (2 * ctx.d + 1) ** 2), shape=(B, C, H, W))
            else:
                _call('unflow_corr_bwd_ex', _ptr(f1), _ptr(f2), _ptr(g), _ptr(gf1), _ptr(gf2), B, C, H, W, ctx.d, ctx.arithmetic,
                      _stream(), nbytes=4 * B * H * W * (4 * C + (2 * ctx.d + 1) ** 2), shape=(B, C, H, W), label='unflow_corr_bwd')
        return gf1, gf2, None, None


def corr(input1, input2, d=4, backward='auto'):
    """Cost volume, PWC_tf.corr_naive (pwc_tf.py:97-106): [B,C,H,W] x2 -> [B,(2d+1)^2,H,W].  ``backward``: the arithmetic of THIS
    call's backward pass (CORR_BACKWARD_MODES; ValueError for an unknown name)."""
    assert (input1.shape == input2.shape)            # pwc_tf.py:99
    if backward not in CORR_BACKWARD_MODES:
        raise ValueError('backward must be one of %s, got %r' % (sorted(CORR_BACKWARD_MODES), backward))
    return _Corr.apply(input1, input2, int(d), CORR_BACKWARD_MODES[backward])


# ------------------------------------------------------------------------------------------
# warp
# ------------------------------------------------------------------------------------------
# the feature-map warps' backward as one gather pass (unflow_warp_bwd_fused) where the shape allows; False: zero-fill + scatter (A/B: bench.py --fused-warp-bwd)
fused_warp_bwd = True


class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow, use_mask, align_corners, deterministic=False, fused=None):
        _dev(x, flow)
        x, flow = x.contiguous(), flow.contiguous()
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device=x.device) if use_mask else None
        # the one-pass backward needs the per-tile displacement table: the forward's tap set-up leaves it behind for free
        lib = _lib.load()
        want = fused if fused is not None else fused_warp_bwd
        table = None
        # (ctx.needs_input_grad, not torch.is_grad_enabled(): grad mode is OFF inside an autograd.Function's forward)
        if want and not use_mask and ctx.needs_input_grad[0] and lib.unflow_warp_bwd_fused_supported(B, C, H, W) == 2:
            table = torch.empty(lib.unflow_warp_bwd_table_bytes(B, C, H, W), dtype=torch.uint8, device=x.device)
        with _on(x.device):
            if table is not None:
                _call('unflow_warp_fwd_table', _ptr(x), _ptr(flow), _ptr(out), _ptr(table), B, C, H, W, int(align_corners), _stream(),
                      nbytes=B * H * W * (8 * C + 8), shape=(B, C, H, W))
            else:
                _call('unflow_warp_fwd', _ptr(x), _ptr(flow), _ptr(out), _ptr(mask), B, C, H, W,
                      int(align_corners), _stream(), nbytes=B * H * W * (8 * C + 8 + (1 if use_mask else 0)),
                      shape=(B, C, H, W))
        ctx.table = table
        ctx.save_for_backward(x, flow, mask)
        ctx.set_materialize_grads(False)             # (no zero-filled "gradient" for the mask: a fill launch per warp otherwise)
        ctx.ac = int(align_corners)
        ctx.entry = 'unflow_warp_bwd_det' if deterministic else 'unflow_warp_bwd'
        ctx.fused = fused                            # None: by shape (ops.fused_warp_bwd); True / False: forced
        if use_mask:
            ctx.mark_non_differentiable(mask)
            return out, mask
        return out, None

    @staticmethod
    def backward(ctx, g, _gmask):
        x, flow, mask = ctx.saved_tensors
        B, C, H, W = x.shape
        if g is None:
            return None, None, None, None, None, None
        g = g.contiguous()
        gsrc = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gflow = torch.empty_like(flow)
        want = ctx.fused if ctx.fused is not None else fused_warp_bwd
        if want and mask is None and gsrc is not None and _lib.load().unflow_warp_bwd_fused_supported(B, C, H, W) >= (1 if ctx.fused else 2):
            # one pass: source gradient as a gather + flow gradient, no zero-fill, no atomics, bitwise reproducible
            table, ready = ctx.table, 1
            if table is None:
                table, ready = torch.empty(_lib.load().unflow_warp_bwd_table_bytes(B, C, H, W), dtype=torch.uint8, device=x.device), 0
            with _on(x.device):
                _call('unflow_warp_bwd_fused', _ptr(x), _ptr(flow), _ptr(g), _ptr(gsrc), _ptr(gflow), _ptr(table), ready, B, C, H, W, ctx.ac, _stream(),
                      nbytes=4 * B * H * W * (3 * C + 4), shape=(B, C, H, W))
            return gsrc, (gflow if ctx.needs_input_grad[1] else None), None, None, None, None
        with _on(x.device):
            _call(ctx.entry, _ptr(x), _ptr(flow), _ptr(g), _ptr(mask), _ptr(gsrc), _ptr(gflow),
                  B, C, H, W, ctx.ac, _stream(),
                  nbytes=4 * B * H * W * ((3 * C + 4) if gsrc is not None else (2 * C + 4)), shape=(B, C, H, W))
        return gsrc, (gflow if ctx.needs_input_grad[1] else None), None, None, None, None


def _check_flow_shape(x, flow):
    B, C, H, W = x.size()
    if tuple(flow.shape) != (B, 2, H, W):            # net_utils.py:35-36
        raise ValueError('the shape of grid {0} is not equal to the shape of flow {1}.'.format(
            torch.Size((B, 2, H, W)), flow.shape))


def warp_flow(x, flow, use_mask=False, align_corners=False, deterministic=False, fused_backward=None):
    """warp_flow (net_utils.py:16-54): backward-warp x [B,C,H,W] by flow [B,2,H,W].  The backward of a feature-map warp whose tiles
    fill the chip (levels 2 / 3 of the 832x256 step) is ONE gather pass (unflow_warp_bwd_fused: no zero-fill, no float atomics, bitwise
    reproducible); other shapes zero-fill and scatter.  ``fused_backward``: True / False force / forbid the gather pass (None: by
    shape); ``deterministic``: the older two-launch gather form (unflow_warp_bwd_det) for the shapes the fused pass does not take."""
    _check_flow_shape(x, flow)
    out, _ = _Warp.apply(x, flow, bool(use_mask), bool(align_corners), bool(deterministic), fused_backward)
    return out


def warp_flow_masked(x, flow, align_corners=False):
    """warp_flow(..., use_mask=True) that also returns the binary uint8 mask of net_utils.py:47-51."""
    _check_flow_shape(x, flow)
    return _Warp.apply(x, flow, True, bool(align_corners))


# ------------------------------------------------------------------------------------------
# fused warp + cost volume
# ------------------------------------------------------------------------------------------
class _WarpCorr(torch.autograd.Function):
    """cv = corr(f1, warp(f2, flow)) with the warped map living only in LDS (csrc/warp_corr.hip).  Nothing but the
    inputs is saved for the backward pass, which recomputes what it needs."""

    @staticmethod
    def forward(ctx, f1, f2, flow, d, align_corners):
        _dev(f1, f2, flow)
        f1, f2, flow = f1.contiguous(), f2.contiguous(), flow.contiguous()
        B, C, H, W = f1.shape
        cv = torch.empty((B, (2 * d + 1) ** 2, H, W), dtype=f1.dtype, device=f1.device)
        with _on(f1.device):
            # algorithmic bytes: those of the two ops it replaces: warp 4*n*(2C+2) + corr 4*n*(2C+D^2)
            _call('unflow_warp_corr_fwd', _ptr(f1), _ptr(f2), _ptr(flow), _ptr(cv), B, C, H, W, d, int(align_corners),
                  _stream(), nbytes=4 * B * H * W * (4 * C + 2 + (2 * d + 1) ** 2), shape=(B, C, H, W))
        ctx.save_for_backward(f1, f2, flow)
        ctx.d, ctx.ac = d, int(align_corners)
        return cv

    @staticmethod
    def backward(ctx, g):
        f1, f2, flow = ctx.saved_tensors
        B, C, H, W = f1.shape
        g = g.contiguous()
        gf1 = torch.empty_like(f1)
        gf2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        gflow = torch.empty_like(flow)
        scratch = torch.empty((2,) + tuple(f2.shape), dtype=f2.dtype, device=f2.device)     # warped map | its gradient: dies here
        D2 = (2 * ctx.d + 1) ** 2
        with _on(f1.device):
            _call('unflow_warp_corr_bwd', _ptr(f1), _ptr(f2), _ptr(flow), _ptr(g), _ptr(gf1), _ptr(gf2), _ptr(gflow),
                  _ptr(scratch), B, C, H, W, ctx.d, ctx.ac, _stream(),
                  nbytes=4 * B * H * W * ((4 * C + D2) + (3 * C + 4)), shape=(B, C, H, W))
        return gf1, gf2, (gflow if ctx.needs_input_grad[2] else None), None, None


def warp_corr_supported(f1, d):
    B, C, H, W = f1.shape
    return bool(_lib.load().unflow_warp_corr_supported(C, H, W, int(d)))


def warp_corr(input1, input2, flow, d=4, align_corners=False):
    """One decoder level's `corr(feat1, warp(feat2, flow))` (pwc_tf.py:121-122 and the three level blocks below it) as ONE
    kernel; shapes the fused kernel does not cover run the two separate operators."""
    assert (input1.shape == input2.shape)            # pwc_tf.py:99
    _check_flow_shape(input2, flow)
    if not warp_corr_supported(input1, d):
        return corr(input1, warp_flow(input2, flow, use_mask=False, align_corners=align_corners), d)
    return _WarpCorr.apply(input1, input2, flow, int(d), bool(align_corners))


# ------------------------------------------------------------------------------------------
# occlusion weights + photometric terms
# ------------------------------------------------------------------------------------------
class _OccWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, from_l, from_r):
        dev = _dev(img, from_l, from_r)
        img, from_l, from_r = img.contiguous(), from_l.contiguous(), from_r.contiguous()
        B, C, H, W = img.shape
        assert C == 3
        f = lambda: torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        u = lambda: torch.empty((B, 1, H, W), dtype=torch.uint8, device=dev)
        diff_l, diff_r, w_bwd, w_fwd, v_bwd, v_fwd = f(), f(), f(), f(), u(), u()
        with _on(dev):
            _call('unflow_occ_weight_fwd', _ptr(img), _ptr(from_l), _ptr(from_r), _ptr(diff_l), _ptr(diff_r),
                  _ptr(w_bwd), _ptr(w_fwd), _ptr(v_bwd), _ptr(v_fwd), B, H, W, _stream(),
                  nbytes=B * H * W * (4 * 13 + 2), shape=(B, 3, H, W))
        ctx.save_for_backward(img, from_l, from_r)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(w_bwd, w_fwd, v_bwd, v_fwd)     # weight is .data (:122)
        return diff_l, diff_r, w_bwd, w_fwd, v_bwd, v_fwd

    @staticmethod
    def backward(ctx, g_l, g_r, *_):
        img, from_l, from_r = ctx.saved_tensors
        B, C, H, W = img.shape
        out = [None, None, None]
        with _on(img.device):
            for k, (src, g) in enumerate(((from_l, g_l), (from_r, g_r))):
                if ctx.needs_input_grad[k + 1] and g is not None:
                    gs = torch.empty_like(src)
                    _call('unflow_absdiff_bwd', _ptr(img), _ptr(src), _ptr(g.contiguous()), _ptr(gs), B, H, W, B,
                          _stream())
                    out[k + 1] = gs
        if ctx.needs_input_grad[0]:
            raise RuntimeError('occ_weight: the centre image is a detached pyramid level '
                               '(model_flow_paper.py:58); no gradient is defined for it')
        return tuple(out)


class _OccWeight2(torch.autograd.Function):
    """Both warp directions of a scale stacked on the batch axis: warped = (from_l | from_r) [2B,3,H,W] against the B centre
    images -> diff = (diff_l | diff_r), weight = (w_bwd | w_fwd) [2B,1,H,W], so that every loss that follows runs as ONE
    2B launch per scale instead of one per direction."""

    @staticmethod
    def forward(ctx, img, warped):
        dev = _dev(img, warped)
        img, warped = img.contiguous(), warped.contiguous()
        B, C, H, W = img.shape
        assert C == 3 and warped.shape[0] == 2 * B
        diff = torch.empty((2 * B, 1, H, W), dtype=torch.float32, device=dev)
        wgt = torch.empty((2 * B, 1, H, W), dtype=torch.float32, device=dev)
        with _on(dev):
            _call('unflow_occ_weight_fwd', _ptr(img), _ptr(warped[:B]), _ptr(warped[B:]), _ptr(diff[:B]), _ptr(diff[B:]),
                  _ptr(wgt[:B]), _ptr(wgt[B:]), _ptr(None), _ptr(None), B, H, W, _stream(),
                  nbytes=B * H * W * 4 * 13, shape=(B, 3, H, W))
        ctx.save_for_backward(img, warped)
        ctx.set_materialize_grads(False)                             # (no zero-filled [2B,1,H,W] "gradient" for the weights)
        ctx.mark_non_differentiable(wgt)                             # weight is .data (model_flow_paper.py:122)
        return diff, wgt

    @staticmethod
    def backward(ctx, g, _gw):
        img, warped = ctx.saved_tensors
        B, C, H, W = img.shape
        if not ctx.needs_input_grad[1] or g is None:
            return None, None
        gs = torch.empty_like(warped)
        with _on(img.device):
            _call('unflow_absdiff_bwd', _ptr(img), _ptr(warped), _ptr(g.contiguous()), _ptr(gs), 2 * B, H, W, B, _stream(),
                  nbytes=4 * B * H * W * (3 + 2 * 7), shape=(2 * B, 3, H, W))
        return None, gs


def occ_weight_stacked(img, warped_lr):
    """compute_diff_weight (model_flow_paper.py:108-132) for one scale with (from_l | from_r) stacked on the batch axis:
    -> (diff [2B,1,H,W] = (diff_bwd | diff_fwd), weight [2B,1,H,W] = (weight_bwd | weight_fwd))."""
    return _OccWeight2.apply(img, warped_lr)


def occ_weight(img, img_from_l, img_from_r):
    """One scale of compute_diff_weight (model_flow_paper.py:108-132).

    Returns (diff_l, diff_r, weight_bwd, weight_fwd, valid_bwd_u8, valid_fwd_u8)."""
    return _OccWeight.apply(img, img_from_l, img_from_r)


class _MaskedMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, diff, w):
        dev = _dev(diff, w)
        diff, w = diff.contiguous(), w.contiguous()
        B, _, H, W = diff.shape
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        sums = torch.empty((B, 2), dtype=torch.float32, device=dev)
        part = _partials(B, H, W, dev)
        later = deferred_loss_sums.enabled
        with _on(dev):
            _call('unflow_masked_mean_fwd', _ptr(diff), _ptr(w), None if later else _ptr(loss), _ptr(sums),
                  _ptr(part), B, H, W, _stream(), nbytes=4 * B * H * W * 2, shape=(B, 1, H, W))       # reads diff, w
        if later:
            hw = _f32(float(H) * float(W))
            deferred_loss_sums.add(part, loss, sums, 0, H, W, 0, hw, hw)
        ctx.save_for_backward(w, sums)
        return loss

    @staticmethod
    def backward(ctx, gl):
        w, sums = ctx.saved_tensors
        B, _, H, W = w.shape
        gdiff = torch.empty_like(w)
        with _on(w.device):
            _call('unflow_masked_mean_bwd', _ptr(w), _ptr(sums), _ptr(gl.contiguous()), _ptr(gdiff), B, H, W,
                  _stream(), nbytes=4 * B * H * W * 2, shape=(B, 1, H, W))                                              # reads w, writes gdiff
        return gdiff, None


def masked_mean(diff, w):
    """One scale of compute_loss_with_mask (model_flow_paper.py:93-97): [B,1,H,W] x2 -> [B]."""
    return _MaskedMean.apply(diff, w.detach())


class _SsimLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, warped, w):
        dev = _dev(img, warped, w)
        img, warped, w = img.contiguous(), warped.contiguous(), w.contiguous()
        B, C, H, W = warped.shape                      # img may hold fewer samples: sample b pairs with image b % len(img)
        assert C == 3 and B % img.shape[0] == 0 and w.shape[0] == B
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        sums = torch.empty((B, 2), dtype=torch.float32, device=dev)
        part = _partials(B, H, W, dev)
        later = deferred_loss_sums.enabled
        with _on(dev):
            _call('unflow_ssim_loss_fwd', _ptr(img), _ptr(warped), _ptr(w), None if later else _ptr(loss), _ptr(sums),
                  _ptr(part), B, H, W, img.shape[0], _stream(), nbytes=4 * B * H * W * 7, shape=(B, 3, H, W))
        if later:
            hw = _f32(float(H) * float(W))
            aligned = ((img.data_ptr() | warped.data_ptr() | w.data_ptr()) & 7) == 0
            deferred_loss_sums.add(part, loss, sums, 1, H, W, 0, _f32(3.0 * hw), hw, aligned)
        ctx.save_for_backward(img, warped, w, sums)
        return loss

    @staticmethod
    def backward(ctx, gl):
        img, warped, w, sums = ctx.saved_tensors
        B, C, H, W = warped.shape
        gw = torch.empty_like(warped)
        with _on(img.device):
            _call('unflow_ssim_loss_bwd', _ptr(img), _ptr(warped), _ptr(w), _ptr(sums), _ptr(gl.contiguous()),
                  _ptr(gw), B, H, W, img.shape[0], _stream(), nbytes=4 * B * H * W * 10, shape=(B, 3, H, W))
        return None, gw, None


def ssim_loss(img, img_warped, w):
    """One scale of compute_loss_ssim (model_flow_paper.py:140-146) -> [B]; grad to img_warped."""
    return _SsimLoss.apply(img.detach(), img_warped, w.detach())


class _SsimMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        dev = _dev(x, y)
        x, y = x.contiguous(), y.contiguous()
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        with _on(dev):
            _call('unflow_ssim_map', _ptr(x), _ptr(y), _ptr(out), B, C, H, W, _stream())
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None
        scratch = torch.empty((4,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        with _on(x.device):
            _call('unflow_ssim_map_bwd', _ptr(x), _ptr(y), _ptr(g.contiguous()), _ptr(gx), _ptr(gy), _ptr(scratch),
                  B, C, H, W, _stream())
        return gx, gy


def ssim_map(x, y):
    """SSIM(x, y) map of pytorch_ssim/ssim.py:4-20, differentiable in both arguments like the reference's function
    (the train step itself uses the fused ``ssim_loss``)."""
    if x.shape != y.shape:
        raise ValueError('SSIM: shapes differ: {} vs {}'.format(tuple(x.shape), tuple(y.shape)))
    return _SsimMap.apply(x, y)


# ------------------------------------------------------------------------------------------
# flow regularisers
# ------------------------------------------------------------------------------------------
class _Smooth2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, img):
        dev = _dev(flow, img)
        flow, img = flow.contiguous(), img.contiguous()
        B, _, H, W = flow.shape                        # img may hold fewer samples: sample b pairs with image b % len(img)
        assert B % img.shape[0] == 0
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        part = _partials(B, H, W, dev)
        later = deferred_loss_sums.enabled
        with _on(dev):
            _call('unflow_smooth2_fwd', _ptr(flow), _ptr(img), None if later else _ptr(loss), _ptr(part),
                  B, H, W, img.shape[0], _stream(), nbytes=4 * H * W * (2 * B + 3 * img.shape[0]), shape=(B, 2, H, W))   # reads flow and the images once
        if later:
            nx = _f32(_f32(2.0 * float(H)) * float(W - 2))
            ny = _f32(_f32(2.0 * float(H - 2)) * float(W))
            deferred_loss_sums.add(part, loss, None, 2, H, W, 1, nx, ny)
        ctx.save_for_backward(flow, img)
        return loss

    @staticmethod
    def backward(ctx, gl):
        flow, img = ctx.saved_tensors
        B, _, H, W = flow.shape
        gflow = torch.empty_like(flow)
        with _on(flow.device):
            _call('unflow_smooth2_bwd', _ptr(flow), _ptr(img), _ptr(gl.contiguous()), _ptr(gflow), B, H, W,
                  img.shape[0], _stream(), nbytes=4 * H * W * (4 * B + 3 * img.shape[0]), shape=(B, 2, H, W))           # reads flow, images; writes gflow
        return gflow, None


def smooth2_loss(flow, img):
    """cal_grad2_error(flow/20, img) (model_flow_paper.py:152-167,174) -> [B]; flow is un-divided."""
    return _Smooth2.apply(flow, img.detach())


class _Consis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ff, fb, w):
        dev = _dev(ff, fb, w)
        ff, fb, w = ff.contiguous(), fb.contiguous(), w.contiguous()
        B, _, H, W = ff.shape
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        sums = torch.empty((B, 2), dtype=torch.float32, device=dev)
        part = _partials(B, H, W, dev)
        later = deferred_loss_sums.enabled
        with _on(dev):
            _call('unflow_consis_fwd', _ptr(ff), _ptr(fb), _ptr(w), None if later else _ptr(loss), _ptr(sums),
                  _ptr(part), B, H, W, _stream(), nbytes=4 * B * H * W * 5, shape=(B, 2, H, W))       # reads both flows and the weight
        if later:
            hw = _f32(float(H) * float(W))
            deferred_loss_sums.add(part, loss, sums, 3, H, W, 0, _f32(2.0 * hw), hw)
        ctx.save_for_backward(ff, fb, w, sums)
        return loss

    @staticmethod
    def backward(ctx, gl):
        ff, fb, w, sums = ctx.saved_tensors
        B, _, H, W = ff.shape
        g = torch.empty_like(ff)
        with _on(ff.device):
            _call('unflow_consis_bwd', _ptr(ff), _ptr(fb), _ptr(w), _ptr(sums), _ptr(gl.contiguous()), _ptr(g),
                  B, H, W, _stream(), nbytes=4 * B * H * W * 7, shape=(B, 2, H, W))                                     # + writes the flow gradient
        return g, None, None


def consis_loss(fwd_flow, bwd_flow, w_fwd):
    """One scale of compute_loss_flow_consis (model_flow_paper.py:183-193) -> [B]; grad to fwd_flow."""
    return _Consis.apply(fwd_flow, bwd_flow.detach(), w_fwd.detach())


# ------------------------------------------------------------------------------------------
# the five losses of Model_flow.forward's scale loop (model_flow_paper.py:224-235), each as ONE launch over the scales per
# direction of the pass (csrc/multiscale.h, C ABI 11): per scale the launch runs the single-scale entry's workgroups with its
# kernel body -- the per-scale ops above give the same bits, these give them in 5 + 5 launches instead of 15 + 15
# ------------------------------------------------------------------------------------------
def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def multiscale_supported(imgs, warped):
    """True when the `_ms` entries serve these scales (what the single-scale fast paths serve: even widths -- the SSIM column pairs --,
    maps of at least 3 x 3 -- the smoothness tiles --, 8-byte aligned tensors -- ssim2_ok of csrc/ssim.hip: a view with an odd storage offset would
    come back as UNFLOW_EINVAL from unflow_ssim_loss_*_ms --, at most 4 scales).  The occlusion weights the SSIM pair also reads are allocated
    inside the op (aligned)."""
    return (0 < len(imgs) <= 4 and all(on_device(t) and t.shape[-1] % 2 == 0 and t.shape[-1] >= 3 and t.shape[-2] >= 3 for t in imgs)
            and all(w.shape[0] <= 65535 for w in warped)
            and all(t.data_ptr() % 8 == 0 and t.is_contiguous() for t in list(imgs) + list(warped)))


def _register_sums(jobs):
    """jobs of one `_ms` forward: finished with everything else inside ``with deferred_loss_sums``, at once otherwise."""
    for j in jobs:
        deferred_loss_sums.add(*j)
    if not deferred_loss_sums.enabled:
        deferred_loss_sums.flush()


class _OccWeightMS(torch.autograd.Function):
    """_OccWeight2 over n scales: (imgs..., warped...) -> (diffs..., weights...)."""

    @staticmethod
    def forward(ctx, n, *ts):
        dev = _dev(*ts)
        imgs = [t.contiguous() for t in ts[:n]]
        warped = [t.contiguous() for t in ts[n:]]
        B = imgs[0].shape[0]
        diffs, wgts = [], []
        for i, w in zip(imgs, warped):
            assert i.shape[1] == 3 and i.shape[0] == B and w.shape[0] == 2 * B and w.shape[1:] == i.shape[1:]
            diffs.append(torch.empty((2 * B, 1) + tuple(i.shape[2:]), dtype=torch.float32, device=dev))
            wgts.append(torch.empty((2 * B, 1) + tuple(i.shape[2:]), dtype=torch.float32, device=dev))
        H, W = [i.shape[2] for i in imgs], [i.shape[3] for i in imgs]
        with _on(dev):
            _call('unflow_occ_weight_fwd_ms', n, _ptrs(imgs), _ptrs(warped), _ptrs(diffs), _ptrs(wgts), _ints(H), _ints(W), B, _stream(),
                  nbytes=sum(B * h * w * 4 * 13 for h, w in zip(H, W)), shape=(n, B, 3, H[0], W[0]))
        ctx.save_for_backward(*imgs, *warped)
        ctx.n = n
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*wgts)                           # weight is .data (model_flow_paper.py:122)
        return (*diffs, *wgts)

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        imgs, warped = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        B = imgs[0].shape[0]
        live = [k for k in range(n) if gs[k] is not None and ctx.needs_input_grad[1 + n + k]]
        out = [None] * n
        if live:
            g = [gs[k].contiguous() for k in live]
            for k in live:
                out[k] = torch.empty_like(warped[k])
            H, W = [imgs[k].shape[2] for k in live], [imgs[k].shape[3] for k in live]
            with _on(imgs[0].device):
                _call('unflow_absdiff_bwd_ms', len(live), _ptrs([imgs[k] for k in live]), _ptrs([warped[k] for k in live]), _ptrs(g),
                      _ptrs([out[k] for k in live]), _ints(H), _ints(W), 2 * B, B, _stream(),
                      nbytes=sum(4 * B * h * w * (3 + 2 * 7) for h, w in zip(H, W)), shape=(len(live), 2 * B, 3, H[0], W[0]))
        return (None, *([None] * n), *out)


class _MaskedMeanMS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, *ts):
        dev = _dev(*ts)
        diffs = [t.contiguous() for t in ts[:n]]
        ws = [t.contiguous() for t in ts[n:]]
        B = diffs[0].shape[0]
        H, W = [d.shape[2] for d in diffs], [d.shape[3] for d in diffs]
        losses = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(n)]
        sums = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(n)]
        parts = [_partials(B, h, w, dev) for h, w in zip(H, W)]
        with _on(dev):
            _call('unflow_masked_mean_fwd_ms', n, _ptrs(diffs), _ptrs(ws), _ptrs(parts), _ints(H), _ints(W), B, _stream(),
                  nbytes=sum(4 * B * h * w * 2 for h, w in zip(H, W)), shape=(n, B, 1, H[0], W[0]))
        jobs = []
        for k in range(n):
            hw = _f32(float(H[k]) * float(W[k]))
            jobs.append((parts[k], losses[k], sums[k], 0, H[k], W[k], 0, hw, hw))
        _register_sums(jobs)
        ctx.save_for_backward(*ws, *sums)
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(losses)

    @staticmethod
    def backward(ctx, *gl):
        n = ctx.n
        ws, sums = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        live = [k for k in range(n) if gl[k] is not None]
        out = [None] * n
        if live:
            B = ws[0].shape[0]
            g = [gl[k].contiguous() for k in live]
            for k in live:
                out[k] = torch.empty_like(ws[k])
            H, W = [ws[k].shape[2] for k in live], [ws[k].shape[3] for k in live]
            with _on(ws[0].device):
                _call('unflow_masked_mean_bwd_ms', len(live), _ptrs([ws[k] for k in live]), _ptrs([sums[k] for k in live]), _ptrs(g),
                      _ptrs([out[k] for k in live]), _ints(H), _ints(W), B, _stream(),
                      nbytes=sum(4 * B * h * w * 2 for h, w in zip(H, W)), shape=(len(live), B, 1, H[0], W[0]))
        return (None, *out, *([None] * n))


class _SsimLossMS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, *ts):
        dev = _dev(*ts)
        imgs = [t.contiguous() for t in ts[:n]]
        warped = [t.contiguous() for t in ts[n:2 * n]]
        ws = [t.contiguous() for t in ts[2 * n:]]
        B, ib = warped[0].shape[0], imgs[0].shape[0]
        assert B % ib == 0 and all(w.shape[0] == B for w in ws)
        H, W = [t.shape[2] for t in warped], [t.shape[3] for t in warped]
        losses = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(n)]
        sums = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(n)]
        parts = [_partials(B, h, w, dev) for h, w in zip(H, W)]
        with _on(dev):
            _call('unflow_ssim_loss_fwd_ms', n, _ptrs(imgs), _ptrs(warped), _ptrs(ws), _ptrs(parts), _ints(H), _ints(W), B, ib, _stream(),
                  nbytes=sum(4 * B * h * w * 7 for h, w in zip(H, W)), shape=(n, B, 3, H[0], W[0]))
        jobs = []
        for k in range(n):
            hw = _f32(float(H[k]) * float(W[k]))
            jobs.append((parts[k], losses[k], sums[k], 1, H[k], W[k], 0, _f32(3.0 * hw), hw, True))
        _register_sums(jobs)
        ctx.save_for_backward(*imgs, *warped, *ws, *sums)
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(losses)

    @staticmethod
    def backward(ctx, *gl):
        n = ctx.n
        t = ctx.saved_tensors
        imgs, warped, ws, sums = t[:n], t[n:2 * n], t[2 * n:3 * n], t[3 * n:]
        live = [k for k in range(n) if gl[k] is not None]
        out = [None] * n
        if live:
            B, ib = warped[0].shape[0], imgs[0].shape[0]
            g = [gl[k].contiguous() for k in live]
            for k in live:
                out[k] = torch.empty_like(warped[k])
            H, W = [warped[k].shape[2] for k in live], [warped[k].shape[3] for k in live]
            pick = lambda seq: _ptrs([seq[k] for k in live])
            with _on(imgs[0].device):
                _call('unflow_ssim_loss_bwd_ms', len(live), pick(imgs), pick(warped), pick(ws), pick(sums), _ptrs(g), pick(out),
                      _ints(H), _ints(W), B, ib, _stream(),
                      nbytes=sum(4 * B * h * w * 10 for h, w in zip(H, W)), shape=(len(live), B, 3, H[0], W[0]))
        return (None, *([None] * n), *out, *([None] * n))


class _Smooth2MS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, *ts):
        dev = _dev(*ts)
        flows = [t.contiguous() for t in ts[:n]]
        imgs = [t.contiguous() for t in ts[n:]]
        B, ib = flows[0].shape[0], imgs[0].shape[0]
        assert B % ib == 0
        H, W = [t.shape[2] for t in flows], [t.shape[3] for t in flows]
        losses = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(n)]
        parts = [_partials(B, h, w, dev) for h, w in zip(H, W)]
        with _on(dev):
            _call('unflow_smooth2_fwd_ms', n, _ptrs(flows), _ptrs(imgs), _ptrs(parts), _ints(H), _ints(W), B, ib, _stream(),
                  nbytes=sum(4 * h * w * (2 * B + 3 * ib) for h, w in zip(H, W)), shape=(n, B, 2, H[0], W[0]))
        jobs = []
        for k in range(n):
            nx = _f32(_f32(2.0 * float(H[k])) * float(W[k] - 2))
            ny = _f32(_f32(2.0 * float(H[k] - 2)) * float(W[k]))
            jobs.append((parts[k], losses[k], None, 2, H[k], W[k], 1, nx, ny))
        _register_sums(jobs)
        ctx.save_for_backward(*flows, *imgs)
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(losses)

    @staticmethod
    def backward(ctx, *gl):
        n = ctx.n
        flows, imgs = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        live = [k for k in range(n) if gl[k] is not None]
        out = [None] * n
        if live:
            B, ib = flows[0].shape[0], imgs[0].shape[0]
            g = [gl[k].contiguous() for k in live]
            for k in live:
                out[k] = torch.empty_like(flows[k])
            H, W = [flows[k].shape[2] for k in live], [flows[k].shape[3] for k in live]
            with _on(flows[0].device):
                _call('unflow_smooth2_bwd_ms', len(live), _ptrs([flows[k] for k in live]), _ptrs([imgs[k] for k in live]), _ptrs(g),
                      _ptrs([out[k] for k in live]), _ints(H), _ints(W), B, ib, _stream(),
                      nbytes=sum(4 * h * w * (4 * B + 3 * ib) for h, w in zip(H, W)), shape=(len(live), B, 2, H[0], W[0]))
        return (None, *out, *([None] * n))


class _ConsisMS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, *ts):
        dev = _dev(*ts)
        ff = [t.contiguous() for t in ts[:n]]
        fb = [t.contiguous() for t in ts[n:2 * n]]
        ws = [t.contiguous() for t in ts[2 * n:]]
        B = ff[0].shape[0]
        H, W = [t.shape[2] for t in ff], [t.shape[3] for t in ff]
        losses = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(n)]
        sums = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(n)]
        parts = [_partials(B, h, w, dev) for h, w in zip(H, W)]
        with _on(dev):
            _call('unflow_consis_fwd_ms', n, _ptrs(ff), _ptrs(fb), _ptrs(ws), _ptrs(parts), _ints(H), _ints(W), B, _stream(),
                  nbytes=sum(4 * B * h * w * 5 for h, w in zip(H, W)), shape=(n, B, 2, H[0], W[0]))
        jobs = []
        for k in range(n):
            hw = _f32(float(H[k]) * float(W[k]))
            jobs.append((parts[k], losses[k], sums[k], 3, H[k], W[k], 0, _f32(2.0 * hw), hw))
        _register_sums(jobs)
        ctx.save_for_backward(*ff, *fb, *ws, *sums)
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(losses)

    @staticmethod
    def backward(ctx, *gl):
        n = ctx.n
        t = ctx.saved_tensors
        ff, fb, ws, sums = t[:n], t[n:2 * n], t[2 * n:3 * n], t[3 * n:]
        live = [k for k in range(n) if gl[k] is not None]
        out = [None] * n
        if live:
            B = ff[0].shape[0]
            g = [gl[k].contiguous() for k in live]
            for k in live:
                out[k] = torch.empty_like(ff[k])
            H, W = [ff[k].shape[2] for k in live], [ff[k].shape[3] for k in live]
            pick = lambda seq: _ptrs([seq[k] for k in live])
            with _on(ff[0].device):
                _call('unflow_consis_bwd_ms', len(live), pick(ff), pick(fb), pick(ws), pick(sums), _ptrs(g), pick(out), _ints(H), _ints(W),
                      B, _stream(), nbytes=sum(4 * B * h * w * 7 for h, w in zip(H, W)), shape=(len(live), B, 2, H[0], W[0]))
        return (None, *out, *([None] * 2 * n))


class _ConsisStackedMS(torch.autograd.Function):
    """_ConsisMS on the decoder's STACKED operands: flows[s] [2B,2,H,W] = (centre->left | centre->right), wgt[s] [2B,1,H,W] = (w_bwd | w_fwd).
    The halves are pointer offsets here instead of `split` nodes in the graph, whose backward costs a zero-fill and a concatenation per
    scale (model_flow_paper.py:183-193: the gradient goes to the forward flow only, the backward flow is detached): the result is a
    [2B,...] gradient whose first half is zero."""

    @staticmethod
    def forward(ctx, n, *ts):
        dev = _dev(*ts)
        flows = [t.contiguous() for t in ts[:n]]
        ws = [t.contiguous() for t in ts[n:]]
        B = flows[0].shape[0] // 2
        assert all(f.shape[0] == 2 * B and w.shape[0] == 2 * B for f, w in zip(flows, ws))
        H, W = [t.shape[2] for t in flows], [t.shape[3] for t in flows]
        losses = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(n)]
        sums = [torch.empty((B, 2), dtype=torch.float32, device=dev) for _ in range(n)]
        parts = [_partials(B, h, w, dev) for h, w in zip(H, W)]
        ptr = lambda addrs: (ctypes.c_void_p * n)(*addrs)
        ff = ptr([f.data_ptr() + 4 * B * 2 * h * w for f, h, w in zip(flows, H, W)])       # the forward half
        fb = ptr([f.data_ptr() for f in flows])
        wf = ptr([t.data_ptr() + 4 * B * h * w for t, h, w in zip(ws, H, W)])              # weight_fwd
        with _on(dev):
            _call('unflow_consis_fwd_ms', n, ff, fb, wf, _ptrs(parts), _ints(H), _ints(W), B, _stream(),
                  nbytes=sum(4 * B * h * w * 5 for h, w in zip(H, W)), shape=(n, B, 2, H[0], W[0]))
        jobs = []
        for k in range(n):
            hw = _f32(float(H[k]) * float(W[k]))
            jobs.append((parts[k], losses[k], sums[k], 3, H[k], W[k], 0, _f32(2.0 * hw), hw))
        _register_sums(jobs)
        ctx.save_for_backward(*flows, *ws, *sums)
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(losses)

    @staticmethod
    def backward(ctx, *gl):
        n = ctx.n
        t = ctx.saved_tensors
        flows, ws, sums = t[:n], t[n:2 * n], t[2 * n:]
        live = [k for k in range(n) if gl[k] is not None]
        out = [None] * n
        if live:
            B = flows[0].shape[0] // 2
            g = [gl[k].contiguous() for k in live]
            H, W = [flows[k].shape[2] for k in live], [flows[k].shape[3] for k in live]
            for k in live:
                out[k] = torch.empty_like(flows[k])
                out[k][:B].zero_()                           # d / d(backward flow) = 0: it is detached in the reference (:186)
            ptr = lambda addrs: (ctypes.c_void_p * len(live))(*addrs)
            ff = ptr([flows[k].data_ptr() + 4 * B * 2 * h * w for k, h, w in zip(live, H, W)])
            fb = ptr([flows[k].data_ptr() for k in live])
            wf = ptr([ws[k].data_ptr() + 4 * B * h * w for k, h, w in zip(live, H, W)])
            go = ptr([out[k].data_ptr() + 4 * B * 2 * h * w for k, h, w in zip(live, H, W)])
            with _on(flows[0].device):
                _call('unflow_consis_bwd_ms', len(live), ff, fb, wf, _ptrs([sums[k] for k in live]), _ptrs(g), go, _ints(H), _ints(W),
                      B, _stream(), nbytes=sum(4 * B * h * w * 7 for h, w in zip(H, W)), shape=(len(live), B, 2, H[0], W[0]))
        return (None, *out, *([None] * n))


class _WarpMaskedMS(torch.autograd.Function):
    """warp_flow(img, flow, use_mask=True) (net_utils.py:16-54) over the n scales of an image pyramid in one launch each way."""

    @staticmethod
    def forward(ctx, n, align_corners, *ts):
        dev = _dev(*ts)
        imgs = [t.contiguous() for t in ts[:n]]
        flows = [t.contiguous() for t in ts[n:]]
        B, C = imgs[0].shape[:2]
        for x, f in zip(imgs, flows):
            _check_flow_shape(x, f)
            assert x.shape[:2] == (B, C)
        H, W = [t.shape[2] for t in imgs], [t.shape[3] for t in imgs]
        outs = [torch.empty_like(x) for x in imgs]
        masks = [torch.empty((B, 1, h, w), dtype=torch.uint8, device=dev) for h, w in zip(H, W)]
        with _on(dev):
            _call('unflow_warp_fwd_ms', n, _ptrs(imgs), _ptrs(flows), _ptrs(outs), _ptrs(masks), _ints(H), _ints(W), B, C, int(align_corners),
                  _stream(), nbytes=sum(B * h * w * (8 * C + 8 + 1) for h, w in zip(H, W)), shape=(n, B, C, H[0], W[0]))
        ctx.save_for_backward(*imgs, *flows, *masks)
        ctx.n, ctx.ac = n, int(align_corners)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*masks)
        return (*outs, *masks)

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        t = ctx.saved_tensors
        imgs, flows, masks = t[:n], t[n:2 * n], t[2 * n:]
        if any(ctx.needs_input_grad[2 + k] for k in range(n)):
            raise RuntimeError('warp_flow_masked_pyramid: the image pyramids are detached (model_flow_paper.py:58); no source gradient is defined here')
        live = [k for k in range(n) if gs[k] is not None and ctx.needs_input_grad[2 + n + k]]
        out = [None] * n
        if live:
            B, C = imgs[0].shape[:2]
            g = [gs[k].contiguous() for k in live]
            for k in live:
                out[k] = torch.empty_like(flows[k])
            H, W = [imgs[k].shape[2] for k in live], [imgs[k].shape[3] for k in live]
            pick = lambda seq: _ptrs([seq[k] for k in live])
            with _on(imgs[0].device):
                _call('unflow_warp_bwd_ms', len(live), pick(imgs), pick(flows), _ptrs(g), pick(masks), pick(out), _ints(H), _ints(W), B, C,
                      ctx.ac, _stream(), nbytes=sum(4 * B * h * w * (2 * C + 4) for h, w in zip(H, W)), shape=(len(live), B, C, H[0], W[0]))
        return (None, None, *([None] * n), *out)


def warp_flow_masked_pyramid(imgs, flows, align_corners=False):
    """Model_flow.warp_flow_pyramid (model_flow_paper.py:62-66): warp_flow(img, flow, use_mask=True) at every scale of an image pyramid
    (C <= 4, detached images) as one launch each way -> the list of masked warped images; values of ops.warp_flow per scale, bit for bit."""
    n = len(imgs)
    res = _WarpMaskedMS.apply(n, bool(align_corners), *[i.detach() for i in imgs], *flows)
    return list(res[:n])


def multiscale_losses(imgs, warped, flows_lr, flows_fwd=None, flows_bwd=None):
    """The scale loop of Model_flow.forward (model_flow_paper.py:224-235) with every loss as ONE launch over the scales:
    imgs[s] [B,3,H,W] centre pyramid, warped[s] [2B,3,H,W] = (from_l | from_r), flows_lr[s] [2B,2,H,W] = (bwd | fwd) flows
    -> (pixel, ssim, smooth, consis): lists of per-scale [2B] / [2B] / [2B] / [B] losses, the values of occ_weight_stacked + masked_mean +
    ssim_loss + smooth2_loss + consis_loss per scale, bit for bit.  flows_fwd[s] / flows_bwd[s] [B,2,H,W]: the halves of flows_lr as tensors
    of their own (a caller that has split them already); None: the consistency term takes its halves from flows_lr by offset, and no
    `split` node -- a concatenation per scale on the way back -- enters the graph."""
    n = len(imgs)
    B = imgs[0].shape[0]
    imgs = [i.detach() for i in imgs]
    ow = _OccWeightMS.apply(n, *imgs, *warped)
    diffs, wgts = ow[:n], [w.detach() for w in ow[n:]]
    pixel = _MaskedMeanMS.apply(n, *diffs, *wgts)
    ssim = _SsimLossMS.apply(n, *imgs, *warped, *wgts)
    smooth = _Smooth2MS.apply(n, *flows_lr, *imgs)
    if flows_fwd is None:
        consis = _ConsisStackedMS.apply(n, *flows_lr, *wgts)
    else:
        consis = _ConsisMS.apply(n, *flows_fwd, *[f.detach() for f in flows_bwd], *[w[B:] for w in wgts])
    return list(pixel), list(ssim), list(smooth), list(consis)


def _ptr_array(tensors):
    """A host array of device pointers (NULL for None) for the entry points that take a list of vectors."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class _LossCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, B, *terms):
        deferred_loss_sums.flush()                     # the per-scale losses are read here
        dev = _dev(*terms)
        terms = [t.contiguous() for t in terms]
        outs = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(4)]
        with _on(dev):
            _call('unflow_loss_combine_fwd', _ptr_array(terms), n, B, _ptr_array(outs), _stream())
        ctx.meta = (n, B)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n, B = ctx.meta
        dev = next(g.device for g in gs if g is not None)
        gs = [None if g is None else g.float().contiguous() for g in gs]
        gin = torch.empty(7 * B, dtype=torch.float32, device=dev)
        with _on(dev):
            _call('unflow_loss_combine_bwd', _ptr_array(gs), B, _ptr(gin), _stream())
        per_loss = [gin[2 * B * k:2 * B * (k + 1)] for k in range(3)] + [gin[6 * B:]]
        return (None, None) + tuple(per_loss[k] for k in range(4) for _ in range(n))      # the same vector for every scale of a loss


def loss_combine(pixel, ssim, smooth, consis):
    """The loss bookkeeping of Model_flow.forward (model_flow_paper.py:224-235) in one launch each way: ``pixel``, ``ssim``,
    ``smooth`` are lists over scales of [2B] vectors (bwd half | fwd half), ``consis`` a list of [B] vectors; returns the four
    [B] tensors of the loss pack -- per loss the terms summed over the scales (in order, from 0), then fwd + bwd."""
    n = len(pixel)
    if not (n == len(ssim) == len(smooth) == len(consis)) or not 1 <= n <= 4:
        raise ValueError('loss_combine: one term per scale and loss, 1 to 4 scales')
    B = consis[0].shape[0]
    for t in list(pixel) + list(ssim) + list(smooth):
        if t.shape != (2 * B,):
            raise ValueError('loss_combine: [2B] vectors expected, got %s for B = %d' % (tuple(t.shape), B))
    for t in consis:
        if t.shape != (B,):
            raise ValueError('loss_combine: [B] consistency terms expected, got %s' % (tuple(t.shape),))
    return _LossCombine.apply(n, B, *pixel, *ssim, *smooth, *consis)


class _WeightedMeanSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        deferred_loss_sums.flush()                     # (a caller that skipped loss_combine reads the losses here)
        dev = _dev(*terms)
        terms = [t.contiguous() for t in terms]
        K, B = len(terms), terms[0].shape[0]
        w = (ctypes.c_float * K)(*weights)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        with _on(dev):
            _call('unflow_weighted_mean_sum_fwd', _ptr_array(terms), w, K, B, _ptr(loss), _stream())
        ctx.meta = (tuple(weights), K, B)
        return loss

    @staticmethod
    def backward(ctx, g):
        weights, K, B = ctx.meta
        g = g.float().contiguous()
        grads = [torch.empty(B, dtype=torch.float32, device=g.device) for _ in range(K)]
        with _on(g.device):
            _call('unflow_weighted_mean_sum_bwd', _ptr(g), (ctypes.c_float * K)(*weights), K, B, _ptr_array(grads), _stream())
        return (None,) + tuple(grads)


def weighted_mean_sum(terms, weights):
    """``sum_k weights[k] * terms[k].mean()`` (train.py:147-150) for up to 8 [B] vectors in one launch each way."""
    terms = list(terms)
    weights = [float(w) for w in weights]
    if not 1 <= len(terms) <= 8 or len(weights) != len(terms) or any(t.dim() != 1 or t.shape != terms[0].shape for t in terms):
        raise ValueError('weighted_mean_sum: 1 to 8 vectors of one length, one weight each')
    return _WeightedMeanSum.apply(weights, *terms)


# ------------------------------------------------------------------------------------------
# conv() epilogue and image pyramid
# ------------------------------------------------------------------------------------------
def _sample_strided(g, shape):
    """(tensor, sample stride in elements) for a gradient whose per-sample (C,H,W) block is dense -- a
    contiguous tensor or a channel slice of a wider NCHW one (a torch.cat operand's gradient); anything
    else is made contiguous first."""
    N, C, H, W = shape
    st = g.stride()
    dense = (st[3] == 1 or W == 1) and (st[2] == W or H == 1) and (st[1] == H * W or C == 1)
    vec = 16 // g.element_size()                      # elements per 16-byte lane access
    if not dense or (N > 1 and st[0] < C * H * W) or ((H * W) % vec == 0 and (st[0] % vec or g.data_ptr() % 16)):
        g = g.contiguous()
        st = g.stride()
    return g, (st[0] if N > 1 else C * H * W)


def _is_nhwc(t):
    """Dense channels_last 4-d tensor that is not ALSO plain-contiguous (C == 1 or H == W == 1 are both)."""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


def _pixel_strided(g, shape):
    """(tensor, pixel stride in elements) for a gradient in channels_last order whose C values of a pixel are dense --
    a channels_last tensor or a channel slice of a wider one (a torch.cat operand's gradient); anything else is
    converted first."""
    N, C, H, W = shape
    st = g.stride()
    ps = st[3] if W > 1 else (st[2] if H > 1 else (st[0] if N > 1 else C))
    ok = (st[1] == 1 and ps >= C and ps % 4 == 0 and g.data_ptr() % 16 == 0 and (W == 1 or st[3] == ps)
          and (H == 1 or st[2] == W * ps) and (N == 1 or st[0] == H * W * ps))
    if not ok:
        g = g.contiguous(memory_format=torch.channels_last)
        ps = C
    return g, ps


def _bias_leaky_backward(ctx, ga, gb):
    (y,) = ctx.saved_tensors
    N, C, H, W = y.shape
    if ga is None:
        ga, gb = gb, None
    if ga is None:
        return None, None, None
    half = y.dtype == torch.bfloat16
    if ga.dtype != y.dtype or (gb is not None and gb.dtype != y.dtype):
        raise TypeError('gradient dtype %s does not match the activation (%s)' % (ga.dtype, y.dtype))
    gin = torch.empty_like(y)                         # (keeps y's memory format)
    gbias = torch.empty(C, dtype=torch.float32, device=y.device)
    if ctx.nhwc:
        ga, sa = _pixel_strided(ga, y.shape)
        sb = 0
        if gb is not None:
            gb, sb = _pixel_strided(gb, y.shape)
        P = N * H * W
        npart = _lib.load().unflow_bias_leaky_partials_nhwc(P, C)
        part = torch.empty(npart, dtype=torch.float32, device=y.device)
        with _on(y.device):
            _call('unflow_bias_leaky_bwd2_nhwc_bf16' if half else 'unflow_bias_leaky_bwd2_nhwc', _ptr(y), _ptr(ga), sa, _ptr(gb), sb,
                  _ptr(gin), _finish_bias_grad(part, gbias, npart // C, C, 1 if half else 0), _ptr(part), P, C, ctypes.c_float(ctx.slope), _stream(),
                  nbytes=(3 if gb is None else 4) * y.element_size() * N * C * H * W, shape=(N, C, H, W))
        return gin, gbias, None
    ga, sa = _sample_strided(ga, y.shape)
    sb = 0
    if gb is not None:
        gb, sb = _sample_strided(gb, y.shape)
    npart = _lib.load().unflow_bias_leaky_partials(N, C, H, W)
    part = torch.empty(npart, dtype=torch.float32, device=y.device)
    with _on(y.device):
        _call('unflow_bias_leaky_bwd2_bf16' if half else 'unflow_bias_leaky_bwd2', _ptr(y), _ptr(ga), sa, _ptr(gb), sb,
              _ptr(gin), _finish_bias_grad(part, gbias, npart // C, C, 1 if half else 0), _ptr(part), N, C, H, W, ctypes.c_float(ctx.slope), _stream(),
              nbytes=(3 if gb is None else 4) * y.element_size() * N * C * H * W, shape=(N, C, H, W))
    return gin, gbias, None


def _bias_leaky_forward(ctx, y, bias, slope):
    half = y.dtype == torch.bfloat16                  # bf16 conv-stack option: bf16 activation, fp32 bias
    _dev(None if half else y, bias)
    if half and (not on_device(y) or y.device != bias.device):
        raise RuntimeError('bias_leaky_relu_: activation and bias must be on the same HIP device')
    N, C, H, W = y.shape
    nhwc = _is_nhwc(y)
    if nhwc:
        if C % 4:
            raise RuntimeError('bias_leaky_relu_: the channels_last epilogue needs C %% 4 == 0 (got C=%d)' % C)
        with _on(y.device):
            _call('unflow_bias_leaky_fwd_nhwc_bf16' if half else 'unflow_bias_leaky_fwd_nhwc', _ptr(y), _ptr(bias), N * H * W, C,
                  ctypes.c_float(slope), _stream(),
                  nbytes=2 * y.element_size() * N * C * H * W, shape=(N, C, H, W))
    else:
        if not y.is_contiguous():
            raise RuntimeError('bias_leaky_relu_ works in place on a contiguous (NCHW or channels_last) convolution output')
        with _on(y.device):
            _call('unflow_bias_leaky_fwd_bf16' if half else 'unflow_bias_leaky_fwd', _ptr(y), _ptr(bias), N, C, H, W,
                  ctypes.c_float(slope), _stream(), nbytes=2 * y.element_size() * N * C * H * W, shape=(N, C, H, W))
    ctx.mark_dirty(y)
    ctx.save_for_backward(y)
    ctx.slope = slope
    ctx.nhwc = nhwc
    ctx.set_materialize_grads(False)


class _BiasLeaky(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias, slope):
        _bias_leaky_forward(ctx, y, bias, slope)
        return y

    @staticmethod
    def backward(ctx, g):
        return _bias_leaky_backward(ctx, g, None)


class _BiasLeakyTwo(torch.autograd.Function):
    """The same activation handed out twice (the second output aliases the first): autograd then delivers
    the gradients of its two consumers separately and the backward kernel adds them while it reads them."""

    @staticmethod
    def forward(ctx, y, bias, slope):
        _bias_leaky_forward(ctx, y, bias, slope)
        return y, y.view_as(y)

    @staticmethod
    def backward(ctx, ga, gb):
        return _bias_leaky_backward(ctx, ga, gb)


def bias_leaky_relu_(y, bias, negative_slope=0.1, consumers=1):
    """In-place ``leaky_relu(y + bias[None,:,None,None])`` on a bias-free conv output: the epilogue of the
    reference's conv() block (net_utils.py:7-11) in one pass; its backward also reduces the bias gradient.
    ``consumers=2`` returns the activation twice (one handle per consumer) so that the backward pass adds the
    two incoming gradients inside the kernel instead of in a separate pass."""
    if y.dim() == 4 and y.shape[1] % 4 and _is_nhwc(y):
        # the channels_last kernels keep a channel quad per thread; other widths (none in the flow network) take the NCHW
        # epilogue on a re-laid-out copy instead of failing
        y = y.contiguous()
    if consumers == 2:
        return _BiasLeakyTwo.apply(y, bias, float(negative_slope))
    return _BiasLeaky.apply(y, bias, float(negative_slope))


class _BiasLeakyInto(torch.autograd.Function):
    """conv() epilogue that FILLS cat buffers: ``leaky_relu(y + bias)`` of a bias-free channels_last conv output ``y`` is written
    into the channel slice ``[off, off + C)`` of up to two wider channels_last buffers -- the inputs of the next two
    convolutions of the decoder, which the reference builds with torch.cat (pwc_tf.py:114-118) -- and, with ``inplace``, into
    ``y`` itself (an activation that is also consumed on its own).  Each buffer is modified in place and handed back, so
    autograd chains the two epilogues that fill a buffer; the backward reads the activation's sign and the (at most two)
    upstream gradients straight out of the slices (one kernel, no slice / cat / add passes)."""

    @staticmethod
    def forward(ctx, y, bias, slope, inplace, buf_a, off_a, buf_b, off_b):
        half = y.dtype == torch.bfloat16
        _dev(None if half else y, bias)
        N, C, H, W = y.shape
        if not (_is_nhwc(y) or (y.is_contiguous() and (C == 1 or H * W == 1))) or C % 4:
            raise RuntimeError('bias_leaky_relu_into needs a dense channels_last convolution output with C % 4 == 0')
        dests = []                                       # (tensor, pixel stride, element offset)
        if inplace:
            dests.append((y, C, 0))
        for buf, off in ((buf_a, off_a), (buf_b, off_b)):
            if buf is None:
                continue
            dense_cl = _is_nhwc(buf) or (buf.dim() == 4 and buf.is_contiguous() and (buf.shape[1] == 1 or H * W == 1))
            if buf.dtype != y.dtype or buf.device != y.device or buf.shape[0] != N or tuple(buf.shape[2:]) != (H, W) or not dense_cl \
                    or off % 4 or off < 0 or off + C > buf.shape[1]:
                raise RuntimeError('bias_leaky_relu_into: a destination must be a dense channels_last [N, C_total, H, W] tensor of the '
                                   "activation's dtype with a 4-aligned channel offset")
            dests.append((buf, buf.shape[1], off))
        if not 1 <= len(dests) <= 2:
            raise RuntimeError('bias_leaky_relu_into writes one or two destinations, got %d' % len(dests))
        es = y.element_size()
        d1 = ctypes.c_void_p(dests[0][0].data_ptr() + dests[0][2] * es)
        d2 = ctypes.c_void_p(dests[1][0].data_ptr() + dests[1][2] * es) if len(dests) > 1 else ctypes.c_void_p(0)
        with _on(y.device):
            _call('unflow_bias_leaky_fwd_nhwc_to_bf16' if half else 'unflow_bias_leaky_fwd_nhwc_to', _ptr(y), _ptr(bias), N * H * W, C,
                  ctypes.c_float(slope), d1, dests[0][1], d2, dests[1][1] if len(dests) > 1 else 0, _stream(),
                  nbytes=(1 + len(dests)) * es * N * C * H * W, shape=(N, C, H, W))
        dirty = [t for t in ((y if inplace else None), buf_a, buf_b) if t is not None]
        ctx.mark_dirty(*dirty)
        # the activation (for its sign): NOT through save_for_backward -- the buffer is legitimately modified again by the
        # epilogue that fills its other channels, which would trip the saved-tensor version check.  A DETACHED alias: the buffer
        # is also an output of this node (mark_dirty), so keeping the tensor itself would close the cycle ctx -> tensor ->
        # grad_fn -> ctx, which only backward breaks (a grad-enabled forward without backward would pin every cat buffer).
        ctx.act, ctx.act_ps, ctx.act_off = dests[0][0].detach(), dests[0][1], dests[0][2]
        ctx.meta = (N, C, H, W, slope, inplace, off_a, off_b, half)
        ctx.set_materialize_grads(False)
        return (y if inplace else None), buf_a, buf_b

    @staticmethod
    def backward(ctx, g_y, g_a, g_b):
        N, C, H, W, slope, inplace, off_a, off_b, half = ctx.meta
        srcs = []                                        # (gradient tensor, channel offset)
        if inplace and g_y is not None:
            srcs.append((g_y, 0))
        if g_a is not None:
            srcs.append((g_a, off_a))
        if g_b is not None:
            srcs.append((g_b, off_b))
        if not srcs:
            return (None,) * 8
        if len(srcs) > 2:
            raise RuntimeError('bias_leaky_relu_into: at most two consumers per activation')
        act = ctx.act
        dt = act.dtype
        es = act.element_size()
        ptrs = []
        for g, off in srcs:
            if g.dtype != dt:
                raise TypeError('gradient dtype %s does not match the activation (%s)' % (g.dtype, dt))
            if not (_is_nhwc(g) or (g.is_contiguous() and (g.shape[1] == 1 or H * W == 1))):
                g = g.contiguous(memory_format=torch.channels_last)
            ptrs.append((g, ctypes.c_void_p(g.data_ptr() + off * es), g.shape[1]))
        gin = torch.empty((N, C, H, W), dtype=dt, device=act.device, memory_format=torch.channels_last)
        gbias = torch.empty(C, dtype=torch.float32, device=act.device)
        P = N * H * W
        npart = _lib.load().unflow_bias_leaky_partials_nhwc(P, C)
        part = torch.empty(npart, dtype=torch.float32, device=act.device)
        g2 = ptrs[1] if len(ptrs) > 1 else (None, ctypes.c_void_p(0), 0)
        with _on(act.device):
            _call('unflow_bias_leaky_bwd2_nhwc_from_bf16' if half else 'unflow_bias_leaky_bwd2_nhwc_from',
                  ctypes.c_void_p(act.data_ptr() + ctx.act_off * es), ctx.act_ps, ptrs[0][1], ptrs[0][2], g2[1], g2[2],
                  _ptr(gin), _finish_bias_grad(part, gbias, npart // C, C, 1 if half else 0), _ptr(part), P, C, ctypes.c_float(slope), _stream(),
                  nbytes=(2 + len(ptrs)) * es * N * C * H * W, shape=(N, C, H, W))
        ctx.act = None
        # the buffers' incoming gradients pass through to whoever filled their other channels
        return gin, gbias, None, None, g_a, None, g_b, None


def bias_leaky_relu_into(y, bias, negative_slope, buf_a, off_a, buf_b=None, off_b=0, inplace=False):
    """The conv() epilogue (net_utils.py:7-11) of a channels_last convolution output ``y``, written into ``buf_a[:, off_a:off_a+C]``
    (and ``buf_b[:, off_b:off_b+C]``; and into ``y`` itself with ``inplace``): the way the decoder's cat((x_k, x_k+1)) inputs
    come into being without a cat.  Returns ``(y or None, buf_a, buf_b)`` -- use the returned buffers from here on."""
    return _BiasLeakyInto.apply(y, bias, float(negative_slope), bool(inplace), buf_a, int(off_a), buf_b, int(off_b))


class _CatChannelsLast(torch.autograd.Function):
    """cat(tensors, 1) of up to three fp32 NCHW tensors, produced directly in channels_last order (fp32, or bf16 for the bf16
    conv-stack option); backward hands each input its fp32 NCHW gradient (one kernel each way)."""

    @staticmethod
    def forward(ctx, half, *ts):
        dev = _dev(*ts)
        ts = [t.contiguous() for t in ts]
        B, _, H, W = ts[0].shape
        cs = [t.shape[1] for t in ts] + [0] * (3 - len(ts))
        out = torch.empty((B, sum(cs), H, W), dtype=torch.bfloat16 if half else torch.float32, device=dev, memory_format=torch.channels_last)
        ps = [_ptr(t) for t in ts] + [None] * (3 - len(ts))
        with _on(dev):
            _call('unflow_cat_nhwc_bf16' if half else 'unflow_cat_nhwc', ps[0], cs[0], ps[1], cs[1], ps[2], cs[2], _ptr(out), B, H * W, _stream(),
                  nbytes=(6 if half else 8) * out.numel(), shape=tuple(out.shape))
        ctx.cs = cs
        ctx.n = len(ts)
        ctx.half = half
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = g.shape
        want = torch.bfloat16 if ctx.half else torch.float32
        if g.dtype != want or not g.is_contiguous(memory_format=torch.channels_last):
            g = g.to(want).contiguous(memory_format=torch.channels_last)
        need = [ctx.needs_input_grad[k + 1] for k in range(ctx.n)] + [False] * (3 - ctx.n)
        outs = [torch.empty((B, ctx.cs[k], H, W), dtype=torch.float32, device=g.device) if need[k] else None for k in range(3)]
        if any(need):
            with _on(g.device):
                _call('unflow_split_nhwc_bf16' if ctx.half else 'unflow_split_nhwc', _ptr(g), _ptr(outs[0]), ctx.cs[0], _ptr(outs[1]), ctx.cs[1],
                      _ptr(outs[2]), ctx.cs[2], B, H * W, _stream(), nbytes=(6 if ctx.half else 8) * g.numel(), shape=(B, C, H, W))
        return (None,) + tuple(outs[:ctx.n])


class _ToNCHW(torch.autograd.Function):
    """A channels_last activation (fp32 or bf16) as a plain fp32 NCHW tensor (and its gradient back into channels_last, in the
    activation's dtype).  ``dup`` > 0: the last ``dup`` samples are written twice (output batch B + dup) and their two gradients
    meet in the backward kernel."""

    @staticmethod
    def forward(ctx, x, dup):
        B, C, H, W = x.shape
        ctx.half = x.dtype == torch.bfloat16
        ctx.dup = dup
        out = torch.empty((B + dup, C, H, W), dtype=torch.float32, device=x.device)
        with _on(x.device):
            _call('unflow_to_nchw_dup_bf16' if ctx.half else 'unflow_to_nchw_dup', _ptr(x), _ptr(out), C, B, dup, H * W, _stream(),
                  nbytes=(2 if ctx.half else 4) * x.numel() + 4 * out.numel(), shape=(B + dup, C, H, W))
        return out

    @staticmethod
    def backward(ctx, g):
        Bd, C, H, W = g.shape
        B = Bd - ctx.dup
        g = g.float().contiguous()
        out = torch.empty((B, C, H, W), dtype=torch.bfloat16 if ctx.half else torch.float32, device=g.device, memory_format=torch.channels_last)
        with _on(g.device):
            _call('unflow_to_nhwc_fold_bf16' if ctx.half else 'unflow_to_nhwc_fold', _ptr(g), _ptr(out), C, B, ctx.dup, H * W, _stream(),
                  nbytes=4 * g.numel() + (2 if ctx.half else 4) * out.numel(), shape=(B, C, H, W))
        return out, None


class _ToNCHWSplit(torch.autograd.Function):
    """_ToNCHW whose output batch comes back as TWO tensors -- the first ``head`` samples and the rest (with the duplicated tail) -- both
    views of the one buffer the forward kernel fills.  What it saves is in the backward: a caller that splits _ToNCHW's result makes
    autograd concatenate the two gradients into one buffer again (a copy of every pyramid level's gradient per step); here each
    gradient goes through the fold kernel from where it is -- two launches of unflow_to_nhwc_fold on the two halves of the result."""

    @staticmethod
    def forward(ctx, x, head, dup):
        B, C, H, W = x.shape
        ctx.half = x.dtype == torch.bfloat16
        ctx.meta = (B, head, dup)
        out = torch.empty((B + dup, C, H, W), dtype=torch.float32, device=x.device)
        with _on(x.device):
            _call('unflow_to_nchw_dup_bf16' if ctx.half else 'unflow_to_nchw_dup', _ptr(x), _ptr(out), C, B, dup, H * W, _stream(),
                  nbytes=(2 if ctx.half else 4) * x.numel() + 4 * out.numel(), shape=(B + dup, C, H, W))
        return out[:head], out[head:]

    @staticmethod
    def backward(ctx, ga, gb):
        B, head, dup = ctx.meta
        g = ga if ga is not None else gb
        if g is None:
            return None, None, None
        C, H, W = g.shape[1:]
        out = torch.empty((B, C, H, W), dtype=torch.bfloat16 if ctx.half else torch.float32, device=g.device, memory_format=torch.channels_last)
        entry = 'unflow_to_nhwc_fold_bf16' if ctx.half else 'unflow_to_nhwc_fold'
        esz = 2 if ctx.half else 4
        # NHWC is sample-major: out[:head] and out[head:] are contiguous pieces of the one gradient buffer
        for piece, first, count, d in ((ga, 0, head, 0), (gb, head, B - head, dup)):
            if count == 0:
                continue
            dst = ctypes.c_void_p(out.data_ptr() + first * C * H * W * esz)
            if piece is None:                            # (that half was not used: its gradient is zero)
                out[first:first + count].zero_()
                continue
            piece = piece.float().contiguous()
            with _on(g.device):
                _call(entry, _ptr(piece), dst, C, count, d, H * W, _stream(), nbytes=4 * piece.numel() + esz * count * C * H * W, shape=(count, C, H, W))
        return out, None, None


def to_nchw_split(x, head, dup_tail=0):
    """``to_nchw(x, dup_tail).split((head, rest))`` without the gradient concatenation a split costs on the way back: returns
    (y[:head], y[head:]) of the fp32 NCHW form y of the channels_last activation x (with its last ``dup_tail`` samples repeated)."""
    head, dup_tail = int(head), int(dup_tail)
    if not (0 <= dup_tail <= x.shape[0] - head and 0 <= head <= x.shape[0]):
        raise ValueError('to_nchw_split: head %d, dup_tail %d for a batch of %d' % (head, dup_tail, x.shape[0]))
    if x.dtype not in (torch.float32, torch.bfloat16) or not _is_nhwc(x):
        return to_nchw(x, dup_tail).split((head, x.shape[0] - head + dup_tail))
    if x.dtype == torch.float32:
        _dev(x)
    elif not on_device(x):
        raise RuntimeError('unopticalflow_amd ops run on an MI355X (HIP) device only; got a %s tensor. There is no CPU fallback.' % x.device)
    return _ToNCHWSplit.apply(x, head, dup_tail)


def cat_channels_last(tensors, dtype=torch.float32):
    """``torch.cat(tensors, 1).contiguous(memory_format=torch.channels_last)`` for up to three fp32 NCHW tensors in ONE
    pass (the decoder input of pwc_tf.py:113 at the border of the channels_last conv stack).  ``dtype=torch.bfloat16``: the
    result rounded to bf16 in the same pass (what autocast would make of it in front of a convolution)."""
    tensors = tuple(tensors)
    if not 1 <= len(tensors) <= 3:
        raise ValueError('cat_channels_last takes 1 to 3 tensors, got %d' % len(tensors))
    if any(t.dtype != torch.float32 or t.dim() != 4 or t.shape[0] != tensors[0].shape[0] or t.shape[2:] != tensors[0].shape[2:]
           for t in tensors):
        raise ValueError('cat_channels_last: fp32 [B, C_k, H, W] tensors with equal B, H, W expected')
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError('cat_channels_last: fp32 or bf16 output, got %s' % dtype)
    return _CatChannelsLast.apply(dtype == torch.bfloat16, *tensors)


def to_nchw(x, dup_tail=0):
    """A dense channels_last fp32 or bf16 activation as a contiguous fp32 NCHW tensor (one transposing kernel each way).
    ``dup_tail`` = d > 0: the result has d more samples, copies of the last d (``torch.cat((y, y[-d:]))`` without the cat: the centre
    frame's pyramid features feed both decoder directions)."""
    dup_tail = int(dup_tail)
    if not 0 <= dup_tail <= x.shape[0]:
        raise ValueError('to_nchw: dup_tail %d for a batch of %d' % (dup_tail, x.shape[0]))
    if x.dtype not in (torch.float32, torch.bfloat16) or not _is_nhwc(x):
        x = x.contiguous()
        return torch.cat((x, x[x.shape[0] - dup_tail:]), 0) if dup_tail else x
    if x.dtype == torch.float32:
        _dev(x)
    elif not on_device(x):
        raise RuntimeError('unopticalflow_amd ops run on an MI355X (HIP) device only; got a %s tensor. There is no CPU fallback.' % x.device)
    return _ToNCHW.apply(x, dup_tail)


class _FlowHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias, res):
        half = y.dtype == torch.bfloat16
        dev = _dev(bias, res) if half else _dev(y, bias, res)
        N, C, H, W = y.shape
        res = None if res is None else res.contiguous()
        out = torch.empty((N, 2, H, W), dtype=torch.float32, device=dev)
        with _on(dev):
            _call('unflow_flow_head_fwd_bf16' if half else 'unflow_flow_head_fwd', _ptr(y), _ptr(bias), _ptr(res), _ptr(out), N, H * W, _stream(),
                  nbytes=N * H * W * 2 * ((2 if half else 4) + 4 + (4 if res is not None else 0)), shape=(N, 2, H, W))
        ctx.half = half
        ctx.has_res = res is not None
        return out

    @staticmethod
    def backward(ctx, g):
        N, _, H, W = g.shape
        g = g.contiguous()
        gy = torch.empty((N, 2, H, W), dtype=torch.bfloat16 if ctx.half else torch.float32, device=g.device, memory_format=torch.channels_last)
        gbias = torch.empty(2, dtype=torch.float32, device=g.device)
        lib = _lib.load()
        partials = torch.empty(lib.unflow_flow_head_partials(), dtype=torch.float32, device=g.device)
        nblk = min((N * H * W + 255) // 256, lib.unflow_flow_head_partials() // 2)
        with _on(g.device):
            _call('unflow_flow_head_bwd_bf16' if ctx.half else 'unflow_flow_head_bwd', _ptr(g), _ptr(gy), _finish_bias_grad(partials, gbias, nblk, 2, 2), _ptr(partials), N, H * W,
                  _stream(), nbytes=N * H * W * 2 * ((2 if ctx.half else 4) + 4), shape=(N, 2, H, W))
        return gy, gbias, (g if ctx.has_res else None)


def flow_head(y, bias, residual=None):
    """``(y + bias.view(1, 2, 1, 1)).float().contiguous() [+ residual]`` for the bias-free output ``y`` [N,2,H,W] (channels_last, fp32
    or bf16) of a predict_flow convolution (pwc_tf.py:93-94,118,130,...): the fp32 NCHW flow in one pass each way."""
    if y.dim() != 4 or y.shape[1] != 2 or y.dtype not in (torch.float32, torch.bfloat16) or not on_device(y):
        raise ValueError('flow_head: a [N,2,H,W] fp32 / bf16 HIP tensor expected, got %s %s' % (tuple(y.shape), y.dtype))
    if not y.is_contiguous(memory_format=torch.channels_last):
        y = y.contiguous(memory_format=torch.channels_last)
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise ValueError('flow_head: residual %s against %s' % (tuple(residual.shape), tuple(y.shape)))
    return _FlowHead.apply(y, bias, residual)


class _UpsampleScaled(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, mul):
        dev = _dev(x)
        x = x.contiguous()
        B, C, Hi, Wi = x.shape
        out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=dev)
        with _on(dev):
            _call('unflow_upsample_scaled_fwd', _ptr(x), _ptr(out), B * C, Hi, Wi, Ho, Wo, ctypes.c_float(mul), _stream(),
                  nbytes=4 * B * C * (Hi * Wi + Ho * Wo), shape=(B, C, Ho, Wo))
        ctx.meta = (B, C, Hi, Wi, Ho, Wo, mul)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, Hi, Wi, Ho, Wo, mul = ctx.meta
        g = g.contiguous()
        gin = torch.empty((B, C, Hi, Wi), dtype=torch.float32, device=g.device)
        with _on(g.device):
            _call('unflow_upsample_scaled_bwd', _ptr(g), _ptr(gin), B * C, Hi, Wi, Ho, Wo, ctypes.c_float(mul), _stream(),
                  nbytes=4 * B * C * (Hi * Wi + Ho * Wo), shape=(B, C, Ho, Wo))
        return gin, None, None, None


def upsample_bilinear_scaled(x, size, mul=1.0):
    """``mul * F.interpolate(x, size, mode='bilinear', align_corners=False)`` for integer up-sampling factors, one kernel each way
    (the decoder's flow up-sampling, pwc_tf.py:119-177: ``interpolate(flow, scale_factor=2) * 2.0`` and ``interpolate(flow * 4.0, size)``)."""
    Ho, Wo = int(size[0]), int(size[1])
    if x.dim() != 4 or Ho % x.shape[2] or Wo % x.shape[3] or Ho < x.shape[2] or Wo < x.shape[3]:
        raise ValueError('upsample_bilinear_scaled: integer up-sampling factors only, got %s -> %s' % (tuple(x.shape[2:]), (Ho, Wo)))
    return _UpsampleScaled.apply(x, Ho, Wo, float(mul))


def img_pyramid(img):
    """Scales 1 and 2 of generate_img_pyramid (model_flow_paper.py:54-60) for a contiguous [N,C,H,W]
    image batch: (2x2 box means [N,C,H/2,W/2], 4x4 box means [N,C,H/4,W/4]).  No gradient (``.data``)."""
    dev = _dev(img)
    img = img.detach().contiguous()
    N, C, H, W = img.shape
    if H % 4 or W % 4:
        raise ValueError('img_pyramid needs H and W to be multiples of 4, got %dx%d' % (H, W))
    half = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=dev)
    quarter = torch.empty((N, C, H // 4, W // 4), dtype=torch.float32, device=dev)
    with _on(dev):
        _call('unflow_img_pyramid', _ptr(img), _ptr(half), _ptr(quarter), N * C, H, W, _stream(),
              nbytes=N * C * H * W * 4 * 21 // 16, shape=(N, C, H, W))
    return half, quarter


def prepare_triplets(images, img_hw, flips=None, device=None, src_is_rgb=True, staging=None):
    """Decoded stacked triplets -> the train step's input ``[B,3,3H,W]`` fp32 on the device.

    ``images``: list of uint8 host arrays / CPU tensors ``[rows, w, 3]`` (sizes may differ: KITTI drives have
    different resolutions).  Device side of ``KITTI_Prepared.__getitem__`` (kitti_prepared.py:63-90,145-148):
    frame split, cv2-compatible 8-bit bilinear resize to ``img_hw``, optional horizontal flip, /255.
    One pinned staging buffer, one async copy, one kernel on the current stream.  A caller that passes its
    own ``staging`` (pinned uint8) must not rewrite it before this call's copy has run (see data.py).
    """
    dev = torch.device(device if device is not None else ('cuda', torch.cuda.current_device()))
    if dev.type != 'cuda':
        raise ValueError('prepare_triplets runs on an MI355X; got device %s' % dev)
    H, W = int(img_hw[0]), int(img_hw[1])
    if W % 4:
        raise ValueError('img_hw[1] must be a multiple of 4, got %d' % W)
    B = len(images)
    imgs = [im.numpy() if isinstance(im, torch.Tensor) else np.asarray(im) for im in images]
    for im in imgs:
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3 or im.shape[0] < 3:
            raise ValueError('expected uint8 [rows>=3, w, 3] images, got %s %s' % (im.dtype, tuple(im.shape)))
    sizes = [im.size for im in imgs]
    head = 16 * B + 8 * B + ((B + 15) // 16) * 16                    # offsets (i64) | dims (2 x i32) | flips (u8, padded)
    offs, pos = [], head
    for n in sizes:
        offs.append(pos)
        pos += (n + 15) // 16 * 16
    if staging is None or staging.numel() < pos:
        staging = torch.empty(pos, dtype=torch.uint8).pin_memory()
    host = staging[:pos]
    # packed with plain single-threaded memcpy (numpy): a torch copy_ fans each 4 MB image out over every OpenMP
    # thread, whose spin-waits then starve the decoding workers (113 -> 599 triplets/s, tools/microbench.py loader)
    hb = host.numpy()
    hb[:8 * B].view(np.int64)[:] = offs
    hb[8 * B:16 * B].view(np.int32)[:] = np.asarray([[im.shape[0], im.shape[1]] for im in imgs], np.int32).reshape(-1)
    hb[16 * B:16 * B + B] = 0 if flips is None else np.asarray(flips, dtype=np.uint8)
    for im, o, n in zip(imgs, offs, sizes):
        hb[o:o + n] = im.reshape(-1)
    buf = host.to(dev, non_blocking=True)
    out = torch.empty((B, 3, 3 * H, W), dtype=torch.float32, device=dev)
    base = buf.data_ptr()
    with _on(dev):
        _call('unflow_prepare_triplets', ctypes.c_void_p(base), ctypes.c_void_p(base), ctypes.c_void_p(base + 8 * B),
              ctypes.c_void_p(base + 16 * B), _ptr(out), B, H, W,
              1 if src_is_rgb else 0, _stream(), nbytes=sum(sizes) + out.numel() * 4, shape=(B, 3, 3 * H, W))
    return out
