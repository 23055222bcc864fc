"""The HIP kernels as PyTorch dispatcher operators: ``torch.ops.unflow_hip.*``.

    from unopticalflow_amd import torch_ops
    torch_ops.load()
    cv = torch.ops.unflow_hip.corr(f1, f2, 4)                      # differentiable
    cv = torch.ops.unflow_hip.corr_fwd(f1, f2, 4)                  # the raw kernels, as registered in csrc/torch_ops.cpp

``libunflow_torch.so`` (csrc/torch_ops.cpp, built by ``python -m unopticalflow_amd.build``) registers the schemas, the HIP
kernels (CUDA dispatch key on PyTorch-ROCm) and Meta shape functions, so FakeTensor / ``torch.compile`` tracing works
without running a kernel.  This module adds the differentiable composites ``corr``, ``warp`` and ``warp_corr`` with their
autograd formulas (``torch.library.register_autograd``).  The ctypes path of ``ops.py`` -- what ``Model_flow`` uses --
does not depend on any of this.
"""
import os

import torch

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, 'libunflow_torch.so')
_loaded = False


def load():
    """Load libunflow_torch.so (once) and register the differentiable composites."""
    global _loaded
    if _loaded:
        return
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libunflow_torch.so is not built (%s); run `python -m unopticalflow_amd.build`' % LIB_PATH)
    torch.ops.load_library(LIB_PATH)
    _register_composites()
    _loaded = True


def _register_composites():
    lib = torch.library

    # ---- corr: PWC_tf.corr_naive (pwc_tf.py:97-106)
    @lib.custom_op('unflow_hip::corr', mutates_args=())
    def corr(f1: torch.Tensor, f2: torch.Tensor, d: int) -> torch.Tensor:
        return torch.ops.unflow_hip.corr_fwd(f1, f2, d)

    @corr.register_fake
    def _(f1, f2, d):
        return f1.new_empty((f1.shape[0], (2 * d + 1) ** 2, f1.shape[2], f1.shape[3]))

    def corr_setup(ctx, inputs, output):
        f1, f2, d = inputs
        ctx.save_for_backward(f1, f2)
        ctx.d = d

    def corr_backward(ctx, g):
        f1, f2 = ctx.saved_tensors
        gf1, gf2 = torch.ops.unflow_hip.corr_bwd(f1, f2, g.contiguous(), ctx.d)
        return gf1, gf2, None

    corr.register_autograd(corr_backward, setup_context=corr_setup)

    # ---- warp: warp_flow(x, flow, use_mask=False) (net_utils.py:16-46)
    @lib.custom_op('unflow_hip::warp', mutates_args=())
    def warp(src: torch.Tensor, flow: torch.Tensor, align_corners: bool) -> torch.Tensor:
        return torch.ops.unflow_hip.warp_fwd(src, flow, align_corners, False)[0]

    @warp.register_fake
    def _(src, flow, align_corners):
        return torch.empty_like(src)

    def warp_setup(ctx, inputs, output):
        src, flow, ac = inputs
        ctx.save_for_backward(src, flow)
        ctx.ac = ac

    def warp_backward(ctx, g):
        src, flow = ctx.saved_tensors
        gsrc, gflow = torch.ops.unflow_hip.warp_bwd(src, flow, g.contiguous(), None, ctx.ac, True)
        return gsrc, gflow, None

    warp.register_autograd(warp_backward, setup_context=warp_setup)

    # ---- warp_corr: one decoder level's corr(f1, warp(f2, flow)) (pwc_tf.py:121-122)
    @lib.custom_op('unflow_hip::warp_corr', mutates_args=())
    def warp_corr(f1: torch.Tensor, f2: torch.Tensor, flow: torch.Tensor, d: int, align_corners: bool) -> torch.Tensor:
        return torch.ops.unflow_hip.warp_corr_fwd(f1, f2, flow, d, align_corners)

    @warp_corr.register_fake
    def _(f1, f2, flow, d, align_corners):
        return f1.new_empty((f1.shape[0], (2 * d + 1) ** 2, f1.shape[2], f1.shape[3]))

    def wc_setup(ctx, inputs, output):
        f1, f2, flow, d, ac = inputs
        ctx.save_for_backward(f1, f2, flow)
        ctx.d, ctx.ac = d, ac

    def wc_backward(ctx, g):
        f1, f2, flow = ctx.saved_tensors
        gf1, gf2, gflow = torch.ops.unflow_hip.warp_corr_bwd(f1, f2, flow, g.contiguous(), ctx.d, ctx.ac)
        return gf1, gf2, gflow, None, None

    warp_corr.register_autograd(wc_backward, setup_context=wc_setup)
