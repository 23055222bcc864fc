import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops


def conv_weight(c):
    """The weight a convolution module contracts with in this forward: its parameter, or -- bf16 conv-stack option -- the
    bf16 copy ``WeightShadows`` made for the whole network in one launch."""
    w = c.__dict__.get('_w_half')
    return c.weight if w is None else w


class _CastAll(torch.autograd.Function):
    """bf16 copies of a list of fp32 parameters by ONE multi-tensor kernel (torch._foreach_copy_), and their bf16 gradients
    widened back to fp32 the same way.  Autocast does the same arithmetic per convolution call: one cast launch per weight
    forward and one per weight gradient backward, ~100 launches of 4-5 us per step."""

    @staticmethod
    def forward(ctx, *ws):
        outs = [torch.empty_like(w, dtype=torch.bfloat16) for w in ws]          # (empty_like keeps the channels_last strides)
        torch._foreach_copy_(outs, [w.detach() for w in ws])
        ctx.set_materialize_grads(False)        # a shadow nobody used (a head's bias when ops.flow_head adds it in fp32) has no gradient: no zero-fill + add
        ctx.like = [(w.shape, w.stride()) for w in ws]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        outs, src, dst = [], [], []
        for g, (shape, stride) in zip(gs, ctx.like):
            if g is None:
                outs.append(None)
                continue
            o = torch.empty_strided(shape, stride, dtype=torch.float32, device=g.device)
            if g.stride() != stride:                    # (the multi-tensor kernel walks both sides by offset)
                g = torch.empty_strided(shape, stride, dtype=g.dtype, device=g.device).copy_(g)
            outs.append(o); src.append(g); dst.append(o)
        if dst:
            torch._foreach_copy_(dst, src)
        return tuple(outs)


class WeightShadows:
    """Context manager around a network pass of the bf16 option: every convolution weight (and the bias of the plain
    Conv2d heads, which autocast would cast per call too) of ``modules`` gets its bf16 copy for the pass; ``conv_weight`` /
    ``HeadConv2d`` pick it up.  Gradients reach the fp32 parameters through ``_CastAll.backward``."""

    def __init__(self, modules, enabled=True, groups=1):
        """``groups``: number of cast nodes the weights are spread over, consecutive in registration order.  One node hands
        EVERY weight gradient to its parameter at the very end of backward; with the eager data-parallel step, whose all-reduce
        pieces leave from hooks the moment their last gradient exists, that would serialise the whole exchange behind
        backward -- the trainer asks for one node per piece then (FlowTrainer: ``model.weight_shadow_groups``)."""
        self.convs = [m for mod in modules for m in mod.modules() if isinstance(m, nn.Conv2d)] if enabled else []
        self.groups = max(1, min(int(groups), len(self.convs))) if self.convs else 1

    def __enter__(self):
        if not self.convs:
            return self
        n = len(self.convs)
        for g in range(self.groups):
            part = self.convs[g * n // self.groups:(g + 1) * n // self.groups]
            heads = [m for m in part if isinstance(m, HeadConv2d) and m.bias is not None]
            halves = _CastAll.apply(*([m.weight for m in part] + [m.bias for m in heads]))
            for m, w in zip(part, halves):
                m.__dict__['_w_half'] = w
            for m, b in zip(heads, halves[len(part):]):
                m.__dict__['_b_half'] = b
        return self

    def __exit__(self, *exc):
        for m in self.convs:
            m.__dict__.pop('_w_half', None)
            m.__dict__.pop('_b_half', None)
        return False


class HeadConv2d(nn.Conv2d):
    """nn.Conv2d (the reference's predict_flow heads, pwc_tf.py:93-94: bias inside the convolution, no activation) that
    contracts with the pass's bf16 shadows when there are any; same parameters, same state-dict keys."""

    def forward(self, x):
        b = self.__dict__.get('_b_half')
        return self._conv_forward(x, conv_weight(self), self.bias if b is None else b)


class ConvLeaky(nn.Sequential):
    """The reference's conv() block, Sequential(Conv2d(bias=True), LeakyReLU(0.1)): same children, same
    ``<name>.0.weight`` / ``<name>.0.bias`` state-dict keys.  On a HIP tensor the forward runs the
    contraction bias-free on MFMA (PyTorch-ROCm / MIOpen) and applies bias + LeakyReLU in one in-place
    HIP pass (csrc/elementwise.hip, elementwise_bf16.hip); the backward of that pass also produces the bias gradient."""

    def forward(self, x, consumers=1):
        """``consumers=2``: returns (y, y) -- two handles of the same activation, one per consumer, so the two
        gradients meet inside the epilogue's backward kernel (ops.bias_leaky_relu_)."""
        c = self[0]
        y = F.conv2d(x, conv_weight(c), None, c.stride, c.padding, c.dilation, c.groups)
        # fp32, or bf16 under the autocast of cfg.precision == 'bf16' (bias stays fp32): the same fused epilogue
        return ops.bias_leaky_relu_(y, c.bias, self[1].negative_slope, consumers)


CL = torch.channels_last


def weights_to_channels_last(module):
    """Store every 4-d convolution weight of ``module`` in channels_last order (same shape, same state-dict keys and
    values; ``load_state_dict`` copies into it whatever the source layout).  With channels_last activations MIOpen is
    called with NHWC tensors and would otherwise re-lay the weight out on every call."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=CL)
    return module


def conv(in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1):
    """Conv2d(bias) + LeakyReLU(0.1) (reference net_utils.py:7-11)."""
    return ConvLeaky(
        nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                  padding=padding, dilation=dilation, bias=True),
        nn.LeakyReLU(0.1))


def deconv(in_planes, out_planes, kernel_size=4, stride=2, padding=1):
    """reference net_utils.py:13-14 -- kept for the symbol; unused on the flow path."""
    return nn.ConvTranspose2d(in_planes, out_planes, kernel_size, stride, padding, bias=True)


def warp_flow(x, flow, use_mask=False, align_corners=False):
    """Warp x (im2) back to im1 by the optical flow (reference net_utils.py:16-54).

    x: [B, C, H, W], flow: [B, 2, H, W] -> [B, C, H, W].  One HIP kernel (csrc/warp.hip); with
    ``use_mask`` the binary validity mask of net_utils.py:47-52 is produced in the same pass and
    multiplied in.  ``align_corners`` picks the grid_sample generation (False: torch >= 1.3, the
    reference as it runs today; True: its torch==1.2.0 pin).
    """
    return ops.warp_flow(x, flow, use_mask=use_mask, align_corners=align_corners)
