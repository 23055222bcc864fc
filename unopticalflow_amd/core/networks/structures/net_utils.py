import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops


class ConvLeaky(nn.Sequential):
    """The reference's conv() block, Sequential(Conv2d(bias=True), LeakyReLU(0.1)): same children, same
    ``<name>.0.weight`` / ``<name>.0.bias`` state-dict keys.  On a HIP tensor the forward runs the
    contraction bias-free on MFMA (PyTorch-ROCm / MIOpen) and applies bias + LeakyReLU in one in-place
    HIP pass (csrc/elementwise.hip, elementwise_bf16.hip); the backward of that pass also produces the bias gradient."""

    def forward(self, x, consumers=1):
        """``consumers=2``: returns (y, y) -- two handles of the same activation, one per consumer, so the two
        gradients meet inside the epilogue's backward kernel (ops.bias_leaky_relu_)."""
        c = self[0]
        y = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
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
