import torch
import torch.nn as nn

from .... import ops
from .net_utils import conv, weights_to_channels_last, CL

_CHANNELS = (16, 32, 64, 96, 128, 196)


class FeaturePyramid(nn.Module):
    """Six (stride-2, stride-1) conv pairs; returns the six stride-1 outputs
    (reference feature_pyramid.py:8-36).  Module names conv1..conv12 fix the checkpoint keys."""

    def __init__(self, channels_last=False):
        """``channels_last``: run the twelve convolutions on channels_last (NHWC) tensors -- MIOpen's implicit-GEMM
        solvers are NHWC kernels and need no transposes then -- and hand the features out as plain NCHW tensors (what
        the cost volume / warp kernels and the decoder's cat read)."""
        super(FeaturePyramid, self).__init__()
        self.channels_last = bool(channels_last)
        cin = 3
        for lvl, cout in enumerate(_CHANNELS):
            self.add_module('conv%d' % (2 * lvl + 1), conv(cin, cout, kernel_size=3, stride=2))
            self.add_module('conv%d' % (2 * lvl + 2), conv(cout, cout, kernel_size=3, stride=1))
            cin = cout
        if self.channels_last:
            weights_to_channels_last(self)

    def forward(self, img, dup_tail=0, split_head=0):
        """``dup_tail`` = d: every returned level (but the unused first) carries d extra samples, copies of its last d -- the train
        step's (left | right | centre) batch comes back as (left | right | centre | centre), both decoder inputs as views.
        ``split_head`` = h > 0 (channels_last stacks only): those levels come back as PAIRS (first h samples, the rest) straight from
        the hand-off (ops.to_nchw_split) -- the caller's split, without the gradient concatenation it costs on the way back."""
        cl = self.channels_last and ops.on_device(img) and img.dtype == torch.float32      # (under bf16 autocast the convs then produce bf16 NHWC)
        outs, t, last = [], (img.contiguous(memory_format=CL) if cl else img), len(_CHANNELS) - 1
        for lvl in range(len(_CHANNELS)):
            t = getattr(self, 'conv%d' % (2 * lvl + 1))(t)
            if lvl < last:      # two consumers (the next level and the caller): one handle each, see ConvLeaky
                t, out = getattr(self, 'conv%d' % (2 * lvl + 2))(t, 2)
            else:
                t = out = getattr(self, 'conv%d' % (2 * lvl + 2))(t)
            # level 1 is never read by the decoder (pwc_tf.py:108-179): it stays as it is
            if cl and lvl > 0:
                out = ops.to_nchw_split(out, split_head, dup_tail) if split_head else ops.to_nchw(out, dup_tail)
            elif dup_tail and lvl > 0:
                out = torch.cat((out, out[out.shape[0] - dup_tail:]), 0)
            outs.append(out)
        return tuple(outs)
