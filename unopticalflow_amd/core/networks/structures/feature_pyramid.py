import torch.nn as nn

from .net_utils import conv

_CHANNELS = (16, 32, 64, 96, 128, 196)


class FeaturePyramid(nn.Module):
    """Six (stride-2, stride-1) conv pairs; returns the six stride-1 outputs
    (reference feature_pyramid.py:8-36).  Module names conv1..conv12 fix the checkpoint keys."""

    def __init__(self):
        super(FeaturePyramid, self).__init__()
        cin = 3
        for lvl, cout in enumerate(_CHANNELS):
            self.add_module('conv%d' % (2 * lvl + 1), conv(cin, cout, kernel_size=3, stride=2))
            self.add_module('conv%d' % (2 * lvl + 2), conv(cout, cout, kernel_size=3, stride=1))
            cin = cout

    def forward(self, img):
        outs, t, last = [], img, len(_CHANNELS) - 1
        for lvl in range(len(_CHANNELS)):
            t = getattr(self, 'conv%d' % (2 * lvl + 1))(t)
            if lvl < last:      # two consumers (the next level and the caller): one handle each, see ConvLeaky
                t, out = getattr(self, 'conv%d' % (2 * lvl + 2))(t, 2)
            else:
                t = out = getattr(self, 'conv%d' % (2 * lvl + 2))(t)
            outs.append(out)
        return tuple(outs)
