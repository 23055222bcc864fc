import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from .net_utils import conv, conv_weight, HeadConv2d, warp_flow, weights_to_channels_last, CL

_DD = (128, 128, 96, 64, 32)                       # decoder widths (reference pwc_tf.py:25)
_FEAT = {6: 0, 5: 128, 4: 96, 3: 64, 2: 32}        # channels of the pyramid level fed to each decoder


class PWC_tf(nn.Module):
    """PWC-style coarse-to-fine flow decoder (reference pwc_tf.py:17-179).

    ``self.corr`` is the reference's own plug point (pwc_tf.py:19-20); here it is the HIP cost
    volume.  Parameter names (conv{6..2}_{0..4}, predict_flow{6..2}, dc_conv{1..7}) match the
    reference checkpoint.
    """

    def __init__(self, md=4, align_corners=False, fused_warp_corr=False, channels_last=False):
        super(PWC_tf, self).__init__()
        # True: the decoder and context convolutions run on channels_last (NHWC) tensors (MIOpen's implicit-GEMM
        # solvers are NHWC kernels; on NCHW tensors each call is wrapped in transposes).  Cost volume, warp, flows and
        # everything outside this module stay NCHW: the decoder input is re-laid out once per level, the 2-channel
        # flows once on the way out.
        self.channels_last = bool(channels_last)
        # channels_last only: the decoder's cat((x_k, x_k+1)) inputs are filled by the producing convolutions' epilogues
        # (_decoder_filled) instead of being copied together by torch.cat; False keeps the cat form (tests compare the two)
        self.fill_cat_buffers = True
        self.fused_head = True            # flow heads: bias + NHWC -> NCHW + residual as one kernel each way (ops.flow_head)
        self.fused_upsample = True        # flow up-sampling + its scale factor as one kernel each way (ops.upsample_bilinear_scaled)
        self.corr = self.corr_naive
        self.corr_backward = 'auto'       # arithmetic of the cost volume's backward pass, per module (ops.CORR_BACKWARD_MODES; bench.py --corr-bwd)
        # True: warp + cost volume of a level as ONE kernel (ops.warp_corr; the warped features never reach HBM).
        # Measured on MI355X (profiles/r2_v1_bench_fused1.json): at parity with the two separate kernels at level 2 and
        # slower below it -- the cost-volume kernel is LDS/VALU-bound, so the warp stage adds to the bound resource --
        # hence off by default.
        self.fused_warp_corr = bool(fused_warp_corr)
        # decoder levels whose warp + cost volume run as the ONE fused kernel (ops.warp_corr): all four with fused_warp_corr, or a chosen few --
        # (the fused kernel lost with levels 2-4 all fused; single levels were never measured alone -- level 5's width 26 is not a multiple of 4: not served)
        self.fused_levels = frozenset((2, 3, 4, 5)) if self.fused_warp_corr else frozenset()
        self.leakyRELU = nn.LeakyReLU(0.1)
        self.align_corners = align_corners
        nd = (2 * md + 1) ** 2
        for lvl in (6, 5, 4, 3, 2):
            od = nd if lvl == 6 else nd + _FEAT[lvl] + 2
            cins = (od, _DD[0], _DD[0] + _DD[1], _DD[1] + _DD[2], _DD[2] + _DD[3])
            for k, cin in enumerate(cins):
                self.add_module('conv%d_%d' % (lvl, k), conv(cin, _DD[k], kernel_size=3, stride=1))
            self.add_module('predict_flow%d' % lvl, self.predict_flow(_DD[3] + _DD[4]))
        ctx = ((_DD[4] + 2, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16), (64, 32, 1))
        for k, (cin, cout, dil) in enumerate(ctx):
            self.add_module('dc_conv%d' % (k + 1), conv(cin, cout, kernel_size=3, stride=1, padding=dil, dilation=dil))
        self.dc_conv7 = self.predict_flow(32)
        if self.channels_last:
            weights_to_channels_last(self)

    def _cl(self, x):
        return self.channels_last and ops.on_device(x) and x.dtype in (torch.float32, torch.bfloat16)

    def _cat(self, parts):
        """The decoder input torch.cat(parts, 1) (pwc_tf.py:113): written directly in channels_last order when the conv
        stack runs in it (one kernel instead of a cat and a re-layout)."""
        if self._cl(parts[0]):
            # (bf16 option: the cat of the fp32 parts is rounded to bf16 in the same pass -- the values autocast would hand the
            # convolution after torch.cat's fp32 result)
            half = parts[0].is_cuda and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            return ops.cat_channels_last([p.float() for p in parts], torch.bfloat16 if half else torch.float32)
        return parts[0] if len(parts) == 1 else torch.cat(parts, 1)

    def _up(self, flow, size, mul):
        """mul * bilinear up-sampling of a flow: one HIP kernel each way for the integer factors of the pyramid (ops.upsample_bilinear_scaled),
        the reference's torch expression for what the kernel does not take (a non-integer ratio; host tensors in module-level tests --
        the model as a whole has no CPU path: its cost volume, warp and loss operators refuse host tensors)."""
        h, w = flow.shape[2], flow.shape[3]
        if self.fused_upsample and ops.on_device(flow) and flow.dtype == torch.float32 and size[0] % h == 0 and size[1] % w == 0 and size[0] >= h and size[1] >= w:
            return ops.upsample_bilinear_scaled(flow, size, mul)
        return F.interpolate(flow * mul, list(size), mode='bilinear') if mul == 4.0 else F.interpolate(flow, list(size), mode='bilinear') * mul

    def _head(self, m, x, residual=None):
        """``predict_flow(x) [+ residual]`` (pwc_tf.py:118,130,143,155,167,171) as the fp32 NCHW flow.  On a channels_last stack: the
        bias-free contraction, then bias, re-layout and residual in ONE pass (ops.flow_head) instead of ATen's bias add, copy and add."""
        if self.fused_head and self._cl(x):
            y = F.conv2d(x, conv_weight(m), None, m.stride, m.padding, m.dilation, m.groups)
            return ops.flow_head(y, m.bias, residual)
        flow = m(x).float().contiguous()
        return flow if residual is None else flow + residual

    def predict_flow(self, in_planes):
        return HeadConv2d(in_planes, 2, kernel_size=3, stride=1, padding=1, bias=True)

    def warp(self, x, flow):
        return warp_flow(x.float(), flow.float(), use_mask=False, align_corners=self.align_corners)

    def corr_naive(self, input1, input2, d=4):
        """Same contract as the reference's corr_naive (pwc_tf.py:97-106); one HIP kernel (fp32 accumulation;
        bf16 features of an autocast run are widened first)."""
        return ops.corr(input1.float(), input2.float(), d, backward=self.corr_backward)

    def _decoder(self, lvl, x, residual=None):
        """``x``: the tuple of tensors the reference concatenates into the decoder input.
        reference pwc_tf.py:113-118 (and the same six lines per level).  Every activation feeds two
        consumers; it is taken as two handles (ConvLeaky(consumers=2)) so the gradients are summed inside
        the epilogue's backward kernel."""
        c = [getattr(self, 'conv%d_%d' % (lvl, k)) for k in range(5)]
        x = self._cat(x)
        if self._cl(x) and self.fill_cat_buffers:
            return self._decoder_filled(lvl, c, x, residual)
        x0, x0b = c[0](x, 2)
        x1, x1b = c[1](x0, 2)
        x2, x2b = c[2](torch.cat((x0b, x1), 1), 2)
        x3, x3b = c[3](torch.cat((x1b, x2), 1), 2)
        if lvl == 2:                                  # x4 of level 2 also feeds the context network
            x4, x4b = c[4](torch.cat((x2b, x3), 1), 2)
        else:
            x4 = x4b = c[4](torch.cat((x2b, x3), 1))
        return self._head(getattr(self, 'predict_flow%d' % lvl), torch.cat((x3b, x4), 1), residual), x4b

    def _decoder_filled(self, lvl, c, x, residual=None):
        """The same five convolutions on channels_last tensors without a single torch.cat: every conv input
        cat((x_k, x_k+1)) (pwc_tf.py:114-118) is ONE buffer whose two channel ranges are written by the epilogues of the
        convolutions that produce x_k and x_k+1 (ops.bias_leaky_relu_into); an activation is written to the (at most two)
        buffers that hold it, so per activation 1 read + 2 writes replace the in-place epilogue's 1 + 1 and two cats' 2 + 2."""
        def raw(m, t):                                   # the bias-free contraction of a conv() block
            k = m[0]
            return F.conv2d(t, conv_weight(k), None, k.stride, k.padding, k.dilation, k.groups), k.bias, m[1].negative_slope

        def buf(like, ch):
            return torch.empty((like.shape[0], ch) + tuple(like.shape[2:]), dtype=like.dtype, device=like.device, memory_format=CL)
        d0, d1, d2, d3, d4 = _DD
        y, b, sl = raw(c[0], x)
        x0, b01, _ = ops.bias_leaky_relu_into(y, b, sl, buf(y, d0 + d1), 0, inplace=True)        # x0: conv1's input and cat(x0, x1)
        y, b, sl = raw(c[1], x0)
        _, b01, b12 = ops.bias_leaky_relu_into(y, b, sl, b01, d0, buf(y, d1 + d2), 0)            # x1 -> cat(x0, x1), cat(x1, x2)
        y, b, sl = raw(c[2], b01)
        _, b12, b23 = ops.bias_leaky_relu_into(y, b, sl, b12, d1, buf(y, d2 + d3), 0)            # x2 -> cat(x1, x2), cat(x2, x3)
        y, b, sl = raw(c[3], b12)
        _, b23, b34 = ops.bias_leaky_relu_into(y, b, sl, b23, d2, buf(y, d3 + d4), 0)            # x3 -> cat(x2, x3), cat(x3, x4)
        y, b, sl = raw(c[4], b23)
        x4, b34, _ = ops.bias_leaky_relu_into(y, b, sl, b34, d3, inplace=(lvl == 2))             # x4 (level 2: also the context network's input)
        return self._head(getattr(self, 'predict_flow%d' % lvl), b34, residual), x4

    def forward(self, feature_list_1, feature_list_2, img_hw):
        f1 = dict(zip(range(1, 7), feature_list_1))
        f2 = dict(zip(range(1, 7), feature_list_2))
        flow, _ = self._decoder(6, (self.corr(f1[6], f2[6]),))
        level_flow = {}
        for lvl in (5, 4, 3, 2):
            up = self._up(flow, (2 * flow.shape[2], 2 * flow.shape[3]), 2.0)         # F.interpolate(flow, scale_factor=2.0, 'bilinear') * 2.0 (pwc_tf.py:119)
            if lvl in self.fused_levels and self.corr == self.corr_naive:   # (a user-supplied self.corr keeps the two-op path)
                cv = ops.warp_corr(f1[lvl].float(), f2[lvl].float(), up.float(), 4, self.align_corners)
            else:
                cv = self.corr(f1[lvl], self.warp(f2[lvl], up))
            flow, x4 = self._decoder(lvl, (cv, f1[lvl], up), up)             # flow = predict_flow(x) + up_flow (pwc_tf.py:130,143,155,167)
            level_flow[lvl] = flow
        fl2 = level_flow[2].contiguous(memory_format=CL) if self._cl(x4) else level_flow[2]
        x = self.dc_conv4(self.dc_conv3(self.dc_conv2(self.dc_conv1(torch.cat([fl2, x4], 1)))))
        level_flow[2] = self._head(self.dc_conv7, self.dc_conv6(self.dc_conv5(x)), level_flow[2])      # flow2 + dc_conv7(...) (pwc_tf.py:171)
        img_h, img_w = img_hw[0], img_hw[1]
        # F.interpolate(flow * 4.0, size, 'bilinear') (pwc_tf.py:174-177): the factor 4.0 commutes with the interpolation exactly
        return [self._up(level_flow[lvl], (img_h // (1 << k), img_w // (1 << k)), 4.0) for k, lvl in enumerate((2, 3, 4, 5))]
