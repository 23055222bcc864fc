"""Kernel-exact times (event pairs attached to the kernels, as bench.py's roofline legs) of the loss-section entry points at the three
scales of the 832x256, B=8 step (2B = 16 directed flows, 8 centre images).  python tools/probes/loss_kernel_times.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unopticalflow_amd import ops, _lib  # noqa: E402
if os.environ.get('UNFLOW_LIB_PATH'):                      # a variant build (python -m unopticalflow_amd.build --tuning with UNFLOW_TUNING_TAG)
    _lib.LIB_PATH = os.environ['UNFLOW_LIB_PATH']
elif os.environ.get('UNFLOW_MICROBENCH_TUNING') == '1':
    from unopticalflow_amd import build as _b
    _lib.LIB_PATH = _b.LIB_TUNING

torch.manual_seed(0)
for s in range(3):
    H, W = 256 >> s, 832 >> s
    img = torch.rand(8, 3, H, W, device='cuda')
    warped = torch.rand(16, 3, H, W, device='cuda', requires_grad=True)
    flow = (torch.randn(16, 2, H, W, device='cuda') * 3).requires_grad_()
    ops.kernel_timer.enable(True)
    for _ in range(int(os.environ.get('UNFLOW_PROBE_ITERS', '12'))):
        diff, wgt = ops.occ_weight_stacked(img, warped)
        loss = ops.masked_mean(diff, wgt).sum() + ops.ssim_loss(img, warped, wgt).sum() + ops.smooth2_loss(flow, img).sum() + \
            ops.consis_loss(flow[8:], flow[:8], wgt[8:]).sum()
        loss.backward()
    torch.cuda.synchronize()
    ops.kernel_timer.disable()
    for r in ops.kernel_timer.rows():
        print('scale %d %-26s %7.1f us  %s GB/s' % (s, r['entry'], r['avg_us'], r['algorithmic_GBps']), flush=True)
