"""Experiment: weight-gradient convolutions with NCHW vs channels_last operands (MIOpen picks NHWC implicit-GEMM
kernels for most of them and transposes NCHW operands around the kernel).  GPU only.
    python tools/wrw_layout_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MIOPEN_USER_DB_PATH', '/tmp/miopen_probe_db')
os.makedirs(os.environ['MIOPEN_USER_DB_PATH'], exist_ok=True)
torch.backends.cudnn.benchmark = True

# (N, Cin, Cout, Hout, Wout, stride, dilation)
SHAPES = [(16, 115, 128, 64, 208, 1, 1), (16, 128, 128, 64, 208, 1, 1), (16, 256, 96, 64, 208, 1, 1),
          (16, 224, 64, 64, 208, 1, 1), (16, 160, 32, 64, 208, 1, 1), (16, 34, 128, 64, 208, 1, 1),
          (16, 128, 128, 64, 208, 1, 2), (16, 128, 128, 64, 208, 1, 4), (16, 128, 96, 64, 208, 1, 8),
          (16, 96, 64, 64, 208, 1, 16), (16, 64, 32, 64, 208, 1, 1),
          (24, 3, 16, 128, 416, 2, 1), (24, 16, 16, 128, 416, 1, 1), (24, 16, 32, 64, 208, 2, 1), (24, 32, 32, 64, 208, 1, 1),
          (16, 147, 128, 32, 104, 1, 1), (16, 128, 128, 32, 104, 1, 1), (16, 256, 96, 32, 104, 1, 1)]


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {'nchw': 0.0, 'nhwc': 0.0, 'nhwc_dgrad': 0.0, 'nchw_dgrad': 0.0, 'nchw_fwd': 0.0, 'nhwc_fwd': 0.0}
for (N, ci, co, ho, wo, st, dil) in SHAPES:
    hi, wi = ho * st, wo * st
    x = torch.randn(N, ci, hi, wi, device='cuda')
    dy = torch.randn(N, co, ho, wo, device='cuda')
    w = torch.randn(co, ci, 3, 3, device='cuda')
    xl, dyl, wl = (t.contiguous(memory_format=torch.channels_last) for t in (x, dy, w))

    def wrw(a, b, c):
        return torch.ops.aten.convolution_backward(b, a, c, None, [st, st], [dil, dil], [dil, dil], False, [0, 0], 1, [False, True, False])

    def dgrad(a, b, c):
        return torch.ops.aten.convolution_backward(b, a, c, None, [st, st], [dil, dil], [dil, dil], False, [0, 0], 1, [True, False, False])

    def fwd(a, c):
        return torch.nn.functional.conv2d(a, c, None, st, dil, dil)
    t0 = time.time()
    r = {'nchw': timeit(lambda: wrw(x, dy, w)), 'nhwc': timeit(lambda: wrw(xl, dyl, wl)),
         'nchw_dgrad': timeit(lambda: dgrad(x, dy, w)) if ci > 3 else 0.0, 'nhwc_dgrad': timeit(lambda: dgrad(xl, dyl, wl)) if ci > 3 else 0.0,
         'nchw_fwd': timeit(lambda: fwd(x, w)), 'nhwc_fwd': timeit(lambda: fwd(xl, wl))}
    for k in r:
        tot[k] += r[k]
    gf = 2.0 * N * ci * co * 9 * ho * wo / 1e12
    print('N%d %3d->%3d %dx%d s%d d%-2d | wrw nchw %7.1f us (%5.1f TF/s) nhwc %7.1f us (%5.1f) | dgrad %7.1f / %7.1f | fwd %7.1f / %7.1f  [%.0fs]' % (
        N, ci, co, ho, wo, st, dil, r['nchw'], gf / r['nchw'] * 1e6, r['nhwc'], gf / r['nhwc'] * 1e6,
        r['nchw_dgrad'], r['nhwc_dgrad'], r['nchw_fwd'], r['nhwc_fwd'], time.time() - t0), flush=True)
print('totals (us):', {k: round(v, 1) for k, v in tot.items()})
