import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['UNFLOW_MICROBENCH_TUNING'] = '1'
from unopticalflow_amd import ops, _lib, build as _b
_lib.LIB_PATH = _b.LIB_TUNING
from oracle import ref_cpu as R
lib = _lib.load(); P = ops._ptr
for (B, C, h, w) in ((2, 1, 256, 256), (2, 3, 260, 256), (2, 5, 64, 64), (1, 1, 8, 64), (1, 4, 8, 64), (1, 1, 64, 64)):
    g = torch.Generator().manual_seed(1)
    f1c = torch.randn(B, C, h, w, generator=g).requires_grad_(); f2c = torch.randn(B, C, h, w, generator=g).requires_grad_()
    cv = R.corr_naive(f1c, f2c, 4); go = torch.randn(cv.shape, generator=g); cv.backward(go)
    f1, f2, gg = f1c.detach().cuda(), f2c.detach().cuda(), go.cuda()
    for fb in (6, 1, 3, 4, 5):
        os.environ['UNFLOW_CORR_BWD'] = str(fb)
        gf1, gf2 = torch.full_like(f1, 7.0), torch.full_like(f2, 7.0)
        lib.unflow_corr_bwd(P(f1), P(f2), P(gg), P(gf1), P(gf2), B, C, h, w, 4, ops._stream())
        torch.cuda.synchronize()
        e1 = (gf1.cpu() - f1c.grad).abs(); e2 = (gf2.cpu() - f2c.grad).abs()
        print((B, C, h, w), 'fb', fb, 'max err gf1 %.3e gf2 %.3e' % (e1.max(), e2.max()), 'n bad', int((e1 > 1e-4).sum()), int((e2 > 1e-4).sum()),
              'first bad', (e1 > 1e-4).nonzero()[:2].tolist(), flush=True)
