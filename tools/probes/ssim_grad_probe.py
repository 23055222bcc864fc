"""How far the SSIM / loss gradients of the HIP kernels are from the golden fixture and the oracle (sets the bars of
tests/test_hip_ops.py::test_occ_weight_and_losses_golden / test_losses_vs_oracle_random).  python tools/probes/ssim_grad_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R          # noqa: E402
from unopticalflow_amd import ops        # noqa: E402

g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g1_losses.npz')))
T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
img, fl, fr, gl = T(g['img']), T(g['from_l']).requires_grad_(), T(g['from_r']).requires_grad_(), T(g['gl'])
_, _, w_b, w_f, _, _ = ops.occ_weight(img, fl, fr)
((ops.ssim_loss(img, fr, w_f) + ops.ssim_loss(img, fl, w_b)) * gl).sum().backward()
for name, a, b in (('from_l', fl.grad, g['ls_g_from_l']), ('from_r', fr.grad, g['ls_g_from_r'])):
    a = a.cpu().numpy(); s = np.abs(b).max()
    d = np.abs(a - b)
    big = np.abs(b) > 1e-2 * s
    print('golden ssim grad %s: max|d|/max %.2e; max rel where |ref| > 1%% of max %.2e; max over (|d| - 2e-4|ref|)/max %.2e' % (
        name, d.max() / s, (d[big] / np.abs(b[big])).max(), (d - 2e-4 * np.abs(b)).max() / s))
rng = np.random.default_rng(0)
for (B, h, w) in ((2, 64, 208), (1, 256, 832), (3, 33, 70)):
    img = torch.from_numpy(rng.random((B, 3, h, w), dtype=np.float32))
    l = torch.from_numpy(rng.random((B, 3, h, w), dtype=np.float32)); r = torch.from_numpy(rng.random((B, 3, h, w), dtype=np.float32))
    glv = torch.from_numpy(rng.standard_normal(B).astype(np.float32))
    lc, rc = l.clone().requires_grad_(), r.clone().requires_grad_()
    d_l, d_r, wb, wf, _, _ = R.diff_weight(img, lc, rc)
    ((R.ssim_loss(img, rc, wf) + R.ssim_loss(img, lc, wb)) * glv).sum().backward()
    lg, rg = l.cuda().requires_grad_(), r.cuda().requires_grad_()
    _, _, Wb, Wf, _, _ = ops.occ_weight(img.cuda(), lg, rg)
    ((ops.ssim_loss(img.cuda(), rg, Wf) + ops.ssim_loss(img.cuda(), lg, Wb)) * glv.cuda()).sum().backward()
    for name, a, b in (('l', lg.grad, lc.grad), ('r', rg.grad, rc.grad)):
        a = a.cpu().numpy(); b = b.numpy(); s = np.abs(b).max(); d = np.abs(a - b); big = np.abs(b) > 1e-2 * s
        print('oracle [%d,3,%d,%d] %s: max|d|/max %.2e; max rel (|ref|>1%%) %.2e; max (|d| - 2e-4|ref|)/max %.2e' % (
            B, h, w, name, d.max() / s, (d[big] / np.abs(b[big])).max(), (d - 2e-4 * np.abs(b)).max() / s))
