"""How far the HIP model is from the fixtures / from its fp32 self, per check -- the numbers the tolerances in
tests/test_hip_model.py are set from (run on the MI355X box: python tools/tolerance_probe.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R                                   # noqa: E402  (a measurement tool, like tests/)
from unopticalflow_amd import get_model, generate_loss_weights_dict   # noqa: E402


def grads_vs_golden():
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g2_module_128.npz')))
    x = R.synthetic_triplets(int(g['B']), int(g['H']), int(g['W']), seed=0, structured=True).cuda()
    for cl in (True, False):
        for ac in (0, 1):
            tag = '_ac%d' % ac
            worst = []
            for rep in range(3):
                cfg = R.default_cfg(align_corners=bool(ac), channels_last=cl)
                model = get_model('flow')(cfg).cuda()
                model.load_state_dict(R.seeded_state_dict(model, 1234, float(g['flow_gain'])))
                w = generate_loss_weights_dict(cfg)
                pack = model(x)
                sum(w[k] * pack[k].mean() for k in pack).backward()
                ga = np.array([p.grad.double().abs().sum().item() for p in model.parameters()])
                gmax = np.array([p.grad.abs().max().item() for p in model.parameters()])
                ref = g['grad_abs' + tag]
                rel = np.abs(ga - ref) / ref
                score = np.abs(ga - ref) / (1e-3 * ref + 1e-3 * gmax)
                i, j = int(rel.argmax()), int(score.argmax())
                names = [n for n, _ in model.named_parameters()]
                worst.append((rel.max(), names[i], score.max(), names[j]))
            print('grad_abs cl=%s ac=%d: worst rel %s ; worst score (must be < 1) %s' % (
                cl, ac, ['%.2e %s' % (a, b) for a, b, _, _ in worst], ['%.2f %s' % (c, d) for _, _, c, d in worst]), flush=True)


def bf16_vs_fp32():
    for (B, H, W, seed) in ((2, 128, 128, 0), (8, 256, 832, 3)):
        x = R.synthetic_triplets(B, H, W, seed=seed, structured=True).cuda()
        flows, packs = {}, {}
        for prec, cl in (('fp32', True), ('bf16', False), ('bf16', True)):
            cfg = R.default_cfg(precision=prec, channels_last=cl)
            model = get_model('flow')(cfg).cuda()
            model.load_state_dict(R.seeded_state_dict(model, 1234, 0.25))
            with torch.no_grad():
                fl = model._flows(x[:, :, :H], x[:, :, H:2 * H], x[:, :, 2 * H:])
                pack = model(x)
            flows[(prec, cl)] = [f.float() for f in fl]
            packs[(prec, cl)] = {k: v.float().cpu() for k, v in pack.items()}
        ref = flows[('fp32', True)]
        for key in (('bf16', False), ('bf16', True)):
            for s in range(4):
                d = (flows[key][s] - ref[s]).abs()
                epe = (flows[key][s] - ref[s]).pow(2).sum(1).sqrt()
                print('%dx%d B=%d %s cl=%s scale %d: max|dflow| %.4f  mean EPE %.5f  max|flow| %.3f  -> max/max %.4f, epe/max %.5f' % (
                    W, H, B, key[0], key[1], s, d.max().item(), epe.mean().item(), ref[s].abs().max().item(),
                    d.max().item() / ref[s].abs().max().item(), epe.mean().item() / ref[s].abs().max().item()), flush=True)
            for k in packs[key]:
                r = (packs[key][k] / packs[('fp32', True)][k])
                print('   %s ratio to fp32: min %.4f max %.4f' % (k, r.min().item(), r.max().item()), flush=True)


if __name__ == '__main__':
    what = sys.argv[1:] or ['grads', 'bf16']
    if 'grads' in what:
        grads_vs_golden()
    if 'bf16' in what:
        bf16_vs_fp32()
