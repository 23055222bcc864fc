"""Diagnostic: the channels_last decoder with epilogue-filled cat buffers against the torch.cat form -- largest differences per
loss and per gradient tensor (relative to the tensor's largest gradient), and the same for two runs of the cat form (the
run-to-run level).  python tools/probes/fill_vs_cat.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R                                                # noqa: E402
from unopticalflow_amd import get_model, generate_loss_weights_dict            # noqa: E402


def run(prec, fill):
    cfg = R.default_cfg(precision=prec, channels_last=True)
    model = get_model('flow')(cfg).cuda()
    model.load_state_dict(R.seeded_state_dict(model, 1234, 0.25))
    model.pwc_model.fill_cat_buffers = fill
    w = generate_loss_weights_dict(cfg)
    x = R.synthetic_triplets(2, 128, 128, seed=0, structured=True).cuda()
    pack = model(x)
    sum(w[k] * pack[k].mean() for k in pack).backward()
    return ({k: v.detach().float().cpu() for k, v in pack.items()}, {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()})


for prec in ('fp32', 'bf16'):
    a, b, c = run(prec, True), run(prec, False), run(prec, False)
    for tag, (u, v) in (('fill vs cat', (a, b)), ('cat vs cat ', (b, c))):
        lo = max(((u[0][k] - v[0][k]).abs() / v[0][k].abs()).max().item() for k in u[0])
        worst = max(((u[1][n] - v[1][n]).abs().max().item() / max(v[1][n].abs().max().item(), 1e-12), n) for n in u[1])
        print('%s %s: worst loss rel diff %.2e; worst gradient diff / max|grad| %.2e (%s)' % (prec, tag, lo, worst[0], worst[1]), flush=True)
