"""What two forms of the same network differ by when MIOpen runs on its MEASURED picks (the shipped find-db: the product
configuration of bench.py / train.py), at the shape the db covers (832x256, B = 8):

  fill vs cat      channels_last decoder with epilogue-filled cat buffers against the torch.cat form (fp32 and bf16)
  same vs same     the cat form twice: the run-to-run level
  shadows vs casts bf16: one multi-tensor weight cast per pass against autocast's cast per convolution call

Prints one JSON object: per comparison the worst relative loss difference and the worst gradient difference in units of the
tensor's largest gradient.  tests/test_hip_model.py::test_two_forms_agree_under_the_find_db runs it as a child process (the find
mode is process-global) and holds the numbers to bars.

    python tools/probes/finddb_run_to_run.py [--batch 8 --hw 256 832]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R                                                # noqa: E402
from unopticalflow_amd import get_model, generate_loss_weights_dict, tuning     # noqa: E402


def run(x, prec, fill=True, shadows=True):
    cfg = R.default_cfg(precision=prec, channels_last=True, weight_shadows=shadows)
    model = get_model('flow')(cfg).cuda()
    model.load_state_dict(R.seeded_state_dict(model, 1234, 0.25))
    model.pwc_model.fill_cat_buffers = fill
    model.pwc_model.fused_head = fill or prec == 'bf16'
    model.pwc_model.fused_upsample = fill
    w = generate_loss_weights_dict(cfg)
    for _ in range(2):                                   # (the second pass: MIOpen's picks are settled)
        for p in model.parameters():
            p.grad = None
        pack = model(x)
        sum(w[k] * pack[k].mean() for k in pack).backward()
    torch.cuda.synchronize()
    return ({k: v.detach().float().cpu() for k, v in pack.items()}, {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()})


def compare(u, v):
    lo = max(((u[0][k] - v[0][k]).abs() / v[0][k].abs().clamp_min(1e-12)).max().item() for k in u[0])
    worst = max(((u[1][n] - v[1][n]).abs().max().item() / max(v[1][n].abs().max().item(), 1e-12), n) for n in u[1])
    return {'loss_rel': lo, 'grad_over_max': worst[0], 'worst_tensor': worst[1]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--hw', type=int, nargs=2, default=[256, 832])
    a = ap.parse_args()
    tuning.enable_miopen_tuning()
    out = {'find_db_in_use': tuning.default_channels_last(), 'batch': a.batch, 'hw': a.hw}
    x = R.synthetic_triplets(a.batch, a.hw[0], a.hw[1], seed=0, structured=True).cuda()
    for prec in ('fp32', 'bf16'):
        f, c1, c2 = run(x, prec, True), run(x, prec, False), run(x, prec, False)
        out[prec + ' fill vs cat'] = compare(f, c1)
        out[prec + ' cat vs cat'] = compare(c1, c2)
        out[prec + ' fill vs fill'] = compare(f, run(x, prec, True))
    s1, s0 = run(x, 'bf16', True, True), run(x, 'bf16', True, False)
    out['bf16 shadows vs casts'] = compare(s1, s0)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
