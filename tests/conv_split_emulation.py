"""What would convolutions on the bf16 matrix cores with SPLIT operands do to the parity bars?  (CPU emulation, no GPU.)

DESIGN.md section 7 "next 4": the train step is convolution-bound (MIOpen's fp32 implicit-GEMM kernels at ~0.65 of the fp32 MFMA peak,
157 TFLOP/s); the bf16 MFMA peak is 16x that, so a convolution whose operands are split into bf16 parts -- the same arithmetic the
cost-volume backward of csrc/corr_mfma.h uses -- has 5x (three products: hi.hi + hi.lo + lo.hi) or 2.7x (six products, three parts per
operand) the fp32 peak to work with.  Before anyone writes that kernel: does the model stay inside north_star's "flow / loss within 1e-4
rel fp32" with it?  A product of two bf16 numbers is exact in fp32 and the MFMA accumulates in fp32, so an fp32 CPU convolution of the
split parts IS that arithmetic up to summation order.  This script swaps every convolution of the CPU oracle (forward, data gradient
and weight gradient) for the split form and compares one train step with the plain fp32 oracle at the quantities and bars
tests/test_hip_model.py holds the HIP model to.

    python tests/conv_split_emulation.py [--size 128 128] [--batch 2] [--steps 3] [--out profiles/r5_conv_split_emulation.json]

It lives under tests/ because it drives oracle/ (test infrastructure); tests/test_host_logic.py runs its split helpers at a tiny size.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def split_parts(x, parts):
    """x = p0 + p1 (+ p2) + r with every p a bf16 number (round to nearest even), as the kernels split: hi = bf16(x), lo = bf16(x - hi)."""
    out, rest = [], x
    for _ in range(parts):
        p = rest.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        rest = rest - p
    return out


def product_pairs(parts):
    """Which (i, j) part products are formed: everything of order i + j < parts (parts = 2: hh, hl, lh; parts = 3: six products)."""
    return [(i, j) for i in range(parts) for j in range(parts) if i + j < parts]


class SplitConv(torch.autograd.Function):
    """conv2d whose three contractions (forward, data gradient, weight gradient) are sums of bf16-part products accumulated in fp32."""
    parts = 2
    which = ('fwd', 'dgrad', 'wgrad')

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation):
        ctx.save_for_backward(x, w)
        ctx.conf = (stride, padding, dilation, b is not None)
        if 'fwd' in SplitConv.which:
            xs, ws = split_parts(x, SplitConv.parts), split_parts(w, SplitConv.parts)
            y = None
            for i, j in reversed(product_pairs(SplitConv.parts)):          # small terms first
                t = F.conv2d(xs[i], ws[j], None, stride, padding, dilation)
                y = t if y is None else y + t
        else:
            y = F.conv2d(x, w, None, stride, padding, dilation)
        return y if b is None else y + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, padding, dilation, has_b = ctx.conf
        P = SplitConv.parts
        gx = gw = None
        if 'dgrad' in SplitConv.which:
            gs, ws = split_parts(g, P), split_parts(w, P)
            for i, j in reversed(product_pairs(P)):
                t = torch.nn.grad.conv2d_input(x.shape, ws[j], gs[i], stride, padding, dilation)
                gx = t if gx is None else gx + t
        else:
            gx = torch.nn.grad.conv2d_input(x.shape, w, g, stride, padding, dilation)
        if 'wgrad' in SplitConv.which:
            gs, xs = split_parts(g, P), split_parts(x, P)
            for i, j in reversed(product_pairs(P)):
                t = torch.nn.grad.conv2d_weight(xs[j], w.shape, gs[i], stride, padding, dilation)
                gw = t if gw is None else gw + t
        else:
            gw = torch.nn.grad.conv2d_weight(x, w.shape, g, stride, padding, dilation)
        gb = g.sum((0, 2, 3)) if has_b else None
        return gx, gw, gb, None, None, None


class patched_convolutions:
    """Within the block every nn.Conv2d forward goes through SplitConv with `parts` bf16 parts per operand."""

    def __init__(self, parts, which=('fwd', 'dgrad', 'wgrad')):
        self.parts, self.which = parts, tuple(which)

    def __enter__(self):
        self.old = torch.nn.Conv2d.forward
        SplitConv.parts, SplitConv.which = self.parts, self.which

        def fwd(m, x):
            return SplitConv.apply(x, m.weight, m.bias, m.stride, m.padding, m.dilation)
        torch.nn.Conv2d.forward = fwd
        return self

    def __exit__(self, *a):
        torch.nn.Conv2d.forward = self.old


def run(parts, H, W, B, steps, which=('fwd', 'dgrad', 'wgrad'), seed=0):
    """`steps` Adam steps of the oracle; parts = 0: plain fp32.  Returns losses, first-step pack / flows / gradients, final parameters."""
    from oracle import ref_cpu as R
    torch.manual_seed(0)
    cfg = R.default_cfg()
    model = R.Model_flow(cfg)
    model.load_state_dict(R.seeded_state_dict(model), strict=False)
    weights = R.generate_loss_weights_dict(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    inputs = R.synthetic_triplets(B, H, W, seed=seed)
    res = {'losses': []}

    def go():
        for it in range(steps):
            opt.zero_grad()
            pack = model(inputs)
            loss = R.total_loss(pack, weights)
            loss.backward()
            if it == 0:
                res['pack'] = {k: v.detach().clone() for k, v in pack.items()}
                res['grads'] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
                with torch.no_grad():
                    res['flow'] = model.inference_flow(inputs[:, :, H:2 * H], inputs[:, :, 2 * H:]).clone()
            res['losses'].append(float(loss))
            opt.step()
    if parts:
        with patched_convolutions(parts, which):
            go()
    else:
        go()
    res['params'] = {n: p.detach().clone() for n, p in model.named_parameters()}
    return res


def compare(ref, got):
    """The quantities of tests/test_hip_model.py::test_module_128_golden / test_kitti_256x832_golden, as numbers (bar in brackets)."""
    out = {}
    out['loss_terms_rel_max [1e-4]'] = max(float(((got['pack'][k] - ref['pack'][k]).abs() / ref['pack'][k].abs().clamp_min(1e-30)).max())
                                           for k in ref['pack'] if ref['pack'][k].numel() and ref['pack'][k].dtype.is_floating_point)
    out['total_loss_rel_step0 [1e-4]'] = abs(got['losses'][0] - ref['losses'][0]) / abs(ref['losses'][0])
    out['total_loss_rel_later_steps [2e-3]'] = max([abs(a - b) / abs(b) for a, b in zip(got['losses'][1:], ref['losses'][1:])] or [0.0])
    out['flow_over_max [1e-4]'] = float((got['flow'] - ref['flow']).abs().max() / ref['flow'].abs().max())
    gn = lambda d: float(torch.sqrt(sum((g.double() ** 2).sum() for g in d.values())))
    out['grad_norm_rel [5e-4]'] = abs(gn(got['grads']) - gn(ref['grads'])) / gn(ref['grads'])
    worst, worst_name = 0.0, ''
    for n, g in ref['grads'].items():
        e = float((got['grads'][n] - g).abs().max() / g.abs().max().clamp_min(1e-30))
        if e > worst:
            worst, worst_name = e, n
    out['grad_over_max_worst_tensor [1e-3..2e-3 at three tensors]'] = worst
    out['grad_worst_tensor'] = worst_name
    for n in ('fpyramid.conv1.0.weight', 'pwc_model.conv2_0.0.weight', 'pwc_model.dc_conv7.weight'):
        if n in ref['grads']:
            out['grad_over_max ' + n] = float((got['grads'][n] - ref['grads'][n]).abs().max() / ref['grads'][n].abs().max())
    pa = lambda d: float(sum(p.abs().double().sum() for p in d.values()))
    out['param_abs_rel_after_steps [5e-4]'] = abs(pa(got['params']) - pa(ref['params'])) / pa(ref['params'])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, nargs=2, default=(128, 128))
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    H, W = a.size
    ref = run(0, H, W, a.batch, a.steps)
    doc = {'size': [H, W], 'batch': a.batch, 'steps': a.steps, 'fp32_losses': ref['losses'], 'variants': {}}
    for name, parts, which in (('bf16x1 (plain bf16 operands, fp32 accumulate)', 1, ('fwd', 'dgrad', 'wgrad')),
                               ('bf16x3 (hi.hi + hi.lo + lo.hi)', 2, ('fwd', 'dgrad', 'wgrad')),
                               ('bf16x3 forward only', 2, ('fwd',)),
                               ('bf16x3 forward + data gradient', 2, ('fwd', 'dgrad')),
                               ('bf16x6 (three parts, six products)', 3, ('fwd', 'dgrad', 'wgrad'))):
        got = run(parts, H, W, a.batch, a.steps, which)
        doc['variants'][name] = compare(ref, got)
        print(name, json.dumps(doc['variants'][name], indent=1), flush=True)
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(doc, f, indent=1)


if __name__ == '__main__':
    main()
