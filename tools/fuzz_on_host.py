"""Randomised shapes through the operators that have not run on a GPU yet, on the build host -- TEST INFRASTRUCTURE (oracle + host-executed
kernel library: tests/hostexec.py; nothing of the product imports this).

    python tools/fuzz_on_host.py [--minutes M] [--seed S] [--what ms,warp,mfma,corr,fused,smallrows]

For M minutes draws shapes and checks, over the product's own autograd wrappers (ops.py) and the REAL kernel sources executed with lanes as
fibers:
  ms     ops.multiscale_losses (1-4 scales, B 1-3, even widths 4-90, heights 3-40, flat / zeroed regions, both consistency forms, inside and
         outside deferred_loss_sums) == the scale-by-scale operators BIT FOR BIT in every loss, saved sum and gradient
  warp   ops.warp_flow_masked_pyramid == ops.warp_flow_masked per scale bit for bit, forward and flow gradient, both conventions, five flow kinds
  mfma   the matrix-core cost-volume backward (ops.corr(..., backward='mfma')) against the oracle's autograd of
         corr_naive at the GPU test's bar, at shapes the dispatch serves (W % 4 == 0, C % 16 == 0, >= 8192 pixels, H >= 4 d), d = 4 and 8
  corr   the fp32 cost volume through its dispatch at random shapes (any W, C), d in {1, 2, 4, 8}, against the oracle at the GPU tests' bars
  fused  ops.warp_corr (the fused warp + cost volume, W % 4 == 0) against corr_naive(f1, warp_flow(f2, flow)) forward and backward, five flow kinds
  smallrows  round 6's small-map cost-volume backward (ops.corr(..., backward='fp32_next'), csrc/corr_small_rows.h) against the oracle, maps of <= 1024 pixels
Under AddressSanitizer + UBSan (every global and LDS access of every lane, random shapes): build the host library instrumented into a directory
of its own and preload the runtime --
    export UNFLOW_HOSTEXEC_DIR=/tmp/hxasan UNFLOW_HOSTEXEC_FLAGS="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -shared-libsan"
    LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
        python tools/fuzz_on_host.py --minutes 30
Prints one line per failure (with the draw that reproduces it) and a summary; exit code = number of failures (capped at 100)."""
import argparse
import contextlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import hostexec  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402
from unopticalflow_amd import ops  # noqa: E402


def tensor(rng, shape, scale=1.0, uniform=False):
    a = rng.random(shape, dtype=np.float32) if uniform else rng.standard_normal(shape).astype(np.float32)
    return torch.from_numpy(a * np.float32(scale))


def flow_of(rng, B, h, w, kind):
    f = tensor(rng, (B, 2, h, w), {'smooth': 0.7, 'mixed': 3.0, 'edge': 1.0, 'outside': 1.0, 'noise': 8.0}[kind])
    if kind == 'outside':
        f += torch.from_numpy(rng.uniform(-1.5, 1.5, (B, 2, 1, 1)).astype(np.float32)) * max(h, w)
    if kind == 'edge':                                                      # taps exactly on pixel centres and on the border
        f = torch.round(f)
        f[:, 0, :, -1] = 0.0
        f[:, 1, -1, :] = 0.0
    return f


def fuzz_ms(rng):
    n, B = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    h, w = int(rng.integers(3, 41)) << (n - 1), int(rng.integers(2, 46)) * 2 << (n - 1)
    hs, ws = [max(h >> s, 3) for s in range(n)], [max((w >> s) // 2 * 2, 4) for s in range(n)]
    imgs = [tensor(rng, (B, 3, hs[s], ws[s]), uniform=True) for s in range(n)]
    warped0 = [(torch.cat((imgs[s], imgs[s])) + tensor(rng, (2 * B, 3, hs[s], ws[s]), 0.1)).clamp(0, 1) for s in range(n)]
    for s in range(n):
        if rng.random() < 0.5:
            warped0[s][:B, :, : max(1, hs[s] // 3), : max(1, ws[s] // 2)] = 0.0      # an invalid (all-zero) region
        if rng.random() < 0.3:
            warped0[s][B:, :, :, :] = imgs[s]                                         # identical images: flat SSIM patches, zero differences
    flows0 = [tensor(rng, (2 * B, 2, hs[s], ws[s]), 3.0 / (1 << s)) for s in range(n)]
    gl = [tensor(rng, (B,)) for _ in range(4)]
    draw = 'ms n=%d B=%d hs=%s ws=%s' % (n, B, hs, ws)
    if not ops.multiscale_supported(imgs, warped0):
        return draw, None
    res = {}
    forms = ['per scale', 'one launch, halves by offset' if rng.random() < 0.5 else 'one launch']
    deferred = rng.random() < 0.7
    for form in forms:
        wp = [t.clone().requires_grad_() for t in warped0]
        fl = [t.clone().requires_grad_() for t in flows0]
        halves = [f.split(B) for f in fl]
        fb, ff = [x[0] for x in halves], [x[1] for x in halves]
        with (ops.deferred_loss_sums if deferred else contextlib.nullcontext()):
            if form == 'per scale':
                pixel, ssim, smooth, consis = [], [], [], []
                for s in range(n):
                    diff, wgt = ops.occ_weight_stacked(imgs[s], wp[s])
                    pixel.append(ops.masked_mean(diff, wgt)); ssim.append(ops.ssim_loss(imgs[s], wp[s], wgt))
                    smooth.append(ops.smooth2_loss(fl[s], imgs[s])); consis.append(ops.consis_loss(ff[s], fb[s], wgt[B:]))
            elif form.endswith('by offset'):
                pixel, ssim, smooth, consis = ops.multiscale_losses(imgs, wp, fl)
            else:
                pixel, ssim, smooth, consis = ops.multiscale_losses(imgs, wp, fl, ff, fb)
            packed = ops.loss_combine(pixel, ssim, smooth, consis)
        sum((p * g).sum() for p, g in zip(packed, gl)).backward()
        res[form] = [t.clone() for t in packed] + [t.grad.clone() for t in wp] + [t.grad.clone() for t in fl] + [t.clone() for t in pixel + ssim + smooth + consis]
    for k, (a, b) in enumerate(zip(res[forms[0]], res[forms[1]])):
        if not torch.equal(a, b):
            return draw + ' deferred=%s %s' % (deferred, forms[1]), 'tensor %d differs by %g' % (k, float((a - b).abs().max()))
    return draw, None


def fuzz_warp(rng):
    n, B, ac = int(rng.integers(1, 5)), int(rng.integers(1, 4)), bool(rng.integers(0, 2))
    h, w = int(rng.integers(2, 41)) << (n - 1), int(rng.integers(2, 70)) << (n - 1)
    hs, ws = [max(h >> s, 2) for s in range(n)], [max(w >> s, 2) for s in range(n)]
    kind = ('smooth', 'mixed', 'edge', 'outside', 'noise')[int(rng.integers(0, 5))]
    imgs = [tensor(rng, (B, 3, hs[s], ws[s]), uniform=True) for s in range(n)]
    flows0 = [flow_of(rng, B, hs[s], ws[s], kind) for s in range(n)]
    gout = [tensor(rng, (B, 3, hs[s], ws[s])) for s in range(n)]
    draw = 'warp n=%d B=%d hs=%s ws=%s ac=%d %s' % (n, B, hs, ws, ac, kind)
    fa = [f.clone().requires_grad_() for f in flows0]
    per = [ops.warp_flow_masked(imgs[s], fa[s], align_corners=ac)[0] for s in range(n)]
    sum((o * g).sum() for o, g in zip(per, gout)).backward()
    fb = [f.clone().requires_grad_() for f in flows0]
    one = ops.warp_flow_masked_pyramid(imgs, fb, align_corners=ac)
    sum((o * g).sum() for o, g in zip(one, gout)).backward()
    for s in range(n):
        if not torch.equal(per[s], one[s]):
            return draw, 'scale %d forward differs' % s
        if not torch.equal(fa[s].grad, fb[s].grad):
            return draw, 'scale %d flow gradient differs by %g' % (s, float((fa[s].grad - fb[s].grad).abs().max()))
    return draw, None


def _corr_against_oracle(rng, d, B, C, h, w, mode, rtol, atol_of):
    f1c, f2c = tensor(rng, (B, C, h, w)).requires_grad_(), tensor(rng, (B, C, h, w)).requires_grad_()
    cv_ref = R.corr_naive(f1c, f2c, d)
    gout = tensor(rng, tuple(cv_ref.shape), 0.05 if mode else 1.0)
    cv_ref.backward(gout)
    f1, f2 = f1c.detach().clone().requires_grad_(), f2c.detach().clone().requires_grad_()
    cv = ops.corr(f1, f2, d, backward=mode or 'auto')
    cv.backward(gout)
    amax = max(float(f1c.grad.abs().max()), float(f2c.grad.abs().max()))
    if not np.allclose(cv.detach().numpy(), cv_ref.detach().numpy(), rtol=1e-5, atol=2e-6):
        return 'forward off by %g' % float((cv - cv_ref).abs().max())
    for name, got, ref in (('gf1', f1.grad, f1c.grad), ('gf2', f2.grad, f2c.grad)):
        if not np.allclose(got.numpy(), ref.numpy(), rtol=rtol, atol=atol_of(amax)):
            return '%s off by %g (largest gradient %g)' % (name, float((got - ref).abs().max()), amax)
    return None


def fuzz_mfma(rng):
    d = 4 if rng.random() < 0.6 else 8
    C = 16 * int(rng.integers(1, 5))
    w = 4 * int(rng.integers(4, 40))
    h = int(rng.integers(4 * d, 4 * d + 40))
    B = max(1, -(-8192 // (h * w)))
    B += int(rng.integers(0, 2))
    draw = 'mfma d=%d [%d,%d,%d,%d]' % (d, B, C, h, w)
    return draw, _corr_against_oracle(rng, d, B, C, h, w, 'mfma', 1e-4, lambda amax: 1e-5 * amax)


def fuzz_smallrows(rng):
    """round 6: the small-map cost-volume backward with the gradient rows through registers (backward='fp32_next', csrc/corr_small_rows.h): maps of
    <= 1024 pixels, any channel count, d = 4 and 8 -- pixel blocks, channel phases, ragged chunks and dead lanes as the draw has them."""
    d = 8 if rng.random() < 0.6 else 4
    h = int(rng.integers(1, 33))
    w = int(rng.integers(1, max(2, min(96, 1024 // h) + 1)))
    C = int(rng.integers(1, 70))
    B = int(rng.integers(1, 4))
    draw = 'smallrows d=%d [%d,%d,%d,%d]' % (d, B, C, h, w)
    return draw, _corr_against_oracle(rng, d, B, C, h, w, 'fp32_next', 1e-4 if d == 8 else 1e-5,
                                      lambda amax: max(5e-6, (1e-5 if d == 8 else 1e-6) * max(amax, 1.0 if d == 8 else 0.0)))


def fuzz_corr(rng):
    d = (1, 2, 4, 4, 4, 8)[int(rng.integers(0, 6))]
    C = int(rng.integers(1, 40))
    big = rng.random() < 0.35
    h, w = (int(rng.integers(20, 80)), int(rng.integers(60, 280))) if big else (int(rng.integers(1, 30)), int(rng.integers(1, 90)))
    if rng.random() < 0.5:
        w = max(4, w // 4 * 4)                                             # the LDS-DMA paths want W % 4 == 0
    B = int(rng.integers(1, 4)) if not big else max(1, int(rng.integers(8192, 140000)) // (h * w))
    B = min(B, 24)
    draw = 'corr d=%d [%d,%d,%d,%d]' % (d, B, C, h, w)
    return draw, _corr_against_oracle(rng, d, B, C, h, w, 'fp32' if d == 8 and rng.random() < 0.5 else None, 1e-4 if d == 8 else 1e-5,
                                      lambda amax: max(5e-6, (1e-5 if d == 8 else 1e-6) * max(amax, 1.0 if d == 8 else 0.0)))


def fuzz_fused(rng):
    C = int(rng.integers(1, 40))
    big = rng.random() < 0.3
    h, w = (int(rng.integers(40, 70)), 4 * int(rng.integers(24, 60))) if big else (int(rng.integers(1, 30)), 4 * int(rng.integers(2, 24)))
    B = max(1, int(rng.integers(100000, 140000)) // (h * w)) if big and rng.random() < 0.5 else int(rng.integers(1, 4))
    ac = bool(rng.integers(0, 2))
    kind = ('smooth', 'mixed', 'edge', 'outside', 'noise')[int(rng.integers(0, 5))]
    draw = 'fused [%d,%d,%d,%d] ac=%d %s' % (B, C, h, w, ac, kind)
    f1c, f2c = tensor(rng, (B, C, h, w)).requires_grad_(), tensor(rng, (B, C, h, w)).requires_grad_()
    fc = flow_of(rng, B, h, w, kind).requires_grad_()
    cvr = R.corr_naive(f1c, R.warp_flow(f2c, fc, False, ac), 4)
    g = tensor(rng, tuple(cvr.shape))
    cvr.backward(g)
    f1, f2, f = (t.detach().clone().requires_grad_() for t in (f1c, f2c, fc))
    if not ops.warp_corr_supported(f1, 4):
        return draw, 'not served'
    cv = ops.warp_corr(f1, f2, f, 4, align_corners=ac)
    cv.backward(g)
    amax = max(float(f1c.grad.abs().max()), float(f2c.grad.abs().max()), 1e-6)
    scale = max(float(fc.grad.abs().max()), 1e-6)
    for name, got, ref, rtol, atol in (('cv', cv.detach(), cvr.detach(), 1e-5, 2e-6), ('gf1', f1.grad, f1c.grad, 1e-4, max(5e-6, 2e-6 * amax)),
                                       ('gf2', f2.grad, f2c.grad, 1e-4, max(2e-5, 2e-6 * amax)), ('gflow', f.grad, fc.grad, 1e-4, 2e-5 * scale)):
        if not np.allclose(got.numpy(), ref.numpy(), rtol=rtol, atol=atol):
            return draw, '%s off by %g (bar %g + %g rel)' % (name, float((got - ref).abs().max()), atol, rtol)
    return draw, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--minutes', type=float, default=2.0)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--what', default='ms,warp,mfma,corr,fused')
    a = ap.parse_args()
    fns = {'ms': fuzz_ms, 'warp': fuzz_warp, 'mfma': fuzz_mfma, 'corr': fuzz_corr, 'fused': fuzz_fused, 'smallrows': fuzz_smallrows}
    which = [fns[k] for k in a.what.split(',')]
    t_end = time.time() + 60 * a.minutes
    runs, fails, k = {f.__name__: 0 for f in which}, 0, 0
    with hostexec.patched(ops):
        while time.time() < t_end and fails < 100:
            fn = which[k % len(which)]
            rng = np.random.default_rng([a.seed, k])
            k += 1
            try:
                draw, bad = fn(rng)
            except Exception as e:                                        # an exception is a finding too (an entry point that rejects a shape its Python predicate accepted)
                draw, bad = '%s draw %d' % (fn.__name__, k - 1), 'raised %s: %s' % (type(e).__name__, str(e)[:200])
            runs[fn.__name__] += 1
            if bad:
                fails += 1
                print('FAIL seed=%d draw=%d %s: %s' % (a.seed, k - 1, draw, bad), flush=True)
    print('draws %s, failures %d' % (runs, fails), flush=True)
    return min(fails, 100)


if __name__ == '__main__':
    sys.exit(main())
