"""GPU parity: every HIP operator (through the C ABI) against the CPU oracle and the golden
fixtures captured from the reference.  Run with ``-m gpu`` on an MI355X.

Bars (BASELINE.json north_star): integer / binary outputs bit-exact; floating point within
1e-4 relative (the tolerance is written at each assert; op-level checks are usually tighter)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle_cache import corr_case

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from unopticalflow_amd import ops as _ops, _lib
    _lib.load()                       # fail loudly if the HIP library is missing
    return _ops


def dev(a):
    t = T(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to('cuda')


def close(a, b, rtol=1e-4, atol=1e-6, what=''):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


# Paths that are OFF by default and have not passed a complete `-m gpu` run on an MI355X (the lease was closed from outside the build for rounds 5-6):
# their tests run, a pass shows as XPASS and a failure as xfailed -- neither stops a `-x` run of the default path's tests, which carry no such mark.
UNVALIDATED = pytest.mark.xfail(strict=False, reason='off-by-default path without a complete GPU suite behind it: XPASS = holds, xfailed = does not')


def rnd(seed, shape, scale=1.0, uniform=False):
    rng = np.random.default_rng(seed)
    a = rng.random(shape, dtype=np.float32) if uniform else rng.standard_normal(shape).astype(np.float32)
    return T(a * np.float32(scale))


# ------------------------------------------------------------------------------------ corr
def test_corr_golden(ops, golden):
    g = golden('g1_corr.npz')
    for k, (d, C, h, w) in enumerate(g['cases']):
        f1 = dev(g['f1_%d' % k]).requires_grad_()
        f2 = dev(g['f2_%d' % k]).requires_grad_()
        cv = ops.corr(f1, f2, int(d))
        close(cv, g['cv_%d' % k], rtol=1e-5, atol=1e-6, what='corr fwd case %d' % k)
        cv.backward(dev(g['g_%d' % k]))
        close(f1.grad, g['gf1_%d' % k], rtol=1e-5, atol=2e-6, what='corr gf1 case %d' % k)
        close(f2.grad, g['gf2_%d' % k], rtol=1e-5, atol=2e-6, what='corr gf2 case %d' % k)


@pytest.mark.parametrize('backward', ['auto', 'fp32', pytest.param('mfma', marks=UNVALIDATED)])
@pytest.mark.parametrize('k', [0, 1, 2])
def test_corr_golden_at_matrix_core_shapes(ops, golden, k, backward):
    """the REFERENCE's gradients (tests/golden/g6_corr_served.npz, gen_golden.py corr_served: pwc_tf.py:97-106 through autograd)
    at shapes the matrix-core backward serves -- d = 8 [1,16,64,128], d = 8 with a ragged last segment [1,16,72,116], d = 4 [1,16,64,128] -- for
    every arithmetic a caller can ask for.  Bars: the fp32 kernels at test_corr_golden's; the matrix-core form at north_star's 1e-4 relative + 1e-5
    of the fixture's largest gradient (a sum of (2d+1)^2 signed products cancels: an element near zero has no relative accuracy in ANY fp32 order)."""
    g = golden('g6_corr_served.npz')
    d, B, C, h, w = (int(v) for v in g['cases'][k])
    s1, s2, s3 = (int(v) for v in g['seeds'][k])
    f1, f2 = dev(rnd(s1, (B, C, h, w))).requires_grad_(), dev(rnd(s2, (B, C, h, w))).requires_grad_()
    cv = ops.corr(f1, f2, d, backward=backward)
    close(cv[:, :, ::8, ::8], g['cv_s_%d' % k], rtol=1e-5, atol=2e-6, what='corr fwd case %d' % k)
    cv.backward(dev(rnd(s3, tuple(cv.shape))))
    amax = max(float(np.abs(g['gf1_%d' % k]).max()), float(np.abs(g['gf2_%d' % k]).max()))
    rtol, atol = (1e-4, 1e-5 * amax) if backward == 'mfma' else (1e-5, 5e-6)
    if backward == 'auto':                       # whatever the library picks by itself holds the tighter bar or is the matrix-core form
        ref1 = dev(rnd(s1, (B, C, h, w))).requires_grad_(); ref2 = dev(rnd(s2, (B, C, h, w))).requires_grad_()
        ops.corr(ref1, ref2, d, backward='fp32').backward(dev(rnd(s3, tuple(cv.shape))))
        if not (torch.equal(ref1.grad, f1.grad) and torch.equal(ref2.grad, f2.grad)):
            rtol, atol = 1e-4, 1e-5 * amax
    close(f1.grad, g['gf1_%d' % k], rtol=rtol, atol=atol, what='corr gf1 case %d (%s)' % (k, backward))
    close(f2.grad, g['gf2_%d' % k], rtol=rtol, atol=atol, what='corr gf2 case %d (%s)' % (k, backward))


@pytest.mark.parametrize('d,C,h,w', [(4, 32, 64, 208), (4, 64, 32, 104), (4, 96, 16, 52), (4, 128, 8, 26),
                                     (4, 196, 4, 13), (8, 32, 32, 104), (3, 6, 10, 70), (2, 9, 17, 130),
                                     (1, 4, 5, 5), (0, 3, 4, 6), (4, 3, 23, 97), (4, 17, 9, 129)])
def test_corr_vs_oracle(ops, d, C, h, w):
    """Pyramid-level shapes of 832x256 (L2..L6), d=8, odd sizes, generic-radius path."""
    f1c, f2c = rnd(1, (2, C, h, w)).requires_grad_(), rnd(2, (2, C, h, w)).requires_grad_()
    cv_ref = R.corr_naive(f1c, f2c, d)
    gout = rnd(3, tuple(cv_ref.shape))
    cv_ref.backward(gout)
    f1, f2 = dev(f1c.detach()).requires_grad_(), dev(f2c.detach()).requires_grad_()
    cv = ops.corr(f1, f2, d)
    close(cv, cv_ref, rtol=1e-5, atol=2e-6)
    cv.backward(dev(gout))
    close(f1.grad, f1c.grad, rtol=1e-5, atol=5e-6)
    close(f2.grad, f2c.grad, rtol=1e-5, atol=5e-6)


@pytest.mark.parametrize('backward', ['auto', pytest.param('mfma', marks=UNVALIDATED)])
@pytest.mark.parametrize('C,h,w', [(32, 64, 208), (64, 32, 104), (96, 16, 52), (128, 8, 26), (196, 4, 13)])
def test_corr_d8_full_pyramid(ops, C, h, w, backward):
    """BASELINE config 5: d=8 cost volume (289 planes) on every pyramid-level shape of 832x256 (the oracle's answer per shape is evaluated once per
    session and shared by the three arithmetics: tests/oracle_cache.py)."""
    o = corr_case(8, 2, C, h, w, seeds=(8, 9, 10))
    f1c, f2c, gout, cv_ref = o['f1'], o['f2'], o['gout'], o['cv']
    f1, f2 = dev(f1c).requires_grad_(), dev(f2c).requires_grad_()
    cv = ops.corr(f1, f2, 8, backward=backward)
    assert cv.shape[1] == 289
    close(cv, cv_ref, rtol=1e-5, atol=2e-6)
    cv.backward(dev(gout))
    # 'auto' (what Model_flow runs): the bar of rounds 1-4.  'mfma': maps of >= 8192 pixels with C % 16 == 0 take the matrix-core backward (bf16
    # hi/lo split products, ~4e-6 of the LARGEST gradient from the fp32 sums -- a sum of 289 signed products cancels, so the absolute part of that
    # arithmetic's bar scales with the largest value; the round-5 GPU run had 1 of 851,968 elements at 1.16e-5 absolute under the fixed 1e-5)
    amax = max(1.0, o['gf1'].abs().max().item(), o['gf2'].abs().max().item())
    atol = 1e-5 * amax if backward == 'mfma' else 1e-5
    close(f1.grad, o['gf1'], rtol=1e-4, atol=atol)
    close(f2.grad, o['gf2'], rtol=1e-4, atol=atol)


@pytest.mark.parametrize('B,C,h,w', [(16, 32, 64, 208), (8, 32, 112, 256), (5, 7, 100, 268), (12, 64, 32, 104),
                                     (2, 1, 256, 256), (2, 3, 260, 256), (3, 2, 40, 72), (1, 5, 12, 700)])
def test_corr_large_map_paths(ops, B, C, h, w):
    """Shapes that take the LDS-DMA ring kernels (level 2 of the 832x256 B=8 step and of 1024x448 B=4;
    ragged ones with partial tiles, odd B; fewer channels than ring stages, C not a multiple of the
    stage size) and the mid-size group kernels, vs the oracle (evaluated once per case and session: tests/oracle_cache.py)."""
    o = corr_case(4, B, C, h, w)
    f1, f2 = dev(o['f1']).requires_grad_(), dev(o['f2']).requires_grad_()
    cv = ops.corr(f1, f2, 4)
    close(cv, o['cv'], rtol=1e-5, atol=2e-6)
    cv.backward(dev(o['gout']))
    # an fp32 sum of 81 products per channel in a different order than ATen's: the error scales with the largest
    # terms, not with the (possibly cancelled) result -- atol relative to the largest gradient (1e-6 of it; with C = 1
    # nothing is divided by C and the terms reach ~30)
    amax = max(o['gf1'].abs().max().item(), o['gf2'].abs().max().item())
    close(f1.grad, o['gf1'], rtol=1e-5, atol=max(5e-6, 1e-6 * amax))
    close(f2.grad, o['gf2'], rtol=1e-5, atol=max(5e-6, 1e-6 * amax))


@pytest.mark.parametrize('backward', ['auto', pytest.param('mfma', marks=UNVALIDATED)])
@pytest.mark.parametrize('B,C,h,w', [(16, 32, 64, 208), (5, 7, 100, 268), (3, 2, 40, 72), (12, 64, 32, 104), (16, 96, 16, 52)])
def test_corr_d8_large_map_paths(ops, B, C, h, w, backward):
    """d=8 (BASELINE config 5) at the batch the step uses (2B=16): the LDS-DMA ring forward with 17 displacement
    rows split over workgroups (4 or 3 rows each, the last group partial), ragged tiles, odd channel counts."""
    o = corr_case(8, B, C, h, w)
    f1, f2 = dev(o['f1']).requires_grad_(), dev(o['f2']).requires_grad_()
    cv = ops.corr(f1, f2, 8, backward=backward)
    close(cv, o['cv'], rtol=1e-5, atol=2e-6)
    cv.backward(dev(o['gout']))
    amax = max(1.0, o['gf1'].abs().max().item(), o['gf2'].abs().max().item())      # (as in test_corr_d8_full_pyramid)
    atol = 1e-5 * amax if backward == 'mfma' else 1e-5
    close(f1.grad, o['gf1'], rtol=1e-4, atol=atol)
    close(f2.grad, o['gf2'], rtol=1e-4, atol=atol)


@pytest.mark.parametrize('B,C,h,w', [(16, 128, 8, 26), (16, 196, 4, 13), (3, 5, 7, 11), (1, 2, 30, 34), (2, 1, 3, 3), (4, 128, 14, 32)])
def test_corr_small_map_backward(ops, B, C, h, w, backward='auto'):
    """Levels 5 / 6 (832x256 and 1024x448) and ragged tiny maps: the whole-map backward kernel (a lane owns a pixel
    and its 81 upstream gradients; channel chunks over workgroups; odd channel counts, several pixel blocks).  (backward='fp32_next' --
    the round-6 kernel with the gradient rows passing through registers, csrc/corr_small_rows.h -- is called with the same body and bars from
    tests/test_zz_round5_gpu.py, in a child process: device code that has never run does not share a process with the validated suite.)"""
    f1c, f2c = rnd(20, (B, C, h, w)).requires_grad_(), rnd(21, (B, C, h, w)).requires_grad_()
    cv_ref = R.corr_naive(f1c, f2c, 4)
    gout = rnd(22, tuple(cv_ref.shape))
    cv_ref.backward(gout)
    f1, f2 = dev(f1c.detach()).requires_grad_(), dev(f2c.detach()).requires_grad_()
    cv = ops.corr(f1, f2, 4, backward=backward)
    close(cv, cv_ref, rtol=1e-5, atol=2e-6)
    cv.backward(dev(gout))
    close(f1.grad, f1c.grad, rtol=1e-5, atol=5e-6)
    close(f2.grad, f2c.grad, rtol=1e-5, atol=5e-6)


@pytest.mark.parametrize('d,B,C,h,w', [(4, 16, 32, 64, 208), (4, 12, 64, 32, 104), (4, 16, 96, 16, 52), (4, 8, 32, 112, 256), (4, 3, 16, 40, 72),
                                       (4, 4, 48, 21, 100), (8, 16, 32, 64, 208), (8, 12, 64, 32, 104)])          # (two more d = 8 shapes: tests/test_zz_round5_gpu.py)
@UNVALIDATED
def test_corr_backward_on_the_matrix_cores(ops, d, B, C, h, w):
    """Round 5 (csrc/corr_mfma.h): the cost-volume backward as banded bf16 hi/lo split products on v_mfma_f32_16x16x32_bf16 -- the
    chosen per call (ops.corr(..., backward='mfma') -> unflow_corr_bwd_ex) -- against the oracle's autograd of corr_naive (pwc_tf.py:97-106) at the pyramid shapes of
    832x256 and 1024x448, ragged last segments (w % 16 != 0), row counts that are not a multiple of the chunk, 16 / 48 channels.
    Bars: north_star's 1e-4 relative (+ 1e-5 of the largest gradient: a sum of (2d+1)^2 signed products cancels); the kernel is
    measured at ~4e-6 of the largest gradient from the fp32 sums.  Deterministic: two launches agree bit for bit; backward='fp32'
    gives the fp32 kernels' bits."""
    o = corr_case(d, B, C, h, w)          # (the case of test_corr_large_map_paths / test_corr_d8_large_map_paths where the shapes coincide: one oracle evaluation)
    f1c, f2c, gout = o['f1'], o['f2'], o['gout']
    amax = max(o['gf1'].abs().max().item(), o['gf2'].abs().max().item())
    runs = []
    for _ in range(2):
        f1, f2 = dev(f1c).requires_grad_(), dev(f2c).requires_grad_()
        ops.corr(f1, f2, d, backward='mfma').backward(dev(gout))
        close(f1.grad, o['gf1'], rtol=1e-4, atol=1e-5 * amax, what='gf1')
        close(f2.grad, o['gf2'], rtol=1e-4, atol=1e-5 * amax, what='gf2')
        runs.append((f1.grad.clone(), f2.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    # the split keeps ~16 bits per factor: clearly away from the fp32 kernels' bits, clearly inside a tenth of the bar
    f1, f2 = dev(f1c).requires_grad_(), dev(f2c).requires_grad_()
    ops.corr(f1, f2, d, backward='fp32').backward(dev(gout))
    err = max((f1.grad - runs[0][0]).abs().max().item(), (f2.grad - runs[0][1]).abs().max().item())
    assert 0 < err < 1e-5 * amax, (err, amax)
    with pytest.raises(ValueError):
        ops.corr(f1, f2, d, backward='fp16')


def test_corr_shape_mismatch_asserts(ops):
    with pytest.raises(AssertionError):                       # pwc_tf.py:99
        ops.corr(torch.zeros(1, 2, 4, 4, device='cuda'), torch.zeros(1, 2, 4, 5, device='cuda'))


def test_corr_full_size_properties(ops):
    """BASELINE config 2 size (B=8 pairs, level-2 maps 32x64x208): size-independent properties."""
    f1, f2, f3 = (rnd(s, (8, 32, 64, 208)).cuda() for s in (11, 12, 13))
    a, b = ops.corr(f1, f2), ops.corr(f1, f3)
    close(ops.corr(f1, f2 + 2.0 * f3), a + 2.0 * b, rtol=1e-4, atol=1e-5, what='linearity in f2')
    centre = ops.corr(f1, f1)[:, 40]                          # zero displacement = mean_c f1^2
    close(centre, (f1 * f1).mean(1), rtol=1e-5, atol=1e-6)
    # swapping the arguments mirrors the displacement: cv21[(-dy,-dx)](p + (dy,dx)) == cv12[(dy,dx)](p)
    cv12, cv21 = ops.corr(f1, f2), ops.corr(f2, f1)
    i, j = 6, 1                                               # dy = +2, dx = -3
    dy, dx = i - 4, j - 4
    lhs = cv12[:, i * 9 + j, 0:64 - dy, 3:208]
    rhs = cv21[:, (8 - i) * 9 + (8 - j), dy:64, 0:208 - 3]
    close(lhs, rhs, rtol=1e-5, atol=1e-6, what='displacement symmetry')


# ------------------------------------------------------------------------------------ warp
@pytest.mark.parametrize('ac', [0, 1])
def test_warp_golden(ops, golden, ac):
    g = golden('g1_warp.npz')
    for k, (C, h, w, s10, um) in enumerate(g['cases']):
        tag = '%d_ac%d' % (k, ac)
        x = dev(g['x_%d' % k]).requires_grad_()
        fl = dev(g['flow_%d' % k]).requires_grad_()
        if um:
            y, m = ops.warp_flow_masked(x, fl, align_corners=bool(ac))
            assert np.array_equal(m.cpu().numpy(), g['mask_' + tag]), 'mask not bit-exact, case %d' % k
        else:
            y = ops.warp_flow(x, fl, use_mask=False, align_corners=bool(ac))
        close(y, g['y_' + tag], rtol=1e-5, atol=1e-6, what='warp fwd %s' % tag)
        y.backward(dev(g['g_%d' % k]))
        close(x.grad, g['gx_' + tag], rtol=1e-4, atol=1e-5, what='warp gsrc %s' % tag)
        close(fl.grad, g['gflow_' + tag], rtol=1e-4, atol=1e-4, what='warp gflow %s' % tag)


@pytest.mark.parametrize('ac', [False, True])
@pytest.mark.parametrize('hw', [(256, 832), (128, 416), (64, 208), (32, 104), (8, 26), (4, 13), (33, 57)])
def test_warp_mask_bit_exact(ops, ac, hw):
    """Binary validity masks (net_utils.py:47-51) must equal the reference CPU path bit for bit,
    on flows that straddle the borders and land near integer positions."""
    h, w = hw
    rng = np.random.default_rng(h * 7 + w + int(ac))
    fl = (rng.standard_normal((2, 2, h, w)) * 6).astype(np.float32)
    fl[0, :, : h // 2] *= 1e-3
    fl[1, :, :, : w // 3] = np.round(fl[1, :, :, : w // 3] * 2) / 2
    x = rng.random((2, 3, h, w), dtype=np.float32)
    m_ref = R.warp_mask(x.shape, T(fl), ac).numpy()
    y_ref = R.warp_flow(T(x), T(fl), True, ac)
    y, m = ops.warp_flow_masked(dev(x), dev(fl), align_corners=ac)
    assert np.array_equal(m.cpu().numpy(), m_ref)
    close(y, y_ref, rtol=1e-5, atol=1e-6)


def test_warp_feature_grads_vs_oracle(ops):
    for (C, h, w, s) in ((32, 64, 208, 2.0), (128, 8, 26, 1.0), (96, 16, 52, 5.0), (7, 9, 11, 3.0)):
        xc = rnd(5, (2, C, h, w)).requires_grad_()
        fc = rnd(6, (2, 2, h, w), s).requires_grad_()
        g = rnd(7, (2, C, h, w))
        yr = R.warp_flow(xc, fc)
        yr.backward(g)
        x, f = dev(xc.detach()).requires_grad_(), dev(fc.detach()).requires_grad_()
        y = ops.warp_flow(x, f)
        close(y, yr, rtol=1e-5, atol=1e-6)
        y.backward(dev(g))
        close(x.grad, xc.grad, rtol=1e-4, atol=1e-5)
        scale = fc.grad.abs().max().item()
        close(f.grad, fc.grad, rtol=1e-4, atol=1e-5 * scale)


def _structured_flow(B, h, w, kind, seed):
    """Flows that exercise every branch of the LDS-tile warp: 'smooth' (translation + low frequencies: the
    source window of a tile fits LDS), 'mixed' (smooth with a noisy band: some tiles fall back to per-tap
    gathers), 'outside' (the whole map samples beyond the border: empty windows), 'edge' (windows clipped
    by the image border), 'noise' (no tile fits)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    u = 2.3 + 1.5 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
    v = -1.1 + 0.8 * np.cos(xx / 29.0 + yy / 41.0)
    fl = np.stack((u, v), 0)[None].repeat(B, 0).astype(np.float32)
    fl += rng.standard_normal((B, 2, 1, 1)).astype(np.float32)            # a different translation per sample
    if kind == 'mixed':
        fl[:, :, h // 3: h // 3 + max(2, h // 6)] += (rng.standard_normal((B, 2, max(2, h // 6), w)) * 9).astype(np.float32)
    elif kind == 'outside':
        fl[:, 0] += 3.0 * w
    elif kind == 'edge':
        fl[:, 0] += w / 2.0 - 3.0
        fl[:, 1] -= h / 2.0
    elif kind == 'noise':
        fl = (rng.standard_normal((B, 2, h, w)) * 7).astype(np.float32)
    return T(fl)


@pytest.mark.parametrize('kind', ['smooth', 'mixed', 'outside', 'edge', 'noise'])
@pytest.mark.parametrize('C,h,w', [(32, 64, 208), (64, 32, 104), (96, 16, 52), (128, 8, 26), (20, 40, 72), (9, 17, 130), (8, 70, 300)])
def test_warp_feature_tiles_vs_oracle(ops, kind, C, h, w):
    """Feature-map warp (LDS-tile kernels) forward and backward against the oracle's grid_sample chain
    (net_utils.py:16-46) for both grid_sample generations."""
    B = 3
    for ac in (False, True):
        xc = rnd(11 + C, (B, C, h, w)).requires_grad_()
        fc = _structured_flow(B, h, w, kind, seed=h * w + C).requires_grad_()
        g = rnd(13 + C, (B, C, h, w))
        yr = R.warp_flow(xc, fc, False, ac)
        yr.backward(g)
        x, f = dev(xc.detach()).requires_grad_(), dev(fc.detach()).requires_grad_()
        y = ops.warp_flow(x, f, align_corners=ac)
        close(y, yr, rtol=1e-5, atol=1e-6, what='fwd %s' % kind)
        y.backward(dev(g))
        close(x.grad, xc.grad, rtol=1e-4, atol=2e-5, what='gsrc %s' % kind)
        scale = max(fc.grad.abs().max().item(), 1e-6)
        close(f.grad, fc.grad, rtol=1e-4, atol=2e-5 * scale, what='gflow %s' % kind)
        # source without gradient (gsrc == NULL at the C ABI)
        f2 = dev(fc.detach()).requires_grad_()
        ops.warp_flow(x.detach(), f2, align_corners=ac).backward(dev(g))
        close(f2.grad, fc.grad, rtol=1e-4, atol=2e-5 * scale, what='gflow only %s' % kind)
        # the gather form of the source gradient (unflow_warp_bwd_det: a tile of gsrc per workgroup, per-tile displacement
        # table, no atomics): same bars, and bit-identical from run to run
        runs = []
        for _ in range(2):
            x3, f3 = dev(xc.detach()).requires_grad_(), dev(fc.detach()).requires_grad_()
            ops.warp_flow(x3, f3, align_corners=ac, deterministic=True).backward(dev(g))
            close(x3.grad, xc.grad, rtol=1e-4, atol=2e-5, what='gather gsrc %s' % kind)
            close(f3.grad, fc.grad, rtol=1e-4, atol=2e-5 * scale, what='gather gflow %s' % kind)
            runs.append(x3.grad.clone())
        if C >= 8 and w >= 8 and h * w >= 512:            # (the shapes the tile kernels serve; smaller maps keep the per-tap atomics)
            assert torch.equal(runs[0], runs[1])
            # round 4: the whole backward as ONE gather pass (unflow_warp_bwd_fused: source gradient by gather + flow gradient by the
            # tile's owner); forced here -- at B = 3 the launch is below the size the op picks it for -- same bars, both gradients
            # bit-identical from run to run
            runs = []
            for _ in range(2):
                x4, f4 = dev(xc.detach()).requires_grad_(), dev(fc.detach()).requires_grad_()
                ops.warp_flow(x4, f4, align_corners=ac, fused_backward=True).backward(dev(g))
                close(x4.grad, xc.grad, rtol=1e-4, atol=2e-5, what='fused gsrc %s' % kind)
                close(f4.grad, fc.grad, rtol=1e-4, atol=2e-5 * scale, what='fused gflow %s' % kind)
                runs.append((x4.grad.clone(), f4.grad.clone()))
            assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_warp_backward_fused_at_the_step_shapes(ops):
    """The shapes the op picks the one-pass backward for by itself (level 2 of the 832x256 step at the step's batch of 16 directed
    pairs, level 2 of 1024x448 at 8): chosen without being asked, equal to the zero-fill + scatter form within the float-atomic noise of the latter,
    and bit-identical from run to run."""
    lib = __import__('unopticalflow_amd._lib', fromlist=['load']).load()
    assert lib.unflow_warp_bwd_fused_supported(16, 32, 64, 208) == 2 and lib.unflow_warp_bwd_fused_supported(16, 64, 32, 104) == 1
    assert lib.unflow_warp_bwd_fused_supported(16, 96, 16, 52) == 1 and lib.unflow_warp_bwd_fused_supported(16, 3, 256, 832) == 0
    assert lib.unflow_warp_bwd_fused_supported(8, 32, 112, 256) == 2                  # (448x1024, bs 4: level 2)
    for C, h, w in ((32, 64, 208), (32, 112, 256)):
        Bq = 16 if h == 64 else 8
        x0, g = rnd(500 + C, (Bq, C, h, w)), rnd(501 + C, (Bq, C, h, w))
        f0 = _structured_flow(Bq, h, w, 'mixed', seed=C)
        res = {}
        for fused in (None, False, None):
            x, f = dev(x0).requires_grad_(), dev(f0).requires_grad_()
            ops.kernel_timer.enable(True)
            ops.warp_flow(x, f, fused_backward=fused).backward(dev(g))
            torch.cuda.synchronize()
            ops.kernel_timer.disable()
            names = {r['entry'] for r in ops.kernel_timer.rows()}
            assert ('unflow_warp_bwd_fused' in names) == (fused is None), names
            # the forward leaves the displacement table behind exactly when the one-pass backward will want it (so that pass is ONE launch)
            assert ('unflow_warp_fwd_table' in names) == (fused is None) and ('unflow_warp_fwd' in names) == (fused is not None), names
            res.setdefault(fused, []).append((x.grad.clone(), f.grad.clone()))
        (a, fa), (b, fb) = res[None]
        assert torch.equal(a, b) and torch.equal(fa, fb)
        close(a, res[False][0][0], rtol=1e-4, atol=2e-5)
        close(fa, res[False][0][1], rtol=1e-4, atol=2e-5 * max(fa.abs().max().item(), 1e-6))


# all five flow kinds at level 2 and at a ragged shape; the two kinds with the most out-of-map taps at the others (the fused operator is an option
# of the model, not its default: 20 cases x 2 conventions of the oracle's 81-shift chain instead of 35 x 2)
_FUSED_CASES = [(C, h, w, kind) for (C, h, w) in ((32, 64, 208), (5, 23, 72)) for kind in ('smooth', 'mixed', 'edge', 'outside', 'noise')] + \
               [(C, h, w, kind) for (C, h, w) in ((64, 32, 104), (96, 16, 52), (128, 8, 26), (196, 4, 12), (3, 70, 260)) for kind in ('mixed', 'outside')]


@pytest.mark.parametrize('C,h,w,kind', _FUSED_CASES)
def test_warp_corr_fused_vs_oracle(ops, C, h, w, kind):
    """Fused warp + cost volume (pwc_tf.py:121-122) at the pyramid-level shapes and ragged ones, forward and backward,
    against the oracle's  corr_naive(f1, warp_flow(f2, flow))  for both grid_sample generations."""
    B = 3
    for ac in (False, True):
        f1c, f2c = rnd(31 + C, (B, C, h, w)).requires_grad_(), rnd(32 + C, (B, C, h, w)).requires_grad_()
        fc = _structured_flow(B, h, w, kind, seed=h + w + C).requires_grad_()
        cvr = R.corr_naive(f1c, R.warp_flow(f2c, fc, False, ac), 4)
        g = rnd(33 + C, tuple(cvr.shape))
        cvr.backward(g)
        f1, f2, f = dev(f1c.detach()).requires_grad_(), dev(f2c.detach()).requires_grad_(), dev(fc.detach()).requires_grad_()
        assert ops.warp_corr_supported(f1, 4) == (w % 4 == 0)         # (W = 26: served by the two separate operators)
        cv = ops.warp_corr(f1, f2, f, 4, align_corners=ac)
        close(cv, cvr, rtol=1e-5, atol=2e-6, what='fused cv %s' % kind)
        cv.backward(dev(g))
        amax = max(f1c.grad.abs().max().item(), f2c.grad.abs().max().item(), 1e-6)
        close(f1.grad, f1c.grad, rtol=1e-4, atol=max(5e-6, 2e-6 * amax), what='fused gf1 %s' % kind)
        close(f2.grad, f2c.grad, rtol=1e-4, atol=max(2e-5, 2e-6 * amax), what='fused gf2 %s' % kind)
        scale = max(fc.grad.abs().max().item(), 1e-6)
        close(f.grad, fc.grad, rtol=1e-4, atol=2e-5 * scale, what='fused gflow %s' % kind)


def test_warp_corr_unsupported_shapes_fall_back(ops):
    """W % 4 != 0 and d != 4 are served by the separate operators behind the same call."""
    for (C, h, w, d) in ((6, 9, 13, 4), (4, 12, 40, 2)):
        f1c, f2c, fc = rnd(1, (2, C, h, w)), rnd(2, (2, C, h, w)), rnd(3, (2, 2, h, w), 1.5)
        cvr = R.corr_naive(f1c, R.warp_flow(f2c, fc), d)
        cv = ops.warp_corr(dev(f1c), dev(f2c), dev(fc), d)
        close(cv, cvr, rtol=1e-5, atol=2e-6)
    with pytest.raises(ValueError):
        ops.warp_corr(dev(rnd(1, (1, 4, 8, 8))), dev(rnd(2, (1, 4, 8, 8))), dev(rnd(3, (1, 2, 8, 9))))


def test_warp_shape_mismatch_raises(ops):
    with pytest.raises(ValueError):                           # net_utils.py:35-36
        ops.warp_flow(torch.zeros(1, 3, 8, 8, device='cuda'), torch.zeros(1, 2, 8, 9, device='cuda'))


def test_warp_identity_and_shift_full_size(ops):
    """Full-size properties: zero flow is the identity under align_corners=True (up to the fp32
    rounding of the reference's own normalise/unnormalise round trip, ~W * 2^-24 px); an integer
    shift reproduces the shifted image with a mask that only drops the uncovered columns."""
    x = rnd(21, (8, 3, 256, 832), uniform=True).cuda()
    z = torch.zeros(8, 2, 256, 832, device='cuda')
    y, m = ops.warp_flow_masked(x, z, align_corners=True)
    close(y, x, rtol=0, atol=2e-4)
    assert bool(m.all())
    z[:, 0] = 3.0
    y, m = ops.warp_flow_masked(x, z, align_corners=True)
    close(y[..., :829], x[..., 3:], rtol=0, atol=2e-4)
    assert int(m[..., 829:].sum()) == 0 and bool(m[..., :828].all())


# ------------------------------------------------------------------------------------ losses
def _loss_inputs(g):
    return (dev(g['img']), dev(g['from_l']).requires_grad_(), dev(g['from_r']).requires_grad_(), dev(g['gl']))


def test_occ_weight_and_losses_golden(ops, golden):
    g = golden('g1_losses.npz')
    img, fl, fr, gl = _loss_inputs(g)
    d_l, d_r, w_b, w_f, v_b, v_f = ops.occ_weight(img, fl, fr)
    close(d_l, g['diff_l'], rtol=1e-6, atol=1e-7); close(d_r, g['diff_r'], rtol=1e-6, atol=1e-7)
    close(w_b, g['w_bwd'], rtol=1e-5, atol=1e-6); close(w_f, g['w_fwd'], rtol=1e-5, atol=1e-6)
    assert np.array_equal(v_b.cpu().numpy() != 0, g['w_bwd'] != 0)
    assert np.array_equal(v_f.cpu().numpy() != 0, g['w_fwd'] != 0)
    lp = ops.masked_mean(d_r, w_f) + ops.masked_mean(d_l, w_b)
    close(lp, g['loss_pixel'], rtol=1e-5)
    ls_f, ls_b = ops.ssim_loss(img, fr, w_f), ops.ssim_loss(img, fl, w_b)
    close(ls_f, g['loss_ssim_f'], rtol=1e-5); close(ls_b, g['loss_ssim_b'], rtol=1e-5)
    (lp * gl).sum().backward(retain_graph=True)
    close(fl.grad, g['lp_g_from_l'], rtol=1e-4, atol=1e-8); close(fr.grad, g['lp_g_from_r'], rtol=1e-4, atol=1e-8)
    fl.grad = None; fr.grad = None
    ((ls_f + ls_b) * gl).sum().backward()
    s = np.abs(g['ls_g_from_l']).max()
    # (measured, tools/probes/ssim_grad_probe.py: worst element 1.0e-6 of the largest gradient, 5e-5 relative where it matters; was 1e-3 / 1e-4)
    close(fl.grad, g['ls_g_from_l'], rtol=1e-4, atol=5e-6 * s); close(fr.grad, g['ls_g_from_r'], rtol=1e-4, atol=5e-6 * s)
    w3 = w_f.repeat(1, 3, 1, 1)
    close(ops.ssim_map(img * w3, fr.detach() * w3), g['ssim_map'], rtol=1e-4, atol=1e-5)

    ff = dev(g['flow_f']).requires_grad_()
    lsm = ops.smooth2_loss(ff, img)
    close(lsm, g['loss_smooth'], rtol=1e-5)
    (lsm * gl).sum().backward()
    close(ff.grad, g['lsm_g_flow'], rtol=1e-4, atol=1e-9)

    ff = dev(g['flow_f']).requires_grad_()
    fb = dev(g['flow_b']).requires_grad_()
    lc = ops.consis_loss(ff, fb, w_f)
    close(lc, g['loss_consis'], rtol=1e-5)
    (lc * gl).sum().backward()
    close(ff.grad, g['lc_g_flow'], rtol=1e-4, atol=1e-8)
    assert fb.grad is None


@pytest.mark.parametrize('hw', [(64, 208), (32, 104), (13, 61), (128, 416)])
def test_losses_vs_oracle_random(ops, hw):
    h, w = hw
    B = 2
    img = rnd(31, (B, 3, h, w), uniform=True)
    from_l = (img + rnd(32, (B, 3, h, w), 0.1)).clamp(0, 1)
    from_r = (img + rnd(33, (B, 3, h, w), 0.1)).clamp(0, 1)
    from_l[:, :, 2:7, 3:19] = 0.0
    from_r[:, :, h // 2:, w // 2:] = 0.0
    flf, flb = rnd(34, (B, 2, h, w), 3.0), rnd(35, (B, 2, h, w), 3.0)
    gl = rnd(36, (B,))
    # oracle
    l_c, r_c = from_l.clone().requires_grad_(), from_r.clone().requires_grad_()
    ff_c = flf.clone().requires_grad_()
    d_l, d_r, w_b, w_f, v_b, v_f = R.diff_weight(img, l_c, r_c)
    tot_c = (R.masked_l1(d_r, w_f) + R.masked_l1(d_l, w_b)) * 0.15 + \
        (R.ssim_loss(img, r_c, w_f) + R.ssim_loss(img, l_c, w_b)) * 0.85 + \
        R.grad2_error(ff_c / 20.0, img) * 10.0 + R.consis_loss(ff_c, flb, w_f) * 0.01
    (tot_c * gl).sum().backward()
    # HIP
    img_g, gl_g = img.cuda(), gl.cuda()
    l_g, r_g = from_l.cuda().requires_grad_(), from_r.cuda().requires_grad_()
    ff_g = flf.cuda().requires_grad_()
    D_l, D_r, W_b, W_f, V_b, V_f = ops.occ_weight(img_g, l_g, r_g)
    assert np.array_equal(V_b.cpu().numpy(), v_b.numpy().astype(np.uint8))
    assert np.array_equal(V_f.cpu().numpy(), v_f.numpy().astype(np.uint8))
    close(W_b, w_b, rtol=1e-5, atol=1e-6); close(W_f, w_f, rtol=1e-5, atol=1e-6)
    tot_g = (ops.masked_mean(D_r, W_f) + ops.masked_mean(D_l, W_b)) * 0.15 + \
        (ops.ssim_loss(img_g, r_g, W_f) + ops.ssim_loss(img_g, l_g, W_b)) * 0.85 + \
        ops.smooth2_loss(ff_g, img_g) * 10.0 + ops.consis_loss(ff_g, flb.cuda(), W_f) * 0.01
    close(tot_g, tot_c, rtol=1e-5)
    (tot_g * gl_g).sum().backward()
    for a, b in ((l_g.grad, l_c.grad), (r_g.grad, r_c.grad), (ff_g.grad, ff_c.grad)):
        close(a, b, rtol=1e-4, atol=1e-5 * b.abs().max().item())         # (SSIM part measured at 1.3e-6 of the largest gradient; was 1e-3 / 1e-4)


def test_ssim_map_vs_oracle(ops):
    for (C, h, w) in ((3, 64, 208), (1, 5, 7), (4, 33, 130), (3, 256, 832)):
        x, y = rnd(41, (2, C, h, w), uniform=True), rnd(42, (2, C, h, w), uniform=True)
        y[:, :, : h // 2] = x[:, :, : h // 2]                    # SSIM == 1 region
        close(ops.ssim_map(x.cuda(), y.cuda()), R.SSIM(x, y), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('shape', [(2, 3, 24, 40), (1, 1, 5, 7), (2, 3, 64, 208)])
def test_ssim_map_is_differentiable_like_the_reference(ops, shape):
    """SSIM(x, y) (pytorch_ssim/ssim.py:4-20) back-propagates into BOTH arguments in the reference; so does the HIP op."""
    xc = rnd(41, shape, uniform=True).requires_grad_()
    yc = (rnd(42, shape, uniform=True) * 0.7 + 0.1).requires_grad_()
    g = rnd(43, shape)
    sr = R.SSIM(xc, yc)
    sr.backward(g)
    x, y = dev(xc.detach()).requires_grad_(), dev(yc.detach()).requires_grad_()
    sm = ops.ssim_map(x, y)
    close(sm, sr, rtol=1e-4, atol=1e-5)
    sm.backward(dev(g))
    scale = max(xc.grad.abs().max().item(), yc.grad.abs().max().item())
    close(x.grad, xc.grad, rtol=1e-3, atol=1e-4 * scale)
    close(y.grad, yc.grad, rtol=1e-3, atol=1e-4 * scale)
    # only one side needs a gradient
    y2 = dev(yc.detach()).requires_grad_()
    ops.ssim_map(x.detach(), y2).backward(dev(g))
    close(y2.grad, yc.grad, rtol=1e-3, atol=1e-4 * scale)


def test_stacked_directions_equal_separate_launches(ops):
    """The model runs both warp directions of a scale as one 2B launch per loss (centre image broadcast through
    ``img_batch``): values and gradients must equal the per-direction calls bit for bit (same kernels, same order)."""
    B, h, w = 3, 40, 72
    img = dev(rnd(51, (B, 3, h, w), uniform=True))
    wl = (img + dev(rnd(52, (B, 3, h, w), 0.1))).clamp(0, 1)
    wr = (img + dev(rnd(53, (B, 3, h, w), 0.1))).clamp(0, 1)
    wl[:, :, 3:9, 5:17] = 0.0
    flows = dev(rnd(54, (2 * B, 2, h, w), 3.0))
    gl = dev(rnd(55, (2 * B,)))
    # separate
    a, b = wl.clone().requires_grad_(), wr.clone().requires_grad_()
    fa = flows.clone().requires_grad_()
    d_l, d_r, w_b, w_f, _, _ = ops.occ_weight(img, a, b)
    sep = [torch.cat((ops.masked_mean(d_l, w_b), ops.masked_mean(d_r, w_f))),
           torch.cat((ops.ssim_loss(img, a, w_b), ops.ssim_loss(img, b, w_f))),
           torch.cat((ops.smooth2_loss(fa[:B], img), ops.smooth2_loss(fa[B:], img)))]
    (sum(sep) * gl).sum().backward()
    # stacked
    st = torch.cat((wl, wr)).requires_grad_()
    fb = flows.clone().requires_grad_()
    diff, wgt = ops.occ_weight_stacked(img, st)
    assert torch.equal(diff, torch.cat((d_l, d_r))) and torch.equal(wgt, torch.cat((w_b, w_f)))
    stk = [ops.masked_mean(diff, wgt), ops.ssim_loss(img, st, wgt), ops.smooth2_loss(fb, img)]
    (sum(stk) * gl).sum().backward()
    for x, y in zip(sep, stk):
        assert torch.equal(x, y)
    assert torch.equal(st.grad, torch.cat((a.grad, b.grad)))
    assert torch.equal(fb.grad, fa.grad)


def test_reductions_are_reproducible(ops):
    """Per-sample reductions use fixed-order partial sums: two launches agree bitwise."""
    img = rnd(51, (8, 3, 256, 832), uniform=True).cuda()
    wp = rnd(52, (8, 3, 256, 832), uniform=True).cuda()
    w = rnd(53, (8, 1, 256, 832), uniform=True).cuda()
    a, b = ops.ssim_loss(img, wp, w), ops.ssim_loss(img, wp, w)
    assert torch.equal(a, b)
    fl = rnd(54, (8, 2, 256, 832), 5.0).cuda()
    assert torch.equal(ops.smooth2_loss(fl, img), ops.smooth2_loss(fl, img))


# ------------------------------------------------------------------------------------ conv epilogue / pyramid
@pytest.mark.parametrize('shape', [(16, 128, 64, 208), (24, 16, 128, 416), (2, 196, 4, 13), (3, 5, 7, 9), (1, 2, 1, 1)])
def test_bias_leaky_relu_vs_torch(ops, shape):
    """conv() epilogue (net_utils.py:7-11): in-place bias + LeakyReLU(0.1) and its backward incl. bias grad."""
    N, C, H, W = shape
    y0 = rnd(61, shape)
    bias = rnd(62, (C,), 0.3)
    g = rnd(63, shape)
    yc = y0.clone().requires_grad_()
    bc = bias.clone().requires_grad_()
    ref = torch.nn.functional.leaky_relu(yc + bc.view(1, C, 1, 1), 0.1)
    ref.backward(g)
    yg = y0.cuda().requires_grad_()
    bg = bias.cuda().requires_grad_()
    out = ops.bias_leaky_relu_(yg * 1.0, bg, 0.1)          # * 1.0: the op works in place on a non-leaf
    assert torch.equal(out.cpu(), ref.detach())
    out.backward(g.cuda())
    assert torch.equal(yg.grad.cpu(), yc.grad)
    close(bg.grad, bc.grad, rtol=1e-4, atol=1e-4 * bc.grad.abs().max().item())


@pytest.mark.parametrize('shape', [(4, 32, 16, 52), (2, 5, 7, 9), (1, 3, 5, 6), (3, 8, 4, 4)])
def test_bias_leaky_two_consumers(ops, shape):
    """Decoder wiring (pwc_tf.py:113-118): the activation feeds a conv-like consumer and a torch.cat; the two
    gradients (one dense, one a channel slice of the cat's gradient) are added inside the backward kernel."""
    N, C, H, W = shape
    y0, bias, other = rnd(67, shape), rnd(68, (C,), 0.3), rnd(69, (N, 3, H, W))
    wa, wb = rnd(70, shape), rnd(71, (N, C + 3, H, W))

    def graph(act_a, act_b, oth):
        return (act_a * wa.to(act_a.device)).sum() + (torch.cat((oth, act_b), 1) * wb.to(act_a.device)).sum()
    yc, bc = y0.clone().requires_grad_(), bias.clone().requires_grad_()
    ref = torch.nn.functional.leaky_relu(yc + bc.view(1, C, 1, 1), 0.1)
    graph(ref, ref, other).backward()
    yg, bg = y0.cuda().requires_grad_(), bias.cuda().requires_grad_()
    a, b = ops.bias_leaky_relu_(yg * 1.0, bg, 0.1, consumers=2)
    assert a.data_ptr() == b.data_ptr() and torch.equal(a.cpu(), ref.detach())
    graph(a, b, other.cuda()).backward()
    close(yg.grad, yc.grad, rtol=1e-6, atol=1e-6)
    close(bg.grad, bc.grad, rtol=1e-4, atol=1e-4 * bc.grad.abs().max().item())
    # one handle unused (pyramid level 1 in training), and both gradients slices of cats
    yg2, bg2 = y0.cuda().requires_grad_(), bias.cuda().requires_grad_()
    a, b = ops.bias_leaky_relu_(yg2 * 1.0, bg2, 0.1, consumers=2)
    (torch.cat((other.cuda(), b), 1) * wb.cuda()).sum().backward()
    yc2 = y0.clone().requires_grad_()
    (torch.cat((other, torch.nn.functional.leaky_relu(yc2 + bias.view(1, C, 1, 1), 0.1)), 1) * wb).sum().backward()
    close(yg2.grad, yc2.grad, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('shape', [(16, 128, 64, 208), (4, 32, 16, 52), (2, 96, 7, 9), (3, 196, 4, 13), (1, 16, 5, 6), (2, 8, 1, 1)])
def test_bias_leaky_channels_last(ops, shape):
    """The same epilogue on channels_last (NHWC) activations -- what the fp32 conv stacks produce: forward bit-equal,
    backward with one dense gradient and one channel slice of a channels_last cat gradient added in the kernel."""
    N, C, H, W = shape
    CL = torch.channels_last
    y0, bias, other = rnd(81, shape), rnd(82, (C,), 0.3), rnd(83, (N, 4, H, W))
    wa, wb = rnd(84, shape), rnd(85, (N, C + 4, H, W))

    def graph(act_a, act_b, oth):
        return (act_a * wa.to(act_a.device)).sum() + (torch.cat((oth, act_b), 1) * wb.to(act_a.device)).sum()
    yc, bc = y0.clone().requires_grad_(), bias.clone().requires_grad_()
    ref = torch.nn.functional.leaky_relu(yc + bc.view(1, C, 1, 1), 0.1)
    graph(ref, ref, other).backward()
    yg, bg = y0.cuda().contiguous(memory_format=CL).requires_grad_(), bias.cuda().requires_grad_()
    a, b = ops.bias_leaky_relu_(yg * 1.0, bg, 0.1, consumers=2)
    assert a.data_ptr() == b.data_ptr() and a.stride() == yg.stride() and torch.equal(a.cpu(), ref.detach())
    graph(a, b, other.cuda().contiguous(memory_format=CL)).backward()
    close(yg.grad, yc.grad, rtol=1e-6, atol=1e-6)
    close(bg.grad, bc.grad, rtol=1e-4, atol=1e-4 * bc.grad.abs().max().item())
    # a gradient that arrives in NCHW order (a consumer outside the channels_last island) is re-laid out first
    yg2, bg2 = y0.cuda().contiguous(memory_format=CL).requires_grad_(), bias.cuda().requires_grad_()
    out = ops.bias_leaky_relu_(yg2 * 1.0, bg2, 0.1)
    out.backward(wa.cuda())
    yc2 = y0.clone().requires_grad_()
    torch.nn.functional.leaky_relu(yc2 + bias.view(1, C, 1, 1), 0.1).backward(wa)
    assert torch.equal(yg2.grad.cpu(), yc2.grad)


@pytest.mark.parametrize('shape,chans', [((16, 64, 208), (81, 32, 2)), ((16, 4, 13), (81,)), ((3, 7, 9), (5, 3)), ((2, 8, 26), (81, 128, 2)),
                                         ((1, 1, 70), (1, 1, 1))])
def test_cat_channels_last_and_back(ops, shape, chans):
    """Layout glue of the channels_last conv stacks: cat -> NHWC in one pass (the decoder input, pwc_tf.py:113), its
    backward (NHWC gradient -> one NCHW gradient per input, unused ones skipped) and the NHWC -> NCHW hand-off of the
    pyramid features.  Pure data movement: bit-equal."""
    B, H, W = shape
    xs = [rnd(90 + k, (B, c, H, W)) for k, c in enumerate(chans)]
    gs = [x.cuda().requires_grad_(k != 1) for k, x in enumerate(xs)]                  # the second input needs no gradient
    out = ops.cat_channels_last(gs)
    ref = torch.cat(xs, 1)
    assert out.shape == ref.shape and (sum(chans) == 1 or H * W == 1 or out.is_contiguous(memory_format=torch.channels_last))
    assert torch.equal(out.cpu(), ref)
    go = rnd(95, tuple(ref.shape))
    out.backward(go.cuda().contiguous(memory_format=torch.channels_last))
    o = 0
    for k, c in enumerate(chans):
        if k == 1:
            assert gs[k].grad is None
        else:
            assert gs[k].grad.is_contiguous() and torch.equal(gs[k].grad.cpu(), go[:, o:o + c])
        o += c
    y = ref.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    z = ops.to_nchw(y * 1.0)
    assert z.is_contiguous() and torch.equal(z.cpu(), ref)
    z.backward(go.cuda())
    assert torch.equal(y.grad.cpu(), go)
    for d in (1, B):                                  # the hand-off that writes its last d samples twice (no torch.cat((c, c)))
        y = ref.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        z = ops.to_nchw(y * 1.0, dup_tail=d)
        assert z.is_contiguous() and torch.equal(z.cpu(), torch.cat((ref, ref[B - d:]), 0))
        gd = rnd(96 + d, (B + d,) + tuple(ref.shape[1:]))
        z.backward(gd.cuda())
        want = gd[:B].clone()
        want[B - d:] += gd[B:]
        assert torch.equal(y.grad.cpu(), want)


@pytest.mark.parametrize('shape,chans', [((16, 64, 208), (81, 32, 2)), ((16, 4, 13), (81,)), ((3, 7, 9), (5, 3)), ((2, 8, 26), (81, 128, 2))])
def test_cat_channels_last_bf16_side(ops, shape, chans):
    """bf16 conv-stack option: the same glue with the NHWC side in bf16 and the NCHW planes fp32 -- the cat rounds once
    (round-to-nearest-even: the values torch.cat(...).bfloat16() holds, what autocast hands the convolution), the way back
    widens exactly.  Bit-equal."""
    B, H, W = shape
    CL = torch.channels_last
    xs = [rnd(190 + k, (B, c, H, W)) for k, c in enumerate(chans)]
    gs = [x.cuda().requires_grad_(k != 1) for k, x in enumerate(xs)]
    out = ops.cat_channels_last(gs, torch.bfloat16)
    ref = torch.cat(xs, 1).to(torch.bfloat16)
    assert out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=CL) and torch.equal(out.cpu(), ref)
    go = rnd(195, tuple(ref.shape)).to(torch.bfloat16)
    out.backward(go.cuda().contiguous(memory_format=CL))
    o = 0
    for k, c in enumerate(chans):
        if k == 1:
            assert gs[k].grad is None
        else:
            assert gs[k].grad.dtype == torch.float32 and gs[k].grad.is_contiguous() and torch.equal(gs[k].grad.cpu(), go[:, o:o + c].float())
        o += c
    y = ref.cuda().contiguous(memory_format=CL).requires_grad_()                       # a bf16 NHWC activation (pyramid feature)
    z = ops.to_nchw(y * 1.0)
    assert z.dtype == torch.float32 and z.is_contiguous() and torch.equal(z.cpu(), ref.float())
    g32 = rnd(196, tuple(ref.shape))
    z.backward(g32.cuda())
    assert y.grad.dtype == torch.bfloat16 and torch.equal(y.grad.cpu(), g32.to(torch.bfloat16))
    y = ref.cuda().contiguous(memory_format=CL).requires_grad_()
    z = ops.to_nchw(y * 1.0, dup_tail=1)
    assert torch.equal(z.cpu(), torch.cat((ref, ref[B - 1:]), 0).float())
    gd = rnd(197, (B + 1,) + tuple(ref.shape[1:]))
    z.backward(gd.cuda())
    want = gd[:B].clone()
    want[B - 1:] += gd[B:]
    assert torch.equal(y.grad.cpu(), want.to(torch.bfloat16))              # one rounding, after the fp32 sum


@pytest.mark.parametrize('shape', [(16, 128, 64, 208), (2, 5, 7, 9), (3, 8, 4, 4), (2, 16, 8, 26)])
def test_bias_leaky_bf16(ops, shape):
    """bf16 conv-stack option: same epilogue on bf16 activations -- fp32 arithmetic, one rounding per element."""
    N, C, H, W = shape
    y0 = rnd(72, shape).to(torch.bfloat16)
    bias = rnd(73, (C,), 0.3)
    ga, gcat = rnd(74, shape).to(torch.bfloat16), rnd(75, (N, C + 3, H, W)).to(torch.bfloat16)
    act = torch.nn.functional.leaky_relu(y0.float() + bias.view(1, C, 1, 1), 0.1)
    ref_out = act.to(torch.bfloat16)
    gsum = ga.float() + gcat[:, 3:].float()
    ref_gin = (gsum * torch.where(ref_out.float() > 0, 1.0, 0.1)).to(torch.bfloat16)
    yg, bg = y0.cuda().requires_grad_(), bias.cuda().requires_grad_()
    a, b = ops.bias_leaky_relu_(yg * 1.0, bg, 0.1, consumers=2)
    assert a.dtype == torch.bfloat16 and torch.equal(a.cpu(), ref_out)
    torch.autograd.backward([a, b], [ga.cuda(), gcat.cuda()[:, 3:]])       # second gradient: a channel slice (cat operand)
    assert yg.grad.dtype == torch.bfloat16 and torch.equal(yg.grad.cpu(), ref_gin)
    assert bg.grad.dtype == torch.float32
    want = ref_gin.float().sum((0, 2, 3))
    close(bg.grad, want, rtol=1e-4, atol=1e-4 * want.abs().max().item())
    one = ops.bias_leaky_relu_(y0.cuda().clone(), bias.cuda(), 0.1)            # single-consumer entry, no grad
    assert torch.equal(one.cpu(), ref_out)


@pytest.mark.parametrize('shape', [(16, 128, 64, 208), (2, 196, 4, 13), (3, 8, 4, 4), (2, 96, 8, 26)])
def test_bias_leaky_bf16_channels_last(ops, shape):
    """The bf16 epilogue on channels_last activations (the bf16 conv stacks with cfg.channels_last): same arithmetic as the NCHW
    bf16 kernels -- fp32 math, one rounding per element, bias gradient over the rounded values."""
    N, C, H, W = shape
    CL = torch.channels_last
    y0 = rnd(72, shape).to(torch.bfloat16)
    bias = rnd(73, (C,), 0.3)
    ga, gcat = rnd(74, shape).to(torch.bfloat16), rnd(75, (N, C + 8, H, W)).to(torch.bfloat16)
    act = torch.nn.functional.leaky_relu(y0.float() + bias.view(1, C, 1, 1), 0.1)
    ref_out = act.to(torch.bfloat16)
    gsum = ga.float() + gcat[:, 8:].float()
    ref_gin = (gsum * torch.where(ref_out.float() > 0, 1.0, 0.1)).to(torch.bfloat16)
    yg, bg = y0.cuda().contiguous(memory_format=CL).requires_grad_(), bias.cuda().requires_grad_()
    a, b = ops.bias_leaky_relu_(yg * 1.0, bg, 0.1, consumers=2)
    assert a.dtype == torch.bfloat16 and a.stride() == yg.stride() and torch.equal(a.cpu(), ref_out)
    torch.autograd.backward([a, b], [ga.cuda().contiguous(memory_format=CL), gcat.cuda().contiguous(memory_format=CL)[:, 8:]])
    assert yg.grad.dtype == torch.bfloat16 and torch.equal(yg.grad.cpu(), ref_gin)
    want = ref_gin.float().sum((0, 2, 3))
    close(bg.grad, want, rtol=1e-4, atol=1e-4 * want.abs().max().item())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(16, 128, 64, 208), (2, 96, 8, 26), (3, 32, 4, 13), (1, 8, 1, 1)])
def test_bias_leaky_into_cat_buffers(ops, shape, dtype):
    """ops.bias_leaky_relu_into: two epilogues fill ONE cat buffer (pwc_tf.py:114-118's cat((x0, x1)) without the cat), the
    first one also in place; forward bit-equal to leaky_relu + torch.cat, backward (the gradient of the cat buffer, sliced in
    the kernel, plus the in-place consumer's gradient) against autograd on the plain ops."""
    N, C, H, W = shape
    C2 = max(4, C // 2 // 4 * 4)                                 # a second, narrower activation
    CL = torch.channels_last
    y0, y1 = rnd(181, shape).to(dtype), rnd(182, (N, C2, H, W)).to(dtype)
    b0, b1 = rnd(183, (C,), 0.3), rnd(184, (C2,), 0.3)
    w_cat, w_own = rnd(185, (N, C + C2, H, W)).to(dtype), rnd(186, shape).to(dtype)

    def act(y, b):                                              # fp32 math, one rounding (what the kernels do)
        return torch.nn.functional.leaky_relu(y.float() + b.view(1, -1, 1, 1), 0.1).to(dtype)
    ref0, ref1 = act(y0, b0), act(y1, b1)
    ref_cat = torch.cat((ref0, ref1), 1)
    # reference gradients: d/dy = (sum of the consumers' gradients) * slope mask, rounded once; bias = sum of the rounded values
    g0 = (w_cat[:, :C].float() + w_own.float()) * torch.where(ref0.float() > 0, 1.0, 0.1)
    g1 = w_cat[:, C:].float() * torch.where(ref1.float() > 0, 1.0, 0.1)
    g0, g1 = g0.to(dtype), g1.to(dtype)

    yg0 = y0.cuda().contiguous(memory_format=CL).requires_grad_()
    yg1 = y1.cuda().contiguous(memory_format=CL).requires_grad_()
    bg0, bg1 = b0.cuda().requires_grad_(), b1.cuda().requires_grad_()
    buf = torch.empty((N, C + C2, H, W), dtype=dtype, device='cuda').contiguous(memory_format=CL)
    x0, buf1, none = ops.bias_leaky_relu_into(yg0 * 1, bg0, 0.1, buf, 0, inplace=True)
    assert none is None and buf1.data_ptr() == buf.data_ptr() and torch.equal(x0.cpu(), ref0)
    none2, buf2, _ = ops.bias_leaky_relu_into(yg1 * 1, bg1, 0.1, buf1, C)
    assert none2 is None and torch.equal(buf2.cpu(), ref_cat)
    torch.autograd.backward([buf2, x0], [w_cat.cuda().contiguous(memory_format=CL), w_own.cuda().contiguous(memory_format=CL)])
    if dtype == torch.float32:
        close(yg0.grad, g0, rtol=1e-6, atol=1e-6); close(yg1.grad, g1, rtol=1e-6, atol=1e-6)
    else:
        assert torch.equal(yg0.grad.cpu(), g0) and torch.equal(yg1.grad.cpu(), g1)
    for bg, gref in ((bg0, g0), (bg1, g1)):
        want = gref.float().sum((0, 2, 3))
        close(bg.grad, want, rtol=1e-4, atol=1e-4 * max(want.abs().max().item(), 1e-3))
    # bad destinations are refused
    with pytest.raises(RuntimeError):
        ops.bias_leaky_relu_into(yg1 * 1, bg1, 0.1, buf.detach(), 2)                 # unaligned channel offset
    with pytest.raises(RuntimeError):
        ops.bias_leaky_relu_into(yg1 * 1, bg1, 0.1, buf.detach(), C + 4)             # slice runs past the buffer


def test_conv_block_matches_reference_block(ops):
    from unopticalflow_amd import conv as conv_hip
    torch.manual_seed(0)
    blk = conv_hip(7, 12, kernel_size=3, stride=2).cuda()
    ref = R.conv(7, 12, kernel_size=3, stride=2)
    ref.load_state_dict({k: v.cpu() for k, v in blk.state_dict().items()})     # same keys: 0.weight, 0.bias
    x = rnd(64, (2, 7, 20, 36))
    xc, xg = x.clone().requires_grad_(), x.cuda().requires_grad_()
    yr, yg = ref(xc), blk(xg)
    close(yg, yr, rtol=1e-4, atol=1e-5)
    go = rnd(65, tuple(yr.shape))
    yr.backward(go); yg.backward(go.cuda())
    close(xg.grad, xc.grad, rtol=1e-4, atol=1e-5)
    for (n, pg), (_, pr) in zip(blk.named_parameters(), ref.named_parameters()):
        close(pg.grad, pr.grad, rtol=1e-4, atol=1e-4 * pr.grad.abs().max().item(), what=n)


@pytest.mark.parametrize('shape', [(24, 3, 256, 832), (3, 3, 64, 128), (2, 1, 4, 4), (1, 3, 12, 20)])
def test_img_pyramid_vs_oracle(ops, shape):
    """generate_img_pyramid scales 1, 2 (model_flow_paper.py:54-60): box means; a few ulps of fp32
    summation-order difference vs ATen's adaptive_avg_pool2d are allowed (values are in [0,1))."""
    x = rnd(66, shape, uniform=True)
    ref = R.img_pyramid(x, 3)
    half, quarter = ops.img_pyramid(x.cuda())
    close(half, ref[1], rtol=0, atol=2e-7)
    close(quarter, ref[2], rtol=0, atol=3e-7)


@pytest.mark.parametrize('shape,size,mul', [((16, 2, 4, 13), (8, 26), 2.0), ((16, 2, 32, 104), (64, 208), 2.0), ((16, 2, 64, 208), (256, 832), 4.0),
                                            ((16, 2, 8, 26), (32, 104), 4.0), ((3, 2, 5, 7), (15, 7), 2.0), ((2, 3, 6, 9), (6, 9), 4.0),
                                            ((1, 2, 1, 1), (2, 2), 2.0), ((2, 2, 7, 4), (7, 24), 1.0)])
def test_upsample_scaled_vs_torch(ops, shape, size, mul):
    """The decoder's flow up-sampling (pwc_tf.py:119-177) as one kernel each way against the ops the reference (and the oracle,
    ref_cpu.py:274,285) runs on the CPU: F.interpolate(bilinear, align_corners=False) and the scalar multiply.  Same taps and
    weights; fp32 contraction order may differ by an ulp or two of the result (bar: 1e-6 relative to the largest value).  The
    gather backward is bitwise reproducible."""
    import torch.nn.functional as F
    x = rnd(71, shape, scale=3.0)
    g = rnd(72, shape[:2] + size)
    xr = x.clone().requires_grad_(True)
    if mul == 4.0:
        ref = F.interpolate(xr * 4.0, list(size), mode='bilinear')
    else:
        ref = F.interpolate(xr, list(size), mode='bilinear') * mul
    ref.backward(g)
    xg = x.cuda().requires_grad_(True)
    out = ops.upsample_bilinear_scaled(xg, size, mul)
    out.backward(g.cuda())
    close(out, ref, rtol=0, atol=1e-6 * ref.abs().max().item(), what='forward')
    close(xg.grad, xr.grad, rtol=0, atol=2e-6 * xr.grad.abs().max().item(), what='backward')
    first = xg.grad.clone()
    xg.grad = None
    ops.upsample_bilinear_scaled(xg, size, mul).backward(g.cuda())
    assert torch.equal(first, xg.grad)
    if shape[2] > 1:                                     # (a 1-row map up-samples to any height)
        with pytest.raises(ValueError):
            ops.upsample_bilinear_scaled(xg, (size[0] + 1, size[1]), mul)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('shape,with_res', [((32, 2, 64, 208), True), ((32, 2, 4, 13), False), ((3, 2, 7, 9), True), ((1, 2, 1, 1), True),
                                            ((2, 2, 300, 700), True)])
def test_flow_head_vs_torch(ops, shape, with_res, dtype):
    """predict_flow's bias, the NHWC -> NCHW hand-off and the residual (pwc_tf.py:93-94,130,171) in one pass: the same fp32
    additions in the same order as conv(bias) ... .float().contiguous() + up, so bit-equal; backward: the gradient re-laid out
    (rounded once for a bf16 stack), the bias gradient (1e-5: a tree sum here), the residual's gradient = the gradient."""
    CL = torch.channels_last
    y = rnd(400, shape, 2.0).to(dtype)
    bias = rnd(401, (2,), 0.5)
    res = rnd(402, shape) if with_res else None
    g = rnd(403, shape)
    ref = y.float() + bias.view(1, 2, 1, 1)
    if with_res:
        ref = ref + res
    yd = y.cuda().contiguous(memory_format=CL).requires_grad_(True)
    bd = bias.cuda().requires_grad_(True)
    rd = res.cuda().requires_grad_(True) if with_res else None
    out = ops.flow_head(yd, bd, rd)
    assert out.dtype == torch.float32 and out.is_contiguous() and torch.equal(out.cpu(), ref)
    out.backward(g.cuda())
    assert yd.grad.dtype == dtype and torch.equal(yd.grad.cpu(), g.to(dtype))
    want = g.sum((0, 2, 3))
    close(bd.grad, want, rtol=1e-5, atol=1e-5 * g.abs().sum().item() / 2)
    if with_res:
        assert torch.equal(rd.grad.cpu(), g)
    first = bd.grad.clone()
    bd.grad = None
    ops.flow_head(yd, bd, rd).backward(g.cuda())
    assert torch.equal(first, bd.grad)                          # fixed summation order
    with pytest.raises(ValueError):
        ops.flow_head(torch.zeros(1, 3, 4, 4, device='cuda'), bd)


def test_deferred_bias_gradients_are_the_same_bits(ops):
    """ABI 9: the second stage of every bias-gradient reduction of a backward pass as ONE launch at its end
    (ops.deferred_bias_grads, unflow_bias_grad_finalize_batch) -- fp32 / bf16 conv epilogues in both layouts, the cat-filling
    epilogue and the flow heads in one graph, 60 jobs (two batches), against the per-node second stages: identical bits."""
    CL = torch.channels_last
    torch.manual_seed(5)

    import contextlib

    def run(deferred):
        biases, total = [], 0
        try:
            for rep in range(10):
                for kind in range(6):
                    gen = torch.Generator().manual_seed(1000 + 10 * rep + kind)
                    if kind == 0:                           # NCHW fp32
                        y = torch.randn(3, 20, 9 + rep, 11, generator=gen).cuda().requires_grad_()
                        b = torch.randn(20, generator=gen).cuda().requires_grad_()
                        out = ops.bias_leaky_relu_(y * 1.0, b, 0.1)
                    elif kind == 1:                         # NHWC fp32
                        y = torch.randn(4, 32, 16, 13 + rep, generator=gen).cuda().contiguous(memory_format=CL).requires_grad_()
                        b = torch.randn(32, generator=gen).cuda().requires_grad_()
                        out = ops.bias_leaky_relu_(y * 1.0, b, 0.1)
                    elif kind == 2:                         # NHWC bf16
                        y = torch.randn(2, 16, 33, 17, generator=gen).cuda().bfloat16().contiguous(memory_format=CL).requires_grad_()
                        b = torch.randn(16, generator=gen).cuda().requires_grad_()
                        out = ops.bias_leaky_relu_(y * 1.0, b, 0.1).float()
                    elif kind == 3:                         # NCHW bf16
                        y = torch.randn(2, 6, 21, 30, generator=gen).cuda().bfloat16().requires_grad_()
                        b = torch.randn(6, generator=gen).cuda().requires_grad_()
                        out = ops.bias_leaky_relu_(y * 1.0, b, 0.1).float()
                    elif kind == 4:                         # cat-filling epilogue
                        y = torch.randn(2, 8, 12, 20, generator=gen).cuda().contiguous(memory_format=CL).requires_grad_()
                        b = torch.randn(8, generator=gen).cuda().requires_grad_()
                        buf = torch.zeros(2, 24, 12, 20, device='cuda').contiguous(memory_format=CL)
                        _, buf, _ = ops.bias_leaky_relu_into(y * 1.0, b, 0.1, buf, 8)
                        out = buf[:, 8:16]
                    else:                                   # flow head
                        y = torch.randn(5, 2, 40, 52, generator=gen).cuda().contiguous(memory_format=CL).requires_grad_()
                        b = torch.randn(2, generator=gen).cuda().requires_grad_()
                        out = ops.flow_head(y, b, None)
                    total = total + (out * torch.randn(out.shape, generator=gen).cuda()).sum()
                    biases.append(b)
            assert not ops.deferred_bias_grads.jobs and not ops.deferred_bias_grads.enabled     # opt-in: off outside the scope
            with (ops.deferred_bias_grads if deferred else contextlib.nullcontext()):
                total.backward()
                assert not ops.deferred_bias_grads.jobs and not ops.deferred_bias_grads.queued  # flushed by the engine's final callback
            torch.cuda.synchronize()
            if deferred:
                assert len(ops.deferred_bias_grads.last_addresses) == 60 and ops.deferred_bias_grads.adopted(biases)
            return [b.grad.clone() for b in biases]
        finally:
            assert not ops.deferred_bias_grads.enabled
    a, b = run(True), run(False)
    assert len(a) == 60
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), (i, (u - v).abs().max().item())
    # outside a backward pass (a direct call of an autograd.Function's backward, no engine) nothing is deferred
    assert not ops._in_backward()


@pytest.mark.parametrize('B,n', [(16, 3), (1, 1), (5, 4), (70, 2)])
def test_loss_bookkeeping_vs_torch(ops, B, n):
    """Model_flow.forward's sums over scales and directions (model_flow_paper.py:224-235) and the step's weighted batch means
    (train.py:147-150) as one launch each way: the same fp32 additions in the same association -> the loss pack is bit-equal to
    the eager form, and so are the gradients (each term's gradient is the loss gradient itself); the weighted mean to 1e-6
    (the batch sum is a tree here, a loop in ATen)."""
    terms = [[rnd(300 + 10 * k + s, (B if k == 3 else 2 * B,), uniform=True) for s in range(n)] for k in range(4)]
    gout = [rnd(350 + k, (B,)) for k in range(4)]
    ref_in = [[t.clone().requires_grad_(True) for t in ts] for ts in terms]
    ref = []
    for k in range(4):
        acc = 0
        for t in ref_in[k]:
            acc = acc + t
        ref.append(acc[B:] + acc[:B] if k < 3 else acc)
    torch.autograd.backward(ref, gout)
    dev_in = [[t.cuda().requires_grad_(True) for t in ts] for ts in terms]
    out = ops.loss_combine(*dev_in)
    torch.autograd.backward(list(out), [g.cuda() for g in gout])
    for k in range(4):
        assert out[k].shape == (B,) and torch.equal(out[k].cpu(), ref[k]), k
        for a, b in zip(dev_in[k], ref_in[k]):
            assert torch.equal(a.grad.cpu(), b.grad), k
    only = [[t.cuda().requires_grad_(True) for t in ts] for ts in terms]             # a loss nobody differentiates: zero gradient
    o = ops.loss_combine(*only)
    (o[0].sum() + o[3].sum()).backward()
    assert all(float(t.grad.abs().max()) == 0.0 for t in only[1] + only[2]) and torch.equal(only[0][0].grad.cpu(), torch.ones(2 * B))

    w = [0.15, 0.85, 10.0, 0.01]
    vr = [t.detach().clone().requires_grad_(True) for t in ref]
    want = sum(wk * t.mean() for wk, t in zip(w, vr))
    want.backward()
    vd = [t.detach().cuda().requires_grad_(True) for t in ref]
    got = ops.weighted_mean_sum(vd, w)
    got.backward()
    close(got, want, rtol=1e-6, atol=0)
    for a, b in zip(vd, vr):
        close(a.grad, b.grad, rtol=1e-6, atol=0)
    with pytest.raises(ValueError):
        ops.loss_combine(dev_in[0], dev_in[1], dev_in[2], dev_in[0])


def test_kernel_exact_timing_slots(ops):
    """bench.py's roofline legs: a timed C call carries an event pair on its kernels (unflow_timing_begin /
    hipExtLaunchKernelGGL).  The slot's time is positive, not longer than a hipEventRecord bracket around the same call,
    and an entry point with two launches (zero-fill + kernel) reports one span."""
    f1, f2 = rnd(1, (8, 32, 64, 208)).cuda(), rnd(2, (8, 32, 64, 208)).cuda()
    fl = (rnd(3, (8, 2, 64, 208)) * 2).cuda().requires_grad_()
    for _ in range(2):
        ops.corr(f1, f2, 4)
    torch.cuda.synchronize()
    ops.kernel_timer.enable(('unflow_corr_fwd', 'unflow_warp_bwd'), reserve=8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.corr(f1, f2, 4)
    e1.record()
    x = f2.clone().requires_grad_()
    ops.warp_flow(x, fl, fused_backward=False).sum().backward()          # (the two-launch form: zero-fill + scatter kernel)
    torch.cuda.synchronize()
    ops.kernel_timer.disable()
    rows = {r['entry']: r for r in ops.kernel_timer.rows()}
    assert set(rows) == {'unflow_corr_fwd', 'unflow_warp_bwd'}
    us = rows['unflow_corr_fwd']['avg_us']
    assert 2.0 < us <= e0.elapsed_time(e1) * 1e3 + 0.5
    assert rows['unflow_warp_bwd']['launches'] == 1 and rows['unflow_warp_bwd']['avg_us'] > 2.0
    assert ops.kernel_timer.rows() == []                      # slots are handed back


# ------------------------------------------------------------------------------------ randomised shapes
def test_fuzz_shapes_against_oracle(ops):
    """30 random (B, C, H, W, d) draws: every kernel-selection branch (tile / ring / group / generic paths,
    W % 4 != 0, W < 64, C < ring stage) must agree with the oracle, forward and backward."""
    rng = np.random.default_rng(2024)
    for trial in range(30):
        B = int(rng.integers(1, 5))
        C = int(rng.choice([1, 2, 3, 5, 8, 17, 32, 40]))
        H = int(rng.choice([3, 8, 9, 16, 31, 64, 100]))
        W = int(rng.choice([4, 13, 52, 64, 97, 104, 208, 260]))
        d = int(rng.choice([1, 2, 4, 4, 4, 8]))
        if B * C * H * W * (2 * d + 1) ** 2 > 6e7:
            continue
        f1c = rnd(1000 + trial, (B, C, H, W)).requires_grad_()
        f2c = rnd(2000 + trial, (B, C, H, W)).requires_grad_()
        cv_ref = R.corr_naive(f1c, f2c, d)
        gout = rnd(3000 + trial, tuple(cv_ref.shape))
        cv_ref.backward(gout)
        f1, f2 = dev(f1c.detach()).requires_grad_(), dev(f2c.detach()).requires_grad_()
        cv = ops.corr(f1, f2, d)
        what = 'corr B%d C%d H%d W%d d%d' % (B, C, H, W, d)
        close(cv, cv_ref, rtol=1e-5, atol=2e-6, what=what)
        cv.backward(dev(gout))
        close(f1.grad, f1c.grad, rtol=1e-4, atol=1e-5, what=what)
        close(f2.grad, f2c.grad, rtol=1e-4, atol=1e-5, what=what)
        # warp (+mask) on the same geometry
        ac = bool(trial & 1)
        Cw = min(C, 8)
        xw = rnd(4000 + trial, (B, Cw, H, W), uniform=True)
        fl = rnd(5000 + trial, (B, 2, H, W), float(rng.choice([0.5, 3.0, 12.0])))
        y_ref = R.warp_flow(xw, fl, True, ac)
        m_ref = R.warp_mask(xw.shape, fl, ac).numpy()
        y, m = ops.warp_flow_masked(xw.cuda(), fl.cuda(), align_corners=ac)
        assert np.array_equal(m.cpu().numpy(), m_ref), 'mask ' + what
        close(y, y_ref, rtol=1e-5, atol=1e-6, what='warp ' + what)
