"""GPU parity tests added in round 5 that have not yet run on an MI355X (the GPU lease was closed from outside the build before they
could, and stayed closed through round 6) -- kept in a file that sorts LAST so that a first run under ``-x`` reaches
them after everything that has been validated.  Same helpers, bars and oracle as tests/test_hip_ops.py.  Tests of what the DEFAULT path
runs (round-4 kernels under new test code) carry no mark and fail hard; tests of off-by-default paths carry test_hip_ops.UNVALIDATED."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from test_hip_ops import T, close, dev, ops, rnd, _structured_flow, UNVALIDATED      # noqa: F401  (``ops`` is the module-scoped fixture)
from test_hip_ops import test_corr_backward_on_the_matrix_cores as _matrix_core_backward_case
from test_hip_ops import test_corr_small_map_backward as _small_map_case, test_corr_d8_full_pyramid as _d8_pyramid_case
from oracle_cache import corr_case

pytestmark = [pytest.mark.gpu]


own_process = pytest.mark.own_process
_batch = {}


def _child(ids, timeout, junit=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '--runxfail', '-p', 'no:cacheprovider'] + (['--junitxml', junit] if junit else ['-x']) + list(ids)
    try:
        return subprocess.run(cmd, cwd=root, env=dict(os.environ, UNFLOW_ZZ_CHILD='1'), capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        return subprocess.CompletedProcess(cmd, 124, stdout=(e.stdout or b'').decode(errors='replace') if isinstance(e.stdout, bytes) else (e.stdout or ''), stderr='timed out')


_CHILD_BUDGET_S = [600.0]        # wall seconds all child processes of a session may take together (the driver caps the whole suite at 1200 s)


def _ran_in_a_child(request):
    """Tests that launch device code which has NEVER run (the `_ms` kernels of csrc/multiscale.h) do it outside the pytest
    process of the validated suite: a GPU fault there fails these tests instead of taking that process down.  All `own_process` tests of
    the session run in ONE child (a process start, torch import and library load once, not a dozen times); a test the child did not get to
    (it died on the way) is run again in a child of its own -- as long as the session's child budget lasts: a child that hangs costs its
    300 s once, not once per test.  -> True when a child ran the test (and it passed: a failure is raised here); False inside the child,
    where the body runs."""
    import time
    if os.environ.get('UNFLOW_ZZ_CHILD'):
        return False
    me = request.node.nodeid
    if not _batch:
        import tempfile
        import xml.etree.ElementTree as ET
        ids = [it.nodeid for it in request.session.items if it.get_closest_marker('own_process')] or [me]
        with tempfile.TemporaryDirectory() as tmp:
            junit = os.path.join(tmp, 'isolated.xml')
            t0 = time.monotonic()
            r = _child(ids, 300, junit)
            _CHILD_BUDGET_S[0] -= time.monotonic() - t0
            _batch['log'] = (r.stdout[-3000:], r.stderr[-1500:])
            _batch['timed_out'] = r.returncode == 124
            if os.path.exists(junit):
                try:
                    cases = list(ET.parse(junit).getroot().iter('testcase'))
                except ET.ParseError:                                   # (a child killed while it wrote the report)
                    cases = []
                for case in cases:
                    bad = [c for c in case if c.tag in ('failure', 'error')]
                    skipped = any(c.tag == 'skipped' for c in case)
                    _batch[case.get('name')] = ('failed', (bad[0].get('message') or '')[:300] + '\n' + (bad[0].text or '')[-2500:]) if bad else (('skipped', '') if skipped else ('passed', ''))
    name = me.split('::')[-1]
    if name in _batch:
        state, why = _batch[name]
        if state == 'skipped':
            pytest.skip('skipped in the child process')
        assert state == 'passed', why
        return True
    # the batch never reached this test
    assert not _batch.get('timed_out'), 'the shared child process timed out before this test ran: %s' % (_batch.get('log'),)
    assert _CHILD_BUDGET_S[0] > 60.0, 'no child-process time left in this session (batch: %s)' % (_batch.get('log'),)
    t0 = time.monotonic()
    r = _child([me], int(min(300.0, _CHILD_BUDGET_S[0])))
    _CHILD_BUDGET_S[0] -= time.monotonic() - t0
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:], 'batch: %s' % (_batch.get('log'),))
    return True


@pytest.mark.parametrize('kind', ['smooth', 'mixed', 'outside', 'edge', 'noise'])
@pytest.mark.parametrize('B,C,h,w', [(8, 32, 112, 256), (16, 32, 64, 208), (8, 64, 56, 128)])
def test_warp_backward_auto_picked_one_pass_vs_oracle(ops, kind, B, C, h, w):
    """the one-pass (gather) backward is what the op picks BY ITSELF at level 2 of 1024x448 (bs 4: [8,32,112,256]) and of
    the 832x256 step ([16,32,64,208]) -- so it is compared with the oracle's grid_sample chain (net_utils.py:16-46) at exactly those
    launches, unforced, for the five flow kinds of the tile tests; level 3 of 1024x448 ([8,64,56,128]: served, not picked) forced."""
    lib = __import__('unopticalflow_amd._lib', fromlist=['load']).load()
    picked = lib.unflow_warp_bwd_fused_supported(B, C, h, w) == 2
    assert picked == (C == 32)
    xc = rnd(311 + C, (B, C, h, w)).requires_grad_()
    fc = _structured_flow(B, h, w, kind, seed=h * w + C + 1).requires_grad_()
    g = rnd(313 + C, (B, C, h, w))
    yr = R.warp_flow(xc, fc)
    yr.backward(g)
    x, f = dev(xc.detach()).requires_grad_(), dev(fc.detach()).requires_grad_()
    ops.kernel_timer.enable(True)
    y = ops.warp_flow(x, f, fused_backward=None if picked else True)
    y.backward(dev(g))
    torch.cuda.synchronize()
    ops.kernel_timer.disable()
    names = {r['entry'] for r in ops.kernel_timer.rows()}
    assert 'unflow_warp_bwd_fused' in names and 'unflow_warp_bwd' not in names, names
    close(y, yr, rtol=1e-5, atol=1e-6, what='fwd %s' % kind)
    close(x.grad, xc.grad, rtol=1e-4, atol=2e-5, what='gsrc %s' % kind)
    close(f.grad, fc.grad, rtol=1e-4, atol=2e-5 * max(fc.grad.abs().max().item(), 1e-6), what='gflow %s' % kind)


def _flat_patch_images(B, h, w, seed):
    """An image / warped-image pair with the regions where a window-SUM formulation of SSIM cancels hardest: saturated flat patches
    (x = 1.0 against y = 1.0, 1 - 1/255, 0.95: KITTI sky), dark flat patches, a step edge, on top of the usual noise."""
    rng = np.random.default_rng(seed)
    x = rng.random((B, 3, h, w), dtype=np.float32)
    y = np.clip(x + 0.1 * rng.standard_normal((B, 3, h, w)).astype(np.float32), 0, 1)
    hq, wq = max(h // 4, 2), max(w // 4, 2)
    for k, dy in enumerate((0.0, 1.0 / 255.0, 0.05)):               # bright flat patches, x != y by a constant
        x[:, :, : hq, k * wq: (k + 1) * wq] = 1.0
        y[:, :, : hq, k * wq: (k + 1) * wq] = np.float32(1.0 - dy)
    x[:, :, hq: 2 * hq, : wq] = 0.0; y[:, :, hq: 2 * hq, : wq] = 0.0                    # dark, equal
    x[:, :, hq: 2 * hq, wq: 2 * wq] = 2.0 / 255.0; y[:, :, hq: 2 * hq, wq: 2 * wq] = 0.0   # dark, unequal
    x[:, :, 2 * hq:, 2 * wq:] = 0.25; x[:, :, 2 * hq:, 3 * wq:] = 0.9                     # a step edge in x only
    y[:, :, 2 * hq:, 2 * wq:] = 0.5
    wgt = rng.random((B, 1, h, w), dtype=np.float32) * 2.0
    wgt[:, :, : hq] = 1.0 + (rng.random((B, 1, hq, w), dtype=np.float32) > 0.5)       # weights 1 or 2 over the flat patches (the model's range is [0, 2])
    return T(x), T(y), T(wgt)


@pytest.mark.parametrize('shape', [(16, 256, 832), (2, 37, 131), (3, 128, 416)])
def test_ssim_loss_flat_patches_and_edges(ops, shape):
    """the loss's fast path forms SSIM from window sums; bright flat regions are where "K2 - Q" would lose K2's low bits.
    Forward and gradient against the oracle (model_flow_paper.py:137-148, pytorch_ssim/ssim.py:4-20) at the 1e-4 bar."""
    B, h, w = shape
    img, wp, wgt = _flat_patch_images(B, h, w, seed=h + w)
    gl = rnd(77, (B,))
    wc = wp.clone().requires_grad_()
    lr = R.ssim_loss(img, wc, wgt)
    (lr * gl).sum().backward()
    wg = dev(wp).requires_grad_()
    lg = ops.ssim_loss(dev(img), wg, dev(wgt))
    close(lg, lr, rtol=1e-4, atol=0, what='ssim loss, flat patches')
    (lg * dev(gl)).sum().backward()
    # the gradient against a float64 evaluation: on flat patches the reference's own fp32 gradient is itself off (sigma = E[x^2] - mu^2 cancels),
    # so the allowance is the reference's distance from the truth (twice that, or 2e-5 of the largest element) -- the kernels may not be further
    # from the truth than the reference is.  (At [16,3,256,832] nine of ten million elements of the host-executed kernels sat 1.2e-5 of the
    # largest element from the fp32 reference, inside this allowance.)
    w64 = wp.detach().double().clone().requires_grad_()
    (R.ssim_loss(img.detach().double(), w64, wgt.detach().double()) * gl.double()).sum().backward()
    truth, ref = w64.grad, wc.grad.double()
    big = truth.abs().max().item()
    allowance = max(2.0 * (ref - truth).abs().max().item(), 2e-5 * big)
    err = (wg.grad.detach().cpu().double() - truth).abs().max().item()
    assert err <= allowance, (err / big, allowance / big)
    # every patch on its own (the mean over a whole map would hide a systematic error of one region): weight 1 inside, 0 outside, so the
    # loss is the patch's mean of (1 - SSIM) / 2.  Bar: 1e-4 of SSIM's own range -- a flat patch with x != y has sigma = 0 exactly, and
    # the reference's E[xy] - mu_x mu_y in fp32 already carries ~7e-5 of C2 there, with one sign over the whole patch
    hq, wq = max(h // 4, 2), max(w // 4, 2)
    for (y0, x0) in ((0, 0), (0, wq), (0, 2 * wq), (hq, 0), (hq, wq), (2 * hq, 2 * wq)):
        m = torch.zeros(B, 1, h, w)
        m[:, :, y0: y0 + hq, x0: x0 + wq] = 1.0
        close(ops.ssim_loss(dev(img), dev(wp), dev(m)), R.ssim_loss(img, wp, m), rtol=1e-4, atol=5e-5, what='patch at %d, %d' % (y0, x0))


@UNVALIDATED
def test_deferred_loss_sums_are_the_same_bits(ops):
    """Round 5: inside ``with ops.deferred_loss_sums:`` the four per-sample loss reductions stop after their partial sums and ONE
    ``unflow_loss_finalize_batch`` launch (triggered by loss_combine, their reader) finishes all of them: values and gradients equal the
    immediate second stages bit for bit, at an even and an odd width (both SSIM kernels), and the launch count says so."""
    for (B, h, w) in ((3, 40, 72), (2, 33, 57), (2, 64, 208)):
        img = dev(rnd(51, (B, 3, h, w), uniform=True))
        st0 = torch.cat(((img + dev(rnd(52, (B, 3, h, w), 0.1))).clamp(0, 1), (img + dev(rnd(53, (B, 3, h, w), 0.1))).clamp(0, 1)))
        st0[:B, :, 3:9, 5:17] = 0.0
        flows0 = dev(rnd(54, (2 * B, 2, h, w), 3.0))
        gl = [dev(rnd(55 + k, (B,))) for k in range(4)]
        res = []
        for deferred in (False, True):
            st, fl = st0.clone().requires_grad_(), flows0.clone().requires_grad_()
            before = ops.deferred_loss_sums.launches
            ctx = ops.deferred_loss_sums if deferred else __import__('contextlib').nullcontext()
            with ctx:
                diff, wgt = ops.occ_weight_stacked(img, st)
                terms = ([ops.masked_mean(diff, wgt)], [ops.ssim_loss(img, st, wgt)], [ops.smooth2_loss(fl, img)],
                         [ops.consis_loss(fl[B:], fl[:B], wgt[B:])])
                packed = ops.loss_combine(*terms)
            assert ops.deferred_loss_sums.launches - before == (1 if deferred else 0) and not ops.deferred_loss_sums.jobs
            sum((p * g).sum() for p, g in zip(packed, gl)).backward()
            res.append([t.clone() for t in packed] + [st.grad.clone(), fl.grad.clone()] + [t[0].clone() for t in terms])
        for a, b in zip(*res):
            assert torch.equal(a, b)


@UNVALIDATED
@pytest.mark.parametrize('d,B,C,h,w', [(8, 16, 96, 32, 52), (8, 6, 16, 37, 44)])
def test_corr_backward_on_the_matrix_cores_ragged_d8(ops, d, B, C, h, w):
    """The d = 8 shapes of test_hip_ops.py::test_corr_backward_on_the_matrix_cores that had not run when the lease closed: a last
    segment of 4 / 12 pixels, three channel super-groups, a row count that is not a multiple of the chunk, 16 channels."""
    _matrix_core_backward_case(ops, d, B, C, h, w)


@UNVALIDATED
@own_process
@pytest.mark.parametrize('B,h,w', [(2, 64, 208), (3, 60, 104), (8, 256, 832)])
def test_multiscale_losses_are_the_same_bits(ops, B, h, w, request):
    """Round 5 (csrc/multiscale.h, C ABI 11): every loss of the scale loop as ONE launch over the three scales -- the `_ms` kernels
    include the single-scale kernels' bodies and run their grids, so losses, saved sums and every gradient equal the per-scale ops
    bit for bit, inside and outside ``deferred_loss_sums``; 5 forward + 1 second-stage + 5 backward loss launches instead of 31.
    Shapes: tiles that divide, a ragged one (60 x 104 -> 15 x 26 at scale 2: partial tiles, SSIM 8-row chunks), and the train step's."""
    if _ran_in_a_child(request):
        return
    n = 3
    hs, ws = [h >> s for s in range(n)], [w >> s for s in range(n)]
    imgs = [dev(rnd(71 + s, (B, 3, hs[s], ws[s]), uniform=True)) for s in range(n)]
    warped0 = [torch.cat(((imgs[s] + dev(rnd(74 + s, (B, 3, hs[s], ws[s]), 0.1))).clamp(0, 1),
                          (imgs[s] + dev(rnd(77 + s, (B, 3, hs[s], ws[s]), 0.1))).clamp(0, 1))) for s in range(n)]
    for s in range(n):
        warped0[s][:B, :, 1:5, 2:9] = 0.0                               # an all-zero (invalid) region in one direction
    flows0 = [dev(rnd(80 + s, (2 * B, 2, hs[s], ws[s]), 3.0 / (1 << s))) for s in range(n)]
    gl = [dev(rnd(90 + k, (B,))) for k in range(4)]
    assert ops.multiscale_supported(imgs, warped0)
    res = {}
    for form in ('per scale', 'one launch', 'one launch, halves by offset', 'one launch, sums at once'):
        wp = [t.clone().requires_grad_() for t in warped0]
        fl = [t.clone().requires_grad_() for t in flows0]
        halves = [f.split(B) for f in fl]
        fb, ff = [x[0] for x in halves], [x[1] for x in halves]
        ops.kernel_timer.enable(True)
        ctx = __import__('contextlib').nullcontext() if form.endswith('at once') else ops.deferred_loss_sums
        with ctx:
            if form == 'per scale':
                pixel, ssim, smooth, consis = [], [], [], []
                for s in range(n):
                    diff, wgt = ops.occ_weight_stacked(imgs[s], wp[s])
                    pixel.append(ops.masked_mean(diff, wgt)); ssim.append(ops.ssim_loss(imgs[s], wp[s], wgt))
                    smooth.append(ops.smooth2_loss(fl[s], imgs[s])); consis.append(ops.consis_loss(ff[s], fb[s], wgt[B:]))
            elif form.endswith('by offset'):                              # (what Model_flow.forward uses: no split nodes behind the consistency term)
                pixel, ssim, smooth, consis = ops.multiscale_losses(imgs, wp, fl)
            else:
                pixel, ssim, smooth, consis = ops.multiscale_losses(imgs, wp, fl, ff, fb)
            packed = ops.loss_combine(pixel, ssim, smooth, consis)
        sum((p * g).sum() for p, g in zip(packed, gl)).backward()
        torch.cuda.synchronize()
        ops.kernel_timer.disable()
        rows = ops.kernel_timer.rows()
        launches = sum(r['launches'] for r in rows if r['entry'] not in ('unflow_loss_combine_fwd', 'unflow_loss_combine_bwd'))
        res[form] = ([t.clone() for t in packed] + [t.grad.clone() for t in wp] + [t.grad.clone() for t in fl] +
                     [t.clone() for t in pixel + ssim + smooth + consis], launches, {r['entry'] for r in rows})
    assert res['per scale'][1] == 10 * n + 1 and res['one launch'][1] == 11, (res['per scale'][1], res['one launch'][1])
    assert all(e.endswith('_ms') or e in ('unflow_loss_finalize_batch', 'unflow_loss_combine_fwd', 'unflow_loss_combine_bwd') for e in res['one launch'][2])
    assert res['one launch, halves by offset'][1] == 11
    for form in ('one launch', 'one launch, halves by offset', 'one launch, sums at once'):
        for k, (a, b) in enumerate(zip(res['per scale'][0], res[form][0])):
            assert torch.equal(a, b), (form, k, float((a - b).abs().max()))


@UNVALIDATED
@own_process
def test_multiscale_losses_in_the_model(ops, request):
    """Model_flow.multiscale_losses: the same loss pack and the same loss-side gradients (the flows' and the warped images' come out of the
    loss kernels; compared here through the total gradient norm, which also crosses MIOpen's run-to-run level) as the per-scale loop."""
    if _ran_in_a_child(request):
        return
    from unopticalflow_amd import get_model, generate_loss_weights_dict
    cfg = R.default_cfg()
    inputs = R.synthetic_triplets(2, 128, 192, seed=5, structured=True).cuda()
    packs, norms = [], []
    for ms in (False, True):
        model = get_model('flow')(cfg).cuda()
        model.load_state_dict(R.seeded_state_dict(model, 1234, 0.25))
        model.multiscale_losses = ms
        pack = model(inputs)
        R.total_loss(pack, generate_loss_weights_dict(cfg)).backward()
        packs.append({k: v.detach().clone() for k, v in pack.items()})
        norms.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))))
    for k in packs[0]:
        assert torch.equal(packs[0][k], packs[1][k]), k
    assert abs(norms[0] - norms[1]) <= 1e-3 * norms[0], norms


@UNVALIDATED
@own_process
@pytest.mark.parametrize('ac', [False, True])
def test_multiscale_image_warps_are_the_same_bits(ops, ac, request):
    """ops.warp_flow_masked_pyramid (C ABI 11: unflow_warp_fwd_ms / unflow_warp_bwd_ms) against ops.warp_flow(use_mask=True) per scale:
    warped images, the flow gradients and -- the integer half of the parity bar -- the masks, bit for bit; a ragged width (W = 100: a
    36-pixel last row segment) and flows that leave the image."""
    if _ran_in_a_child(request):
        return
    B, n = 3, 3
    for (h, w) in ((64, 208), (40, 100)):
        hs, ws = [h >> s for s in range(n)], [w >> s for s in range(n)]
        imgs = [dev(rnd(101 + s, (B, 3, hs[s], ws[s]), uniform=True)) for s in range(n)]
        flows0 = [dev(rnd(104 + s, (B, 2, hs[s], ws[s]), 6.0 / (1 << s))) for s in range(n)]
        gout = [dev(rnd(107 + s, (B, 3, hs[s], ws[s]))) for s in range(n)]
        fa = [f.clone().requires_grad_() for f in flows0]
        fb = [f.clone().requires_grad_() for f in flows0]
        per = [ops.warp_flow_masked(imgs[s], fa[s], align_corners=ac) for s in range(n)]
        sum((o * g).sum() for (o, _), g in zip(per, gout)).backward()
        ms = ops._WarpMaskedMS.apply(n, ac, *imgs, *fb)
        sum((o * g).sum() for o, g in zip(ms[:n], gout)).backward()
        for s in range(n):
            assert torch.equal(per[s][0], ms[s]) and torch.equal(per[s][1], ms[n + s]) and torch.equal(fa[s].grad, fb[s].grad), (h, w, s)
            assert 0 < int(per[s][1].sum()) < per[s][1].numel()               # (both mask values occur)
        assert all(torch.equal(a, b) for a, b in zip(ops.warp_flow_masked_pyramid(imgs, flows0, ac), [p[0] for p in per]))


@UNVALIDATED
def test_fused_warp_corr_at_level_4_only(ops):
    """cfg.fused_warp_corr_levels = '4': only the 16 x 52 level takes the fused warp + cost-volume kernel (the op of
    test_fused_warp_corr_model_matches_golden, chosen per level; level 5's width 26 is not a multiple of 4, the fused kernel does not
    serve it) -- same losses as the two-kernel path to 1e-5."""
    from unopticalflow_amd import get_model
    x = R.synthetic_triplets(2, 256, 832, seed=0, structured=True).cuda()
    packs = []
    for lv in (None, '4'):
        model = get_model('flow')(R.default_cfg(fused_warp_corr_levels=lv)).cuda()
        model.load_state_dict(R.seeded_state_dict(model, 1234, 0.25))
        ops.kernel_timer.enable(('unflow_warp_corr_fwd',))
        with torch.no_grad():
            packs.append({k: v.clone() for k, v in model(x).items()})
        torch.cuda.synchronize()
        ops.kernel_timer.disable()
        assert sum(r['launches'] for r in ops.kernel_timer.rows()) == (1 if lv else 0)       # (both flow directions ride one 2B pass of the decoder)
    for k in packs[0]:
        close(packs[1][k], packs[0][k], rtol=1e-5, what=k)


@UNVALIDATED
def test_pyramid_handoff_as_two_tensors_on_the_gpu(ops):
    """ops.to_nchw_split (validated kernels behind a new autograd node; wiring checked on the CPU in tests/test_ops_plumbing_cpu.py):
    values and the gradient against cat + split in torch at a pyramid level's shape, fp32 and bf16; and Model_flow.split_handoff gives
    the same loss pack as the split it replaces."""
    Bp, C, H, W = 4, 64, 32, 104
    B = 3 * Bp
    for dt, tol in ((torch.float32, 0.0), (torch.bfloat16, 1e-2)):
        x0 = dev(rnd(120, (B, C, H, W))).to(dt).contiguous(memory_format=torch.channels_last)
        ga, gb = dev(rnd(121, (2 * Bp, C, H, W))), dev(rnd(122, (2 * Bp, C, H, W)))
        xr = x0.clone().requires_grad_()
        ra, rb = torch.cat((xr, xr[B - Bp:]), 0).float().contiguous().split((2 * Bp, 2 * Bp))
        ((ra * ga).sum() + (rb * gb).sum()).backward()
        x = x0.clone(memory_format=torch.channels_last).requires_grad_()
        a, b = ops.to_nchw_split(x, 2 * Bp, Bp)
        ((a * ga).sum() + (b * gb).sum()).backward()
        assert torch.equal(a, ra) and torch.equal(b, rb)
        close(x.grad.float(), xr.grad.float(), rtol=tol, atol=tol * float(xr.grad.float().abs().max()) + (0 if tol else 1e-6), what=str(dt))
    from unopticalflow_amd import get_model
    inputs = R.synthetic_triplets(2, 128, 192, seed=5, structured=True).cuda()
    packs = []
    for sh in (False, True):
        model = get_model('flow')(R.default_cfg(channels_last=True)).cuda()
        model.load_state_dict(R.seeded_state_dict(model, 1234, 0.25))
        model.split_handoff = sh
        pack = model(inputs)
        sum(v.mean() for v in pack.values()).backward()
        packs.append({k: v.detach().clone() for k, v in pack.items()})
    for k in packs[0]:
        assert torch.equal(packs[0][k], packs[1][k]), k


def _loss_section_flow_gradients(g, dtype):
    """The oracle's evaluation of g5_loss_section.npz's flow gradients in `dtype` (float64: the truth both fp32 sides are measured against)."""
    T_ = torch.from_numpy
    imgl, img, imgr = (T_(g[k]).to(dtype) for k in ('imgl', 'img', 'imgr'))
    fb = [T_(g['flow_b%d' % s]).to(dtype).requires_grad_() for s in range(4)]
    ff = [T_(g['flow_f%d' % s]).to(dtype).requires_grad_() for s in range(4)]
    pl, pc, pr = R.img_pyramid(imgl, 4), R.img_pyramid(img, 4), R.img_pyramid(imgr, 4)
    lp = ls = lsm = lc = 0
    for s in range(3):
        from_l, from_r = R.warp_flow(pl[s], fb[s], True), R.warp_flow(pr[s], ff[s], True)
        d_l, d_r, w_b, w_f, _, _ = R.diff_weight(pc[s], from_l, from_r)
        lp = lp + R.masked_l1(d_r, w_f) + R.masked_l1(d_l, w_b)
        ls = ls + R.ssim_loss(pc[s], from_r, w_f) + R.ssim_loss(pc[s], from_l, w_b)
        lsm = lsm + R.grad2_error(ff[s] / 20.0, pc[s]) + R.grad2_error(fb[s] / 20.0, pc[s])
        lc = lc + R.consis_loss(ff[s], fb[s], w_f)
    sum((l * T_(g['gl%d' % k]).to(dtype)).sum() for k, l in enumerate((lp, ls, lsm, lc))).backward()
    return [torch.cat((fb[s].grad, ff[s].grad)).double().numpy() for s in range(3)]


@own_process
@pytest.mark.parametrize('ms', [False, pytest.param(True, marks=UNVALIDATED)])
def test_loss_section_against_the_reference_fixture(ops, golden, monkeypatch, request, ms):
    """g5_loss_section.npz -- the REFERENCE's own run of model_flow_paper.py:227-251 from given flows, on frames with saturated / dark
    flat patches and a step edge -- against Model_flow.forward from the flows on, i.e. the HIP image pyramid, masked warps, occlusion
    weights and the four loss kernels with their backward passes, in both launch forms: losses at 1e-4 rel (north_star's bar), the
    masked warped images, flow gradients at 1e-4 + 2e-6 of the largest."""
    if ms and _ran_in_a_child(request):
        return
    from unopticalflow_amd import get_model
    g = golden('g5_loss_section.npz')
    imgl, img, imgr = (dev(g[k]) for k in ('imgl', 'img', 'imgr'))
    inputs = torch.cat((imgl, img, imgr), 2)
    model = get_model('flow')(R.default_cfg()).cuda()
    model.multiscale_losses = ms
    fl = [torch.cat((dev(g['flow_b%d' % s]), dev(g['flow_f%d' % s]))).requires_grad_() for s in range(4)]
    monkeypatch.setattr(model, '_flows', lambda *a, **k: fl)
    pack = model(inputs)
    keys = ('loss_pixel', 'loss_ssim', 'loss_flow_smooth', 'loss_flow_consis')
    sum((pack[k] * dev(g['gl%d' % i])).sum() for i, k in enumerate(keys)).backward()
    for k in keys:
        close(pack[k], g[k], rtol=1e-4, what='%s (multiscale_losses=%s)' % (k, ms))
    # flow gradients: these frames are ill-conditioned ON PURPOSE (saturated flat patches: SSIM's variances cancel) -- the reference's own fp32
    # gradient is 5.7e-4 / 9e-5 / 1.6e-5 of the largest element away from a float64 evaluation at scales 0 / 1 / 2, the kernels (sum-space SSIM)
    # 2.0e-4 / 3.8e-5 / 3.9e-5 (measured on the host-executed kernels).  So the bar is the float64 oracle, and the allowance the reference's own
    # distance from it (twice that, or 1e-4 of the largest element): the kernels may not be further from the truth than the reference is
    truth = _loss_section_flow_gradients(g, torch.float64)
    for s in range(3):
        ref = np.concatenate((g['g_flow_b%d' % s], g['g_flow_f%d' % s])).astype(np.float64)
        big = float(np.abs(truth[s]).max())
        allowance = max(2.0 * float(np.abs(ref - truth[s]).max()), 1e-4 * big)
        got = fl[s].grad.detach().cpu().double().numpy()
        assert float(np.abs(got - truth[s]).max()) <= allowance, (s, float(np.abs(got - truth[s]).max()) / big, allowance / big)
    assert fl[3].grad is None


@UNVALIDATED
@own_process
@pytest.mark.parametrize('d,B,C,h,w', [(4, 16, 128, 8, 26), (4, 16, 196, 4, 13), (4, 3, 5, 7, 11), (4, 1, 2, 30, 34), (4, 2, 1, 3, 3), (4, 4, 128, 14, 32),
                                       (8, 2, 128, 8, 26), (8, 2, 196, 4, 13)])
def test_corr_small_map_backward_rows_through_registers(ops, request, d, B, C, h, w):
    """Round 6, csrc/corr_small_rows.h (``ops.corr(..., backward='fp32_next')``: maps of <= 1024 pixels at any radius -- levels 5 / 6 at d = 8 ran one
    lane per output element): the bodies and bars of test_corr_small_map_backward (d = 4) and test_corr_d8_full_pyramid (d = 8) of
    tests/test_hip_ops.py with that arithmetic, in a process of its own -- the kernel has been executed on the build host only."""
    if _ran_in_a_child(request):
        return
    if d == 4:
        _small_map_case(ops, B, C, h, w, backward='fp32_next')
    else:
        _d8_pyramid_case(ops, C, h, w, 'fp32_next')
