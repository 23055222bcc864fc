"""The Python side of the round-5 multi-scale operators, run on the CPU through tests/abi_emulator.py (a stand-in for the C entries, built
on the oracle): argument order, per-scale pointer tables, output shapes, what is saved for the backward and where every gradient is
returned -- the part of ops.py that otherwise first runs in a GPU session.  The values come from the oracle either way, so agreement
here checks the WIRING (which tensor feeds which loss, (bwd | fwd) stacking, detached operands), not the kernels."""
import numpy as np
import torch

from oracle import ref_cpu as R
from abi_emulator import patched


def _rnd(seed, shape, scale=1.0, uniform=False):
    rng = np.random.default_rng(seed)
    a = rng.random(shape, dtype=np.float32) if uniform else rng.standard_normal(shape).astype(np.float32)
    return torch.from_numpy(a * np.float32(scale))


def _scene(B, h, w, n=3):
    hs, ws = [h >> s for s in range(n)], [w >> s for s in range(n)]
    imgs = [_rnd(1 + s, (B, 3, hs[s], ws[s]), uniform=True) for s in range(n)]
    warped = [torch.cat(((imgs[s] + _rnd(4 + s, (B, 3, hs[s], ws[s]), 0.1)).clamp(0, 1), (imgs[s] + _rnd(7 + s, (B, 3, hs[s], ws[s]), 0.1)).clamp(0, 1)))
              for s in range(n)]
    for s in range(n):
        warped[s][:B, :, 1:4, 2:7] = 0.0
    flows = [_rnd(10 + s, (2 * B, 2, hs[s], ws[s]), 3.0 / (1 << s)) for s in range(n)]
    return imgs, warped, flows


def _oracle_pack(imgs, warped, flows, B):
    """The scale loop of the reference (model_flow_paper.py:224-235) on (from_l | from_r) / (bwd | fwd) stacked operands."""
    lp = ls = lsm = lc = 0
    for img, wp, fl in zip(imgs, warped, flows):
        from_l, from_r, f_bwd, f_fwd = wp[:B], wp[B:], fl[:B], fl[B:]
        d_l, d_r, w_b, w_f, _, _ = R.diff_weight(img, from_l, from_r)
        lp = lp + R.masked_l1(d_r, w_f) + R.masked_l1(d_l, w_b)
        ls = ls + R.ssim_loss(img, from_r, w_f) + R.ssim_loss(img, from_l, w_b)
        lsm = lsm + R.grad2_error(f_fwd / 20.0, img) + R.grad2_error(f_bwd / 20.0, img)
        lc = lc + R.consis_loss(f_fwd, f_bwd, w_f)
    return [lp, ls, lsm, lc]


def test_multiscale_losses_wiring_against_the_oracle():
    from unopticalflow_amd import ops
    B = 2
    imgs, warped0, flows0 = _scene(B, 24, 40)
    gl = [_rnd(20 + k, (B,)) for k in range(4)]
    wp_ref = [t.clone().requires_grad_() for t in warped0]
    fl_ref = [t.clone().requires_grad_() for t in flows0]
    ref = _oracle_pack(imgs, wp_ref, fl_ref, B)
    sum((p * g).sum() for p, g in zip(ref, gl)).backward()
    for deferred, stacked in ((True, False), (False, False), (True, True)):
        wp = [t.clone().requires_grad_() for t in warped0]
        fl = [t.clone().requires_grad_() for t in flows0]
        halves = [f.split(B) for f in fl]
        fb, ff = [x[0] for x in halves], [x[1] for x in halves]
        with patched(ops) as emu:
            with (ops.deferred_loss_sums if deferred else __import__('contextlib').nullcontext()):
                # stacked: the consistency term finds its halves in flows_lr by offset (what Model_flow.forward uses: no split nodes in the graph)
                pixel, ssim, smooth, consis = ops.multiscale_losses(imgs, wp, fl) if stacked else ops.multiscale_losses(imgs, wp, fl, ff, fb)
                assert [tuple(t.shape) for t in pixel + ssim + smooth] == [(2 * B,)] * 9 and [tuple(t.shape) for t in consis] == [(B,)] * 3
                packed = ops.loss_combine(pixel, ssim, smooth, consis)
            fwd_calls = list(emu.calls)
            sum((p * g).sum() for p, g in zip(packed, gl)).backward()
            bwd_calls = emu.calls[len(fwd_calls):]
        ms5 = ['unflow_occ_weight_fwd_ms', 'unflow_masked_mean_fwd_ms', 'unflow_ssim_loss_fwd_ms', 'unflow_smooth2_fwd_ms', 'unflow_consis_fwd_ms']
        if deferred:                                                     # 5 first stages, ONE second stage, the bookkeeping launch
            assert fwd_calls == ms5 + ['unflow_loss_finalize_batch', 'unflow_loss_combine_fwd'], fwd_calls
        else:                                                            # (outside the block every `_ms` forward finishes its own sums)
            assert [c for c in fwd_calls if c != 'unflow_loss_finalize_batch'] == ms5 + ['unflow_loss_combine_fwd'] and fwd_calls.count('unflow_loss_finalize_batch') == 4
        assert sorted(bwd_calls) == sorted(['unflow_loss_combine_bwd', 'unflow_absdiff_bwd_ms', 'unflow_masked_mean_bwd_ms', 'unflow_ssim_loss_bwd_ms',
                                            'unflow_smooth2_bwd_ms', 'unflow_consis_bwd_ms']), bwd_calls
        assert not ops.deferred_loss_sums.jobs and not ops.deferred_loss_sums.enabled
        for a, b, name in zip(packed, ref, ('pixel', 'ssim', 'smooth', 'consis')):
            np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=2e-5, atol=1e-7, err_msg=name)
        for s in range(3):
            np.testing.assert_allclose(wp[s].grad.numpy(), wp_ref[s].grad.numpy(), rtol=1e-4, atol=1e-7 * float(wp_ref[s].grad.abs().max()) + 1e-12, err_msg='warped %d' % s)
            np.testing.assert_allclose(fl[s].grad.numpy(), fl_ref[s].grad.numpy(), rtol=1e-4, atol=1e-6 * float(fl_ref[s].grad.abs().max()) + 1e-12, err_msg='flow %d' % s)


def test_multiscale_losses_with_an_unused_scale():
    """A caller that drops one scale's terms gets None gradients there and a launch over the remaining scales (the `live` subsets)."""
    from unopticalflow_amd import ops
    B = 2
    imgs, warped0, flows0 = _scene(B, 16, 24)
    wp = [t.clone().requires_grad_() for t in warped0]
    fl = [t.clone().requires_grad_() for t in flows0]
    halves = [f.split(B) for f in fl]
    with patched(ops):
        pixel, ssim, smooth, consis = ops.multiscale_losses(imgs, wp, fl, [x[1] for x in halves], [x[0] for x in halves])
        (pixel[0].sum() + ssim[0].sum() + smooth[0].sum() + consis[0].sum() + pixel[2].sum()).backward()
    assert wp[0].grad is not None and wp[2].grad is not None and fl[0].grad is not None
    assert wp[1].grad is None and fl[1].grad is None and fl[2].grad is None


def test_masked_image_warp_pyramid_wiring():
    from unopticalflow_amd import ops
    B = 2
    imgs, _, flows0 = _scene(B, 20, 36)
    flows0 = [f[:B] * 2.0 for f in flows0]
    gout = [_rnd(40 + s, tuple(i.shape)) for s, i in enumerate(imgs)]
    for ac in (False, True):
        fr = [f.clone().requires_grad_() for f in flows0]
        ref = [R.warp_flow(i, f, True, ac) for i, f in zip(imgs, fr)]
        sum((o * g).sum() for o, g in zip(ref, gout)).backward()
        fl = [f.clone().requires_grad_() for f in flows0]
        with patched(ops) as emu:
            res = ops._WarpMaskedMS.apply(3, ac, *imgs, *fl)
            outs, masks = res[:3], res[3:]
            sum((o * g).sum() for o, g in zip(outs, gout)).backward()
            assert emu.calls == ['unflow_warp_fwd_ms', 'unflow_warp_bwd_ms']
            again = ops.warp_flow_masked_pyramid(imgs, flows0, ac)
        for s in range(3):
            assert torch.equal(outs[s], ref[s]) and torch.equal(again[s], ref[s])
            assert masks[s].dtype == torch.uint8 and torch.equal(masks[s], R.warp_mask(imgs[s].shape, flows0[s], ac))
            assert torch.equal(fl[s].grad, fr[s].grad)
    # the reference's error for a flow of another shape (net_utils.py:35-36) survives the multi-scale form
    import pytest
    with patched(ops):
        with pytest.raises(ValueError):
            ops.warp_flow_masked_pyramid(imgs, [flows0[0], flows0[1], flows0[1]])


def test_model_loss_section_wiring(monkeypatch):
    """Model_flow.forward from the flows on -- image pyramids, masked warps, the scale loop, loss_combine -- through the emulator, with and
    without `multiscale_losses`, against the reference's own sequence (model_flow_paper.py:214-235) evaluated by the oracle on the same
    flows: the (left | right) / (bwd | fwd) stacking, the per-scale operands and the association of the sums are what is being checked."""
    from unopticalflow_amd import get_model, ops
    B, H, W = 2, 32, 48
    cfg = R.default_cfg()
    inputs = R.synthetic_triplets(B, H, W, seed=3)
    # flows as the decoder would hand them over: [2B,2,h,w] = (centre->left | centre->right) at scales 0..3
    flows0 = [_rnd(60 + s, (2 * B, 2, H >> s, W >> s), 2.0 / (1 << s)) for s in range(4)]
    imgl, img, imgr = inputs[:, :, :H], inputs[:, :, H:2 * H], inputs[:, :, 2 * H:]
    fr = [f.clone().requires_grad_() for f in flows0]
    pyr_l, pyr_c, pyr_r = R.img_pyramid(imgl, 3), R.img_pyramid(img, 3), R.img_pyramid(imgr, 3)
    warped = [torch.cat((R.warp_flow(pl, f[:B], True), R.warp_flow(pr, f[B:], True))) for pl, pr, f in zip(pyr_l, pyr_r, fr)]
    ref = _oracle_pack(pyr_c, warped, fr[:3], B)
    gl = [_rnd(70 + k, (B,)) for k in range(4)]
    sum((p * g).sum() for p, g in zip(ref, gl)).backward()
    monkeypatch.setattr(ops, 'multiscale_supported', lambda imgs, warped: 0 < len(imgs) <= 4 and all(t.shape[-1] % 2 == 0 for t in imgs))
    for ms in (False, True):
        model = get_model('flow')(cfg)
        model.multiscale_losses = ms
        fl = [f.clone().requires_grad_() for f in flows0]
        monkeypatch.setattr(model, '_flows', lambda *a, **k: fl)
        with patched(ops) as emu:
            pack = model(inputs)
            sum((pack[k] * g).sum() for k, g in zip(('loss_pixel', 'loss_ssim', 'loss_flow_smooth', 'loss_flow_consis'), gl)).backward()
        assert ('unflow_warp_fwd_ms' in emu.calls) == ms and ('unflow_ssim_loss_fwd_ms' in emu.calls) == ms and ('unflow_ssim_loss_fwd' in emu.calls) != ms
        for k, r in zip(('loss_pixel', 'loss_ssim', 'loss_flow_smooth', 'loss_flow_consis'), ref):
            assert pack[k].shape == (B,)
            np.testing.assert_allclose(pack[k].detach().numpy(), r.detach().numpy(), rtol=2e-5, atol=1e-7, err_msg='%s (multiscale_losses=%s)' % (k, ms))
        for s in range(3):
            np.testing.assert_allclose(fl[s].grad.numpy(), fr[s].grad.numpy(), rtol=2e-4, atol=2e-6 * float(fr[s].grad.abs().max()), err_msg='flow gradient, scale %d' % s)
        assert fl[3].grad is None                                           # the reference builds a fourth scale and never uses it (num_scales = 3)


def test_pyramid_handoff_as_two_tensors():
    """ops.to_nchw_split: the (left | right), (centre | centre) decoder inputs straight out of the hand-off kernel, and both gradients
    back through the fold kernel from where they are -- against `cat` + `split` in plain torch, values and gradient, with either half
    unused, and with no duplicated tail."""
    from unopticalflow_amd import ops
    Bp = 2                                                                  # pairs: the batch is (left | right | centre) = 3 * Bp samples
    for (C, H, W, head, dup) in ((5, 4, 6, 2 * Bp, Bp), (3, 2, 3, 2, 0), (4, 3, 5, 0, 3), (2, 2, 2, 6, 0)):
        B = 3 * Bp
        x0 = _rnd(80 + C, (B, C, H, W)).contiguous(memory_format=torch.channels_last)
        ga, gb = _rnd(81, (head, C, H, W)), _rnd(82, (B - head + dup, C, H, W))
        for use in ('both', 'head', 'tail'):
            xr = x0.clone().requires_grad_()
            full = torch.cat((xr, xr[B - dup:]), 0) if dup else xr
            ra, rb = full.contiguous().split((head, B - head + dup))
            x = x0.clone(memory_format=torch.channels_last).requires_grad_()
            with patched(ops) as emu:
                a, b = ops.to_nchw_split(x, head, dup)
                assert a.is_contiguous() and b.is_contiguous() and a.shape == ra.shape and b.shape == rb.shape
                assert torch.equal(a, ra) and torch.equal(b, rb)
                if (use == 'head' and head == 0) or (use == 'tail' and B - head + dup == 0):
                    continue
                loss = lambda p, q: ((p * ga).sum() if use != 'tail' else 0) + ((q * gb).sum() if use != 'head' else 0)
                loss(a, b).backward()
                loss(ra, rb).backward()
                assert emu.calls[0] == 'unflow_to_nchw_dup' and set(emu.calls[1:]) <= {'unflow_to_nhwc_fold'} and len(emu.calls) <= 3
            assert x.grad.is_contiguous(memory_format=torch.channels_last) or x.grad.numel() == 0
            np.testing.assert_allclose(x.grad.numpy(), xr.grad.numpy(), rtol=0, atol=1e-6, err_msg=str((C, H, W, head, dup, use)))


def test_model_loss_section_against_the_reference_fixture(monkeypatch, golden):
    """The same path against g5_loss_section.npz -- the REFERENCE's own run of model_flow_paper.py:227-251 from given flows, on frames with
    saturated / dark flat patches: Model_flow.forward's Python (both launch forms) on the reference's inputs gives the reference's four
    losses and flow gradients (the arithmetic behind the C entries is the oracle's here; tests -m gpu run the same comparison on the kernels)."""
    from unopticalflow_amd import get_model, ops
    g = golden('g5_loss_section.npz')
    T = torch.from_numpy
    imgl, img, imgr = T(g['imgl']), T(g['img']), T(g['imgr'])
    B = img.shape[0]
    inputs = torch.cat((imgl, img, imgr), 2)                                  # [B,3,3H,W]: left, centre, right stacked on H
    monkeypatch.setattr(ops, 'multiscale_supported', lambda imgs, warped: 0 < len(imgs) <= 4 and all(t.shape[-1] % 2 == 0 for t in imgs))
    for ms in (False, True):
        model = get_model('flow')(R.default_cfg())
        model.multiscale_losses = ms
        fl = [torch.cat((T(g['flow_b%d' % s]), T(g['flow_f%d' % s]))).requires_grad_() for s in range(4)]      # (centre->left | centre->right)
        monkeypatch.setattr(model, '_flows', lambda *a, **k: fl)
        with patched(ops):
            pack = model(inputs)
            sum((pack[k] * T(g['gl%d' % i])).sum() for i, k in enumerate(('loss_pixel', 'loss_ssim', 'loss_flow_smooth', 'loss_flow_consis'))).backward()
        for k in ('loss_pixel', 'loss_ssim', 'loss_flow_smooth', 'loss_flow_consis'):
            np.testing.assert_allclose(pack[k].detach().numpy(), g[k], rtol=1e-5, err_msg='%s (multiscale_losses=%s)' % (k, ms))
        for s in range(3):
            ref = np.concatenate((g['g_flow_b%d' % s], g['g_flow_f%d' % s]))
            np.testing.assert_allclose(fl[s].grad.numpy(), ref, rtol=1e-4, atol=2e-6 * float(np.abs(ref).max()), err_msg='flow gradient, scale %d' % s)
        assert fl[3].grad is None
