"""The oracle (oracle/ref_cpu.py) against fixtures captured from the reference itself
(tests/golden/gen_golden.py).  CPU only.  Tolerances: the oracle calls the same ATen
ops as the reference, so op-level values agree to fp32 rounding (1e-6); masks bit-equal."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_corr_values_and_grads(golden):
    g = golden('g1_corr.npz')
    for k, (d, C, h, w) in enumerate(g['cases']):
        f1 = T(g['f1_%d' % k]).requires_grad_()
        f2 = T(g['f2_%d' % k]).requires_grad_()
        cv = R.corr_naive(f1, f2, int(d))
        assert cv.shape[1] == (2 * d + 1) ** 2
        close(cv, g['cv_%d' % k])
        cv.backward(T(g['g_%d' % k]))
        close(f1.grad, g['gf1_%d' % k])
        close(f2.grad, g['gf2_%d' % k])


def _rnd(seed, shape):
    return T(np.random.default_rng(int(seed)).standard_normal(shape).astype(np.float32))


@pytest.mark.parametrize('k', [0, 1, 2])
def test_corr_at_matrix_core_shapes(golden, k):
    """g6_corr_served.npz: the reference's corr_naive + autograd (pwc_tf.py:97-106) at shapes the matrix-core backward serves (d = 8 and 4, >= 8192
    pixels, 16 channels, a ragged last segment); inputs re-drawn from the stored seeds.  The oracle reproduces the reference to rounding."""
    g = golden('g6_corr_served.npz')
    d, B, C, h, w = (int(v) for v in g['cases'][k])
    s1, s2, s3 = g['seeds'][k]
    f1, f2 = _rnd(s1, (B, C, h, w)).requires_grad_(), _rnd(s2, (B, C, h, w)).requires_grad_()
    cv = R.corr_naive(f1, f2, d)
    close(cv[:, :, ::8, ::8], g['cv_s_%d' % k])
    cv.backward(_rnd(s3, tuple(cv.shape)))
    close(f1.grad, g['gf1_%d' % k], atol=2e-6)
    close(f2.grad, g['gf2_%d' % k], atol=2e-6)


def test_corr_shape_mismatch_asserts():
    with pytest.raises(AssertionError):          # pwc_tf.py:99
        R.corr_naive(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 5))


@pytest.mark.parametrize('ac', [0, 1])
def test_warp_values_masks_grads(golden, ac):
    g = golden('g1_warp.npz')
    for k, (C, h, w, s10, um) in enumerate(g['cases']):
        tag = '%d_ac%d' % (k, ac)
        x = T(g['x_%d' % k]).requires_grad_()
        fl = T(g['flow_%d' % k]).requires_grad_()
        y = R.warp_flow(x, fl, use_mask=bool(um), align_corners=bool(ac))
        close(y, g['y_' + tag])
        y.backward(T(g['g_%d' % k]))
        close(x.grad, g['gx_' + tag], atol=1e-5)
        close(fl.grad, g['gflow_' + tag], rtol=1e-4, atol=1e-4)
        # elementwise numpy restatement: values to rounding, mask bit-exact
        y_np, m_np = R.warp_flow_np(g['x_%d' % k], g['flow_%d' % k], bool(um), bool(ac))
        np.testing.assert_allclose(y_np, g['y_' + tag], rtol=1e-5, atol=1e-6)
        if um:
            m = R.warp_mask(x.shape, fl.detach(), bool(ac)).numpy()
            assert np.array_equal(m, g['mask_' + tag])
            assert np.array_equal(m_np, g['mask_' + tag])


def test_warp_shape_mismatch_raises():
    with pytest.raises(ValueError):              # net_utils.py:35-36
        R.warp_flow(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 9))


def test_warp_mask_np_bit_exact_random():
    """The formula the HIP kernel follows vs ATen CPU grid_sample, both conventions, on
    flows that straddle the borders (where the 0.9999 threshold decides)."""
    rng = np.random.default_rng(7)
    for ac in (False, True):
        for (h, w) in ((32, 104), (7, 5), (64, 208)):
            fl = (rng.standard_normal((2, 2, h, w)) * 6).astype(np.float32)
            fl[0, :, : h // 2] *= 1e-3              # near-integer sample positions
            x = rng.random((2, 3, h, w), dtype=np.float32)
            m_t = R.warp_mask(x.shape, T(fl), ac).numpy()
            y_np, m_np = R.warp_flow_np(x, fl, True, ac)
            assert np.array_equal(m_np, m_t)
            y_t = R.warp_flow(T(x), T(fl), True, ac).numpy()
            np.testing.assert_allclose(y_np, y_t, rtol=1e-5, atol=1e-6)


def test_losses(golden):
    g = golden('g1_losses.npz')
    img = T(g['img'])
    fl = T(g['from_l']).requires_grad_()
    fr = T(g['from_r']).requires_grad_()
    gl = T(g['gl'])
    d_l, d_r, w_b, w_f, v_b, v_f = R.diff_weight(img, fl, fr)
    close(d_l, g['diff_l']); close(d_r, g['diff_r'])
    close(w_b, g['w_bwd']); close(w_f, g['w_fwd'])
    assert np.array_equal(v_b.numpy() != 0, g['w_bwd'] != 0)      # valid == (weight != 0)
    assert np.array_equal(v_f.numpy() != 0, g['w_fwd'] != 0)
    lp = R.masked_l1(d_r, w_f) + R.masked_l1(d_l, w_b)
    close(lp, g['loss_pixel'])
    ls_f, ls_b = R.ssim_loss(img, fr, w_f), R.ssim_loss(img, fl, w_b)
    close(ls_f, g['loss_ssim_f']); close(ls_b, g['loss_ssim_b'])
    (lp * gl).sum().backward(retain_graph=True)
    close(fl.grad, g['lp_g_from_l']); close(fr.grad, g['lp_g_from_r'])
    fl.grad = None; fr.grad = None
    ((ls_f + ls_b) * gl).sum().backward()
    close(fl.grad, g['ls_g_from_l'], atol=1e-5); close(fr.grad, g['ls_g_from_r'], atol=1e-5)
    w3 = w_f.repeat(1, 3, 1, 1)
    close(R.SSIM(img * w3, fr.detach() * w3), g['ssim_map'], atol=1e-5)

    ff = T(g['flow_f']).requires_grad_()
    lsm = R.grad2_error(ff / 20.0, img)
    close(lsm, g['loss_smooth'])
    (lsm * gl).sum().backward()
    close(ff.grad, g['lsm_g_flow'])

    ff = T(g['flow_f']).requires_grad_()
    fb = T(g['flow_b']).requires_grad_()
    lc = R.consis_loss(ff, fb, w_f.detach())
    close(lc, g['loss_consis'])
    (lc * gl).sum().backward()
    close(ff.grad, g['lc_g_flow'])
    assert fb.grad is None


def _run_module(g, ac, steps):
    B, H, W = int(g['B']), int(g['H']), int(g['W'])
    cfg = R.default_cfg()
    weights = R.generate_loss_weights_dict(cfg)
    x = R.synthetic_triplets(B, H, W, seed=0, structured=True)
    model = R.Model_flow(cfg, align_corners=bool(ac))
    model.load_state_dict(R.seeded_state_dict(model, 1234, float(g['flow_gain'])))
    opt = torch.optim.Adam([{'params': [p for p in model.parameters() if p.requires_grad], 'lr': cfg.lr}])
    return cfg, weights, x, model, opt


@pytest.mark.parametrize('ac', [0, 1])
def test_module_128(golden, ac):
    g = golden('g2_module_128.npz')
    tag = '_ac%d' % ac
    cfg, weights, x, model, opt = _run_module(g, ac, 3)
    H, W = int(g['H']), int(g['W'])
    pack, aux = model(x, return_aux=True)
    for k in pack:
        close(pack[k], g[k + tag], rtol=1e-5)
    close(aux['feats'][4], g['feat5' + tag], atol=1e-5); close(aux['feats'][5], g['feat6' + tag], atol=1e-5)
    for s in range(4):
        st = 1 if s >= 1 else 8
        close(aux['flows_fwd'][s][:, :, ::st, ::st], g['flow_fwd%d%s' % (s, tag)], rtol=1e-4, atol=1e-4)
        close(aux['flows_bwd'][s][:, :, ::st, ::st], g['flow_bwd%d%s' % (s, tag)], rtol=1e-4, atol=1e-4)
        for nm, fl in (('fwd', aux['flows_fwd'][s]), ('bwd', aux['flows_bwd'][s])):
            m = R.warp_mask((x.shape[0], 1, H >> s, W >> s), fl.detach(), bool(ac)).numpy()
            assert np.array_equal(np.packbits(m), g['mask_%s%d%s' % (nm, s, tag)])
    with torch.no_grad():
        inf = model.inference_flow(x[:, :, H:2 * H], x[:, :, 2 * H:])
    close(inf[:, :, ::8, ::8], g['inference_flow' + tag], rtol=1e-4, atol=1e-4)
    # three optimisation steps, train.py:139-152
    for it in range(3):
        loss, _ = R.train_step(model, opt, x, weights)
        np.testing.assert_allclose(loss.item(), g['loss_step%d%s' % (it, tag)], rtol=2e-5)
        if it == 0:
            gs = np.array([p.grad.double().sum().item() for p in model.parameters()])
            ga = np.array([p.grad.double().abs().sum().item() for p in model.parameters()])
            np.testing.assert_allclose(ga, g['grad_abs' + tag], rtol=1e-3)
            np.testing.assert_allclose(gs, g['grad_sum' + tag], rtol=1e-3, atol=1e-4 * ga.max())
        if it in (0, 2):
            pa = np.array([p.detach().double().abs().sum().item() for p in model.parameters()])
            np.testing.assert_allclose(pa, g['param_abs_step%d%s' % (it + 1, tag)], rtol=1e-4)  # Adam normalises tiny grads


def test_module_kitti_256x832(golden):
    g = golden('g3_kitti_256x832.npz')
    cfg, weights, x, model, opt = _run_module(g, 0, 1)
    with torch.no_grad():
        pack = model(x)
    for k in pack:
        close(pack[k], g[k + '_ac0'], rtol=1e-5)
    np.testing.assert_allclose(R.total_loss(pack, weights).item(), g['total_ac0'], rtol=1e-5)


def test_input_size_must_be_multiple_of_64():
    """net_utils.py:35-36 raises when a pyramid level is not exactly half the previous one."""
    cfg = R.default_cfg()
    model = R.Model_flow(cfg)
    with pytest.raises(ValueError):
        model(torch.rand(1, 3, 300, 140))


def test_state_dict_layout():
    sd = R.Model_flow(R.default_cfg()).state_dict()
    assert len(sd) == 98
    assert tuple(sd['fpyramid.conv1.0.weight'].shape) == (16, 3, 3, 3)
    assert tuple(sd['pwc_model.conv5_0.0.weight'].shape) == (128, 211, 3, 3)
    assert tuple(sd['pwc_model.dc_conv1.0.weight'].shape) == (128, 34, 3, 3)
    assert tuple(sd['pwc_model.dc_conv7.weight'].shape) == (2, 32, 3, 3)
    assert sum(v.numel() for v in sd.values()) == 5134324


def test_loss_section_over_three_scales(golden):
    """g5_loss_section.npz: the reference's own run of model_flow_paper.py:227-251 from given flows -- image pyramids, masked warps,
    compute_diff_weight, the four losses summed over num_scales = 3 -- on frames with saturated / dark flat patches and a step edge (where
    SSIM's window sums cancel hardest).  Pins the oracle's multi-scale association, its masked warps bit for bit where the mask is
    concerned, and every flow gradient; the fourth scale is built and unused."""
    g = golden('g5_loss_section.npz')
    imgl, img, imgr = T(g['imgl']), T(g['img']), T(g['imgr'])
    B = img.shape[0]
    fb = [T(g['flow_b%d' % s]).requires_grad_() for s in range(4)]
    ff = [T(g['flow_f%d' % s]).requires_grad_() for s in range(4)]
    pl, pc, pr = R.img_pyramid(imgl, 4), R.img_pyramid(img, 4), R.img_pyramid(imgr, 4)
    lp = ls = lsm = lc = 0
    for s in range(3):
        from_l, from_r = R.warp_flow(pl[s], fb[s], True), R.warp_flow(pr[s], ff[s], True)
        close(from_l, g['from_l%d' % s], atol=1e-6); close(from_r, g['from_r%d' % s], atol=1e-6)
        d_l, d_r, w_b, w_f, _, _ = R.diff_weight(pc[s], from_l, from_r)
        close(w_b, g['w_bwd%d' % s], atol=2e-6); close(w_f, g['w_fwd%d' % s], atol=2e-6)
        assert np.array_equal(w_b.numpy() != 0, g['w_bwd%d' % s] != 0) and np.array_equal(w_f.numpy() != 0, g['w_fwd%d' % s] != 0)
        lp = lp + R.masked_l1(d_r, w_f) + R.masked_l1(d_l, w_b)
        ls = ls + R.ssim_loss(pc[s], from_r, w_f) + R.ssim_loss(pc[s], from_l, w_b)
        lsm = lsm + R.grad2_error(ff[s] / 20.0, pc[s]) + R.grad2_error(fb[s] / 20.0, pc[s])
        lc = lc + R.consis_loss(ff[s], fb[s], w_f)
    for v, k in ((lp, 'loss_pixel'), (ls, 'loss_ssim'), (lsm, 'loss_flow_smooth'), (lc, 'loss_flow_consis')):
        close(v, g[k], rtol=1e-5)
    sum((l * T(g['gl%d' % k])).sum() for k, l in enumerate((lp, ls, lsm, lc))).backward()
    for s in range(3):
        for t, k in ((fb[s], 'g_flow_b%d' % s), (ff[s], 'g_flow_f%d' % s)):
            close(t.grad, g[k], rtol=1e-4, atol=1e-6 * float(np.abs(g[k]).max()))
    assert fb[3].grad is None and ff[3].grad is None
