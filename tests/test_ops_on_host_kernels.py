"""The product's operators -- unopticalflow_amd/ops.py as it is -- on CPU tensors over the HOST-EXECUTED kernel library (tests/hostexec.py:
the kernel source files compiled for the build host, lanes as fibers): Python wrapper, C entry and kernel source together, against the
oracle and the reference's fixtures, without a GPU.  The `-m gpu` tests make the same comparisons on the device build."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
import hostexec

T = torch.from_numpy


def rnd(seed, shape, scale=1.0, uniform=False):
    rng = np.random.default_rng(seed)
    a = rng.random(shape, dtype=np.float32) if uniform else rng.standard_normal(shape).astype(np.float32)
    return T(a * np.float32(scale))


def close(a, b, rtol=1e-4, atol=1e-6, what=''):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


@pytest.fixture(scope='module')
def ops():
    from unopticalflow_amd import ops as _ops
    return _ops


def test_multiscale_losses_are_the_same_bits_and_the_oracles_values(ops):
    """tests/test_zz_round5_gpu.py::test_multiscale_losses_are_the_same_bits, here: every loss of the scale loop as one launch over the scales
    (ops.multiscale_losses, C ABI 11) against the scale-by-scale operators -- bit for bit in losses and every gradient, inside and outside
    deferred_loss_sums, halves by offset or as tensors -- and both against the oracle."""
    B, n = 2, 3
    h, w = 24, 72
    hs, ws = [h >> s for s in range(n)], [w >> s for s in range(n)]
    imgs = [rnd(71 + s, (B, 3, hs[s], ws[s]), uniform=True) for s in range(n)]
    warped0 = [torch.cat(((imgs[s] + rnd(74 + s, (B, 3, hs[s], ws[s]), 0.1)).clamp(0, 1), (imgs[s] + rnd(77 + s, (B, 3, hs[s], ws[s]), 0.1)).clamp(0, 1))) for s in range(n)]
    for s in range(n):
        warped0[s][:B, :, 1:5, 2:9] = 0.0
    flows0 = [rnd(80 + s, (2 * B, 2, hs[s], ws[s]), 3.0 / (1 << s)) for s in range(n)]
    gl = [rnd(90 + k, (B,)) for k in range(4)]
    res, moved, launches = {}, {}, {}
    call = ops._call

    def counting(name, *a, nbytes=0, shape=None):                  # the algorithmic bytes an entry declares (bench.py's roofline.losses sums them)
        moved[form] = moved.get(form, 0) + nbytes
        launches[form] = launches.get(form, 0) + 1
        return call(name, *a, nbytes=nbytes, shape=shape)
    with hostexec.patched(ops), pytest.MonkeyPatch.context() as mp:
        mp.setattr(ops, '_call', counting)
        for form in ('per scale', 'one launch', 'one launch, halves by offset', 'one launch, sums at once'):
            wp = [t.clone().requires_grad_() for t in warped0]
            fl = [t.clone().requires_grad_() for t in flows0]
            halves = [f.split(B) for f in fl]
            fb, ff = [x[0] for x in halves], [x[1] for x in halves]
            with (__import__('contextlib').nullcontext() if form.endswith('at once') else ops.deferred_loss_sums):
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
    for form in list(res)[1:]:
        for k, (a, b) in enumerate(zip(res['per scale'], res[form])):
            assert torch.equal(a, b), (form, k, float((a - b).abs().max()))
        # one launch over the scales declares the bytes of the launches it replaces: the loss section's roofline fraction moves with its TIME only
        assert moved[form] == moved['per scale'], (form, moved)
    assert launches['one launch'] < launches['per scale'] and launches['one launch'] <= 14, launches
    # ... and the oracle (model_flow_paper.py:224-235 on the stacked operands)
    wp = [t.clone().requires_grad_() for t in warped0]
    fl = [t.clone().requires_grad_() for t in flows0]
    lp = ls = lsm = lc = 0
    for img, w_, f_ in zip(imgs, wp, fl):
        d_l, d_r, w_b, w_f, _, _ = R.diff_weight(img, w_[:B], w_[B:])
        lp = lp + R.masked_l1(d_r, w_f) + R.masked_l1(d_l, w_b)
        ls = ls + R.ssim_loss(img, w_[B:], w_f) + R.ssim_loss(img, w_[:B], w_b)
        lsm = lsm + R.grad2_error(f_[B:] / 20.0, img) + R.grad2_error(f_[:B] / 20.0, img)
        lc = lc + R.consis_loss(f_[B:], f_[:B], w_f)
    sum((p * g).sum() for p, g in zip((lp, ls, lsm, lc), gl)).backward()
    got = res['one launch, halves by offset']
    for k, (a, b) in enumerate(zip(got[:4], (lp, ls, lsm, lc))):
        close(a, b, rtol=1e-4, what='loss %d' % k)
    for s in range(n):
        close(got[4 + s], wp[s].grad, rtol=1e-4, atol=2e-5 * float(wp[s].grad.abs().max()), what='warped gradient %d' % s)
        close(got[4 + n + s], fl[s].grad, rtol=1e-4, atol=2e-5 * float(fl[s].grad.abs().max()), what='flow gradient %d' % s)


@pytest.mark.parametrize('ac,cl,switches', [(0, False, {}), (1, True, {}), (0, True, {'multiscale_losses': True, 'split_handoff': True})] +
                         ([(1, False, {'multiscale_losses': True})] if __import__('os').environ.get('UNFLOW_HOST_CHECK_SANITIZE') == 'all' else []))
def test_module_128_golden_on_host_kernels(golden, ops, ac, cl, switches):
    """tests/test_hip_model.py::test_module_128_golden on the CPU tier: the product's Model_flow over the host-executed kernel sources (convolutions:
    torch's CPU conv2d) against g2_module_128.npz, the REFERENCE's own run of BASELINE config 1 -- features, flows at four scales, the inference
    flow, the image warps' validity masks, the four losses, the total, the gradient norm, every gradient tensor's L1 norm, three full gradient
    tensors, and three Adam steps -- at the GPU test's bars, for both grid_sample conventions, both memory formats of the conv stacks, and with
    the round-5 switches that have not run on a GPU yet (one launch per loss over the scales, the hand-off as two tensors)."""
    from unopticalflow_amd import get_model, generate_loss_weights_dict
    g = golden('g2_module_128.npz')
    tag = '_ac%d' % ac
    cfg = R.default_cfg(align_corners=bool(ac), channels_last=cl)
    model = get_model('flow')(cfg)
    model.load_state_dict(R.seeded_state_dict(model, 1234, float(g['flow_gain'])))
    for k, v in switches.items():
        setattr(model, k, v)
    weights = generate_loss_weights_dict(cfg)
    B, H, W = int(g['B']), int(g['H']), int(g['W'])
    x = R.synthetic_triplets(B, H, W, seed=0, structured=True)
    imgl, img, imgr = x[:, :, :H], x[:, :, H:2 * H], x[:, :, 2 * H:]
    with hostexec.patched(ops):
        with torch.no_grad():
            feats = model.fpyramid(img)
            close(feats[4], g['feat5' + tag], rtol=1e-4, atol=1e-5); close(feats[5], g['feat6' + tag], rtol=1e-4, atol=1e-5)
            stacked = model._flows(imgl, img, imgr)
            fb, ff = [f[:B] for f in stacked], [f[B:] for f in stacked]
            for s in range(4):
                st = 1 if s >= 1 else 8
                scale = np.abs(g['flow_fwd%d%s' % (s, tag)]).max()
                close(ff[s][:, :, ::st, ::st], g['flow_fwd%d%s' % (s, tag)], rtol=1e-4, atol=1e-4 * scale)
                close(fb[s][:, :, ::st, ::st], g['flow_bwd%d%s' % (s, tag)], rtol=1e-4, atol=1e-4 * scale)
            inf = model.inference_flow(img, imgr)
            close(inf[:, :, ::8, ::8], g['inference_flow' + tag], rtol=1e-4, atol=1e-4 * np.abs(g['inference_flow' + tag]).max())
            pyr_r = model.generate_img_pyramid(imgr, 4)
            for s in range(3):
                _, m = ops.warp_flow_masked(pyr_r[s], ff[s], align_corners=bool(ac))
                ref_bits = np.unpackbits(g['mask_fwd%d%s' % (s, tag)])[: m.numel()].reshape(m.shape)
                assert (m.numpy() != ref_bits).mean() <= 1e-3
        opt = torch.optim.Adam([{'params': [p for p in model.parameters() if p.requires_grad], 'lr': cfg.lr}])
        for it in range(3):
            opt.zero_grad()
            pack = model(x)
            loss = sum(weights[k] * pack[k].mean() for k in pack)
            loss.backward()
            if it == 0:
                for k in pack:
                    close(pack[k], g[k + tag], rtol=1e-4, what=k)
                close(loss, g['total' + tag], rtol=1e-4)
                gn = float(np.sqrt(sum((p.grad.double() ** 2).sum().item() for p in model.parameters())))
                np.testing.assert_allclose(gn, float(g['grad_norm' + tag]), rtol=5e-4)
                ga = np.array([p.grad.double().abs().sum().item() for p in model.parameters()])
                gmax = np.array([p.grad.abs().max().item() for p in model.parameters()])
                ref_ga = g['grad_abs' + tag]
                bad = np.abs(ga - ref_ga) > 2e-3 * ref_ga + 2e-3 * gmax
                assert not bad.any(), [(n, a, b) for (n, _), a, b, z in zip(model.named_parameters(), ga, ref_ga, bad) if z]
                if ac == 0:
                    named = dict(model.named_parameters())
                    for name, tol in (('fpyramid.conv1.0.weight', 1e-3), ('pwc_model.conv2_0.0.weight', 2e-3), ('pwc_model.dc_conv7.weight', 1e-3)):
                        ref_g = g['gradfull_' + name + tag]
                        close(named[name].grad, ref_g, rtol=0, atol=tol * np.abs(ref_g).max(), what='grad ' + name)
            opt.step()
            np.testing.assert_allclose(loss.item(), g['loss_step%d%s' % (it, tag)], rtol=1e-4 if it == 0 else 2e-3)
            if it in (0, 2):
                pa = np.array([p.detach().double().abs().sum().item() for p in model.parameters()])
                np.testing.assert_allclose(pa, g['param_abs_step%d%s' % (it + 1, tag)], rtol=5e-4)


def test_loss_section_fixture_kernels_vs_reference_vs_float64(golden, ops, monkeypatch):
    """g5_loss_section.npz (the reference's own run of model_flow_paper.py:227-251 on frames with saturated / dark flat patches) through the
    product's Model_flow over the host-executed kernels, both launch forms: the four losses at 1e-4 rel of the reference -- and the flow
    gradients three ways.  These frames are ill-conditioned on purpose (SSIM's variances cancel on flat patches): the reference's own fp32
    gradient is 5.7e-4 / 9e-5 / 1.6e-5 of the largest element away from a float64 evaluation at scales 0 / 1 / 2; the kernels (sum-space
    SSIM, fma-contracted pair factors) are 2.0e-4 / 3.8e-5 / 3.9e-5 away.  Bar: the kernels may not be further from the float64 truth than
    twice the reference's own distance (or 1e-4 of the largest element)."""
    from unopticalflow_amd import get_model
    g = golden('g5_loss_section.npz')
    inputs = torch.cat((T(g['imgl']), T(g['img']), T(g['imgr'])), 2)
    keys = ('loss_pixel', 'loss_ssim', 'loss_flow_smooth', 'loss_flow_consis')

    def oracle(dtype):
        imgl, img, imgr = (T(g[k]).to(dtype) for k in ('imgl', 'img', 'imgr'))
        fb = [T(g['flow_b%d' % s]).to(dtype).requires_grad_() for s in range(4)]
        ff = [T(g['flow_f%d' % s]).to(dtype).requires_grad_() for s in range(4)]
        pl, pc, pr = R.img_pyramid(imgl, 4), R.img_pyramid(img, 4), R.img_pyramid(imgr, 4)
        lp = ls = lsm = lc = 0
        for s in range(3):
            from_l, from_r = R.warp_flow(pl[s], fb[s], True), R.warp_flow(pr[s], ff[s], True)
            d_l, d_r, w_b, w_f, _, _ = R.diff_weight(pc[s], from_l, from_r)
            lp = lp + R.masked_l1(d_r, w_f) + R.masked_l1(d_l, w_b)
            ls = ls + R.ssim_loss(pc[s], from_r, w_f) + R.ssim_loss(pc[s], from_l, w_b)
            lsm = lsm + R.grad2_error(ff[s] / 20.0, pc[s]) + R.grad2_error(fb[s] / 20.0, pc[s])
            lc = lc + R.consis_loss(ff[s], fb[s], w_f)
        sum((l * T(g['gl%d' % k]).to(dtype)).sum() for k, l in enumerate((lp, ls, lsm, lc))).backward()
        return [torch.cat((fb[s].grad, ff[s].grad)).double().numpy() for s in range(3)]
    truth = oracle(torch.float64)
    for ms in (False, True):
        model = get_model('flow')(R.default_cfg())
        model.multiscale_losses = ms
        fl = [torch.cat((T(g['flow_b%d' % s]), T(g['flow_f%d' % s]))).requires_grad_() for s in range(4)]
        monkeypatch.setattr(model, '_flows', lambda *a, **k: fl)
        with hostexec.patched(ops):
            pack = model(inputs)
            sum((pack[k] * T(g['gl%d' % i])).sum() for i, k in enumerate(keys)).backward()
        for k in keys:
            close(pack[k], g[k], rtol=1e-4, what='%s (multiscale_losses=%s)' % (k, ms))
        for s in range(3):
            ref = np.concatenate((g['g_flow_b%d' % s], g['g_flow_f%d' % s])).astype(np.float64)
            big = float(np.abs(truth[s]).max())
            ref_err = float(np.abs(ref - truth[s]).max())
            err = float(np.abs(fl[s].grad.double().numpy() - truth[s]).max())
            assert err <= max(2.0 * ref_err, 1e-4 * big), (ms, s, err / big, ref_err / big)
        assert fl[3].grad is None


def test_kitti_256x832_golden_on_host_kernels(golden, ops):
    """tests/test_hip_model.py::test_kitti_256x832_golden on the CPU tier: the headline resolution (832x256, B = 1) against the reference's fixture
    g3_kitti_256x832.npz over the host-executed kernel sources, channels_last conv stacks with one launch per loss over the scales: inference
    flow, the four losses, the total, the gradient norm, the L1 norm of each of the 98 gradients, the parameters after one Adam step."""
    from unopticalflow_amd import get_model, generate_loss_weights_dict
    g = golden('g3_kitti_256x832.npz')
    cfg = R.default_cfg(align_corners=False, channels_last=True)
    model = get_model('flow')(cfg)
    model.load_state_dict(R.seeded_state_dict(model, 1234, float(g['flow_gain'])))
    model.multiscale_losses = True
    weights = generate_loss_weights_dict(cfg)
    x = R.synthetic_triplets(1, 256, 832, seed=0, structured=True)
    with hostexec.patched(ops):
        with torch.no_grad():
            inf = model.inference_flow(x[:, :, 256:512], x[:, :, 512:])
            ref = g['inference_flow_ac0']
            close(inf[:, :, ::8, ::8], ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
        opt = torch.optim.Adam([{'params': [p for p in model.parameters() if p.requires_grad], 'lr': cfg.lr}])
        opt.zero_grad()
        pack = model(x)
        for k in pack:
            close(pack[k], g[k + '_ac0'], rtol=1e-4, what=k)
        loss = sum(weights[k] * pack[k].mean() for k in pack)
        close(loss, g['total_ac0'], rtol=1e-4)
        loss.backward()
    gn = float(np.sqrt(sum((p.grad.double() ** 2).sum().item() for p in model.parameters())))
    np.testing.assert_allclose(gn, float(g['grad_norm_ac0']), rtol=5e-4)
    ga = np.array([p.grad.double().abs().sum().item() for p in model.parameters()])
    gmax = np.array([p.grad.abs().max().item() for p in model.parameters()])
    bad = np.abs(ga - g['grad_abs_ac0']) > 2e-3 * g['grad_abs_ac0'] + 2e-3 * gmax
    assert not bad.any(), [(n, a, b) for (n, _), a, b, z in zip(model.named_parameters(), ga, g['grad_abs_ac0'], bad) if z]
    opt.step()
    pa = np.array([p.detach().double().abs().sum().item() for p in model.parameters()])
    np.testing.assert_allclose(pa, g['param_abs_step1_ac0'], rtol=5e-4)


@pytest.mark.parametrize('fixture,scales,acs', [('g2_module_128.npz', (0, 1, 2, 3), (0, 1)), ('g3_kitti_256x832.npz', (1, 2, 3), (0,))])
def test_model_scale_masks_bit_exact_on_host_kernels(golden, ops, fixture, scales, acs):
    """tests/test_hip_model.py::test_model_scale_masks_bit_exact on the CPU tier -- the integer half of north_star's parity bar: the
    reference's OWN flows (fixture ``flowfull_*``) through the warp kernel's source, executed on the host; the uint8 validity masks equal
    the reference's bit for bit at every pyramid scale, both directions, both grid_sample conventions; no mismatch allowance."""
    g = golden(fixture)
    with hostexec.patched(ops):
        for ac in acs:
            tag = '_ac%d' % ac
            per_scale = {}
            for s in scales:
                for nm in ('fwd', 'bwd'):
                    fl = T(g['flowfull_%s%d%s' % (nm, s, tag)])
                    ones = torch.ones((fl.shape[0], 1) + tuple(fl.shape[2:]))
                    out, m = ops.warp_flow_masked(ones, fl, align_corners=bool(ac))
                    per_scale[nm, s] = (ones, fl, out)
                    ref_bits = np.unpackbits(g['mask_%s%d%s' % (nm, s, tag)])[: m.numel()].reshape(m.shape)
                    assert np.array_equal(m.numpy(), ref_bits), (fixture, nm, s, ac, int((m.numpy() != ref_bits).sum()))
                    o = out.numpy()
                    assert np.array_equal(o != 0, ref_bits != 0) and (o[ref_bits != 0] >= 0.9999).all()
            # ... and as ONE launch over the scales (ops.warp_flow_masked_pyramid, the `_ms` warp kernel): the same bytes
            for nm in ('fwd', 'bwd'):
                outs = ops.warp_flow_masked_pyramid([per_scale[nm, s][0] for s in scales], [per_scale[nm, s][1] for s in scales], align_corners=bool(ac))
                for s, o in zip(scales, outs):
                    assert torch.equal(o, per_scale[nm, s][2]), (fixture, nm, s, ac)


def test_gpu_op_tests_rehearsed_on_host_kernels():
    """The `-m gpu` operator tests' OWN code on the CPU tier: the node ids of tests/rehearsed_on_host.txt (114 of tests/test_hip_ops.py and
    tests/test_zz_round5_gpu.py: the reference's golden fixtures for cost volume, warp and losses, masks bit for bit, the oracle comparisons of
    every operator family at small shapes, bf16 epilogues, the round-5 tests that have never seen a GPU) run in a child pytest with
    UNFLOW_TESTS_ON_HOST=1 (tests/conftest.py): CPU tensors, the product's ops.py, the REAL kernel sources executed on the host -- same
    inputs, same bars.  A GPU test that rots (a renamed argument, a changed shape) or a kernel-source change that breaks parity fails HERE,
    not in the next GPU session."""
    import os
    import subprocess
    import sys
    if hostexec.library() is None:
        pytest.skip('the host-executed library needs the ROCm clang++')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ids = [l.strip() for l in open(os.path.join(root, 'tests', 'rehearsed_on_host.txt')) if l.strip() and not l.startswith('#')]
    assert len(ids) >= 100
    if os.environ.get('UNFLOW_REHEARSE_ALL') != '1' and os.environ.get('UNFLOW_HOST_CHECK_SANITIZE') != 'all':
        # every pass: at most three parametrisations per test function (first, middle, last of the file's order) -- every function of the list still
        # runs, the fourteen flows of one mask test do not; UNFLOW_REHEARSE_ALL=1: the whole list
        groups = {}
        for i in ids:
            groups.setdefault(i.split('[')[0], []).append(i)
        ids = [i for g in groups.values() for i in (g if len(g) <= 3 else [g[0], g[len(g) // 2], g[-1]])]
        assert len(ids) >= 60
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '--runxfail', '-p', 'no:cacheprovider', '-p', 'no:xdist'] + ids, cwd=root,
                       env=dict(os.environ, UNFLOW_TESTS_ON_HOST='1'), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and ('%d passed' % len(ids)) in r.stdout, r.stdout[-4000:]


def _product_rank(rank, world, port, steps, out_path, flat=False):
    """One rank of the data-parallel train step with the PRODUCT on it: unopticalflow_amd.Model_flow over the host-executed kernels, FlowTrainer as
    bench.py / train.py build it for several ranks (flat gradient buffer, pieces all-reduced from hooks during backward, bias gradients
    deferred to the end of the pass, the one-launch Adam of csrc/optim.hip) -- over gloo, since the ranks have no GPU."""
    import os
    from unopticalflow_amd import get_model, ops as _ops
    from unopticalflow_amd.parallel import init_distributed, shard_batch
    from unopticalflow_amd.trainer import FlowTrainer
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    init_distributed('gloo')
    with hostexec.patched(_ops):
        cfg = R.default_cfg()
        model = get_model('flow')(cfg)
        model.load_state_dict(R.seeded_state_dict(model, 1234 if rank == 0 else 999, 0.25))       # rank 0's weights must win
        trainer = FlowTrainer(cfg, model, distributed=True, use_graph=flat)
        assert type(trainer.optimizer).__name__ == 'FlowAdam' and trainer._defer_bias_grads and trainer.grads.overlap == (not flat)
        x = R.synthetic_triplets(2 * world, 64, 128, seed=5, structured=True)
        for _ in range(steps):
            if not flat:
                loss, _ = trainer.step(shard_batch(x, rank, world))
                continue
            # the replayed multi-rank step's exchange without its graphs (they need a GPU): backward assigns, pack_all copies every piece into
            # the flat buffer (what graph A ends with), ONE all-reduce of the buffer, Adam on the buffer's views (graph B)
            trainer.grads.zero()
            trainer._backward(trainer.total_loss(trainer.model(shard_batch(x, rank, world))))
            assert trainer.grads.launched_early == 0
            trainer.grads.pack_all()
            trainer.grads.all_reduce_flat()
            trainer.optimizer.step()
        assert trainer.optimizer.native_steps == steps                                            # the kernel stepped, not torch's Adam
        trainer.grads.check_views()
        if rank == 0:
            torch.save({'grad': trainer.grads.vector(), 'params': [p.detach().clone() for p in model.parameters()]}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,steps,flat', [(2, 1, False), (8, 1, True)] + ([(2, 2, False)] if __import__('os').environ.get('UNFLOW_HOST_CHECK_SANITIZE') == 'all' else []))
def test_ranks_of_the_product_step_on_host_kernels(tmp_path, world, steps, flat):
    """SURVEY 8(e) with the product's own model on the ranks: two gloo ranks (and eight, with the one-all-reduce exchange of the replayed step; a two-step case with UNFLOW_HOST_CHECK_SANITIZE=all), each running Model_flow over the host-executed kernel sources on its
    half of the batch, exchange the flat gradient piece by piece during backward and step the one-launch Adam; rank 0's averaged gradient
    and its parameters after two steps equal the ORACLE's single-process steps on the whole batch (train.py:137-152 with DataParallel's
    batch split, train.py:36-37).  tests/test_data_parallel.py makes the same comparison with the oracle's model on the ranks."""
    import socket
    import torch.multiprocessing as mp
    from unopticalflow_amd.trainer import FlowTrainer
    if hostexec.library() is None:
        pytest.skip('the host-executed library needs the ROCm clang++')
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    out = str(tmp_path / 'rank0.pt')
    mp.spawn(_product_rank, args=(world, port, steps, out, flat), nprocs=world, join=True)
    got = torch.load(out)
    cfg = R.default_cfg()
    ref = R.Model_flow(cfg)
    ref.load_state_dict(R.seeded_state_dict(ref, 1234, 0.25))
    trainer = FlowTrainer(cfg, ref, distributed=False, fused_adam=False)
    x = R.synthetic_triplets(2 * world, 64, 128, seed=5, structured=True)
    for _ in range(steps):
        trainer.step(x)
    g_ref = trainer.grads.vector()
    pa, pb = torch.cat([a.reshape(-1) for a in got['params']]), torch.cat([b.detach().reshape(-1) for b in ref.parameters()])
    if steps == 1:
        # one step: the averaged gradient element by element (the GPU suite's bar for a whole-model gradient), the parameters inside Adam's step
        np.testing.assert_allclose(got['grad'].numpy(), g_ref.numpy(), rtol=1e-3, atol=2e-5 * g_ref.abs().max().item())
        # Adam's first update is lr * sign(g) whatever |g|: an element whose gradient is zero to rounding may step the other way (2 lr apart)
        off = (pa - pb).abs()
        assert float(off.max()) <= 2.1e-4 and float((off > 2e-5).float().mean()) < 1e-4, (float(off.max()), float((off > 2e-5).float().mean()))
    else:
        # two steps: the second starts from parameters 2e-4 apart in those few elements -- the comparison is by norm and by share, as in test_hip_model.py
        rel = float((got['grad'] - g_ref).norm() / g_ref.norm())
        assert rel < 2e-3, rel
        off = (pa - pb).abs()
        assert float(off.max()) <= 4.2e-4 and float((off > 2e-5).float().mean()) < 2e-3, (float(off.max()), float((off > 2e-5).float().mean()))


def test_smoke_entry_rehearsed_on_host_kernels():
    """__graft_entry__.smoke() -- the driver's round-end check: one train step of the product on `cuda:0` against the oracle -- with ITS OWN code on
    CPU tensors (tests/rehearsal.py) over the host-executed kernels, in a process of its own: the losses inside 1e-4, the gradient norm inside
    2e-4 of the oracle's, as on the device."""
    import os
    import subprocess
    import sys
    if hostexec.library() is None:
        pytest.skip('the host-executed library needs the ROCm clang++')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import rehearsal, hostexec; rehearsal.install()\n"
            "from unopticalflow_amd import ops\n"
            "import __graft_entry__ as g\n"
            "with hostexec.patched(ops):\n"
            "    g.smoke()\n") % (root, os.path.join(root, 'tests'))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'smoke ok' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_line_rehearsed_on_host_kernels():
    """bench.py's OWN code (tests/bench_on_host.py: CPU tensors, host-executed kernels, made-up timer values) with the switches that have not been
    measured yet: the step loop, the kernel-timer bookkeeping and the JSON assembly run through and the LAST stdout line is the contract's
    object -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config
    {workload} / roofline {bound, achieved, peak, unit, frac, traffic} (+ aggregate, losses) / cpu_baseline -- with the one-launch-per-loss
    entries in the loss section (10 loss launches per step, 36 -> 12 counting the image warps).  No number in it means anything."""
    import json
    import os
    import subprocess
    import sys
    if hostexec.library() is None:
        pytest.skip('the host-executed library needs the ROCm clang++')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'bench_on_host.py'), '--steps', '2', '--warmup', '1', '--batch', '1', '--hw', '64', '128',
                        '--graph', '0', '--no-cpu-baseline', '--multiscale-losses', '1', '--split-handoff', '1'], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'pairs/s' and d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert d['config']['loss_launch_form'] == 'one per loss over the scales' and 'not a BASELINE configuration' in d['config']['workload']
    roof = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'aggregate', 'losses'):
        assert k in roof, k
    assert roof['bound'] == 'hbm' and roof['peak'] == 8000.0 and roof['unit'] == 'GB/s'
    entries = {e['entry'] for e in roof['losses']['per_entry']}
    assert entries == {'unflow_%s_ms' % n for n in ('occ_weight_fwd', 'masked_mean_fwd', 'ssim_loss_fwd', 'smooth2_fwd', 'consis_fwd', 'consis_bwd', 'smooth2_bwd',
                                                     'ssim_loss_bwd', 'masked_mean_bwd', 'absdiff_bwd')}, entries
    assert roof['losses']['launches_per_step'] == 10.0
    warps = {e['entry'] for e in roof['aggregate']['per_level']}
    assert {'unflow_warp_fwd_ms', 'unflow_warp_bwd_ms'} <= warps
