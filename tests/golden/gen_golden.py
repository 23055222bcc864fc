"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container (needs /root/reference); nothing here is
imported by the tests or by the product.  The reference is imported unmodified
with the two harness shims of SURVEY.md section 8c:
  * an empty ``cv2`` module (model_flow_paper.py:10 imports it, never calls it),
  * ``Tensor.get_device`` returning the device for CPU tensors (net_utils.py:48).
``align_corners=True`` fixtures (torch-1.2 semantics of ``grid_sample``) are made
by wrapping ``nn.functional.grid_sample`` around the reference's call sites
(net_utils.py:46,49).

Parameters and inputs come from numpy PCG64 streams (``oracle.ref_cpu.
seeded_state_dict`` / ``synthetic_triplets``) so the fixtures hold only small
inputs and the reference's outputs.

    python tests/golden/gen_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

sys.modules.setdefault('cv2', types.ModuleType('cv2'))
_gd = torch.Tensor.get_device
torch.Tensor.get_device = lambda t: t.device if not t.is_cuda else _gd(t)
sys.path.insert(0, '/root/reference')
from core.networks import get_model                      # noqa: E402  (the reference)
from core.networks.structures import warp_flow, PWC_tf   # noqa: E402
from core.networks.pytorch_ssim import SSIM              # noqa: E402

from oracle import ref_cpu as R                          # noqa: E402  (seeding helpers only)

_GS = nn.functional.grid_sample
FLOW_GAIN = 0.25


class align_corners_ctx:
    """Make the reference's bare grid_sample calls use the given align_corners."""

    def __init__(self, ac):
        self.ac = ac

    def __enter__(self):
        ac = self.ac
        nn.functional.grid_sample = lambda x, g, **k: _GS(x, g, align_corners=ac, **k)

    def __exit__(self, *a):
        nn.functional.grid_sample = _GS


def rnd(seed, shape, scale=1.0, uniform=False):
    rng = np.random.default_rng(seed)
    a = rng.random(shape, dtype=np.float32) if uniform else rng.standard_normal(shape).astype(np.float32)
    return torch.from_numpy(a * np.float32(scale))


def npy(t):
    return t.detach().cpu().numpy()


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print('%-22s %8.1f KB  %d arrays' % (name, os.path.getsize(path) / 1024, len(d)))


# ------------------------------------------------------------------ G1: corr
def gen_corr():
    out = {}
    pwc = PWC_tf()
    cases = [(2, 5, 7, 11), (4, 32, 9, 13), (8, 196, 5, 7), (4, 196, 4, 13), (4, 7, 16, 20), (1, 3, 3, 5)]
    out['cases'] = np.array(cases, np.int64)
    for k, (d, C, h, w) in enumerate(cases):
        f1 = rnd(100 + k, (2, C, h, w)).requires_grad_()
        f2 = rnd(200 + k, (2, C, h, w)).requires_grad_()
        cv = pwc.corr_naive(f1, f2, d=d)
        g = rnd(300 + k, tuple(cv.shape))
        cv.backward(g)
        out.update({'f1_%d' % k: npy(f1), 'f2_%d' % k: npy(f2), 'g_%d' % k: npy(g), 'cv_%d' % k: npy(cv),
                    'gf1_%d' % k: npy(f1.grad), 'gf2_%d' % k: npy(f2.grad)})
    save('g1_corr.npz', out)


# ------------------------------------------------------------------ G6: corr at shapes the matrix-core backward serves
def gen_corr_served():
    """g1_corr's cases are too small to reach the matrix-core backward (csrc/corr_mfma.h serves >= 8192 pixels, C % 16 == 0,
    W % 4 == 0, H >= 4 d), so nothing from the REFERENCE pinned it.  The reference's own corr_naive + autograd (pwc_tf.py:97-106) at three served
    shapes: d = 8 on a 64 x 128 map, d = 8 with a ragged last 16-pixel segment (116 = 7 x 16 + 4) and a row count that is no multiple of the
    kernel's chunk, d = 4 (the form on request).  Inputs and the upstream gradient come from the seeds stored here (tests re-draw them with the
    same numpy PCG64 streams); the file holds only the reference's two gradients per case and a strided sample of its cost volume."""
    out = {}
    pwc = PWC_tf()
    cases = [(8, 1, 16, 64, 128), (8, 1, 16, 72, 116), (4, 1, 16, 64, 128)]
    out['cases'] = np.array(cases, np.int64)
    out['seeds'] = np.array([[1100 + k, 1200 + k, 1300 + k] for k in range(len(cases))], np.int64)
    for k, (d, B, C, h, w) in enumerate(cases):
        f1 = rnd(1100 + k, (B, C, h, w)).requires_grad_()
        f2 = rnd(1200 + k, (B, C, h, w)).requires_grad_()
        cv = pwc.corr_naive(f1, f2, d=d)
        g = rnd(1300 + k, tuple(cv.shape))
        cv.backward(g)
        out.update({'cv_s_%d' % k: npy(cv)[:, :, ::8, ::8].copy(), 'gf1_%d' % k: npy(f1.grad), 'gf2_%d' % k: npy(f2.grad)})
    save('g6_corr_served.npz', out)


# ------------------------------------------------------------------ G1: warp
def gen_warp():
    out = {}
    cases = [  # C, h, w, flow scale, use_mask
        (3, 16, 24, 2.0, 1), (3, 16, 24, 9.0, 1), (32, 9, 13, 3.0, 0), (5, 8, 8, 6.0, 0), (3, 1, 1, 0.3, 1),
        (3, 32, 104, 4.0, 1)]
    out['cases'] = np.array([(c, h, w, int(s * 10), m) for c, h, w, s, m in cases], np.int64)
    for k, (C, h, w, s, um) in enumerate(cases):
        x0 = rnd(400 + k, (2, C, h, w), uniform=True)
        fl0 = rnd(500 + k, (2, 2, h, w), s)
        if k == 0:
            fl0[:, :, :4] = 0.0                      # exact zero flow rows
            fl0[:, :, 4:6] = torch.round(fl0[:, :, 4:6])   # integer flows
        g = rnd(600 + k, (2, C, h, w))
        out['x_%d' % k], out['flow_%d' % k], out['g_%d' % k] = npy(x0), npy(fl0), npy(g)
        for ac in (0, 1):
            with align_corners_ctx(bool(ac)):
                x = x0.clone().requires_grad_()
                fl = fl0.clone().requires_grad_()
                y = warp_flow(x, fl, use_mask=bool(um))
                y.backward(g)
                tag = '%d_ac%d' % (k, ac)
                out['y_' + tag], out['gx_' + tag], out['gflow_' + tag] = npy(y), npy(x.grad), npy(fl.grad)
                if um:   # the mask itself: a ones image comes back non-zero exactly where mask==1
                    m = warp_flow(torch.ones(2, 1, h, w), fl0, use_mask=True)
                    out['mask_' + tag] = (npy(m) != 0).astype(np.uint8)
    save('g1_warp.npz', out)


# ------------------------------------------------------------------ G1: losses
def gen_losses():
    out = {}
    cfg = R.default_cfg(num_scales=1)
    m = get_model('flow')(cfg)
    B, h, w = 2, 24, 40
    img = rnd(700, (B, 3, h, w), uniform=True)
    from_l = (img + rnd(701, (B, 3, h, w), 0.1)).clamp(0, 1)
    from_r = (img + rnd(702, (B, 3, h, w), 0.1)).clamp(0, 1)
    from_l[:, :, 3:9, 5:17] = 0.0                   # masked-out (all-zero) pixels
    from_r[:, :, 10:20, 22:38] = 0.0
    from_r[0, 0, 0, 0] = 0.0                        # single-channel zero: still valid
    flow_f = rnd(703, (B, 2, h, w), 3.0)
    flow_b = rnd(704, (B, 2, h, w), 3.0)
    out.update(img=npy(img), from_l=npy(from_l), from_r=npy(from_r), flow_f=npy(flow_f), flow_b=npy(flow_b))
    gl = rnd(705, (B,))                             # upstream d(total)/d(loss[b])
    out['gl'] = npy(gl)

    fl = from_l.clone().requires_grad_()
    fr = from_r.clone().requires_grad_()
    d_b, d_f, w_b, w_f = m.compute_diff_weight([fl], [img], [fr])
    out.update(diff_l=npy(d_b[0]), diff_r=npy(d_f[0]), w_bwd=npy(w_b[0]), w_fwd=npy(w_f[0]))
    lp = m.compute_loss_with_mask(d_f, w_f) + m.compute_loss_with_mask(d_b, w_b)
    ls_f = m.compute_loss_ssim([img], [fr], w_f)
    ls_b = m.compute_loss_ssim([img], [fl], w_b)
    out.update(loss_pixel=npy(lp), loss_ssim_f=npy(ls_f), loss_ssim_b=npy(ls_b))
    (lp * gl).sum().backward(retain_graph=True)
    out.update(lp_g_from_l=npy(fl.grad), lp_g_from_r=npy(fr.grad))
    fl.grad = None; fr.grad = None
    ((ls_f + ls_b) * gl).sum().backward()
    out.update(ls_g_from_l=npy(fl.grad), ls_g_from_r=npy(fr.grad))
    w3 = w_f[0].repeat(1, 3, 1, 1)
    out['ssim_map'] = npy(SSIM(img * w3, from_r * w3))

    ff = flow_f.clone().requires_grad_()
    lsm = m.compute_loss_flow_smooth([ff], [img])
    (lsm * gl).sum().backward()
    out.update(loss_smooth=npy(lsm), lsm_g_flow=npy(ff.grad))

    ff = flow_f.clone().requires_grad_()
    fb = flow_b.clone().requires_grad_()
    lc = m.compute_loss_flow_consis([ff], [fb], [w_f[0].detach()])
    (lc * gl).sum().backward()
    assert fb.grad is None
    out.update(loss_consis=npy(lc), lc_g_flow=npy(ff.grad))
    save('g1_losses.npz', out)


# ------------------------------------------------------------------ G5: the loss section of forward() over three scales
def gen_loss_section():
    """model_flow_paper.py:227-251 as the reference runs it -- image pyramids, warp_flow_pyramid with masks, compute_diff_weight and the
    four losses summed over num_scales = 3 -- from GIVEN flows (4 scales, as PWC_tf returns them), on a triplet whose frames carry the
    regions where SSIM's window sums cancel hardest: saturated flat patches (1.0 against 1.0, 1 - 1/255, 0.95), dark flat patches, a
    step edge, plus a black occluder.  Outputs: the four [B] losses, the masked warped images, the gradients w.r.t. all flows."""
    cfg = R.default_cfg()
    m = get_model('flow')(cfg)
    B, H, W = 2, 64, 96
    rng = np.random.default_rng(900)
    base = rng.random((B, 3, H, W), dtype=np.float32)
    frames = [np.clip(base + 0.05 * rng.standard_normal((B, 3, H, W)).astype(np.float32), 0, 1) for _ in range(3)]      # left, centre, right
    hq, wq = H // 4, W // 4
    for k, dy in enumerate((0.0, 1.0 / 255.0, 0.05)):
        frames[1][:, :, :hq, k * wq:(k + 1) * wq] = 1.0
        frames[0][:, :, :hq, k * wq:(k + 1) * wq] = np.float32(1.0 - dy)
        frames[2][:, :, :hq, k * wq:(k + 1) * wq] = np.float32(1.0 - dy)
    for f in frames:
        f[:, :, hq:2 * hq, :wq] = 0.0                                     # dark, equal in all three (also: invalid where warped to 0)
    frames[1][:, :, hq:2 * hq, wq:2 * wq] = 2.0 / 255.0
    frames[0][:, :, hq:2 * hq, wq:2 * wq] = 0.0
    frames[1][:, :, 2 * hq:, 2 * wq:] = 0.25; frames[1][:, :, 2 * hq:, 3 * wq:] = 0.9      # a step edge in the centre frame only
    imgl, img, imgr = (torch.from_numpy(f) for f in frames)
    flows_b = [rnd(910 + s, (B, 2, H >> s, W >> s), 2.0 / (1 << s)).requires_grad_() for s in range(4)]       # centre -> left
    flows_f = [rnd(920 + s, (B, 2, H >> s, W >> s), 2.0 / (1 << s)).requires_grad_() for s in range(4)]       # centre -> right
    n = len(flows_f)
    pl, pc, pr = m.generate_img_pyramid(imgl, n), m.generate_img_pyramid(img, n), m.generate_img_pyramid(imgr, n)
    from_l = m.warp_flow_pyramid(pl, flows_b)
    from_r = m.warp_flow_pyramid(pr, flows_f)
    d_b, d_f, w_b, w_f = m.compute_diff_weight(from_l, pc, from_r)
    loss_pixel = m.compute_loss_with_mask(d_f, w_f) + m.compute_loss_with_mask(d_b, w_b)
    loss_ssim = m.compute_loss_ssim(pc, from_r, w_f) + m.compute_loss_ssim(pc, from_l, w_b)
    loss_smooth = m.compute_loss_flow_smooth(flows_f, pc) + m.compute_loss_flow_smooth(flows_b, pc)
    loss_consis = m.compute_loss_flow_consis(flows_f, flows_b, w_f)
    gl = [rnd(930 + k, (B,)) for k in range(4)]
    sum((l * g).sum() for l, g in zip((loss_pixel, loss_ssim, loss_smooth, loss_consis), gl)).backward()
    out = {'imgl': npy(imgl), 'img': npy(img), 'imgr': npy(imgr),
           'loss_pixel': npy(loss_pixel), 'loss_ssim': npy(loss_ssim), 'loss_flow_smooth': npy(loss_smooth), 'loss_flow_consis': npy(loss_consis)}
    for k in range(4):
        out['gl%d' % k] = npy(gl[k])
    for s in range(4):
        out['flow_b%d' % s], out['flow_f%d' % s] = npy(flows_b[s]), npy(flows_f[s])
    for s in range(3):
        out['from_l%d' % s], out['from_r%d' % s] = npy(from_l[s]), npy(from_r[s])
        out['w_bwd%d' % s], out['w_fwd%d' % s] = npy(w_b[s]), npy(w_f[s])
        out['g_flow_b%d' % s], out['g_flow_f%d' % s] = npy(flows_b[s].grad), npy(flows_f[s].grad)
    assert flows_b[3].grad is None and flows_f[3].grad is None          # the fourth scale is built and never used (num_scales = 3)
    save('g5_loss_section.npz', out)


# ------------------------------------------------------------------ G2 / G3: modules
def grad_stats(model):
    names, s, a = [], [], []
    for n, p in model.named_parameters():
        names.append(n); s.append(p.grad.double().sum().item()); a.append(p.grad.double().abs().sum().item())
    return names, np.array(s), np.array(a)


def param_stats(model):
    return np.array([p.detach().double().sum().item() for p in model.parameters()]), \
        np.array([p.detach().double().abs().sum().item() for p in model.parameters()])


FULL_GRADS = ('fpyramid.conv1.0.weight', 'pwc_model.conv2_0.0.weight', 'pwc_model.dc_conv7.weight')


def gen_module(name, B, H, W, steps, full_flows, mask_flow_scales=(), full_grads=False):
    out = {'B': B, 'H': H, 'W': W, 'flow_gain': FLOW_GAIN}
    cfg = R.default_cfg()
    weights = R.generate_loss_weights_dict(cfg)
    x = R.synthetic_triplets(B, H, W, seed=0, structured=True)
    for ac in (0, 1):
        tag = '_ac%d' % ac
        with align_corners_ctx(bool(ac)):
            model = get_model('flow')(cfg)
            model.load_state_dict(R.seeded_state_dict(model, 1234, FLOW_GAIN))
            opt = torch.optim.Adam([{'params': filter(lambda p: p.requires_grad, model.parameters()),
                                     'lr': cfg.lr}])
            # forward pieces (model_flow_paper.py:208-235), for feature / flow / mask fixtures
            with torch.no_grad():
                imgl, img, imgr = x[:, :, :H], x[:, :, H:2 * H], x[:, :, 2 * H:]
                feats = model.fpyramid(img)
                fb = model.pwc_model(feats, model.fpyramid(imgl), [H, W])
                ff = model.pwc_model(feats, model.fpyramid(imgr), [H, W])
                out['feat_sum' + tag] = np.array([f.double().sum().item() for f in feats])
                out['feat_abs' + tag] = np.array([f.double().abs().sum().item() for f in feats])
                out['feat5' + tag], out['feat6' + tag] = npy(feats[4]), npy(feats[5])
                for s in range(4):
                    for nm, fl in (('fwd', ff[s]), ('bwd', fb[s])):
                        st = 1 if (full_flows and s >= 1) else max(1, 8 >> s)
                        out['flow_%s%d%s' % (nm, s, tag)] = npy(fl[:, :, ::st, ::st])
                        out['flow_%s%d_sum%s' % (nm, s, tag)] = fl.double().sum().item()
                        if s in mask_flow_scales and (ac == 0 or full_flows):
                            # the exact flow the mask fixtures below were warped with: lets the GPU test demand
                            # bit-equal masks at model scale (no conv rounding in between)
                            out['flowfull_%s%d%s' % (nm, s, tag)] = npy(fl)
                pyr_l, pyr_r = model.generate_img_pyramid(imgl, 4), model.generate_img_pyramid(imgr, 4)
                for s in range(4):
                    for nm, pyr, fl in (('bwd', pyr_l, fb), ('fwd', pyr_r, ff)):
                        m = warp_flow(torch.ones_like(pyr[s][:, :1]), fl[s], use_mask=True)
                        out['mask_%s%d%s' % (nm, s, tag)] = np.packbits((npy(m) != 0).astype(np.uint8))
                out['inference_flow' + tag] = npy(model.inference_flow(img, imgr)[:, :, ::8, ::8])
            # train steps (train.py:139-152)
            for it in range(steps):
                opt.zero_grad()
                pack = model(x)
                loss = sum(weights[k] * pack[k].mean() for k in pack)
                loss.backward()
                if it == 0:
                    for k in pack:
                        out[k + tag] = npy(pack[k])
                    out['total' + tag] = loss.item()
                    names, gs, ga = grad_stats(model)
                    out['grad_sum' + tag], out['grad_abs' + tag] = gs, ga
                    out['grad_norm' + tag] = float(np.sqrt(sum((p.grad.double() ** 2).sum().item()
                                                               for p in model.parameters())))
                    if full_grads and ac == 0:
                        for n, p in model.named_parameters():
                            if n in FULL_GRADS:
                                out['gradfull_' + n + tag] = npy(p.grad)
                opt.step()
                if it in (0, 2):
                    ps, pa = param_stats(model)
                    out['param_sum_step%d%s' % (it + 1, tag)] = ps
                    out['param_abs_step%d%s' % (it + 1, tag)] = pa
                out['loss_step%d%s' % (it, tag)] = loss.item()
    save(name, out)


# ------------------------------------------------------------------ G4: KITTI flow metrics (row N1)
def eval_inputs(seed=11, n=3, H=37, W=61):
    """Synthetic KITTI-style ground truth at NATIVE resolution (prediction size == ground-truth size == cfg.img_hw, so the
    reference's cv2.resize is the identity): flows with validity, non-occluded masks (a subset of valid), moving-object
    masks, predictions = ground truth + noise with a band of gross outliers."""
    rng = np.random.default_rng(seed)
    gt, noc, pred, move = [], [], [], []
    for i in range(n):
        f = rng.standard_normal((H, W, 2)) * 12.0
        f[: H // 3] *= 0.05                                     # small-magnitude region: the 5 % relative test decides
        valid = (rng.random((H, W)) > 0.25).astype(np.float64)
        nocm = valid * (rng.random((H, W)) > 0.3)
        if i == 1:
            nocm = valid.copy()                                 # no occluded pixel at all: the max(.., 1.0) divisor
        p = f + rng.standard_normal((H, W, 2)) * 1.5
        p[:, W // 2: W // 2 + 9] += 7.0                         # outliers (> 3 px)
        m = np.zeros((H, W)); m[5:20, 10:30] = 1.0
        gt.append(np.concatenate([f, valid[:, :, None]], 2)); noc.append(nocm); pred.append(p); move.append(m)
    return gt, noc, pred, move


def gen_eval():
    """core/evaluation/evaluate_flow.py imported unmodified: calculate_error_rate (:85-90) and eval_flow_avg (:93-174).
    Harness shims: a ``png`` module whose Reader serves pixel rows from an array (flowlib.py:11 imports pypng; read_flow_png, flowlib.py:107-127,
    runs over it: the arithmetic after the decode is the reference's), ``cv2.resize`` that only accepts the identity case.  The PNG byte decode
    itself and the real bilinear resize stay unpinned (no pypng / cv2 in the image)."""
    sys.modules.setdefault('png', types.ModuleType('png'))
    cv2 = sys.modules['cv2']

    def resize(img, size, interpolation=None):
        assert (img.shape[1], img.shape[0]) == tuple(size), 'fixture is native-resolution only'
        return np.copy(img)
    cv2.resize, cv2.INTER_LINEAR = resize, 1
    # (the module file itself, not the ``core.evaluation`` package: its __init__ pulls in the depth evaluation -> skimage)
    import importlib.util
    sys.path.insert(0, '/root/reference/core/evaluation')                    # evaluate_flow.py:3 does ``from flowlib import ...``
    spec = importlib.util.spec_from_file_location('evaluate_flow', '/root/reference/core/evaluation/evaluate_flow.py')
    EF = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(EF)
    gt, noc, pred, move = eval_inputs()
    H, W = gt[0].shape[:2]
    cfg = types.SimpleNamespace(img_hw=(H, W), model_dir='/nonexistent')
    out = {'gt': np.stack(gt), 'noc': np.stack(noc), 'pred': np.stack(pred), 'move': np.stack(move)}
    rates = []
    for g, n_, p, m in zip(gt, noc, pred, move):
        epe = np.sqrt(np.sum(np.square(p - g[:, :, :2]), axis=2))
        rates.append([EF.calculate_error_rate(epe, g[:, :, :2], g[:, :, 2]), EF.calculate_error_rate(epe, g[:, :, :2], g[:, :, 2] * m),
                      EF.calculate_error_rate(epe, g[:, :, :2], g[:, :, 2] * (1.0 - m)), EF.calculate_error_rate(epe, g[:, :, :2], n_)])
    out['error_rates'] = np.array(rates, np.float64)
    # flowlib.read_flow_png (flowlib.py:107-127), the reference's own function, over a stand-in for the third-party decoder it calls: pypng's
    # Reader(...).asDirect() hands it (w, h, rows of 3 w flat uint16 values, info); the stand-in serves those rows from an array instead of a
    # file, so what is pinned here is everything AFTER the PNG decode: channel order, (v - 2^15) / 64, the validity channel, zeros where invalid
    import flowlib
    rng = np.random.default_rng(23)
    raw = rng.integers(0, 65536, (29, 47, 3), dtype=np.uint16)
    raw[:, :, 2] = rng.integers(0, 2, (29, 47))                   # validity: 0 / 1 as in KITTI
    raw[3, 5] = (2 ** 15, 2 ** 15, 1); raw[4, 6] = (0, 65535, 1); raw[5, 7] = (12345, 54321, 0)

    class _Reader:
        def __init__(self, filename=None):
            pass

        def asDirect(self):
            return raw.shape[1], raw.shape[0], iter([row.reshape(-1) for row in raw]), {'size': (raw.shape[1], raw.shape[0])}
    flowlib.png.Reader = _Reader
    out['flowpng_raw'] = raw
    out['flowpng_flow'] = flowlib.read_flow_png('unused')
    out['result_plain'] = np.array(EF.eval_flow_avg(gt, noc, pred, cfg))
    out['result_moving'] = np.array(EF.eval_flow_avg(gt, noc, pred, cfg, moving_masks=move))
    # un-rounded per-metric averages, recomputed from the reference's own formulas' pieces (the strings carry 4 decimals)
    save('g4_eval.npz', out)
    print(out['result_plain']); print(out['result_moving'])


def gen_prepare():
    """core/dataset/kitti_prepared.py imported unmodified: KITTI_Prepared.preprocess_img (:91-99: resize_img :63-76, random_flip_img :78-82, / 255.0)
    and the tail of __getitem__ (:146-148,154: transpose(2, 0, 1), torch.from_numpy(img).float()) on a decoded stacked triplet at its NATIVE size.
    Harness shims for the third-party calls: ``cv2.resize`` accepts only the identity case (OpenCV's same-size resize is a copy), ``cv2.flip(img, 1)``
    is the horizontal mirror it documents.  So the frame split, the flip, the scaling, the channel / layout order and the float32 rounding are the
    reference's own; OpenCV's fixed-point bilinear at other sizes stays unpinned (no cv2 in the image; oracle/prepare_cpu.py restates it)."""
    cv2 = sys.modules['cv2']

    def resize(img, size, interpolation=None):
        assert (img.shape[1], img.shape[0]) == tuple(size), 'fixture is native-resolution only'
        return np.copy(img)
    cv2.resize, cv2.flip, cv2.INTER_LINEAR = resize, (lambda img, code: np.ascontiguousarray(img[:, ::-1]) if code == 1 else None), 1
    import importlib.util
    spec = importlib.util.spec_from_file_location('kitti_prepared', '/root/reference/core/dataset/kitti_prepared.py')
    KP = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(KP)
    ds = object.__new__(KP.KITTI_Prepared)                 # (no train.txt to read: only the image path of __getitem__ is exercised)
    h, w = 24, 40
    ds.img_hw = (h, w)
    rng = np.random.default_rng(31)
    img = rng.integers(0, 256, (3 * h, w, 3), dtype=np.uint8)
    img[0, 0] = (0, 255, 1); img[h, 1] = (254, 127, 128)
    out = {'img': img, 'img_hw': np.array([h, w])}
    seen = {}
    for seed in range(64):                                   # the reference draws the flip from numpy's global generator (:79)
        np.random.seed(seed)
        flipped = np.random.rand() > 0.5
        if flipped in seen:
            continue
        np.random.seed(seed)
        x = ds.preprocess_img(img.copy(), ds.img_hw)
        x = torch.from_numpy(x.transpose(2, 0, 1)).float()
        seen[flipped] = True
        out['out_flip%d' % int(flipped)] = x.numpy()
        if len(seen) == 2:
            break
    assert len(seen) == 2
    # the test-time loader: KITTI_2012.__getitem__ (kitti_2012.py:38-55: two frames read, stacked, preprocess_img_origin(is_test=True) --
    # kitti_prepared.py:51-61,101-109 -- transpose, .float()) with ``cv2.imread`` serving the two decoded frames
    sys.modules.setdefault('png', types.ModuleType('png'))
    spec = importlib.util.spec_from_file_location('kitti_2012', '/root/reference/core/dataset/kitti_2012.py')
    K12 = importlib.util.module_from_spec(spec)
    sys.path.insert(0, '/root/reference/core/dataset')
    spec.loader.exec_module(K12)
    a, b = rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ds12 = K12.KITTI_2012('/data', img_hw=(h, w))
    frames = {ds12.data_list[0]['img1_dir']: a, ds12.data_list[0]['img2_dir']: b}
    K12.cv2.imread = lambda path, *args: np.copy(frames[path])
    out['eval_img1'], out['eval_img2'] = a, b
    out['eval_out'] = ds12[0].numpy()
    save('g7_prepare.npz', out)


def gen_config():
    """The reference's hyper-parameter files and its loss weighting: config/kitti.yaml and config/sintel.yaml as yaml reads them (minus machine
    paths) and what core/config/config_utils.py:3-9 ``generate_loss_weights_dict`` (imported unmodified) makes of them -- a JSON file, since these
    are names and scalars."""
    import importlib.util
    import json
    import yaml
    spec = importlib.util.spec_from_file_location('config_utils', '/root/reference/core/config/config_utils.py')
    CU = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(CU)
    doc = {}
    for name in ('kitti', 'sintel'):
        cfg = yaml.safe_load(open('/root/reference/config/%s.yaml' % name))
        keep = {k: v for k, v in cfg.items() if not (isinstance(v, str) and v.startswith('/'))}
        doc[name] = {'yaml': keep, 'loss_weights': CU.generate_loss_weights_dict(types.SimpleNamespace(**cfg))}
    path = os.path.join(HERE, 'g8_config.json')
    with open(path, 'w') as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print('%-22s %8.1f KB' % ('g8_config.json', os.path.getsize(path) / 1024))


if __name__ == '__main__':
    if sys.argv[1:] == ['config']:
        gen_config()
        sys.exit(0)
    if sys.argv[1:] == ['prepare']:
        gen_prepare()
        sys.exit(0)
    if sys.argv[1:] == ['eval']:
        gen_eval()
        sys.exit(0)
    if sys.argv[1:] == ['loss_section']:
        gen_loss_section()
        sys.exit(0)
    if sys.argv[1:] == ['corr_served']:
        gen_corr_served()
        sys.exit(0)
    torch.manual_seed(0)
    gen_corr()
    gen_corr_served()
    gen_warp()
    gen_losses()
    gen_loss_section()
    gen_module('g2_module_128.npz', 2, 128, 128, steps=3, full_flows=True, mask_flow_scales=(0, 1, 2, 3), full_grads=True)
    gen_module('g3_kitti_256x832.npz', 1, 256, 832, steps=1, full_flows=False, mask_flow_scales=(1, 2, 3))
    gen_eval()
    gen_prepare()
    gen_config()
