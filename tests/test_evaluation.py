"""Evaluation path (row N1): PNG codec, KITTI flow encoding, eval_flow_avg on a synthetic KITTI-2015
layout.  CPU; the GPU-marked test at the bottom runs the whole test.py --task kitti_flow path."""
import os
import types

import numpy as np
import pytest
import torch

from unopticalflow_amd import evaluation as E


def test_png_codec_against_pil(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    a = (rng.random((37, 53, 3)) * 255).astype(np.uint8)
    a[:, :20] = (np.arange(20)[None, :, None] * 3).astype(np.uint8)          # smooth part -> Sub/Paeth/Average rows
    Image.fromarray(a).save(str(tmp_path / 'a.png'), optimize=True)
    assert np.array_equal(E.read_png(str(tmp_path / 'a.png')), a)
    g = (rng.random((19, 31)) * 65535).astype(np.uint16)
    Image.fromarray(g).save(str(tmp_path / 'g.png'))
    assert np.array_equal(E.read_png(str(tmp_path / 'g.png')), g)
    E.write_png(str(tmp_path / 'b.png'), a)
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / 'b.png'))), a)
    c = (rng.random((40, 64, 3)) * 65535).astype(np.uint16)                   # 16-bit RGB: what PIL cannot read
    E.write_png(str(tmp_path / 'c.png'), c)
    assert np.array_equal(E.read_png(str(tmp_path / 'c.png')), c)
    with pytest.raises(ValueError):
        open(str(tmp_path / 'x.png'), 'wb').write(b'not a png')
        E.read_png(str(tmp_path / 'x.png'))


def test_flow_png_encoding(tmp_path):
    """u16 = flow*64 + 2^15, valid in channel 2, invalid pixels read back as zero flow (flowlib.py:107-127)."""
    rng = np.random.default_rng(1)
    fl = rng.standard_normal((20, 30, 2)) * 10
    valid = (rng.random((20, 30)) > 0.3).astype(np.float64)
    E.write_flow_png(fl, str(tmp_path / 'f.png'), valid)
    r = E.read_flow_png(str(tmp_path / 'f.png'))
    assert np.array_equal(r[:, :, 2], valid)
    assert np.abs((r[:, :, :2] - fl) * valid[:, :, None]).max() <= 1 / 64 + 1e-12
    assert np.all(r[valid == 0][:, :2] == 0)
    raw = E.read_png(str(tmp_path / 'f.png'))
    assert raw.dtype == np.uint16 and raw[0, 0, 0] == int(np.clip(fl[0, 0, 0] * 64 + 2 ** 15, 0, 65535))


def test_flow_png_matches_the_references_own_reader(tmp_path, golden):
    """g4_eval.npz holds what the REFERENCE's flowlib.read_flow_png (flowlib.py:107-127, imported unmodified) returns for given 16-bit RGB pixel
    rows (its third-party decoder, pypng, replaced by a stand-in that serves the rows: tests/golden/gen_golden.py eval).  The same pixels written as a
    real PNG and read back by this package's reader: the float64 flow and the validity channel bit for bit -- everything after the byte decode is
    pinned on the reference; the decode itself on PIL where PIL can read the format (test_png_codec_against_pil) and on the round trip here."""
    g = golden('g4_eval.npz')
    raw = g['flowpng_raw']
    E.write_png(str(tmp_path / 'k.png'), raw)
    assert np.array_equal(E.read_png(str(tmp_path / 'k.png')), raw)
    got = E.read_flow_png(str(tmp_path / 'k.png'))
    assert got.dtype == np.float64 and np.array_equal(got, g['flowpng_flow'])


def _make_kitti(root, n, H=48, W=160, flow=(4.0, -2.0)):
    for d in ('flow_occ', 'flow_noc', 'obj_map', 'image_2'):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    rng = np.random.default_rng(2)
    for i in range(n):
        fl = np.zeros((H, W, 2)); fl[:, :, 0] = flow[0]; fl[:, :, 1] = flow[1]
        occ_valid = np.ones((H, W)); occ_valid[:4] = 0
        noc_valid = occ_valid.copy(); noc_valid[:, :16] = 0
        name = str(i).zfill(6)
        E.write_flow_png(fl, os.path.join(root, 'flow_occ', name + '_10.png'), occ_valid)
        E.write_flow_png(fl, os.path.join(root, 'flow_noc', name + '_10.png'), noc_valid)
        obj = np.zeros((H, W), np.uint8); obj[10:20, 30:60] = 3
        E.write_png(os.path.join(root, 'obj_map', name + '_10.png'), obj)
        for k in ('_10', '_11'):
            E.write_png(os.path.join(root, 'image_2', name + k + '.png'), (rng.random((H, W, 3)) * 255).astype(np.uint8))


def test_eval_flow_avg_known_errors(tmp_path):
    root = str(tmp_path)
    _make_kitti(root, 3)
    gt, noc = E.load_gt_flow_kitti(root, 'kitti_2015', 3)
    masks = E.load_gt_mask(root, 3)
    assert gt[0].shape == (48, 160, 3) and masks[0].max() == 1.0 and masks[0].sum() == 300
    cfg = types.SimpleNamespace(img_hw=(24, 80), model_dir=root)
    # prediction at half resolution: exact flow / 2 -> rescaled to exact; plus a 1 px error in u on one sample
    preds = [np.broadcast_to(np.array([2.0, -1.0]), (24, 80, 2)).copy() for _ in range(3)]
    preds[1][:, :, 0] += 0.5            # +0.5 at half res = +1 px at GT res
    res = E.eval_flow_avg(gt, noc, preds, cfg, moving_masks=masks)
    lines = res.strip().split('\n')
    assert [t.strip() for t in lines[0].split(',')][:3] == ['epe', 'epe_noc', 'epe_occ']
    vals = [float(t) for t in lines[1].split(',')]
    assert vals[0] == pytest.approx(1.0 / 3, abs=1e-4) and vals[1] == pytest.approx(1.0 / 3, abs=1e-4)
    assert vals[-1] == 0.0                                             # 1 px error is below the 3 px Fl threshold
    preds[2][:, :, 1] += 4.0                                           # 8 px error in v: every valid pixel is an outlier
    vals2 = [float(t) for t in E.eval_flow_avg(gt, noc, preds, cfg).strip().split('\n')[1].split(',')]
    assert vals2[-1] == pytest.approx(1.0 / 3, abs=1e-4) and vals2[0] == pytest.approx((1.0 + 8.0) / 3, abs=1e-3)


def test_resize_matches_half_pixel_bilinear():
    a = np.arange(12, dtype=np.float64).reshape(3, 4, 1)
    r = E.resize_bilinear(a, 8, 6)
    assert r.shape == (6, 8, 1)
    assert r[0, 0, 0] == a[0, 0, 0] and r[-1, -1, 0] == a[-1, -1, 0]          # edge clamp
    assert r[0, 1, 0] == pytest.approx(0.25)                                  # (1+0.5)/2-0.5 = 0.25


def test_kitti_dataset_reader(tmp_path):
    _make_kitti(str(tmp_path), 2)
    ds = E.KITTI_2015(str(tmp_path), (64, 128), 2)
    x = ds[1]
    assert x.shape == (3, 128, 128) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0


@pytest.mark.gpu
def test_kitti_flow_task_end_to_end(tmp_path):
    from oracle import ref_cpu as R
    from unopticalflow_amd import test as T, get_model
    root = str(tmp_path)
    _make_kitti(root, 2, H=96, W=320)
    cfg = R.default_cfg(img_hw=(64, 128), gt_2015_dir=root, model_dir=root, config_file=None)
    model = get_model('flow')(cfg).cuda().eval()
    model.load_state_dict(R.seeded_state_dict(model, 1234, 0.25))
    gt, noc = E.load_gt_flow_kitti(root, 'kitti_2015', 2)
    masks = E.load_gt_mask(root, 2)
    res = T.test_kitti_2015(cfg, model, gt, noc, masks, num=2)
    vals = [float(t) for t in res.strip().split('\n')[1].split(',')]
    assert len(vals) == 8 and all(np.isfinite(vals))
    # the same task on the CPU oracle (test.py:43-76 restated: per pair inference_flow at img_hw, then eval_flow_avg on the same PNGs,
    # the same seeded weights): the eight numbers of the result line agree
    cpu = R.get_model('flow')(cfg).eval()
    cpu.load_state_dict(R.seeded_state_dict(cpu, 1234, 0.25))
    ds = E.KITTI_2015(root, cfg.img_hw, 2)
    preds = []
    for idx in range(len(ds)):
        img = ds[idx][None]
        hh = img.shape[2] // 2
        with torch.no_grad():
            preds.append(cpu.inference_flow(img[:, :, :hh], img[:, :, hh:])[0].numpy().transpose(1, 2, 0))
    ref = [float(t) for t in E.eval_flow_avg(gt, noc, preds, cfg, moving_masks=masks).strip().split('\n')[1].split(',')]
    assert res.strip().split('\n')[0] == E.eval_flow_avg(gt, noc, preds, cfg, moving_masks=masks).strip().split('\n')[0]
    np.testing.assert_allclose(vals, ref, rtol=1e-3, atol=1e-3)
    assert max(vals[:3]) > 0.5                               # (a random-weight network is far from the ground truth: the numbers are not trivially 0)


def test_metrics_match_the_reference_fixture(golden):
    """The metric half of N1 pinned on the reference itself: ``tests/golden/g4_eval.npz`` holds what the reference's own
    ``calculate_error_rate`` (evaluate_flow.py:85-90) and ``eval_flow_avg`` (:93-174) return for native-resolution synthetic
    ground truth (prediction size == ground-truth size == cfg.img_hw, where cv2.resize is the identity; generated by
    tests/golden/gen_golden.py from /root/reference imported unmodified).  Same numbers, same result strings.  The PNG
    decode and the bilinear resize to another resolution stay unpinned (no pypng / cv2 in this image)."""
    g = golden('g4_eval.npz')
    gt, noc, pred, move = (list(g[k]) for k in ('gt', 'noc', 'pred', 'move'))
    H, W = gt[0].shape[:2]
    cfg = types.SimpleNamespace(img_hw=(H, W), model_dir='/nonexistent')
    assert np.array_equal(E.resize_bilinear(pred[0], W, H), pred[0])              # the identity case is exact here too
    for k, (gf, n_, p, m) in enumerate(zip(gt, noc, pred, move)):
        epe = np.sqrt(np.sum(np.square(p - gf[:, :, :2]), axis=2))
        got = [E.calculate_error_rate(epe, gf[:, :, :2], gf[:, :, 2]), E.calculate_error_rate(epe, gf[:, :, :2], gf[:, :, 2] * m),
               E.calculate_error_rate(epe, gf[:, :, :2], gf[:, :, 2] * (1.0 - m)), E.calculate_error_rate(epe, gf[:, :, :2], n_)]
        assert np.array_equal(np.array(got, np.float64), g['error_rates'][k])     # integer counts over integer counts: bit-equal
    assert E.eval_flow_avg(gt, noc, pred, cfg) == str(g['result_plain'])
    assert E.eval_flow_avg(gt, noc, pred, cfg, moving_masks=move) == str(g['result_moving'])
