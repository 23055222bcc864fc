"""Input stage: decoded stacked triplet -> [3,3H,W] float, kitti_prepared.py:63-90,145-148.

cv2 is not installed here (parity of the resize is unpinned, see oracle/prepare_cpu.py); the CPU tests
check the restatement's structural properties, the GPU tests check the HIP kernel bit-for-bit against it.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.prepare_cpu import cv2_resize_linear_u8, prepare_triplet

KITTI_SIZES = [(375, 1242), (370, 1224), (374, 1238), (376, 1241)]


def _float_bilinear(img, W, H):
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    return F.interpolate(t, size=(H, W), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()


@pytest.mark.parametrize('h,w,H,W', [(375, 1242, 256, 832), (436, 1024, 448, 1024), (90, 300, 256, 832), (31, 57, 64, 64)])
def test_resize_within_one_grey_level_of_float_bilinear(h, w, H, W):
    img = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    out = cv2_resize_linear_u8(img, W, H)
    assert out.shape == (H, W, 3) and out.dtype == np.uint8
    assert np.abs(out.astype(np.float64) - _float_bilinear(img, W, H)).max() < 1.0


def test_resize_identities():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (128, 192, 3), dtype=np.uint8)
    assert (cv2_resize_linear_u8(img, 192, 128) == img).all()                       # same size: copy
    a = img.astype(np.int64)
    box = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2   # OpenCV's 2x fast-area path
    assert (cv2_resize_linear_u8(img, 96, 64) == box).all()
    flat = np.full((50, 70, 3), 201, np.uint8)                                       # constants survive the fixed point
    assert (cv2_resize_linear_u8(flat, 832, 256) == 201).all()


def test_prepare_triplet_layout_and_flip():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (3 * 100 + 2, 160, 3), dtype=np.uint8)                # 2 leftover rows are ignored
    out = prepare_triplet(img, (64, 128), False)
    assert out.shape == (3, 192, 128) and out.dtype == np.float32
    for k in range(3):
        frame = cv2_resize_linear_u8(img[k * 100:(k + 1) * 100], 128, 64)
        np.testing.assert_array_equal(out[:, k * 64:(k + 1) * 64], (frame / 255.0).transpose(2, 0, 1).astype(np.float32))
    np.testing.assert_array_equal(prepare_triplet(img, (64, 128), True), out[:, :, ::-1])


def test_prepare_triplet_matches_the_references_own_pipeline(golden):
    """g7_prepare.npz: the REFERENCE's KITTI_Prepared.preprocess_img and the tail of its __getitem__ (kitti_prepared.py:63-99,146-154, imported
    unmodified; cv2.resize in its identity case, cv2.flip as the mirror it documents) on a decoded stacked triplet at its native size, with and
    without the flip.  The restatement gives the same float32 bytes: frame split, flip, / 255.0 in float64 then .float(), HWC -> CHW."""
    g = golden('g7_prepare.npz')
    hw = tuple(int(v) for v in g['img_hw'])
    for flip in (0, 1):
        np.testing.assert_array_equal(prepare_triplet(g['img'], hw, bool(flip)), g['out_flip%d' % flip])


def test_product_host_resize_is_the_restated_opencv_arithmetic():
    """unopticalflow_amd.data.resize_linear_u8 (the host-side twin of the kernel's resize: `--host_input 1` training, the KITTI evaluation loaders)
    against the oracle's restatement of OpenCV's 8-bit INTER_LINEAR, bit for bit: KITTI's native sizes down to 256x832, up-scaling, odd sizes,
    the exact-half reduction, the identity."""
    from unopticalflow_amd.data import resize_linear_u8
    rng = np.random.default_rng(17)
    for (h, w), (H, W) in (((375, 1242), (256, 832)), ((370, 1224), (256, 832)), ((436, 1024), (448, 1024)), ((90, 300), (256, 832)),
                           ((31, 57), (64, 64)), ((128, 192), (64, 96)), ((24, 40), (24, 40)), ((5, 7), (3, 2))):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        np.testing.assert_array_equal(resize_linear_u8(img, W, H), cv2_resize_linear_u8(img, W, H), err_msg='%dx%d -> %dx%d' % (h, w, H, W))
    with pytest.raises(ValueError):
        resize_linear_u8(np.zeros((4, 4, 3), np.float32), 2, 2)


def test_host_input_path_and_eval_loader_match_the_reference(tmp_path, golden):
    """The all-CPU input path (`data.PreparedTriplets`, train.py --host_input 1) and the evaluation loader (`evaluation.KITTI_2012`, test.py --task
    kitti_flow) read PNG files and must hand the network what the reference's loaders do.  At native size against the reference's own output
    (g7_prepare.npz: KITTI_Prepared / KITTI_2012 `__getitem__`, imported unmodified, cv2.imread serving the decoded frames), bit for bit; at
    another size against the oracle (OpenCV's resize restated) -- the reference resizes the uint8 frame BEFORE the division by 255."""
    from unopticalflow_amd.data import PreparedTriplets
    from unopticalflow_amd.evaluation import KITTI_2012, write_png
    g = golden('g7_prepare.npz')
    hw = tuple(int(v) for v in g['img_hw'])
    os.makedirs(tmp_path / 'seq')
    write_png(str(tmp_path / 'seq' / '0.png'), np.ascontiguousarray(g['img'][:, :, ::-1]))        # the file holds RGB, cv2.imread hands out BGR
    (tmp_path / 'train.txt').write_text('seq/0.png seq/0_cam.txt\n')
    ds = PreparedTriplets(str(tmp_path), img_hw=hw)
    seen = set()
    for seed in range(64):
        np.random.seed(seed)
        flip = int(np.random.rand() > 0.5)
        if flip in seen:
            continue
        seen.add(flip)
        np.random.seed(seed)
        np.testing.assert_array_equal(ds[0].numpy(), g['out_flip%d' % flip])
    assert seen == {0, 1}
    ds2 = PreparedTriplets(str(tmp_path), img_hw=(16, 36))                                        # another size: the restated resize
    np.random.seed(3)
    flip = bool(np.random.rand() > 0.5)
    np.random.seed(3)
    np.testing.assert_array_equal(ds2[0].numpy(), prepare_triplet(g['img'], (16, 36), flip))
    os.makedirs(tmp_path / 'image_2')
    write_png(str(tmp_path / 'image_2' / '000000_10.png'), np.ascontiguousarray(g['eval_img1'][:, :, ::-1]))
    write_png(str(tmp_path / 'image_2' / '000000_11.png'), np.ascontiguousarray(g['eval_img2'][:, :, ::-1]))
    np.testing.assert_array_equal(KITTI_2012(str(tmp_path), hw, 1)[0].numpy(), g['eval_out'])
    got = KITTI_2012(str(tmp_path), (16, 36), 1)[0].numpy()
    want = np.concatenate([cv2_resize_linear_u8(g['eval_img1'], 36, 16), cv2_resize_linear_u8(g['eval_img2'], 36, 16)], 0) / 255.0
    np.testing.assert_array_equal(got, want.transpose(2, 0, 1).astype(np.float32))


# ----------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@gpu
def test_prepare_triplets_kernel_matches_the_references_own_pipeline(golden):
    """The HIP input-stage kernel against the reference fixture g7_prepare.npz (see the CPU test above), bit for bit."""
    from unopticalflow_amd import ops
    g = golden('g7_prepare.npz')
    hw = tuple(int(v) for v in g['img_hw'])
    out = ops.prepare_triplets([g['img'], g['img']], hw, [False, True], 'cuda:0', src_is_rgb=False)
    torch.cuda.synchronize()
    for flip in (0, 1):
        np.testing.assert_array_equal(out[flip].cpu().numpy(), g['out_flip%d' % flip])


@gpu
@pytest.mark.parametrize('img_hw', [(256, 832), (64, 128)])
def test_prepare_triplets_kernel_bit_exact(img_hw):
    from unopticalflow_amd import ops
    rng = np.random.default_rng(11)
    sizes = KITTI_SIZES + [(img_hw[0], img_hw[1]), (2 * img_hw[0], 2 * img_hw[1]), (40, 50), (img_hw[0] + 1, 3 * img_hw[1] + 7)]
    images = [rng.integers(0, 256, (3 * h + (i % 3), w, 3), dtype=np.uint8) for i, (h, w) in enumerate(sizes)]
    flips = [bool(i & 1) for i in range(len(images))]
    out = ops.prepare_triplets(images, img_hw, flips, 'cuda:0', src_is_rgb=False)
    torch.cuda.synchronize()
    assert out.shape == (len(images), 3, 3 * img_hw[0], img_hw[1])
    for i, im in enumerate(images):
        np.testing.assert_array_equal(out[i].cpu().numpy(), prepare_triplet(im, img_hw, flips[i]), err_msg='image %d' % i)
    rgb = ops.prepare_triplets(images[:2], img_hw, None, 'cuda:0', src_is_rgb=True)   # RGB-decoded PNG -> BGR planes
    np.testing.assert_array_equal(rgb[1].cpu().numpy(), prepare_triplet(images[1][:, :, ::-1], img_hw, False))


@gpu
def test_prepare_triplets_rejects_bad_input():
    from unopticalflow_amd import ops
    with pytest.raises(ValueError):
        ops.prepare_triplets([np.zeros((30, 40, 3), np.float32)], (64, 128), None, 'cuda:0')
    with pytest.raises(ValueError):
        ops.prepare_triplets([np.zeros((30, 40, 3), np.uint8)], (64, 130), None, 'cuda:0')
    with pytest.raises(ValueError):
        ops.prepare_triplets([np.zeros((30, 40, 3), np.uint8)], (64, 128), None, 'cpu')


@gpu
def test_device_loader_matches_oracle(tmp_path):
    """train.txt + stacked PNGs -> DeviceTripletLoader batches == oracle on the decoded files (same flip draws)."""
    from unopticalflow_amd.data import DecodedTriplets, DeviceTripletLoader
    from unopticalflow_amd.evaluation import read_png, write_png
    rng = np.random.default_rng(2)
    names = []
    for i, (h, w) in enumerate(KITTI_SIZES + [(100, 320)]):
        os.makedirs(tmp_path / 'seq', exist_ok=True)
        write_png(str(tmp_path / 'seq' / ('%d.png' % i)), rng.integers(0, 256, (3 * h, w, 3), dtype=np.uint8))
        names.append('seq/%d.png seq/%d_cam.txt' % (i, i))
    (tmp_path / 'train.txt').write_text('\n'.join(names) + '\n')
    ds = DecodedTriplets(str(tmp_path), img_hw=(64, 128))
    loader = DeviceTripletLoader(ds, 2, 'cuda:0', (64, 128), num_workers=0, shuffle=False)
    np.random.seed(123)
    got = [b.cpu().numpy() for b in loader]
    assert [g.shape[0] for g in got] == [2, 2, 1]
    np.random.seed(123)
    for i in range(5):
        img = read_png(str(tmp_path / 'seq' / ('%d.png' % i)))
        flip = bool(np.random.rand() > 0.5)
        want = prepare_triplet(np.ascontiguousarray(img[:, :, ::-1]), (64, 128), flip)   # file order RGB -> BGR
        np.testing.assert_array_equal(got[i // 2][i % 2], want)
