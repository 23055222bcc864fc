"""KITTI flow evaluation for ``test.py --task kitti_flow`` (row N1 after the hot path).

Own numpy restatement of what the reference's evaluation needs, without cv2 / pypng:
  * ``read_png`` / ``write_png``: 8- and 16-bit, gray / RGB, non-interlaced PNG (zlib + the five
    row filters) -- KITTI flow ground truth is 16-bit RGB, which PIL cannot read;
  * ``read_flow_png`` / ``write_flow_png``: KITTI encoding, u16 = flow * 64 + 2^15, third channel =
    validity (reference core/evaluation/flowlib.py:107-127,130-141);
  * ``load_gt_flow_kitti`` (evaluate_flow.py:53-83), ``load_gt_mask`` (evaluate_mask.py:195-213);
  * ``eval_flow_avg`` (evaluate_flow.py:85-174): EPE over all / non-occluded / occluded pixels and the
    Fl outlier rate (EPE > 3 px and > 5 %), optionally split by moving-object masks; same result string;
  * ``KITTI_2012`` / ``KITTI_2015`` image-pair readers (core/dataset/kitti_2012.py:12-55).
The predicted flow is rescaled and resized to the ground-truth size with half-pixel-centre bilinear
interpolation (what ``cv2.resize(..., INTER_LINEAR)`` does for float input).
"""
import os
import struct
import zlib

import numpy as np
import torch
import torch.nn.functional as F

_PNG_SIG = b'\x89PNG\r\n\x1a\n'


def read_png(path):
    """-> uint8 or uint16 array [H,W] or [H,W,C]."""
    data = open(path, 'rb').read()
    if data[:8] != _PNG_SIG:
        raise ValueError('%s is not a PNG file' % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif typ == b'IDAT':
            idat.append(body)
        elif typ == b'IEND':
            break
        pos += 12 + n
    W, H, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 4, 6):
        raise ValueError('unsupported PNG (depth %d, colour type %d, interlace %d)' % (depth, ctype, interlace))
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    bpp = ch * depth // 8
    stride = W * bpp
    raw = np.frombuffer(zlib.decompress(b''.join(idat)), np.uint8).reshape(H, stride + 1).copy()
    from . import _lib                                   # sequential byte filters: native helper (csrc/png_host.cpp),
    import ctypes                                        # from the HIP-free host library (safe in forked loader workers)
    _lib.check(_lib.load_host().unflow_png_unfilter(ctypes.c_void_p(raw.ctypes.data), H, stride, bpp), 'unflow_png_unfilter')
    out = raw[:, 1:]
    if depth == 16:
        arr = out.reshape(H, W, ch, 2)
        arr = (arr[..., 0].astype(np.uint16) << 8) | arr[..., 1]
    else:
        arr = out.reshape(H, W, ch)
    return arr[:, :, 0] if ch == 1 else arr


def write_png(path, arr):
    """uint8 / uint16 array [H,W] or [H,W,3] -> PNG (filter 0, used for fixtures and saved predictions)."""
    arr = np.asarray(arr)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    H, W, ch = arr.shape
    depth = 16 if arr.dtype == np.uint16 else 8
    ctype = {1: 0, 3: 2}[ch]
    if depth == 16:
        b = np.stack(((arr >> 8).astype(np.uint8), (arr & 255).astype(np.uint8)), -1).reshape(H, -1)
    else:
        b = arr.astype(np.uint8).reshape(H, -1)
    raw = np.concatenate((np.zeros((H, 1), np.uint8), b), 1).tobytes()

    def chunk(t, body):
        return struct.pack('>I', len(body)) + t + body + struct.pack('>I', zlib.crc32(t + body) & 0xffffffff)
    with open(path, 'wb') as f:
        f.write(_PNG_SIG + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, depth, ctype, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def read_flow_png(flow_file):
    """KITTI flow PNG -> float64 [H,W,3] (u, v, valid); flowlib.py:107-127."""
    d = read_png(flow_file).astype(np.float64)
    flow = np.zeros_like(d)
    flow[:, :, 2] = d[:, :, 2]
    invalid = d[:, :, 2] == 0
    flow[:, :, 0:2] = (d[:, :, 0:2] - 2 ** 15) / 64.0
    flow[invalid, 0] = 0
    flow[invalid, 1] = 0
    return flow


def write_flow_png(flo, flow_file, valid=None):
    """flowlib.py:130-141 (valid defaults to all ones)."""
    h, w, _ = flo.shape
    out = np.ones((h, w, 3), np.float64)
    out[:, :, 0] = np.clip(flo[:, :, 0] * 64.0 + 2 ** 15, 0, 2 ** 16 - 1)
    out[:, :, 1] = np.clip(flo[:, :, 1] * 64.0 + 2 ** 15, 0, 2 ** 16 - 1)
    if valid is not None:
        out[:, :, 2] = valid
    write_png(flow_file, out.astype(np.uint16))


def load_gt_flow_kitti(gt_dataset_dir, mode, num_gt=None):
    """-> (gt_flows [H,W,3] each, noc_masks [H,W] each); evaluate_flow.py:53-83."""
    if mode not in ('kitti_2012', 'kitti_2015'):
        raise ValueError('Mode {} not found.'.format(mode))
    if num_gt is None:
        num_gt = 194 if mode == 'kitti_2012' else 200
    gt_flows, noc_masks = [], []
    for i in range(num_gt):
        name = str(i).zfill(6) + '_10.png'
        gt_flows.append(read_flow_png(os.path.join(gt_dataset_dir, 'flow_occ', name)))
        noc_masks.append(read_flow_png(os.path.join(gt_dataset_dir, 'flow_noc', name))[:, :, 2])
    return gt_flows, noc_masks


def load_gt_mask(gt_dataset_dir, num_gt=200):
    """Moving-object masks (KITTI 2015 obj_map > 0); evaluate_mask.py:195-213."""
    masks = []
    for i in range(num_gt):
        m = read_png(os.path.join(gt_dataset_dir, 'obj_map', str(i).zfill(6) + '_10.png')).astype(np.float64)
        m[m > 0.0] = 1.0
        masks.append(m)
    return masks


def resize_bilinear(arr, W, H):
    """[h,w,C] float -> [H,W,C], half-pixel centres, edge clamp (cv2.resize INTER_LINEAR on float data)."""
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).permute(2, 0, 1)[None]
    return F.interpolate(t, size=(H, W), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()


def calculate_error_rate(epe_map, gt_flow, mask):
    """Fl: EPE > 3 px and > 5 % of the ground-truth magnitude; evaluate_flow.py:85-90."""
    bad = np.logical_and(epe_map * mask > 3,
                         epe_map * mask / np.maximum(np.sqrt(np.sum(np.square(gt_flow), axis=2)), 1e-10) > 0.05)
    return bad.sum() / mask.sum()


def eval_flow_avg(gt_flows, noc_masks, pred_flows, cfg, moving_masks=None, write_img=False):
    """evaluate_flow.py:93-174; returns the same two-line result string."""
    error = error_noc = error_occ = error_move = error_static = error_rate = 0.0
    error_move_rate = error_static_rate = 0.0
    num = len(gt_flows)
    for i, (gt_flow, noc_mask, pred_flow) in enumerate(zip(gt_flows, noc_masks, pred_flows)):
        H, W = gt_flow.shape[0:2]
        pred_flow = np.array(pred_flow, dtype=np.float64, copy=True)
        pred_flow[:, :, 0] = pred_flow[:, :, 0] / cfg.img_hw[1] * W
        pred_flow[:, :, 1] = pred_flow[:, :, 1] / cfg.img_hw[0] * H
        flo_pred = resize_bilinear(pred_flow, W, H)
        if write_img:
            os.makedirs(os.path.join(cfg.model_dir, 'pred_flow'), exist_ok=True)
            write_flow_png(flo_pred, os.path.join(cfg.model_dir, 'pred_flow', str(i).zfill(6) + '_10.png'))
        valid = gt_flow[:, :, 2]
        epe_map = np.sqrt(np.sum(np.square(flo_pred[:, :, 0:2] - gt_flow[:, :, 0:2]), axis=2))
        error += np.sum(epe_map * valid) / np.sum(valid)
        error_noc += np.sum(epe_map * noc_mask) / np.sum(noc_mask)
        error_occ += np.sum(epe_map * (valid - noc_mask)) / max(np.sum(valid - noc_mask), 1.0)
        error_rate += calculate_error_rate(epe_map, gt_flow[:, :, 0:2], valid)
        if moving_masks:
            mv = moving_masks[i]
            error_move_rate += calculate_error_rate(epe_map, gt_flow[:, :, 0:2], valid * mv)
            error_static_rate += calculate_error_rate(epe_map, gt_flow[:, :, 0:2], valid * (1.0 - mv))
            error_move += np.sum(epe_map * valid * mv) / np.sum(valid * mv)
            error_static += np.sum(epe_map * valid * (1.0 - mv)) / np.sum(valid * (1.0 - mv))
    if moving_masks:
        result = "{:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10} \n".format(
            'epe', 'epe_noc', 'epe_occ', 'epe_move', 'epe_static', 'move_err_rate', 'static_err_rate', 'err_rate')
        result += "{:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f} \n".format(
            error / num, error_noc / num, error_occ / num, error_move / num, error_static / num,
            error_move_rate / num, error_static_rate / num, error_rate / num)
        return result
    result = "{:>10}, {:>10}, {:>10}, {:>10} \n".format('epe', 'epe_noc', 'epe_occ', 'err_rate')
    result += "{:10.4f}, {:10.4f}, {:10.4f}, {:10.4f} \n".format(error / num, error_noc / num, error_occ / num, error_rate / num)
    return result


class KITTI_2012(torch.utils.data.Dataset):
    """image_2/%06d_10.png + _11.png -> float [3, 2H, W] BGR/255 at img_hw (kitti_2012.py:12-55)."""
    num_total = 194

    def __init__(self, data_dir, img_hw=(256, 832), num_total=None):
        self.data_dir, self.img_hw = data_dir, img_hw
        if num_total is not None:
            self.num_total = num_total

    def __len__(self):
        return self.num_total

    def _load(self, path):
        """One frame as ``cv2.imread`` + ``cv2.resize`` leave it (kitti_2012.py:47-52 through kitti_prepared.py:51-61): uint8 BGR, resized with
        OpenCV's 8-bit fixed-point INTER_LINEAR -- the reference resizes BEFORE it divides by 255, so the network sees quantised grey levels."""
        from .data import resize_linear_u8
        img = read_png(path)
        if img.ndim == 2:
            img = np.repeat(img[:, :, None], 3, 2)
        if img.dtype != np.uint8:                                         # (a 16-bit frame: cv2.imread's default flag reads it as 8 bits)
            img = (img >> 8).astype(np.uint8)
        return resize_linear_u8(np.ascontiguousarray(img[:, :, ::-1]), self.img_hw[1], self.img_hw[0])      # RGB file order -> BGR

    def __getitem__(self, idx):
        a = self._load(os.path.join(self.data_dir, 'image_2', str(idx).zfill(6) + '_10.png'))
        b = self._load(os.path.join(self.data_dir, 'image_2', str(idx).zfill(6) + '_11.png'))
        img = np.concatenate([a, b], 0) / 255.0                           # float64 (kitti_prepared.py:107), then .float() (kitti_2012.py:55)
        return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).float()


class KITTI_2015(KITTI_2012):
    num_total = 200
