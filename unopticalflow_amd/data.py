"""Inputs for the flow stage: frame triplets [B,3,3H,W] (left, centre, right stacked on H), BGR/255.

``SyntheticTriplets`` generates batches on the device (bench / smoke / --synthetic training).
``PreparedTriplets`` reads the stacked-triplet PNGs + train.txt that the reference's
``KITTI_RAW.prepare_data_mp`` / ``SINTEL_RAW.prepare_data_mp`` write (kitti_prepared.py:10-42,133-153)
with PIL's decoder instead of cv2's (then OpenCV's 8-bit resize arithmetic restated in ``resize_linear_u8``, random horizontal flip, / 255,
channel order BGR so the published checkpoints see what they were trained on), everything on the CPU like the reference.
``DecodedTriplets`` + ``DeviceTripletLoader`` are the MI355X input stage:
the workers only decode the PNG; resize / flip / scaling / layout run on the GPU in
``unflow_prepare_triplets`` with OpenCV's 8-bit arithmetic, on a side stream one batch ahead of the step.
"""
import os

import numpy as np
import torch
import torch.utils.data


def _linear_taps(dst_n, src_n, columns):
    """Tap indices and 11-bit fixed-point coefficients of OpenCV's INTER_LINEAR for one axis (cv::resize, resize.cpp: half-pixel centres,
    ``f = float((d + 0.5) * scale - 0.5)``; a column tap that leaves the row snaps to the border pixel with weight 1, a row tap is clamped
    and keeps its fraction; coefficients ``short(rint(c * 2048))``)."""
    scale = 1.0 / (float(dst_n) / float(src_n))
    f = ((np.arange(dst_n, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if columns:
        lo, hi = s < 0, s >= src_n - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, src_n - 1, s))
    i0, i1 = np.clip(s, 0, src_n - 1), np.clip(s + 1, 0, src_n - 1)
    return i0, i1, np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64), np.rint(f * np.float32(2048)).astype(np.int64)


def resize_linear_u8(img, W, H):
    """``cv2.resize(img, (W, H))`` for a uint8 [h, w, C] image -- what the reference applies to every frame it loads (kitti_prepared.py:51-76,
    kitti_2012.py:50-52): OpenCV's 8-bit INTER_LINEAR is integer arithmetic (horizontal pass in int32 with 11-bit coefficients, vertical pass
    ``(((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2``), and its uint8 result is what the network sees.  The host-side twin of
    ``unflow_prepare_triplets``'s resize (csrc/prepare.hip); equal sizes are a copy."""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 3:
        raise ValueError('resize_linear_u8 takes a uint8 [h, w, C] image, got %s %s' % (img.dtype, img.shape))
    h, w = img.shape[:2]
    if (h, w) == (H, W):
        return img.copy()
    x0, x1, a0, a1 = _linear_taps(W, w, True)
    y0, y1, b0, b1 = _linear_taps(H, h, False)
    p = img.astype(np.int64)
    rows = p[:, x0] * a0[None, :, None] + p[:, x1] * a1[None, :, None]
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


class SyntheticTriplets:
    def __init__(self, batch_size, img_hw, device, seed=0):
        self.shape = (batch_size, 3, 3 * img_hw[0], img_hw[1])
        self.device = device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)

    def __iter__(self):
        return self

    def __next__(self):
        return torch.rand(self.shape, generator=self.gen, device=self.device, dtype=torch.float32)


class PreparedTriplets(torch.utils.data.Dataset):
    def __init__(self, data_dir, num_scales=3, img_hw=(256, 832), num_iterations=None):
        self.data_dir, self.num_scales, self.img_hw, self.num_iterations = data_dir, num_scales, img_hw, num_iterations
        with open(os.path.join(data_dir, 'train.txt')) as f:
            self.files = [os.path.join(data_dir, ln.split()[0]) for ln in f if ln.strip()]
        print('A total of {} image pairs found'.format(len(self.files)))

    def count(self):
        return len(self.files)

    def __len__(self):
        return self.count() if self.num_iterations is None else self.num_iterations

    def rand_num(self, idx):                       # kitti_prepared.py:38-42
        np.random.seed(idx)
        return np.random.randint(self.count())

    def __getitem__(self, idx):
        from PIL import Image
        if self.num_iterations is not None:
            idx = self.rand_num(idx)
        img = np.asarray(Image.open(self.files[idx]).convert('RGB'))[:, :, ::-1]     # RGB -> BGR (what cv2.imread hands the reference)
        h = int(img.shape[0] / 3)                                                     # kitti_prepared.py:69
        frames = [resize_linear_u8(img[k * h:(k + 1) * h], self.img_hw[1], self.img_hw[0]) for k in range(3)]
        arr = np.concatenate(frames, 0)
        if np.random.rand() > 0.5:                                                   # kitti_prepared.py:79-81
            arr = arr[:, ::-1]
        arr = arr / 255.0                                                             # float64, then .float() (kitti_prepared.py:96,154)
        return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).float()


class DecodedTriplets(torch.utils.data.Dataset):
    """Same file list, sampling and flip draw as ``PreparedTriplets`` / kitti_prepared.py:38-42,80-84, but
    ``__getitem__`` stops after the decode: (uint8 [rows, w, 3] in file (RGB) order, flip flag)."""

    def __init__(self, data_dir, num_scales=3, img_hw=(256, 832), num_iterations=None):
        self.data_dir, self.num_scales, self.img_hw, self.num_iterations = data_dir, num_scales, img_hw, num_iterations
        with open(os.path.join(data_dir, 'train.txt')) as f:
            self.files = [os.path.join(data_dir, ln.split()[0]) for ln in f if ln.strip()]
        print('A total of {} image pairs found'.format(len(self.files)))

    def count(self):
        return len(self.files)

    def __len__(self):
        return self.count() if self.num_iterations is None else self.num_iterations

    def rand_num(self, idx):
        np.random.seed(idx)
        return np.random.randint(self.count())

    def __getitem__(self, idx):
        from .evaluation import read_png
        if self.num_iterations is not None:
            idx = self.rand_num(idx)
        img = read_png(self.files[idx])
        if img.dtype != np.uint8:
            raise ValueError('%s: expected an 8-bit PNG' % self.files[idx])
        if img.ndim == 2:
            img = np.repeat(img[:, :, None], 3, 2)
        flip = bool(np.random.rand() > 0.5)
        return torch.from_numpy(np.ascontiguousarray(img[:, :, :3])), flip


def _collate_decoded(samples):
    return [s[0] for s in samples], [s[1] for s in samples]


class DeviceTripletLoader:
    """Iterates ``[B,3,3H,W]`` device batches built by ``ops.prepare_triplets`` from a ``DecodedTriplets``.

    The upload + kernel of batch i+1 are enqueued on a side stream while the caller trains on batch i; the
    consumer stream waits on the batch's event, so there is no host synchronisation in the loop.  Two
    pinned staging buffers alternate; a buffer is rewritten only after the copy that read it has finished.
    """

    def __init__(self, dataset, batch_size, device, img_hw, num_workers=4, sampler=None, shuffle=True):
        self.loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=(shuffle and sampler is None),
                                                  sampler=sampler, num_workers=num_workers, drop_last=False,
                                                  collate_fn=_collate_decoded)
        self.device, self.img_hw = torch.device(device), img_hw
        self.stream = torch.cuda.Stream(device=self.device)
        self.staging = [None, None]
        self.copied = [None, None]

    def __len__(self):
        return len(self.loader)

    def _enqueue(self, images, flips, slot):
        from . import ops
        need = sum((im.numel() + 15) // 16 * 16 for im in images) + 64 * len(images) + 64
        if self.copied[slot] is not None:
            self.copied[slot].synchronize()                           # the copy that last read this buffer
        if self.staging[slot] is None or self.staging[slot].numel() < need:
            self.staging[slot] = torch.empty(need * 5 // 4, dtype=torch.uint8).pin_memory()
        with torch.cuda.stream(self.stream):
            batch = ops.prepare_triplets(images, self.img_hw, flips, self.device, True, self.staging[slot])
            ready = torch.cuda.Event()
            ready.record(self.stream)
        self.copied[slot] = ready
        return batch, ready

    def __iter__(self):
        pending, slot = None, 0
        for images, flips in self.loader:
            nxt = self._enqueue(images, flips, slot)
            slot ^= 1
            if pending is not None:
                yield self._hand_over(pending)
            pending = nxt
        if pending is not None:
            yield self._hand_over(pending)

    def _hand_over(self, item):
        batch, ready = item
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        batch.record_stream(cur)
        return batch
