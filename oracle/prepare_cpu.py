"""CPU oracle for the input stage of the flow train step.

TEST INFRASTRUCTURE ONLY -- same rule as ``oracle/ref_cpu.py``: only ``tests/``, ``smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this; the product (``unflow_prepare_triplets``) never does.

Restates ``KITTI_Prepared.__getitem__`` after the PNG decode (core/dataset/kitti_prepared.py:63-90,
133-148): split the stacked image into three frames of ``int(rows / 3)`` rows, ``cv2.resize`` each to
``img_hw`` (default interpolation INTER_LINEAR), ``cv2.flip(img, 1)`` with probability 0.5, ``/ 255.0``
(float64), HWC -> CHW, ``.float()``.  Channel order is whatever ``cv2.imread`` produced (BGR).

PARITY UNPINNED for the resize: the arithmetic lives in a third-party dependency that is neither under
/root/reference nor installed here -- OpenCV, pinned upstream at ``opencv-python==4.1.1.26``
(requirements.txt:13).  ``cv2_resize_linear_u8`` restates that library's published 8-bit INTER_LINEAR
algorithm (modules/imgproc/src/resize.cpp: ``resizeGeneric_`` with ``HResizeLinear<uchar,int,short,2048>``
and the fixed-point ``VResizeLinear``; the IPP branch is skipped for 8-bit linear because it is not
bit-compatible):
  * ``scale = 1.0 / (double(dst) / src)``; per destination index ``f = float((d + 0.5) * scale - 0.5)``,
    ``s = floor(f)``, ``f -= s``;
  * columns: ``s < 0 -> (f, s) = (0, 0)``; ``s >= src_w - 1 -> (f, s) = (0, src_w - 1)``;
    rows: the two taps are clamped to ``[0, src_h - 1]``, ``f`` is kept;
  * coefficients ``short(round_half_even(c * 2048))`` for ``c in (1 - f, f)`` (float arithmetic);
  * horizontal pass in int32: ``S = p[s] * a0 + p[s + 1] * a1``;
  * vertical pass: ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``.
An exact 2x reduction takes OpenCV's INTER_AREA fast path, ``(a + b + c + d + 2) >> 2``, which this
formula reproduces identically; equal sizes are a copy.  What the tests can check without cv2: the result
is within 1 grey level of the float half-pixel-centre bilinear interpolation, the 2x and 1x identities, and
bit-equality between this restatement and the HIP kernel.
"""
import numpy as np


def _taps(dst_n, src_n, clamp_fraction):
    """-> (i0, i1, c0, c1) int arrays of length dst_n."""
    scale = 1.0 / (float(dst_n) / float(src_n))                       # doubles, like cv::resize / hal::resize
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_fraction:                                                 # columns
        lo, hi = s < 0, s >= src_n - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, src_n - 1, s))
        i0, i1 = s, np.minimum(s + 1, src_n - 1)
    else:                                                              # rows: clip the tap indices only
        i0, i1 = np.clip(s, 0, src_n - 1), np.clip(s + 1, 0, src_n - 1)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return i0, i1, c0, c1


def cv2_resize_linear_u8(img, W, H):
    """uint8 [h,w,C] -> uint8 [H,W,C]; OpenCV 4.1 ``cv2.resize(img, (W, H))`` for 8-bit input."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    x0, x1, a0, a1 = _taps(W, w, True)
    y0, y1, b0, b1 = _taps(H, h, False)
    p = img.astype(np.int64)
    rows = p[:, x0] * a0[None, :, None] + p[:, x1] * a1[None, :, None]          # [h, W, C]
    s0, s1 = rows[y0] >> 4, rows[y1] >> 4
    out = (((b0[:, None, None] * s0) >> 16) + ((b1[:, None, None] * s1) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def prepare_triplet(img, img_hw, flip):
    """Decoded stacked triplet uint8 [rows, w, 3] -> float32 [3, 3H, W] (kitti_prepared.py:63-90,145-148)."""
    H, W = img_hw
    h = int(img.shape[0] / 3)
    frames = [cv2_resize_linear_u8(img[k * h:(k + 1) * h], W, H) for k in range(3)]
    out = np.concatenate(frames, 0)
    if flip:
        out = out[:, ::-1]                                             # cv2.flip(img, 1)
    out = out / 255.0                                                  # float64, like the reference
    return np.ascontiguousarray(out.transpose(2, 0, 1)).astype(np.float32)
