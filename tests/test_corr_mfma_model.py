"""CPU model of the matrix-core cost-volume backward (unopticalflow_amd/csrc/corr_mfma.h) against the oracle's autograd of
corr_naive (reference pwc_tf.py:97-106).  Not the kernel -- the ALGORITHM the kernel implements, restated lane-free in numpy:
a wave = (16-pixel segment, chunk of rows); it walks the source rows of F; source row r feeds output rows y = r + R - i with
displacement row i as a banded 16 x 32 matrix product (band column = xl + j + 8 - R); the 2R + 1 rows in flight live in slots
that slide (acc[i + 1] = A B + acc[i]); row pairs outside the chunk are skipped; the chunk's last rows are read out of their
slots; gf2 takes plane (2R - i, 2R - j) at the displaced pixel.  With float64 operands the model must equal the oracle to
rounding (indexing proof); with both operands split into bf16 hi + lo parts and the lo x lo product dropped it must stay
within the error the GPU tests allow the kernel (tests/test_hip_ops.py::test_corr_backward_on_the_matrix_cores)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R


def _bf16_round(x):
    """float32 -> nearest-even bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _split(x):
    hi = _bf16_round(x)
    lo = _bf16_round((x.astype(np.float32) - hi).astype(np.float32))
    return hi.astype(np.float64), lo.astype(np.float64)


def model_backward(f1, f2, g, d, rows, mode, split, skip=True):
    """gf1 (mode 0) or gf2 (mode 1) of one cost volume the way corr_bwd_mf_body computes it."""
    B, C, H, W = f1.shape
    DD, SH = 2 * d + 1, 8 - d
    F = f1 if mode else f2
    out = np.full((B, C, H, W), np.nan)
    nseg, nchunk = -(-W // 16), -(-H // rows)
    gflat = g.reshape(B, -1)
    for b in range(B):
        for chunk in range(nchunk):
            ya, ybp = chunk * rows, min(chunk * rows + rows, H)
            r_begin, r_end = max(ya - d, 0), min(ybp - 1 + d, H - 1)
            nsteps = ((r_end - r_begin + 1 + 1) // 2) * 2              # two request sets: an even number of steps
            r_last = r_begin + nsteps - 1
            for S in range(nseg):
                acc = np.zeros((DD + 2, 16, C))
                for r in range(r_begin, r_last + 1):
                    Bm = np.zeros((32, C))                              # source columns 16 S - 8 .. 16 S + 23 of row r (zeros outside)
                    if r <= r_end:
                        for k in range(32):
                            x = 16 * S - 8 + k
                            if 0 <= x < W:
                                Bm[k] = F[b, :, r, x]
                    for i in range(DD - 1, -1, -1):
                        y = r + d - i
                        if skip == 'round':                               # (per round of four displacement rows: tuning variant)
                            lo_i, hi_i = 4 * (i // 4), min(4 * (i // 4) + 3, DD - 1)
                            if not (r + d - lo_i >= ya and r + d - hi_i < ybp):
                                continue
                        elif skip and not (ya <= y < ybp):
                            continue                                     # the slot keeps a stale value of a row outside the chunk
                        A = np.zeros((16, 32))
                        for xl in range(16):
                            xg = 16 * S + xl
                            if not (xg < W and ya <= y < ybp and r <= r_end):
                                continue
                            for n in range(DD):
                                if mode:
                                    j, off = 2 * d - n, ((2 * d - i) * DD + n) * H * W + r * W + xg + d - n
                                else:
                                    j, off = n, (i * DD + n) * H * W + y * W + xg
                                A[xl, xl + j + SH] = gflat[b, off] if 0 <= off < gflat.shape[1] else 0.0
                        prev = acc[i] if i else 0.0
                        if split:
                            (ah, al), (bh, bl) = _split(A), _split(Bm)
                            acc[i + 1] = ((prev + al @ bh) + ah @ bl) + ah @ bh
                        else:
                            acc[i + 1] = A @ Bm + prev
                    y = r - d
                    if ya <= y < ybp:
                        out[b, :, y, 16 * S: 16 * S + 16] = (acc[DD][: min(16, W - 16 * S)] / C).T
                for s in range(1, 2 * d + 1):
                    y = r_last + d + 1 - s
                    if ya <= y < ybp:
                        out[b, :, y, 16 * S: 16 * S + 16] = (acc[s][: min(16, W - 16 * S)] / C).T
    return out


@pytest.mark.parametrize('d,C,H,W,rows', [(4, 3, 13, 24, 8), (2, 2, 7, 20, 4), (8, 2, 19, 36, 16), (4, 2, 9, 16, 16), (4, 2, 5, 12, 8), (4, 2, 21, 40, 16)])
def test_walker_model_equals_the_oracle(d, C, H, W, rows):
    rng = np.random.default_rng(d * 100 + H)
    f1, f2 = rng.standard_normal((2, C, H, W)), rng.standard_normal((2, C, H, W))
    g = rng.standard_normal((2, (2 * d + 1) ** 2, H, W))
    t1, t2 = torch.from_numpy(f1).requires_grad_(), torch.from_numpy(f2).requires_grad_()
    R.corr_naive(t1, t2, d).backward(torch.from_numpy(g))
    for mode, ref in ((0, t1.grad.numpy()), (1, t2.grad.numpy())):
        got = model_backward(f1, f2, g, d, rows, mode, split=False)
        assert not np.isnan(got).any()
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
        # the stale-slot argument: skipping row pairs outside the chunk changes nothing
        np.testing.assert_array_equal(got, model_backward(f1, f2, g, d, rows, mode, split=False, skip=False))
        np.testing.assert_array_equal(got, model_backward(f1, f2, g, d, rows, mode, split=False, skip='round'))


def test_bf16_hi_lo_split_products_stay_inside_the_kernel_bar():
    """hi x hi + lo x hi + hi x lo with fp32-rounded inputs: a few 1e-6 of the largest gradient (the kernel measures 3.9e-6,
    profiles/r5_corr_bwd_mfma.md); the GPU test allows rtol 1e-4 + 1e-5 of the largest gradient."""
    d, C, H, W = 4, 32, 12, 48
    rng = np.random.default_rng(7)
    f1 = rng.uniform(-1, 1, (1, C, H, W)).astype(np.float32).astype(np.float64)
    f2 = rng.uniform(-1, 1, (1, C, H, W)).astype(np.float32).astype(np.float64)
    g = (0.05 * rng.standard_normal((1, 81, H, W))).astype(np.float32).astype(np.float64)
    t1, t2 = torch.from_numpy(f1).requires_grad_(), torch.from_numpy(f2).requires_grad_()
    R.corr_naive(t1, t2, d).backward(torch.from_numpy(g))
    for mode, ref in ((0, t1.grad.numpy()), (1, t2.grad.numpy())):
        got = model_backward(f1, f2, g, d, 8, mode, split=True)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert 0 < err < 1e-5, err
