"""A CPU stand-in for a slice of the C ABI (include/unflow_hip.h), built on the oracle -- TEST INFRASTRUCTURE ONLY.

The autograd wrappers of unopticalflow_amd/ops.py hand raw device pointers to libunflow_hip.so; on the build host there is no GPU, so
their own logic -- which tensor goes into which argument, the per-scale pointer tables, output shapes, what is saved for the backward and
in which position every gradient is returned -- never runs before a GPU session.  This module lets it run: ``patched(ops)`` swaps
``ops._call`` for an emulator that reads the same argument lists (written from the header, entry by entry), turns the pointers back into
views of the CPU tensors they came from, and computes what the entry is documented to compute with the oracle's functions
(oracle/ref_cpu.py) and torch autograd.  tests/test_ops_plumbing_cpu.py drives the round-5 multi-scale operators through it and compares
them with the oracle called directly.

It says nothing about the kernels (tests -m gpu do), and the product never imports it: no CPU fallback exists outside this file.
"""
import contextlib
import ctypes

import numpy as np
import torch

from oracle import ref_cpu as R


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, ctypes.c_void_p):
        return p.value or 0
    return int(p)


def f32(p, *shape):
    """A float32 view of host memory at address p."""
    n = int(np.prod(shape))
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(_addr(p)))).view(*shape)


def u8(p, *shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(_addr(p)))).view(*shape)


def _sums_into(partials, nblk, s0, s1):
    """What a first stage leaves: nblk (s0, s1) pairs per sample -- here all of a sample's sum in its first pair."""
    B = s0.shape[0]
    p = partials.view(-1)[:B * nblk * 2].view(B, nblk, 2)
    p.zero_()
    p[:, 0, 0], p[:, 0, 1] = s0, s1


class Emulator:
    def __init__(self, lib):
        self.lib = lib                                                  # the real library: its HOST entries (block counts) are called as they are
        self.calls = []

    def __call__(self, name, *args, nbytes=0, shape=None):
        self.calls.append(name)
        fn = getattr(self, name, None)
        if fn is None:
            raise NotImplementedError('abi_emulator: %s is not emulated' % name)
        with torch.enable_grad():                                       # (called from inside autograd.Function.backward, where grad mode is off)
            fn(*args)

    # ---- int unflow_loss_finalize_batch(partials[], loss[], sums[], nblk[], B[], kind[], n0[], n1[], njobs, stream)
    def unflow_loss_finalize_batch(self, partials, loss, sums, nblk, B, kind, n0, n1, njobs, stream):
        for q in range(njobs):
            p = f32(partials[q], B[q], nblk[q], 2)
            s0, s1 = p[:, :, 0].sum(1), p[:, :, 1].sum(1)
            out = f32(loss[q], B[q])
            if kind[q] == 0:
                out.copy_((s0 / n0[q]) / (s1 / n1[q] + 1e-12))
                if sums[q]:
                    f32(sums[q], B[q], 2).copy_(torch.stack((s0, s1), 1))
            else:
                out.copy_((s0 / n0[q] + s1 / n1[q]) / 2.0)

    # ---- int unflow_loss_combine_fwd(terms[4 * n], n_scales, B, outs[4], stream) / _bwd(gouts[4], B, gin, stream)
    def unflow_loss_combine_fwd(self, terms, n, B, outs, stream):
        for k in range(4):
            width = 2 * B if k < 3 else B
            tot = torch.zeros(width)
            for s in range(n):
                tot = tot + f32(terms[k * n + s], width)
            f32(outs[k], B).copy_(tot[B:] + tot[:B] if k < 3 else tot)

    def unflow_loss_combine_bwd(self, gouts, B, gin, stream):
        g = f32(gin, 7 * B)
        for k in range(3):
            v = f32(gouts[k], B) if gouts[k] else torch.zeros(B)
            g[k * 2 * B:k * 2 * B + B] = v
            g[k * 2 * B + B:(k + 1) * 2 * B] = v
        g[6 * B:] = f32(gouts[3], B) if gouts[3] else torch.zeros(B)

    # ---- the `_ms` entries (ABI 11): every array one entry per scale
    def unflow_occ_weight_fwd_ms(self, n, img, warped, diff, wgt, H, W, B, stream):
        for k in range(n):
            i, w = f32(img[k], B, 3, H[k], W[k]), f32(warped[k], 2 * B, 3, H[k], W[k])
            d_l, d_r, w_b, w_f, _, _ = R.diff_weight(i, w[:B], w[B:])
            f32(diff[k], 2 * B, 1, H[k], W[k]).copy_(torch.cat((d_l, d_r)))
            f32(wgt[k], 2 * B, 1, H[k], W[k]).copy_(torch.cat((w_b, w_f)))

    def unflow_absdiff_bwd_ms(self, n, img, frm, gdiff, gfrom, H, W, B, img_batch, stream):
        for k in range(n):
            i = f32(img[k], img_batch, 3, H[k], W[k]).repeat(B // img_batch, 1, 1, 1)
            f = f32(frm[k], B, 3, H[k], W[k]).clone().requires_grad_()
            torch.abs(i - f).mean(1, True).backward(f32(gdiff[k], B, 1, H[k], W[k]))
            f32(gfrom[k], B, 3, H[k], W[k]).copy_(f.grad)

    def unflow_masked_mean_fwd_ms(self, n, diff, w, partials, H, W, B, stream):
        for k in range(n):
            d, wt = f32(diff[k], B, H[k] * W[k]), f32(w[k], B, H[k] * W[k])
            _sums_into(f32(partials[k], B * self.lib.unflow_partials_per_sample(H[k], W[k])), self.lib.unflow_loss_partial_blocks(0, H[k], W[k], B, 1),
                       (d * wt).sum(1), wt.sum(1))

    def unflow_masked_mean_bwd_ms(self, n, w, sums, gloss, gdiff, H, W, B, stream):
        for k in range(n):
            hw = float(H[k] * W[k])
            wt, s, g = f32(w[k], B, H[k] * W[k]), f32(sums[k], B, 2), f32(gloss[k], B)
            f32(gdiff[k], B, H[k] * W[k]).copy_((g / hw / (s[:, 1] / hw + 1e-12)).view(B, 1) * wt)

    def unflow_ssim_loss_fwd_ms(self, n, img, warped, w, partials, H, W, B, img_batch, stream):
        for k in range(n):
            i = f32(img[k], img_batch, 3, H[k], W[k]).repeat(B // img_batch, 1, 1, 1)
            y, wt = f32(warped[k], B, 3, H[k], W[k]), f32(w[k], B, 1, H[k], W[k])
            w3 = wt.repeat(1, 3, 1, 1)
            c = torch.clamp((1.0 - R.SSIM(i * w3, y * w3)) / 2.0, 0, 1)
            _sums_into(f32(partials[k], B * self.lib.unflow_partials_per_sample(H[k], W[k])), self.lib.unflow_loss_partial_blocks(1, H[k], W[k], B, 1),
                       c.sum((1, 2, 3)), wt.sum((1, 2, 3)))

    def unflow_ssim_loss_bwd_ms(self, n, img, warped, w, sums, gloss, gwarped, H, W, B, img_batch, stream):
        for k in range(n):
            i = f32(img[k], img_batch, 3, H[k], W[k]).repeat(B // img_batch, 1, 1, 1)
            y, wt = f32(warped[k], B, 3, H[k], W[k]).clone().requires_grad_(), f32(w[k], B, 1, H[k], W[k])
            R.ssim_loss(i, y, wt).backward(f32(gloss[k], B))
            f32(gwarped[k], B, 3, H[k], W[k]).copy_(y.grad)

    def unflow_smooth2_fwd_ms(self, n, flow, img, partials, H, W, B, img_batch, stream):
        gx = lambda t: t[:, :, :, 1:] - t[:, :, :, :-1]
        gy = lambda t: t[:, :, 1:, :] - t[:, :, :-1, :]
        for k in range(n):
            f = f32(flow[k], B, 2, H[k], W[k]) / 20.0
            i = f32(img[k], img_batch, 3, H[k], W[k]).repeat(B // img_batch, 1, 1, 1)
            w_x = torch.exp(-10.0 * torch.abs(gx(i)).mean(1).unsqueeze(1))
            w_y = torch.exp(-10.0 * torch.abs(gy(i)).mean(1).unsqueeze(1))
            _sums_into(f32(partials[k], B * self.lib.unflow_partials_per_sample(H[k], W[k])), self.lib.unflow_loss_partial_blocks(2, H[k], W[k], B, 1),
                       (w_x[:, :, :, 1:] * torch.abs(gx(gx(f)))).sum((1, 2, 3)), (w_y[:, :, 1:, :] * torch.abs(gy(gy(f)))).sum((1, 2, 3)))

    def unflow_smooth2_bwd_ms(self, n, flow, img, gloss, gflow, H, W, B, img_batch, stream):
        for k in range(n):
            f = f32(flow[k], B, 2, H[k], W[k]).clone().requires_grad_()
            i = f32(img[k], img_batch, 3, H[k], W[k]).repeat(B // img_batch, 1, 1, 1)
            R.grad2_error(f / 20.0, i).backward(f32(gloss[k], B))
            f32(gflow[k], B, 2, H[k], W[k]).copy_(f.grad)

    def unflow_consis_fwd_ms(self, n, ff, fb, w, partials, H, W, B, stream):
        for k in range(n):
            a, b, wt = f32(ff[k], B, 2, H[k], W[k]), f32(fb[k], B, 2, H[k], W[k]), f32(w[k], B, 1, H[k], W[k])
            occ = 1 - wt
            s0 = (torch.abs(R.flow_normalization(a) + R.flow_normalization(b)) * occ).sum((1, 2, 3))
            _sums_into(f32(partials[k], B * self.lib.unflow_partials_per_sample(H[k], W[k])), self.lib.unflow_loss_partial_blocks(3, H[k], W[k], B, 1),
                       s0, occ.sum((1, 2, 3)))

    def unflow_consis_bwd_ms(self, n, ff, fb, w, sums, gloss, gflow, H, W, B, stream):
        for k in range(n):
            a = f32(ff[k], B, 2, H[k], W[k]).clone().requires_grad_()
            R.consis_loss(a, f32(fb[k], B, 2, H[k], W[k]), f32(w[k], B, 1, H[k], W[k])).backward(f32(gloss[k], B))
            f32(gflow[k], B, 2, H[k], W[k]).copy_(a.grad)

    def unflow_warp_fwd_ms(self, n, src, flow, out, mask, H, W, B, C, align_corners, stream):
        for k in range(n):
            x, f = f32(src[k], B, C, H[k], W[k]), f32(flow[k], B, 2, H[k], W[k])
            f32(out[k], B, C, H[k], W[k]).copy_(R.warp_flow(x, f, True, bool(align_corners)))
            u8(mask[k], B, 1, H[k], W[k]).copy_(R.warp_mask(x.shape, f, bool(align_corners)))

    def unflow_warp_bwd_ms(self, n, src, flow, gout, mask, gflow, H, W, B, C, align_corners, stream):
        for k in range(n):
            x, f = f32(src[k], B, C, H[k], W[k]), f32(flow[k], B, 2, H[k], W[k]).clone().requires_grad_()
            R.warp_flow(x, f, True, bool(align_corners)).backward(f32(gout[k], B, C, H[k], W[k]))
            f32(gflow[k], B, 2, H[k], W[k]).copy_(f.grad)


    # ---- the single-scale entries of the same operators: the `_ms` emulation at n = 1, plus the immediate second stage when `loss` is given
    @staticmethod
    def _one(*v):
        return [[x] for x in v]

    def _finish(self, op, partials, loss, sums, B, H, W, kind, n0, n1):
        if loss:
            nblk = self.lib.unflow_loss_partial_blocks(op, H, W, B, 1)
            self.unflow_loss_finalize_batch([partials], [loss], [sums], [nblk], [B], [kind], [n0], [n1], 1, None)

    def unflow_occ_weight_fwd(self, img, from_l, from_r, diff_l, diff_r, w_bwd, w_fwd, valid_bwd, valid_fwd, B, H, W, stream):
        i = f32(img, B, 3, H, W)
        d_l, d_r, w_b, w_f, v_b, v_f = R.diff_weight(i, f32(from_l, B, 3, H, W), f32(from_r, B, 3, H, W))
        for dst, src in ((diff_l, d_l), (diff_r, d_r), (w_bwd, w_b), (w_fwd, w_f)):
            f32(dst, B, 1, H, W).copy_(src)
        for dst, src in ((valid_bwd, v_b), (valid_fwd, v_f)):
            if _addr(dst):
                u8(dst, B, 1, H, W).copy_(src.to(torch.uint8))

    def unflow_absdiff_bwd(self, img, frm, gdiff, gfrom, B, H, W, img_batch, stream):
        self.unflow_absdiff_bwd_ms(1, *self._one(img, frm, gdiff, gfrom, H, W), B, img_batch, stream)

    def unflow_masked_mean_fwd(self, diff, w, loss, sums, partials, B, H, W, stream):
        self.unflow_masked_mean_fwd_ms(1, *self._one(diff, w, partials, H, W), B, stream)
        self._finish(0, partials, loss, sums, B, H, W, 0, float(H * W), float(H * W))

    def unflow_masked_mean_bwd(self, w, sums, gloss, gdiff, B, H, W, stream):
        self.unflow_masked_mean_bwd_ms(1, *self._one(w, sums, gloss, gdiff, H, W), B, stream)

    def unflow_ssim_loss_fwd(self, img, warped, w, loss, sums, partials, B, H, W, img_batch, stream):
        self.unflow_ssim_loss_fwd_ms(1, *self._one(img, warped, w, partials, H, W), B, img_batch, stream)
        self._finish(1, partials, loss, sums, B, H, W, 0, 3.0 * H * W, float(H * W))

    def unflow_ssim_loss_bwd(self, img, warped, w, sums, gloss, gwarped, B, H, W, img_batch, stream):
        self.unflow_ssim_loss_bwd_ms(1, *self._one(img, warped, w, sums, gloss, gwarped, H, W), B, img_batch, stream)

    def unflow_smooth2_fwd(self, flow, img, loss, partials, B, H, W, img_batch, stream):
        self.unflow_smooth2_fwd_ms(1, *self._one(flow, img, partials, H, W), B, img_batch, stream)
        self._finish(2, partials, loss, None, B, H, W, 1, 2.0 * H * (W - 2), 2.0 * (H - 2) * W)

    def unflow_smooth2_bwd(self, flow, img, gloss, gflow, B, H, W, img_batch, stream):
        self.unflow_smooth2_bwd_ms(1, *self._one(flow, img, gloss, gflow, H, W), B, img_batch, stream)

    def unflow_consis_fwd(self, ff, fb, w, loss, sums, partials, B, H, W, stream):
        self.unflow_consis_fwd_ms(1, *self._one(ff, fb, w, partials, H, W), B, stream)
        self._finish(3, partials, loss, sums, B, H, W, 0, 2.0 * H * W, float(H * W))

    def unflow_consis_bwd(self, ff, fb, w, sums, gloss, gflow, B, H, W, stream):
        self.unflow_consis_bwd_ms(1, *self._one(ff, fb, w, sums, gloss, gflow, H, W), B, stream)

    # masked image warps only (mask given, no source gradient): what Model_flow.warp_flow_pyramid asks for
    def unflow_warp_fwd(self, src, flow, out, mask, B, C, H, W, align_corners, stream):
        assert _addr(mask), 'abi_emulator: only the masked image warp is emulated'
        self.unflow_warp_fwd_ms(1, *self._one(src, flow, out, mask, H, W), B, C, align_corners, stream)

    def unflow_warp_bwd(self, src, flow, gout, mask, gsrc, gflow, B, C, H, W, align_corners, stream):
        assert _addr(mask) and not _addr(gsrc), 'abi_emulator: only the masked image warp (no source gradient) is emulated'
        self.unflow_warp_bwd_ms(1, *self._one(src, flow, gout, mask, gflow, H, W), B, C, align_corners, stream)

    # ---- int unflow_img_pyramid(img, half, quarter, planes, H, W, stream): 2x2 and 4x4 box means
    def unflow_img_pyramid(self, img, half, quarter, planes, H, W, stream):
        x = f32(img, 1, planes, H, W)
        f32(half, 1, planes, H // 2, W // 2).copy_(torch.nn.functional.adaptive_avg_pool2d(x, [H // 2, W // 2]))
        f32(quarter, 1, planes, H // 4, W // 4).copy_(torch.nn.functional.adaptive_avg_pool2d(x, [H // 4, W // 4]))

    # ---- int unflow_to_nchw_dup(in NHWC [Bin][HW][C], out NCHW [Bin + dup][C][HW], C, Bin, dup, HW, stream) and the gradient's way back,
    # int unflow_to_nhwc_fold(g NCHW [Bout + dup][C][HW], out NHWC [Bout][HW][C], C, Bout, dup, HW, stream)
    def unflow_to_nchw_dup(self, src, out, C, Bin, dup, HW, stream):
        x = f32(src, Bin, HW, C).permute(0, 2, 1)
        f32(out, Bin + dup, C, HW).copy_(torch.cat((x, x[Bin - dup:]), 0) if dup else x)

    def unflow_to_nhwc_fold(self, g, out, C, Bout, dup, HW, stream):
        gg = f32(g, Bout + dup, C, HW)
        r = gg[:Bout].clone()
        if dup:
            r[Bout - dup:] += gg[Bout:]
        f32(out, Bout, HW, C).copy_(r.permute(0, 2, 1))

    # ---- int unflow_weighted_mean_sum_fwd(terms[K], weights[K], K, B, loss, stream) / _bwd(gloss, weights, K, B, grads[K], stream)
    def unflow_weighted_mean_sum_fwd(self, terms, weights, K, B, loss, stream):
        f32(loss, 1).copy_(sum(weights[k] * f32(terms[k], B).mean() for k in range(K)).reshape(1))

    def unflow_weighted_mean_sum_bwd(self, gloss, weights, K, B, grads, stream):
        for k in range(K):
            f32(grads[k], B).copy_((f32(gloss, 1) * weights[k] / B).expand(B))

@contextlib.contextmanager
def patched(ops):
    """Inside the block ``ops`` accepts CPU tensors and its C calls go to the emulator (yielded: ``.calls`` lists the entry names)."""
    from unopticalflow_amd import _lib
    emu = Emulator(_lib.load())
    old = (ops._call, ops._dev, ops._stream, ops._on)

    def dev(*tensors):
        for t in tensors:
            if t is not None and t.dtype not in (torch.float32,):
                raise TypeError('unopticalflow_amd ops compute in fp32; got %s' % t.dtype)
        return torch.device('cpu')
    ops._call, ops._dev, ops._stream, ops._on = emu, dev, (lambda: None), (lambda d: contextlib.nullcontext())
    try:
        yield emu
    finally:
        ops._call, ops._dev, ops._stream, ops._on = old
