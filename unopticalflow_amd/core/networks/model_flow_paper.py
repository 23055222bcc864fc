import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops, tuning
from .structures import FeaturePyramid, PWC_tf, warp_flow
from .structures.net_utils import WeightShadows


class Model_flow(nn.Module):
    """Flow-stage model + unsupervised losses (reference core/networks/model_flow_paper.py:14-255).

    Same constructor, attributes (``fpyramid``, ``pwc_model`` -> same checkpoint keys), methods and
    return values as the reference.  What changed underneath:
      * the three frames go through the pyramid as ONE 3B batch and the two directed pairs
        (centre->left, centre->right) through the decoder as ONE 2B batch (the weights are
        shared, samples are independent), so every conv / corr / warp launch is 2-3x larger;
      * corr, warp(+mask), occlusion weights, masked-L1, SSIM, smoothness and consistency
        are HIP kernels (``unopticalflow_amd.ops``) instead of eager op chains;
      * the 1/8-scale image warp the reference computes and discards
        (model_flow_paper.py:62-66 with num_scales=3) is not computed.
    ``cfg.align_corners`` (optional, default False) selects the grid_sample generation;
    ``cfg.precision`` (optional, 'fp32' | 'bf16') the conv-stack precision; ``cfg.channels_last`` (optional, default:
    ``tuning.default_channels_last()``) the memory format of the conv stacks' activations and weights.
    """

    def __init__(self, cfg):
        super(Model_flow, self).__init__()
        self.align_corners = bool(getattr(cfg, 'align_corners', False))
        # 'fp32' (default, the parity configuration) or 'bf16': the conv stacks run under bf16 autocast (MFMA
        # bf16), cost volume / warp / losses keep fp32 inputs and accumulation (BASELINE config 3)
        self.precision = getattr(cfg, 'precision', 'fp32')
        if self.precision not in ('fp32', 'bf16'):
            raise ValueError('precision must be fp32 or bf16, got {}'.format(self.precision))
        # conv stacks on channels_last tensors (NHWC: no MIOpen transposes) when asked for, by default when MIOpen runs on the
        # shipped find-db's measured picks (tuning.default_channels_last: the one place that decides), else NCHW
        cl = getattr(cfg, 'channels_last', None)
        self.channels_last = tuning.default_channels_last() if cl is None else bool(cl)
        # bf16 option: one multi-tensor cast of all convolution weights per pass (False: autocast's per-call casts; tests compare)
        self.weight_shadows = bool(getattr(cfg, 'weight_shadows', True))
        self.weight_shadow_groups = int(getattr(cfg, 'weight_shadow_groups', 1))     # cast nodes of the bf16 option (net_utils.WeightShadows)
        self.dup_centre = True           # the pyramid hand-off writes the centre features twice (False: torch.cat((c, c)); A/B)
        self.fused_loss_sums = True      # the sums over scales and directions of forward() as one launch each way (False: eager adds; tests compare)
        # True: the second stage of the per-sample loss reductions as ONE launch in front of loss_combine (False: one per reduction; same bits on the
        # build host).  Round 5's GPU evidence for it is smoke() + bench lines, no complete `-m gpu` run, so it is off until one has passed
        # (tests/test_zz_round5_gpu.py::test_deferred_loss_sums_are_the_same_bits; bench.py --deferred-loss-sums 1)
        self.deferred_loss_sums = bool(getattr(cfg, 'deferred_loss_sums', False))
        # every loss of the scale loop as ONE launch over the scales per direction of the pass (ops.multiscale_losses, csrc/multiscale.h): same kernel
        # bodies, same bits, 30 loss launches -> 10.  Off until its GPU tests have run (written after the round-5 GPU lease closed); bench.py
        # --multiscale-losses 1 / tests/test_zz_round5_gpu.py switch it on
        self.multiscale_losses = bool(getattr(cfg, 'multiscale_losses', False))
        # the pyramid hand-off returns (left | right) and (centre | centre) as two tensors (ops.to_nchw_split): no `split`, whose backward copies every
        # level's gradient into one buffer again (125 us/step of ATen copies in the round-4 trace).  Validated kernels, new autograd node: off until measured
        self.split_handoff = bool(getattr(cfg, 'split_handoff', False))
        self.fpyramid = FeaturePyramid(channels_last=self.channels_last)
        self.pwc_model = PWC_tf(align_corners=self.align_corners, fused_warp_corr=bool(getattr(cfg, 'fused_warp_corr', False)),
                                channels_last=self.channels_last)
        lv = getattr(cfg, 'fused_warp_corr_levels', None)        # e.g. '5' or (4, 5): only those levels take the fused warp + cost-volume kernel
        if lv:
            self.pwc_model.fused_levels = frozenset(int(v) for v in (str(lv).split(',') if isinstance(lv, (str, int)) else lv))
        if cfg.mode == 'depth' or cfg.mode == 'flowposenet':
            # Stage 2 training
            for param in self.fpyramid.parameters():
                param.requires_grad = False
            for param in self.pwc_model.parameters():
                param.requires_grad = False

        # hyperparameters
        self.dataset = cfg.dataset
        self.num_scales = cfg.num_scales
        self.flow_consist_alpha = cfg.h_flow_consist_alpha
        self.flow_consist_beta = cfg.h_flow_consist_beta

    # ---- small helpers kept for surface parity (reference :36-60) ----
    def get_flow_norm(self, flow, p=2):
        return torch.norm(flow, p=p, dim=1).unsqueeze(1) + 1e-12

    def get_flow_normalization(self, flow, p=2):
        flow_norm = torch.norm(flow, p=p, dim=1).unsqueeze(1) + 1e-12
        return flow / flow_norm.repeat(1, 2, 1, 1)

    def generate_img_pyramid(self, img, num_pyramid):
        img_h, img_w = img.shape[2], img.shape[3]
        return [F.adaptive_avg_pool2d(img, [int(img_h / (2 ** s)), int(img_w / (2 ** s))]).data
                for s in range(num_pyramid)]

    def warp_flow_pyramid(self, img_pyramid, flow_pyramid):
        n = min(len(img_pyramid), len(flow_pyramid))
        if (self.multiscale_losses and ops.multiscale_supported(img_pyramid[:n], flow_pyramid[:n]) and img_pyramid[0].shape[1] <= 4
                and not any(i.requires_grad for i in img_pyramid[:n])):
            # one launch over the scales each way (ops.warp_flow_masked_pyramid; same bits)
            return ops.warp_flow_masked_pyramid(img_pyramid[:n], flow_pyramid[:n], align_corners=self.align_corners)
        return [warp_flow(img, flow, use_mask=True, align_corners=self.align_corners)
                for img, flow in zip(img_pyramid, flow_pyramid)]

    def compute_loss_pixel(self, img_pyramid, img_warped_pyramid, occ_mask_list):
        """reference :68-77 (not called by forward(); kept for surface parity, same HIP masked mean)"""
        loss = 0
        for scale in range(self.num_scales):
            diff = torch.abs(img_pyramid[scale] - img_warped_pyramid[scale]).mean(1, True)
            loss = loss + ops.masked_mean(diff, occ_mask_list[scale])
        return loss

    def compute_loss_pixel_without_mask(self, img_pyramid, img_warped_pyramid):
        """reference :79-87 (not called by forward())"""
        loss = 0
        for scale in range(self.num_scales):
            loss = loss + torch.abs(img_pyramid[scale] - img_warped_pyramid[scale]).mean((1, 2, 3))
        return loss

    # ---- losses (one HIP op per scale each) ----
    def compute_loss_with_mask(self, diff_list, occ_mask_list):
        """reference :90-99"""
        loss = 0
        for scale in range(self.num_scales):
            loss = loss + ops.masked_mean(diff_list[scale], occ_mask_list[scale])
        return loss

    def compute_diff_weight(self, img_pyramid_from_l, img_pyramid, img_pyramid_from_r):
        """reference :101-134 -> diff_bwd, diff_fwd, weight_bwd, weight_fwd (lists over scales)"""
        diff_fwd, diff_bwd, weight_fwd, weight_bwd = [], [], [], []
        for scale in range(self.num_scales):
            d_l, d_r, w_b, w_f, _, _ = ops.occ_weight(img_pyramid[scale], img_pyramid_from_l[scale],
                                                      img_pyramid_from_r[scale])
            diff_bwd.append(d_l); diff_fwd.append(d_r)
            weight_bwd.append(w_b); weight_fwd.append(w_f)
        return diff_bwd, diff_fwd, weight_bwd, weight_fwd

    def compute_loss_ssim(self, img_pyramid, img_warped_pyramid, occ_mask_list):
        """reference :137-148"""
        loss = 0
        for scale in range(self.num_scales):
            loss = loss + ops.ssim_loss(img_pyramid[scale], img_warped_pyramid[scale], occ_mask_list[scale])
        return loss

    def gradients(self, img):
        dy = img[:, :, 1:, :] - img[:, :, :-1, :]
        dx = img[:, :, :, 1:] - img[:, :, :, :-1]
        return dx, dy

    def cal_grad2_error(self, flow, img):
        """reference :157-167; ``flow`` is already divided by 20 as at the call site :174."""
        return ops.smooth2_loss(flow * 20.0, img)

    def compute_loss_flow_smooth(self, optical_flows, img_pyramid):
        """reference :169-177 (the /20 of :174 happens inside the kernel)"""
        loss = 0
        for scale in range(self.num_scales):
            loss = loss + ops.smooth2_loss(optical_flows[scale], img_pyramid[scale])
        return loss

    def compute_loss_flow_consis(self, fwd_flow_pyramid, bwd_flow_pyramid, occ_mask_list):
        """reference :180-195"""
        loss = 0
        for scale in range(self.num_scales):
            loss = loss + ops.consis_loss(fwd_flow_pyramid[scale], bwd_flow_pyramid[scale], occ_mask_list[scale])
        return loss

    # ---- network passes ----
    def inference_flow(self, img1, img2):
        """reference :198-202 -> full-resolution flow [B,2,H,W]"""
        img_hw = [img1.shape[2], img1.shape[3]]
        B = img1.shape[0]
        with self._autocast():
            feats = self.fpyramid(torch.cat((img1, img2), 0))
            feature_list_1 = [f[:B] for f in feats]
            feature_list_2 = [f[B:] for f in feats]
            return self.pwc_model(feature_list_1, feature_list_2, img_hw)[0].float()

    @contextlib.contextmanager
    def _autocast(self):
        """The conv stacks' precision for one network pass: nothing for fp32; for bf16 the autocast region plus one bf16 copy of
        every convolution weight for the whole pass (net_utils.WeightShadows) instead of autocast's cast per call."""
        if self.precision != 'bf16':
            yield
            return
        with torch.autocast('cuda', dtype=torch.bfloat16):
            on_gpu = self.weight_shadows and next(self.fpyramid.parameters()).is_cuda
            with WeightShadows((self.fpyramid, self.pwc_model), enabled=on_gpu, groups=self.weight_shadow_groups):
                yield

    def _flows(self, imgl, img, imgr, frames=None):
        """Both directed flow pyramids with one 3B pyramid pass and one 2B decoder pass.  ``frames`` is the
        [3B,3,H,W] batch ordered (left, right, centre), so the decoder's second inputs (left | right) are a view of
        the pyramid output; returns per scale the stacked flows [2B,2,h,w] = (centre->left | centre->right)."""
        B, _, img_h, img_w = img.shape
        if frames is None:
            frames = torch.cat((imgl, imgr, img), 0)
        with self._autocast():
            # [4B, ...] per level: (left | right | centre | centre) -- the hand-off out of the conv stack writes the centre features
            # twice (FeaturePyramid dup_tail), so both decoder inputs are views and the two gradients of the centre features meet
            # in the hand-off's backward kernel
            feats = self.fpyramid(frames, dup_tail=B if self.dup_centre else 0, split_head=2 * B if (self.dup_centre and self.split_handoff) else 0)
            # the decoder never reads pyramid level 1 (pwc_tf.py:108-179 uses c12..c16 / c22..c26): nothing built for it.
            # split (not slices): its backward is one cat, a slice's is a zero-fill + copy + add of the whole map
            if self.dup_centre:
                parts = [f if isinstance(f, tuple) else f.split((2 * B, 2 * B)) for f in feats[1:]]         # (left | right), (centre | centre)
            else:
                parts = [f.split((2 * B, B)) for f in feats[1:]]
                parts = [(lr, torch.cat((c, c), 0)) for lr, c in parts]
            feat_lr = [None] + [lr for lr, _ in parts]
            feat_c2 = [None] + [c2 for _, c2 in parts]
            flows = self.pwc_model(feat_c2, feat_lr, [img_h, img_w])         # [2B, 2, h, w] per scale
        return [f.float() for f in flows]

    def forward(self, inputs, output_flow=False, use_flow_loss=True, is_second_phase=False):
        images = inputs
        assert (images.shape[1] == 3)
        img_h, img_w = int(images.shape[2] / 3), images.shape[3]
        imgl, img, imgr = images[:, :, :img_h, :], images[:, :, img_h:2 * img_h, :], images[:, :, 2 * img_h:3 * img_h, :]

        B = images.shape[0]
        # the three frames as one contiguous [3B,3,H,W] batch (left, right, centre): feeds the 3B pyramid pass and,
        # through one HIP pooling kernel, all three image pyramids; (left | right) stay adjacent so both warp
        # directions of a scale run as ONE 2B launch
        # (no index tensors here: the step must stay capturable as a hipGraph)
        v = images[:, :, :3 * img_h].reshape(B, 3, 3, img_h, img_w)
        frames = torch.cat((v[:, :, 0], v[:, :, 2], v[:, :, 1]), 0)
        flows_lr = self._flows(imgl, img, imgr, frames)
        halves = [f.split(B) for f in flows_lr]
        optical_flows_bwd, optical_flows_fwd = [h[0] for h in halves], [h[1] for h in halves]   # centre->left, centre->right

        loss_pack = {}
        n = self.num_scales          # the reference also builds the unused 4th level
        if n <= 3 and img_h % 4 == 0 and img_w % 4 == 0:
            scales = (frames.detach(),) + ops.img_pyramid(frames)
            imglr_pyramid = [t[:2 * B] for t in scales[:n]]
            img_pyramid = [t[2 * B:] for t in scales[:n]]
        else:
            pl, pr = self.generate_img_pyramid(imgl, n), self.generate_img_pyramid(imgr, n)
            imglr_pyramid = [torch.cat((a, b), 0) for a, b in zip(pl, pr)]
            img_pyramid = self.generate_img_pyramid(img, n)

        # warp_flow_pyramid(imgl, flows_bwd) and (imgr, flows_fwd) (reference :221-222) as one 2B call per scale:
        # warped[s] = (img_from_l | img_from_r) [2B,3,h,w]
        warped = self.warp_flow_pyramid(imglr_pyramid, flows_lr)

        # Every loss of the reference's :224-234 is per sample and per direction; the two directions of a scale share the
        # centre image, so each kernel runs ONCE per scale over 2B samples (bwd | fwd) against the B centre images:
        # compute_diff_weight :224-225, compute_loss_with_mask x2 :226-227, compute_loss_ssim x2 :229-230,
        # compute_loss_flow_smooth x2 :232-233, compute_loss_flow_consis :235.  Sums over scales first, then fwd + bwd,
        # exactly the reference's association.
        pixel, ssim, smooth, consis = [], [], [], []
        fused_sums = self.fused_loss_sums and n <= 4
        # round 5: the second stage of the 4 x n per-sample reductions below is ONE launch in front of loss_combine (its only reader)
        # instead of one per reduction -- ops.deferred_loss_sums, same bits
        with (ops.deferred_loss_sums if (fused_sums and self.deferred_loss_sums and ops.on_device(images)) else contextlib.nullcontext()):
            one_launch_per_loss = self.multiscale_losses and ops.multiscale_supported(img_pyramid[:n], warped[:n])
            if one_launch_per_loss:
                pixel, ssim, smooth, consis = ops.multiscale_losses(img_pyramid[:n], warped[:n], flows_lr[:n])     # (the halves by offset: no split nodes)
            for s in (() if one_launch_per_loss else range(n)):
                diff, wgt = ops.occ_weight_stacked(img_pyramid[s], warped[s])       # (diff_bwd | diff_fwd), (weight_bwd | weight_fwd)
                pixel.append(ops.masked_mean(diff, wgt))
                ssim.append(ops.ssim_loss(img_pyramid[s], warped[s], wgt))
                smooth.append(ops.smooth2_loss(flows_lr[s], img_pyramid[s]))
                consis.append(ops.consis_loss(optical_flows_fwd[s], optical_flows_bwd[s], wgt[B:]))
            if fused_sums:
                # `loss = 0; loss += term(scale)` per loss, then fwd + bwd (:226-233), as one launch each way (ops.loss_combine)
                packed = ops.loss_combine(pixel, ssim, smooth, consis)
        if not fused_sums:
            packed = [sum(t[1:], t[0]) for t in (pixel, ssim, smooth, consis)]
            packed = [t[B:] + t[:B] for t in packed[:3]] + packed[3:]           # fwd + bwd (:226-227)
        loss_pack['loss_pixel'], loss_pack['loss_ssim'], loss_pack['loss_flow_smooth'], loss_pack['loss_flow_consis'] = packed

        return loss_pack
