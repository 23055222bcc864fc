/*
 * unflow_hip.h -- C ABI of libunflow_hip.so, the MI355X (gfx950) kernels behind the
 * UnOpticalFlow `--mode flow` hot path.
 *
 * The reference (jianfenglihg/UnOpticalFlow) is pure Python on PyTorch and has no FFI;
 * its operator plug points for this path are Python callables.  Each entry point below
 * replaces the eager-PyTorch op chain of one of them (reference file:line cited per
 * function); the Python host side (unopticalflow_amd/ops.py) binds them with ctypes and
 * wraps them in torch.autograd.Function under the reference's own names.
 *
 * Conventions
 *   - every tensor is a dense fp32 NCHW device buffer (masks: uint8; the *_bf16 epilogue entries
 *     and unflow_prepare_triplets state their own element types), caller-allocated; kernels never
 *     allocate, free or synchronise;
 *   - `stream` is a hipStream_t (NULL = the default stream); work is only enqueued;
 *   - the return value is the hipError_t of the launch (0 = hipSuccess),
 *     UNFLOW_EINVAL (-22) for a bad argument (NULL pointer, non-positive size, d < 0);
 *   - `partials` scratch buffers hold per-workgroup partial sums of the per-sample
 *     reductions; size them with unflow_partials_per_sample(); results are bitwise
 *     reproducible run to run (fixed summation order, no float atomics) except
 *     unflow_warp_bwd's gsrc scatter-add.
 *   - align_corners selects the grid_sample generation: 0 = torch >= 1.3 default
 *     (the reference as it runs today), 1 = torch 1.2.0 (the reference's pin).
 */
#ifndef UNFLOW_HIP_H
#define UNFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNFLOW_EINVAL (-22)

/* ABI version of this header (bumped on any signature change).  The library returns the value it was BUILT with; a C user
 * compares `unflow_abi_version() == UNFLOW_ABI_VERSION` (tools/capi_bench.cpp), the Python binding reads this very line
 * (unopticalflow_amd/_lib.py). */
#define UNFLOW_ABI_VERSION 12
int unflow_abi_version(void);

/* ---- kernel-exact timing (bench.py's roofline legs; nothing in the reference corresponds) ----
 * unflow_timing_begin() arms a fresh HIP event pair for the CALLING THREAD and returns its slot id; until
 * unflow_timing_end(), the kernels this thread launches through the library carry the pair (hipExtLaunchKernelGGL): `start` is
 * stamped when the first of them begins, `stop` when the last one ends.  unflow_timing_elapsed_us(slot, &us) after the stream has
 * been synchronised; UNFLOW_EINVAL for a slot that saw no launch.  unflow_timing_reset() makes the slot ids start over (the event
 * pairs are kept and re-used); unflow_timing_reserve(n) creates n pairs ahead of a timed loop.  Do not arm while the stream is
 * being captured into a hipGraph. */
int unflow_timing_reserve(int n);
int unflow_timing_begin(void);
int unflow_timing_end(void);
int unflow_timing_elapsed_us(int slot, float* us);
int unflow_timing_reset(void);

/* Number of float slots per sample a `partials` buffer needs for an H x W map (K = 2 included;
 * multiply by B). */
int unflow_partials_per_sample(int H, int W);

/* ---- cost volume: PWC_tf.corr_naive, core/networks/structures/pwc_tf.py:97-106 ----
 * cv[b, i*(2d+1)+j, y, x] = (1/C) * sum_c f1[b,c,y,x] * f2[b,c,y+i-d,x+j-d]  (zero outside)
 * f1,f2: [B,C,H,W]; cv: [B,(2d+1)^2,H,W]. */
int unflow_corr_fwd(const float* f1, const float* f2, float* cv,
                    int B, int C, int H, int W, int d, void* stream);
/* autograd of the above: gcv [B,(2d+1)^2,H,W] -> gf1, gf2 [B,C,H,W] (both written in full). */
int unflow_corr_bwd(const float* f1, const float* f2, const float* gcv, float* gf1, float* gf2,
                    int B, int C, int H, int W, int d, void* stream);
/* (ABI 12) The same backward with the ARITHMETIC chosen per call -- the library keeps no process-wide mode (the setter of ABI 10-11,
 * unflow_corr_set_backward, is gone: two models in one process could not differ).  The reference has one arithmetic (ATen's fp32 sums,
 * pwc_tf.py:97-106 through autograd); every form below holds its 1e-4 bar.
 *   UNFLOW_CORR_BWD_AUTO   what unflow_corr_bwd() runs: per shape, the fastest kernel that has passed a complete GPU parity run;
 *   UNFLOW_CORR_BWD_FP32   fp32 FMA kernels everywhere (the results of ABI <= 9, bit for bit);
 *   UNFLOW_CORR_BWD_MFMA   banded products on the matrix cores (v_mfma_f32_16x16x32_bf16, both operands split into bf16 hi + lo parts, fp32
 *                          accumulation: ~4e-6 of the largest gradient away from the fp32 sums, deterministic) wherever the shape is served
 *                          (d = 4 or 8, >= 8192 pixels, C % 16 == 0, W % 4 == 0, H >= 4 d, 16-byte aligned tensors), the fp32 kernels elsewhere.
 *                          Difference a caller can see besides rounding: for gf2 the kernel multiplies weights read at displaced pixels
 *                          (neighbouring rows / planes of gcv) with exact-zero features outside the image, so an Inf or NaN anywhere in gcv
 *                          reaches border outputs that the fp32 kernels leave finite;
 *   UNFLOW_CORR_BWD_FP32_NEXT  fp32 FMA kernels, taking the kernels written in round 6 that no GPU has run yet where one exists (executed and checked
 *                          on the build host only): the small-map backward with the gradient rows passing through registers (csrc/corr_small_rows.h)
 *                          for maps of <= 1024 pixels at d = 8 -- where _AUTO runs one lane per output element -- and at d = 4.  Same sums in another
 *                          fp32 order, deterministic.  The candidate for _AUTO once measured.
 * UNFLOW_EINVAL for any other value. */
#define UNFLOW_CORR_BWD_AUTO 0
#define UNFLOW_CORR_BWD_FP32 1
#define UNFLOW_CORR_BWD_MFMA 2
#define UNFLOW_CORR_BWD_FP32_NEXT 3
int unflow_corr_bwd_ex(const float* f1, const float* f2, const float* gcv, float* gf1, float* gf2,
                       int B, int C, int H, int W, int d, int arithmetic, void* stream);

/* ---- flow warp: warp_flow, core/networks/structures/net_utils.py:16-54 ----
 * out[b,c,y,x] = bilinear sample of src[b,c] at (x+u, y+v) through the reference's
 * normalise -> grid_sample(bilinear, zeros) chain.  mask (may be NULL = use_mask False):
 * [B,1,H,W] uint8, 1 where the sampled ones-image is >= 0.9999 (net_utils.py:47-51); when
 * given, out is multiplied by it (net_utils.py:52).  flow: [B,2,H,W], channel 0 = x. */
int unflow_warp_fwd(const float* src, const float* flow, float* out, uint8_t* mask,
                    int B, int C, int H, int W, int align_corners, void* stream);
/* autograd of the above.  mask NULL = unmasked.  gsrc may be NULL (source has no grad, as for
 * the detached image pyramids, model_flow_paper.py:58); otherwise it is zeroed and
 * scatter-added with float atomics.  gflow [B,2,H,W] is written in full. */
int unflow_warp_bwd(const float* src, const float* flow, const float* gout, const uint8_t* mask,
                    float* gsrc, float* gflow,
                    int B, int C, int H, int W, int align_corners, void* stream);
/* The same backward with the source gradient computed as a GATHER (ABI 7): every workgroup owns a tile of gsrc, finds the
 * pixels whose taps land on it through a per-tile table of tap displacements (a pre-pass; gflow's memory is its scratch until
 * the flow gradient is written), and stores each element once -- no zero-fill, no float atomics, bitwise reproducible from
 * run to run.  Feature maps only (mask == NULL, >= 8 channels, >= 512 pixels); other shapes take unflow_warp_bwd's path.
 * Slower than the scatter form on MI355X (level 2: 51.6 vs 48.6 us), hence a separate entry point, not the default. */
int unflow_warp_bwd_det(const float* src, const float* flow, const float* gout, const uint8_t* mask,
                    float* gsrc, float* gflow,
                    int B, int C, int H, int W, int align_corners, void* stream);
/* (ABI 9) The whole backward of a feature-map warp in ONE pass over the upstream gradient (net_utils.py:16-46 through autograd): the
 * source gradient as a gather (as unflow_warp_bwd_det: no zero-fill, no float atomics, every element written once) AND the flow gradient,
 * both by the workgroup that owns the tile -- bitwise reproducible, the upstream gradient is read from HBM once.  `table`:
 * unflow_warp_bwd_table_bytes(B, C, H, W) bytes of scratch.  unflow_warp_bwd_fused_supported(): 0 = shape not served (fewer than 8
 * channels or 512 pixels: UNFLOW_EINVAL, use unflow_warp_bwd; the masked image warps have no source gradient), 1 = served, but fewer
 * than 256 workgroups of full-width tiles (levels 3-5 of the 832x256 step: the scatter form with channel groups measures faster
 * there), 2 = served and recommended (level 2; 448x1024 levels 2 and 3). */
int unflow_warp_bwd_fused_supported(int B, int C, int H, int W);
int unflow_warp_bwd_table_bytes(int B, int C, int H, int W);
int unflow_warp_bwd_fused(const float* src, const float* flow, const float* gout, float* gsrc, float* gflow, void* table,
                          int table_ready, int B, int C, int H, int W, int align_corners, void* stream);
/* The table is a by-product of the forward's tap set-up: unflow_warp_fwd_table == unflow_warp_fwd(mask = NULL) that also fills `table`
 * (only where unflow_warp_bwd_fused_supported() == 2), and unflow_warp_bwd_fused(table_ready = 1) for the SAME flow then skips its
 * pre-pass over the flow; table_ready = 0: the backward computes the table itself. */
int unflow_warp_fwd_table(const float* src, const float* flow, float* out, void* table,
                          int B, int C, int H, int W, int align_corners, void* stream);

/* ---- fused warp + cost volume: one decoder level of PWC_tf.forward, pwc_tf.py:121-122 (134-135, 146-147, 159-160) ----
 *   feat2_warped = self.warp(f2, flow);  cv = self.corr(f1, feat2_warped)
 * in one kernel: the warped map only ever exists as LDS tiles.  f1, f2: [B,C,H,W]; flow: [B,2,H,W] (the x2-upsampled
 * flow of the level above); cv: [B,(2d+1)^2,H,W].  Same numerics as unflow_warp_fwd followed by unflow_corr_fwd.
 * unflow_warp_corr_supported() tells whether a shape is covered (d = 4, W a multiple of 4, 16-byte aligned inputs);
 * otherwise the entry returns UNFLOW_EINVAL and the caller runs the two separate entry points. */
int unflow_warp_corr_supported(int C, int H, int W, int d);
int unflow_warp_corr_fwd(const float* f1, const float* f2, const float* flow, float* cv,
                         int B, int C, int H, int W, int d, int align_corners, void* stream);
/* autograd of the above without anything saved by the forward pass: the warped map is recomputed.  gcv
 * [B,(2d+1)^2,H,W] -> gf1 [B,C,H,W], gf2 [B,C,H,W] (zeroed, then scatter-added; may be NULL), gflow [B,2,H,W].
 * scratch: 2*B*C*H*W floats (warped map | gradient w.r.t. it), caller-owned.  Any shape unflow_corr_bwd accepts. */
int unflow_warp_corr_bwd(const float* f1, const float* f2, const float* flow, const float* gcv,
                         float* gf1, float* gf2, float* gflow, float* scratch,
                         int B, int C, int H, int W, int d, int align_corners, void* stream);

/* ---- occlusion weights: Model_flow.compute_diff_weight, model_flow_paper.py:101-134 ----
 * img, from_l, from_r: [B,3,H,W].  diff_*: mean_c|img-from_*| [B,1,H,W]; w_*: soft occlusion
 * weight * validity [B,1,H,W]; valid_* (may be NULL): uint8 1 - prod_c[from_* == 0]. */
int unflow_occ_weight_fwd(const float* img, const float* from_l, const float* from_r,
                          float* diff_l, float* diff_r, float* w_bwd, float* w_fwd,
                          uint8_t* valid_bwd, uint8_t* valid_fwd,
                          int B, int H, int W, void* stream);
/* grad of diff = mean_c|img-from| w.r.t. from: gfrom[b,c,p] = -sign(img-from) * gdiff[b,p] / 3.
 * img_batch: samples in `img` (B % img_batch == 0); sample b of `from` pairs with image b % img_batch -- the model
 * runs both warp directions of a pair as ONE launch of 2B samples over the B centre images (img_batch = B). */
int unflow_absdiff_bwd(const float* img, const float* from, const float* gdiff, float* gfrom,
                       int B, int H, int W, int img_batch, void* stream);

/* ---- the per-sample reductions below are two stages: partial sums per workgroup, then one workgroup per sample adds them in a
 * fixed order.  Round 5: a forward entry called with loss == NULL (masked mean, SSIM loss, smoothness, consistency) stops after the
 * first stage, and unflow_loss_finalize_batch finishes any number of them with ONE launch and the same bits (12 second-stage
 * launches per train step -> 1).  Job q: partials[q] as the entry filled them, nblk[q] = unflow_loss_partial_blocks(op, H, W, B,
 * aligned) (op 0 masked mean, 1 SSIM loss -- aligned: every tensor 8-byte aligned --, 2 smoothness, 3 consistency), B[q] samples,
 * kind[q] 0: loss = (s0/n0) / (s1/n1 + 1e-12), sums[q][b] = {s0, s1};  1: loss = (s0/n0 + s1/n1) / 2, sums[q] ignored (may be NULL).
 * n0 / n1: masked mean HW / HW, SSIM 3HW / HW, consistency 2HW / HW, smoothness 2 H (W-2) / 2 (H-2) W (as floats). */
int unflow_loss_partial_blocks(int op, int H, int W, int B, int aligned);
int unflow_loss_finalize_batch(const void* const* partials, void* const* loss, void* const* sums, const int* nblk,
                               const int* B, const int* kind, const float* n0, const float* n1, int njobs, void* stream);

/* ---- masked mean: Model_flow.compute_loss_with_mask (one scale), model_flow_paper.py:93-97 ----
 * loss[b] = mean_p(diff*w) / (mean_p(w) + 1e-12).  partials: K=2.  sums[b] = {sum diff*w, sum w}
 * is saved for the backward. */
int unflow_masked_mean_fwd(const float* diff, const float* w, float* loss, float* sums,
                           float* partials, int B, int H, int W, void* stream);
int unflow_masked_mean_bwd(const float* w, const float* sums, const float* gloss, float* gdiff,
                           int B, int H, int W, void* stream);

/* ---- SSIM loss: SSIM, pytorch_ssim/ssim.py:4-20 + compute_loss_ssim (one scale),
 * model_flow_paper.py:140-146 ----
 * x = img*w, y = warped*w; loss[b] = mean_{c,p} clamp((1-SSIM(x,y))/2, 0, 1) / (mean_p(w)+1e-12).
 * partials: K=2.  sums[b] = {sum clamp(..), sum w}. */
int unflow_ssim_loss_fwd(const float* img, const float* warped, const float* w, float* loss,
                         float* sums, float* partials, int B, int H, int W, int img_batch, void* stream);
/* gradient w.r.t. warped only (img is a detached pyramid, w is detached).  img_batch as for unflow_absdiff_bwd
 * (img: [img_batch,3,H,W]; warped, w: B samples). */
int unflow_ssim_loss_bwd(const float* img, const float* warped, const float* w, const float* sums,
                         const float* gloss, float* gwarped, int B, int H, int W, int img_batch, void* stream);
/* the bare SSIM map of ssim.py:4-20 for [B,C,H,W] inputs (test/diagnostic surface). */
int unflow_ssim_map(const float* x, const float* y, float* out, int B, int C, int H, int W,
                    void* stream);
/* autograd of unflow_ssim_map w.r.t. both arguments (the reference's SSIM(x, y) is an ordinary differentiable
 * function, pytorch_ssim/ssim.py:4-20): gmap [B,C,H,W] -> gx, gy [B,C,H,W] (either may be NULL).
 * scratch: 4*B*C*H*W floats, caller-owned. */
int unflow_ssim_map_bwd(const float* x, const float* y, const float* gmap, float* gx, float* gy,
                        float* scratch, int B, int C, int H, int W, void* stream);

/* ---- 2nd-order smoothness: cal_grad2_error + compute_loss_flow_smooth (one scale),
 * model_flow_paper.py:152-177 ----  flow [B,2,H,W] (un-divided; the /20 is inside), img [B,3,H,W].
 * partials: K=2. */
int unflow_smooth2_fwd(const float* flow, const float* img, float* loss, float* partials,
                       int B, int H, int W, int img_batch, void* stream);       /* img: [img_batch,3,H,W], see unflow_absdiff_bwd */
int unflow_smooth2_bwd(const float* flow, const float* img, const float* gloss, float* gflow,
                       int B, int H, int W, int img_batch, void* stream);

/* ---- forward/backward consistency: compute_loss_flow_consis (one scale),
 * model_flow_paper.py:44-51,183-193 ----  grad flows to fwd_flow only.  partials: K=2. */
int unflow_consis_fwd(const float* fwd_flow, const float* bwd_flow, const float* w_fwd, float* loss,
                      float* sums, float* partials, int B, int H, int W, void* stream);
int unflow_consis_bwd(const float* fwd_flow, const float* bwd_flow, const float* w_fwd,
                      const float* sums, const float* gloss, float* gflow,
                      int B, int H, int W, void* stream);

/* ---- the loss entries above, ONE launch over the n <= 4 scales of Model_flow.forward's loop (model_flow_paper.py:224-235) each
 * (ABI 11).  Every array argument is a HOST array with one entry per scale (device pointers / sizes of that scale); B and img_batch
 * are common to the scales.  The launch runs, per scale, exactly the workgroups of the single-scale entry with exactly its kernel
 * body -- same values bit for bit; what it saves is 2 of 3 launches per loss and direction (scales 1 and 2 hold 1/4 and 1/16 of the
 * pixels and sit on a 4-10 us floor each as launches of their own).  The FORWARD entries stop after the first stage of the
 * per-sample reductions, like the single-scale entries called with loss == NULL: finish with unflow_loss_finalize_batch
 * (nblk[q] = unflow_loss_partial_blocks(op, H, W, B, 1) at that scale).  -22 for what the single-scale entries serve through another
 * kernel -- B > 65535; H or W < 3 (smoothness backward); an odd width or a tensor that is not 8-byte aligned (SSIM pair): call
 * those per scale.
 * occ_weight: the stacked layout of the train step -- warped[k] = (from_l | from_r) [2B,3,H,W] against img[k] [B,3,H,W] ->
 * diff[k] = (diff_l | diff_r), wgt[k] = (w_bwd | w_fwd) [2B,1,H,W]; no validity masks. */
int unflow_occ_weight_fwd_ms(int n, const float* const* img, const float* const* warped, float* const* diff, float* const* wgt,
                             const int* H, const int* W, int B, void* stream);
int unflow_absdiff_bwd_ms(int n, const float* const* img, const float* const* from, const float* const* gdiff, float* const* gfrom,
                          const int* H, const int* W, int B, int img_batch, void* stream);
int unflow_masked_mean_fwd_ms(int n, const float* const* diff, const float* const* w, float* const* partials,
                              const int* H, const int* W, int B, void* stream);
int unflow_masked_mean_bwd_ms(int n, const float* const* w, const float* const* sums, const float* const* gloss, float* const* gdiff,
                              const int* H, const int* W, int B, void* stream);
int unflow_ssim_loss_fwd_ms(int n, const float* const* img, const float* const* warped, const float* const* w, float* const* partials,
                            const int* H, const int* W, int B, int img_batch, void* stream);
int unflow_ssim_loss_bwd_ms(int n, const float* const* img, const float* const* warped, const float* const* w, const float* const* sums,
                            const float* const* gloss, float* const* gwarped, const int* H, const int* W, int B, int img_batch,
                            void* stream);
int unflow_smooth2_fwd_ms(int n, const float* const* flow, const float* const* img, float* const* partials,
                          const int* H, const int* W, int B, int img_batch, void* stream);
int unflow_smooth2_bwd_ms(int n, const float* const* flow, const float* const* img, const float* const* gloss, float* const* gflow,
                          const int* H, const int* W, int B, int img_batch, void* stream);
int unflow_consis_fwd_ms(int n, const float* const* fwd_flow, const float* const* bwd_flow, const float* const* w_fwd,
                         float* const* partials, const int* H, const int* W, int B, void* stream);
int unflow_consis_bwd_ms(int n, const float* const* fwd_flow, const float* const* bwd_flow, const float* const* w_fwd,
                         const float* const* sums, const float* const* gloss, float* const* gflow, const int* H, const int* W,
                         int B, void* stream);
/* the masked image warps of Model_flow.warp_flow_pyramid (model_flow_paper.py:62-66, net_utils.py:47-52), one launch over the scales
 * each way: src[k] [B,C,H,W] with C <= 4, flow[k] [B,2,H,W] -> out[k] = warp * mask[k], mask[k] [B,1,H,W] uint8 (required); the backward
 * writes the flow gradient only (unflow_warp_bwd with gsrc = NULL: the image pyramids carry no gradient). */
int unflow_warp_fwd_ms(int n, const float* const* src, const float* const* flow, float* const* out, uint8_t* const* mask,
                       const int* H, const int* W, int B, int C, int align_corners, void* stream);
int unflow_warp_bwd_ms(int n, const float* const* src, const float* const* flow, const float* const* gout, const uint8_t* const* mask,
                       float* const* gflow, const int* H, const int* W, int B, int C, int align_corners, void* stream);

/* ---- conv() epilogue: Conv2d bias + LeakyReLU(0.1), core/networks/structures/net_utils.py:7-11 ----
 * y [N,C,H,W] is a bias-free convolution output, updated in place: y = leaky_relu(y + bias[c]). */
int unflow_bias_leaky_fwd(float* y, const float* bias, int N, int C, int H, int W, float slope,
                          void* stream);
/* floats of `partials` scratch the backward needs. */
int unflow_bias_leaky_partials(int N, int C, int H, int W);
/* y: the activation output; gin = gout * (y > 0 ? 1 : slope) (may alias gout); gbias[c] = sum gin,
 * reduced in a fixed order (bitwise reproducible). */
int unflow_bias_leaky_bwd(const float* y, const float* gout, float* gin, float* gbias, float* partials,
                          int N, int C, int H, int W, float slope, void* stream);

/* The same with the gradients of up to two consumers of the activation added on the fly
 * (gin = (gout + gout2) * ...; gout2 may be NULL), each possibly a channel slice of a wider NCHW tensor
 * -- what autograd hands out for a torch.cat operand (pwc_tf.py:113-117: every decoder activation feeds
 * the next conv and a cat): sample stride in elements (>= C*H*W), the (C,H,W) block of a sample dense.
 * Replaces the separate gradient-accumulation pass and the .contiguous() copy of a sliced gradient. */
int unflow_bias_leaky_bwd2(const float* y, const float* gout, long long gout_stride, const float* gout2,
                           long long gout2_stride, float* gin, float* gbias, float* partials,
                           int N, int C, int H, int W, float slope, void* stream);

/* Channels-last twins (the fp32 conv stacks run in torch.channels_last: MIOpen's implicit-GEMM solvers are NHWC
 * kernels and need no transposes then): y / gin dense [P][C], P = N*H*W pixels, C a multiple of 4 (<= 1024);
 * gout / gout2 may be channel slices of wider NHWC tensors -- pixel stride in elements (>= C, multiple of 4).
 * Same arithmetic and the same fixed-order bias-gradient reduction as the NCHW entries. */
int unflow_bias_leaky_fwd_nhwc(float* y, const float* bias, long long P, int C, float slope, void* stream);
int unflow_bias_leaky_partials_nhwc(long long P, int C);
int unflow_bias_leaky_bwd2_nhwc(const float* y, const float* gout, long long gout_pstride, const float* gout2,
                                long long gout2_pstride, float* gin, float* gbias, float* partials,
                                long long P, int C, float slope, void* stream);

/* Layout glue at the borders of the channels_last conv stacks (pwc_tf.py:113 `torch.cat((corr, c1, up_flow), 1)` is the
 * decoder input): out [B][HW][Ca+Cb+Cc] (NHWC) = the channel concatenation of up to three dense NCHW tensors (Cb / Cc may
 * be 0), and its inverse (destinations that are NULL are skipped: gradients nobody needs).  With one tensor these are
 * the NCHW <-> NHWC transposes. */
int unflow_cat_nhwc(const float* a, int Ca, const float* b, int Cb, const float* c, int Cc, float* out,
                    int B, int HW, void* stream);
int unflow_split_nhwc(const float* in, float* a, int Ca, float* b, int Cb, float* c, int Cc, int B, int HW, void* stream);
/* (ABI 8) the same two with the NHWC side in bf16 (raw uint16_t; the bf16 conv-stack option): the NCHW planes stay fp32, one
 * round-to-nearest-even per element into the NHWC tensor, exact widening out of it. */
int unflow_cat_nhwc_bf16(const float* a, int Ca, const float* b, int Cb, const float* c, int Cc, uint16_t* out,
                         int B, int HW, void* stream);
int unflow_split_nhwc_bf16(const uint16_t* in, float* a, int Ca, float* b, int Cb, float* c, int Cc, int B, int HW, void* stream);

/* bf16 activations (the bf16 conv-stack option: torch.autocast around the reference's conv() blocks): y, gout,
 * gout2, gin are bf16 (raw uint16_t), bias / gbias / partials fp32; arithmetic in fp32, one round-to-nearest-even
 * per element; gbias sums the rounded gin values.  Same scratch size (unflow_bias_leaky_partials). */
int unflow_bias_leaky_fwd_bf16(uint16_t* y, const float* bias, int N, int C, int H, int W, float slope,
                               void* stream);
int unflow_bias_leaky_bwd2_bf16(const uint16_t* y, const uint16_t* gout, long long gout_stride,
                                const uint16_t* gout2, long long gout2_stride, uint16_t* gin, float* gbias,
                                float* partials, int N, int C, int H, int W, float slope, void* stream);

/* ... and their channels_last twins ([P][C] bf16, C % 4 == 0; scratch size: unflow_bias_leaky_partials_nhwc). */
int unflow_bias_leaky_fwd_nhwc_bf16(uint16_t* y, const float* bias, long long P, int C, float slope, void* stream);
int unflow_bias_leaky_bwd2_nhwc_bf16(const uint16_t* y, const uint16_t* gout, long long gout_pstride, const uint16_t* gout2,
                                     long long gout2_pstride, uint16_t* gin, float* gbias, float* partials,
                                     long long P, int C, float slope, void* stream);

/* Epilogues that FILL the decoder's cat buffers (ABI 7).  The reference's decoder feeds every convolution the
 * torch.cat of the two previous activations (pwc_tf.py:114-118: x = cat((conv0(x), x)) ... per level); here the epilogue
 * of the convolution that produces an activation writes it straight into the channel slices of the (at most two) NHWC
 * buffers the next convolutions read, so the cat copies do not exist:
 *   dst1[p * dst1_pstride + c] = dst2[...] = leaky(y[p * C + c] + bias[c])     dst2 may be NULL; dst1 may be y (in place)
 * and the backward reads the activation (for its sign) from such a slice (act, act_pstride) and its two upstream
 * gradients from the slices of the buffers' gradients (as unflow_bias_leaky_bwd2_nhwc).  Pixel strides in elements,
 * multiples of 4; fp32 pointers 16-byte, bf16 pointers 8-byte aligned. */
int unflow_bias_leaky_fwd_nhwc_to(const float* y, const float* bias, long long P, int C, float slope, float* dst1,
                                  long long dst1_pstride, float* dst2, long long dst2_pstride, void* stream);
int unflow_bias_leaky_bwd2_nhwc_from(const float* act, long long act_pstride, const float* gout, long long gout_pstride,
                                     const float* gout2, long long gout2_pstride, float* gin, float* gbias,
                                     float* partials, long long P, int C, float slope, void* stream);
int unflow_bias_leaky_fwd_nhwc_to_bf16(const uint16_t* y, const float* bias, long long P, int C, float slope, uint16_t* dst1,
                                       long long dst1_pstride, uint16_t* dst2, long long dst2_pstride, void* stream);
int unflow_bias_leaky_bwd2_nhwc_from_bf16(const uint16_t* act, long long act_pstride, const uint16_t* gout, long long gout_pstride,
                                          const uint16_t* gout2, long long gout2_pstride, uint16_t* gin, float* gbias,
                                          float* partials, long long P, int C, float slope, void* stream);

/* (ABI 8) The pyramid hand-off with the last `dup` samples written twice: in [Bin][HW][C] (NHWC; fp32 / bf16) -> out fp32 NCHW
 * [Bin + dup][C][HW], samples Bin .. Bin + dup - 1 = samples Bin - dup .. Bin - 1 (the centre frame's features are the first input of
 * both decoder directions: no torch.cat((c, c))); and the gradient's way back, g fp32 NCHW [Bout + dup][C][HW] -> out NHWC
 * [Bout][HW][C] with g[Bout + i] added to sample Bout - dup + i. */
int unflow_to_nchw_dup(const float* in, float* out, int C, int Bin, int dup, int HW, void* stream);
int unflow_to_nchw_dup_bf16(const uint16_t* in, float* out, int C, int Bin, int dup, int HW, void* stream);
int unflow_to_nhwc_fold(const float* g, float* out, int C, int Bout, int dup, int HW, void* stream);
int unflow_to_nhwc_fold_bf16(const float* g, uint16_t* out, int C, int Bout, int dup, int HW, void* stream);

/* ---- flow heads (ABI 8): predict_flow (pwc_tf.py:93-94, Conv2d(c, 2, 3, bias=True), no activation) and the residual that follows
 * it (`flow = predict_flow(x) + up_flow`, :130,143,155,167; `flow2 + dc_conv7(x)`, :171) at the border of the channels_last conv
 * stack.  y [N*HW][2]: the bias-free convolution output (NHWC; fp32, or bf16 as raw uint16_t); res (may be NULL), out, g: fp32
 * NCHW [N][2][HW];  out = y + bias (+ res).  Backward: gy = g re-laid out (rounded once for bf16), gbias[2] = sum g (fixed
 * summation order); the residual's gradient is g itself.  partials: unflow_flow_head_partials() floats of scratch. */
int unflow_flow_head_partials(void);
int unflow_flow_head_fwd(const float* y, const float* bias, const float* res, float* out, int N, int HW, void* stream);
int unflow_flow_head_fwd_bf16(const uint16_t* y, const float* bias, const float* res, float* out, int N, int HW, void* stream);
int unflow_flow_head_bwd(const float* g, float* gy, float* gbias, float* partials, int N, int HW, void* stream);
int unflow_flow_head_bwd_bf16(const float* g, uint16_t* gy, float* gbias, float* partials, int N, int HW, void* stream);

/* ---- deferred bias-gradient reduction (ABI 9).  Every backward entry point above that takes `gbias` (unflow_bias_leaky_bwd*,
 * unflow_flow_head_bwd*) finishes with a small second-stage launch that adds its per-workgroup partial sums into gbias in a
 * fixed order.  A backward pass of the flow network (43 conv() blocks net_utils.py:7-11 + the predict_flow heads
 * pwc_tf.py:93-94) paid 49 of those launches, ~5 us each.  Since ABI 9, gbias == NULL makes such an entry point stop after
 * its first stage (the partial sums stay in `partials`, which the caller keeps alive), and ONE call of this function
 * finishes all pending reductions with one launch (per 56 jobs), each channel summed in exactly the order the entry point's
 * own second stage uses -- the same bits.
 *   partials[j], gbias[j]   device pointers of job j (HOST arrays of njobs entries)
 *   n[j]                    partial sums per channel: N * ceil(H*W / 4096) for the NCHW entries (unflow_bias_leaky_partials / C),
 *                           ceil(P / 128) for the NHWC ones (unflow_bias_leaky_partials_nhwc / C), ceil(N*HW / 256) capped at 512
 *                           for a flow head
 *   C[j]                    channels (2 for a flow head)
 *   mode[j]                 0: fp32 conv epilogue (NCHW or NHWC), 1: bf16 conv epilogue, 2: flow head */
int unflow_bias_grad_finalize_batch(const void* const* partials, void* const* gbias, const int* n, const int* C,
                                    const int* mode, int njobs, void* stream);

/* ---- Adam (ABI 9): torch.optim.Adam(lr, betas, eps) of the reference's train step (train.py:39,151; no weight decay, no amsgrad) over
 * ALL parameter tensors in one launch.  A block updates one chunk of unflow_adam_chunk() elements of one tensor:
 *   m = m + (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * with t = steps[0] + 1 (the arithmetic of torch's fused Adam).
 *   slots      DEVICE array [ntensors]: parameter / exp_avg / exp_avg_sq addresses (walked by memory offset: all three and the
 *              gradient must share one dense layout) and the element count
 *   chunk_map  DEVICE int pairs [nchunks][2] = (tensor, chunk of that tensor)
 *   grads      HOST array [ntensors] of the gradients' device addresses (they change with every eager backward pass; passed to the
 *              kernel by value); NULL: that tensor is skipped and its counter does not advance
 *   steps      DEVICE float [ntensors]: the step counters (all equal when every tensor has a gradient), advanced by a one-block
 *              launch behind the update */
typedef struct { float* p; float* m; float* v; long long numel; } unflow_adam_slot;
int unflow_adam_chunk(void);
int unflow_adam_multi(const unflow_adam_slot* slots, const int* chunk_map, int nchunks, const void* const* grads, int ntensors,
                      float* steps, float lr, float beta1, float beta2, float eps, void* stream);

/* ---- loss bookkeeping (ABI 8).  Model_flow.forward sums every per-sample loss over the scales (`loss = 0; loss += term(scale)`,
 * model_flow_paper.py:92-99,140-148,171-177,183-195) and adds the two directions (:226-233); train.py:147-150 weights the four batch
 * means.  One launch each way per stage, same association of the fp32 additions.
 *   terms   HOST array of 4 * n_scales device pointers, [loss][scale]; losses 0..2 (pixel, ssim, smooth) are [2B] vectors
 *           (bwd half | fwd half), loss 3 (consis) is [B];  outs / gouts: HOST arrays of 4 device pointers, [B] each (a NULL
 *           gradient pointer counts as zeros);  gin: 7 * B floats = [3][2B] then [B] -- the gradient of every scale's term of
 *           a loss is the same vector.
 *   unflow_weighted_mean_sum_*: loss = sum_k weights[k] * mean_b(terms[k][b]), K <= 8; weights is a HOST array. */
int unflow_loss_combine_fwd(const float* const* terms, int n_scales, int B, float* const* outs, void* stream);
int unflow_loss_combine_bwd(const float* const* gouts, int B, float* gin, void* stream);
int unflow_weighted_mean_sum_fwd(const float* const* terms, const float* weights, int K, int B, float* loss, void* stream);
int unflow_weighted_mean_sum_bwd(const float* gloss, const float* weights, int K, int B, float* const* grads, void* stream);

/* ---- flow up-sampling of the decoder (ABI 8): out = mul * F.interpolate(x, bilinear, align_corners=False) for an integer
 * up-sampling factor per axis (Ho % Hi == 0, Wo % Wi == 0) -- pwc_tf.py:119,131,144,156 (factor 2, mul 2.0) and :174-177
 * (F.interpolate(flow * 4.0, size): the factor 4.0 commutes exactly).  x [planes,Hi,Wi] -> out [planes,Ho,Wo]; the backward is
 * a gather over the output pixels that read an input pixel (fixed order, bitwise reproducible). */
int unflow_upsample_scaled_fwd(const float* x, float* out, int planes, int Hi, int Wi, int Ho, int Wo, float mul, void* stream);
int unflow_upsample_scaled_bwd(const float* gout, float* gin, int planes, int Hi, int Wi, int Ho, int Wo, float mul, void* stream);

/* ---- image pyramid: Model_flow.generate_img_pyramid scales 1 and 2, model_flow_paper.py:54-60 ----
 * img [planes,H,W] (planes = B*C, any leading layout) -> half [planes,H/2,W/2] (2x2 box means) and
 * quarter [planes,H/4,W/4] (4x4 box means of img).  H, W multiples of 4. */
int unflow_img_pyramid(const float* img, float* half, float* quarter, int planes, int H, int W,
                       void* stream);

/* ---- input stage: KITTI_Prepared.__getitem__ after the PNG decode, core/dataset/kitti_prepared.py:63-90,145-148 ----
 * src: B decoded stacked triplets, uint8, 3 interleaved channels, back to back in one device buffer;
 * offsets[b]: first byte of image b; dims[2b], dims[2b+1]: its rows (three frames of int(rows/3) rows)
 * and columns; flip[b] != 0: horizontal flip (cv2.flip(img, 1)), NULL = none.  dst [B,3,3H,W] fp32 =
 * each frame resized to HxW with OpenCV's 8-bit INTER_LINEAR fixed-point arithmetic, / 255.0.
 * swap_rb != 0 writes source channel c to plane 2-c (RGB-decoded PNG -> the BGR planes cv2.imread gives).
 * W multiple of 4.  offsets, dims, flip are device pointers. */
int unflow_prepare_triplets(const uint8_t* src, const long long* offsets, const int* dims,
                            const uint8_t* flip, float* dst, int B, int H, int W, int swap_rb,
                            void* stream);

/* ---- host helper of the evaluation path (KITTI 16-bit flow PNGs, core/evaluation/flowlib.py:107-127) ----
 * rows: height x (1 + stride) host bytes (filter byte + filtered scanline per row), decoded in place.
 * Runs on the CPU; no GPU work. */
int unflow_png_unfilter(uint8_t* rows, int height, int stride, int bpp);

#ifdef __cplusplus
}
#endif
#endif /* UNFLOW_HIP_H */
