// Occlusion-aware photometric terms for gfx950: occlusion weights, masked mean, and the flow
// regularisers (2nd-order smoothness, forward/backward consistency).
//
// Each kernel replaces a chain of ~10-20 eager elementwise / reduction launches of the
// reference (core/networks/model_flow_paper.py, lines cited per kernel) with one pass over the
// pixels: coalesced reads along x, per-sample sums reduced with wave shuffles -> LDS -> one
// partial per workgroup -> fixed-order finalize (bitwise reproducible, no float atomics).
// Built with -ffp-contract=off: the elementwise algebra follows the reference op by op.
#include "common.h"
#include "multiscale.h"
#include <stdlib.h>

namespace {

constexpr int TILE = UNFLOW_RED_TILE;

// ---------------------------------------------------------------------------------------------
// compute_diff_weight, model_flow_paper.py:108-132 (one scale)
// ---------------------------------------------------------------------------------------------
__global__ void occ_weight_fwd_kernel(const float* __restrict__ img, const float* __restrict__ from_l,
                                      const float* __restrict__ from_r, float* __restrict__ diff_l,
                                      float* __restrict__ diff_r, float* __restrict__ w_bwd,
                                      float* __restrict__ w_fwd, uint8_t* __restrict__ valid_bwd,
                                      uint8_t* __restrict__ valid_fwd, int B, int HW) {
#include "bodies/occ_weight_fwd.inc"
}

// d mean_c|img - from| / d from  (torch: abs' = sign, with sign(0) = 0)
// img_b: samples in `img`; sample b of `from` is compared with image b % img_b (both warp directions of a pair share
// the centre image, so the two directions run as one 2B launch)
__global__ void absdiff_bwd_kernel(const float* __restrict__ img, const float* __restrict__ from,
                                   const float* __restrict__ gdiff, float* __restrict__ gfrom, int B, int HW, int img_b) {
#include "bodies/absdiff_bwd.inc"
}


// ---------------------------------------------------------------------------------------------
// compute_loss_with_mask, model_flow_paper.py:93-97 (one scale)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masked_mean_partial_kernel(const float* __restrict__ diff,
                                                                  const float* __restrict__ w,
                                                                  float* __restrict__ partials, int HW) {
#include "bodies/masked_mean_partial.inc"
}

// generic finalize for loss = (s0 / n0) / (s1 / n1 + 1e-12)
__global__ void ratio_finalize_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ loss,
                                      float* __restrict__ sums, float n0, float n1) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* p = partials + (size_t)b * nblk * 2;
    const float s0 = sum_partials(p, nblk, 2, 0, red);
    const float s1 = sum_partials(p, nblk, 2, 1, red);
    if (threadIdx.x == 0) {
        loss[b] = (s0 / n0) / (s1 / n1 + 1e-12f);
        if (sums) { sums[b * 2] = s0; sums[b * 2 + 1] = s1; }
    }
}

__global__ void masked_mean_bwd_kernel(const float* __restrict__ w, const float* __restrict__ sums,
                                       const float* __restrict__ gloss, float* __restrict__ gdiff,
                                       int B, int HW) {
#include "bodies/masked_mean_bwd.inc"
}


// ---------------------------------------------------------------------------------------------
// cal_grad2_error / compute_loss_flow_smooth, model_flow_paper.py:152-177 (one scale)
//   err = ( mean_{2,H,W-2}( wx[x+1] * |f[x+2]-2f[x+1]+f[x]| ) + mean_{2,H-2,W}( wy[y+1] * |...| ) ) / 2
//   wx[x] = exp(-10 * mean_c|img[x+1]-img[x]|), f = flow / 20
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float edge_w(const float* __restrict__ im, size_t hw, size_t i0, size_t i1) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) s = s + fabsf(im[c * hw + i1] - im[c * hw + i0]);
    return expf(-10.0f * (s / 3.0f));
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void smooth2_partial_kernel(const float* __restrict__ flow,
                                                              const float* __restrict__ img,
                                                              float* __restrict__ partials, int H, int W, int img_b) {
    __shared__ float red[8];
    const int b = blockIdx.y;
    const int HW = H * W;
    const float* f = flow + (size_t)b * 2 * HW;
    const float* im = img + (size_t)(b % img_b) * 3 * HW;
    float acc[2] = {0.f, 0.f};
    const int p0 = blockIdx.x * TILE;
#pragma unroll 2
    for (int k = 0; k < TILE / 256; ++k) {
        const int p = p0 + k * 256 + threadIdx.x;
        if (p >= HW) break;
        const int y = p / W, x = p - y * W;
        if (x + 2 < W) {                                 // dx2 at x, weight w_x[x+1]
            const float wx = edge_w(im, HW, p + 1, p + 2);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float a = f[c * HW + p] / 20.0f, m = f[c * HW + p + 1] / 20.0f, z = f[c * HW + p + 2] / 20.0f;
                acc[0] += wx * fabsf((z - m) - (m - a));
            }
        }
        if (y + 2 < H) {
            const float wy = edge_w(im, HW, p + W, p + 2 * W);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float a = f[c * HW + p] / 20.0f, m = f[c * HW + p + W] / 20.0f, z = f[c * HW + p + 2 * W] / 20.0f;
                acc[1] += wy * fabsf((z - m) - (m - a));
            }
        }
    }
    block_sum_256<2>(acc, red);
    if (threadIdx.x == 0) {
        float* o = partials + ((size_t)b * gridDim.x + blockIdx.x) * 2;
        o[0] = acc[0]; o[1] = acc[1];
    }
}

__global__ void smooth2_finalize_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ loss,
                                        int H, int W) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* p = partials + (size_t)b * nblk * 2;
    const float s0 = sum_partials(p, nblk, 2, 0, red);
    const float s1 = sum_partials(p, nblk, 2, 1, red);
    if (threadIdx.x == 0) {
        const float nx = 2.0f * (float)H * (float)(W - 2), ny = 2.0f * (float)(H - 2) * (float)W;
        loss[b] = (s0 / nx + s1 / ny) / 2.0f;     // empty means (W<3 or H<3) are NaN, as in torch
    }
}

// Backward through LDS tiles (round 3).  With S_x[c][q] = wx(q) * sgn(dx2_c(q)) (q = where a second difference starts) the gradient is
//   g_c[p] = kx * (S_x[c][p-2] - 2 S_x[c][p-1] + S_x[c][p]) + ky * (the same down the column),
// so every S -- one exp, six image and six flow loads -- is worth computing ONCE per position instead of once per pixel that uses it
// (three times in each direction in the per-pixel form below: 36.5 us for 75 MB at scale 0).  A 256-thread workgroup owns 64 x 8
// pixels: S_x on 8 x 66 positions, S_y on 10 x 64, as float2 (both flow channels) in LDS.
constexpr int SM_TW = 64, SM_TH = 8;
__global__ __launch_bounds__(256) void smooth2_bwd_tile_kernel(const float* __restrict__ flow, const float* __restrict__ img,
                                                               const float* __restrict__ gloss, float* __restrict__ gflow,
                                                               int H, int W, int img_b) {
    constexpr int NX = SM_TH * (SM_TW + 2), NY = (SM_TH + 2) * SM_TW;
    __shared__ float2 s_x[NX], s_y[NY];
    const int b = blockIdx.z, x0 = blockIdx.x * SM_TW, y0 = blockIdx.y * SM_TH;
    const int HW = H * W;
    const float* f = flow + (size_t)b * 2 * HW;
    const float* im = img + (size_t)(b % img_b) * 3 * HW;
    for (int i = threadIdx.x; i < NX + NY; i += 256) {
        float2 v = make_float2(0.f, 0.f);
        if (i < NX) {
            const int r = i / (SM_TW + 2), c = i - r * (SM_TW + 2);
            const int y = y0 + r, x = x0 - 2 + c;                    // dx2 starting at x, weight w_x[x+1]
            if (y < H && x >= 0 && x + 2 < W) {
                const int q = y * W + x;
                const float w = edge_w(im, HW, q + 1, q + 2);
                const float a0 = f[q] / 20.0f, m0 = f[q + 1] / 20.0f, z0 = f[q + 2] / 20.0f;
                const float a1 = f[HW + q] / 20.0f, m1 = f[HW + q + 1] / 20.0f, z1 = f[HW + q + 2] / 20.0f;
                v = make_float2(w * sgn((z0 - m0) - (m0 - a0)), w * sgn((z1 - m1) - (m1 - a1)));
            }
            s_x[i] = v;
        } else {
            const int j = i - NX;
            const int r = j / SM_TW, c = j - r * SM_TW;
            const int y = y0 - 2 + r, x = x0 + c;
            if (x < W && y >= 0 && y + 2 < H) {
                const int q = y * W + x;
                const float w = edge_w(im, HW, q + W, q + 2 * W);
                const float a0 = f[q] / 20.0f, m0 = f[q + W] / 20.0f, z0 = f[q + 2 * W] / 20.0f;
                const float a1 = f[HW + q] / 20.0f, m1 = f[HW + q + W] / 20.0f, z1 = f[HW + q + 2 * W] / 20.0f;
                v = make_float2(w * sgn((z0 - m0) - (m0 - a0)), w * sgn((z1 - m1) - (m1 - a1)));
            }
            s_y[j] = v;
        }
    }
    __syncthreads();
    const float kx = gloss[b] / (2.0f * (2.0f * (float)H * (float)(W - 2))) / 20.0f;
    const float ky = gloss[b] / (2.0f * (2.0f * (float)(H - 2) * (float)W)) / 20.0f;
#pragma unroll
    for (int k = 0; k < SM_TW * SM_TH / 256; ++k) {
        const int idx = k * 256 + (int)threadIdx.x;
        const int r = idx / SM_TW, c = idx - r * SM_TW;
        const int y = y0 + r, x = x0 + c;
        if (x >= W || y >= H) continue;
        const float2 xa = s_x[r * (SM_TW + 2) + c], xb = s_x[r * (SM_TW + 2) + c + 1], xc = s_x[r * (SM_TW + 2) + c + 2];      // starts x-2, x-1, x
        const float2 ya = s_y[r * SM_TW + c], yb = s_y[(r + 1) * SM_TW + c], yc = s_y[(r + 2) * SM_TW + c];                    // starts y-2, y-1, y
        const size_t o = (size_t)b * 2 * HW + (size_t)y * W + x;
        gflow[o] = kx * ((xa.x - 2.f * xb.x) + xc.x) + ky * ((ya.x - 2.f * yb.x) + yc.x);
        gflow[o + HW] = kx * ((xa.y - 2.f * xb.y) + xc.y) + ky * ((ya.y - 2.f * yb.y) + yc.y);
    }
}

// Round 4: both directions through ONE staged tile.  The per-pixel forward re-read every flow value six times and divided it by 20
// each time (12 IEEE divisions and 18 loads per pixel: 38.8 us for 47.7 MB at scale 0, 0.15 of the HBM roofline); round 3's
// backward computed its S values from global memory with the same redundancy (44.5 us, 0.21).  Here a workgroup stages its
// 64 x 8 pixels plus a 2-pixel ring ONCE: flow / 20 (both channels) and the three image planes, 68 x 12 positions, 16 KB of LDS;
// every second difference and every edge weight is then computed from LDS, with the operations of the per-pixel forms in the same
// order (same values; only the grouping of the forward's partial sums changes).
constexpr int SG_W = SM_TW + 4, SG_H = SM_TH + 4, SG_N = SG_W * SG_H;
struct SmoothStage { float f0[SG_N], f1[SG_N], i0[SG_N], i1[SG_N], i2[SG_N]; };

__device__ __forceinline__ void smooth_stage(SmoothStage& t, const float* __restrict__ f, const float* __restrict__ im, int HW,
                                             int H, int W, int x0, int y0) {
    for (int i = threadIdx.x; i < SG_N; i += 256) {
        const int ry = i / SG_W, rx = i - ry * SG_W;
        const int y = y0 - 2 + ry, x = x0 - 2 + rx;
        float a = 0.f, b = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const int q = y * W + x;
            a = f[q] / 20.0f; b = f[HW + q] / 20.0f;
            c0 = im[q]; c1 = im[HW + q]; c2 = im[2 * HW + q];
        }
        t.f0[i] = a; t.f1[i] = b; t.i0[i] = c0; t.i1[i] = c1; t.i2[i] = c2;
    }
}

// edge_w() between two staged positions
__device__ __forceinline__ float stage_edge_w(const SmoothStage& t, int j0, int j1) {
    float s = 0.f;
    s = s + fabsf(t.i0[j1] - t.i0[j0]);
    s = s + fabsf(t.i1[j1] - t.i1[j0]);
    s = s + fabsf(t.i2[j1] - t.i2[j0]);
    return expf(-10.0f * (s / 3.0f));
}

// grid = (tiles_x, tiles_y, B); partials[(b * tiles + tile) * 2] = {sum over the tile of wx |dx2|, of wy |dy2|}
__global__ __launch_bounds__(256) void smooth2_fwd_tile_kernel(const float* __restrict__ flow, const float* __restrict__ img,
                                                               float* __restrict__ partials, int H, int W, int img_b) {
#include "bodies/smooth2_fwd_tile.inc"
}

// the backward of smooth2_bwd_tile_kernel with its S values computed from the staged tile
__global__ __launch_bounds__(256) void smooth2_bwd_stage_kernel(const float* __restrict__ flow, const float* __restrict__ img,
                                                                const float* __restrict__ gloss, float* __restrict__ gflow,
                                                                int H, int W, int img_b) {
#include "bodies/smooth2_bwd_stage.inc"
}

// gather form of the backward: pixel q collects the three second differences it takes part in.
__global__ void smooth2_bwd_kernel(const float* __restrict__ flow, const float* __restrict__ img,
                                   const float* __restrict__ gloss, float* __restrict__ gflow,
                                   int B, int H, int W, int img_b) {
    const int HW = H * W;
    const size_t n = (size_t)B * HW;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(t / HW), p = (int)(t - (size_t)b * HW);
        const int y = p / W, x = p - y * W;
        const float* f = flow + (size_t)b * 2 * HW;
        const float* im = img + (size_t)(b % img_b) * 3 * HW;
        const float kx = gloss[b] / (2.0f * (2.0f * (float)H * (float)(W - 2))) / 20.0f;
        const float ky = gloss[b] / (2.0f * (2.0f * (float)(H - 2) * (float)W)) / 20.0f;
        float g[2] = {0.f, 0.f};
        // x direction: second difference starting at x0 = x-2, x-1, x with coefficient +1, -2, +1
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int x0 = x - 2 + k;
            if (x0 < 0 || x0 + 2 >= W) continue;
            const int q = p - 2 + k;
            const float wx = edge_w(im, HW, q + 1, q + 2);
            const float coef = (k == 1) ? -2.f : 1.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float a = f[c * HW + q] / 20.0f, m = f[c * HW + q + 1] / 20.0f, z = f[c * HW + q + 2] / 20.0f;
                g[c] += kx * wx * coef * sgn((z - m) - (m - a));
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int y0 = y - 2 + k;
            if (y0 < 0 || y0 + 2 >= H) continue;
            const int q = p + (k - 2) * W;
            const float wy = edge_w(im, HW, q + W, q + 2 * W);
            const float coef = (k == 1) ? -2.f : 1.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float a = f[c * HW + q] / 20.0f, m = f[c * HW + q + W] / 20.0f, z = f[c * HW + q + 2 * W] / 20.0f;
                g[c] += ky * wy * coef * sgn((z - m) - (m - a));
            }
        }
        gflow[(size_t)b * 2 * HW + p] = g[0];
        gflow[(size_t)b * 2 * HW + HW + p] = g[1];
    }
}

// ---------------------------------------------------------------------------------------------
// get_flow_normalization + compute_loss_flow_consis, model_flow_paper.py:44-51,183-193 (one scale)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void consis_partial_kernel(const float* __restrict__ ff,
                                                             const float* __restrict__ fb,
                                                             const float* __restrict__ w_fwd,
                                                             float* __restrict__ partials, int HW) {
#include "bodies/consis_partial.inc"
}

__global__ void consis_bwd_kernel(const float* __restrict__ ff, const float* __restrict__ fb,
                                  const float* __restrict__ w_fwd, const float* __restrict__ sums,
                                  const float* __restrict__ gloss, float* __restrict__ gflow, int B, int HW) {
#include "bodies/consis_bwd.inc"
}

// ---------------------------------------------------------------------------------------------
// Loss bookkeeping of Model_flow.forward (model_flow_paper.py:224-235) and of the step (train.py:147-150).  Every term is a
// [B]- or [2B]-vector per scale; summing them over scales, adding the two directions and weighting the four batch means are
// ~50 launches of 4-5 us in eager ops (adds, slices and their zero-fill + copy + add backward, means, scalar multiplies).
// Here: one launch each way for each of the two stages, same association of the fp32 additions as the reference.
// ---------------------------------------------------------------------------------------------
constexpr int LOSS_MAX_SCALES = 4;
struct LossTerms { const float* t[4][LOSS_MAX_SCALES]; };     // [pixel, ssim, smooth: (bwd | fwd) [2B]; consis: [B]][scale]
struct LossOuts { float* o[4]; };
struct LossGrads { const float* g[4]; };

__global__ void loss_combine_fwd_kernel(LossTerms in, LossOuts out, int n, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float bwd = 0.f, fwd = 0.f;                     // loss = 0; loss = loss + term(scale) (:92-99 ...): 0 + x is exact
        for (int s = 0; s < n; ++s) { bwd += in.t[k][s][b]; fwd += in.t[k][s][B + b]; }
        out.o[k][b] = fwd + bwd;                        // loss_with_mask(fwd) + loss_with_mask(bwd) (:226-233)
    }
    float c = 0.f;
    for (int s = 0; s < n; ++s) c += in.t[3][s][b];
    out.o[3][b] = c;
}

// d(out_k[b]) / d(term_k[s][b]) = d / d(term_k[s][B + b]) = 1 for every scale: one (bwd | fwd) gradient vector per loss serves all scales
__global__ void loss_combine_bwd_kernel(LossGrads g, float* __restrict__ gin, int B) {      // gin: [3][2B] then [B]
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = g.g[k] ? g.g[k][b] : 0.f;
        gin[k * 2 * B + b] = v;
        gin[k * 2 * B + B + b] = v;
    }
    gin[6 * B + b] = g.g[3] ? g.g[3][b] : 0.f;
}

struct MeanTerms { const float* t[8]; float w[8]; };
struct MeanGrads { float* g[8]; float w[8]; };

// loss = sum_k w_k * mean_b(t_k[b]) (train.py:147-150), keys in order; one workgroup, lanes along the batch
__global__ __launch_bounds__(256) void weighted_mean_sum_fwd_kernel(MeanTerms in, int K, int B, float* __restrict__ loss) {
    __shared__ float red[4];
    float total = 0.f;
    for (int k = 0; k < K; ++k) {
        const float s = sum_partials(in.t[k], B, 1, 0, red);
        total += in.w[k] * (s / (float)B);
    }
    if (threadIdx.x == 0) *loss = total;
}

__global__ void weighted_mean_sum_bwd_kernel(MeanGrads out, int K, int B, const float* __restrict__ gloss) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float g = *gloss / (float)B;
    for (int k = 0; k < K; ++k) out.g[k][b] = out.w[k] * g;
}

// ---------------------------------------------------------------------------------------------
// One launch over the scales (csrc/multiscale.h): the bodies above once more, behind a table of per-scale arguments.
// ---------------------------------------------------------------------------------------------
#include "ms_flat_photo.h"      // occ_weight_fwd, absdiff_bwd, masked_mean_bwd, consis_bwd (also compiled for the host by the tests)

struct MeanMsArgs { const float *diff, *w; float* partials; int HW; };
__global__ __launch_bounds__(256) void masked_mean_partial_ms_kernel(MsTable<MeanMsArgs> ms_table_) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    const float* __restrict__ diff = ms_a_.diff; const float* __restrict__ w = ms_a_.w; float* __restrict__ partials = ms_a_.partials;
    const int HW = ms_a_.HW;
#include "bodies/masked_mean_partial.inc"
}

struct SmoothMsArgs { const float *flow, *img; float* partials; int H, W; };
__global__ __launch_bounds__(256) void smooth2_fwd_tile_ms_kernel(MsTable<SmoothMsArgs> ms_table_, int img_b) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    const float* __restrict__ flow = ms_a_.flow; const float* __restrict__ img = ms_a_.img; float* __restrict__ partials = ms_a_.partials;
    const int H = ms_a_.H, W = ms_a_.W;
#include "bodies/smooth2_fwd_tile.inc"
}

struct SmoothBwdMsArgs { const float *flow, *img, *gloss; float* gflow; int H, W; };
__global__ __launch_bounds__(256) void smooth2_bwd_stage_ms_kernel(MsTable<SmoothBwdMsArgs> ms_table_, int img_b) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    const float* __restrict__ flow = ms_a_.flow; const float* __restrict__ img = ms_a_.img; const float* __restrict__ gloss = ms_a_.gloss;
    float* __restrict__ gflow = ms_a_.gflow;
    const int H = ms_a_.H, W = ms_a_.W;
#include "bodies/smooth2_bwd_stage.inc"
}

struct ConsisMsArgs { const float *ff, *fb, *w_fwd; float* partials; int HW; };
__global__ __launch_bounds__(256) void consis_partial_ms_kernel(MsTable<ConsisMsArgs> ms_table_) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    const float* __restrict__ ff = ms_a_.ff; const float* __restrict__ fb = ms_a_.fb; const float* __restrict__ w_fwd = ms_a_.w_fwd;
    float* __restrict__ partials = ms_a_.partials;
    const int HW = ms_a_.HW;
#include "bodies/consis_partial.inc"
}

inline int flat_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

}  // namespace

int unflow_ssim_blocks(int H, int W);   // ssim.hip

extern "C" int unflow_abi_version(void) { return UNFLOW_ABI_VERSION; }   // 2: + unflow_warp_corr_*; 3: img_batch on the 2B loss entries; 4: + *_nhwc epilogues; 5: + unflow_timing_*; 6: + *_nhwc_bf16 epilogues; 7: + *_nhwc_to / *_nhwc_from (epilogues that fill cat buffers), unflow_warp_bwd_det; 8: + unflow_upsample_scaled_*, *_nhwc_bf16 glue, unflow_loss_combine_*, unflow_weighted_mean_sum_*, unflow_flow_head_*

// ---- kernel-exact timing slots (see UNFLOW_LAUNCH in common.h) ----
#include <mutex>
#include <vector>
namespace {
struct TimingSlot { hipEvent_t start, stop; bool used; };
std::mutex g_timing_mutex;
std::vector<TimingSlot> g_timing_slots;
thread_local UnflowTimingArm t_arm = {nullptr, nullptr, false};
thread_local int t_slot = -1;
}  // namespace
UnflowTimingArm& unflow_timing_arm() { return t_arm; }

// event pairs are created once and re-used after unflow_timing_reset(): arming a slot inside a timed loop costs no hipEventCreate
static int timing_grow(size_t n) {                   // (caller holds the mutex)
    while (g_timing_slots.size() < n) {
        TimingSlot sl = {nullptr, nullptr, false};
        if (hipEventCreate(&sl.start) != hipSuccess || hipEventCreate(&sl.stop) != hipSuccess) return UNFLOW_EINVAL;
        g_timing_slots.push_back(sl);
    }
    return 0;
}
static size_t g_timing_next = 0;

extern "C" int unflow_timing_reserve(int n) {
    if (n < 0) return UNFLOW_EINVAL;
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    return timing_grow((size_t)n);
}

extern "C" int unflow_timing_begin(void) {
    if (t_arm.stop) return UNFLOW_EINVAL;            // already armed
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    if (timing_grow(g_timing_next + 1) != 0) return UNFLOW_EINVAL;
    t_slot = (int)g_timing_next++;
    g_timing_slots[t_slot].used = false;
    t_arm = {g_timing_slots[t_slot].start, g_timing_slots[t_slot].stop, false};
    return t_slot;
}

extern "C" int unflow_timing_end(void) {
    if (!t_arm.stop) return UNFLOW_EINVAL;
    {
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        g_timing_slots[t_slot].used = t_arm.started;
    }
    t_arm = {nullptr, nullptr, false};
    t_slot = -1;
    return 0;
}

extern "C" int unflow_timing_elapsed_us(int slot, float* us) {
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    if (!us || slot < 0 || slot >= (int)g_timing_next || !g_timing_slots[slot].used) return UNFLOW_EINVAL;
    float ms = 0.f;
    const hipError_t e = hipEventElapsedTime(&ms, g_timing_slots[slot].start, g_timing_slots[slot].stop);
    if (e != hipSuccess) return (int)e;
    *us = ms * 1000.f;
    return 0;
}

extern "C" int unflow_timing_reset(void) {
    if (t_arm.stop) return UNFLOW_EINVAL;
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    g_timing_next = 0;                               // slot ids start over; the event pairs are kept for re-use
    return 0;
}

// terms: HOST array of 4 * n_scales device pointers, [loss][scale]: pixel, ssim, smooth ([2B] each: bwd half | fwd half), consis ([B]);
// outs: HOST array of 4 device pointers, [B] each
extern "C" int unflow_loss_combine_fwd(const float* const* terms, int n_scales, int B, float* const* outs, void* stream) {
    UNFLOW_REQUIRE(terms && outs && n_scales > 0 && n_scales <= LOSS_MAX_SCALES && B > 0);
    LossTerms in = {};
    LossOuts out;
    for (int k = 0; k < 4; ++k) {
        UNFLOW_REQUIRE(outs[k]);
        out.o[k] = outs[k];
        for (int s = 0; s < n_scales; ++s) { UNFLOW_REQUIRE(terms[k * n_scales + s]); in.t[k][s] = terms[k * n_scales + s]; }
    }
    UNFLOW_LAUNCH(loss_combine_fwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, in, out, n_scales, B);
    return unflow_launch_status();
}

// gouts: HOST array of 4 device pointers ([B] each; NULL = no gradient for that loss); gin: [3][2B] followed by [B] (7 * B floats)
extern "C" int unflow_loss_combine_bwd(const float* const* gouts, int B, float* gin, void* stream) {
    UNFLOW_REQUIRE(gouts && gin && B > 0);
    LossGrads g;
    for (int k = 0; k < 4; ++k) g.g[k] = gouts[k];
    UNFLOW_LAUNCH(loss_combine_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, g, gin, B);
    return unflow_launch_status();
}

// terms / grads: HOST arrays of K device pointers ([B] each), weights: HOST array of K floats; loss, gloss: one device float
extern "C" int unflow_weighted_mean_sum_fwd(const float* const* terms, const float* weights, int K, int B, float* loss, void* stream) {
    UNFLOW_REQUIRE(terms && weights && loss && K > 0 && K <= 8 && B > 0);
    MeanTerms in = {};
    for (int k = 0; k < K; ++k) { UNFLOW_REQUIRE(terms[k]); in.t[k] = terms[k]; in.w[k] = weights[k]; }
    UNFLOW_LAUNCH(weighted_mean_sum_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, K, B, loss);
    return unflow_launch_status();
}

extern "C" int unflow_weighted_mean_sum_bwd(const float* gloss, const float* weights, int K, int B, float* const* grads, void* stream) {
    UNFLOW_REQUIRE(gloss && weights && grads && K > 0 && K <= 8 && B > 0);
    MeanGrads out = {};
    for (int k = 0; k < K; ++k) { UNFLOW_REQUIRE(grads[k]); out.g[k] = grads[k]; out.w[k] = weights[k]; }
    UNFLOW_LAUNCH(weighted_mean_sum_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, out, K, B, gloss);
    return unflow_launch_status();
}

extern "C" int unflow_partials_per_sample(int H, int W) {
    if (H <= 0 || W <= 0) return UNFLOW_EINVAL;
    const int flat = ceil_div(H * W, TILE);
    const int ss = unflow_ssim_blocks(H, W);
    const int sm = ceil_div(W, SM_TW) * ceil_div(H, SM_TH);              // smooth2_fwd_tile_kernel: one partial pair per 64 x 8 tile
    const int m = flat > ss ? flat : ss;
    return 2 * (m > sm ? m : sm);     // floats per sample (K = 2 everywhere)
}

// ---- the second stage of several per-sample reductions in ONE launch (round 5): a forward entry called with loss == NULL stops after
// its partial sums; this finishes up to LOSS_JOBS of them -- a block = one (job, sample), summed exactly like the entry's own second
// stage (same sum_partials, same divisions): same bits.  kind 0: loss = (s0 / n0) / (s1 / n1 + 1e-12), sums = {s0, s1} (masked mean,
// SSIM loss, consistency); kind 1: loss = (s0 / n0 + s1 / n1) / 2 (second-order smoothness).
constexpr int LOSS_JOBS = 16;
struct LossJob { const float* partials; float* loss; float* sums; int nblk, B, kind; float n0, n1; };
struct LossBatch { LossJob job[LOSS_JOBS]; };
__global__ __launch_bounds__(256) void loss_finalize_batch_kernel(LossBatch batch, int njobs) {
    __shared__ float red[4];
    int b = blockIdx.x, k = 0;
    while (k < njobs - 1 && b >= batch.job[k].B) { b -= batch.job[k].B; ++k; }      // (block-uniform)
    const LossJob j = batch.job[k];
    const float* p = j.partials + (size_t)b * j.nblk * 2;
    const float s0 = sum_partials(p, j.nblk, 2, 0, red);
    const float s1 = sum_partials(p, j.nblk, 2, 1, red);
    if (threadIdx.x == 0) {
        if (j.kind == 0) {
            j.loss[b] = (s0 / j.n0) / (s1 / j.n1 + 1e-12f);
            if (j.sums) { j.sums[b * 2] = s0; j.sums[b * 2 + 1] = s1; }
        } else {
            j.loss[b] = (s0 / j.n0 + s1 / j.n1) / 2.0f;
        }
    }
}

int unflow_ssim_loss_blocks(int H, int W, int fast);      // ssim.hip

extern "C" int unflow_loss_partial_blocks(int op, int H, int W, int B, int aligned) {
    if (H <= 0 || W <= 0 || B <= 0) return UNFLOW_EINVAL;
    switch (op) {
        case 0: case 3: return ceil_div(H * W, TILE);                                   // masked mean, consistency
        case 1: return unflow_ssim_loss_blocks(H, W, aligned);                          // SSIM loss
        case 2:
#ifdef UNFLOW_TUNING
            if (getenv("UNFLOW_SMOOTH_OLD") != nullptr) return ceil_div(H * W, TILE);
#endif
            return B <= 65535 ? ceil_div(W, SM_TW) * ceil_div(H, SM_TH) : ceil_div(H * W, TILE);    // smoothness
        default: return UNFLOW_EINVAL;
    }
}

extern "C" int unflow_loss_finalize_batch(const void* const* partials, void* const* loss, void* const* sums, const int* nblk,
                                          const int* B, const int* kind, const float* n0, const float* n1, int njobs, void* stream) {
    UNFLOW_REQUIRE(partials && loss && sums && nblk && B && kind && n0 && n1 && njobs > 0);
    hipStream_t s = (hipStream_t)stream;
    for (int first = 0; first < njobs; first += LOSS_JOBS) {
        LossBatch batch;
        const int n = njobs - first < LOSS_JOBS ? njobs - first : LOSS_JOBS;
        int blocks = 0;
        for (int k = 0; k < n; ++k) {
            const int q = first + k;
            UNFLOW_REQUIRE(partials[q] && loss[q] && nblk[q] > 0 && B[q] > 0 && (kind[q] == 0 || kind[q] == 1));
            batch.job[k] = LossJob{(const float*)partials[q], (float*)loss[q], (float*)sums[q], nblk[q], B[q], kind[q], n0[q], n1[q]};
            blocks += B[q];
        }
        UNFLOW_LAUNCH(loss_finalize_batch_kernel, dim3(blocks), dim3(256), 0, s, batch, n);
    }
    return unflow_launch_status();
}

extern "C" int unflow_occ_weight_fwd(const float* img, const float* from_l, const float* from_r,
                                     float* diff_l, float* diff_r, float* w_bwd, float* w_fwd,
                                     uint8_t* valid_bwd, uint8_t* valid_fwd, int B, int H, int W, void* stream) {
    UNFLOW_REQUIRE(img && from_l && from_r && diff_l && diff_r && w_bwd && w_fwd && B > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    UNFLOW_LAUNCH(occ_weight_fwd_kernel, dim3(flat_blocks((size_t)B * H * W)), dim3(256), 0, s, img, from_l,
                       from_r, diff_l, diff_r, w_bwd, w_fwd, valid_bwd, valid_fwd, B, H * W);
    return unflow_launch_status();
}

extern "C" int unflow_absdiff_bwd(const float* img, const float* from, const float* gdiff, float* gfrom,
                                  int B, int H, int W, int img_batch, void* stream) {
    UNFLOW_REQUIRE(img && from && gdiff && gfrom && B > 0 && H > 0 && W > 0 && img_batch > 0 && B % img_batch == 0);
    hipStream_t s = (hipStream_t)stream;
    UNFLOW_LAUNCH(absdiff_bwd_kernel, dim3(flat_blocks((size_t)B * 3 * H * W)), dim3(256), 0, s, img, from,
                       gdiff, gfrom, B, H * W, img_batch);
    return unflow_launch_status();
}

extern "C" int unflow_masked_mean_fwd(const float* diff, const float* w, float* loss, float* sums,
                                      float* partials, int B, int H, int W, void* stream) {
    UNFLOW_REQUIRE(diff && w && (loss == nullptr || sums) && partials && B > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = ceil_div(H * W, TILE);
    UNFLOW_LAUNCH(masked_mean_partial_kernel, dim3(nblk, B), dim3(256), 0, s, diff, w, partials, H * W);
    if (loss)                                            // (NULL: the caller finishes the sums later, unflow_loss_finalize_batch)
        UNFLOW_LAUNCH(ratio_finalize_kernel, dim3(B), dim3(256), 0, s, partials, nblk, loss, sums,
                           (float)H * (float)W, (float)H * (float)W);
    return unflow_launch_status();
}

extern "C" int unflow_masked_mean_bwd(const float* w, const float* sums, const float* gloss, float* gdiff,
                                      int B, int H, int W, void* stream) {
    UNFLOW_REQUIRE(w && sums && gloss && gdiff && B > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    UNFLOW_LAUNCH(masked_mean_bwd_kernel, dim3(flat_blocks((size_t)B * H * W)), dim3(256), 0, s, w, sums,
                       gloss, gdiff, B, H * W);
    return unflow_launch_status();
}

extern "C" int unflow_smooth2_fwd(const float* flow, const float* img, float* loss, float* partials,
                                  int B, int H, int W, int img_batch, void* stream) {
    UNFLOW_REQUIRE(flow && img && partials && B > 0 && H > 0 && W > 0 && img_batch > 0 && B % img_batch == 0);
    hipStream_t s = (hipStream_t)stream;
#ifdef UNFLOW_TUNING
    const bool per_pixel = getenv("UNFLOW_SMOOTH_OLD") != nullptr;      // A/B against the per-pixel form (tools/probes/loss_kernel_times.py)
#else
    const bool per_pixel = false;
#endif
    if (!per_pixel && B <= 65535) {
        const dim3 grid(ceil_div(W, SM_TW), ceil_div(H, SM_TH), B);
        UNFLOW_LAUNCH(smooth2_fwd_tile_kernel, grid, dim3(256), 0, s, flow, img, partials, H, W, img_batch);
        if (loss) UNFLOW_LAUNCH(smooth2_finalize_kernel, dim3(B), dim3(256), 0, s, partials, (int)(grid.x * grid.y), loss, H, W);
        return unflow_launch_status();
    }
    const int nblk = ceil_div(H * W, TILE);
    UNFLOW_LAUNCH(smooth2_partial_kernel, dim3(nblk, B), dim3(256), 0, s, flow, img, partials, H, W, img_batch);
    if (loss) UNFLOW_LAUNCH(smooth2_finalize_kernel, dim3(B), dim3(256), 0, s, partials, nblk, loss, H, W);
    return unflow_launch_status();
}

extern "C" int unflow_smooth2_bwd(const float* flow, const float* img, const float* gloss, float* gflow,
                                  int B, int H, int W, int img_batch, void* stream) {
    UNFLOW_REQUIRE(flow && img && gloss && gflow && B > 0 && H > 0 && W > 0 && img_batch > 0 && B % img_batch == 0);
    hipStream_t s = (hipStream_t)stream;
#ifdef UNFLOW_TUNING
    const bool per_pixel = getenv("UNFLOW_SMOOTH_OLD") != nullptr;      // A/B against the per-pixel form (tools/probes/loss_kernel_times.py)
#else
    const bool per_pixel = false;
#endif
    if (!per_pixel && B <= 65535 && H >= 3 && W >= 3) {
#ifdef UNFLOW_TUNING
        if (getenv("UNFLOW_SMOOTH_R3") != nullptr) {                     // round 3's tile kernel (S values from global memory)
            UNFLOW_LAUNCH(smooth2_bwd_tile_kernel, dim3(ceil_div(W, SM_TW), ceil_div(H, SM_TH), B), dim3(256), 0, s, flow, img, gloss, gflow,
                               H, W, img_batch);
            return unflow_launch_status();
        }
#endif
        UNFLOW_LAUNCH(smooth2_bwd_stage_kernel, dim3(ceil_div(W, SM_TW), ceil_div(H, SM_TH), B), dim3(256), 0, s, flow, img, gloss, gflow,
                           H, W, img_batch);
        return unflow_launch_status();
    }
    UNFLOW_LAUNCH(smooth2_bwd_kernel, dim3(flat_blocks((size_t)B * H * W)), dim3(256), 0, s, flow, img, gloss,
                       gflow, B, H, W, img_batch);
    return unflow_launch_status();
}

extern "C" int unflow_consis_fwd(const float* fwd_flow, const float* bwd_flow, const float* w_fwd, float* loss,
                                 float* sums, float* partials, int B, int H, int W, void* stream) {
    UNFLOW_REQUIRE(fwd_flow && bwd_flow && w_fwd && (loss == nullptr || sums) && partials && B > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = ceil_div(H * W, TILE);
    UNFLOW_LAUNCH(consis_partial_kernel, dim3(nblk, B), dim3(256), 0, s, fwd_flow, bwd_flow, w_fwd, partials,
                       H * W);
    if (loss)
        UNFLOW_LAUNCH(ratio_finalize_kernel, dim3(B), dim3(256), 0, s, partials, nblk, loss, sums,
                           2.0f * (float)H * (float)W, (float)H * (float)W);
    return unflow_launch_status();
}

extern "C" int unflow_consis_bwd(const float* fwd_flow, const float* bwd_flow, const float* w_fwd,
                                 const float* sums, const float* gloss, float* gflow, int B, int H, int W,
                                 void* stream) {
    UNFLOW_REQUIRE(fwd_flow && bwd_flow && w_fwd && sums && gloss && gflow && B > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    UNFLOW_LAUNCH(consis_bwd_kernel, dim3(flat_blocks((size_t)B * H * W)), dim3(256), 0, s, fwd_flow, bwd_flow,
                       w_fwd, sums, gloss, gflow, B, H * W);
    return unflow_launch_status();
}

// ---------------------------------------------------------------------------------------------
// The loss entries above, ONE launch for n <= 4 scales each (csrc/multiscale.h).  Every array argument is a HOST array with one
// entry per scale; B / img_batch are common to the scales.  The forward entries stop after the first stage of the per-sample
// reductions (what the single-scale entries do with loss == NULL): unflow_loss_finalize_batch finishes them, with
// unflow_loss_partial_blocks partial sums per sample at a scale -- the same grids, the same partial sums and the same bits as n
// single-scale calls.  -22 for shapes the single-scale entries serve by another kernel (B > 65535, H or W < 3 in the smoothness
// backward, odd W / unaligned tensors in the SSIM pair): call those per scale.
// ---------------------------------------------------------------------------------------------
#define UNFLOW_MS_REQUIRE_N(n) UNFLOW_REQUIRE((n) > 0 && (n) <= MS_MAX)

#include "ms_flat_photo_entries.h"     // unflow_occ_weight_fwd_ms, unflow_absdiff_bwd_ms, unflow_masked_mean_bwd_ms, unflow_consis_bwd_ms

extern "C" int unflow_masked_mean_fwd_ms(int n, const float* const* diff, const float* const* w, float* const* partials,
                                         const int* H, const int* W, int B, void* stream) {
    UNFLOW_REQUIRE(diff && w && partials && H && W && B > 0 && B <= 65535);
    UNFLOW_MS_REQUIRE_N(n);
    MsTable<MeanMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(diff[k] && w[k] && partials[k] && H[k] > 0 && W[k] > 0);
        t.a[k] = MeanMsArgs{diff[k], w[k], partials[k], H[k] * W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(ceil_div(H[k] * W[k], TILE), B)));
    }
    UNFLOW_LAUNCH(masked_mean_partial_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(256), 0, (hipStream_t)stream, t);
    return unflow_launch_status();
}

extern "C" int unflow_smooth2_fwd_ms(int n, const float* const* flow, const float* const* img, float* const* partials,
                                     const int* H, const int* W, int B, int img_batch, void* stream) {
    UNFLOW_REQUIRE(flow && img && partials && H && W && B > 0 && B <= 65535 && img_batch > 0 && B % img_batch == 0);
    UNFLOW_MS_REQUIRE_N(n);
    MsTable<SmoothMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(flow[k] && img[k] && partials[k] && H[k] > 0 && W[k] > 0);
        t.a[k] = SmoothMsArgs{flow[k], img[k], partials[k], H[k], W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(ceil_div(W[k], SM_TW), ceil_div(H[k], SM_TH), B)));
    }
    UNFLOW_LAUNCH(smooth2_fwd_tile_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(256), 0, (hipStream_t)stream, t, img_batch);
    return unflow_launch_status();
}

extern "C" int unflow_smooth2_bwd_ms(int n, const float* const* flow, const float* const* img, const float* const* gloss,
                                     float* const* gflow, const int* H, const int* W, int B, int img_batch, void* stream) {
    UNFLOW_REQUIRE(flow && img && gloss && gflow && H && W && B > 0 && B <= 65535 && img_batch > 0 && B % img_batch == 0);
    UNFLOW_MS_REQUIRE_N(n);
    MsTable<SmoothBwdMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(flow[k] && img[k] && gloss[k] && gflow[k] && H[k] >= 3 && W[k] >= 3);
        t.a[k] = SmoothBwdMsArgs{flow[k], img[k], gloss[k], gflow[k], H[k], W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(ceil_div(W[k], SM_TW), ceil_div(H[k], SM_TH), B)));
    }
    UNFLOW_LAUNCH(smooth2_bwd_stage_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(256), 0, (hipStream_t)stream, t, img_batch);
    return unflow_launch_status();
}

extern "C" int unflow_consis_fwd_ms(int n, const float* const* fwd_flow, const float* const* bwd_flow, const float* const* w_fwd,
                                    float* const* partials, const int* H, const int* W, int B, void* stream) {
    UNFLOW_REQUIRE(fwd_flow && bwd_flow && w_fwd && partials && H && W && B > 0 && B <= 65535);
    UNFLOW_MS_REQUIRE_N(n);
    MsTable<ConsisMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(fwd_flow[k] && bwd_flow[k] && w_fwd[k] && partials[k] && H[k] > 0 && W[k] > 0);
        t.a[k] = ConsisMsArgs{fwd_flow[k], bwd_flow[k], w_fwd[k], partials[k], H[k] * W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(ceil_div(H[k] * W[k], TILE), B)));
    }
    UNFLOW_LAUNCH(consis_partial_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(256), 0, (hipStream_t)stream, t);
    return unflow_launch_status();
}

