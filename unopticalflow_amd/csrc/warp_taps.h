// Bilinear tap set-up shared by the warp kernels (warp.hip) and the fused warp + cost-volume kernels (warp_corr.hip).
//
// Bit-exact mask: the sample position is computed with the reference's exact fp32 operation
// sequence (no contraction except the one ATen's CPU kernel itself performs):
//   v  = x + u                       net_utils.py:39
//   g  = (2*v) / max(W-1,1) - 1      net_utils.py:42-43   (IEEE divide)
//   ix = fma(g+1, W/2, -0.5)         grid_sample, align_corners=False (ATen CPU contracts it)
//   ix = (g+1) * ((W-1)/2)           align_corners=True
//   w = ix-floor(ix), e = 1-w, n = iy-floor(iy), s = 1-n; taps nw=s*e ne=s*w sw=n*e se=n*w
//   mask = (((nw'+ne')+sw')+se') >= 0.9999f   with out-of-image taps' weights zeroed
#pragma once
#ifndef UNFLOW_HOST_CHECK        // (tests/host_check/ms_flat_check.cpp compiles this header with g++ behind a few one-line stand-ins)
#include "common.h"
#endif

namespace {

struct Taps {
    float nw, ne, sw, se;      // weights with out-of-range taps zeroed
    float w, e, n, s;          // raw fractional weights (for the flow gradient)
    int o_nw, o_ne, o_sw, o_se;  // element offsets inside one HxW plane (clamped, always valid)
    int xc0, xc1, yc0, yc1;      // the clamped tap coordinates those offsets were built from
    int x0, y0;                  // position of the nw tap before clamping to the image (itself limited to [-2, size+1])
    bool v_nw, v_ne, v_sw, v_se;
    bool mask;
};

__device__ __forceinline__ float unnormalise(float v, int size, int align_corners) {
    const float den = (float)(size > 1 ? size - 1 : 1);
    const float g = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, v), den), 1.0f);
    const float gp = __fadd_rn(g, 1.0f);
    if (align_corners) return __fmul_rn(gp, (float)(size - 1) * 0.5f);
    return fmaf(gp, (float)size * 0.5f, -0.5f);
}

__device__ __forceinline__ Taps make_taps(float u, float v, int x, int y, int H, int W, int ac) {
    Taps t;
    const float ix = unnormalise(__fadd_rn((float)x, u), W, ac);
    const float iy = unnormalise(__fadd_rn((float)y, v), H, ac);
    const float fx = floorf(ix), fy = floorf(iy);
    t.w = __fsub_rn(ix, fx);
    t.e = __fsub_rn(1.0f, t.w);
    t.n = __fsub_rn(iy, fy);
    t.s = __fsub_rn(1.0f, t.n);
    // clamp before the int conversion so wild flows cannot overflow
    const float cx = fminf(fmaxf(fx, -2.0f), (float)W + 1.0f);
    const float cy = fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
    const int x0 = (int)cx, y0 = (int)cy, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = (x0 >= 0 && x0 < W), vx1 = (x1 >= 0 && x1 < W);
    const bool vy0 = (y0 >= 0 && y0 < H), vy1 = (y1 >= 0 && y1 < H);
    const bool finite = (ix == ix) && (iy == iy);     // NaN flow: every tap dropped, as ATen does
    t.v_nw = vx0 && vy0 && finite; t.v_ne = vx1 && vy0 && finite;
    t.v_sw = vx0 && vy1 && finite; t.v_se = vx1 && vy1 && finite;
    t.nw = t.v_nw ? __fmul_rn(t.s, t.e) : 0.f;
    t.ne = t.v_ne ? __fmul_rn(t.s, t.w) : 0.f;
    t.sw = t.v_sw ? __fmul_rn(t.n, t.e) : 0.f;
    t.se = t.v_se ? __fmul_rn(t.n, t.w) : 0.f;
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y1, 0), H - 1);
    t.o_nw = yc0 * W + xc0; t.o_ne = yc0 * W + xc1;
    t.o_sw = yc1 * W + xc0; t.o_se = yc1 * W + xc1;
    t.xc0 = xc0; t.xc1 = xc1; t.yc0 = yc0; t.yc1 = yc1;
    t.x0 = x0; t.y0 = y0;
    const float m = __fadd_rn(__fadd_rn(__fadd_rn(t.nw, t.ne), t.sw), t.se);
    t.mask = (m >= 0.9999f);
    return t;
}

}  // namespace
