// Bilinear flow warp (+ binary validity mask) forward / backward for gfx950.
//
// Replaces warp_flow (reference core/networks/structures/net_utils.py:16-54): the CPU meshgrid +
// H2D copy, add, 2x normalise, permute, grid_sample (and, with use_mask, a second grid_sample of
// a CPU-built ones tensor plus two masked fills and a multiply) become one kernel; the
// backward replaces grid_sampler_2d_backward and the chain through the normalisation.
//
// Bit-exact mask: the sample position is computed with the reference's exact fp32 operation
// sequence (no contraction except the one ATen's CPU kernel itself performs):
//   v  = x + u                       net_utils.py:39
//   g  = (2*v) / max(W-1,1) - 1      net_utils.py:42-43   (IEEE divide)
//   ix = fma(g+1, W/2, -0.5)         grid_sample, align_corners=False (ATen CPU contracts it)
//   ix = (g+1) * ((W-1)/2)           align_corners=True
//   w = ix-floor(ix), e = 1-w, n = iy-floor(iy), s = 1-n; taps nw=s*e ne=s*w sw=n*e se=n*w
//   mask = (((nw'+ne')+sw')+se') >= 0.9999f   with out-of-image taps' weights zeroed
//
// Parallelisation: lanes run along x (coalesced flow reads / out writes; source gathers are
// near-coalesced because flow is smooth); threadIdx.y strides the channel loop so small pyramid
// levels still fill the chip, and the per-pixel tap set-up is shared by a lane's channels.
#include "common.h"

namespace {

struct Taps {
    float nw, ne, sw, se;      // weights with out-of-range taps zeroed
    float w, e, n, s;          // raw fractional weights (for the flow gradient)
    int o_nw, o_ne, o_sw, o_se;  // element offsets inside one HxW plane (clamped, always valid)
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
    const float m = __fadd_rn(__fadd_rn(__fadd_rn(t.nw, t.ne), t.sw), t.se);
    t.mask = (m >= 0.9999f);
    return t;
}

// block = (64, NY): x lanes x channel phases.  grid = (ceil(W/64), H, B).
template <int NY, bool MASKED>
__global__ void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                float* __restrict__ out, uint8_t* __restrict__ mask,
                                int C, int H, int W, int ac) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= W) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    const float u = flow[((size_t)b * 2) * plane + pix];
    const float v = flow[((size_t)b * 2 + 1) * plane + pix];
    const Taps t = make_taps(u, v, x, y, H, W, ac);
    const float keep = (!MASKED || t.mask) ? 1.f : 0.f;
    if (MASKED && threadIdx.y == 0) mask[(size_t)b * plane + pix] = t.mask ? 1 : 0;
    const float* sp = src + (size_t)b * C * plane;
    float* op = out + (size_t)b * C * plane + pix;
#pragma unroll 4
    for (int c = threadIdx.y; c < C; c += NY) {
        const float* p = sp + (size_t)c * plane;
        // same accumulation order as ATen: nw, ne, sw, se
        float r = p[t.o_nw] * t.nw;
        r = fmaf(p[t.o_ne], t.ne, r);
        r = fmaf(p[t.o_sw], t.sw, r);
        r = fmaf(p[t.o_se], t.se, r);
        op[(size_t)c * plane] = r * keep;
    }
}

// gsrc (optional) is scatter-added; gflow is reduced over the channel phases through LDS and
// written once per pixel (no atomics, reproducible).
template <int NY, bool MASKED, bool WITH_GSRC>
__global__ void warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                const float* __restrict__ gout, const uint8_t* __restrict__ mask,
                                float* __restrict__ gsrc, float* __restrict__ gflow,
                                int C, int H, int W, int ac) {
    __shared__ float red[2][NY][64];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    const bool live = (x < W);
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + (live ? x : 0);
    float gix = 0.f, giy = 0.f;
    {
        // dead lanes (x >= W) run the loop too, with keep == false: the in-wave shuffles below need every lane
        const float u = live ? flow[((size_t)b * 2) * plane + pix] : 0.f;
        const float v = live ? flow[((size_t)b * 2 + 1) * plane + pix] : 0.f;
        const Taps t = make_taps(u, v, live ? x : 0, y, H, W, ac);
        const bool keep = live && (!MASKED || (mask[(size_t)b * plane + pix] != 0));
        // Scatter-add of the source gradient.  With a locally smooth flow, lane l's right-hand taps (ne, se) hit
        // the same source pixels as lane l+1's left-hand taps (nw, sw): the pair is summed in-wave (one DPP
        // shuffle per row) and issued as ONE atomic by lane l+1, which halves the float atomics -- the op runs
        // at the chip-wide atomic rate (~1.3 TB/s of added bytes), not at HBM rate.
        const int lane = threadIdx.x;                      // blockDim.x == 64: one wave per threadIdx.y
        const int my_nw = (keep && t.v_nw) ? t.o_nw : -1, my_sw = (keep && t.v_sw) ? t.o_sw : -1;
        const int my_ne = (keep && t.v_ne) ? t.o_ne : -2, my_se = (keep && t.v_se) ? t.o_se : -2;
        const int left_ne = __shfl_up(my_ne, 1, 64), left_se = __shfl_up(my_se, 1, 64);
        // take over the left neighbour's ne / se contribution when it targets my nw / sw pixel
        const bool take_n = WITH_GSRC && lane > 0 && left_ne == my_nw && my_nw >= 0;
        const bool take_s = WITH_GSRC && lane > 0 && left_se == my_sw && my_sw >= 0;
        const bool give_n = __shfl_down((int)take_n, 1, 64) && lane < 63;   // my ne is handled by lane+1
        const bool give_s = __shfl_down((int)take_s, 1, 64) && lane < 63;
        {
            const float* sp = src + (size_t)b * C * plane;
            const float* gp = gout + (size_t)b * C * plane + pix;
            float* dp = WITH_GSRC ? gsrc + (size_t)b * C * plane : nullptr;
#pragma unroll 4
            for (int c = threadIdx.y; c < C; c += NY) {
                const float* p = sp + (size_t)c * plane;
                const float g = keep ? gp[(size_t)c * plane] : 0.f;
                if (keep) {
                    const float a = t.v_nw ? p[t.o_nw] : 0.f, bq = t.v_ne ? p[t.o_ne] : 0.f;
                    const float cq = t.v_sw ? p[t.o_sw] : 0.f, dq = t.v_se ? p[t.o_se] : 0.f;
                    gix += g * ((bq - a) * t.s + (dq - cq) * t.n);
                    giy += g * ((cq - a) * t.e + (dq - bq) * t.w);
                }
                if (WITH_GSRC) {
                    float* d = dp + (size_t)c * plane;
                    const float c_ne = g * t.ne, c_se = g * t.se;           // all lanes shuffle (no divergence here)
                    const float from_n = __shfl_up(c_ne, 1, 64), from_s = __shfl_up(c_se, 1, 64);
                    if (my_nw >= 0) atomicAdd(d + my_nw, g * t.nw + (take_n ? from_n : 0.f));
                    if (my_sw >= 0) atomicAdd(d + my_sw, g * t.sw + (take_s ? from_s : 0.f));
                    if (my_ne >= 0 && !give_n) atomicAdd(d + my_ne, c_ne);
                    if (my_se >= 0 && !give_s) atomicAdd(d + my_se, c_se);
                }
            }
        }
    }
    if (NY > 1) {
        red[0][threadIdx.y][threadIdx.x] = gix;
        red[1][threadIdx.y][threadIdx.x] = giy;
        __syncthreads();
        if (threadIdx.y != 0) return;
        gix = 0.f; giy = 0.f;
#pragma unroll
        for (int k = 0; k < NY; ++k) { gix += red[0][k][threadIdx.x]; giy += red[1][k][threadIdx.x]; }
    }
    if (!live) return;
    // d(ix)/d(g) = W/2 (or (W-1)/2); d(g)/d(v) = 2/max(W-1,1)   (net_utils.py:42-43 in reverse)
    const float mx = ac ? (float)(W - 1) * 0.5f : (float)W * 0.5f;
    const float my = ac ? (float)(H - 1) * 0.5f : (float)H * 0.5f;
    const float dx = (float)(W > 1 ? W - 1 : 1), dy = (float)(H > 1 ? H - 1 : 1);
    gflow[((size_t)b * 2) * plane + pix] = (gix * mx) / dx * 2.0f;
    gflow[((size_t)b * 2 + 1) * plane + pix] = (giy * my) / dy * 2.0f;
}

}  // namespace

extern "C" int unflow_warp_fwd(const float* src, const float* flow, float* out, uint8_t* mask,
                               int B, int C, int H, int W, int align_corners, void* stream) {
    UNFLOW_REQUIRE(src && flow && out && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ceil_div(W, 64), H, B);
    const int ac = align_corners ? 1 : 0;
    if (C <= 4) {
        if (mask) hipLaunchKernelGGL((warp_fwd_kernel<1, true>), grid, dim3(64, 1), 0, s, src, flow, out, mask, C, H, W, ac);
        else      hipLaunchKernelGGL((warp_fwd_kernel<1, false>), grid, dim3(64, 1), 0, s, src, flow, out, mask, C, H, W, ac);
    } else if ((long)grid.x * H * B < 4096 && C >= 64) {      // small maps: more channel phases per workgroup
        if (mask) hipLaunchKernelGGL((warp_fwd_kernel<16, true>), grid, dim3(64, 16), 0, s, src, flow, out, mask, C, H, W, ac);
        else      hipLaunchKernelGGL((warp_fwd_kernel<16, false>), grid, dim3(64, 16), 0, s, src, flow, out, mask, C, H, W, ac);
    } else {
        if (mask) hipLaunchKernelGGL((warp_fwd_kernel<4, true>), grid, dim3(64, 4), 0, s, src, flow, out, mask, C, H, W, ac);
        else      hipLaunchKernelGGL((warp_fwd_kernel<4, false>), grid, dim3(64, 4), 0, s, src, flow, out, mask, C, H, W, ac);
    }
    return unflow_launch_status();
}

extern "C" int unflow_warp_bwd(const float* src, const float* flow, const float* gout, const uint8_t* mask,
                               float* gsrc, float* gflow, int B, int C, int H, int W, int align_corners,
                               void* stream) {
    UNFLOW_REQUIRE(src && flow && gout && gflow && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    if (gsrc) unflow_zero_async(gsrc, (size_t)B * C * H * W, s);
    dim3 grid(ceil_div(W, 64), H, B);
    const int ac = align_corners ? 1 : 0;
#define LAUNCH(NY, M, G) hipLaunchKernelGGL((warp_bwd_kernel<NY, M, G>), grid, dim3(64, NY), 0, s, src, flow, gout, mask, gsrc, gflow, C, H, W, ac)
    // few pixels, many channels (pyramid levels 4-6): 16 channel phases per workgroup instead of 4 -- the launch
    // has too few pixel rows to fill the chip, and the per-wave channel loop is a serial chain of atomics
    const bool small_map = (long)grid.x * H * B < 4096 && C >= 64;
    if (C <= 4) {
        if (mask) { if (gsrc) LAUNCH(1, true, true); else LAUNCH(1, true, false); }
        else      { if (gsrc) LAUNCH(1, false, true); else LAUNCH(1, false, false); }
    } else if (small_map) {
        if (mask) { if (gsrc) LAUNCH(16, true, true); else LAUNCH(16, true, false); }
        else      { if (gsrc) LAUNCH(16, false, true); else LAUNCH(16, false, false); }
    } else {
        if (mask) { if (gsrc) LAUNCH(4, true, true); else LAUNCH(4, true, false); }
        else      { if (gsrc) LAUNCH(4, false, true); else LAUNCH(4, false, false); }
    }
#undef LAUNCH
    return unflow_launch_status();
}
