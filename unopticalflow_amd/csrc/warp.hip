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
#include "multiscale.h"
#include "warp_taps.h"

namespace {

// block = (64, NY): x lanes x channel phases.  grid = (ceil(W/64), H, B).
template <int NY, bool MASKED>
__global__ void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                float* __restrict__ out, uint8_t* __restrict__ mask,
                                int C, int H, int W, int ac) {
#include "bodies/warp_fwd.inc"
}

// gsrc (optional) is scatter-added; gflow is reduced over the channel phases through LDS and
// written once per pixel (no atomics, reproducible).
template <int NY, bool MASKED, bool WITH_GSRC>
__global__ void warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                const float* __restrict__ gout, const uint8_t* __restrict__ mask,
                                float* __restrict__ gsrc, float* __restrict__ gflow,
                                int C, int H, int W, int ac) {
#include "bodies/warp_bwd.inc"
}


// ---- the masked image warps of Model_flow.warp_flow_pyramid (model_flow_paper.py:62-66, net_utils.py:47-52) as ONE launch over
// the scales (csrc/multiscale.h): the two bodies above at one channel phase (C <= 4), with the mask, without a source gradient ----
#include "ms_flat_warp.h"        // warp_fwd_ms_kernel (also compiled for the host by the tests)

struct WarpBwdMsArgs { const float *src, *flow, *gout; const uint8_t* mask; float* gflow; int H, W; };
__global__ void warp_bwd_ms_kernel(MsTable<WarpBwdMsArgs> ms_table_, int C, int ac) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    constexpr int NY = 1; constexpr bool MASKED = true, WITH_GSRC = false;
    const float* __restrict__ src = ms_a_.src; const float* __restrict__ flow = ms_a_.flow; const float* __restrict__ gout = ms_a_.gout;
    const uint8_t* __restrict__ mask = ms_a_.mask; float* __restrict__ gsrc = nullptr; float* __restrict__ gflow = ms_a_.gflow;
    const int H = ms_a_.H, W = ms_a_.W;
#include "bodies/warp_bwd.inc"
}

// ---------------------------------------------------------------------------------------------
// Feature-map warp through LDS tiles (pyramid levels: many channels, smooth flow).
//
// A 256-thread workgroup owns a TW x TH tile of destination pixels (PPT pixels per thread, lanes along
// x) and a group of channels.  The taps of its pixels are computed once; their bounding box in the
// source map is the tile's SOURCE WINDOW.  When the window fits the LDS budget (WXS x WY floats per
// channel: the tile plus its flow range) the channels stream through LDS CC at a time:
//   forward : window rows are loaded coalesced (one dword per lane) into LDS, the four taps of a pixel
//             are LDS reads, outputs are coalesced stores;
//   backward: the same window feeds the flow gradient; the source gradient is accumulated into a second
//             LDS window (ds_add_f32) and flushed once per chunk with row-contiguous float atomics --
//             every source element is added ~1.2x (tile interior once, halo twice) instead of 4x by
//             scattered per-tap atomics (the op runs at the chip's float-atomic rate, not at HBM rate).
// A tile whose flow spreads its taps over more than the window falls back to per-tap global gathers /
// atomics for that tile only (noise flows, motion boundaries): always correct, decided per workgroup.
// ---------------------------------------------------------------------------------------------
constexpr int kBig = 0x3fffffff;
struct TileBounds { int dxmin, dxmax, dymin, dymax; };     // range of a tile's tap displacements (nw tap - pixel); empty: dxmin > dxmax

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

template <int PPT>
struct TileCtx {
    Taps t[PPT];
    bool live[PPT];            // the pixel exists (inside the tile and the image)
    int pix[PPT];              // y * W + x (0 for dead slots)
    int wx0, wy0, ww, wh;      // source window, workgroup-uniform; ww == 0: no tap of the tile lies inside the image
};

template <int PPT>
__device__ __forceinline__ void tile_setup(TileCtx<PPT>& k, const float* __restrict__ flow_b, int H, int W, int ac,
                                           int x0t, int y0t, int TW, int TH, int* s_box) {
    const int plane = H * W;
    int bx0 = kBig, bx1 = -1, by0 = kBig, by1 = -1;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int idx = q * 256 + (int)threadIdx.x;
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0t + lx, y = y0t + ly;
        const bool live = ly < TH && x < W && y < H;
        const int xs = live ? x : 0, ys = live ? y : 0;
        const int pix = ys * W + xs;
        const float u = live ? flow_b[pix] : 0.f;
        const float v = live ? flow_b[plane + pix] : 0.f;
        k.t[q] = make_taps(u, v, xs, ys, H, W, ac);
        k.live[q] = live;
        k.pix[q] = pix;
        if (live && (k.t[q].v_nw || k.t[q].v_ne || k.t[q].v_sw || k.t[q].v_se)) {
            bx0 = min(bx0, k.t[q].xc0); bx1 = max(bx1, k.t[q].xc1);
            by0 = min(by0, k.t[q].yc0); by1 = max(by1, k.t[q].yc1);
        }
    }
    bx0 = wave_min_i(bx0); by0 = wave_min_i(by0); bx1 = wave_max_i(bx1); by1 = wave_max_i(by1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_box[wave * 4 + 0] = bx0; s_box[wave * 4 + 1] = by0; s_box[wave * 4 + 2] = bx1; s_box[wave * 4 + 3] = by1; }
    __syncthreads();
    bx0 = min(min(s_box[0], s_box[4]), min(s_box[8], s_box[12]));
    by0 = min(min(s_box[1], s_box[5]), min(s_box[9], s_box[13]));
    bx1 = max(max(s_box[2], s_box[6]), max(s_box[10], s_box[14]));
    by1 = max(max(s_box[3], s_box[7]), max(s_box[11], s_box[15]));
    k.wx0 = bx0; k.wy0 = by0;
    k.ww = bx1 >= bx0 ? bx1 - bx0 + 1 : 0;
    k.wh = by1 >= by0 ? by1 - by0 + 1 : 0;
}


// The source window in LDS: rows [wy0, wy0+wh) x [wxa, wxa+rs) of a channel plane, packed with row stride `rs`
// (runtime: the window is as wide as this tile's flow makes it), channel stride WIN floats.
//   vec (W % 4 == 0, 16-byte aligned source): wxa = wx0 rounded down to 4, rs a multiple of 4; the window is moved by
//       global_load_lds_dwordx4 (LDS-DMA: no staging registers, every piece of a chunk in flight at once); piece p of a
//       channel = float4 slots [64p, 64p+64) of the packed window, whose plane offsets are computed once per workgroup;
//   otherwise: dword loads through registers, one row per wave-instruction.
template <int WIN>
struct Window {
    static constexpr int MAXP = (WIN / 4 + 63) / 64;
    int wxa, wy0, rs, wh, per_ch;
    bool vec;
    int goff[MAXP];            // plane offset of this lane's float4 slot in piece p, -1 beyond the window
};

template <int WIN, int PPT>
__device__ __forceinline__ bool window_setup(Window<WIN>& w, const TileCtx<PPT>& k, int W, bool vec_ok) {
    w.vec = vec_ok;
    w.wxa = vec_ok ? (k.wx0 & ~3) : k.wx0;
    w.wy0 = k.wy0;
    const int wwa = k.wx0 - w.wxa + k.ww;
    w.rs = vec_ok ? ((wwa + 3) & ~3) : wwa;
    w.wh = k.wh;
    if (k.ww == 0 || w.rs * w.wh > WIN || w.rs > 128) return false;      // (rows are moved / flushed as two 64-lane pieces)
    const int ww4 = w.rs >> 2, slots = ww4 * w.wh;
    w.per_ch = (slots + 63) >> 6;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int p = 0; p < Window<WIN>::MAXP; ++p) {
        const int sl = p * 64 + lane;
        const int r = vec_ok ? sl / ww4 : 0, c4 = sl - r * ww4;
        w.goff[p] = (vec_ok && sl < slots) ? (w.wy0 + r) * W + w.wxa + c4 * 4 : -1;
    }
    return true;
}

// Issue the staging of `nc` channel planes (wave v moves channels v, v+4, ...).  The caller waits (vmcnt(0)) and
// barriers before the first read.
template <int STRIDE, int WIN, int CC>
__device__ __forceinline__ void window_stage(float* __restrict__ dst, const Window<WIN>& w, const float* __restrict__ planes,
                                             int plane, int W, int nc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (w.vec) {
#pragma unroll
        for (int cc = 0; cc < (CC + 3) / 4; ++cc) {
            const int c = wave + 4 * cc;
            if (c < nc) {
                const float* g = planes + (size_t)c * plane;
#pragma unroll
                for (int p = 0; p < Window<WIN>::MAXP; ++p)
                    if (p < w.per_ch && w.goff[p] >= 0)
                        __builtin_amdgcn_global_load_lds((wgas_ptr)(g + w.goff[p]), (wlds_ptr)(dst + c * STRIDE + p * 256), 16, 0, 0);
            }
        }
    } else {
        const int rows = nc * w.wh;
        for (int r = wave; r < rows; r += 4) {
            const int c = r / w.wh, ry = r - c * w.wh;
            const float* g = planes + (size_t)c * plane + (size_t)(w.wy0 + ry) * W + w.wxa;
            float* d = dst + c * STRIDE + ry * w.rs;
            for (int x = lane; x < w.rs; x += 64) d[x] = g[x];
        }
    }
}

__device__ __forceinline__ void stage_fence() {
    UNFLOW_WAIT_VMCNT0();                                 // this wave's LDS-DMA pieces have landed ...
    __syncthreads();                                      // ... and everyone else's
}

// The same window moved through REGISTERS (vec windows only): window_fetch issues the loads of the next chunk, which then fly
// under the current chunk's arithmetic; window_commit writes them to LDS once the current chunk is done with the window.
// (LDS-DMA cannot be used for this: with a DMA in flight hipcc drains vmcnt(0) before every compiler-visible LDS access.)
template <int WIN, int CC>
struct WindowRegs { float4 v[(CC + 3) / 4][Window<WIN>::MAXP]; };

template <int WIN, int CC>
__device__ __forceinline__ void window_fetch(WindowRegs<WIN, CC>& r, const Window<WIN>& w, const float* __restrict__ planes,
                                             int plane, int nc) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int cc = 0; cc < (CC + 3) / 4; ++cc) {
        const int c = wave + 4 * cc;
        const float* g = planes + (size_t)c * plane;
#pragma unroll
        for (int p = 0; p < Window<WIN>::MAXP; ++p)
            r.v[cc][p] = (c < nc && p < w.per_ch && w.goff[p] >= 0) ? *reinterpret_cast<const float4*>(g + w.goff[p])
                                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int STRIDE, int WIN, int CC>
__device__ __forceinline__ void window_commit(float* __restrict__ dst, const WindowRegs<WIN, CC>& r, const Window<WIN>& w, int nc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int cc = 0; cc < (CC + 3) / 4; ++cc) {
        const int c = wave + 4 * cc;
#pragma unroll
        for (int p = 0; p < Window<WIN>::MAXP; ++p)
            if (c < nc && p < w.per_ch && w.goff[p] >= 0)
                *reinterpret_cast<float4*>(dst + c * STRIDE + p * 256 + lane * 4) = r.v[cc][p];
    }
}

// LDS offsets of a pixel's four taps inside the packed window (0 for pixels that sample nothing)
template <int WIN, int PPT>
__device__ __forceinline__ void tap_offsets(const Window<WIN>& w, const TileCtx<PPT>& k, int (&l_nw)[PPT], int (&l_ne)[PPT],
                                            int (&l_sw)[PPT], int (&l_se)[PPT]) {
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const Taps& tp = k.t[q];
        const bool any = k.live[q] && (tp.v_nw || tp.v_ne || tp.v_sw || tp.v_se);
        const int ya = any ? (tp.yc0 - w.wy0) * w.rs : 0, yb = any ? (tp.yc1 - w.wy0) * w.rs : 0;
        const int xa = any ? tp.xc0 - w.wxa : 0, xb = any ? tp.xc1 - w.wxa : 0;
        l_nw[q] = ya + xa; l_ne[q] = ya + xb; l_sw[q] = yb + xa; l_se[q] = yb + xb;
    }
}

// `table` (may be NULL): the forward leaves, per tile, the range of its pixels' tap displacements -- what the one-pass backward
// (warp_bwd_gather_kernel) needs to know before it can gather, and what would otherwise cost it a pre-pass over the flow
// (warp_tile_bounds_kernel): the taps are computed here anyway.
template <int PPT, int WIN, int CC>
__global__ __launch_bounds__(256) void warp_fwd_tile_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                            float* __restrict__ out, int C, int H, int W, int ac,
                                                            int TW, int TH, int tiles_x, int tiles_y, int cpg, int vec_ok,
                                                            TileBounds* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) float s_src[CC * WIN];
    __shared__ int s_box[16];
    // XCD-local tile order (workgroups are dealt round-robin over the XCDs in x-major order: with grid.y channel groups the
    // linear id is x + gridDim.x * y, and a tile's groups land on the same XCD when gridDim.x is a multiple of 8 -- 512 at level 2):
    // vertically / horizontally adjacent tiles share window rows, which then come from ONE L2
    const int tile_id = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int t = tile_id;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int c_begin = blockIdx.y * cpg, c_end = min(C, c_begin + cpg);
    const int plane = H * W;
    TileCtx<PPT> k;
    tile_setup<PPT>(k, flow + (size_t)b * 2 * plane, H, W, ac, bx * TW, by * TH, TW, TH, s_box);
    if (table != nullptr && blockIdx.y == 0) {              // (workgroup-uniform)
        int x0 = kBig, x1 = -kBig, y0 = kBig, y1 = -kBig;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const Taps& tp = k.t[q];
            if (k.live[q] && (tp.v_nw || tp.v_ne || tp.v_sw || tp.v_se)) {
                const int py = k.pix[q] / W, px = k.pix[q] - py * W;
                x0 = min(x0, tp.x0 - px); x1 = max(x1, tp.x0 - px);
                y0 = min(y0, tp.y0 - py); y1 = max(y1, tp.y0 - py);
            }
        }
        x0 = wave_min_i(x0); y0 = wave_min_i(y0); x1 = wave_max_i(x1); y1 = wave_max_i(y1);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();                                    // (tile_setup's readers of s_box are done)
        if (lane == 0) { s_box[wave * 4 + 0] = x0; s_box[wave * 4 + 1] = y0; s_box[wave * 4 + 2] = x1; s_box[wave * 4 + 3] = y1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            TileBounds tb;
            tb.dxmin = min(min(s_box[0], s_box[4]), min(s_box[8], s_box[12]));
            tb.dymin = min(min(s_box[1], s_box[5]), min(s_box[9], s_box[13]));
            tb.dxmax = max(max(s_box[2], s_box[6]), max(s_box[10], s_box[14]));
            tb.dymax = max(max(s_box[3], s_box[7]), max(s_box[11], s_box[15]));
            table[tile_id] = tb;
        }
        __syncthreads();                                    // (s_box is not reused below, but keep the waves together for the staging)
    }
    const float* sp = src + (size_t)b * C * plane;
    float* op = out + (size_t)b * C * plane;
    Window<WIN> w;
    if (!window_setup<WIN, PPT>(w, k, W, vec_ok != 0)) {
        // nothing sampled (all zeros) or the flow spreads the taps beyond the LDS window: per-tap global gathers
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (!k.live[q]) continue;
            const Taps& tp = k.t[q];
#pragma unroll 4
            for (int c = c_begin; c < c_end; ++c) {
                const float* p = sp + (size_t)c * plane;
                float r = p[tp.o_nw] * tp.nw;
                r = fmaf(p[tp.o_ne], tp.ne, r);
                r = fmaf(p[tp.o_sw], tp.sw, r);
                r = fmaf(p[tp.o_se], tp.se, r);
                op[(size_t)c * plane + k.pix[q]] = r;
            }
        }
        return;
    }
    int l_nw[PPT], l_ne[PPT], l_sw[PPT], l_se[PPT];
    tap_offsets<WIN, PPT>(w, k, l_nw, l_ne, l_sw, l_se);
    for (int c0 = c_begin; c0 < c_end; c0 += CC) {
        const int nc = min(CC, c_end - c0);
        window_stage<WIN, WIN, CC>(s_src, w, sp + (size_t)c0 * plane, plane, W, nc);
        stage_fence();
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const Taps& tp = k.t[q];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                if (c < nc) {
                    const float* win = s_src + c * WIN;
                    // same accumulation order as ATen: nw, ne, sw, se
                    float r = win[l_nw[q]] * tp.nw;
                    r = fmaf(win[l_ne[q]], tp.ne, r);
                    r = fmaf(win[l_sw[q]], tp.sw, r);
                    r = fmaf(win[l_se[q]], tp.se, r);
                    if (k.live[q]) op[(size_t)(c0 + c) * plane + k.pix[q]] = r;
                }
            }
        }
        __syncthreads();                                      // the window may be overwritten
    }
}

// Phase ablations (tuning builds only): 1 no LDS accumulation, 2 no flush, 4 flush without the global atomics.
#ifdef UNFLOW_TUNING
#define WARP_DBG(bit) (dbg & (bit))
#else
#define WARP_DBG(bit) 0
#endif

// SPLIT: the channels of a tile are spread over gridDim.y workgroups; their flow-gradient partials are
// added atomically into a zeroed gflow (otherwise gflow is written once, reproducibly).
template <int PPT, int WIN, int CC, bool SPLIT>
__global__ __launch_bounds__(256) void warp_bwd_tile_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                            const float* __restrict__ gout, float* __restrict__ gsrc,
                                                            float* __restrict__ gflow, int C, int H, int W, int ac,
                                                            int TW, int TH, int tiles_x, int tiles_y, int cpg, int vec_ok, int dbg) {
    // Per channel: WIN floats of window + one private cell per thread.  A tap that does not exist (outside the image,
    // dead lane) is pointed at its thread's private cell: the source copy reads 0 there, the accumulator copy absorbs
    // the (zero) contribution -- so the inner loops carry no validity branches at all.
    constexpr int WINP = WIN + 256;
    __shared__ __attribute__((aligned(16))) float s_src[CC * WINP];
    __shared__ __attribute__((aligned(16))) float s_acc[CC * WINP];
    __shared__ int s_box[16];
    int t = xcd_remap((int)blockIdx.x, (int)gridDim.x);          // (XCD-local tile order, as the forward)
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int c_begin = blockIdx.y * cpg, c_end = min(C, c_begin + cpg);
    const int plane = H * W;
    TileCtx<PPT> k;
    tile_setup<PPT>(k, flow + (size_t)b * 2 * plane, H, W, ac, bx * TW, by * TH, TW, TH, s_box);
    const float* sp = src + (size_t)b * C * plane;
    const float* gp = gout + (size_t)b * C * plane;
    float* dp = gsrc ? gsrc + (size_t)b * C * plane : nullptr;
    float gix[PPT], giy[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) { gix[q] = 0.f; giy[q] = 0.f; }

    Window<WIN> w;
    const bool fits = window_setup<WIN, PPT>(w, k, W, vec_ok != 0);
    if (k.ww == 0) {
        // no tap inside the image: both gradients are zero
    } else if (!fits) {
        // taps spread beyond the LDS window: per-tap global gathers and atomics for this tile
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (!k.live[q]) continue;
            const Taps& tp = k.t[q];
#pragma unroll 2
            for (int c = c_begin; c < c_end; ++c) {
                const float* p = sp + (size_t)c * plane;
                const float g = gp[(size_t)c * plane + k.pix[q]];
                const float a = tp.v_nw ? p[tp.o_nw] : 0.f, bq = tp.v_ne ? p[tp.o_ne] : 0.f;
                const float cq = tp.v_sw ? p[tp.o_sw] : 0.f, dq = tp.v_se ? p[tp.o_se] : 0.f;
                gix[q] += g * ((bq - a) * tp.s + (dq - cq) * tp.n);
                giy[q] += g * ((cq - a) * tp.e + (dq - bq) * tp.w);
                if (dp) {
                    float* d = dp + (size_t)c * plane;
                    if (tp.v_nw) atomicAdd(d + tp.o_nw, g * tp.nw);
                    if (tp.v_ne) atomicAdd(d + tp.o_ne, g * tp.ne);
                    if (tp.v_sw) atomicAdd(d + tp.o_sw, g * tp.sw);
                    if (tp.v_se) atomicAdd(d + tp.o_se, g * tp.se);
                }
            }
        }
    } else {
        int o_nw[PPT], o_ne[PPT], o_sw[PPT], o_se[PPT];
        {
            int l_nw[PPT], l_ne[PPT], l_sw[PPT], l_se[PPT];
            tap_offsets<WIN, PPT>(w, k, l_nw, l_ne, l_sw, l_se);
            const int mine = WIN + (int)threadIdx.x;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const Taps& tp = k.t[q];
                o_nw[q] = (k.live[q] && tp.v_nw) ? l_nw[q] : mine;
                o_ne[q] = (k.live[q] && tp.v_ne) ? l_ne[q] : mine;
                o_sw[q] = (k.live[q] && tp.v_sw) ? l_sw[q] : mine;
                o_se[q] = (k.live[q] && tp.v_se) ? l_se[q] : mine;
            }
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int used = w.rs * w.wh;                         // floats of a channel's window actually in use
        // ds_add_f32 runs ~20x slower than an LDS read + write pair, so the taps are added with PLAIN read-add-write
        // wherever that is race free: every pixel posts its id at the (unclamped) position of its nw tap; the pixel
        // that reads its own id back is the cell's "solo" owner.  Solo pixels have pairwise different positions, so
        // within one tap kind (nw / ne / sw / se) they touch pairwise different accumulator cells: four plain passes
        // separated by barriers.  The others (where the flow compresses, two pixels floor to the same position: a few
        // per cent of a smooth flow) add theirs with LDS atomics in a fifth pass.
        bool solo[PPT];
        bool any_dup = false;
        {
            int* marker = reinterpret_cast<int*>(s_acc);      // (wh + 1) x (rs + 1) cells, position (x0 + 1, y0 + 1) relative to the window
            const int mrs = w.rs + 1;
            int cell[PPT];
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const Taps& tp = k.t[q];
                const bool any = k.live[q] && (tp.v_nw || tp.v_ne || tp.v_sw || tp.v_se);
                cell[q] = any ? (tp.y0 + 1 - w.wy0) * mrs + (tp.x0 + 1 - w.wxa) : -1;
                if (cell[q] >= 0) marker[cell[q]] = q * 256 + (int)threadIdx.x;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                solo[q] = cell[q] < 0 || marker[cell[q]] == q * 256 + (int)threadIdx.x;
                any_dup = any_dup || !solo[q];
            }
            any_dup = __syncthreads_or(any_dup ? 1 : 0) != 0;
        }
        for (int i = threadIdx.x; i < CC * WINP; i += 256) { s_acc[i] = 0.f; s_src[i] = 0.f; }
        __syncthreads();                                      // (the staging below must not be overtaken by this fill)
        // Software pipeline over the channel chunks (vec windows): the NEXT chunk's window pieces and upstream gradients are
        // requested into registers before the current chunk's arithmetic and written to LDS after it, so their latency
        // (one HBM round trip per chunk, 4-8 chunks per workgroup, nothing else resident to hide it at levels 3 / 4) is off
        // the critical path.
        WindowRegs<WIN, CC> wnext;
        float gnext[PPT][CC];
        auto fetch = [&](int c0n) {
            const int ncn = min(CC, c_end - c0n);
            window_fetch<WIN, CC>(wnext, w, sp + (size_t)c0n * plane, plane, ncn);
#pragma unroll
            for (int q = 0; q < PPT; ++q)
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    gnext[q][c] = (k.live[q] && c < ncn) ? gp[(size_t)(c0n + c) * plane + k.pix[q]] : 0.f;
        };
        if (w.vec) fetch(c_begin);
        for (int c0 = c_begin; c0 < c_end; c0 += CC) {
            const int nc = min(CC, c_end - c0);
            float g[PPT][CC];                                 // upstream gradients of the chunk
            if (w.vec) {
                window_commit<WINP, WIN, CC>(s_src, wnext, w, nc);
#pragma unroll
                for (int q = 0; q < PPT; ++q)
#pragma unroll
                    for (int c = 0; c < CC; ++c) g[q][c] = gnext[q][c];
                __syncthreads();                              // window staged, accumulator zero
                if (c0 + CC < c_end) fetch(c0 + CC);
                __builtin_amdgcn_sched_barrier(0);            // (the requests go out here, not where their values are used)
            } else {
                window_stage<WINP, WIN, CC>(s_src, w, sp + (size_t)c0 * plane, plane, W, nc);
#pragma unroll
                for (int q = 0; q < PPT; ++q)
#pragma unroll
                    for (int c = 0; c < CC; ++c)
                        g[q][c] = (k.live[q] && c < nc) ? gp[(size_t)(c0 + c) * plane + k.pix[q]] : 0.f;
                stage_fence();                                // window staged, accumulator zero
            }
            // channels beyond nc hold an older chunk's (finite) values and g == 0: they add nothing
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const Taps& tp = k.t[q];
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    const float* win = s_src + c * WINP;
                    const float gv = g[q][c];
                    const float a = win[o_nw[q]], bq = win[o_ne[q]], cq = win[o_sw[q]], dq = win[o_se[q]];
                    gix[q] += gv * ((bq - a) * tp.s + (dq - cq) * tp.n);
                    giy[q] += gv * ((cq - a) * tp.e + (dq - bq) * tp.w);
                }
                __builtin_amdgcn_sched_barrier(0);            // one pixel's taps in flight at a time (registers)
            }
            if (dp && !WARP_DBG(1)) {
                const int mine = WIN + (int)threadIdx.x;
                // one tap kind at a time: within a kind the solo pixels' cells are distinct, so plain read-add-write is race free
#define TAP_PHASE(OFF, WGT)                                                                          \
                _Pragma("unroll") for (int q = 0; q < PPT; ++q) {                                    \
                    const int o = solo[q] ? OFF[q] : mine;                                           \
                    _Pragma("unroll") for (int c = 0; c < CC; ++c)                                   \
                        s_acc[c * WINP + o] += g[q][c] * k.t[q].WGT;                                 \
                    __builtin_amdgcn_sched_barrier(0);                                               \
                }
                TAP_PHASE(o_nw, nw)
                __syncthreads();
                TAP_PHASE(o_ne, ne)
                __syncthreads();
                TAP_PHASE(o_sw, sw)
                __syncthreads();
                TAP_PHASE(o_se, se)
#undef TAP_PHASE
                if (any_dup) {
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < PPT; ++q) {
                        if (!solo[q]) {
#pragma unroll
                            for (int c = 0; c < CC; ++c) {
                                float* acc = s_acc + c * WINP;
                                atomicAdd(acc + o_nw[q], g[q][c] * k.t[q].nw);
                                atomicAdd(acc + o_ne[q], g[q][c] * k.t[q].ne);
                                atomicAdd(acc + o_sw[q], g[q][c] * k.t[q].sw);
                                atomicAdd(acc + o_se[q], g[q][c] * k.t[q].se);
                            }
                        }
                    }
                }
            }
            __syncthreads();                                  // every tap of the chunk is in the LDS accumulator
            if (dp && !WARP_DBG(2)) {
                // flush: one window row per wave-instruction (row-contiguous float atomics), re-zero behind it; wave v owns
                // channels v, v+4, ..., four rows in flight per wave
#pragma unroll
                for (int cc = 0; cc < (CC + 3) / 4; ++cc) {
                    const int c = wave + 4 * cc;
                    if (c < nc) {
                        float* d = dp + (size_t)(c0 + c) * plane + (size_t)w.wy0 * W + w.wxa;
                        float* a = s_acc + c * WINP;
                        for (int ry = 0; ry < w.wh; ry += 4) {
                            float v[4], v2[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const bool row = ry + u < w.wh;
                                v[u] = (row && lane < w.rs) ? a[(ry + u) * w.rs + lane] : 0.f;
                                v2[u] = (row && lane + 64 < w.rs) ? a[(ry + u) * w.rs + lane + 64] : 0.f;
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (v[u] != 0.f) { if (!WARP_DBG(4)) atomicAdd(d + (size_t)(ry + u) * W + lane, v[u]); a[(ry + u) * w.rs + lane] = 0.f; }
                                if (v2[u] != 0.f) { if (!WARP_DBG(4)) atomicAdd(d + (size_t)(ry + u) * W + lane + 64, v2[u]); a[(ry + u) * w.rs + lane + 64] = 0.f; }
                            }
                        }
                    }
                }
            }
            // (the next chunk's staging writes s_src only; its fence orders the next taps against this flush)
        }
    }
    const float mx = ac ? (float)(W - 1) * 0.5f : (float)W * 0.5f;
    const float my = ac ? (float)(H - 1) * 0.5f : (float)H * 0.5f;
    const float dx = (float)(W > 1 ? W - 1 : 1), dy = (float)(H > 1 ? H - 1 : 1);
    float* gf = gflow + (size_t)b * 2 * plane;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (!k.live[q]) continue;
        const float vx = (gix[q] * mx) / dx * 2.0f, vy = (giy[q] * my) / dy * 2.0f;
        if (SPLIT) { atomicAdd(gf + k.pix[q], vx); atomicAdd(gf + plane + k.pix[q], vy); }
        else { gf[k.pix[q]] = vx; gf[plane + k.pix[q]] = vy; }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward through LDS tiles, "cell gather" form (shipped for maps below 32768 pixels per launch: pyramid level 4).  The scatter of the source gradient is turned
// around inside the workgroup: every pixel posts its id at the position of its nw tap (the marker table the tile
// kernel above uses to find race-free pixels); then every CELL of the source window looks up the <= 4 pixels whose
// se / sw / ne / nw tap lands on it -- once per tile -- and per channel computes its sum from their upstream gradients
// (staged in LDS by pixel id) and adds it to global memory with one row-contiguous float atomic.  No LDS
// accumulator, no read-add-write passes, no flush pass: per chunk two barriers instead of seven and about half
// the LDS operations.  Pixels that lost their marker cell to another pixel (the flow compresses there: a few per
// cent of a smooth flow) add their four taps with global atomics themselves.
// ---------------------------------------------------------------------------------------------
template <int PPT, int WIN, int CC, bool SPLIT>
__global__ __launch_bounds__(256) void warp_bwd_cell_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                            const float* __restrict__ gout, float* __restrict__ gsrc,
                                                            float* __restrict__ gflow, int C, int H, int W, int ac,
                                                            int TW, int TH, int tiles_x, int tiles_y, int cpg, int vec_ok, int dbg) {
    constexpr int WINP = WIN + 256;                 // + one private (always zero) cell per thread for taps that do not exist
    constexpr int NPX = PPT * 256;                  // pixel ids; id NPX is the "nobody" entry (gradient 0)
    constexpr int CPT = (WIN + 255) / 256;          // window cells per thread
    constexpr int MK = WIN + WIN / 4 + 136;         // (wh + 1) x (rs + 1) marker cells
    __shared__ __attribute__((aligned(16))) float s_src[CC * WINP];
    __shared__ float s_g[CC][NPX + 1];              // upstream gradients of the chunk by pixel id
    __shared__ float s_w[4][NPX];                   // tap weights (nw, ne, sw, se) by pixel id
    __shared__ float s_dup[CC * WINP];              // side accumulator for the pixels the cells cannot find (LDS atomics)
    __shared__ unsigned char s_flag[WINP];          // window cells that receive something through s_dup
    __shared__ int s_mark[MK];
    __shared__ int s_box[16];
    int t = xcd_remap((int)blockIdx.x, (int)gridDim.x);          // (XCD-local tile order, as the tile kernels)
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int c_begin = blockIdx.y * cpg, c_end = min(C, c_begin + cpg);
    const int plane = H * W;
    TileCtx<PPT> k;
    tile_setup<PPT>(k, flow + (size_t)b * 2 * plane, H, W, ac, bx * TW, by * TH, TW, TH, s_box);
    const float* sp = src + (size_t)b * C * plane;
    const float* gp = gout + (size_t)b * C * plane;
    float* dp = gsrc ? gsrc + (size_t)b * C * plane : nullptr;
    float gix[PPT], giy[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) { gix[q] = 0.f; giy[q] = 0.f; }

    Window<WIN> w;
    const bool fits = window_setup<WIN, PPT>(w, k, W, vec_ok != 0);
    if (k.ww == 0) {
        // no tap inside the image: both gradients are zero
    } else if (!fits) {
        // taps spread beyond the LDS window: per-tap global gathers and atomics for this tile
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (!k.live[q]) continue;
            const Taps& tp = k.t[q];
#pragma unroll 2
            for (int c = c_begin; c < c_end; ++c) {
                const float* p = sp + (size_t)c * plane;
                const float g = gp[(size_t)c * plane + k.pix[q]];
                const float a = tp.v_nw ? p[tp.o_nw] : 0.f, bq = tp.v_ne ? p[tp.o_ne] : 0.f;
                const float cq = tp.v_sw ? p[tp.o_sw] : 0.f, dq = tp.v_se ? p[tp.o_se] : 0.f;
                gix[q] += g * ((bq - a) * tp.s + (dq - cq) * tp.n);
                giy[q] += g * ((cq - a) * tp.e + (dq - bq) * tp.w);
                if (dp) {
                    float* d = dp + (size_t)c * plane;
                    if (tp.v_nw) atomicAdd(d + tp.o_nw, g * tp.nw);
                    if (tp.v_ne) atomicAdd(d + tp.o_ne, g * tp.ne);
                    if (tp.v_sw) atomicAdd(d + tp.o_sw, g * tp.sw);
                    if (tp.v_se) atomicAdd(d + tp.o_se, g * tp.se);
                }
            }
        }
    } else {
        // LDS offsets of the four taps for the flow gradient (taps that do not exist read this thread's zero cell)
        int o_nw[PPT], o_ne[PPT], o_sw[PPT], o_se[PPT];
        const int mine = WIN + (int)threadIdx.x;
        {
            int l_nw[PPT], l_ne[PPT], l_sw[PPT], l_se[PPT];
            tap_offsets<WIN, PPT>(w, k, l_nw, l_ne, l_sw, l_se);
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const Taps& tp = k.t[q];
                o_nw[q] = (k.live[q] && tp.v_nw) ? l_nw[q] : mine;
                o_ne[q] = (k.live[q] && tp.v_ne) ? l_ne[q] : mine;
                o_sw[q] = (k.live[q] && tp.v_sw) ? l_sw[q] : mine;
                o_se[q] = (k.live[q] && tp.v_se) ? l_se[q] : mine;
            }
        }
        const int mrs = w.rs + 1, msz = (w.wh + 1) * mrs;
        for (int i = threadIdx.x; i < msz; i += 256) s_mark[i] = -1;
        if (threadIdx.x < CC) s_g[threadIdx.x][NPX] = 0.f;
        int cell[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const Taps& tp = k.t[q];
            const int id = q * 256 + (int)threadIdx.x;
            s_w[0][id] = tp.nw; s_w[1][id] = tp.ne; s_w[2][id] = tp.sw; s_w[3][id] = tp.se;     // zero where the tap does not exist
            const bool any = k.live[q] && (tp.v_nw || tp.v_ne || tp.v_sw || tp.v_se);
            cell[q] = any ? (tp.y0 + 1 - w.wy0) * mrs + (tp.x0 + 1 - w.wxa) : -1;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PPT; ++q)
            if (cell[q] >= 0) s_mark[cell[q]] = q * 256 + (int)threadIdx.x;
        __syncthreads();
        bool solo[PPT];                                  // this pixel is the one the cells will find
        bool any_dup = false;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            solo[q] = cell[q] < 0 || s_mark[cell[q]] == q * 256 + (int)threadIdx.x;
            any_dup = any_dup || !solo[q];
        }
        // (s_src too: a last chunk with fewer than CC channels multiplies whatever the other planes hold by g == 0)
        for (int i = threadIdx.x; i < CC * WINP; i += 256) { s_dup[i] = 0.f; s_src[i] = 0.f; }
        for (int i = threadIdx.x; i < WINP; i += 256) s_flag[i] = 0;
        any_dup = __syncthreads_or(any_dup ? 1 : 0) != 0;      // (also: the fills are done before the first chunk is staged)
        if (any_dup) {
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                if (!solo[q]) { s_flag[o_nw[q]] = 1; s_flag[o_ne[q]] = 1; s_flag[o_sw[q]] = 1; s_flag[o_se[q]] = 1; }
            __syncthreads();
        }

        // this thread's window cells: who lands here with which weight (loop invariant), and where the cell is in the plane
        int cid[CPT][4], coff[CPT];
        float cw[CPT][4];
        bool via_dup[CPT];
        const int used = w.rs * w.wh;
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int e = u * 256 + (int)threadIdx.x;
            const bool in = e < used;
            via_dup[u] = any_dup && in && s_flag[e] != 0;
            const int row = in ? e / w.rs : 0, col = in ? e - row * w.rs : 0;
            coff[u] = in ? (w.wy0 + row) * W + w.wxa + col : -1;
            // marker position of a pixel = its nw tap + (1, 1): the cell's own position holds the pixel whose SE tap is here
            const int m = row * mrs + col;
            const int ids[4] = {in ? s_mark[m + mrs + 1] : -1, in ? s_mark[m + mrs] : -1, in ? s_mark[m + 1] : -1, in ? s_mark[m] : -1};
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) {             // tk: 0 nw, 1 ne, 2 sw, 3 se
                cid[u][tk] = ids[tk] >= 0 ? ids[tk] : NPX;
                cw[u][tk] = ids[tk] >= 0 ? s_w[tk][ids[tk]] : 0.f;
            }
        }

        // the next chunk's window pieces and upstream gradients are requested before this chunk's arithmetic (see the
        // accumulate-and-flush kernel above)
        WindowRegs<WIN, CC> wnext;
        float gnext[PPT][CC];
        auto fetch = [&](int c0n) {
            const int ncn = min(CC, c_end - c0n);
            window_fetch<WIN, CC>(wnext, w, sp + (size_t)c0n * plane, plane, ncn);
#pragma unroll
            for (int q = 0; q < PPT; ++q)
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    gnext[q][c] = (k.live[q] && c < ncn) ? gp[(size_t)(c0n + c) * plane + k.pix[q]] : 0.f;
        };
        if (w.vec) fetch(c_begin);
        for (int c0 = c_begin; c0 < c_end; c0 += CC) {
            const int nc = min(CC, c_end - c0);
            float g[PPT][CC];                                 // upstream gradients of the chunk
            if (w.vec) {
                window_commit<WINP, WIN, CC>(s_src, wnext, w, nc);
#pragma unroll
                for (int q = 0; q < PPT; ++q)
#pragma unroll
                    for (int c = 0; c < CC; ++c) { g[q][c] = gnext[q][c]; s_g[c][q * 256 + (int)threadIdx.x] = g[q][c]; }
                __syncthreads();                              // window staged, gradients posted
                if (c0 + CC < c_end) fetch(c0 + CC);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                window_stage<WINP, WIN, CC>(s_src, w, sp + (size_t)c0 * plane, plane, W, nc);
#pragma unroll
                for (int q = 0; q < PPT; ++q)
#pragma unroll
                    for (int c = 0; c < CC; ++c)
                        g[q][c] = (k.live[q] && c < nc) ? gp[(size_t)(c0 + c) * plane + k.pix[q]] : 0.f;
#pragma unroll
                for (int q = 0; q < PPT; ++q)
#pragma unroll
                    for (int c = 0; c < CC; ++c) s_g[c][q * 256 + (int)threadIdx.x] = g[q][c];
                stage_fence();                                // window staged, gradients posted
            }
            // channels beyond nc hold an older chunk's (finite) values and g == 0: they add nothing
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const Taps& tp = k.t[q];
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    const float* win = s_src + c * WINP;
                    const float gv = g[q][c];
                    const float a = win[o_nw[q]], bq = win[o_ne[q]], cq = win[o_sw[q]], dq = win[o_se[q]];
                    gix[q] += gv * ((bq - a) * tp.s + (dq - cq) * tp.n);
                    giy[q] += gv * ((cq - a) * tp.e + (dq - bq) * tp.w);
                }
                __builtin_amdgcn_sched_barrier(0);            // one pixel's taps in flight at a time (registers)
            }
            if (dp && !WARP_DBG(1)) {
                if (any_dup) {                                // workgroup-uniform
                    // the pixels the cells cannot find add their taps to the side accumulator (taps that do not exist land
                    // in the thread's private cell, which nobody reads)
#pragma unroll
                    for (int q = 0; q < PPT; ++q) {
                        if (solo[q] || WARP_DBG(8)) continue;
                        const Taps& tp = k.t[q];
#pragma unroll
                        for (int c = 0; c < CC; ++c) {
                            float* acc = s_dup + c * WINP;
                            atomicAdd(acc + o_nw[q], g[q][c] * tp.nw);
                            atomicAdd(acc + o_ne[q], g[q][c] * tp.ne);
                            atomicAdd(acc + o_sw[q], g[q][c] * tp.sw);
                            atomicAdd(acc + o_se[q], g[q][c] * tp.se);
                        }
                    }
                    __syncthreads();
                }
                // the four waves walk the chunk's planes in rotated order: at any moment their atomics go to four different
                // planes instead of neighbouring lines of one (measured: 16.5 -> 9.3 us of atomics at level 2)
                auto cells = [&](auto rot) {
#pragma unroll
                    for (int u = 0; u < CPT; ++u) {
                        if (coff[u] < 0) continue;
                        const int e = u * 256 + (int)threadIdx.x;  // = the cell's offset inside the packed window
#pragma unroll
                        for (int kk = 0; kk < CC; ++kk) {
                            constexpr int ROT = decltype(rot)::value;
                            const int c = (kk + ROT) % CC;
                            // fixed order nw, ne, sw, se: a cell's value does not depend on the launch geometry
                            float r = s_g[c][cid[u][0]] * cw[u][0];
                            r = fmaf(s_g[c][cid[u][1]], cw[u][1], r);
                            r = fmaf(s_g[c][cid[u][2]], cw[u][2], r);
                            r = fmaf(s_g[c][cid[u][3]], cw[u][3], r);
                            if (via_dup[u]) { r += s_dup[c * WINP + e]; s_dup[c * WINP + e] = 0.f; }
                            if (c < nc && r != 0.f && !WARP_DBG(4)) atomicAdd(dp + (size_t)(c0 + c) * plane + coff[u], r);
                        }
                    }
                };
                const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                if (wave == 0) cells(std::integral_constant<int, 0>{});
                else if (wave == 1) cells(std::integral_constant<int, 1>{});
                else if (wave == 2) cells(std::integral_constant<int, 2>{});
                else cells(std::integral_constant<int, 3>{});
            }
            __syncthreads();                                  // s_src / s_g / s_dup may be overwritten
        }
    }
    const float mx = ac ? (float)(W - 1) * 0.5f : (float)W * 0.5f;
    const float my = ac ? (float)(H - 1) * 0.5f : (float)H * 0.5f;
    const float dx = (float)(W > 1 ? W - 1 : 1), dy = (float)(H > 1 ? H - 1 : 1);
    float* gf = gflow + (size_t)b * 2 * plane;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (!k.live[q]) continue;
        const float vx = (gix[q] * mx) / dx * 2.0f, vy = (giy[q] * my) / dy * 2.0f;
        if (SPLIT) { atomicAdd(gf + k.pix[q], vx); atomicAdd(gf + plane + k.pix[q], vy); }
        else { gf[k.pix[q]] = vx; gf[plane + k.pix[q]] = vy; }
    }
}

// ---------------------------------------------------------------------------------------------
// Source gradient as a GATHER (round 3): no zero-fill, no float atomics, every element of gsrc written exactly once with a
// plain coalesced store, bitwise reproducible.
//
// The kernels above own DESTINATION tiles and scatter; their windows overlap, so the overlaps meet in global float atomics on
// a pre-zeroed gsrc (level 2: 27 MB fill + an atomic read-modify-write per window element = 1.86x the algorithmic HBM bytes).
// Here a workgroup owns a SOURCE tile S.  Which destination pixels reach it is decided by the flow, which a workgroup cannot
// know beyond what it reads -- so a pre-pass (warp_tile_bounds_kernel, 3 MB of flow) leaves for every destination tile the
// range of its pixels' tap displacements  d = (x0 - x, y0 - y)  (nw tap position minus pixel position).  The owner of S scans
// its sample's table: tile T reaches S iff  T + [dmin, dmax + 1]  meets S;  over the tiles that do, the displacements span
// [DXmin, DXmax] x [DYmin, DYmax], and a source cell s can only be hit by the pixels  p = s + k,  k in
// [-DXmax - 1, -DXmin] x [-DYmax - 1, -DYmin]  -- a window of (span + 2)^2 candidates that does not depend on how LARGE the
// flow is, only on how much it varies across a few tiles (a constant flow: 2 x 2).  The gather is then dense and branch free:
//     gsrc[c][s] = sum_k  w(s, s + k) * gout[c][s + k],   w = the bilinear weight of pixel s + k's tap that lands on s (or 0)
// with the <= 16 weights of a cell in registers (computed once per tile from per-pixel tap records in LDS) and the gout
// rows of  S + window  staged in LDS per chunk of channels.  Windows beyond 4 x 4 (motion boundaries with several pixels of
// spread, noise) take a generic per-cell loop over the candidates with global reads: slow, exact, rare.
// The flow gradient stays with the destination-tile kernel (gsrc == nullptr: no accumulation, no flush).
// ---------------------------------------------------------------------------------------------

template <int PPT>
__global__ __launch_bounds__(256) void warp_tile_bounds_kernel(const float* __restrict__ flow, TileBounds* __restrict__ table,
                                                               int H, int W, int ac, int TW, int TH, int tiles_x, int tiles_y) {
    __shared__ int s_box[16];
    int t = blockIdx.x;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int plane = H * W;
    const float* fb = flow + (size_t)b * 2 * plane;
    int x0 = kBig, x1 = -kBig, y0 = kBig, y1 = -kBig;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int idx = q * 256 + (int)threadIdx.x;
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = bx * TW + lx, y = by * TH + ly;
        if (ly < TH && x < W && y < H) {
            const Taps tp = make_taps(fb[y * W + x], fb[plane + y * W + x], x, y, H, W, ac);
            if (tp.v_nw || tp.v_ne || tp.v_sw || tp.v_se) {
                x0 = min(x0, tp.x0 - x); x1 = max(x1, tp.x0 - x);
                y0 = min(y0, tp.y0 - y); y1 = max(y1, tp.y0 - y);
            }
        }
    }
    x0 = wave_min_i(x0); y0 = wave_min_i(y0); x1 = wave_max_i(x1); y1 = wave_max_i(y1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_box[wave * 4 + 0] = x0; s_box[wave * 4 + 1] = y0; s_box[wave * 4 + 2] = x1; s_box[wave * 4 + 3] = y1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        TileBounds tb;
        tb.dxmin = min(min(s_box[0], s_box[4]), min(s_box[8], s_box[12]));
        tb.dymin = min(min(s_box[1], s_box[5]), min(s_box[9], s_box[13]));
        tb.dxmax = max(max(s_box[2], s_box[6]), max(s_box[10], s_box[14]));
        tb.dymax = max(max(s_box[3], s_box[7]), max(s_box[11], s_box[15]));
        table[blockIdx.x] = tb;
    }
}

// the weight with which pixel p (tap record: nw-tap position (x0, y0), weights with non-existent taps zeroed) lands on cell (cx, cy)
__device__ __forceinline__ float tap_weight_on(int x0, int y0, float nw, float ne, float sw, float se, int cx, int cy) {
    const float top = (x0 == cx) ? nw : ((x0 + 1 == cx) ? ne : 0.f);
    const float bot = (x0 == cx) ? sw : ((x0 + 1 == cx) ? se : 0.f);
    return (y0 == cy) ? top : ((y0 + 1 == cy) ? bot : 0.f);
}

constexpr int GK = 12;                 // largest candidate window of the LDS path: GK x GK pixels per cell
constexpr int GN = 8;                  // contributions a cell keeps in registers (a smooth flow gives 2-6; more: that cell rescans)

// one cell's gradient from scratch: every candidate straight from global memory (windows beyond GK, and the few cells of
// the LDS path that collect more than GN contributions)
template <int CC>
__device__ __forceinline__ void gather_cell_global(float (&acc)[CC], const float* __restrict__ fb, const float* __restrict__ g0, int nc,
                                                   int cx, int cy, int kx0, int ky0, int kw, int kh, int H, int W, int ac) {
    const int plane = H * W;
#pragma unroll
    for (int c = 0; c < CC; ++c) acc[c] = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
        const int py = cy + ky0 + ky;
        if (py < 0 || py >= H) continue;
        for (int kx = 0; kx < kw; ++kx) {
            const int px = cx + kx0 + kx;
            if (px < 0 || px >= W) continue;
            const Taps tp = make_taps(fb[py * W + px], fb[plane + py * W + px], px, py, H, W, ac);
            const float wgt = tap_weight_on(tp.x0, tp.y0, tp.nw, tp.ne, tp.sw, tp.se, cx, cy);
            if (wgt != 0.f) {
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    if (c < nc) acc[c] = fmaf(wgt, g0[(size_t)c * plane + py * W + px], acc[c]);
            }
        }
    }
}

// WITH_FLOW (round 4): the workgroup also owes the FLOW gradient of the destination pixels at its tile's coordinates (the same
// 2 pixels per thread): per channel one coalesced read of the upstream gradient and the four taps of the source map (L1 / L2: the
// neighbours of a warp's lanes share cache lines), accumulated over the channel chunks in registers and written once, plain
// stores -- so one launch does the whole backward, and the upstream gradient leaves HBM once.  (Several channel groups per tile
// would each hold a partial flow gradient: the host only asks for WITH_FLOW when a tile's channels stay in one workgroup.)
template <int PPT, int CC, bool WITH_FLOW>
__global__ __launch_bounds__(256) void warp_bwd_gather_kernel(const float* __restrict__ flow, const float* __restrict__ gout,
                                                              const TileBounds* __restrict__ table, float* __restrict__ gsrc,
                                                              int C, int H, int W, int ac, int TW, int TH, int tiles_x, int tiles_y,
                                                              int cpg, int vec_ok, const float* __restrict__ src, float* __restrict__ gflow) {
    constexpr int DWP = 80, DHMAX = 8 + GK - 1;             // LDS row stride (64 + GK - 1 + 3 alignment columns) / rows of the staged region
    constexpr int REGION = DHMAX * DWP;                     // floats per channel
    constexpr int kNone = -0x7fffffff - 1;
    __shared__ __attribute__((aligned(16))) float s_g[CC * REGION];      // tap records first (6 x REGION words), then gout chunks
    __shared__ int s_red[16];
    static_assert(CC >= 6, "the tap records alias the gout staging area");
    // (XCD-local tile order when the launch is one-dimensional: neighbouring tiles share halo lines of gout, flow and src)
    int t = gridDim.y == 1 ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int c_begin = blockIdx.y * cpg, c_end = min(C, c_begin + cpg);
    const int plane = H * W;
    const int sx0 = bx * TW, sy0 = by * TH;
    const int sx1 = min(sx0 + TW, W) - 1, sy1 = min(sy0 + TH, H) - 1;
    const float* fb = flow + (size_t)b * 2 * plane;
    const float* gp = gout + (size_t)b * C * plane;
    float* dp = gsrc + (size_t)b * C * plane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // ---- which destination tiles reach S, and with which displacements
    int dx0 = kBig, dx1 = -kBig, dy0 = kBig, dy1 = -kBig;
    {
        const TileBounds* tb = table + (size_t)b * tiles_x * tiles_y;
        for (int i = threadIdx.x; i < tiles_x * tiles_y; i += 256) {
            const TileBounds v = tb[i];
            const int ty = i / tiles_x, tx = i - ty * tiles_x;
            const int tx0 = tx * TW, tx1 = min(tx0 + TW, W) - 1, ty0 = ty * TH, ty1 = min(ty0 + TH, H) - 1;
            const bool reach = v.dxmin <= v.dxmax && tx0 + v.dxmin <= sx1 && tx1 + v.dxmax + 1 >= sx0 &&
                               ty0 + v.dymin <= sy1 && ty1 + v.dymax + 1 >= sy0;
            if (reach) { dx0 = min(dx0, v.dxmin); dx1 = max(dx1, v.dxmax); dy0 = min(dy0, v.dymin); dy1 = max(dy1, v.dymax); }
        }
        dx0 = wave_min_i(dx0); dy0 = wave_min_i(dy0); dx1 = wave_max_i(dx1); dy1 = wave_max_i(dy1);
        if (lane == 0) { s_red[wave * 4 + 0] = dx0; s_red[wave * 4 + 1] = dy0; s_red[wave * 4 + 2] = dx1; s_red[wave * 4 + 3] = dy1; }
        __syncthreads();
        dx0 = min(min(s_red[0], s_red[4]), min(s_red[8], s_red[12]));
        dy0 = min(min(s_red[1], s_red[5]), min(s_red[9], s_red[13]));
        dx1 = max(max(s_red[2], s_red[6]), max(s_red[10], s_red[14]));
        dy1 = max(max(s_red[3], s_red[7]), max(s_red[11], s_red[15]));
    }
    // this thread's cells
    int cx[PPT], cy[PPT];
    bool live[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int idx = q * 256 + (int)threadIdx.x;
        const int ly = idx / TW, lx = idx - ly * TW;
        cx[q] = sx0 + lx; cy[q] = sy0 + ly;
        live[q] = ly < TH && cx[q] < W && cy[q] < H;
    }
    // the flow gradient of this thread's pixels (destination role): taps once, then per channel chunk
    float t_s[PPT], t_n[PPT], t_e[PPT], t_w[PPT];
    int t_o[PPT][4];
    bool t_v[PPT][4];
    float gix[PPT], giy[PPT];
    if (WITH_FLOW) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            gix[q] = 0.f; giy[q] = 0.f;
            const int pix = live[q] ? cy[q] * W + cx[q] : 0;
            const Taps tp = make_taps(live[q] ? fb[pix] : 0.f, live[q] ? fb[plane + pix] : 0.f, live[q] ? cx[q] : 0, live[q] ? cy[q] : 0, H, W, ac);
            t_s[q] = tp.s; t_n[q] = tp.n; t_e[q] = tp.e; t_w[q] = tp.w;
            t_o[q][0] = tp.o_nw; t_o[q][1] = tp.o_ne; t_o[q][2] = tp.o_sw; t_o[q][3] = tp.o_se;
            t_v[q][0] = live[q] && tp.v_nw; t_v[q][1] = live[q] && tp.v_ne; t_v[q][2] = live[q] && tp.v_sw; t_v[q][3] = live[q] && tp.v_se;
        }
    }
    const float* sp = WITH_FLOW ? src + (size_t)b * C * plane : nullptr;
    // `g_lds`: the chunk's staged upstream gradients (channel stride `g_cs` floats) when this thread's own pixels lie inside the
    // staged region (`own_idx`), else NULL: read them from global memory.  All tap loads of a pixel's chunk go out before the first
    // is used (the offsets are clamped, always valid: no exec-masked loads; validity is applied to the values).
    int own_idx[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) own_idx[q] = 0;
    auto flow_grad = [&](int c0, int nc, const float* g_lds, int g_cs) {      // channels [c0, c0 + nc) of this workgroup's pixels
        if (!WITH_FLOW) return;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (!live[q]) continue;
            const int pix = cy[q] * W + cx[q];
            float gv[CC], tv[CC][4];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                if (c < nc) {
                    const float* p = sp + (size_t)(c0 + c) * plane;
                    gv[c] = g_lds ? g_lds[c * g_cs + own_idx[q]] : gp[(size_t)(c0 + c) * plane + pix];
#pragma unroll
                    for (int k = 0; k < 4; ++k) tv[c][k] = p[t_o[q][k]];
                }
            }
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                if (c < nc) {
                    const float a = t_v[q][0] ? tv[c][0] : 0.f, bq = t_v[q][1] ? tv[c][1] : 0.f;
                    const float cq = t_v[q][2] ? tv[c][2] : 0.f, dq = t_v[q][3] ? tv[c][3] : 0.f;
                    gix[q] += gv[c] * ((bq - a) * t_s[q] + (dq - cq) * t_n[q]);
                    giy[q] += gv[c] * ((cq - a) * t_e[q] + (dq - bq) * t_w[q]);
                }
            }
        }
    };
    auto flow_grad_store = [&]() {
        if (!WITH_FLOW) return;
        const float mx = ac ? (float)(W - 1) * 0.5f : (float)W * 0.5f;
        const float my = ac ? (float)(H - 1) * 0.5f : (float)H * 0.5f;
        const float ddx = (float)(W > 1 ? W - 1 : 1), ddy = (float)(H > 1 ? H - 1 : 1);
        float* gf = gflow + (size_t)b * 2 * plane;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (!live[q]) continue;
            const int pix = cy[q] * W + cx[q];
            gf[pix] = (gix[q] * mx) / ddx * 2.0f;
            gf[plane + pix] = (giy[q] * my) / ddy * 2.0f;
        }
    };
    if (dx0 > dx1) {                                        // nothing lands on this tile
#pragma unroll
        for (int q = 0; q < PPT; ++q)
            if (live[q])
                for (int c = c_begin; c < c_end; ++c) dp[(size_t)c * plane + cy[q] * W + cx[q]] = 0.f;
        for (int c0 = c_begin; c0 < c_end; c0 += CC) flow_grad(c0, min(CC, c_end - c0), nullptr, 0);
        flow_grad_store();
        return;
    }
    const int kx0 = -dx1 - 1, ky0 = -dy1 - 1;               // candidate p = s + (kx0 + kx, ky0 + ky)
    const int kw = dx1 - dx0 + 2, kh = dy1 - dy0 + 2;

    if (kw > GK || kh > GK || TW > 64 || TH > 8) {
        // ---- windows beyond the LDS region: every cell from global memory
#pragma unroll 1
        for (int c0 = c_begin; c0 < c_end; c0 += CC) {
            const int nc = min(CC, c_end - c0);
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                if (!live[q]) continue;
                float acc[CC];
                gather_cell_global<CC>(acc, fb, gp + (size_t)c0 * plane, nc, cx[q], cy[q], kx0, ky0, kw, kh, H, W, ac);
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    if (c < nc) dp[(size_t)(c0 + c) * plane + cy[q] * W + cx[q]] = acc[c];
            }
            flow_grad(c0, nc, nullptr, 0);
        }
        flow_grad_store();
        return;
    }

    // ---- LDS path.  Region D = S + window, origin (ox, oy) (may start outside the image: those entries stay zero)
    const int ox = sx0 + kx0, oy = sy0 + ky0;
    const int oxa = vec_ok ? (ox & ~3) : ox;                // staged rows start 16-byte aligned
    const int DW = (sx1 - sx0 + 1) + kw - 1 + (ox - oxa), DH = (sy1 - sy0 + 1) + kh - 1;      // <= 78, <= DHMAX
    // staging: thread t moves the 16-byte piece (region row t / (DWP/4), column t % (DWP/4)) of EVERY channel of a chunk --
    // one constant division per tile, no per-slot index arithmetic
    const int s_ry = (int)threadIdx.x / (DWP / 4), s_r4 = (int)threadIdx.x - s_ry * (DWP / 4);
    int s_goff = kNone;
    {
        const int px = oxa + s_r4 * 4, py = oy + s_ry;
        if (s_ry < DH && s_r4 * 4 < DW && py >= 0 && py < H && px >= 0 && px < W) s_goff = py * W + px;      // (vec: all or nothing)
    }
    const int s_loff = s_ry * DWP + s_r4 * 4;
    static_assert(DHMAX * (DWP / 4) <= 512, "one or two staging pieces per thread and channel");
    // (rows beyond 256 / (DWP/4) = 12: a second piece per thread)
    const int s_ry2 = s_ry + 256 / (DWP / 4);
    int s_goff2 = kNone;
    {
        const int t2 = (int)threadIdx.x + 256;
        const int ry2 = t2 / (DWP / 4), r42 = t2 - ry2 * (DWP / 4);
        const int px = oxa + r42 * 4, py = oy + ry2;
        if (ry2 < DH && r42 * 4 < DW && py >= 0 && py < H && px >= 0 && px < W) s_goff2 = py * W + px;
    }
    const int s_loff2 = ((int)threadIdx.x + 256) / (DWP / 4) * DWP + (((int)threadIdx.x + 256) % (DWP / 4)) * 4;
    (void)s_ry2;
    // software pipeline over the channel chunks: the NEXT chunk's pieces are requested into registers before the current
    // chunk's gather and written to LDS after it (one HBM round trip per chunk would otherwise sit on the critical path of a
    // workgroup that has only one or two neighbours on its CU)
    float4 vpre[CC];
    auto fetch = [&](int c0n) {
        const int ncn = min(CC, c_end - c0n);
        const float* gn = gp + (size_t)c0n * plane;
#pragma unroll
        for (int c = 0; c < CC; ++c)
            vpre[c] = (s_goff != kNone && c < ncn) ? *reinterpret_cast<const float4*>(gn + (size_t)c * plane + s_goff) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (vec_ok) fetch(c_begin);                              // (the first chunk's pieces fly under the record / list phases below)
    // this thread's own pixels inside the staged region? (workgroup-uniform: the candidate window then contains displacement 0)
    const bool own_in = WITH_FLOW && kx0 <= 0 && kx0 + kw > 0 && ky0 <= 0 && ky0 + kh > 0;
    if (own_in) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) own_idx[q] = live[q] ? (cy[q] - oy) * DWP + (cx[q] - oxa) : 0;
    }
    // tap records of D's pixels: ints x0, y0 and the four weights, region-indexed [y][x] with row stride DWP
    int* r_x0 = reinterpret_cast<int*>(s_g);
    int* r_y0 = r_x0 + REGION;
    float* r_w = s_g + 2 * REGION;                          // [4][REGION]
    for (int i = threadIdx.x; i < DH * DWP; i += 256) {
        const int ry = i / DWP, rx = i - ry * DWP;
        const int px = oxa + rx, py = oy + ry;
        int x0 = -kBig, y0 = -kBig;
        float nw = 0.f, ne = 0.f, sw = 0.f, se = 0.f;
        if (rx < DW && px >= 0 && px < W && py >= 0 && py < H) {
            const Taps tp = make_taps(fb[py * W + px], fb[plane + py * W + px], px, py, H, W, ac);
            x0 = tp.x0; y0 = tp.y0; nw = tp.nw; ne = tp.ne; sw = tp.sw; se = tp.se;
        }
        r_x0[i] = x0; r_y0[i] = y0;
        r_w[i] = nw; r_w[REGION + i] = ne; r_w[2 * REGION + i] = sw; r_w[3 * REGION + i] = se;
    }
    __syncthreads();
    // each cell collects the pixels that land on it: (region index, weight), at most GN in registers
    float lw[PPT][GN];
    int li[PPT][GN];
    bool over[PPT];                                         // more than GN contributions: this cell rescans from global memory
    int nmax = 0;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
#pragma unroll
        for (int j = 0; j < GN; ++j) { lw[q][j] = 0.f; li[q][j] = 0; }
        over[q] = false;
        int n = 0;
        if (live[q]) {
            const int base = (cy[q] - sy0) * DWP + (cx[q] - sx0) + (ox - oxa);          // candidate (kx, ky) at base + ky * DWP + kx
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx) {
                    const int i = base + ky * DWP + kx;
                    const int x0 = r_x0[i], y0 = r_y0[i];
                    if ((x0 == cx[q] || x0 + 1 == cx[q]) && (y0 == cy[q] || y0 + 1 == cy[q])) {
                        const float wv = r_w[((y0 == cy[q] ? 0 : 2) + (x0 == cx[q] ? 0 : 1)) * REGION + i];
                        if (wv != 0.f) {
#pragma unroll
                            for (int j = 0; j < GN; ++j)
                                if (n == j) { lw[q][j] = wv; li[q][j] = i; }
                            n++;
                        }
                    }
                }
            over[q] = n > GN;
        }
        nmax = max(nmax, min(n, GN));
    }
    nmax = wave_max_i(nmax);
    if (lane == 0) s_red[wave] = nmax;
    __syncthreads();                                        // (also: everyone is done with the records)
    nmax = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    for (int i = threadIdx.x; i < CC * REGION; i += 256) s_g[i] = 0.f;        // (entries outside the image are never staged)
    __syncthreads();
#pragma unroll 1
    for (int c0 = c_begin; c0 < c_end; c0 += CC) {
        const int nc = min(CC, c_end - c0);
        const float* g0 = gp + (size_t)c0 * plane;
        if (vec_ok) {
#pragma unroll
            for (int c = 0; c < CC; ++c)
                if (s_goff != kNone) *reinterpret_cast<float4*>(s_g + c * REGION + s_loff) = vpre[c];
            if (DH > 256 / (DWP / 4)) {                    // (workgroup-uniform: windows taller than 5 -- a second piece, not pipelined)
                float4 v[CC];
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    v[c] = (s_goff2 != kNone && c < nc) ? *reinterpret_cast<const float4*>(g0 + (size_t)c * plane + s_goff2) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    if (s_goff2 != kNone) *reinterpret_cast<float4*>(s_g + c * REGION + s_loff2) = v[c];
            }
        } else {
            // rows that are not 16-byte aligned (W % 4 != 0): element by element
            for (int i = threadIdx.x; i < CC * DH * DW; i += 256) {
                const int c = i / (DH * DW), r = i - c * (DH * DW);
                const int ry = r / DW, rx = r - ry * DW;
                const int px = oxa + rx, py = oy + ry;
                if (px >= 0 && px < W && py >= 0 && py < H)
                    s_g[c * REGION + ry * DWP + rx] = c < nc ? g0[(size_t)c * plane + py * W + px] : 0.f;
            }
        }
        __syncthreads();
        if (vec_ok && c0 + CC < c_end) fetch(c0 + CC);
        __builtin_amdgcn_sched_barrier(0);                  // (the requests go out here, not where their values are used)
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            float acc[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = 0.f;
#pragma unroll
            for (int j = 0; j < GN; ++j) {
                if (j >= nmax) break;                       // (workgroup-uniform)
                const float wv = lw[q][j];
                const float* col = s_g + li[q][j];
#pragma unroll
                for (int c = 0; c < CC; ++c) acc[c] = fmaf(wv, col[c * REGION], acc[c]);
            }
            if (over[q]) gather_cell_global<CC>(acc, fb, g0, nc, cx[q], cy[q], kx0, ky0, kw, kh, H, W, ac);
            if (live[q]) {
#pragma unroll
                for (int c = 0; c < CC; ++c)
                    if (c < nc) dp[(size_t)(c0 + c) * plane + cy[q] * W + cx[q]] = acc[c];
            }
        }
        flow_grad(c0, nc, own_in ? s_g : nullptr, REGION);
        __syncthreads();
    }
    flow_grad_store();
}

// Tile geometry for an H x W map with C channels: tiles as wide as the map allows up to 64 (52 for the 13 * 2^k wide
// KITTI levels: no dead lanes), channel groups so that the launch has >= ~1024 workgroups.
struct TilePlan { int TW, TH, tiles_x, tiles_y, groups, cpg; };
inline TilePlan plan_tiles(int B, int C, int H, int W, int TH, int CC, int want_wgs, int maxw = 64) {
    TilePlan p;
    p.tiles_x = ceil_div(W, maxw);
    p.TW = ceil_div(ceil_div(W, p.tiles_x), 4) * 4;
    p.tiles_x = ceil_div(W, p.TW);
    p.TH = TH;
    p.tiles_y = ceil_div(H, TH);
    const int tiles = p.tiles_x * p.tiles_y * B;
    int groups = ceil_div(want_wgs, tiles);
    const int max_groups = ceil_div(C, CC);
    if (groups > max_groups) groups = max_groups;
    if (groups < 1) groups = 1;
    p.cpg = ceil_div(ceil_div(C, groups), CC) * CC;
    p.groups = ceil_div(C, p.cpg);
    return p;
}

}  // namespace


// Tuning builds (tools/, -DUNFLOW_TUNING) may override the tile height / workgroup target; the shipped library
// never reads the environment.
#ifdef UNFLOW_TUNING
#include <stdlib.h>
static int wenv(const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; }
#else
static inline int wenv(const char*, int dflt) { return dflt; }
#endif

// Feature maps (no mask, >= 8 channels, >= 512 pixels: pyramid levels 2-4) go through the LDS-tile kernels.
static bool use_tiles(const uint8_t* mask, int C, int H, int W) { return mask == nullptr && C >= 8 && W >= 8 && H * W >= wenv("UNFLOW_WARP_MINPIX", 512); }

static int warp_fused_plan(int B, int C, int H, int W, TilePlan* out);

static int warp_fwd_impl(const float* src, const float* flow, float* out, uint8_t* mask, TileBounds* table,
                         int B, int C, int H, int W, int align_corners, void* stream) {
    UNFLOW_REQUIRE(src && flow && out && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const int ac = align_corners ? 1 : 0;
    if (use_tiles(mask, C, H, W) && wenv("UNFLOW_WARP_TILES", 1)) {
        const int th = wenv("UNFLOW_WARP_TH", 8);              // measured: 64x8 tiles, ~1024 workgroups (levels 2-4: 17 / 9 / 7.5 us)
        const TilePlan p = plan_tiles(B, C, H, W, th, 8, wenv("UNFLOW_WARP_WGS", 1024));
        if (table) {                                             // the table is only good for a backward with the very same tiles
            TilePlan pb;
            UNFLOW_REQUIRE(warp_fused_plan(B, C, H, W, &pb) == 2 && pb.TW == p.TW && pb.TH == p.TH && pb.tiles_x == p.tiles_x && pb.tiles_y == p.tiles_y);
        }
        dim3 grid(p.tiles_x * p.tiles_y * B, p.groups);
        const int vec_ok = ((W & 3) == 0 && (((size_t)src) & 15) == 0) ? 1 : 0;
#define LAUNCH_T(PPT, WIN) UNFLOW_LAUNCH((warp_fwd_tile_kernel<PPT, WIN, 8>), grid, dim3(256), 0, s, src, flow, out, \
                                              C, H, W, ac, p.TW, p.TH, p.tiles_x, p.tiles_y, p.cpg, vec_ok, table)
        if (p.TH == 16) LAUNCH_T(4, 1600); else LAUNCH_T(2, 1024);
#undef LAUNCH_T
        return unflow_launch_status();
    }
    UNFLOW_REQUIRE(table == nullptr);                            // (only the tile kernels know tiles)
    dim3 grid(ceil_div(W, 64), H, B);
    if (C <= 4) {
        if (mask) UNFLOW_LAUNCH((warp_fwd_kernel<1, true>), grid, dim3(64, 1), 0, s, src, flow, out, mask, C, H, W, ac);
        else      UNFLOW_LAUNCH((warp_fwd_kernel<1, false>), grid, dim3(64, 1), 0, s, src, flow, out, mask, C, H, W, ac);
    } else if ((long)grid.x * H * B < 4096 && C >= 64) {      // small maps: more channel phases per workgroup
        if (mask) UNFLOW_LAUNCH((warp_fwd_kernel<16, true>), grid, dim3(64, 16), 0, s, src, flow, out, mask, C, H, W, ac);
        else      UNFLOW_LAUNCH((warp_fwd_kernel<16, false>), grid, dim3(64, 16), 0, s, src, flow, out, mask, C, H, W, ac);
    } else {
        if (mask) UNFLOW_LAUNCH((warp_fwd_kernel<4, true>), grid, dim3(64, 4), 0, s, src, flow, out, mask, C, H, W, ac);
        else      UNFLOW_LAUNCH((warp_fwd_kernel<4, false>), grid, dim3(64, 4), 0, s, src, flow, out, mask, C, H, W, ac);
    }
    return unflow_launch_status();
}

extern "C" int unflow_warp_fwd(const float* src, const float* flow, float* out, uint8_t* mask,
                               int B, int C, int H, int W, int align_corners, void* stream) {
    return warp_fwd_impl(src, flow, out, mask, nullptr, B, C, H, W, align_corners, stream);
}

// (ABI 9) the forward of a feature-map warp that also leaves the displacement table unflow_warp_bwd_fused(table_ready = 1) reads
extern "C" int unflow_warp_fwd_table(const float* src, const float* flow, float* out, void* table,
                                     int B, int C, int H, int W, int align_corners, void* stream) {
    UNFLOW_REQUIRE(table);
    return warp_fwd_impl(src, flow, out, nullptr, (TileBounds*)table, B, C, H, W, align_corners, stream);
}

static int warp_bwd_impl(const float* src, const float* flow, const float* gout, const uint8_t* mask,
                         float* gsrc, float* gflow, int B, int C, int H, int W, int align_corners,
                         void* stream, int want_gather) {
    UNFLOW_REQUIRE(src && flow && gout && gflow && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const int ac = align_corners ? 1 : 0;
    const bool tiles = use_tiles(mask, C, H, W) && wenv("UNFLOW_WARP_TILES", 1);
    // round 3: the source gradient of the feature warps as a GATHER (no zero-fill, no atomics, bitwise reproducible).  Measured
    // against the scatter forms below (tools/microbench.py warp_gather, back to back): level 2 51.6 vs 48.6 us, level 3 37.4 vs
    // 32.2, level 4 34.0 vs 18.6 -- three launches (bounds, gather, flow gradient) against two, and a per-tile set-up that a
    // handful of channel chunks does not amortise -- so it is NOT the default: unflow_warp_bwd_det asks for it
    // (UNFLOW_WARP_GATHER=1 in tuning builds).
    const bool gather = tiles && gsrc && (want_gather || wenv("UNFLOW_WARP_GATHER", 0)) &&
                        (size_t)B * 2 * H * W * sizeof(float) >= (size_t)B * ceil_div(W, 4) * ceil_div(H, 8) * sizeof(TileBounds);
    if (gather) {
        const TilePlan p = plan_tiles(B, C, H, W, 8, 8, wenv("UNFLOW_WARP_GATHER_WGS", 512));   // (the per-tile set-up -- tap records, contribution lists -- is worth ~2 chunks: few channel groups)
        const int ntiles = p.tiles_x * p.tiles_y * B;
        TileBounds* table = reinterpret_cast<TileBounds*>(gflow);       // gflow is scratch until its own kernel (below) writes it
        UNFLOW_LAUNCH((warp_tile_bounds_kernel<2>), dim3(ntiles), dim3(256), 0, s, flow, table, H, W, ac, p.TW, p.TH, p.tiles_x, p.tiles_y);
        const int vec_g = ((W & 3) == 0 && (((size_t)gout) & 15) == 0) ? 1 : 0;
        UNFLOW_LAUNCH((warp_bwd_gather_kernel<2, 8, false>), dim3(ntiles, p.groups), dim3(256), 0, s, flow, gout, table, gsrc,
                           C, H, W, ac, p.TW, p.TH, p.tiles_x, p.tiles_y, p.cpg, vec_g, (const float*)nullptr, (float*)nullptr);
        gsrc = nullptr;                                                  // the kernels below only owe the flow gradient
    }
    if (tiles) {
        const int th = 8;                                       // 64x8 tiles, 2 px per lane: 152 VGPRs, 3 workgroups per CU
        const TilePlan p = plan_tiles(B, C, H, W, th, 4, wenv("UNFLOW_WARP_WGS", 512));   // level 2: 512 tiles, all channels in one workgroup
        dim3 tgrid(p.tiles_x * p.tiles_y * B, p.groups);
        const int vec_ok = ((W & 3) == 0 && (((size_t)src) & 15) == 0) ? 1 : 0;
        // the scatter targets are zeroed by ONE launch: gsrc, and gflow when channel groups add their partials to it
        if (gsrc && p.groups > 1) unflow_zero2_async(gsrc, (size_t)B * C * H * W, gflow, (size_t)B * 2 * H * W, s);
        else if (gsrc) unflow_zero_async(gsrc, (size_t)B * C * H * W, s);
        else if (p.groups > 1) unflow_zero_async(gflow, (size_t)B * 2 * H * W, s);
#define LAUNCH_T(KERNEL, PPT, WIN, SPLIT) UNFLOW_LAUNCH((KERNEL<PPT, WIN, 4, SPLIT>), tgrid, dim3(256), 0, s, src, flow, \
                                                     gout, gsrc, gflow, C, H, W, ac, p.TW, p.TH, p.tiles_x, p.tiles_y, p.cpg, vec_ok, wenv("UNFLOW_WARP_DEBUG", 0))
#define LAUNCH_S(KERNEL, PPT, WIN) do { if (p.groups > 1) LAUNCH_T(KERNEL, PPT, WIN, true); else LAUNCH_T(KERNEL, PPT, WIN, false); } while (0)
        // Two forms of the same tile kernel.  In the train step (in-step A/B, tools/gpu_r2_k.sh; network flows, not the
        // microbench's synthetic field): levels 2 / 3 accumulate-and-flush 55.8 / 39.6 us vs cell gather 57.4 / 41.1; level 4
        // 25.2 vs 23.7.  (Back-to-back launches on a smooth synthetic flow favour the cell form everywhere: 49 / 32 / 18 vs 52 / 33 / 18.)
        const int form = wenv("UNFLOW_WARP_BWD", (long)B * H * W < 32768 ? 1 : 0);
        if (form == 0) LAUNCH_S(warp_bwd_tile_kernel, 2, 1024);
        else LAUNCH_S(warp_bwd_cell_kernel, 2, 1024);           // (4 px per lane would need 256 VGPRs: one wave per SIMD)
#undef LAUNCH_S
#undef LAUNCH_T
        return unflow_launch_status();
    }
    if (gsrc) unflow_zero_async(gsrc, (size_t)B * C * H * W, s);
    dim3 grid(ceil_div(W, 64), H, B);
#define LAUNCH(NY, M, G) UNFLOW_LAUNCH((warp_bwd_kernel<NY, M, G>), grid, dim3(64, NY), 0, s, src, flow, gout, mask, gsrc, gflow, C, H, W, ac)
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

extern "C" int unflow_warp_bwd(const float* src, const float* flow, const float* gout, const uint8_t* mask,
                               float* gsrc, float* gflow, int B, int C, int H, int W, int align_corners,
                               void* stream) {
    return warp_bwd_impl(src, flow, gout, mask, gsrc, gflow, B, C, H, W, align_corners, stream, 0);
}

// (ABI 9) The whole backward of a feature-map warp in one pass over the upstream gradient: source gradient as a gather (no
// zero-fill, no float atomics, every element written once) AND the flow gradient, by the workgroup that owns the tile -- bitwise
// reproducible.  `table` = unflow_warp_bwd_table_bytes() bytes of scratch (the per-tile displacement ranges a pre-pass leaves
// there).  Shapes this form does not take (masked image warps, < 8 channels, small maps whose tiles would need several channel
// groups) return UNFLOW_EINVAL from unflow_warp_bwd_fused_supported() == 0: use unflow_warp_bwd.
// -> 0: the shape is not served; 1: served, but the scatter form with channel groups is the faster one (fewer than 256 workgroups of
// full-width tiles); 2: served and recommended
static int warp_fused_plan(int B, int C, int H, int W, TilePlan* out) {
    if (C < 8 || W < 8 || H * W < 512 || B <= 0) return 0;
    TilePlan p;
    for (int maxw = 64; maxw >= 32; maxw >>= 1) {                        // narrower tiles on smaller maps
        p = plan_tiles(B, C, H, W, 8, 8, 0, maxw);                       // one channel group: the flow gradient stays in one workgroup
        if (p.groups != 1 || p.TW > 64) return 0;
        if (p.tiles_x * p.tiles_y * B >= 256) {
            if (out) *out = p;
            // measured in the step (profiles/r4_warp_bwd_fused.md): level 2 (512 tiles of 52 x 8, 4 chunks of 8 channels) 50.0 us against
            // 50.6 for zero-fill + scatter; level 3 (256 tiles of 28 x 8, 8 chunks) 48.6 against 37.6 -- the per-tile set-up and the
            // exposed latency of a chunk do not amortise on 224-pixel tiles, so only full-width tiles are recommended
            return maxw == 64 ? 2 : 1;
        }
    }
    if (out) *out = p;
    return 1;
}
extern "C" int unflow_warp_bwd_fused_supported(int B, int C, int H, int W) { return warp_fused_plan(B, C, H, W, nullptr); }
extern "C" int unflow_warp_bwd_table_bytes(int B, int C, int H, int W) {
    TilePlan p;
    if (warp_fused_plan(B, C, H, W, &p) == 0) return 0;
    return (int)((size_t)p.tiles_x * p.tiles_y * B * sizeof(TileBounds));
}
extern "C" int unflow_warp_bwd_fused(const float* src, const float* flow, const float* gout, float* gsrc, float* gflow, void* table,
                                     int table_ready, int B, int C, int H, int W, int align_corners, void* stream) {
    UNFLOW_REQUIRE(src && flow && gout && gsrc && gflow && table && B > 0 && C > 0 && H > 0 && W > 0);
    TilePlan p;
    UNFLOW_REQUIRE(warp_fused_plan(B, C, H, W, &p) > 0);
    hipStream_t s = (hipStream_t)stream;
    const int ac = align_corners ? 1 : 0;
    const int ntiles = p.tiles_x * p.tiles_y * B;
    if (!table_ready)                                            // (1: unflow_warp_fwd_table filled it for this very flow)
        UNFLOW_LAUNCH((warp_tile_bounds_kernel<2>), dim3(ntiles), dim3(256), 0, s, flow, (TileBounds*)table, H, W, ac, p.TW, p.TH, p.tiles_x, p.tiles_y);
    const int vec_g = ((W & 3) == 0 && (((size_t)gout) & 15) == 0) ? 1 : 0;
    UNFLOW_LAUNCH((warp_bwd_gather_kernel<2, 8, true>), dim3(ntiles, 1), dim3(256), 0, s, flow, gout, (const TileBounds*)table, gsrc,
                       C, H, W, ac, p.TW, p.TH, p.tiles_x, p.tiles_y, p.cpg, vec_g, src, gflow);
    return unflow_launch_status();
}

extern "C" int unflow_warp_bwd_det(const float* src, const float* flow, const float* gout, const uint8_t* mask,
                                   float* gsrc, float* gflow, int B, int C, int H, int W, int align_corners,
                                   void* stream) {
    return warp_bwd_impl(src, flow, gout, mask, gsrc, gflow, B, C, H, W, align_corners, stream, 1);
}

// ---- (ABI 11) the masked image warps of a pyramid, ONE launch over n <= 4 scales each way (the conventions of the `_ms` loss entries,
// photo.hip): src[k] [B,C,H,W] with C <= 4, flow[k] [B,2,H,W] -> out[k] = warp * mask, mask[k] [B,1,H,W] uint8; the backward writes the
// flow gradient only (the images carry none).  Per scale the grid and the kernel body of unflow_warp_fwd / unflow_warp_bwd(gsrc = NULL).
#include "ms_flat_warp_entries.h"       // unflow_warp_fwd_ms

extern "C" int unflow_warp_bwd_ms(int n, const float* const* src, const float* const* flow, const float* const* gout,
                                  const uint8_t* const* mask, float* const* gflow, const int* H, const int* W, int B, int C,
                                  int align_corners, void* stream) {
    UNFLOW_REQUIRE(src && flow && gout && mask && gflow && H && W && n > 0 && n <= MS_MAX && B > 0 && B <= 65535 && C > 0 && C <= 4);
    MsTable<WarpBwdMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(src[k] && flow[k] && gout[k] && mask[k] && gflow[k] && H[k] > 0 && H[k] <= 65535 && W[k] > 0);
        t.a[k] = WarpBwdMsArgs{src[k], flow[k], gout[k], mask[k], gflow[k], H[k], W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(ceil_div(W[k], 64), H[k], B)));
    }
    UNFLOW_LAUNCH(warp_bwd_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(64, 1), 0, (hipStream_t)stream, t, C, align_corners ? 1 : 0);
    return unflow_launch_status();
}
