// HBM-bound glue around the convolutions, fused for gfx950.
//
//  * bias + LeakyReLU(0.1) applied in place to a bias-free convolution output, and its backward
//    fused with the bias-gradient reduction.  The reference's conv() block (net_utils.py:7-11,
//    Conv2d(bias=True) + LeakyReLU) costs, per convolution, a bias-add pass and an activation pass
//    forward, and an activation-backward pass plus a reduction pass backward -- each a full
//    read+write of the activation (109 MB at level 2).  Here: one pass forward, one backward.
//  * the image pyramid of Model_flow.generate_img_pyramid (model_flow_paper.py:54-60,
//    adaptive_avg_pool2d to H/2, H/4): both scales from one read of the full-resolution frames.
//
// Layout NCHW fp32; lanes along the contiguous H*W axis, 16 bytes per lane when H*W % 4 == 0.
#include "common.h"
#include <stdint.h>

namespace {

constexpr int EW_TILE = 4096;      // elements of one (n, c) plane per workgroup

__global__ __launch_bounds__(256) void bias_leaky_fwd_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                             int C, int HW, float slope) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float b = bias[c];
    float* p = y + ((size_t)n * C + c) * HW;
    const int e0 = blockIdx.x * EW_TILE;
    if ((HW & 3) == 0) {
#pragma unroll
        for (int k = 0; k < EW_TILE / 1024; ++k) {
            const int e = e0 + (k * 256 + threadIdx.x) * 4;
            if (e < HW) {
                float4 v = *reinterpret_cast<float4*>(p + e);
                v.x += b; v.y += b; v.z += b; v.w += b;
                v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
                v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
                *reinterpret_cast<float4*>(p + e) = v;
            }
        }
    } else {
        for (int e = e0 + threadIdx.x; e < min(e0 + EW_TILE, HW); e += 256) {
            float v = p[e] + b;
            p[e] = v > 0.f ? v : v * slope;
        }
    }
}

// gin = (gout [+ gout2]) * (y > 0 ? 1 : slope)   (y is the activation OUTPUT; LeakyReLU keeps the sign)
// partials[(c * N + n) * nchunk + chunk] = sum of gin over the workgroup's elements
// gout / gout2 may be channel slices of a wider NCHW tensor (the views autograd hands out for a
// torch.cat operand): their sample stride is passed in elements, the (C,H,W) block of a sample is dense.
// TWO: the activation had two consumers; adding their gradients here replaces a separate add pass.
template <bool TWO>
__global__ __launch_bounds__(256) void bias_leaky_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gout,
                                                             long long gstride, const float* __restrict__ gout2,
                                                             long long gstride2, float* __restrict__ gin,
                                                             float* __restrict__ partials, int C, int HW, float slope) {
    __shared__ float red[4];
    const int c = blockIdx.y, n = blockIdx.z, N = gridDim.z;
    const size_t base = ((size_t)n * C + c) * HW;
    const float* ga = gout + (size_t)n * gstride + (size_t)c * HW;
    const float* gb = TWO ? gout2 + (size_t)n * gstride2 + (size_t)c * HW : nullptr;
    const int e0 = blockIdx.x * EW_TILE;
    float acc[1] = {0.f};
    if ((HW & 3) == 0) {
#pragma unroll
        for (int k = 0; k < EW_TILE / 1024; ++k) {
            const int e = e0 + (k * 256 + threadIdx.x) * 4;
            if (e < HW) {
                const float4 v = *reinterpret_cast<const float4*>(y + base + e);
                float4 g = *reinterpret_cast<const float4*>(ga + e);
                if (TWO) {
                    const float4 h = *reinterpret_cast<const float4*>(gb + e);
                    g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
                }
                g.x = v.x > 0.f ? g.x : g.x * slope; g.y = v.y > 0.f ? g.y : g.y * slope;
                g.z = v.z > 0.f ? g.z : g.z * slope; g.w = v.w > 0.f ? g.w : g.w * slope;
                *reinterpret_cast<float4*>(gin + base + e) = g;
                acc[0] += (g.x + g.y) + (g.z + g.w);
            }
        }
    } else {
        for (int e = e0 + threadIdx.x; e < min(e0 + EW_TILE, HW); e += 256) {
            float g = ga[e];
            if (TWO) g += gb[e];
            g = y[base + e] > 0.f ? g : g * slope;
            gin[base + e] = g;
            acc[0] += g;
        }
    }
    block_sum_256<1>(acc, red);
    if (threadIdx.x == 0) partials[((size_t)c * N + n) * gridDim.x + blockIdx.x] = acc[0];
}

__global__ void bias_grad_finalize_kernel(const float* __restrict__ partials, int per_channel, float* __restrict__ gbias) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const float s = sum_partials(partials + (size_t)c * per_channel, per_channel, 1, 0, red);
    if (threadIdx.x == 0) gbias[c] = s;
}

// The second stage of MANY bias-gradient reductions in one launch (ABI 9).  A backward pass of the flow network runs 43 conv
// epilogues + 6 flow heads, each of which used to end in its own ~5 us finalize launch (218 us per step, profiles/r3); with
// gbias == NULL the backward entry points leave their per-workgroup partial sums behind and the host hands all of them to
// this kernel at the end of the pass.  A block = one channel of one job, summed exactly as the job's own finalize kernel
// would have (mode 0: sum_partials over partials[c][n]; 1: the bf16 epilogues' block_sum_256 order; 2: a flow head's
// partials[n][2]) -- bitwise the same gradients.
struct BiasJob { const float* partials; float* gbias; int n, C, mode, pad; };
constexpr int BIAS_JOBS = 56;
struct BiasBatch { BiasJob job[BIAS_JOBS]; int first[BIAS_JOBS + 1]; };

__global__ __launch_bounds__(256) void bias_grad_finalize_batch_kernel(BiasBatch b, int njobs) {
    __shared__ float red[4];
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= b.first[j + 1]) ++j;
    const int c = (int)blockIdx.x - b.first[j];
    const float* p = b.job[j].partials;
    const int n = b.job[j].n, mode = b.job[j].mode;
    float s;
    if (mode == 0) {
        s = sum_partials(p + (size_t)c * n, n, 1, 0, red);
    } else if (mode == 1) {
        float acc[1] = {0.f};
        for (int i = threadIdx.x; i < n; i += 256) acc[0] += p[(size_t)c * n + i];
        block_sum_256<1>(acc, red);
        s = acc[0];
    } else {
        s = sum_partials(p, n, 2, c, red);
    }
    if (threadIdx.x == 0) b.job[j].gbias[c] = s;
}


// ---- channels-last (NHWC) twins: the conv stacks run in channels_last (MIOpen's implicit-GEMM solvers are NHWC
// kernels; on NCHW tensors each of them is wrapped in batched_transpose launches, 2.2 ms of a 26 ms step), so their
// epilogue has to work on [P = N*H*W pixels][C] too.  A workgroup = QUADS channel quads x ROWS pixel rows
// (QUADS = C/4, ROWS = 256 / QUADS): a thread keeps ONE channel quad (its bias / its bias-gradient sum stay in
// registers) and walks pixels; consecutive threads read consecutive 16-byte pieces.
constexpr int NHWC_PIX = 128;      // pixels per workgroup
constexpr int NHWC_UNROLL = 4;     // pixels in flight per thread

__device__ __forceinline__ float4 leaky4(float4 v, float slope) {
    v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
    return v;
}

// dst1 (and dst2 when given) receive the activation with their own pixel strides: in place (dst1 == y, stride C), and / or as
// a channel slice of a wider NHWC tensor -- the decoder's cat((x_k, x_k+1)) buffers are filled by the epilogues of the two
// convolutions that produce their halves, so no torch.cat copy exists (round 3).
__global__ __launch_bounds__(256) void bias_leaky_fwd_nhwc_kernel(const float* y, const float* __restrict__ bias,
                                                                  float* dst1, long long ps1, float* dst2, long long ps2,
                                                                  long long P, int C, int rows, float slope) {
    const int quads = C >> 2;
    const int q = threadIdx.x % quads, r = threadIdx.x / quads;
    if (r >= rows) return;
    const float4 b = *reinterpret_cast<const float4*>(bias + q * 4);
    const long long p0 = (long long)blockIdx.x * NHWC_PIX, p1 = min(p0 + NHWC_PIX, P);
    for (long long p = p0 + r; p < p1; p += (long long)rows * NHWC_UNROLL) {
        float4 v[NHWC_UNROLL];
#pragma unroll
        for (int u = 0; u < NHWC_UNROLL; ++u)
            if (p + (long long)u * rows < p1) v[u] = reinterpret_cast<const float4*>(y + (p + (long long)u * rows) * C)[q];
#pragma unroll
        for (int u = 0; u < NHWC_UNROLL; ++u)
            if (p + (long long)u * rows < p1) {
                float4 t = v[u];
                t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
                t = leaky4(t, slope);
                reinterpret_cast<float4*>(dst1 + (p + (long long)u * rows) * ps1)[q] = t;
                if (dst2) reinterpret_cast<float4*>(dst2 + (p + (long long)u * rows) * ps2)[q] = t;
            }
    }
}

// gin[p][c] = (gout[p][c] [+ gout2[p][c]]) * (y > 0 ? 1 : slope); gout / gout2 may be channel slices of wider NHWC
// tensors: pixel stride in elements.  partials[c * nblocks + block] = the workgroup's sum for channel c, rows added in
// a fixed order (bitwise reproducible), finished by bias_grad_finalize_kernel.
template <bool TWO>
__global__ __launch_bounds__(256) void bias_leaky_bwd_nhwc_kernel(const float* __restrict__ y, long long yps, const float* __restrict__ gout,
                                                                  long long gps, const float* __restrict__ gout2, long long gps2,
                                                                  float* __restrict__ gin, float* __restrict__ partials,
                                                                  long long P, int C, int rows, float slope) {
    UNFLOW_DYNAMIC_LDS(float, red);                 // [rows][C]
    const int quads = C >> 2;
    const int q = threadIdx.x % quads, r = threadIdx.x / quads;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
        const long long p0 = (long long)blockIdx.x * NHWC_PIX, p1 = min(p0 + NHWC_PIX, P);
        for (long long p = p0 + r; p < p1; p += (long long)rows * NHWC_UNROLL) {
            float4 v[NHWC_UNROLL], g[NHWC_UNROLL], h[NHWC_UNROLL];
#pragma unroll
            for (int u = 0; u < NHWC_UNROLL; ++u) {
                const long long pp = p + (long long)u * rows;
                if (pp < p1) {
                    v[u] = reinterpret_cast<const float4*>(y + pp * yps)[q];      // (the activation may live in a channel slice)
                    g[u] = reinterpret_cast<const float4*>(gout + pp * gps)[q];
                    if (TWO) h[u] = reinterpret_cast<const float4*>(gout2 + pp * gps2)[q];
                }
            }
#pragma unroll
            for (int u = 0; u < NHWC_UNROLL; ++u) {
                const long long pp = p + (long long)u * rows;
                if (pp < p1) {
                    float4 t = g[u];
                    if (TWO) { t.x += h[u].x; t.y += h[u].y; t.z += h[u].z; t.w += h[u].w; }
                    t.x = v[u].x > 0.f ? t.x : t.x * slope; t.y = v[u].y > 0.f ? t.y : t.y * slope;
                    t.z = v[u].z > 0.f ? t.z : t.z * slope; t.w = v[u].w > 0.f ? t.w : t.w * slope;
                    reinterpret_cast<float4*>(gin + pp * C)[q] = t;
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                }
            }
        }
        *reinterpret_cast<float4*>(red + r * C + q * 4) = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int k = 0; k < rows; ++k) s += red[k * C + c];
        partials[(size_t)c * gridDim.x + blockIdx.x] = s;
    }
}

// ---- layout glue at the borders of the channels_last conv stacks: the decoder input cat((cost volume, features, flow))
// is written as NHWC directly, and NHWC activations / gradients come back as NCHW planes, through a 64-pixel x C LDS
// tile: both sides of the transpose are coalesced (rows of 64 pixels on the NCHW side, the tile's 64 * C contiguous
// floats on the NHWC side).  Row stride 65 floats: a column walk hits 32 different banks.
constexpr int LG_PIX = 64, LG_LD = LG_PIX + 1;

struct Planes3 { const float* p[3]; int c[3]; };      // up to three NCHW tensors, concatenated along C
struct PlanesOut3 { float* p[3]; int c[3]; };

// (both loops keep LG_UNROLL loads in flight per lane before the dependent LDS / global writes: a one-load-at-a-time loop
// is a chain of memory round trips)
constexpr int LG_UNROLL = 8;

__device__ __forceinline__ const float* plane_of(const Planes3& src, int b, int c, int HW) {
    int k = 0, cc = c;
    if (cc >= src.c[0]) { cc -= src.c[0]; k = 1; if (cc >= src.c[1]) { cc -= src.c[1]; k = 2; } }
    return (k == 0 ? src.p[0] : k == 1 ? src.p[1] : src.p[2]) + ((size_t)b * (k == 0 ? src.c[0] : k == 1 ? src.c[1] : src.c[2]) + cc) * HW;
}

// T: the element type of the NHWC side -- float, or unsigned short = bf16 bits for the bf16 conv-stack option (the NCHW planes
// stay fp32: what the cost volume / warp kernels read and write), one round-to-nearest-even per element on the way in
__device__ __forceinline__ float nhwc_load(const float* p) { return *p; }
__device__ __forceinline__ float nhwc_load(const unsigned short* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void nhwc_store(float* p, float v) { *p = v; }
__device__ __forceinline__ void nhwc_store(unsigned short* p, float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) { *p = 0x7fc0; return; }      // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    *p = (unsigned short)(u >> 16);
}

// fold > 0 (one source tensor): the last ``fold`` samples of the output also receive the source samples ``fold`` further on --
// the gradient of a hand-off that duplicated its last samples (split_nhwc_kernel with Bin < gridDim.y)
template <typename T>
__global__ __launch_bounds__(256) void cat_nhwc_fwd_kernel(Planes3 src, T* __restrict__ dst, int HW, int C, int fold) {
    UNFLOW_DYNAMIC_LDS(float, tile);                   // [C][LG_LD]
    const int b = blockIdx.y, p0 = blockIdx.x * LG_PIX, npx = min(LG_PIX, HW - p0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int c0 = wave; c0 < C; c0 += 4 * LG_UNROLL) {
        float v[LG_UNROLL];
#pragma unroll
        for (int u = 0; u < LG_UNROLL; ++u) {
            const int c = c0 + 4 * u;
            v[u] = (c < C && lane < npx) ? plane_of(src, b, c, HW)[p0 + lane] : 0.f;
            if (fold > 0 && b >= (int)gridDim.y - fold && c < C && lane < npx) v[u] += plane_of(src, b + fold, c, HW)[p0 + lane];
        }
#pragma unroll
        for (int u = 0; u < LG_UNROLL; ++u) {
            const int c = c0 + 4 * u;
            if (c < C) tile[c * LG_LD + lane] = v[u];
        }
    }
    __syncthreads();
    T* d = dst + ((size_t)b * HW + p0) * C;
    const int n = npx * C;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int px = e / C, c = e - px * C;
        nhwc_store(d + e, tile[c * LG_LD + px]);
    }
}

// the inverse: NHWC [B][HW][C] -> up to three NCHW tensors (destinations with a null pointer are skipped)
// Bin < gridDim.y: output samples beyond Bin repeat the LAST gridDim.y - Bin source samples (the centre frame's features feed both
// decoder directions, model_flow_paper.py:198-201 run twice -> one hand-off writes them twice instead of a torch.cat((c, c)))
template <typename T>
__global__ __launch_bounds__(256) void split_nhwc_kernel(const T* __restrict__ srcp, PlanesOut3 dst, int HW, int C, int Bin) {
    UNFLOW_DYNAMIC_LDS(float, tile);
    const int b = blockIdx.y, p0 = blockIdx.x * LG_PIX, npx = min(LG_PIX, HW - p0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sb = b < Bin ? b : b - ((int)gridDim.y - Bin);
    const T* s = srcp + ((size_t)sb * HW + p0) * C;
    const int n = npx * C;
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * LG_UNROLL) {
        float v[LG_UNROLL];
#pragma unroll
        for (int u = 0; u < LG_UNROLL; ++u) v[u] = (e0 + 256 * u < n) ? nhwc_load(s + e0 + 256 * u) : 0.f;
#pragma unroll
        for (int u = 0; u < LG_UNROLL; ++u) {
            const int e = e0 + 256 * u;
            if (e < n) { const int px = e / C, c = e - px * C; tile[c * LG_LD + px] = v[u]; }
        }
    }
    __syncthreads();
    for (int c = wave; c < C; c += 4) {
        int k = 0, cc = c;
        if (cc >= dst.c[0]) { cc -= dst.c[0]; k = 1; if (cc >= dst.c[1]) { cc -= dst.c[1]; k = 2; } }
        float* d = (k == 0 ? dst.p[0] : k == 1 ? dst.p[1] : dst.p[2]);
        if (d == nullptr) continue;
        d += ((size_t)b * (k == 0 ? dst.c[0] : k == 1 ? dst.c[1] : dst.c[2]) + cc) * HW + p0;
        if (lane < npx) d[lane] = tile[c * LG_LD + lane];
    }
}

// one lane = one 4x4 input block: 4 float4 row reads -> four 2x2 means and one 4x4 mean, summed in
// ATen's adaptive_avg_pool2d order (row-major over the window, then * 1/count: exact for 4 and 16).
__global__ void img_pyramid_kernel(const float* __restrict__ img, float* __restrict__ s1, float* __restrict__ s2,
                                   int planes, int H, int W) {
    const int H4 = H >> 2, W4 = W >> 2;
    const size_t n = (size_t)planes * H4 * W4;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(t % W4), y4 = (int)((t / W4) % H4);
        const size_t pl = t / ((size_t)W4 * H4);
        const float* p = img + (pl * H + (size_t)y4 * 4) * W + (size_t)x4 * 4;
        float4 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const float4*>(p + (size_t)k * W);
        float* o1 = s1 + (pl * (H >> 1) + (size_t)y4 * 2) * (W >> 1) + (size_t)x4 * 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float a = ((r[2 * k].x + r[2 * k].y) + r[2 * k + 1].x) + r[2 * k + 1].y;
            const float b = ((r[2 * k].z + r[2 * k].w) + r[2 * k + 1].z) + r[2 * k + 1].w;
            *reinterpret_cast<float2*>(o1 + (size_t)k * (W >> 1)) = make_float2(a * 0.25f, b * 0.25f);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s = (((s + r[k].x) + r[k].y) + r[k].z) + r[k].w;
        s2[(pl * H4 + y4) * W4 + x4] = s * 0.0625f;
    }
}

}  // namespace

extern "C" int unflow_bias_leaky_fwd(float* y, const float* bias, int N, int C, int H, int W, float slope,
                                     void* stream) {
    UNFLOW_REQUIRE(y && bias && N > 0 && C > 0 && H > 0 && W > 0 && N <= 65535 && C <= 65535);
    const int HW = H * W;
    UNFLOW_LAUNCH(bias_leaky_fwd_kernel, dim3(ceil_div(HW, EW_TILE), C, N), dim3(256), 0, (hipStream_t)stream,
                       y, bias, C, HW, slope);
    return unflow_launch_status();
}

extern "C" int unflow_bias_leaky_partials(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return UNFLOW_EINVAL;
    return N * C * ceil_div(H * W, EW_TILE);
}

extern "C" int unflow_bias_leaky_bwd2(const float* y, const float* gout, long long gout_stride, const float* gout2,
                                      long long gout2_stride, float* gin, float* gbias, float* partials,
                                      int N, int C, int H, int W, float slope, void* stream) {
    UNFLOW_REQUIRE(y && gout && gin && partials && N > 0 && C > 0 && H > 0 && W > 0 && N <= 65535 && C <= 65535);
    const int HW = H * W, nchunk = ceil_div(HW, EW_TILE);
    UNFLOW_REQUIRE(gout_stride >= (long long)C * HW && (!gout2 || gout2_stride >= (long long)C * HW));
    if ((HW & 3) == 0)      // the 16-byte path needs every sample block on a 16-byte boundary
        UNFLOW_REQUIRE((gout_stride & 3) == 0 && ((size_t)gout & 15) == 0 &&
                       (!gout2 || ((gout2_stride & 3) == 0 && ((size_t)gout2 & 15) == 0)));
    hipStream_t s = (hipStream_t)stream;
    if (gout2)
        UNFLOW_LAUNCH(bias_leaky_bwd_kernel<true>, dim3(nchunk, C, N), dim3(256), 0, s, y, gout, gout_stride, gout2,
                           gout2_stride, gin, partials, C, HW, slope);
    else
        UNFLOW_LAUNCH(bias_leaky_bwd_kernel<false>, dim3(nchunk, C, N), dim3(256), 0, s, y, gout, gout_stride, gout2,
                           gout2_stride, gin, partials, C, HW, slope);
    if (gbias) UNFLOW_LAUNCH(bias_grad_finalize_kernel, dim3(C), dim3(256), 0, s, partials, N * nchunk, gbias);     // (NULL: deferred, unflow_bias_grad_finalize_batch)
    return unflow_launch_status();
}

extern "C" int unflow_bias_leaky_bwd(const float* y, const float* gout, float* gin, float* gbias, float* partials,
                                     int N, int C, int H, int W, float slope, void* stream) {
    return unflow_bias_leaky_bwd2(y, gout, (long long)C * H * W, nullptr, 0, gin, gbias, partials, N, C, H, W, slope, stream);
}


// channels-last twins: y / gin dense [P][C] (P = N*H*W), C a multiple of 4 and <= 1024
static inline int nhwc_rows(int C) { const int r = 256 / (C >> 2); return r < 1 ? 1 : r; }

static int launch_fwd_nhwc(const float* y, const float* bias, long long P, int C, float slope, float* dst1, long long ps1,
                           float* dst2, long long ps2, void* stream) {
    UNFLOW_REQUIRE(y && bias && dst1 && P > 0 && C >= 4 && (C & 3) == 0 && C <= 1024 && (((size_t)y | (size_t)bias | (size_t)dst1) & 15) == 0);
    UNFLOW_REQUIRE(ps1 >= C && (ps1 & 3) == 0 && (!dst2 || (ps2 >= C && (ps2 & 3) == 0 && ((size_t)dst2 & 15) == 0)));
    const long long blocks = (P + NHWC_PIX - 1) / NHWC_PIX;
    UNFLOW_REQUIRE(blocks < (1ll << 31));
    UNFLOW_LAUNCH(bias_leaky_fwd_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, bias, dst1, ps1, dst2, ps2,
                       P, C, nhwc_rows(C), slope);
    return unflow_launch_status();
}

extern "C" int unflow_bias_leaky_fwd_nhwc(float* y, const float* bias, long long P, int C, float slope, void* stream) {
    return launch_fwd_nhwc(y, bias, P, C, slope, y, C, nullptr, 0, stream);
}

extern "C" int unflow_bias_leaky_fwd_nhwc_to(const float* y, const float* bias, long long P, int C, float slope, float* dst1,
                                             long long dst1_pstride, float* dst2, long long dst2_pstride, void* stream) {
    return launch_fwd_nhwc(y, bias, P, C, slope, dst1, dst1_pstride, dst2, dst2_pstride, stream);
}

extern "C" int unflow_bias_leaky_partials_nhwc(long long P, int C) {
    if (P <= 0 || C <= 0) return UNFLOW_EINVAL;
    const long long n = (P + NHWC_PIX - 1) / NHWC_PIX * C;
    return n < (1ll << 31) ? (int)n : UNFLOW_EINVAL;
}

static int launch_bwd_nhwc(const float* y, long long yps, const float* gout, long long gout_pstride, const float* gout2,
                           long long gout2_pstride, float* gin, float* gbias, float* partials,
                           long long P, int C, float slope, void* stream) {
    UNFLOW_REQUIRE(y && gout && gin && partials && P > 0 && C >= 4 && (C & 3) == 0 && C <= 1024);
    UNFLOW_REQUIRE(yps >= C && (yps & 3) == 0);
    UNFLOW_REQUIRE(gout_pstride >= C && (gout_pstride & 3) == 0 && (((size_t)gout | (size_t)y | (size_t)gin) & 15) == 0);
    UNFLOW_REQUIRE(!gout2 || (gout2_pstride >= C && (gout2_pstride & 3) == 0 && ((size_t)gout2 & 15) == 0));
    const long long blocks = (P + NHWC_PIX - 1) / NHWC_PIX;
    UNFLOW_REQUIRE(blocks * C < (1ll << 31));
    hipStream_t s = (hipStream_t)stream;
    const int rows = nhwc_rows(C);
    const size_t shmem = (size_t)rows * C * sizeof(float);
    if (gout2)
        UNFLOW_LAUNCH(bias_leaky_bwd_nhwc_kernel<true>, dim3((unsigned)blocks), dim3(256), shmem, s, y, yps, gout, gout_pstride,
                           gout2, gout2_pstride, gin, partials, P, C, rows, slope);
    else
        UNFLOW_LAUNCH(bias_leaky_bwd_nhwc_kernel<false>, dim3((unsigned)blocks), dim3(256), shmem, s, y, yps, gout, gout_pstride,
                           gout2, gout2_pstride, gin, partials, P, C, rows, slope);
    if (gbias) UNFLOW_LAUNCH(bias_grad_finalize_kernel, dim3(C), dim3(256), 0, s, partials, (int)blocks, gbias);
    return unflow_launch_status();
}

extern "C" int unflow_bias_leaky_bwd2_nhwc(const float* y, const float* gout, long long gout_pstride, const float* gout2,
                                           long long gout2_pstride, float* gin, float* gbias, float* partials,
                                           long long P, int C, float slope, void* stream) {
    return launch_bwd_nhwc(y, C, gout, gout_pstride, gout2, gout2_pstride, gin, gbias, partials, P, C, slope, stream);
}

extern "C" int unflow_bias_leaky_bwd2_nhwc_from(const float* act, long long act_pstride, const float* gout, long long gout_pstride,
                                                const float* gout2, long long gout2_pstride, float* gin, float* gbias,
                                                float* partials, long long P, int C, float slope, void* stream) {
    return launch_bwd_nhwc(act, act_pstride, gout, gout_pstride, gout2, gout2_pstride, gin, gbias, partials, P, C, slope, stream);
}

template <typename T>
static int launch_cat_nhwc(const float* a, int Ca, const float* b, int Cb, const float* c, int Cc, T* out, int B, int HW, void* stream,
                           int fold = 0) {
    UNFLOW_REQUIRE(a && out && Ca > 0 && Cb >= 0 && Cc >= 0 && (Cb == 0 || b) && (Cc == 0 || c) && (Cb > 0 || Cc == 0) &&
                   B > 0 && B <= 65535 && HW > 0 && fold >= 0 && fold <= B && (fold == 0 || Cb == 0));
    const int C = Ca + Cb + Cc;
    const size_t shmem = (size_t)C * LG_LD * sizeof(float);
    UNFLOW_REQUIRE(shmem <= 160 * 1024);
    Planes3 src = {{a, b, c}, {Ca, Cb, Cc}};
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)cat_nhwc_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    UNFLOW_LAUNCH(cat_nhwc_fwd_kernel<T>, dim3(ceil_div(HW, LG_PIX), B), dim3(256), shmem, (hipStream_t)stream, src, out, HW, C, fold);
    return unflow_launch_status();
}

template <typename T>
static int launch_split_nhwc(const T* in, float* a, int Ca, float* b, int Cb, float* c, int Cc, int B, int HW, void* stream, int dup = 0) {
    UNFLOW_REQUIRE(in && Ca > 0 && Cb >= 0 && Cc >= 0 && (Cb > 0 || Cc == 0) && B > 0 && B <= 65535 && HW > 0 && dup >= 0 && dup < B);
    const int C = Ca + Cb + Cc;
    const size_t shmem = (size_t)C * LG_LD * sizeof(float);
    UNFLOW_REQUIRE(shmem <= 160 * 1024);
    PlanesOut3 dst = {{a, b, c}, {Ca, Cb, Cc}};
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)split_nhwc_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    UNFLOW_LAUNCH(split_nhwc_kernel<T>, dim3(ceil_div(HW, LG_PIX), B), dim3(256), shmem, (hipStream_t)stream, in, dst, HW, C, B - dup);
    return unflow_launch_status();
}

extern "C" int unflow_cat_nhwc(const float* a, int Ca, const float* b, int Cb, const float* c, int Cc, float* out,
                               int B, int HW, void* stream) {
    return launch_cat_nhwc<float>(a, Ca, b, Cb, c, Cc, out, B, HW, stream);
}

extern "C" int unflow_split_nhwc(const float* in, float* a, int Ca, float* b, int Cb, float* c, int Cc, int B, int HW, void* stream) {
    return launch_split_nhwc<float>(in, a, Ca, b, Cb, c, Cc, B, HW, stream);
}

// NHWC [Bin][HW][C] -> NCHW [Bin + dup][C][HW] whose last dup samples repeat samples Bin - dup .. Bin - 1, and the gradient's way back:
// NCHW [Bout + dup][C][HW] -> NHWC [Bout][HW][C] with the repeated samples' gradients added to their originals
extern "C" int unflow_to_nchw_dup(const float* in, float* out, int C, int Bin, int dup, int HW, void* stream) {
    UNFLOW_REQUIRE(dup >= 0 && dup <= Bin);
    return launch_split_nhwc<float>(in, out, C, nullptr, 0, nullptr, 0, Bin + dup, HW, stream, dup);
}
extern "C" int unflow_to_nchw_dup_bf16(const uint16_t* in, float* out, int C, int Bin, int dup, int HW, void* stream) {
    UNFLOW_REQUIRE(dup >= 0 && dup <= Bin);
    return launch_split_nhwc<unsigned short>(in, out, C, nullptr, 0, nullptr, 0, Bin + dup, HW, stream, dup);
}
extern "C" int unflow_to_nhwc_fold(const float* g, float* out, int C, int Bout, int dup, int HW, void* stream) {
    return launch_cat_nhwc<float>(g, C, nullptr, 0, nullptr, 0, out, Bout, HW, stream, dup);
}
extern "C" int unflow_to_nhwc_fold_bf16(const float* g, uint16_t* out, int C, int Bout, int dup, int HW, void* stream) {
    return launch_cat_nhwc<unsigned short>(g, C, nullptr, 0, nullptr, 0, out, Bout, HW, stream, dup);
}

// bf16 conv-stack option: the NHWC side in bf16 (what the convolutions read and write under autocast), the NCHW planes fp32
extern "C" int unflow_cat_nhwc_bf16(const float* a, int Ca, const float* b, int Cb, const float* c, int Cc, uint16_t* out,
                                    int B, int HW, void* stream) {
    return launch_cat_nhwc<unsigned short>(a, Ca, b, Cb, c, Cc, out, B, HW, stream);
}

extern "C" int unflow_split_nhwc_bf16(const uint16_t* in, float* a, int Ca, float* b, int Cb, float* c, int Cc, int B, int HW, void* stream) {
    return launch_split_nhwc<unsigned short>(in, a, Ca, b, Cb, c, Cc, B, HW, stream);
}

// ---------------------------------------------------------------------------------------------
// Flow up-sampling of the decoder (pwc_tf.py:119,131,144,156: F.interpolate(flow, scale_factor=2, bilinear) * 2.0; :174-177:
// F.interpolate(flow * 4.0, size) -- the scale factor 4 commutes with the interpolation exactly, a power of two) as ONE kernel
// each way instead of interpolate + multiply (+ their two backward kernels): ATen's up-sampling spends 13-15 us per launch on
// these 2-channel maps and its backward scatters with atomics.  Same arithmetic as ATen's upsample_bilinear2d with
// align_corners = False:  src = max((dst + 0.5) * (in / out) - 0.5, 0), i0 = floor(src), i1 = i0 + (i0 < in - 1), l1 = src - i0,
// out = l0y * (l0x * a + l1x * b) + l1y * (l0x * c + l1x * d), then * mul.  The backward is a gather (every input pixel sums the
// output pixels that read it, in a fixed order): bitwise reproducible.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void up_src(int dst, float ratio, int in, int& i0, int& i1, float& l0, float& l1) {
    float src = ((float)dst + 0.5f) * ratio - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;                                     // src >= 0: truncation = floor
    i0 = min(i0, in - 1);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

__global__ __launch_bounds__(256) void upsample_scaled_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                  int Hi, int Wi, int Ho, int Wo, float ry, float rx, float mul) {
    const int p = blockIdx.y;
    const float* xp = x + (size_t)p * Hi * Wi;
    float* op = out + (size_t)p * Ho * Wo;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Ho * Wo; t += gridDim.x * 256) {
        const int oy = t / Wo, ox = t - oy * Wo;
        int y0, y1, x0, x1; float a0, a1, b0, b1;
        up_src(oy, ry, Hi, y0, y1, a0, a1);
        up_src(ox, rx, Wi, x0, x1, b0, b1);
        const float v = a0 * (b0 * xp[y0 * Wi + x0] + b1 * xp[y0 * Wi + x1]) + a1 * (b0 * xp[y1 * Wi + x0] + b1 * xp[y1 * Wi + x1]);
        op[t] = v * mul;
    }
}

// gin[y][x] = mul * sum over the output pixels whose taps include (y, x); the candidate rows / columns of an input pixel are
// [F y - F/2 - 1, F y + 3F/2] for an up-sampling factor F = out / in (one spare on each side; the clamped first and last rows fall inside)
__global__ __launch_bounds__(256) void upsample_scaled_bwd_kernel(const float* __restrict__ g, float* __restrict__ gin,
                                                                  int Hi, int Wi, int Ho, int Wo, float ry, float rx, float mul, int Fy, int Fx) {
    const int p = blockIdx.y;
    const float* gp = g + (size_t)p * Ho * Wo;
    float* ip = gin + (size_t)p * Hi * Wi;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Hi * Wi; t += gridDim.x * 256) {
        const int y = t / Wi, x = t - y * Wi;
        const int oy_lo = max(0, Fy * y - Fy / 2 - 1), oy_hi = min(Ho - 1, Fy * y + (3 * Fy) / 2);
        const int ox_lo = max(0, Fx * x - Fx / 2 - 1), ox_hi = min(Wo - 1, Fx * x + (3 * Fx) / 2);
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float a0, a1;
            up_src(oy, ry, Hi, y0, y1, a0, a1);
            const float wy = (y0 == y ? a0 : 0.f) + (y1 == y ? a1 : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float b0, b1;
                up_src(ox, rx, Wi, x0, x1, b0, b1);
                const float wx = (x0 == x ? b0 : 0.f) + (x1 == x ? b1 : 0.f);
                row = fmaf(wx, gp[oy * Wo + ox], row);
            }
            acc = fmaf(wy, row, acc);
        }
        ip[t] = acc * mul;
    }
}

// Up-sampling factor F = 2 or 4 on both axes (every call of the decoder): the taps of an input pixel are the 2F x 2F outputs
// oy = F y - F/2 + i, ox = F x - F/2 + j with the closed-form weights  w(i) = (i + 0.5) / F  for i < F (the pixel is the lower /
// right tap i1 of those outputs)  and  1 - (i - F + 0.5) / F  for i >= F (it is their tap i0) -- exact in fp32 for a power of two,
// the values up_src() computes.  At the borders both taps of an output fall on the same pixel (src clamped to 0 / i1 clamped to
// in - 1) and the weight is l0 + l1 = 1.  No per-candidate source arithmetic: 2F x 2F loads and FMAs per input pixel.
template <int F>
__device__ __forceinline__ void up_tap_weights(int v, int n_in, float (&w)[2 * F], int (&o)[2 * F]) {
    const int n_out = F * n_in;
#pragma unroll
    for (int i = 0; i < 2 * F; ++i) {
        const int q = F * v - F / 2 + i;
        float t = i < F ? ((float)i + 0.5f) / (float)F : 1.0f - ((float)(i - F) + 0.5f) / (float)F;
        if ((v == 0 && i < F) || (v == n_in - 1 && i >= F)) t = 1.0f;
        const bool in = q >= 0 && q < n_out;
        w[i] = in ? t : 0.f;
        o[i] = in ? q : 0;
    }
}

template <int F>
__global__ __launch_bounds__(256) void upsample_scaled_bwd_pow2_kernel(const float* __restrict__ g, float* __restrict__ gin,
                                                                       int Hi, int Wi, float mul) {
    const int p = blockIdx.y, Wo = F * Wi;
    const float* gp = g + (size_t)p * (F * Hi) * Wo;
    float* ip = gin + (size_t)p * Hi * Wi;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Hi * Wi; t += gridDim.x * 256) {
        const int y = t / Wi, x = t - y * Wi;
        float wy[2 * F], wx[2 * F];
        int oy[2 * F], ox[2 * F];
        up_tap_weights<F>(y, Hi, wy, oy);
        up_tap_weights<F>(x, Wi, wx, ox);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * F; ++i) {
            const float* row = gp + (size_t)oy[i] * Wo;
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * F; ++j) r = fmaf(wx[j], row[ox[j]], r);
            acc = fmaf(wy[i], r, acc);
        }
        ip[t] = acc * mul;
    }
}

static int upsample_args_ok(const void* a, const void* b, int planes, int Hi, int Wi, int Ho, int Wo) {
    return a && b && planes > 0 && planes <= 65535 && Hi > 0 && Wi > 0 && Ho >= Hi && Wo >= Wi && Ho % Hi == 0 && Wo % Wi == 0;
}

extern "C" int unflow_upsample_scaled_fwd(const float* x, float* out, int planes, int Hi, int Wi, int Ho, int Wo, float mul, void* stream) {
    UNFLOW_REQUIRE(upsample_args_ok(x, out, planes, Hi, Wi, Ho, Wo));
    const int blocks = ceil_div(Ho * Wo, 256) < 1024 ? ceil_div(Ho * Wo, 256) : 1024;
    UNFLOW_LAUNCH(upsample_scaled_fwd_kernel, dim3(blocks, planes), dim3(256), 0, (hipStream_t)stream, x, out, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo, mul);
    return unflow_launch_status();
}

extern "C" int unflow_upsample_scaled_bwd(const float* gout, float* gin, int planes, int Hi, int Wi, int Ho, int Wo, float mul, void* stream) {
    UNFLOW_REQUIRE(upsample_args_ok(gout, gin, planes, Hi, Wi, Ho, Wo));
    const int blocks = ceil_div(Hi * Wi, 256) < 1024 ? ceil_div(Hi * Wi, 256) : 1024;
    if (Ho == 2 * Hi && Wo == 2 * Wi) {
        UNFLOW_LAUNCH(upsample_scaled_bwd_pow2_kernel<2>, dim3(blocks, planes), dim3(256), 0, (hipStream_t)stream, gout, gin, Hi, Wi, mul);
        return unflow_launch_status();
    }
    if (Ho == 4 * Hi && Wo == 4 * Wi) {
        UNFLOW_LAUNCH(upsample_scaled_bwd_pow2_kernel<4>, dim3(blocks, planes), dim3(256), 0, (hipStream_t)stream, gout, gin, Hi, Wi, mul);
        return unflow_launch_status();
    }
    UNFLOW_LAUNCH(upsample_scaled_bwd_kernel, dim3(blocks, planes), dim3(256), 0, (hipStream_t)stream, gout, gin, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo, mul, Ho / Hi, Wo / Wi);
    return unflow_launch_status();
}

// ---------------------------------------------------------------------------------------------
// The flow heads (pwc_tf.py:93-94 predict_flow: Conv2d(c, 2, 3, bias=True), no activation; :118,130,143,155,167,171
// `flow = predict_flow(x) [+ up_flow]`) at the border of the channels_last conv stack: the bias-free convolution output y
// [P][2] (NHWC, fp32 or bf16) becomes the fp32 NCHW flow  out[b][c][hw] = y[p][c] + bias[c] (+ res[b][c][hw])  in one pass --
// instead of ATen's bias add, re-layout copy and residual add -- and the way back  gy[p][c] = g[b][c][hw],
// gbias[c] = sum g  (partials[block][2], rows summed in a fixed order) replaces a re-layout copy and a 10 us reduction.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void flow_head_fwd_kernel(const T* __restrict__ y, const float* __restrict__ bias,
                                                            const float* __restrict__ res, float* __restrict__ out, int HW, long long P) {
    const float b0 = bias[0], b1 = bias[1];
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const long long n = p / HW;
        const int hw = (int)(p - n * HW);
        float v0 = nhwc_load(y + 2 * p) + b0, v1 = nhwc_load(y + 2 * p + 1) + b1;
        const size_t o = (size_t)n * 2 * HW + hw;
        if (res) { v0 += res[o]; v1 += res[o + HW]; }
        out[o] = v0;
        out[o + HW] = v1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void flow_head_bwd_kernel(const float* __restrict__ g, T* __restrict__ gy, float* __restrict__ partials,
                                                            int HW, long long P) {
    __shared__ float red[2][4];
    float s0 = 0.f, s1 = 0.f;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const long long n = p / HW;
        const int hw = (int)(p - n * HW);
        const size_t o = (size_t)n * 2 * HW + hw;
        const float g0 = g[o], g1 = g[o + HW];
        nhwc_store(gy + 2 * p, g0);
        nhwc_store(gy + 2 * p + 1, g1);
        s0 += g0; s1 += g1;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[0][wid] = s0; red[1][wid] = s1; }
    __syncthreads();
    if (threadIdx.x < 2) partials[blockIdx.x * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void flow_head_bias_finalize_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ gbias) {
    __shared__ float red[4];
    const float s0 = sum_partials(partials, nblk, 2, 0, red);
    const float s1 = sum_partials(partials, nblk, 2, 1, red);
    if (threadIdx.x == 0) { gbias[0] = s0; gbias[1] = s1; }
}

constexpr int FLOW_HEAD_MAX_BLOCKS = 512;

template <typename T>
static int launch_flow_head_fwd(const T* y, const float* bias, const float* res, float* out, int N, int HW, void* stream) {
    UNFLOW_REQUIRE(y && bias && out && N > 0 && HW > 0);
    const long long P = (long long)N * HW;
    const int blocks = (int)((P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048);
    UNFLOW_LAUNCH(flow_head_fwd_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, bias, res, out, HW, P);
    return unflow_launch_status();
}

template <typename T>
static int launch_flow_head_bwd(const float* g, T* gy, float* gbias, float* partials, int N, int HW, void* stream) {
    UNFLOW_REQUIRE(g && gy && partials && N > 0 && HW > 0);
    const long long P = (long long)N * HW;
    const int blocks = (int)((P + 255) / 256 < FLOW_HEAD_MAX_BLOCKS ? (P + 255) / 256 : FLOW_HEAD_MAX_BLOCKS);
    UNFLOW_LAUNCH(flow_head_bwd_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, gy, partials, HW, P);
    if (gbias) UNFLOW_LAUNCH(flow_head_bias_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, blocks, gbias);
    return unflow_launch_status();
}

extern "C" int unflow_flow_head_partials(void) { return 2 * FLOW_HEAD_MAX_BLOCKS; }

extern "C" int unflow_flow_head_fwd(const float* y, const float* bias, const float* res, float* out, int N, int HW, void* stream) {
    return launch_flow_head_fwd<float>(y, bias, res, out, N, HW, stream);
}
extern "C" int unflow_flow_head_fwd_bf16(const uint16_t* y, const float* bias, const float* res, float* out, int N, int HW, void* stream) {
    return launch_flow_head_fwd<unsigned short>(y, bias, res, out, N, HW, stream);
}
extern "C" int unflow_flow_head_bwd(const float* g, float* gy, float* gbias, float* partials, int N, int HW, void* stream) {
    return launch_flow_head_bwd<float>(g, gy, gbias, partials, N, HW, stream);
}
extern "C" int unflow_flow_head_bwd_bf16(const float* g, uint16_t* gy, float* gbias, float* partials, int N, int HW, void* stream) {
    return launch_flow_head_bwd<unsigned short>(g, gy, gbias, partials, N, HW, stream);
}

extern "C" int unflow_img_pyramid(const float* img, float* half, float* quarter, int planes, int H, int W,
                                  void* stream) {
    UNFLOW_REQUIRE(img && half && quarter && planes > 0 && H > 0 && W > 0 && (H & 3) == 0 && (W & 3) == 0);
    const size_t n = (size_t)planes * (H >> 2) * (W >> 2);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    UNFLOW_LAUNCH(img_pyramid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, half, quarter, planes, H, W);
    return unflow_launch_status();
}

extern "C" int unflow_bias_grad_finalize_batch(const void* const* partials, void* const* gbias, const int* n, const int* C,
                                               const int* mode, int njobs, void* stream) {
    UNFLOW_REQUIRE(partials && gbias && n && C && mode && njobs > 0);
    for (int j0 = 0; j0 < njobs; j0 += BIAS_JOBS) {
        BiasBatch b;
        const int nj = njobs - j0 < BIAS_JOBS ? njobs - j0 : BIAS_JOBS;
        int blocks = 0;
        for (int j = 0; j < nj; ++j) {
            UNFLOW_REQUIRE(partials[j0 + j] && gbias[j0 + j] && n[j0 + j] > 0 && C[j0 + j] > 0 && mode[j0 + j] >= 0 && mode[j0 + j] <= 2 &&
                           (mode[j0 + j] != 2 || C[j0 + j] == 2));
            b.job[j].partials = (const float*)partials[j0 + j]; b.job[j].gbias = (float*)gbias[j0 + j];
            b.job[j].n = n[j0 + j]; b.job[j].C = C[j0 + j]; b.job[j].mode = mode[j0 + j]; b.job[j].pad = 0;
            b.first[j] = blocks;
            blocks += C[j0 + j];
        }
        for (int j = nj; j <= BIAS_JOBS; ++j) b.first[j] = blocks;
        for (int j = nj; j < BIAS_JOBS; ++j) { b.job[j].partials = nullptr; b.job[j].gbias = nullptr; b.job[j].n = 0; b.job[j].C = 0; b.job[j].mode = 0; b.job[j].pad = 0; }
        UNFLOW_LAUNCH(bias_grad_finalize_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b, nj);
    }
    return unflow_launch_status();
}
