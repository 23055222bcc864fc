// conv() epilogue for the bf16 conv-stack option (BASELINE config 3: bf16 autocast on the convolutions):
// the same two passes as elementwise.hip -- in-place bias + LeakyReLU(0.1) forward, and a backward that
// also reduces the bias gradient and adds the gradients of up to two consumers -- on bf16 activations.
// Arithmetic in fp32 (one rounding to bf16 per element, round-to-nearest-even), bias and its gradient fp32
// (the parameter's dtype under autocast).  8 bf16 per lane (16 bytes) when H*W % 8 == 0.
#include "common.h"
#include <stdint.h>

namespace {

constexpr int EWH_TILE = 4096;     // elements of one (n, c) plane per workgroup

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;                 // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                    // round to nearest even
    return (unsigned short)(u >> 16);
}

struct alignas(16) bf8 { unsigned short v[8]; };

__global__ __launch_bounds__(256) void bias_leaky_fwd_bf16_kernel(unsigned short* __restrict__ y, const float* __restrict__ bias,
                                                                  int C, int HW, float slope) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float b = bias[c];
    unsigned short* p = y + ((size_t)n * C + c) * HW;
    const int e0 = blockIdx.x * EWH_TILE;
    if ((HW & 7) == 0) {
#pragma unroll
        for (int k = 0; k < EWH_TILE / 2048; ++k) {
            const int e = e0 + (k * 256 + threadIdx.x) * 8;
            if (e < HW) {
                bf8 v = *reinterpret_cast<bf8*>(p + e);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float f = bf2f(v.v[q]) + b;
                    v.v[q] = f2bf(f > 0.f ? f : f * slope);
                }
                *reinterpret_cast<bf8*>(p + e) = v;
            }
        }
    } else {
        for (int e = e0 + threadIdx.x; e < min(e0 + EWH_TILE, HW); e += 256) {
            const float f = bf2f(p[e]) + b;
            p[e] = f2bf(f > 0.f ? f : f * slope);
        }
    }
}

// gin = bf16((gout [+ gout2]) * (y > 0 ? 1 : slope)); the bias-gradient partial sums use the ROUNDED gin
// values (what a separate reduction over the bf16 gradient tensor would see), accumulated in fp32.
template <bool TWO>
__global__ __launch_bounds__(256) void bias_leaky_bwd_bf16_kernel(const unsigned short* __restrict__ y,
                                                                  const unsigned short* __restrict__ gout, long long gstride,
                                                                  const unsigned short* __restrict__ gout2, long long gstride2,
                                                                  unsigned short* __restrict__ gin, float* __restrict__ partials,
                                                                  int C, int HW, float slope) {
    __shared__ float red[4];
    const int c = blockIdx.y, n = blockIdx.z, N = gridDim.z;
    const size_t base = ((size_t)n * C + c) * HW;
    const unsigned short* ga = gout + (size_t)n * gstride + (size_t)c * HW;
    const unsigned short* gb = TWO ? gout2 + (size_t)n * gstride2 + (size_t)c * HW : nullptr;
    const int e0 = blockIdx.x * EWH_TILE;
    float acc[1] = {0.f};
    if ((HW & 7) == 0) {
#pragma unroll
        for (int k = 0; k < EWH_TILE / 2048; ++k) {
            const int e = e0 + (k * 256 + threadIdx.x) * 8;
            if (e < HW) {
                const bf8 v = *reinterpret_cast<const bf8*>(y + base + e);
                bf8 g = *reinterpret_cast<const bf8*>(ga + e);
                bf8 h;
                if (TWO) h = *reinterpret_cast<const bf8*>(gb + e);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float f = bf2f(g.v[q]);
                    if (TWO) f += bf2f(h.v[q]);
                    f = bf2f(v.v[q]) > 0.f ? f : f * slope;
                    g.v[q] = f2bf(f);
                    acc[0] += bf2f(g.v[q]);
                }
                *reinterpret_cast<bf8*>(gin + base + e) = g;
            }
        }
    } else {
        for (int e = e0 + threadIdx.x; e < min(e0 + EWH_TILE, HW); e += 256) {
            float f = bf2f(ga[e]);
            if (TWO) f += bf2f(gb[e]);
            f = bf2f(y[base + e]) > 0.f ? f : f * slope;
            const unsigned short r = f2bf(f);
            gin[base + e] = r;
            acc[0] += bf2f(r);
        }
    }
    block_sum_256<1>(acc, red);
    if (threadIdx.x == 0) partials[((size_t)c * N + n) * gridDim.x + blockIdx.x] = acc[0];
}

__global__ void bias_grad_finalize_bf16_kernel(const float* __restrict__ partials, int per_channel, float* __restrict__ gbias) {
    __shared__ float red[4];
    const float* p = partials + (size_t)blockIdx.x * per_channel;
    float acc[1] = {0.f};
    for (int i = threadIdx.x; i < per_channel; i += 256) acc[0] += p[i];
    block_sum_256<1>(acc, red);
    if (threadIdx.x == 0) gbias[blockIdx.x] = acc[0];
}

// ---- channels-last (NHWC) twins for the bf16 conv stacks: [P = N*H*W pixels][C] bf16, C % 4 == 0.  Same geometry as the fp32
// NHWC epilogue (elementwise.hip): a thread keeps one channel quad (8 bytes per access) and walks pixels, 4 in flight.
constexpr int HN_PIX = 128, HN_UNROLL = 4;
struct alignas(8) bf4 { unsigned short v[4]; };

__global__ __launch_bounds__(256) void bias_leaky_fwd_nhwc_bf16_kernel(const unsigned short* y, const float* __restrict__ bias,
                                                                       unsigned short* dst1, long long ps1, unsigned short* dst2, long long ps2,
                                                                       long long P, int C, int rows, float slope) {
    const int quads = C >> 2;
    const int q = threadIdx.x % quads, r = threadIdx.x / quads;
    if (r >= rows) return;
    const float4 b = *reinterpret_cast<const float4*>(bias + q * 4);
    const float bb[4] = {b.x, b.y, b.z, b.w};
    const long long p0 = (long long)blockIdx.x * HN_PIX, p1 = min(p0 + HN_PIX, P);
    for (long long p = p0 + r; p < p1; p += (long long)rows * HN_UNROLL) {
        bf4 v[HN_UNROLL];
#pragma unroll
        for (int u = 0; u < HN_UNROLL; ++u)
            if (p + (long long)u * rows < p1) v[u] = reinterpret_cast<const bf4*>(y + (p + (long long)u * rows) * C)[q];
#pragma unroll
        for (int u = 0; u < HN_UNROLL; ++u)
            if (p + (long long)u * rows < p1) {
                bf4 t = v[u];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float f = bf2f(t.v[k]) + bb[k];
                    t.v[k] = f2bf(f > 0.f ? f : f * slope);
                }
                reinterpret_cast<bf4*>(dst1 + (p + (long long)u * rows) * ps1)[q] = t;
                if (dst2) reinterpret_cast<bf4*>(dst2 + (p + (long long)u * rows) * ps2)[q] = t;
            }
    }
}

template <bool TWO>
__global__ __launch_bounds__(256) void bias_leaky_bwd_nhwc_bf16_kernel(const unsigned short* __restrict__ y, long long yps,
                                                                       const unsigned short* __restrict__ gout, long long gps,
                                                                       const unsigned short* __restrict__ gout2, long long gps2,
                                                                       unsigned short* __restrict__ gin, float* __restrict__ partials,
                                                                       long long P, int C, int rows, float slope) {
    UNFLOW_DYNAMIC_LDS(float, red);                 // [rows][C]
    const int quads = C >> 2;
    const int q = threadIdx.x % quads, r = threadIdx.x / quads;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
        const long long p0 = (long long)blockIdx.x * HN_PIX, p1 = min(p0 + HN_PIX, P);
        for (long long p = p0 + r; p < p1; p += (long long)rows * HN_UNROLL) {
            bf4 v[HN_UNROLL], g[HN_UNROLL], h[HN_UNROLL];
#pragma unroll
            for (int u = 0; u < HN_UNROLL; ++u) {
                const long long pp = p + (long long)u * rows;
                if (pp < p1) {
                    v[u] = reinterpret_cast<const bf4*>(y + pp * yps)[q];
                    g[u] = reinterpret_cast<const bf4*>(gout + pp * gps)[q];
                    if (TWO) h[u] = reinterpret_cast<const bf4*>(gout2 + pp * gps2)[q];
                }
            }
#pragma unroll
            for (int u = 0; u < HN_UNROLL; ++u) {
                const long long pp = p + (long long)u * rows;
                if (pp < p1) {
                    bf4 t;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float f = bf2f(g[u].v[k]);
                        if (TWO) f += bf2f(h[u].v[k]);
                        f = bf2f(v[u].v[k]) > 0.f ? f : f * slope;
                        t.v[k] = f2bf(f);
                        acc[k] += bf2f(t.v[k]);          // the rounded value, as in the NCHW kernel
                    }
                    reinterpret_cast<bf4*>(gin + pp * C)[q] = t;
                }
            }
        }
        *reinterpret_cast<float4*>(red + r * C + q * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int k = 0; k < rows; ++k) s += red[k * C + c];
        partials[(size_t)c * gridDim.x + blockIdx.x] = s;
    }
}

}  // namespace

extern "C" int unflow_bias_leaky_fwd_bf16(uint16_t* y, const float* bias, int N, int C, int H, int W, float slope,
                                          void* stream) {
    UNFLOW_REQUIRE(y && bias && N > 0 && C > 0 && H > 0 && W > 0 && N <= 65535 && C <= 65535);
    const int HW = H * W;
    UNFLOW_LAUNCH(bias_leaky_fwd_bf16_kernel, dim3(ceil_div(HW, EWH_TILE), C, N), dim3(256), 0, (hipStream_t)stream,
                       y, bias, C, HW, slope);
    return unflow_launch_status();
}

extern "C" int unflow_bias_leaky_bwd2_bf16(const uint16_t* y, const uint16_t* gout, long long gout_stride,
                                           const uint16_t* gout2, long long gout2_stride, uint16_t* gin, float* gbias,
                                           float* partials, int N, int C, int H, int W, float slope, void* stream) {
    UNFLOW_REQUIRE(y && gout && gin && partials && N > 0 && C > 0 && H > 0 && W > 0 && N <= 65535 && C <= 65535);
    const int HW = H * W, nchunk = ceil_div(HW, EWH_TILE);
    UNFLOW_REQUIRE(gout_stride >= (long long)C * HW && (!gout2 || gout2_stride >= (long long)C * HW));
    if ((HW & 7) == 0)      // the 16-byte path needs every sample block on a 16-byte boundary
        UNFLOW_REQUIRE((gout_stride & 7) == 0 && ((size_t)gout & 15) == 0 &&
                       (!gout2 || ((gout2_stride & 7) == 0 && ((size_t)gout2 & 15) == 0)));
    hipStream_t s = (hipStream_t)stream;
    if (gout2)
        UNFLOW_LAUNCH(bias_leaky_bwd_bf16_kernel<true>, dim3(nchunk, C, N), dim3(256), 0, s, y, gout, gout_stride, gout2,
                           gout2_stride, gin, partials, C, HW, slope);
    else
        UNFLOW_LAUNCH(bias_leaky_bwd_bf16_kernel<false>, dim3(nchunk, C, N), dim3(256), 0, s, y, gout, gout_stride, gout2,
                           gout2_stride, gin, partials, C, HW, slope);
    if (gbias) UNFLOW_LAUNCH(bias_grad_finalize_bf16_kernel, dim3(C), dim3(256), 0, s, partials, N * nchunk, gbias);
    return unflow_launch_status();
}

// channels-last twins: y / gin dense [P][C] bf16 (P = N*H*W), C a multiple of 4 and <= 1024; scratch as unflow_bias_leaky_partials_nhwc
static inline int hn_rows(int C) { const int r = 256 / (C >> 2); return r < 1 ? 1 : r; }

static int launch_fwd_nhwc_bf16(const uint16_t* y, const float* bias, long long P, int C, float slope, uint16_t* dst1, long long ps1,
                                uint16_t* dst2, long long ps2, void* stream) {
    UNFLOW_REQUIRE(y && bias && dst1 && P > 0 && C >= 4 && (C & 3) == 0 && C <= 1024 && (((size_t)y | (size_t)dst1) & 7) == 0 && ((size_t)bias & 15) == 0);
    UNFLOW_REQUIRE(ps1 >= C && (ps1 & 3) == 0 && (!dst2 || (ps2 >= C && (ps2 & 3) == 0 && ((size_t)dst2 & 7) == 0)));
    const long long blocks = (P + HN_PIX - 1) / HN_PIX;
    UNFLOW_REQUIRE(blocks < (1ll << 31));
    UNFLOW_LAUNCH(bias_leaky_fwd_nhwc_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, bias, dst1, ps1, dst2, ps2,
                  P, C, hn_rows(C), slope);
    return unflow_launch_status();
}

extern "C" int unflow_bias_leaky_fwd_nhwc_bf16(uint16_t* y, const float* bias, long long P, int C, float slope, void* stream) {
    return launch_fwd_nhwc_bf16(y, bias, P, C, slope, y, C, nullptr, 0, stream);
}

extern "C" int unflow_bias_leaky_fwd_nhwc_to_bf16(const uint16_t* y, const float* bias, long long P, int C, float slope, uint16_t* dst1,
                                                  long long dst1_pstride, uint16_t* dst2, long long dst2_pstride, void* stream) {
    return launch_fwd_nhwc_bf16(y, bias, P, C, slope, dst1, dst1_pstride, dst2, dst2_pstride, stream);
}

static int launch_bwd_nhwc_bf16(const uint16_t* y, long long yps, const uint16_t* gout, long long gout_pstride, const uint16_t* gout2,
                                long long gout2_pstride, uint16_t* gin, float* gbias, float* partials,
                                long long P, int C, float slope, void* stream) {
    UNFLOW_REQUIRE(y && gout && gin && partials && P > 0 && C >= 4 && (C & 3) == 0 && C <= 1024);
    UNFLOW_REQUIRE(yps >= C && (yps & 3) == 0);
    UNFLOW_REQUIRE(gout_pstride >= C && (gout_pstride & 3) == 0 && (((size_t)gout | (size_t)y | (size_t)gin) & 7) == 0);
    UNFLOW_REQUIRE(!gout2 || (gout2_pstride >= C && (gout2_pstride & 3) == 0 && ((size_t)gout2 & 7) == 0));
    const long long blocks = (P + HN_PIX - 1) / HN_PIX;
    UNFLOW_REQUIRE(blocks * C < (1ll << 31));
    hipStream_t s = (hipStream_t)stream;
    const int rows = hn_rows(C);
    const size_t shmem = (size_t)rows * C * sizeof(float);
    if (gout2)
        UNFLOW_LAUNCH(bias_leaky_bwd_nhwc_bf16_kernel<true>, dim3((unsigned)blocks), dim3(256), shmem, s, y, yps, gout, gout_pstride,
                      gout2, gout2_pstride, gin, partials, P, C, rows, slope);
    else
        UNFLOW_LAUNCH(bias_leaky_bwd_nhwc_bf16_kernel<false>, dim3((unsigned)blocks), dim3(256), shmem, s, y, yps, gout, gout_pstride,
                      gout2, gout2_pstride, gin, partials, P, C, rows, slope);
    if (gbias) UNFLOW_LAUNCH(bias_grad_finalize_bf16_kernel, dim3(C), dim3(256), 0, s, partials, (int)blocks, gbias);
    return unflow_launch_status();
}

extern "C" int unflow_bias_leaky_bwd2_nhwc_bf16(const uint16_t* y, const uint16_t* gout, long long gout_pstride, const uint16_t* gout2,
                                                long long gout2_pstride, uint16_t* gin, float* gbias, float* partials,
                                                long long P, int C, float slope, void* stream) {
    return launch_bwd_nhwc_bf16(y, C, gout, gout_pstride, gout2, gout2_pstride, gin, gbias, partials, P, C, slope, stream);
}

extern "C" int unflow_bias_leaky_bwd2_nhwc_from_bf16(const uint16_t* act, long long act_pstride, const uint16_t* gout, long long gout_pstride,
                                                     const uint16_t* gout2, long long gout2_pstride, uint16_t* gin, float* gbias,
                                                     float* partials, long long P, int C, float slope, void* stream) {
    return launch_bwd_nhwc_bf16(act, act_pstride, gout, gout_pstride, gout2, gout2_pstride, gin, gbias, partials, P, C, slope, stream);
}
