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
    UNFLOW_REQUIRE(y && gout && gin && gbias && partials && N > 0 && C > 0 && H > 0 && W > 0 && N <= 65535 && C <= 65535);
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
    UNFLOW_LAUNCH(bias_grad_finalize_bf16_kernel, dim3(C), dim3(256), 0, s, partials, N * nchunk, gbias);
    return unflow_launch_status();
}
