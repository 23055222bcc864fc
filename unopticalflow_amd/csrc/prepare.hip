// Input stage of the flow train step on the device: what
// KITTI_Prepared.__getitem__ does after the PNG decode (core/dataset/kitti_prepared.py:63-90,145-148)
// -- split the stacked triplet into three frames, cv2.resize each to img_hw, optional horizontal
// flip, / 255.0, HWC -> CHW float -- as one HBM-bound byte kernel over a whole batch of decoded
// images, so the host only decodes PNGs and ships uint8 (4.2 MB per KITTI triplet instead of the
// 7.7 MB float tensor, and no CPU resize).
//
// The resize is OpenCV's 8-bit INTER_LINEAR (opencv-python==4.1.1.26, requirements.txt:13) in its
// published fixed-point form: 11-bit coefficients rounded half-to-even from float fractions, int32
// horizontal pass, vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  All integer
// work after the two coefficient roundings: bit-exact against oracle/prepare_cpu.py.
//
// One lane produces 4 adjacent output pixels of one row for the 3 channels (three 16-byte stores, one per
// plane); source bytes are gathered through L2/TA (each source byte is touched by ~1.5 lanes).
#include "common.h"

namespace {

struct Taps { int i0, i1, c0, c1; };

__device__ __forceinline__ Taps column_taps(int d, double scale, int n) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    Taps t;
    t.i0 = s; t.i1 = min(s + 1, n - 1);
    t.c0 = (int)rintf((1.f - f) * 2048.f); t.c1 = (int)rintf(f * 2048.f);
    return t;
}

__device__ __forceinline__ Taps row_taps(int d, double scale, int n) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    Taps t;
    t.i0 = min(max(s, 0), n - 1); t.i1 = min(max(s + 1, 0), n - 1);
    t.c0 = (int)rintf((1.f - f) * 2048.f); t.c1 = (int)rintf(f * 2048.f);
    return t;
}

// src: all decoded images back to back; offsets[b] = first byte of image b; dims[2b] = rows (three
// frames of int(rows/3) rows, leftover rows ignored), dims[2b+1] = columns; 3 interleaved channels.
__global__ __launch_bounds__(256) void prepare_triplets_kernel(const unsigned char* __restrict__ src,
                                                               const long long* __restrict__ offsets,
                                                               const int* __restrict__ dims,
                                                               const unsigned char* __restrict__ flip,
                                                               float* __restrict__ dst, int H, int W, int swap_rb) {
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)((double)threadIdx.x / 255.0);          // img / 255.0 in float64, then .float()
    __syncthreads();
    const int b = blockIdx.z, WQ = W >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;                      // over 3 frames x H rows x W/4 quads
    if (q >= 3 * H * WQ) return;
    const int xq = q % WQ, yy = q / WQ, k = yy / H, y = yy - k * H;
    const int rows = dims[2 * b], w = dims[2 * b + 1], h = rows / 3;
    const double scale_x = 1.0 / ((double)W / (double)w), scale_y = 1.0 / ((double)H / (double)h);
    const bool fl = flip != nullptr && flip[b] != 0;
    const Taps ty = row_taps(y, scale_y, h);
    const unsigned char* r0 = src + offsets[b] + (size_t)(k * h + ty.i0) * w * 3;
    const unsigned char* r1 = src + offsets[b] + (size_t)(k * h + ty.i1) * w * 3;
    float o[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = xq * 4 + j;
        const Taps tx = column_taps(fl ? W - 1 - x : x, scale_x, w);     // cv2.flip(img, 1) after the resize
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s0 = (int)r0[tx.i0 * 3 + c] * tx.c0 + (int)r0[tx.i1 * 3 + c] * tx.c1;
            const int s1 = (int)r1[tx.i0 * 3 + c] * tx.c0 + (int)r1[tx.i1 * 3 + c] * tx.c1;
            const int v = (((ty.c0 * (s0 >> 4)) >> 16) + ((ty.c1 * (s1 >> 4)) >> 16) + 2) >> 2;
            o[c][j] = lut[v & 255];
        }
    }
    const size_t plane = (size_t)3 * H * W;
    float* out = dst + (size_t)b * 3 * plane + (size_t)yy * W + xq * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int pc = swap_rb ? 2 - c : c;                              // RGB-decoded source -> cv2's BGR planes
        *reinterpret_cast<float4*>(out + pc * plane) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
    }
}

}  // namespace

extern "C" int unflow_prepare_triplets(const unsigned char* src, const long long* offsets, const int* dims,
                                       const unsigned char* flip, float* dst, int B, int H, int W, int swap_rb,
                                       void* stream) {
    if (!src || !offsets || !dims || !dst || B <= 0 || H <= 0 || W <= 0 || (W & 3)) return UNFLOW_EINVAL;
    const int quads = 3 * H * (W >> 2);
    dim3 grid((quads + 255) / 256, 1, B);
    UNFLOW_LAUNCH(prepare_triplets_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, offsets, dims, flip, dst, H, W, swap_rb);
    return unflow_launch_status();
}
