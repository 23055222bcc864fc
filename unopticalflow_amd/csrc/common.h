// Shared device helpers for the gfx950 kernels of libunflow_hip.so.
#pragma once
#include <stdint.h>
#include "../../include/unflow_hip.h"
#ifdef UNFLOW_HOST_CHECK            // the TEST build (tests/host_check/: kernel sources compiled for the build host, lanes as fibers): that header holds
#include "hip_on_host.h"            // the host form of every name device_forms.h defines -- the one place csrc/ knows about it
#else
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "device_forms.h"
#endif

#define UNFLOW_WAVE 64

#define UNFLOW_REQUIRE(cond) do { if (!(cond)) return UNFLOW_EINVAL; } while (0)

static inline int unflow_launch_status() { return (int)hipGetLastError(); }

// Every kernel of the library is launched through UNFLOW_LAUNCH.  Normally that is hipLaunchKernelGGL.  When the calling
// thread has armed a timing slot (unflow_timing_begin, photo.hip -- bench.py's roofline legs), the launches of the entry
// point carry the slot's events (hipExtLaunchKernelGGL): the command processor stamps `start` when the entry point's FIRST
// kernel begins and `stop` when its LAST one ends, so start -> stop is what a kernel trace reports for them -- without the
// few microseconds of marker packets that hipEventRecord brackets add around a short kernel.
struct UnflowTimingArm { hipEvent_t start, stop; bool started; };
UnflowTimingArm& unflow_timing_arm();             // this thread's armed pair (stop == nullptr: none)
#define UNFLOW_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                        \
    do {                                                                                                              \
        UnflowTimingArm& arm_ = unflow_timing_arm();                                                                  \
        if (arm_.stop) {                                                                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, arm_.started ? nullptr : arm_.start, arm_.stop, 0, __VA_ARGS__); \
            arm_.started = true;                                                                                      \
        } else {                                                                                                      \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                      \
        }                                                                                                             \
    } while (0)

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Bijective remap of the linear workgroup id so that consecutive work items share an XCD (workgroups are dealt round-robin over
// the 8 XCDs, each with its own L2): neighbouring tiles then find each other's halo lines in L2 instead of fetching them from
// HBM once per XCD.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7;          // XCD x gets q (+1 if x < r) workgroups
    const int xcd = lin & 7, k = lin >> 3;
    return xcd * q + (xcd < r ? xcd : r) + k;
}

// Zero-fill as a kernel launch (not hipMemsetAsync): every piece of work of an entry point is then an ordinary
// kernel node when the caller captures the stream into a hipGraph.  n floats, p 4-byte aligned.
__global__ static void unflow_zero_kernel(float* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if ((((size_t)p) & 15) == 0) {
        float4* p4 = reinterpret_cast<float4*>(p);
        const size_t n4 = n >> 2;
        for (size_t k = i; k < n4; k += stride) p4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t k = (n4 << 2) + i; k < n; k += stride) p[k] = 0.f;
    } else {
        for (size_t k = i; k < n; k += stride) p[k] = 0.f;
    }
}
static inline void unflow_zero_async(float* p, size_t n, hipStream_t s) {
    const size_t want = (n / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    UNFLOW_LAUNCH(unflow_zero_kernel, dim3(blocks), dim3(256), 0, s, p, n);
}

// Two buffers in ONE launch (the warp backward's scatter form zeroes gsrc and, with channel groups, gflow: a launch less).
__global__ static void unflow_zero2_kernel(float* __restrict__ p, size_t n, float* __restrict__ q, size_t m) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (((((size_t)p) | ((size_t)q)) & 15) == 0) {
        float4* p4 = reinterpret_cast<float4*>(p);
        float4* q4 = reinterpret_cast<float4*>(q);
        const size_t n4 = n >> 2, m4 = m >> 2;
        for (size_t k = i; k < n4; k += stride) p4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t k = i; k < m4; k += stride) q4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t k = (n4 << 2) + i; k < n; k += stride) p[k] = 0.f;
        for (size_t k = (m4 << 2) + i; k < m; k += stride) q[k] = 0.f;
    } else {
        for (size_t k = i; k < n; k += stride) p[k] = 0.f;
        for (size_t k = i; k < m; k += stride) q[k] = 0.f;
    }
}
static inline void unflow_zero2_async(float* p, size_t n, float* q, size_t m, hipStream_t s) {
    const size_t want = ((n > m ? n : m) / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    UNFLOW_LAUNCH(unflow_zero2_kernel, dim3(blocks), dim3(256), 0, s, p, n, q, m);
}

// 64-lane butterfly sum (DPP/ds_swizzle shuffles, no LDS).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, UNFLOW_WAVE);
    return v;
}

// Sum K values over a 256-thread workgroup; result valid in thread 0.  `red` is K*4 floats of LDS.
template <int K>
__device__ __forceinline__ void block_sum_256(float (&v)[K], float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[wid * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = (red[k] + red[K + k]) + (red[2 * K + k] + red[3 * K + k]);
    }
}

// Per-sample reductions: every workgroup stores its K partial sums, a second one-workgroup-per-
// sample kernel adds them in a fixed order (bitwise reproducible; no float atomics).
// Layout partials[b][blk][K]; blocks per sample = unflow_partials_per_sample(H, W).
#define UNFLOW_RED_TILE 2048   /* pixels per workgroup in the flat per-sample reductions */

__device__ __forceinline__ float sum_partials(const float* p, int n, int K, int k, float* red) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i * K + k];
    s = wave_sum(s);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = s;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
