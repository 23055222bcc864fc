// Adam over all parameter tensors of the model in ONE launch (the train step of the reference: torch.optim.Adam with default
// betas / eps, lr 1e-4, train.py:39,151).
//
// torch's fused Adam (multi_tensor_apply) walks the 98 tensors in chunks of 65,536 elements on 512-thread blocks: ~180 blocks for
// 5.1 M parameters, less than one per CU, 137 us for 144 MB (profiles/r3_aten_sources_fp32.txt).  Here a block owns a chunk of
// 4,096 elements of one tensor (16-byte accesses: four streams in, three out), ~1,300 blocks; the gradients' addresses travel as
// kernel arguments (they change with every eager backward pass), parameter / moment addresses and the chunk map sit in a device
// table written once.  The step counters live on the device (hipGraph replay): every block reads step[0]; a one-block launch behind the
// update writes step + 1 to every tensor's counter.
#include "common.h"

namespace {

constexpr int ADAM_MAX_TENSORS = 128;
constexpr int ADAM_CHUNK = 4096;

struct AdamGrads { const float* g[ADAM_MAX_TENSORS]; };

__global__ __launch_bounds__(256) void adam_multi_kernel(const unflow_adam_slot* __restrict__ slots, const int* __restrict__ chunk_map,
                                                         AdamGrads grads, const float* __restrict__ steps,
                                                         float lr, float beta1, float beta2, float eps) {
    const int t = chunk_map[2 * blockIdx.x], ck = chunk_map[2 * blockIdx.x + 1];
    const float step = steps[0] + 1.0f;
    const float bc1 = 1.0f - powf(beta1, step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, step));
    const float step_size = lr / bc1;
    const unflow_adam_slot s = slots[t];
    const float* __restrict__ g = grads.g[t];
    if (g != nullptr) {
        const long long e0 = (long long)ck * ADAM_CHUNK;
        const long long n = s.numel - e0 < ADAM_CHUNK ? s.numel - e0 : ADAM_CHUNK;
        float* p = s.p + e0; float* m = s.m + e0; float* v = s.v + e0; g += e0;
        auto upd = [&](float& pv, float gv, float& mv, float& vv) {
            mv = mv + (gv - mv) * (1.0f - beta1);                   // lerp(exp_avg, grad, 1 - beta1)
            vv = beta2 * vv + (1.0f - beta2) * gv * gv;
            const float denom = sqrtf(vv) / bc2_sqrt + eps;
            pv = pv - step_size * mv / denom;
        };
        if (((((size_t)p) | ((size_t)m) | ((size_t)v) | ((size_t)g)) & 15) == 0) {
#pragma unroll
            for (int k = 0; k < ADAM_CHUNK / 1024; ++k) {
                const long long e = (long long)(k * 256 + threadIdx.x) * 4;
                if (e + 3 < n) {
                    float4 pv = *reinterpret_cast<float4*>(p + e), mv = *reinterpret_cast<float4*>(m + e), vv = *reinterpret_cast<float4*>(v + e);
                    const float4 gv = *reinterpret_cast<const float4*>(g + e);
                    upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
                    *reinterpret_cast<float4*>(p + e) = pv; *reinterpret_cast<float4*>(m + e) = mv; *reinterpret_cast<float4*>(v + e) = vv;
                } else {
                    for (long long i = e; i < n; ++i) upd(p[i], g[i], m[i], v[i]);
                }
            }
        } else {
            for (long long i = threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
        }
    }
}

// steps[i] <- steps[0] + 1 for every tensor that had a gradient: its own one-block launch BEHIND the update (a "last block to finish
// advances the counters" inside the update kernel needs a device-scope fence per block, and on this chip that fence writes the XCD's
// dirty L2 lines back -- the 61 MB the kernel has just written: measured 253 us for the update instead of ~40)
__global__ __launch_bounds__(128) void adam_advance_kernel(AdamGrads grads, float* __restrict__ steps, int ntensors) {
    const float step = steps[0] + 1.0f;
    __syncthreads();
    for (int i = threadIdx.x; i < ntensors; i += 128)
        if (grads.g[i] != nullptr) steps[i] = step;
}

}  // namespace

extern "C" int unflow_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int unflow_adam_multi(const unflow_adam_slot* slots, const int* chunk_map, int nchunks, const void* const* grads, int ntensors,
                                 float* steps, float lr, float beta1, float beta2, float eps, void* stream) {
    UNFLOW_REQUIRE(slots && chunk_map && grads && steps && nchunks > 0 && ntensors > 0 && ntensors <= ADAM_MAX_TENSORS);
    AdamGrads a;
    for (int i = 0; i < ADAM_MAX_TENSORS; ++i) a.g[i] = i < ntensors ? (const float*)grads[i] : nullptr;
    UNFLOW_LAUNCH(adam_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, slots, chunk_map, a, (const float*)steps,
                  lr, beta1, beta2, eps);
    UNFLOW_LAUNCH(adam_advance_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, a, steps, ntensors);
    return unflow_launch_status();
}
