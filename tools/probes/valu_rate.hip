// VALU issue-rate probe (gfx950): cycles per wave-instruction of v_fma_f32 and v_pk_fma_f32, for 1..4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
    v2f a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.0f + i};
    v2f w = v2f{1.0001f, 0.9999f}, b = v2f{0.5f, 0.25f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(w), "v"(b));
                else asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i].x) : "v"(w.x), "v"(b.x));
            }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int waves = 4; waves <= 16; waves *= 2) {          // waves per workgroup = per CU (one workgroup per CU)
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (pk) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
                else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double c = (double)h[7] / (iters * 64.0);
            printf("%s  %2d waves/CU (%d per SIMD): %.2f ticks per wave-instruction per wave -> %.2f ticks of SIMD time each;  kernel %.1f us for %llu ticks -> s_memtime at %.0f MHz (lower bound)\n",
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", waves, waves / 4, c, c / (waves / 4), ms * 1e3, h[7], h[7] / (ms * 1e3));
        }
    return 0;
}
