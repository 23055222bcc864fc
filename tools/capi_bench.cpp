// Standalone use of the C ABI (include/unflow_hip.h) without Python or torch: times the cost-volume
// forward/backward at one pyramid-level shape with hipEvents and prints algorithmic GB/s.  Also the
// program to put after `rocprofv3 ... --` when a Python-free profile is wanted.
//
//   hipcc -O2 --offload-arch=gfx950 tools/capi_bench.cpp -Iinclude -Lunopticalflow_amd -lunflow_hip \
//         -Wl,-rpath,$PWD/unopticalflow_amd -o tools/capi_bench
//   tools/capi_bench [B C H W d iters]        (default 16 32 64 208 4 50 = level 2 of the 832x256, B=8 step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "unflow_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    int B = 16, C = 32, H = 64, W = 208, d = 4, iters = 50;
    if (argc >= 6) { B = atoi(argv[1]); C = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); d = atoi(argv[5]); }
    if (argc >= 7) iters = atoi(argv[6]);
    const int DD = (2 * d + 1) * (2 * d + 1);
    const size_t nf = (size_t)B * C * H * W, nc = (size_t)B * DD * H * W;
    std::vector<float> h(nf > nc ? nf : nc);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    float *f1, *f2, *cv, *g, *gf1, *gf2;
    CK(hipMalloc(&f1, nf * 4)); CK(hipMalloc(&f2, nf * 4)); CK(hipMalloc(&gf1, nf * 4)); CK(hipMalloc(&gf2, nf * 4));
    CK(hipMalloc(&cv, nc * 4)); CK(hipMalloc(&g, nc * 4));
    CK(hipMemcpy(f1, h.data(), nf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(f2, h.data() + 7, (nf - 7) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, h.data(), nc * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (unflow_abi_version() != 1) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 3; ++i) {
            int rc = pass ? unflow_corr_bwd(f1, f2, g, gf1, gf2, B, C, H, W, d, s) : unflow_corr_fwd(f1, f2, cv, B, C, H, W, d, s);
            if (rc) { fprintf(stderr, "launch failed: %d\n", rc); return 1; }
        }
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i)
            pass ? unflow_corr_bwd(f1, f2, g, gf1, gf2, B, C, H, W, d, s) : unflow_corr_fwd(f1, f2, cv, B, C, H, W, d, s);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = 4.0 * B * H * W * ((pass ? 4.0 : 2.0) * C + DD);
        printf("corr %s [%d,%d,%d,%d] d=%d: %.2f us/launch, %.0f GB/s algorithmic\n", pass ? "bwd" : "fwd", B, C, H, W, d,
               ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e9);
    }
    std::vector<float> out(4);
    CK(hipMemcpy(out.data(), cv, 16, hipMemcpyDeviceToHost));
    printf("cv[0..3] = %g %g %g %g\n", out[0], out[1], out[2], out[3]);
    return 0;
}
