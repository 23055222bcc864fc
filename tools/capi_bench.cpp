// Standalone use of the C ABI (include/unflow_hip.h) without Python or torch: runs the cost-volume forward / backward at one
// pyramid-level shape on a deterministic input, times them with hipEvents, prints algorithmic GB/s and a few values + checksums
// that tests/test_abi.py compares with the CPU oracle.  Also the program to put after `rocprofv3 --kernel-trace --stats --`
// when a Python-free capture of the cost-volume kernels is wanted (tools/gpu_r4.sh corr_capi).
//
//   hipcc -O2 --offload-arch=gfx950 tools/capi_bench.cpp -Iinclude -Lunopticalflow_amd -lunflow_hip \
//         -Wl,-rpath,$PWD/unopticalflow_amd -o tools/capi_bench
//   tools/capi_bench [B C H W d iters [mask.bin]]        (default 16 32 64 208 4 50 = level 2 of the 832x256, B=8 step)
//
// Input (restated by the test in numpy): v(i) = float((i * 2654435761) mod 2001) / 1000 - 1 over 64-bit i;
// f1[i] = v(i), f2[i] = v(i + 7), g[i] = v(i + 3) (the upstream gradient of the backward).
// Then the masked image warp (net_utils.py:47-52, the integer half of the parity bar) of img[i] = (v(i + 11) + 1) / 2, [B,3,H,W],
// by flow[i] = 6 v(i + 5), [B,2,H,W]: the uint8 mask goes to `mask.bin` byte for byte (the test compares it with the oracle's).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "unflow_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float v(size_t i) { return (float)((i * 2654435761ull) % 2001ull) / 1000.f - 1.f; }

int main(int argc, char** argv) {
    int B = 16, C = 32, H = 64, W = 208, d = 4, iters = 50;
    if (argc >= 6) { B = atoi(argv[1]); C = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); d = atoi(argv[5]); }
    if (argc >= 7) iters = atoi(argv[6]);
    if (unflow_abi_version() != UNFLOW_ABI_VERSION) {
        fprintf(stderr, "ABI mismatch: library %d, header %d\n", unflow_abi_version(), UNFLOW_ABI_VERSION);
        return 1;
    }
    const int D = 2 * d + 1, DD = D * D;
    const size_t nf = (size_t)B * C * H * W, nc = (size_t)B * DD * H * W, nmax = nf > nc ? nf : nc;
    std::vector<float> h(nmax + 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = v(i);
    float *f1, *f2, *cv, *g, *gf1, *gf2;
    CK(hipMalloc(&f1, nf * 4)); CK(hipMalloc(&f2, nf * 4)); CK(hipMalloc(&gf1, nf * 4)); CK(hipMalloc(&gf2, nf * 4));
    CK(hipMalloc(&cv, nc * 4)); CK(hipMalloc(&g, nc * 4));
    CK(hipMemcpy(f1, h.data(), nf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(f2, h.data() + 7, nf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, h.data() + 3, nc * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
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
    // values at the centre displacement (dy = dx = 0) of sample 0, row H/2, columns 0..3, and checksums of all three outputs
    std::vector<float> out(nmax);
    CK(hipMemcpy(out.data(), cv, nc * 4, hipMemcpyDeviceToHost));
    const size_t centre = ((size_t)(d * D + d) * H + H / 2) * W;
    printf("cv_centre = %.9g %.9g %.9g %.9g\n", out[centre], out[centre + 1], out[centre + 2], out[centre + 3]);
    double sa = 0; for (size_t i = 0; i < nc; ++i) sa += out[i] < 0 ? -out[i] : out[i];
    printf("sum_abs_cv = %.9g\n", sa);
    CK(hipMemcpy(out.data(), gf1, nf * 4, hipMemcpyDeviceToHost));
    sa = 0; for (size_t i = 0; i < nf; ++i) sa += out[i] < 0 ? -out[i] : out[i];
    printf("sum_abs_gf1 = %.9g\n", sa);
    CK(hipMemcpy(out.data(), gf2, nf * 4, hipMemcpyDeviceToHost));
    sa = 0; for (size_t i = 0; i < nf; ++i) sa += out[i] < 0 ? -out[i] : out[i];
    printf("sum_abs_gf2 = %.9g\n", sa);

    // masked image warp through unflow_warp_fwd: values and the binary mask
    const size_t ni = (size_t)B * 3 * H * W, nfl = (size_t)B * 2 * H * W, nm = (size_t)B * H * W;
    std::vector<float> himg(ni), hfl(nfl);
    for (size_t i = 0; i < ni; ++i) himg[i] = (v(i + 11) + 1.f) / 2.f;
    for (size_t i = 0; i < nfl; ++i) hfl[i] = 6.f * v(i + 5);
    float *img, *fl, *wimg; uint8_t* mask;
    CK(hipMalloc(&img, ni * 4)); CK(hipMalloc(&fl, nfl * 4)); CK(hipMalloc(&wimg, ni * 4)); CK(hipMalloc(&mask, nm));
    CK(hipMemcpy(img, himg.data(), ni * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(fl, hfl.data(), nfl * 4, hipMemcpyHostToDevice));
    CK(hipMemset(mask, 0xff, nm));
    { int rc = unflow_warp_fwd(img, fl, wimg, mask, B, 3, H, W, 0, s); if (rc) { fprintf(stderr, "warp launch failed: %d\n", rc); return 1; } }
    CK(hipStreamSynchronize(s));
    std::vector<uint8_t> hm(nm);
    CK(hipMemcpy(hm.data(), mask, nm, hipMemcpyDeviceToHost));
    CK(hipMemcpy(himg.data(), wimg, ni * 4, hipMemcpyDeviceToHost));
    size_t ones = 0, other = 0;
    for (size_t i = 0; i < nm; ++i) { ones += hm[i] == 1; other += hm[i] > 1; }
    sa = 0; for (size_t i = 0; i < ni; ++i) sa += himg[i] < 0 ? -himg[i] : himg[i];
    printf("mask_ones = %zu\nmask_not_binary = %zu\nsum_abs_warped = %.9g\n", ones, other, sa);
    if (argc >= 8) {
        FILE* f = fopen(argv[7], "wb");
        if (!f || fwrite(hm.data(), 1, nm, f) != nm) { fprintf(stderr, "cannot write %s\n", argv[7]); return 1; }
        fclose(f);
    }
    return 0;
}
