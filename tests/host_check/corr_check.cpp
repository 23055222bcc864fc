// The fp32 cost-volume kernels -- csrc/corr.hip (tile kernels, the LDS-DMA ring kernels with their hand-issued LDS reads, the group-split and
// row-streamed backward, the whole-map backward, the per-element kernels) and the fused warp + cost volume of csrc/warp_corr.hip -- compiled
// for the build host and EXECUTED with lanes as fibers, through the library's own C entry points and its own dispatch by shape (TEST
// INFRASTRUCTURE: tests/test_kernels_on_host.py links this file with corr.hip, warp_corr.hip and warp.hip built with -DUNFLOW_HOST_CHECK).  On the host LDS-DMA is a copy that has landed when the call returns and the counted waits mean nothing: what is
// checked is WHAT is computed and where every access lands (the program runs once more under AddressSanitizer), not when.
//
//   corr_check in.bin out.bin
// in : int32 ncases; per case int32 kind (0: cost volume, 1: fused warp + cost volume), d, B, C, H, W, align_corners, backward mode
//      (unflow_corr_bwd_ex's arithmetic); float f1[B,C,H,W], f2[B,C,H,W], flow[B,2,H,W] (kind 1 only), g[B,(2d+1)^2,H,W]
// out: per case cv[B,(2d+1)^2,H,W], gf1[B,C,H,W], gf2[B,C,H,W], gflow[B,2,H,W] (kind 1 only)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"

UnflowTimingArm& unflow_timing_arm() { static UnflowTimingArm arm = {nullptr, nullptr, false}; return arm; }      // (photo.hip's, not linked here)

// 16-byte aligned blocks (what the LDS-DMA paths require and torch's allocator provides), exactly sized: red zones start at the last float
static float* block(size_t n) {
    void* p = nullptr;
    if (posix_memalign(&p, 16, n * 4) != 0) exit(2);
    return (float*)p;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    FILE* o = fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int ncases;
    if (fread(&ncases, 4, 1, f) != 1) return 2;
    for (int k = 0; k < ncases; ++k) {
        int h[8];
        if (fread(h, 4, 8, f) != 8) return 2;
        const int kind = h[0], d = h[1], B = h[2], C = h[3], H = h[4], W = h[5], ac = h[6], mode = h[7];
        const int DD = 2 * d + 1;
        const size_t n = (size_t)B * C * H * W, ng = (size_t)B * DD * DD * H * W, nf = (size_t)B * 2 * H * W;
        float *f1 = block(n), *f2 = block(n), *flow = block(nf), *g = block(ng), *cv = block(ng), *gf1 = block(n), *gf2 = block(n), *gflow = block(nf),
              *scratch = block(2 * n);
        if (fread(f1, 4, n, f) != n || fread(f2, 4, n, f) != n) return 2;
        if (kind == 1 && fread(flow, 4, nf, f) != nf) return 2;
        if (fread(g, 4, ng, f) != ng) return 2;
        for (size_t i = 0; i < ng; ++i) cv[i] = -7.f;
        for (size_t i = 0; i < n; ++i) gf1[i] = gf2[i] = -7.f;
        for (size_t i = 0; i < nf; ++i) gflow[i] = -7.f;
        int rc;
        if (kind == 0) {
            rc = unflow_corr_fwd(f1, f2, cv, B, C, H, W, d, nullptr);
            if (rc == 0) rc = unflow_corr_bwd_ex(f1, f2, g, gf1, gf2, B, C, H, W, d, mode, nullptr);
        } else {
            if (!unflow_warp_corr_supported(C, H, W, d)) { printf("case %d: not served\n", k); return 1; }
            rc = unflow_warp_corr_fwd(f1, f2, flow, cv, B, C, H, W, d, ac, nullptr);
            if (rc == 0) rc = unflow_warp_corr_bwd(f1, f2, flow, g, gf1, gf2, gflow, scratch, B, C, H, W, d, ac, nullptr);
        }
        if (rc != 0) { printf("case %d: rc %d\n", k, rc); return 1; }
        fwrite(cv, 4, ng, o); fwrite(gf1, 4, n, o); fwrite(gf2, 4, n, o);
        if (kind == 1) fwrite(gflow, 4, nf, o);
        free(f1); free(f2); free(flow); free(g); free(cv); free(gf1); free(gf2); free(gflow); free(scratch);
    }
    fclose(f); fclose(o);
    printf("OK\n");
    return 0;
}
