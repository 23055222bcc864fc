// The cost-volume backward on the matrix cores -- csrc/corr_mfma.h (shipped, on request: UNFLOW_CORR_BWD_MFMA) and, built with -DWITH_PROTO, the
// never-run pixel-pair prototype tools/proto/corr_mfma2.h -- compiled for the build host with the ROCm clang++ and EXECUTED with lanes as fibers
// (tests/host_check/hip_on_host.h; TEST INFRASTRUCTURE, tests/test_kernels_on_host.py).  The matrix instruction is a function there that
// gathers the 64 lanes' A / B fragments by the lane layouts of the CDNA4 ISA and forms the 16 x 16 x 32 product in fp32; buffer loads /
// stores range-check like the hardware (the kernels' predicate); LDS is a static array.
//
//   mfma_check in.bin out.bin
// in : int32 ncases; per case int32 R, B, C, H, W, rows, which (0: corr_mfma.h, 1: corr_mfma2.h); float f1[B,C,H,W], f2[B,C,H,W], g[B,(2R+1)^2,H,W]
// out: per case gf1[B,C,H,W], gf2[B,C,H,W]
#include "corr_mfma.h"
#ifdef WITH_PROTO
#include "corr_mfma2.h"
#endif

UnflowTimingArm& unflow_timing_arm() { static UnflowTimingArm arm = {nullptr, nullptr, false}; return arm; }

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    FILE* o = fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int ncases;
    if (fread(&ncases, 4, 1, f) != 1) return 2;
    for (int k = 0; k < ncases; ++k) {
        int h[7];
        if (fread(h, 4, 7, f) != 7) return 2;
        const int R = h[0], B = h[1], C = h[2], H = h[3], W = h[4], rows = h[5], which = h[6];
        const int DD = 2 * R + 1;
        const size_t n = (size_t)B * C * H * W, ng = (size_t)B * DD * DD * H * W;
        std::vector<float> f1(n), f2(n), g(ng), gf1(n, -7.f), gf2(n, -7.f);
        if (fread(f1.data(), 4, n, f) != n || fread(f2.data(), 4, n, f) != n || fread(g.data(), 4, ng, f) != ng) return 2;
        int rc = -1;
        if (which == 0 && R == 4) rc = launch_bwd_mf<4, 2, 1, 1>(f1.data(), f2.data(), g.data(), gf1.data(), gf2.data(), B, C, H, W, rows, nullptr);
        if (which == 0 && R == 8) rc = launch_bwd_mf<8, 2, 2, 1>(f1.data(), f2.data(), g.data(), gf1.data(), gf2.data(), B, C, H, W, rows, nullptr);
#ifdef WITH_PROTO
        if (which == 1 && R == 4) rc = launch_bwd_mf2<4, 2, 1>(f1.data(), f2.data(), g.data(), gf1.data(), gf2.data(), B, C, H, W, rows, nullptr);
        if (which == 1 && R == 8) rc = launch_bwd_mf2<8, 2, 1>(f1.data(), f2.data(), g.data(), gf1.data(), gf2.data(), B, C, H, W, rows, nullptr);
#endif
        if (rc != 0) { printf("case %d: rc %d\n", k, rc); return 1; }
        fwrite(gf1.data(), 4, n, o); fwrite(gf2.data(), 4, n, o);
    }
    fclose(f); fclose(o);
    printf("OK\n");
    return 0;
}
