// The cost-volume entry points of the HOST-EXECUTED test library (tests/host_check/build_hostexec.py -> libunflow_hostexec.so; TEST
// INFRASTRUCTURE: the product never loads it).  The fast fp32 kernels of csrc/corr.hip have no host form (LDS-DMA rings, hand-issued
// ds_read, packed-FMA register layouts), so unflow_corr_fwd / unflow_corr_bwd dispatch here to what does: the any-radius one-lane-per-element
// kernels (csrc/corr_generic.h) and -- where csrc/corr.hip would take them -- the matrix-core backward kernels (csrc/corr_mfma.h,
// csrc/corr_mfma2.h) with the library's own mode switch.
#include "corr_mfma.h"
#include "corr_mfma2.h"
#include <atomic>

namespace {
#include "corr_generic.h"
std::atomic<int> g_mode{0};
bool served(const float* f1, const float* f2, const float* g, const float* gf1, const float* gf2, int B, int C, int H, int W, int R) {      // corr.hip: mfma_served
    return (W % 4) == 0 && (C % 16) == 0 && (long)B * H * W >= 8192 && W >= 16 && H >= 4 * R &&
           ((((size_t)f1 | (size_t)f2 | (size_t)g | (size_t)gf1 | (size_t)gf2) & 15) == 0) && mf_offsets_fit(C, H, W, R);
}
}  // namespace

extern "C" int unflow_corr_set_backward(int mode) {
    if (mode < 0 || mode > 3) return UNFLOW_EINVAL;
    return g_mode.exchange(mode);
}

extern "C" int unflow_corr_fwd(const float* f1, const float* f2, float* cv, int B, int C, int H, int W, int d, void* stream) {
    UNFLOW_REQUIRE(f1 && f2 && cv && B > 0 && C > 0 && H > 0 && W > 0 && d >= 0);
    const size_t n = (size_t)B * (2 * d + 1) * (2 * d + 1) * H * W;
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    UNFLOW_LAUNCH(corr_fwd_generic, dim3(blocks), dim3(256), 0, (hipStream_t)stream, f1, f2, cv, B, C, H, W, d, 1.0f / C);
    return unflow_launch_status();
}

extern "C" int unflow_corr_bwd(const float* f1, const float* f2, const float* gcv, float* gf1, float* gf2, int B, int C, int H, int W, int d, void* stream) {
    UNFLOW_REQUIRE(f1 && f2 && gcv && gf1 && gf2 && B > 0 && C > 0 && H > 0 && W > 0 && d >= 0);
    hipStream_t s = (hipStream_t)stream;
    const int mode = g_mode.load();
    const int rows = H >= 32 ? 16 : 8;
    if (d == 4 && served(f1, f2, gcv, gf1, gf2, B, C, H, W, 4)) {
        if (mode == 2) return launch_bwd_mf<4, 2, 1, 1>(f1, f2, gcv, gf1, gf2, B, C, H, W, rows, s);
        if (mode == 3) return launch_bwd_mf2<4, 2, 1>(f1, f2, gcv, gf1, gf2, B, C, H, W, rows, s);
    }
    if (d == 8 && served(f1, f2, gcv, gf1, gf2, B, C, H, W, 8)) {
        if (mode == 3) return launch_bwd_mf2<8, 2, 1>(f1, f2, gcv, gf1, gf2, B, C, H, W, rows, s);
        if (mode != 1) return launch_bwd_mf<8, 2, 2, 1>(f1, f2, gcv, gf1, gf2, B, C, H, W, rows, s);
    }
    const size_t n = (size_t)B * C * H * W;
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    UNFLOW_LAUNCH(corr_bwd_generic, dim3(blocks), dim3(256), 0, s, f1, f2, gcv, gf1, gf2, B, C, H, W, d, 1.0f / C);
    return unflow_launch_status();
}
