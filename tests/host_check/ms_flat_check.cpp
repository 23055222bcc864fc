// Runs the FLAT `_ms` kernels of csrc/multiscale.h ON THE BUILD HOST (test infrastructure; tests/test_kernels_on_host.py builds and runs it).
//
// The kernels whose bodies are plain grid-stride loops -- no LDS, no barrier, no wave shuffle: occlusion weights, |.| backward, masked-mean
// backward, consistency backward, masked image warp forward -- are ordinary C++ once threadIdx / blockIdx / blockDim / gridDim are
// variables.  This program compiles the very definitions the library ships -- the C entries (csrc/ms_flat_*_entries.h: argument checks,
// per-scale pointer arithmetic, launch geometry), the kernels (csrc/ms_flat_photo.h, ms_flat_warp.h: table structs, UNFLOW_MS_PROLOGUE /
// ms_locate / ms_grid_add, the hook-up of a scale's arguments to the body's parameter names) and the bodies (csrc/bodies/*.inc) -- with g++, executes every workgroup and lane of a launch in a loop, and does the same with the single-scale kernels (the same body files
// behind the real coordinates).  It checks that the one launch over three scales leaves bit for bit what three single-scale launches
// leave, and writes the results to a file that the Python test compares with the oracle.
//
//   g++ -O1 -std=c++17 -ffp-contract=off -DUNFLOW_HOST_CHECK -I unopticalflow_amd/csrc tests/host_check/ms_flat_check.cpp -o ms_flat_check
//   ms_flat_check out.bin        (inputs: v(i) = float((i * 2654435761) mod 2001) / 1000 - 1, as tools/capi_bench.cpp)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

// ---- the few names of the HIP dialect the flat kernels use ----
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static dim3 threadIdx, blockIdx, blockDim, gridDim;          // set by launch(); an `_ms` kernel shadows blockIdx / gridDim with its VDim locals
using std::max;
using std::min;
static inline float __fadd_rn(float a, float b) { return a + b; }      // (built with -ffp-contract=off: no fused forms sneak in)
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }      // photo.hip
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }                          // common.h
// ... and what the C entries use: argument checks as in common.h, a launch that runs every workgroup and lane here
#define UNFLOW_EINVAL (-22)
#define UNFLOW_REQUIRE(cond) do { if (!(cond)) return UNFLOW_EINVAL; } while (0)
typedef void* hipStream_t;
static inline int unflow_launch_status() { return 0; }
template <class K, class... A> static void launch(K kernel, dim3 grid, dim3 block, A... args);
#define UNFLOW_LAUNCH(kernel, grid, block, shmem, stream, ...) launch(kernel, grid, block, __VA_ARGS__)

#include "multiscale.h"
#include "warp_taps.h"
namespace {
#include "ms_flat_photo.h"
#include "ms_flat_warp.h"

// ---- the single-scale kernels: the same body files behind the real coordinates (photo.hip / warp.hip, verbatim signatures) ----
void occ_weight_fwd_kernel(const float* __restrict__ img, const float* __restrict__ from_l, const float* __restrict__ from_r,
                           float* __restrict__ diff_l, float* __restrict__ diff_r, float* __restrict__ w_bwd, float* __restrict__ w_fwd,
                           uint8_t* __restrict__ valid_bwd, uint8_t* __restrict__ valid_fwd, int B, int HW) {
#include "bodies/occ_weight_fwd.inc"
}
void absdiff_bwd_kernel(const float* __restrict__ img, const float* __restrict__ from, const float* __restrict__ gdiff,
                        float* __restrict__ gfrom, int B, int HW, int img_b) {
#include "bodies/absdiff_bwd.inc"
}
void masked_mean_bwd_kernel(const float* __restrict__ w, const float* __restrict__ sums, const float* __restrict__ gloss,
                            float* __restrict__ gdiff, int B, int HW) {
#include "bodies/masked_mean_bwd.inc"
}
void consis_bwd_kernel(const float* __restrict__ ff, const float* __restrict__ fb, const float* __restrict__ w_fwd,
                       const float* __restrict__ sums, const float* __restrict__ gloss, float* __restrict__ gflow, int B, int HW) {
#include "bodies/consis_bwd.inc"
}
template <int NY, bool MASKED>
void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow, float* __restrict__ out, uint8_t* __restrict__ mask,
                     int C, int H, int W, int ac) {
#include "bodies/warp_fwd.inc"
}
inline int flat_blocks(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 8192 ? (b ? b : 1) : 8192); }   // photo.hip
}  // namespace

// the C entries themselves (photo.hip / warp.hip include the same two headers at file scope)
#define UNFLOW_MS_REQUIRE_N(n) UNFLOW_REQUIRE((n) > 0 && (n) <= MS_MAX)
#include "ms_flat_photo_entries.h"
#include "ms_flat_warp_entries.h"

// every workgroup and every lane of a launch, one after the other (legal for kernels without barriers or cross-lane traffic)
template <class K, class... A>
static void launch(K kernel, dim3 grid, dim3 block, A... args) {
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned ty = 0; ty < block.y; ++ty) for (unsigned tx = 0; tx < block.x; ++tx) { threadIdx = dim3(tx, ty, 0); kernel(args...); }
    }
}

static float v(size_t i) { return (float)((i * 2654435761ull) % 2001ull) / 1000.f - 1.f; }
static std::vector<float> gen(size_t n, size_t seed, float scale = 1.f, float shift = 0.f) {
    std::vector<float> a(n);
    for (size_t i = 0; i < n; ++i) a[i] = v(i + seed) * scale + shift;
    return a;
}
static int failures = 0;
static void same(const char* what, int s, const void* a, const void* b, size_t bytes) {
    if (memcmp(a, b, bytes) != 0) { printf("MISMATCH %s scale %d\n", what, s); ++failures; }
}

int main(int argc, char** argv) {
    const int n = 3, B = 2, C = 3;
    static const int Hs[n] = {12, 6, 5}, Ws[n] = {70, 36, 17};                // (70 = one full 64-lane row segment + a ragged one; 17 x 5 = odd everything)
    FILE* f = argc > 1 ? fopen(argv[1], "wb") : nullptr;
    auto dump = [&](const std::vector<float>& a) { if (f) fwrite(a.data(), 4, a.size(), f); };
    auto dump8 = [&](const std::vector<uint8_t>& a) { if (f) fwrite(a.data(), 1, a.size(), f); };
    struct Scale {
        int H, W, HW;
        std::vector<float> img, warped, flow_img, gdiff, sums, gloss, ff, fb, wgt_in;                   // inputs
        std::vector<float> diff, wgt, gfrom, gmm, gflow, wout, diff1, wgt1, gfrom1, gmm1, gflow1, wout1;   // outputs: `_ms` launch / single-scale launches
        std::vector<uint8_t> mask, mask1;
    } S[n];
    for (int s = 0; s < n; ++s) {
        Scale& q = S[s];
        q.H = Hs[s]; q.W = Ws[s]; q.HW = q.H * q.W;
        const size_t hw = q.HW;
        q.img = gen(B * 3 * hw, 11 + s, 0.5f, 0.5f);                      // [B,3,H,W] in [0, 1]
        q.warped = gen(2 * B * 3 * hw, 101 + s, 0.5f, 0.5f);              // [2B,3,H,W] = (from_l | from_r)
        for (size_t i = 0; i < hw / 3; ++i) q.warped[i] = q.warped[hw + i] = q.warped[2 * hw + i] = 0.f;      // an all-zero (invalid) region, sample 0
        q.flow_img = gen(B * 2 * hw, 201 + s, 4.f);                       // flows that leave the image here and there
        q.gdiff = gen(2 * B * hw, 301 + s);
        q.sums = gen(2 * B * 2, 401 + s, 0.25f, 0.75f * hw);              // {sum diff*w, sum w} per sample: positive
        q.gloss = gen(2 * B, 501 + s);
        q.ff = gen(B * 2 * hw, 601 + s, 3.f); q.fb = gen(B * 2 * hw, 701 + s, 3.f);
        q.wgt_in = gen(B * hw, 801 + s, 1.f, 1.f);                        // weights in [0, 2]
        q.diff.assign(2 * B * hw, -7.f); q.wgt = q.diff; q.diff1 = q.diff; q.wgt1 = q.diff;
        q.gfrom.assign(2 * B * 3 * hw, -7.f); q.gfrom1 = q.gfrom;
        q.gmm.assign(2 * B * hw, -7.f); q.gmm1 = q.gmm;
        q.gflow.assign(B * 2 * hw, -7.f); q.gflow1 = q.gflow;
        q.wout.assign(B * 3 * hw, -7.f); q.wout1 = q.wout;
        q.mask.assign(B * hw, 9); q.mask1 = q.mask;
    }
    // ---- one launch over the scales, through the C entries as the ctypes binding calls them (host arrays of per-scale pointers / sizes)
    const float *img[n], *warped[n], *gdiff[n], *sums[n], *gloss[n], *ff[n], *fb[n], *win[n], *wgt_c[n], *flow_img[n];
    float *diff[n], *wgt[n], *gfrom[n], *gmm[n], *gflow[n], *wout[n];
    uint8_t* mask[n];
    for (int s = 0; s < n; ++s) {
        Scale& q = S[s];
        img[s] = q.img.data(); warped[s] = q.warped.data(); gdiff[s] = q.gdiff.data(); sums[s] = q.sums.data(); gloss[s] = q.gloss.data();
        ff[s] = q.ff.data(); fb[s] = q.fb.data(); win[s] = q.wgt_in.data(); wgt_c[s] = q.wgt.data(); flow_img[s] = q.flow_img.data();
        diff[s] = q.diff.data(); wgt[s] = q.wgt.data(); gfrom[s] = q.gfrom.data(); gmm[s] = q.gmm.data(); gflow[s] = q.gflow.data();
        wout[s] = q.wout.data(); mask[s] = q.mask.data();
    }
    if (unflow_occ_weight_fwd_ms(n, img, warped, diff, wgt, Hs, Ws, B, nullptr) != 0) return 2;
    if (unflow_absdiff_bwd_ms(n, img, warped, gdiff, gfrom, Hs, Ws, 2 * B, B, nullptr) != 0) return 2;
    if (unflow_masked_mean_bwd_ms(n, wgt_c, sums, gloss, gmm, Hs, Ws, 2 * B, nullptr) != 0) return 2;
    if (unflow_consis_bwd_ms(n, ff, fb, win, sums, gloss, gflow, Hs, Ws, B, nullptr) != 0) return 2;
    // (and what they must refuse, as tests/test_abi.py asks the built library)
    if (unflow_occ_weight_fwd_ms(0, img, warped, diff, wgt, Hs, Ws, B, nullptr) != UNFLOW_EINVAL || unflow_occ_weight_fwd_ms(5, img, warped, diff, wgt, Hs, Ws, B, nullptr) != UNFLOW_EINVAL ||
        unflow_absdiff_bwd_ms(n, img, warped, gdiff, gfrom, Hs, Ws, 3, 2, nullptr) != UNFLOW_EINVAL) { printf("argument checks\n"); ++failures; }
    for (int ac = 0; ac < 2; ++ac) {
        if (unflow_warp_fwd_ms(n, img, flow_img, wout, mask, Hs, Ws, B, C, ac, nullptr) != 0) return 2;
        for (int s = 0; s < n; ++s) {
            Scale& q = S[s];
            launch(warp_fwd_kernel<1, true>, dim3(ceil_div(q.W, 64), q.H, B), dim3(64, 1), q.img.data(), q.flow_img.data(), q.wout1.data(), q.mask1.data(), C, q.H, q.W, ac);
            same(ac ? "warp fwd (align_corners)" : "warp fwd", s, q.wout.data(), q.wout1.data(), q.wout.size() * 4);
            same(ac ? "warp mask (align_corners)" : "warp mask", s, q.mask.data(), q.mask1.data(), q.mask.size());
            dump(q.wout); dump8(q.mask);                                   // (both align_corners settings go to the file)
        }
    }
    // ---- the same work as single-scale launches, and the comparison
    for (int s = 0; s < n; ++s) {
        Scale& q = S[s];
        const size_t hw = q.HW;
        launch(occ_weight_fwd_kernel, dim3(flat_blocks((size_t)B * hw)), dim3(256), q.img.data(), q.warped.data(), q.warped.data() + (size_t)B * 3 * hw,
               q.diff1.data(), q.diff1.data() + (size_t)B * hw, q.wgt1.data(), q.wgt1.data() + (size_t)B * hw, (uint8_t*)nullptr, (uint8_t*)nullptr, B, q.HW);
        launch(absdiff_bwd_kernel, dim3(flat_blocks((size_t)2 * B * 3 * hw)), dim3(256), q.img.data(), q.warped.data(), q.gdiff.data(), q.gfrom1.data(), 2 * B, q.HW, B);
        launch(masked_mean_bwd_kernel, dim3(flat_blocks((size_t)2 * B * hw)), dim3(256), q.wgt1.data(), q.sums.data(), q.gloss.data(), q.gmm1.data(), 2 * B, q.HW);
        launch(consis_bwd_kernel, dim3(flat_blocks((size_t)B * hw)), dim3(256), q.ff.data(), q.fb.data(), q.wgt_in.data(), q.sums.data(), q.gloss.data(), q.gflow1.data(), B, q.HW);
        same("occlusion diff", s, q.diff.data(), q.diff1.data(), q.diff.size() * 4);
        same("occlusion weight", s, q.wgt.data(), q.wgt1.data(), q.wgt.size() * 4);
        same("|.| backward", s, q.gfrom.data(), q.gfrom1.data(), q.gfrom.size() * 4);
        same("masked-mean backward", s, q.gmm.data(), q.gmm1.data(), q.gmm.size() * 4);
        same("consistency backward", s, q.gflow.data(), q.gflow1.data(), q.gflow.size() * 4);
        for (const auto* a : {&q.diff, &q.wgt, &q.gfrom, &q.gmm, &q.gflow})
            for (float x : *a) if (x == -7.f) { printf("UNWRITTEN element, scale %d\n", s); ++failures; break; }
        dump(q.diff); dump(q.wgt); dump(q.gfrom); dump(q.gmm); dump(q.gflow);
        // the validity masks of compute_diff_weight (model_flow_paper.py:111-112: bit-exact target #2) come out of the single-scale kernel only
        std::vector<uint8_t> vb((size_t)B * hw, 9), vf((size_t)B * hw, 9);
        launch(occ_weight_fwd_kernel, dim3(flat_blocks((size_t)B * hw)), dim3(256), q.img.data(), q.warped.data(), q.warped.data() + (size_t)B * 3 * hw,
               q.diff1.data(), q.diff1.data() + (size_t)B * hw, q.wgt1.data(), q.wgt1.data() + (size_t)B * hw, vb.data(), vf.data(), B, q.HW);
        dump8(vb); dump8(vf);
    }
    if (f) fclose(f);
    printf("%s: %d mismatches\n", failures ? "FAILED" : "OK", failures);
    return failures ? 1 : 0;
}
