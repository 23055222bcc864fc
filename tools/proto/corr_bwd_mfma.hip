// PROTOTYPE HARNESS (not part of libunflow_hip.so): the cost-volume backward on the matrix cores (csrc/corr_mfma.h) against the
// shipped fp32 entry point of the same translation unit -- error statistics (the MFMA form is not bit-identical) and times.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude tools/proto/corr_bwd_mfma.hip -o tools/proto/corr_bwd_mfma
//   tools/proto/corr_bwd_mfma [B C H W [iters]]
#include "../../unopticalflow_amd/csrc/corr.hip"
#include "corr_mfma2.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

UnflowTimingArm& unflow_timing_arm() { static thread_local UnflowTimingArm arm{nullptr, nullptr, false}; return arm; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// values with full fp32 mantissas (a hash, not a short rational): the split must be exercised
static float vv(size_t i, float scale) {
    unsigned long long z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 29; z *= 0x94D049BB133111EBull; z ^= z >> 32;
    return scale * ((float)(z & 0xffffff) / 8388608.f - 1.f);
}

struct Err { double worst_abs, worst_rel_to_max, ref_max; size_t beyond; };
static Err compare(const std::vector<float>& ref, const std::vector<float>& got) {
    Err e{0, 0, 0, 0};
    for (float v : ref) e.ref_max = std::fmax(e.ref_max, std::fabs((double)v));
    for (size_t i = 0; i < ref.size(); ++i) {
        const double d = std::fabs((double)ref[i] - (double)got[i]);
        if (!(d <= e.worst_abs)) e.worst_abs = d;
        if (!(d <= 1e-5 * std::fabs((double)ref[i]) + 2e-6 * e.ref_max)) ++e.beyond;
    }
    e.worst_rel_to_max = e.worst_abs / e.ref_max;
    return e;
}

constexpr int NV = 9;

template <int R>
static int run_shape(int B, int C, int H, int W, int iters) {
    const int d = R, DD = (2 * d + 1) * (2 * d + 1);
    const size_t nf = (size_t)B * C * H * W, nc = (size_t)B * DD * H * W;
    std::vector<float> h1(nf), h2(nf), hg(nc);
    for (size_t i = 0; i < nf; ++i) { h1[i] = vv(i, 1.f); h2[i] = vv(i + 77777777ull, 1.f); }
    for (size_t i = 0; i < nc; ++i) hg[i] = vv(i + 555555555ull, 0.05f);
    float *f1, *f2, *g, *out[NV][2];
    CK(hipMalloc(&f1, nf * 4)); CK(hipMalloc(&f2, nf * 4)); CK(hipMalloc(&g, nc * 4));
    for (int v = 0; v < NV; ++v) for (int k = 0; k < 2; ++k) { CK(hipMalloc(&out[v][k], nf * 4)); CK(hipMemset(out[v][k], 0xff, nf * 4)); }
    CK(hipMemcpy(f1, h1.data(), nf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(f2, h2.data(), nf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), nc * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[NV] = {"shipped entry (unflow_corr_bwd)", "mfma 16 rows, 1 set, skip per pair", "mfma 16 rows, 1 set, skip per round",
                             "mfma 32 rows, 2 sets, skip per pair", "mfma 32 rows, 2 sets, skip per round", "mfma 16 rows, 2 sets, skip per round",
                             "mfma2 (pixel pairs) 16 rows, per pair", "mfma2 (pixel pairs) 16 rows, per round", "mfma2 (pixel pairs) 32 rows, per pair"};
    auto run = [&](int v) -> int {
        switch (v) {
            case 0: return unflow_corr_bwd(f1, f2, g, out[0][0], out[0][1], B, C, H, W, d, s);
            case 1: return launch_bwd_mf<R, 2, 1, 1>(f1, f2, g, out[1][0], out[1][1], B, C, H, W, 16, s);
            case 2: return launch_bwd_mf<R, 2, 1, 2>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, 16, s);
            case 3: return launch_bwd_mf<R, 2, 2, 1>(f1, f2, g, out[3][0], out[3][1], B, C, H, W, 32, s);
            case 4: return launch_bwd_mf<R, 2, 2, 2>(f1, f2, g, out[4][0], out[4][1], B, C, H, W, 32, s);
            case 5: return launch_bwd_mf<R, 2, 2, 2>(f1, f2, g, out[5][0], out[5][1], B, C, H, W, 16, s);
            case 6: return launch_bwd_mf2<R, 2, 1>(f1, f2, g, out[6][0], out[6][1], B, C, H, W, 16, s);
            case 7: return launch_bwd_mf2<R, 2, 2>(f1, f2, g, out[7][0], out[7][1], B, C, H, W, 16, s);
            default: return launch_bwd_mf2<R, 2, 1>(f1, f2, g, out[8][0], out[8][1], B, C, H, W, 32, s);
        }
    };
    printf("== d = %d  [%d,%d,%d,%d]  (algorithmic bytes %.1f MB)\n", d, B, C, H, W, 4.0 * B * H * W * (4.0 * C + DD) / 1e6);
    for (int round = 0; round < 2; ++round)
        for (int v = 0; v < NV; ++v) {
            for (int i = 0; i < 3; ++i) { int rc = run(v); if (rc) { fprintf(stderr, "%s: launch failed %d\n", names[v], rc); return 1; } }
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) run(v);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d  %-44s %8.2f us/launch\n", round, names[v], ms * 1e3 / iters);
        }
    std::vector<float> ref(nf), got(nf);
    for (int v = 1; v < NV; ++v)
        for (int k = 0; k < 2; ++k) {
            CK(hipMemcpy(ref.data(), out[0][k], nf * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(got.data(), out[v][k], nf * 4, hipMemcpyDeviceToHost));
            const Err e = compare(ref, got);
            if (e.worst_rel_to_max > 1e-3) {                 // a real error, not rounding: where?
                int shown = 0; size_t nbad = 0, by_cg[8] = {0}, by_seg[64] = {0}, by_y[256] = {0};
                for (size_t i = 0; i < nf; ++i) {
                    if (std::fabs((double)ref[i] - (double)got[i]) > 1e-4 * e.ref_max) {
                        const int x = i % W, y = (i / W) % H, c = (i / ((size_t)W * H)) % C, b = i / ((size_t)W * H * C);
                        ++nbad; ++by_cg[(c / 16) & 7]; ++by_seg[(x / 16) & 63]; ++by_y[y & 255];
                        if (shown++ < 6) printf("   bad at b %d c %d y %d x %d: %.6g vs %.6g\n", b, c, y, x, got[i], ref[i]);
                    }
                }
                printf("   %zu bad; by channel group:", nbad); for (int k = 0; k < (C + 15) / 16; ++k) printf(" %zu", by_cg[k]);
                printf("; by segment:"); for (int k = 0; k < (W + 15) / 16; ++k) printf(" %zu", by_seg[k]);
                printf("; by row:"); for (int k = 0; k < H && k < 256; ++k) printf(" %zu", by_y[k]);
                printf("\n");
            }
            printf("%-44s gf%d: worst |diff| %.3g = %.3g of max |ref| %.3g; %zu of %zu beyond rtol 1e-5 + 2e-6 max\n", names[v], k + 1,
                   e.worst_abs, e.worst_rel_to_max, e.ref_max, e.beyond, nf);
        }
    // a float64 sum on the host for a sample of outputs: the error of BOTH kernels against the exact value
    {
        CK(hipMemcpy(ref.data(), out[0][0], nf * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), out[2][0], nf * 4, hipMemcpyDeviceToHost));
        double e_ref = 0, e_got = 0, vmax = 0;
        const int DDr = 2 * d + 1;
        for (size_t k = 0; k < 4000; ++k) {
            const size_t idx = (k * 2654435761ull) % nf;
            const int x = idx % W, y = (idx / W) % H, c = (idx / ((size_t)W * H)) % C, b = idx / ((size_t)W * H * C);
            double sum = 0;
            for (int i = 0; i < DDr; ++i)
                for (int j = 0; j < DDr; ++j) {
                    const int yy = y + i - d, xx = x + j - d;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    sum += (double)hg[(((size_t)b * DD + i * DDr + j) * H + y) * W + x] * (double)h2[(((size_t)b * C + c) * H + yy) * W + xx];
                }
            sum /= C;
            e_ref = std::fmax(e_ref, std::fabs(sum - ref[idx])); e_got = std::fmax(e_got, std::fabs(sum - got[idx])); vmax = std::fmax(vmax, std::fabs(sum));
        }
        printf("gf1 against a float64 sum (4000 samples, max |value| %.3g): fp32 kernel worst %.3g, mfma kernel worst %.3g (%.3g of max)\n", vmax, e_ref, e_got, e_got / vmax);
    }
    for (int v = 0; v < NV; ++v) for (int k = 0; k < 2; ++k) CK(hipFree(out[v][k]));
    CK(hipFree(f1)); CK(hipFree(f2)); CK(hipFree(g));
    return 0;
}

int main(int argc, char** argv) {
    int B = 16, C = 32, H = 64, W = 208, iters = 50;
    if (argc >= 5) { B = atoi(argv[1]); C = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); }
    if (argc >= 6) iters = atoi(argv[5]);
    if (run_shape<4>(B, C, H, W, iters)) return 1;
    if (run_shape<8>(B, C, H, W, iters)) return 1;
    return 0;
}
