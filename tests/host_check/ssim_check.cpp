// csrc/ssim.hip -- the SSIM loss kernels (pytorch_ssim/ssim.py:4-20 + compute_loss_ssim, model_flow_paper.py:137-148), forward and backward, the
// column-pair kernels and the general ones, single-scale and `_ms`, kernels AND C entries -- compiled for the build host and EXECUTED with lanes
// as fibers (tests/host_check/hip_on_host.h; TEST INFRASTRUCTURE, tests/test_kernels_on_host.py builds and runs it with the ROCm clang++: the
// kernels use clang's vector extensions).  The wave shifts (DPP on the device) are exchanges between fibers, the hardware reciprocal is the IEEE one.
//
// Inputs carry what VERDICT r4 asked for: saturated flat patches (x = 1.0 against y = 1.0, 1 - 1/255, 0.95), dark flat patches and a step
// edge on top of noise.  Checked here: the `_ms` launch over three scales leaves the bits of three single-scale launches (partial sums and
// gradients); written to a file: losses and gradients, which the Python test compares with the oracle.
#include "ssim.hip"

UnflowTimingArm& unflow_timing_arm() { static UnflowTimingArm arm = {nullptr, nullptr, false}; return arm; }      // photo.hip's: never armed here

static float v(size_t i) { return (float)((i * 2654435761ull) % 2001ull) / 1000.f - 1.f; }
static int failures = 0;
static void same(const char* what, int s, const std::vector<float>& a, const std::vector<float>& b) {
    if (a.size() != b.size() || memcmp(a.data(), b.data(), a.size() * 4) != 0) { printf("MISMATCH %s scale %d\n", what, s); ++failures; }
}
struct Case { int H, W; std::vector<float> img, warped, w, gloss; };
static Case make(int B, int B2, int H, int W, size_t seed) {
    Case c; c.H = H; c.W = W;
    const size_t hw = (size_t)H * W;
    c.img.resize(B * 3 * hw); c.warped.resize(B2 * 3 * hw); c.w.resize(B2 * hw); c.gloss.resize(B2);
    for (size_t i = 0; i < c.img.size(); ++i) c.img[i] = v(i + seed) * 0.5f + 0.5f;
    for (size_t i = 0; i < c.warped.size(); ++i) { const float t = c.img[i % c.img.size()] + 0.1f * v(i + seed + 7); c.warped[i] = t < 0.f ? 0.f : (t > 1.f ? 1.f : t); }
    for (size_t i = 0; i < c.w.size(); ++i) c.w[i] = v(i + seed + 13) + 1.f;                       // weights in [0, 2]
    for (int i = 0; i < B2; ++i) c.gloss[i] = v(i + seed + 17);
    const int hq = H / 4 > 2 ? H / 4 : 2, wq = W / 4 > 2 ? W / 4 : 2;
    auto I = [&](int b, int ch, int y, int x) -> float& { return c.img[(((size_t)b * 3 + ch) * H + y) * W + x]; };
    auto Y = [&](int b, int ch, int y, int x) -> float& { return c.warped[(((size_t)b * 3 + ch) * H + y) * W + x]; };
    const float dys[3] = {0.f, 1.f / 255.f, 0.05f};
    for (int b = 0; b < B2; ++b) for (int ch = 0; ch < 3; ++ch) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const int bi = b % B;
        if (y < hq && x < 3 * wq) { I(bi, ch, y, x) = 1.0f; Y(b, ch, y, x) = 1.0f - dys[x / wq]; }                 // bright flat patches, x != y by a constant
        else if (y < 2 * hq && y >= hq && x < wq) { I(bi, ch, y, x) = 0.f; Y(b, ch, y, x) = 0.f; }                 // dark, equal
        else if (y < 2 * hq && y >= hq && x < 2 * wq) { I(bi, ch, y, x) = 2.f / 255.f; Y(b, ch, y, x) = 0.f; }     // dark, unequal
        else if (y >= 2 * hq && x >= 2 * wq) { I(bi, ch, y, x) = x >= 3 * wq ? 0.9f : 0.25f; Y(b, ch, y, x) = 0.5f; }   // a step edge in x only
    }
    for (int b = 0; b < B2; ++b) for (int y = 0; y < hq; ++y) for (int x = 0; x < W; ++x) c.w[((size_t)b * H + y) * W + x] = (v(x + y * 31 + b) > 0.f) ? 2.f : 1.f;
    return c;
}

int main(int argc, char** argv) {
    constexpr int n = 3, B = 1, B2 = 2;
    static const int Hs[n] = {128, 64, 32}, Ws[n] = {64, 32, 16};            // 16 rows per wave at scale 0, 8 below (both instantiations of the `_ms` kernel)
    FILE* f = argc > 1 ? fopen(argv[1], "wb") : nullptr;
    auto dump = [&](const std::vector<float>& a) { if (f) fwrite(a.data(), 4, a.size(), f); };
    Case C[n];
    std::vector<float> loss[n], sums[n], pa[n], pm[n], ga[n], gm[n];
    const float *img[n], *warped[n], *w[n], *sm[n], *gl[n];
    float *pmp[n], *gmp[n];
    for (int s = 0; s < n; ++s) {
        C[s] = make(B, B2, Hs[s], Ws[s], 1000 * (s + 1));
        const size_t hw = (size_t)Hs[s] * Ws[s];
        const int pps = 2 * unflow_ssim_blocks(Hs[s], Ws[s]);
        loss[s].assign(B2, -7.f); sums[s].assign(B2 * 2, -7.f); pa[s].assign((size_t)B2 * pps, 0.f); pm[s] = pa[s];
        ga[s].assign(B2 * 3 * hw, -7.f); gm[s] = ga[s];
        if (unflow_ssim_loss_fwd(C[s].img.data(), C[s].warped.data(), C[s].w.data(), loss[s].data(), sums[s].data(), pa[s].data(), B2, Hs[s], Ws[s], B, nullptr)) return 2;
        if (unflow_ssim_loss_bwd(C[s].img.data(), C[s].warped.data(), C[s].w.data(), sums[s].data(), C[s].gloss.data(), ga[s].data(), B2, Hs[s], Ws[s], B, nullptr)) return 2;
        img[s] = C[s].img.data(); warped[s] = C[s].warped.data(); w[s] = C[s].w.data(); sm[s] = sums[s].data(); gl[s] = C[s].gloss.data();
        pmp[s] = pm[s].data(); gmp[s] = gm[s].data();
    }
    if (unflow_ssim_loss_fwd_ms(n, img, warped, w, pmp, Hs, Ws, B2, B, nullptr)) return 3;
    if (unflow_ssim_loss_bwd_ms(n, img, warped, w, sm, gl, gmp, Hs, Ws, B2, B, nullptr)) return 3;
    for (int s = 0; s < n; ++s) {
        same("SSIM partial sums", s, pa[s], pm[s]); same("SSIM backward", s, ga[s], gm[s]);
        for (float e : gm[s]) if (e == -7.f) { printf("UNWRITTEN element, scale %d\n", s); ++failures; break; }
        dump(C[s].img); dump(C[s].warped); dump(C[s].w); dump(C[s].gloss); dump(loss[s]); dump(gm[s]);
    }
    // an odd width: the general kernels (one lane per column, neighbours by __shfl_up / __shfl_down)
    {
        const int H = 33, W = 57;
        Case c = make(B, B2, H, W, 5000);
        std::vector<float> l(B2, -7.f), su(B2 * 2, -7.f), p((size_t)B2 * 2 * unflow_ssim_blocks(H, W), 0.f), g((size_t)B2 * 3 * H * W, -7.f);
        if (unflow_ssim_loss_fwd(c.img.data(), c.warped.data(), c.w.data(), l.data(), su.data(), p.data(), B2, H, W, B, nullptr)) return 4;
        if (unflow_ssim_loss_bwd(c.img.data(), c.warped.data(), c.w.data(), su.data(), c.gloss.data(), g.data(), B2, H, W, B, nullptr)) return 4;
        dump(c.img); dump(c.warped); dump(c.w); dump(c.gloss); dump(l); dump(g);
    }
    if (f) fclose(f);
    printf("%s: %d mismatches\n", failures ? "FAILED" : "OK", failures);
    return failures ? 1 : 0;
}
