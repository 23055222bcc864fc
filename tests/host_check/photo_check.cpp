// csrc/photo.hip -- the occlusion-weight, masked-mean, smoothness and consistency kernels with their C entries, single-scale and `_ms` --
// compiled with g++ and EXECUTED on the build host, lanes as threads (tests/host_check/hip_on_host.h; TEST INFRASTRUCTURE,
// tests/test_kernels_on_host.py builds and runs it).  The program includes the shipped source file itself; LDS tiles, barriers and the
// butterfly reductions of common.h run as written.
//
// It drives the entries as ops.py does for the three scales of a train step -- first stages, ONE unflow_loss_finalize_batch, backward
// entries -- twice: scale by scale through the single-scale entries and once per loss through the `_ms` entries; checks that both leave
// the same bits everywhere (partial sums included); and writes the results to a file the Python test compares with the oracle.
//
//   g++ -O1 -std=c++20 -pthread -ffp-contract=off -DUNFLOW_HOST_CHECK -I tests/host_check -I unopticalflow_amd/csrc tests/host_check/photo_check.cpp -o photo_check
#include "photo.hip"

// (ssim.hip's two host functions: they only size the partial-sum scratch here -- anything large enough will do)
int unflow_ssim_blocks(int H, int W) { return ceil_div(W, 60) * ceil_div(H, 8) + 8; }
int unflow_ssim_loss_blocks(int H, int W, int) { return unflow_ssim_blocks(H, W); }

static float v(size_t i) { return (float)((i * 2654435761ull) % 2001ull) / 1000.f - 1.f; }
static std::vector<float> gen(size_t n, size_t seed, float scale = 1.f, float shift = 0.f) {
    std::vector<float> a(n);
    for (size_t i = 0; i < n; ++i) a[i] = v(i + seed) * scale + shift;
    return a;
}
static int failures = 0;
static void same(const char* what, int s, const std::vector<float>& a, const std::vector<float>& b) {
    if (a.size() != b.size() || memcmp(a.data(), b.data(), a.size() * 4) != 0) { printf("MISMATCH %s scale %d\n", what, s); ++failures; }
}

int main(int argc, char** argv) {
    constexpr int n = 3, B = 2;                                       // B centre images; 2B stacked samples (bwd | fwd)
    static const int Hs[n] = {20, 10, 5}, Ws[n] = {136, 68, 34};     // 136 = two 64-wide smoothness tiles + a ragged one; 20 rows = 2.5 tiles of 8
    FILE* f = argc > 1 ? fopen(argv[1], "wb") : nullptr;
    auto dump = [&](const std::vector<float>& a) { if (f) fwrite(a.data(), 4, a.size(), f); };
    struct Out { std::vector<float> diff, wgt, l_pix, s_pix, l_sm, l_co, s_co, gdiff, gfrom, gflow_sm, gflow_co, p_pix, p_sm, p_co; };
    struct Scale { int H, W, HW; std::vector<float> img, warped, flows, gl_pix, gl_sm, gl_co; Out a, m; } S[n];     // a: per scale; m: `_ms`
    for (int s = 0; s < n; ++s) {
        Scale& q = S[s];
        q.H = Hs[s]; q.W = Ws[s]; q.HW = q.H * q.W;
        const size_t hw = q.HW;
        q.img = gen(B * 3 * hw, 11 + s, 0.5f, 0.5f);
        q.warped = gen(2 * B * 3 * hw, 101 + s, 0.5f, 0.5f);
        for (size_t i = 0; i < hw / 3; ++i) q.warped[i] = q.warped[hw + i] = q.warped[2 * hw + i] = 0.f;
        q.flows = gen(2 * B * 2 * hw, 201 + s, 3.f);                    // [2B,2,H,W] = (centre->left | centre->right)
        q.gl_pix = gen(2 * B, 301 + s); q.gl_sm = gen(2 * B, 311 + s); q.gl_co = gen(B, 321 + s);
        for (Out* o : {&q.a, &q.m}) {
            const int pps = unflow_partials_per_sample(q.H, q.W);
            o->diff.assign(2 * B * hw, -7.f); o->wgt = o->diff; o->gdiff = o->diff;
            o->gfrom.assign(2 * B * 3 * hw, -7.f); o->gflow_sm.assign(2 * B * 2 * hw, -7.f); o->gflow_co.assign(B * 2 * hw, -7.f);
            o->l_pix.assign(2 * B, -7.f); o->s_pix.assign(2 * B * 2, -7.f); o->l_sm.assign(2 * B, -7.f); o->l_co.assign(B, -7.f); o->s_co.assign(B * 2, -7.f);
            o->p_pix.assign((size_t)2 * B * pps, 0.f); o->p_sm = o->p_pix; o->p_co = o->p_pix;
        }
    }
    // ================= scale by scale through the single-scale entries (deferred second stages: loss == NULL)
    for (int s = 0; s < n; ++s) {
        Scale& q = S[s]; Out& o = q.a;
        const size_t hw = q.HW;
        if (unflow_occ_weight_fwd(q.img.data(), q.warped.data(), q.warped.data() + B * 3 * hw, o.diff.data(), o.diff.data() + B * hw, o.wgt.data(),
                                  o.wgt.data() + B * hw, nullptr, nullptr, B, q.H, q.W, nullptr)) return 2;
        if (unflow_masked_mean_fwd(o.diff.data(), o.wgt.data(), nullptr, o.s_pix.data(), o.p_pix.data(), 2 * B, q.H, q.W, nullptr)) return 2;
        if (unflow_smooth2_fwd(q.flows.data(), q.img.data(), nullptr, o.p_sm.data(), 2 * B, q.H, q.W, B, nullptr)) return 2;
        if (unflow_consis_fwd(q.flows.data() + B * 2 * hw, q.flows.data(), o.wgt.data() + B * hw, nullptr, o.s_co.data(), o.p_co.data(), B, q.H, q.W, nullptr)) return 2;
    }
    auto finalize = [&](bool ms) {                                       // ONE second-stage launch for the nine reductions, as ops.deferred_loss_sums
        const void* P[9]; void* L[9]; void* Sm[9]; int N[9], Bq[9], K[9]; float n0[9], n1[9];
        int j = 0;
        for (int s = 0; s < n; ++s) {
            Scale& q = S[s]; Out& o = ms ? q.m : q.a;
            const float hw = (float)q.H * (float)q.W;
            P[j] = o.p_pix.data(); L[j] = o.l_pix.data(); Sm[j] = o.s_pix.data(); N[j] = unflow_loss_partial_blocks(0, q.H, q.W, 2 * B, 1); Bq[j] = 2 * B; K[j] = 0; n0[j] = hw; n1[j] = hw; ++j;
            P[j] = o.p_sm.data(); L[j] = o.l_sm.data(); Sm[j] = nullptr; N[j] = unflow_loss_partial_blocks(2, q.H, q.W, 2 * B, 1); Bq[j] = 2 * B; K[j] = 1;
            n0[j] = 2.0f * (float)q.H * (float)(q.W - 2); n1[j] = 2.0f * (float)(q.H - 2) * (float)q.W; ++j;
            P[j] = o.p_co.data(); L[j] = o.l_co.data(); Sm[j] = o.s_co.data(); N[j] = unflow_loss_partial_blocks(3, q.H, q.W, B, 1); Bq[j] = B; K[j] = 0; n0[j] = 2.0f * hw; n1[j] = hw; ++j;
        }
        return unflow_loss_finalize_batch(P, L, Sm, N, Bq, K, n0, n1, j, nullptr);
    };
    if (finalize(false)) return 2;
    for (int s = 0; s < n; ++s) {
        Scale& q = S[s]; Out& o = q.a;
        const size_t hw = q.HW;
        if (unflow_masked_mean_bwd(o.wgt.data(), o.s_pix.data(), q.gl_pix.data(), o.gdiff.data(), 2 * B, q.H, q.W, nullptr)) return 2;
        if (unflow_absdiff_bwd(q.img.data(), q.warped.data(), o.gdiff.data(), o.gfrom.data(), 2 * B, q.H, q.W, B, nullptr)) return 2;
        if (unflow_smooth2_bwd(q.flows.data(), q.img.data(), q.gl_sm.data(), o.gflow_sm.data(), 2 * B, q.H, q.W, B, nullptr)) return 2;
        if (unflow_consis_bwd(q.flows.data() + B * 2 * hw, q.flows.data(), o.wgt.data() + B * hw, o.s_co.data(), q.gl_co.data(), o.gflow_co.data(), B, q.H, q.W, nullptr)) return 2;
    }
    // ================= once per loss through the `_ms` entries
    {
        const float *img[n], *warped[n], *flows[n], *ff[n], *fb[n], *wf[n], *diffc[n], *wgtc[n], *spix[n], *sco[n], *glp[n], *gls[n], *glc[n], *gdiffc[n];
        float *diff[n], *wgt[n], *ppix[n], *psm[n], *pco[n], *gdiff[n], *gfrom[n], *gsm[n], *gco[n];
        for (int s = 0; s < n; ++s) {
            Scale& q = S[s]; Out& o = q.m;
            const size_t hw = q.HW;
            img[s] = q.img.data(); warped[s] = q.warped.data(); flows[s] = q.flows.data(); ff[s] = q.flows.data() + B * 2 * hw; fb[s] = q.flows.data();
            diff[s] = o.diff.data(); wgt[s] = o.wgt.data(); diffc[s] = diff[s]; wgtc[s] = wgt[s]; wf[s] = o.wgt.data() + B * hw;
            ppix[s] = o.p_pix.data(); psm[s] = o.p_sm.data(); pco[s] = o.p_co.data(); spix[s] = o.s_pix.data(); sco[s] = o.s_co.data();
            glp[s] = q.gl_pix.data(); gls[s] = q.gl_sm.data(); glc[s] = q.gl_co.data();
            gdiff[s] = o.gdiff.data(); gdiffc[s] = gdiff[s]; gfrom[s] = o.gfrom.data(); gsm[s] = o.gflow_sm.data(); gco[s] = o.gflow_co.data();
        }
        if (unflow_occ_weight_fwd_ms(n, img, warped, diff, wgt, Hs, Ws, B, nullptr)) return 3;
        if (unflow_masked_mean_fwd_ms(n, diffc, wgtc, ppix, Hs, Ws, 2 * B, nullptr)) return 3;
        if (unflow_smooth2_fwd_ms(n, flows, img, psm, Hs, Ws, 2 * B, B, nullptr)) return 3;
        if (unflow_consis_fwd_ms(n, ff, fb, wf, pco, Hs, Ws, B, nullptr)) return 3;
        if (finalize(true)) return 3;
        if (unflow_masked_mean_bwd_ms(n, wgtc, spix, glp, gdiff, Hs, Ws, 2 * B, nullptr)) return 3;
        if (unflow_absdiff_bwd_ms(n, img, warped, gdiffc, gfrom, Hs, Ws, 2 * B, B, nullptr)) return 3;
        if (unflow_smooth2_bwd_ms(n, flows, img, gls, gsm, Hs, Ws, 2 * B, B, nullptr)) return 3;
        if (unflow_consis_bwd_ms(n, ff, fb, wf, sco, glc, gco, Hs, Ws, B, nullptr)) return 3;
    }
    for (int s = 0; s < n; ++s) {
        Scale& q = S[s]; Out &a = q.a, &m = q.m;
        same("diff", s, a.diff, m.diff); same("weight", s, a.wgt, m.wgt);
        same("masked-mean partial sums", s, a.p_pix, m.p_pix); same("smoothness partial sums", s, a.p_sm, m.p_sm); same("consistency partial sums", s, a.p_co, m.p_co);
        same("masked-mean loss", s, a.l_pix, m.l_pix); same("masked-mean sums", s, a.s_pix, m.s_pix);
        same("smoothness loss", s, a.l_sm, m.l_sm); same("consistency loss", s, a.l_co, m.l_co); same("consistency sums", s, a.s_co, m.s_co);
        same("masked-mean backward", s, a.gdiff, m.gdiff); same("|.| backward", s, a.gfrom, m.gfrom);
        same("smoothness backward", s, a.gflow_sm, m.gflow_sm); same("consistency backward", s, a.gflow_co, m.gflow_co);
        for (const auto* x : {&m.diff, &m.wgt, &m.l_pix, &m.l_sm, &m.l_co, &m.gdiff, &m.gfrom, &m.gflow_sm, &m.gflow_co})
            for (float e : *x) if (e == -7.f) { printf("UNWRITTEN element, scale %d\n", s); ++failures; break; }
        dump(m.l_pix); dump(m.l_sm); dump(m.l_co); dump(m.gfrom); dump(m.gflow_sm); dump(m.gflow_co);
    }
    if (f) fclose(f);
    printf("%s: %d mismatches\n", failures ? "FAILED" : "OK", failures);
    return failures ? 1 : 0;
}
