// TORCH_LIBRARY(unflow_hip, ...): the C ABI of libunflow_hip.so as PyTorch dispatcher operators
// (SURVEY.md section 8b item 1).  Every operator takes / returns contiguous NCHW at::Tensor on the current HIP device
// and enqueues on at::hip::getCurrentHIPStream(); outputs come from the caching allocator; nothing synchronises.
// Schemas are registered for all backends, the kernels under the CUDA dispatch key (HIP on PyTorch-ROCm), and shape
// functions under Meta, so FakeTensor / torch.compile tracing sees the output shapes without running a kernel.
// Autograd formulas are attached from Python (unopticalflow_amd/torch_ops.py, torch.library.register_autograd).
//
// Built as libunflow_torch.so by unopticalflow_amd/build.py (host-only C++: g++ against the torch headers, linked to
// libunflow_hip.so); the ctypes path of ops.py does not need it.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/unflow_hip.h"

namespace {

using at::Tensor;

void* stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void check(int rc, const char* name) {
    TORCH_CHECK(rc == 0, name, " failed with status ", rc, rc == UNFLOW_EINVAL ? " (invalid argument)" : " (hipError_t)");
}

Tensor nchw(const Tensor& t, const char* what) {
    TORCH_CHECK(t.is_cuda(), what, ": unopticalflow_amd ops run on an MI355X (HIP) device only; there is no CPU fallback");
    TORCH_CHECK(t.scalar_type() == at::kFloat, what, ": fp32 expected, got ", t.scalar_type());
    TORCH_CHECK(t.dim() == 4, what, ": [N,C,H,W] expected");
    return t.contiguous();
}

const float* fp(const Tensor& t) { return t.data_ptr<float>(); }
float* fpm(Tensor& t) { return t.data_ptr<float>(); }

int partials_per_sample(int64_t H, int64_t W) { return unflow_partials_per_sample((int)H, (int)W); }

// ---------------------------------------------------------------------------------------- cost volume
Tensor corr_fwd(const Tensor& f1_, const Tensor& f2_, int64_t d) {
    TORCH_CHECK(f1_.sizes() == f2_.sizes(), "corr: shapes differ");           // pwc_tf.py:99
    TORCH_CHECK(d >= 0, "corr: d >= 0");
    Tensor f1 = nchw(f1_, "corr f1"), f2 = nchw(f2_, "corr f2");
    const auto B = f1.size(0), C = f1.size(1), H = f1.size(2), W = f1.size(3);
    Tensor cv = at::empty({B, (2 * d + 1) * (2 * d + 1), H, W}, f1.options());
    check(unflow_corr_fwd(fp(f1), fp(f2), fpm(cv), B, C, H, W, (int)d, stream()), "unflow_corr_fwd");
    return cv;
}
Tensor corr_fwd_meta(const Tensor& f1, const Tensor& f2, int64_t d) {
    TORCH_CHECK(f1.sizes() == f2.sizes(), "corr: shapes differ");
    return at::empty({f1.size(0), (2 * d + 1) * (2 * d + 1), f1.size(2), f1.size(3)}, f1.options());
}

std::tuple<Tensor, Tensor> corr_bwd(const Tensor& f1_, const Tensor& f2_, const Tensor& g_, int64_t d) {
    Tensor f1 = nchw(f1_, "corr f1"), f2 = nchw(f2_, "corr f2"), g = nchw(g_, "corr gcv");
    const auto B = f1.size(0), C = f1.size(1), H = f1.size(2), W = f1.size(3);
    Tensor gf1 = at::empty_like(f1), gf2 = at::empty_like(f2);
    check(unflow_corr_bwd(fp(f1), fp(f2), fp(g), fpm(gf1), fpm(gf2), B, C, H, W, (int)d, stream()), "unflow_corr_bwd");
    return {gf1, gf2};
}
std::tuple<Tensor, Tensor> corr_bwd_meta(const Tensor& f1, const Tensor& f2, const Tensor&, int64_t) {
    return {at::empty_like(f1), at::empty_like(f2)};
}

// ---------------------------------------------------------------------------------------- warp
void check_flow(const Tensor& x, const Tensor& flow) {                         // net_utils.py:35-36
    TORCH_CHECK_VALUE(flow.dim() == 4 && flow.size(0) == x.size(0) && flow.size(1) == 2 && flow.size(2) == x.size(2) &&
                          flow.size(3) == x.size(3),
                      "the shape of grid [", x.size(0), ", 2, ", x.size(2), ", ", x.size(3), "] is not equal to the shape of flow ",
                      flow.sizes());
}

std::tuple<Tensor, Tensor> warp_fwd(const Tensor& x_, const Tensor& flow_, bool align_corners, bool want_mask) {
    Tensor x = nchw(x_, "warp src"), flow = nchw(flow_, "warp flow");
    check_flow(x, flow);
    const auto B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    Tensor out = at::empty_like(x);
    Tensor mask = want_mask ? at::empty({B, 1, H, W}, x.options().dtype(at::kByte)) : at::empty({0}, x.options().dtype(at::kByte));
    check(unflow_warp_fwd(fp(x), fp(flow), fpm(out), want_mask ? mask.data_ptr<uint8_t>() : nullptr, B, C, H, W,
                          align_corners ? 1 : 0, stream()), "unflow_warp_fwd");
    return {out, mask};
}
std::tuple<Tensor, Tensor> warp_fwd_meta(const Tensor& x, const Tensor& flow, bool, bool want_mask) {
    check_flow(x, flow);
    return {at::empty_like(x), want_mask ? at::empty({x.size(0), 1, x.size(2), x.size(3)}, x.options().dtype(at::kByte))
                                         : at::empty({0}, x.options().dtype(at::kByte))};
}

std::tuple<Tensor, Tensor> warp_bwd(const Tensor& x_, const Tensor& flow_, const Tensor& g_, const c10::optional<Tensor>& mask,
                                    bool align_corners, bool need_gsrc) {
    Tensor x = nchw(x_, "warp src"), flow = nchw(flow_, "warp flow"), g = nchw(g_, "warp gout");
    const auto B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
    Tensor gsrc = need_gsrc ? at::empty_like(x) : at::empty({0}, x.options());
    Tensor gflow = at::empty_like(flow);
    const uint8_t* mp = nullptr;
    Tensor m;
    if (mask.has_value() && mask->numel() > 0) { m = mask->contiguous(); mp = m.data_ptr<uint8_t>(); }
    check(unflow_warp_bwd(fp(x), fp(flow), fp(g), mp, need_gsrc ? fpm(gsrc) : nullptr, fpm(gflow), B, C, H, W,
                          align_corners ? 1 : 0, stream()), "unflow_warp_bwd");
    return {gsrc, gflow};
}
std::tuple<Tensor, Tensor> warp_bwd_meta(const Tensor& x, const Tensor& flow, const Tensor&, const c10::optional<Tensor>&, bool,
                                         bool need_gsrc) {
    return {need_gsrc ? at::empty_like(x) : at::empty({0}, x.options()), at::empty_like(flow)};
}

// ---------------------------------------------------------------------------------------- fused warp + cost volume
Tensor warp_corr_fwd(const Tensor& f1_, const Tensor& f2_, const Tensor& flow_, int64_t d, bool align_corners) {
    TORCH_CHECK(f1_.sizes() == f2_.sizes(), "warp_corr: shapes differ");
    Tensor f1 = nchw(f1_, "warp_corr f1"), f2 = nchw(f2_, "warp_corr f2"), flow = nchw(flow_, "warp_corr flow");
    check_flow(f2, flow);
    const auto B = f1.size(0), C = f1.size(1), H = f1.size(2), W = f1.size(3);
    Tensor cv = at::empty({B, (2 * d + 1) * (2 * d + 1), H, W}, f1.options());
    if (unflow_warp_corr_supported(C, H, W, (int)d)) {
        check(unflow_warp_corr_fwd(fp(f1), fp(f2), fp(flow), fpm(cv), B, C, H, W, (int)d, align_corners ? 1 : 0, stream()),
              "unflow_warp_corr_fwd");
    } else {                                                   // shapes the fused kernel does not cover: the two entry points
        Tensor warped = at::empty_like(f2);
        check(unflow_warp_fwd(fp(f2), fp(flow), fpm(warped), nullptr, B, C, H, W, align_corners ? 1 : 0, stream()), "unflow_warp_fwd");
        check(unflow_corr_fwd(fp(f1), fp(warped), fpm(cv), B, C, H, W, (int)d, stream()), "unflow_corr_fwd");
    }
    return cv;
}
Tensor warp_corr_fwd_meta(const Tensor& f1, const Tensor&, const Tensor&, int64_t d, bool) {
    return at::empty({f1.size(0), (2 * d + 1) * (2 * d + 1), f1.size(2), f1.size(3)}, f1.options());
}

std::tuple<Tensor, Tensor, Tensor> warp_corr_bwd(const Tensor& f1_, const Tensor& f2_, const Tensor& flow_, const Tensor& g_,
                                                 int64_t d, bool align_corners) {
    Tensor f1 = nchw(f1_, "warp_corr f1"), f2 = nchw(f2_, "warp_corr f2"), flow = nchw(flow_, "warp_corr flow"), g = nchw(g_, "gcv");
    const auto B = f1.size(0), C = f1.size(1), H = f1.size(2), W = f1.size(3);
    Tensor gf1 = at::empty_like(f1), gf2 = at::empty_like(f2), gflow = at::empty_like(flow);
    Tensor scratch = at::empty({2, B, C, H, W}, f1.options());
    check(unflow_warp_corr_bwd(fp(f1), fp(f2), fp(flow), fp(g), fpm(gf1), fpm(gf2), fpm(gflow), fpm(scratch), B, C, H, W, (int)d,
                               align_corners ? 1 : 0, stream()), "unflow_warp_corr_bwd");
    return {gf1, gf2, gflow};
}
std::tuple<Tensor, Tensor, Tensor> warp_corr_bwd_meta(const Tensor& f1, const Tensor& f2, const Tensor& flow, const Tensor&, int64_t,
                                                      bool) {
    return {at::empty_like(f1), at::empty_like(f2), at::empty_like(flow)};
}

// ---------------------------------------------------------------------------------------- occlusion weights
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> occ_weight(const Tensor& img_, const Tensor& l_, const Tensor& r_) {
    Tensor img = nchw(img_, "occ_weight img"), l = nchw(l_, "from_l"), r = nchw(r_, "from_r");
    TORCH_CHECK(img.size(1) == 3 && l.sizes() == img.sizes() && r.sizes() == img.sizes(), "occ_weight: three [B,3,H,W] images");
    const auto B = img.size(0), H = img.size(2), W = img.size(3);
    auto f = [&] { return at::empty({B, 1, H, W}, img.options()); };
    auto u = [&] { return at::empty({B, 1, H, W}, img.options().dtype(at::kByte)); };
    Tensor dl = f(), dr = f(), wb = f(), wf = f(), vb = u(), vf = u();
    check(unflow_occ_weight_fwd(fp(img), fp(l), fp(r), fpm(dl), fpm(dr), fpm(wb), fpm(wf), vb.data_ptr<uint8_t>(),
                                vf.data_ptr<uint8_t>(), B, H, W, stream()), "unflow_occ_weight_fwd");
    return {dl, dr, wb, wf, vb, vf};
}
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> occ_weight_meta(const Tensor& img, const Tensor&, const Tensor&) {
    auto f = [&] { return at::empty({img.size(0), 1, img.size(2), img.size(3)}, img.options()); };
    auto u = [&] { return at::empty({img.size(0), 1, img.size(2), img.size(3)}, img.options().dtype(at::kByte)); };
    return {f(), f(), f(), f(), u(), u()};
}

Tensor absdiff_bwd(const Tensor& img_, const Tensor& from_, const Tensor& g_) {
    Tensor img = nchw(img_, "img"), from = nchw(from_, "from"), g = nchw(g_, "gdiff");
    TORCH_CHECK(from.size(0) % img.size(0) == 0, "absdiff_bwd: batch of `from` must be a multiple of the image batch");
    Tensor out = at::empty_like(from);
    check(unflow_absdiff_bwd(fp(img), fp(from), fp(g), fpm(out), from.size(0), from.size(2), from.size(3), img.size(0), stream()),
          "unflow_absdiff_bwd");
    return out;
}
Tensor absdiff_bwd_meta(const Tensor&, const Tensor& from, const Tensor&) { return at::empty_like(from); }

// ---------------------------------------------------------------------------------------- per-sample losses
// masked_l1 = compute_loss_with_mask (one scale): returns (loss [B], sums [B,2])
std::tuple<Tensor, Tensor> masked_l1_fwd(const Tensor& diff_, const Tensor& w_) {
    Tensor diff = nchw(diff_, "diff"), w = nchw(w_, "w");
    const auto B = diff.size(0), H = diff.size(2), W = diff.size(3);
    Tensor loss = at::empty({B}, diff.options()), sums = at::empty({B, 2}, diff.options());
    Tensor part = at::empty({B * partials_per_sample(H, W)}, diff.options());
    check(unflow_masked_mean_fwd(fp(diff), fp(w), fpm(loss), fpm(sums), fpm(part), B, H, W, stream()), "unflow_masked_mean_fwd");
    return {loss, sums};
}
std::tuple<Tensor, Tensor> loss_sums_meta(const Tensor& a, const Tensor&) {
    return {at::empty({a.size(0)}, a.options()), at::empty({a.size(0), 2}, a.options())};
}
Tensor masked_l1_bwd(const Tensor& w_, const Tensor& sums, const Tensor& gl) {
    Tensor w = nchw(w_, "w");
    Tensor g = at::empty_like(w);
    Tensor s = sums.contiguous(), glc = gl.contiguous();
    check(unflow_masked_mean_bwd(fp(w), fp(s), fp(glc), fpm(g), w.size(0), w.size(2), w.size(3), stream()), "unflow_masked_mean_bwd");
    return g;
}
Tensor like_first_meta3(const Tensor& a, const Tensor&, const Tensor&) { return at::empty_like(a); }

std::tuple<Tensor, Tensor> ssim_loss_fwd(const Tensor& img_, const Tensor& warped_, const Tensor& w_) {
    Tensor img = nchw(img_, "img"), warped = nchw(warped_, "warped"), w = nchw(w_, "w");
    TORCH_CHECK(warped.size(1) == 3 && warped.size(0) % img.size(0) == 0 && w.size(0) == warped.size(0), "ssim_loss: shapes");
    const auto B = warped.size(0), H = warped.size(2), W = warped.size(3);
    Tensor loss = at::empty({B}, warped.options()), sums = at::empty({B, 2}, warped.options());
    Tensor part = at::empty({B * partials_per_sample(H, W)}, warped.options());
    check(unflow_ssim_loss_fwd(fp(img), fp(warped), fp(w), fpm(loss), fpm(sums), fpm(part), B, H, W, img.size(0), stream()),
          "unflow_ssim_loss_fwd");
    return {loss, sums};
}
std::tuple<Tensor, Tensor> ssim_loss_fwd_meta(const Tensor&, const Tensor& warped, const Tensor&) {
    return {at::empty({warped.size(0)}, warped.options()), at::empty({warped.size(0), 2}, warped.options())};
}
Tensor ssim_loss_bwd(const Tensor& img_, const Tensor& warped_, const Tensor& w_, const Tensor& sums, const Tensor& gl) {
    Tensor img = nchw(img_, "img"), warped = nchw(warped_, "warped"), w = nchw(w_, "w");
    Tensor g = at::empty_like(warped);
    Tensor s = sums.contiguous(), glc = gl.contiguous();
    check(unflow_ssim_loss_bwd(fp(img), fp(warped), fp(w), fp(s), fp(glc), fpm(g), warped.size(0), warped.size(2), warped.size(3),
                               img.size(0), stream()), "unflow_ssim_loss_bwd");
    return g;
}
Tensor ssim_loss_bwd_meta(const Tensor&, const Tensor& warped, const Tensor&, const Tensor&, const Tensor&) { return at::empty_like(warped); }

Tensor smooth2_fwd(const Tensor& flow_, const Tensor& img_) {
    Tensor flow = nchw(flow_, "flow"), img = nchw(img_, "img");
    TORCH_CHECK(flow.size(0) % img.size(0) == 0, "smooth2: batch of `flow` must be a multiple of the image batch");
    const auto B = flow.size(0), H = flow.size(2), W = flow.size(3);
    Tensor loss = at::empty({B}, flow.options());
    Tensor part = at::empty({B * partials_per_sample(H, W)}, flow.options());
    check(unflow_smooth2_fwd(fp(flow), fp(img), fpm(loss), fpm(part), B, H, W, img.size(0), stream()), "unflow_smooth2_fwd");
    return loss;
}
Tensor smooth2_fwd_meta(const Tensor& flow, const Tensor&) { return at::empty({flow.size(0)}, flow.options()); }
Tensor smooth2_bwd(const Tensor& flow_, const Tensor& img_, const Tensor& gl) {
    Tensor flow = nchw(flow_, "flow"), img = nchw(img_, "img");
    Tensor g = at::empty_like(flow);
    Tensor glc = gl.contiguous();
    check(unflow_smooth2_bwd(fp(flow), fp(img), fp(glc), fpm(g), flow.size(0), flow.size(2), flow.size(3), img.size(0), stream()),
          "unflow_smooth2_bwd");
    return g;
}

std::tuple<Tensor, Tensor> consis_fwd(const Tensor& ff_, const Tensor& fb_, const Tensor& w_) {
    Tensor ff = nchw(ff_, "fwd flow"), fb = nchw(fb_, "bwd flow"), w = nchw(w_, "w");
    const auto B = ff.size(0), H = ff.size(2), W = ff.size(3);
    Tensor loss = at::empty({B}, ff.options()), sums = at::empty({B, 2}, ff.options());
    Tensor part = at::empty({B * partials_per_sample(H, W)}, ff.options());
    check(unflow_consis_fwd(fp(ff), fp(fb), fp(w), fpm(loss), fpm(sums), fpm(part), B, H, W, stream()), "unflow_consis_fwd");
    return {loss, sums};
}
std::tuple<Tensor, Tensor> consis_fwd_meta(const Tensor& ff, const Tensor&, const Tensor&) {
    return {at::empty({ff.size(0)}, ff.options()), at::empty({ff.size(0), 2}, ff.options())};
}
Tensor consis_bwd(const Tensor& ff_, const Tensor& fb_, const Tensor& w_, const Tensor& sums, const Tensor& gl) {
    Tensor ff = nchw(ff_, "fwd flow"), fb = nchw(fb_, "bwd flow"), w = nchw(w_, "w");
    Tensor g = at::empty_like(ff);
    Tensor s = sums.contiguous(), glc = gl.contiguous();
    check(unflow_consis_bwd(fp(ff), fp(fb), fp(w), fp(s), fp(glc), fpm(g), ff.size(0), ff.size(2), ff.size(3), stream()),
          "unflow_consis_bwd");
    return g;
}
Tensor consis_bwd_meta(const Tensor& ff, const Tensor&, const Tensor&, const Tensor&, const Tensor&) { return at::empty_like(ff); }

}  // namespace

TORCH_LIBRARY(unflow_hip, m) {
    m.def("corr_fwd(Tensor f1, Tensor f2, int d) -> Tensor");
    m.def("corr_bwd(Tensor f1, Tensor f2, Tensor gcv, int d) -> (Tensor, Tensor)");
    m.def("warp_fwd(Tensor src, Tensor flow, bool align_corners, bool want_mask) -> (Tensor, Tensor)");
    m.def("warp_bwd(Tensor src, Tensor flow, Tensor gout, Tensor? mask, bool align_corners, bool need_gsrc) -> (Tensor, Tensor)");
    m.def("warp_corr_fwd(Tensor f1, Tensor f2, Tensor flow, int d, bool align_corners) -> Tensor");
    m.def("warp_corr_bwd(Tensor f1, Tensor f2, Tensor flow, Tensor gcv, int d, bool align_corners) -> (Tensor, Tensor, Tensor)");
    m.def("occ_weight(Tensor img, Tensor from_l, Tensor from_r) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("absdiff_bwd(Tensor img, Tensor src, Tensor gdiff) -> Tensor");
    m.def("masked_l1_fwd(Tensor diff, Tensor w) -> (Tensor, Tensor)");
    m.def("masked_l1_bwd(Tensor w, Tensor sums, Tensor gloss) -> Tensor");
    m.def("ssim_loss_fwd(Tensor img, Tensor warped, Tensor w) -> (Tensor, Tensor)");
    m.def("ssim_loss_bwd(Tensor img, Tensor warped, Tensor w, Tensor sums, Tensor gloss) -> Tensor");
    m.def("smooth2_fwd(Tensor flow, Tensor img) -> Tensor");
    m.def("smooth2_bwd(Tensor flow, Tensor img, Tensor gloss) -> Tensor");
    m.def("consis_fwd(Tensor fwd_flow, Tensor bwd_flow, Tensor w_fwd) -> (Tensor, Tensor)");
    m.def("consis_bwd(Tensor fwd_flow, Tensor bwd_flow, Tensor w_fwd, Tensor sums, Tensor gloss) -> Tensor");
}

TORCH_LIBRARY_IMPL(unflow_hip, CUDA, m) {       // the HIP backend of PyTorch-ROCm dispatches under the CUDA key
    m.impl("corr_fwd", corr_fwd);
    m.impl("corr_bwd", corr_bwd);
    m.impl("warp_fwd", warp_fwd);
    m.impl("warp_bwd", warp_bwd);
    m.impl("warp_corr_fwd", warp_corr_fwd);
    m.impl("warp_corr_bwd", warp_corr_bwd);
    m.impl("occ_weight", occ_weight);
    m.impl("absdiff_bwd", absdiff_bwd);
    m.impl("masked_l1_fwd", masked_l1_fwd);
    m.impl("masked_l1_bwd", masked_l1_bwd);
    m.impl("ssim_loss_fwd", ssim_loss_fwd);
    m.impl("ssim_loss_bwd", ssim_loss_bwd);
    m.impl("smooth2_fwd", smooth2_fwd);
    m.impl("smooth2_bwd", smooth2_bwd);
    m.impl("consis_fwd", consis_fwd);
    m.impl("consis_bwd", consis_bwd);
}

TORCH_LIBRARY_IMPL(unflow_hip, Meta, m) {
    m.impl("corr_fwd", corr_fwd_meta);
    m.impl("corr_bwd", corr_bwd_meta);
    m.impl("warp_fwd", warp_fwd_meta);
    m.impl("warp_bwd", warp_bwd_meta);
    m.impl("warp_corr_fwd", warp_corr_fwd_meta);
    m.impl("warp_corr_bwd", warp_corr_bwd_meta);
    m.impl("occ_weight", occ_weight_meta);
    m.impl("absdiff_bwd", absdiff_bwd_meta);
    m.impl("masked_l1_fwd", loss_sums_meta);
    m.impl("masked_l1_bwd", like_first_meta3);
    m.impl("ssim_loss_fwd", ssim_loss_fwd_meta);
    m.impl("ssim_loss_bwd", ssim_loss_bwd_meta);
    m.impl("smooth2_fwd", smooth2_fwd_meta);
    m.impl("smooth2_bwd", like_first_meta3);
    m.impl("consis_fwd", consis_fwd_meta);
    m.impl("consis_bwd", consis_bwd_meta);
}
