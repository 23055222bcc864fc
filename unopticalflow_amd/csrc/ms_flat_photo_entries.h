// The C entries of the flat `_ms` kernels of ms_flat_photo.h, in a header of their own for the same reason: tests/host_check/ms_flat_check.cpp
// compiles THESE definitions with g++ (UNFLOW_LAUNCH, UNFLOW_REQUIRE and the stream type behind one-line stand-ins) and calls them as the ctypes
// binding does -- argument checks, per-scale pointer arithmetic, the workgroup table and the launch geometry included.  Included by photo.hip
// at file scope (C linkage, exported), after flat_blocks() and UNFLOW_MS_REQUIRE_N.
#pragma once

// stacked layout of the train step: warped = (from_l | from_r) [2B,3,H,W] against the B centre images -> diff = (diff_l | diff_r),
// wgt = (w_bwd | w_fwd) [2B,1,H,W]  (model_flow_paper.py:108-132)
extern "C" int unflow_occ_weight_fwd_ms(int n, const float* const* img, const float* const* warped, float* const* diff,
                                        float* const* wgt, const int* H, const int* W, int B, void* stream) {
    UNFLOW_REQUIRE(img && warped && diff && wgt && H && W && B > 0);
    UNFLOW_MS_REQUIRE_N(n);
    MsTable<OccMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(img[k] && warped[k] && diff[k] && wgt[k] && H[k] > 0 && W[k] > 0);
        const int HW = H[k] * W[k];
        t.a[k] = OccMsArgs{img[k], warped[k], warped[k] + (size_t)B * 3 * HW, diff[k], diff[k] + (size_t)B * HW, wgt[k], wgt[k] + (size_t)B * HW, HW};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(flat_blocks((size_t)B * HW))));
    }
    UNFLOW_LAUNCH(occ_weight_fwd_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(256), 0, (hipStream_t)stream, t, B);
    return unflow_launch_status();
}

extern "C" int unflow_absdiff_bwd_ms(int n, const float* const* img, const float* const* from, const float* const* gdiff,
                                     float* const* gfrom, const int* H, const int* W, int B, int img_batch, void* stream) {
    UNFLOW_REQUIRE(img && from && gdiff && gfrom && H && W && B > 0 && img_batch > 0 && B % img_batch == 0);
    UNFLOW_MS_REQUIRE_N(n);
    MsTable<AbsdiffMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(img[k] && from[k] && gdiff[k] && gfrom[k] && H[k] > 0 && W[k] > 0);
        t.a[k] = AbsdiffMsArgs{img[k], from[k], gdiff[k], gfrom[k], H[k] * W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(flat_blocks((size_t)B * 3 * H[k] * W[k]))));
    }
    UNFLOW_LAUNCH(absdiff_bwd_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(256), 0, (hipStream_t)stream, t, B, img_batch);
    return unflow_launch_status();
}

extern "C" int unflow_masked_mean_bwd_ms(int n, const float* const* w, const float* const* sums, const float* const* gloss,
                                         float* const* gdiff, const int* H, const int* W, int B, void* stream) {
    UNFLOW_REQUIRE(w && sums && gloss && gdiff && H && W && B > 0);
    UNFLOW_MS_REQUIRE_N(n);
    MsTable<MeanBwdMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(w[k] && sums[k] && gloss[k] && gdiff[k] && H[k] > 0 && W[k] > 0);
        t.a[k] = MeanBwdMsArgs{w[k], sums[k], gloss[k], gdiff[k], H[k] * W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(flat_blocks((size_t)B * H[k] * W[k]))));
    }
    UNFLOW_LAUNCH(masked_mean_bwd_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(256), 0, (hipStream_t)stream, t, B);
    return unflow_launch_status();
}

extern "C" int unflow_consis_bwd_ms(int n, const float* const* fwd_flow, const float* const* bwd_flow, const float* const* w_fwd,
                                    const float* const* sums, const float* const* gloss, float* const* gflow, const int* H,
                                    const int* W, int B, void* stream) {
    UNFLOW_REQUIRE(fwd_flow && bwd_flow && w_fwd && sums && gloss && gflow && H && W && B > 0);
    UNFLOW_MS_REQUIRE_N(n);
    MsTable<ConsisBwdMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(fwd_flow[k] && bwd_flow[k] && w_fwd[k] && sums[k] && gloss[k] && gflow[k] && H[k] > 0 && W[k] > 0);
        t.a[k] = ConsisBwdMsArgs{fwd_flow[k], bwd_flow[k], w_fwd[k], sums[k], gloss[k], gflow[k], H[k] * W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(flat_blocks((size_t)B * H[k] * W[k]))));
    }
    UNFLOW_LAUNCH(consis_bwd_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(256), 0, (hipStream_t)stream, t, B);
    return unflow_launch_status();
}
