// The C entry of warp_fwd_ms_kernel (ms_flat_warp.h) in a header of its own: compiled with g++ and called by tests/host_check/ms_flat_check.cpp
// (see ms_flat_photo_entries.h).  Included by warp.hip at file scope.
#pragma once

extern "C" int unflow_warp_fwd_ms(int n, const float* const* src, const float* const* flow, float* const* out, uint8_t* const* mask,
                                  const int* H, const int* W, int B, int C, int align_corners, void* stream) {
    UNFLOW_REQUIRE(src && flow && out && mask && H && W && n > 0 && n <= MS_MAX && B > 0 && B <= 65535 && C > 0 && C <= 4);
    MsTable<WarpMsArgs> t = {};
    for (int k = 0; k < n; ++k) {
        UNFLOW_REQUIRE(src[k] && flow[k] && out[k] && mask[k] && H[k] > 0 && H[k] <= 65535 && W[k] > 0);
        t.a[k] = WarpMsArgs{src[k], flow[k], out[k], mask[k], H[k], W[k]};
        UNFLOW_REQUIRE(ms_grid_add(t.grid, dim3(ceil_div(W[k], 64), H[k], B)));
    }
    UNFLOW_LAUNCH(warp_fwd_ms_kernel, dim3(ms_grid_blocks(t.grid)), dim3(64, 1), 0, (hipStream_t)stream, t, C, align_corners ? 1 : 0);
    return unflow_launch_status();
}
