// The `_ms` form of the masked image warp's forward (csrc/multiscale.h) in a header of its own: tests/host_check/ms_flat_check.cpp compiles
// exactly this definition with g++ and runs it on the build host (see ms_flat_photo.h).  Included by warp.hip inside its anonymous
// namespace, after warp_taps.h.
#pragma once
#include "multiscale.h"

struct WarpMsArgs { const float *src, *flow; float* out; uint8_t* mask; int H, W; };
__global__ void warp_fwd_ms_kernel(MsTable<WarpMsArgs> ms_table_, int C, int ac) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    constexpr int NY = 1; constexpr bool MASKED = true;
    const float* __restrict__ src = ms_a_.src; const float* __restrict__ flow = ms_a_.flow;
    float* __restrict__ out = ms_a_.out; uint8_t* __restrict__ mask = ms_a_.mask;
    const int H = ms_a_.H, W = ms_a_.W;
#include "bodies/warp_fwd.inc"
}
