// One launch over the scales of a loss (round 5).
//
// Model_flow.forward (model_flow_paper.py:224-235) evaluates every loss at num_scales = 3 pyramid scales; scales 1 and 2 hold 1/4 and
// 1/16 of scale 0's pixels, and as launches of their own each sits on a 4-10 us floor for 0.2-8 MB (20 of the 30 loss launches of a
// step, 132 of 320 us).  Here a kernel's body is a file of its own (csrc/bodies/*.inc) that two kernels include: the single-scale
// kernel, unchanged token for token (tests/test_abi.py holds its instruction-stream hash to the validated build), and the `_ms` kernel,
// which finds its scale from the linear workgroup id, brings that scale's arguments into scope under the body's parameter names and
// shadows blockIdx / gridDim with the scale's virtual coordinates -- same body, same per-scale grid, same partial sums, same bits.
// A scale's range of workgroups starts at a multiple of 8, so `id mod 8` -- the XCD a workgroup lands on, which xcd_remap() relies
// on -- is the same for the virtual id as for the real one.
#pragma once
#ifndef UNFLOW_HOST_CHECK        // (tests/host_check/ms_flat_check.cpp compiles this header and the flat `_ms` kernels with g++: no HIP there)
#include "common.h"
#endif

// what a `_ms` kernel declares as LOCALS named blockIdx / gridDim in front of the included body (they shadow the built-ins)
struct VDim { unsigned x, y, z; };

constexpr int MS_MAX = 4;                                // == LOSS_MAX_SCALES (photo.hip)
struct MsGrid { int n; unsigned first[MS_MAX + 1], gx[MS_MAX], gy[MS_MAX], gz[MS_MAX]; };
template <class A> struct MsTable { MsGrid grid; A a[MS_MAX]; };

// host: append a scale with grid `d`; returns false when the table is full or the grid is empty
static inline bool ms_grid_add(MsGrid& g, dim3 d) {
    if (g.n >= MS_MAX || d.x == 0 || d.y == 0 || d.z == 0) return false;
    const unsigned long long blocks = (unsigned long long)d.x * d.y * d.z;
    if (blocks + g.first[g.n] + 8 > 0x7fffffffull) return false;
    g.gx[g.n] = d.x; g.gy[g.n] = d.y; g.gz[g.n] = d.z;
    g.first[g.n + 1] = g.first[g.n] + (unsigned)((blocks + 7) & ~7ull);
    ++g.n;
    return true;
}
static inline MsGrid ms_grid_empty() { MsGrid g = {}; return g; }
static inline unsigned ms_grid_blocks(const MsGrid& g) { return g.first[g.n]; }

// device: which scale this workgroup belongs to (block-uniform; -1: one of the padding workgroups behind a scale's range) and its
// virtual coordinates there
__device__ __forceinline__ int ms_locate(const MsGrid& g, VDim& vb, VDim& vg) {
    int s = 0;
    while (s < g.n - 1 && blockIdx.x >= g.first[s + 1]) ++s;
    const unsigned li = blockIdx.x - g.first[s];
    const unsigned gx = g.gx[s], gy = g.gy[s], gz = g.gz[s];
    if (li >= gx * gy * gz) return -1;
    const unsigned q = li / gx;
    vb.x = li - q * gx; vb.z = q / gy; vb.y = q - vb.z * gy;
    vg.x = gx; vg.y = gy; vg.z = gz;
    return s;
}

// the prologue of every `_ms` kernel (`t`: its MsTable<Args> parameter): afterwards `ms_a_` is this workgroup's scale's arguments and
// the names blockIdx / gridDim mean the virtual coordinates
#define UNFLOW_MS_PROLOGUE(t)                                   \
    VDim ms_vb_, ms_vg_;                                        \
    const int ms_scale_ = ms_locate((t).grid, ms_vb_, ms_vg_);  \
    if (ms_scale_ < 0) return;                                  \
    const auto& ms_a_ = (t).a[ms_scale_];                       \
    const VDim blockIdx = ms_vb_, gridDim = ms_vg_;             \
    (void)blockIdx; (void)gridDim
