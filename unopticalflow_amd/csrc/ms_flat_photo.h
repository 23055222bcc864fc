// The `_ms` kernels (csrc/multiscale.h) of photo.hip's FLAT bodies -- grid-stride loops without LDS, barriers or wave shuffles -- in a
// header of their own, so that tests/host_check/ms_flat_check.cpp can compile exactly these definitions (table structs, prologue, the
// hook-up of a scale's arguments to the body's parameter names, the bodies themselves) with g++ and RUN them on the build host:
// tests/test_kernels_on_host.py compares them with the single-scale kernels run the same way, bit for bit, and with the oracle.
// Included by photo.hip inside its anonymous namespace, after `sgn`.
#pragma once
#include "multiscale.h"

struct OccMsArgs { const float *img, *from_l, *from_r; float *diff_l, *diff_r, *w_bwd, *w_fwd; int HW; };
__global__ void occ_weight_fwd_ms_kernel(MsTable<OccMsArgs> ms_table_, int B) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    const float* __restrict__ img = ms_a_.img; const float* __restrict__ from_l = ms_a_.from_l; const float* __restrict__ from_r = ms_a_.from_r;
    float* __restrict__ diff_l = ms_a_.diff_l; float* __restrict__ diff_r = ms_a_.diff_r;
    float* __restrict__ w_bwd = ms_a_.w_bwd; float* __restrict__ w_fwd = ms_a_.w_fwd;
    uint8_t* __restrict__ valid_bwd = nullptr; uint8_t* __restrict__ valid_fwd = nullptr;      // (the train step does not take the masks)
    const int HW = ms_a_.HW;
#include "bodies/occ_weight_fwd.inc"
}

struct AbsdiffMsArgs { const float *img, *from, *gdiff; float* gfrom; int HW; };
__global__ void absdiff_bwd_ms_kernel(MsTable<AbsdiffMsArgs> ms_table_, int B, int img_b) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    const float* __restrict__ img = ms_a_.img; const float* __restrict__ from = ms_a_.from; const float* __restrict__ gdiff = ms_a_.gdiff;
    float* __restrict__ gfrom = ms_a_.gfrom;
    const int HW = ms_a_.HW;
#include "bodies/absdiff_bwd.inc"
}

struct MeanBwdMsArgs { const float *w, *sums, *gloss; float* gdiff; int HW; };
__global__ void masked_mean_bwd_ms_kernel(MsTable<MeanBwdMsArgs> ms_table_, int B) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    const float* __restrict__ w = ms_a_.w; const float* __restrict__ sums = ms_a_.sums; const float* __restrict__ gloss = ms_a_.gloss;
    float* __restrict__ gdiff = ms_a_.gdiff;
    const int HW = ms_a_.HW;
#include "bodies/masked_mean_bwd.inc"
}

struct ConsisBwdMsArgs { const float *ff, *fb, *w_fwd, *sums, *gloss; float* gflow; int HW; };
__global__ void consis_bwd_ms_kernel(MsTable<ConsisBwdMsArgs> ms_table_, int B) {
    UNFLOW_MS_PROLOGUE(ms_table_);
    const float* __restrict__ ff = ms_a_.ff; const float* __restrict__ fb = ms_a_.fb; const float* __restrict__ w_fwd = ms_a_.w_fwd;
    const float* __restrict__ sums = ms_a_.sums; const float* __restrict__ gloss = ms_a_.gloss; float* __restrict__ gflow = ms_a_.gflow;
    const int HW = ms_a_.HW;
#include "bodies/consis_bwd.inc"
}
