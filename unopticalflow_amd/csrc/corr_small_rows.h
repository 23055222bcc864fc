// Small-map cost-volume backward, any radius, "rows in registers" (round 6; NOT RUN ON A GPU YET: reached only through
// unflow_corr_bwd_ex(UNFLOW_CORR_BWD_FP32_NEXT), executed and checked on the build host, tests/host_check/corr_check.cpp).
// Autograd of PWC_tf.corr_naive, /root/reference core/networks/structures/pwc_tf.py:97-106, for maps of <= 1024 pixels:
//     gf1[c, q] = 1/C  sum_{i, j} g[i (2R+1) + j][q]                      f2[c][q + (i-R, j-R)]
//     gf2[c, q] = 1/C  sum_{i, j} g[(2R-i)(2R+1) + (2R-j)][q + (i-R, j-R)] f1[c][q + (i-R, j-R)]
// What it replaces: at d = 8 the pyramid's levels 5 / 6 (8x26, 4x13) run corr_bwd_generic -- one lane per output element, 2 x 289 dependent global
// loads each: 119 / 57 us for 7.9 / 2.9 MB, 0.01 of the roofline (profiles/r5_corr_d8_microbench.txt).  corr_bwd_small_kernel<R> (d = 4) keeps all
// (2R+1)^2 gradients of a lane's pixel in registers, which at R = 8 is 289: more than the register file.
// Here a lane owns one pixel of the WHOLE map and CPL channels (one accumulator each); the chunk's F planes sit zero-padded in LDS, so taps need
// no bounds checks; the upstream gradient passes through the registers DGS displacement rows at a time (DGS (2R+1) values, the next group requested
// while the current one is used), each value used by all CPL channels.  Per FMA: one 4-byte LDS read (stride-1 across lanes: conflict-free) -- the
// kernel is LDS-read bound at about twice its FMA time, ~10-15 us at level 5 by that count.  Workgroup = pxl pixel lanes x 256 / pxl channel phases of
// one (sample, gradient, chunk of CPL x phases channels); grid (pixel blocks, channel chunks, 2 B).  One accumulator chain per (lane, channel), rows
// in order i = 0 .. 2R, j = 0 .. 2R: deterministic, a different fp32 order than corr_bwd_small_kernel's two alternating chains.
#pragma once
#include <type_traits>

namespace {

template <int R, int DGS, int CPL>
__global__ __launch_bounds__(256) void corr_bwd_smallrows_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                 const float* __restrict__ g, float* __restrict__ gf1,
                                                                 float* __restrict__ gf2, int C, int H, int W, int pxl, float inv_c) {
    constexpr int DD = 2 * R + 1, NG = (DD + DGS - 1) / DGS;
    UNFLOW_DYNAMIC_LDS(float, planes);                 // CPL * csub planes of (H + 2R) x (W + 2R), zero padded
    const int mode = blockIdx.z & 1, b = blockIdx.z >> 1;
    const float* __restrict__ F = mode ? f1 : f2;
    float* __restrict__ out = mode ? gf2 : gf1;
    const int PH = H + 2 * R, PW = W + 2 * R, plane = H * W, pplane = PH * PW;
    const int csub = 256 / pxl, cch = CPL * csub;      // channel phases per workgroup; channels per workgroup
    const int lp = threadIdx.x % pxl, cs = threadIdx.x / pxl;
    const int q = blockIdx.x * pxl + lp;               // this lane's pixel
    const bool live = q < plane;
    const int y = live ? q / W : 0, x = live ? q - y * W : 0;
    const int c0 = blockIdx.y * cch;
    const int nch = min(cch, C - c0);

    // the gradient values of displacement rows [i0, i0 + DGS) at this lane's pixel (mode 1: at the displaced pixels, planes flipped), through ONE
    // buffer descriptor over the sample's (2R+1)^2 planes: "outside the image" (and a dead lane) is an offset with kOut added, which the hardware's
    // range check answers with 0 -- no branch around a load, no 64-bit address per value; the plane / displacement part of an offset is scalar
    constexpr unsigned kOut = 0x40000000u;            // (2R+1)^2 x 1024 pixels x 4 bytes stays far below
    const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g + (size_t)b * DD * DD * plane), 0, DD * DD * plane * 4, 0x00020000);
    // validity is ADDED to the offset, never selected per load (a select in front of each load made hipcc wrap the loads in EXEC-masked branches,
    // one wait per load): a column term per j, computed once and pinned, a row term per displacement row; two kOut's still sum to an offset
    // beyond num_records
    unsigned colterm[DD];                              // the lane's own byte offset where column x + j - R is inside (mode 0: always), else kOut
#pragma unroll
    for (int j = 0; j < DD; ++j) {
        colterm[j] = (live && ((mode == 0) || ((unsigned)(x + j - R) < (unsigned)W))) ? (unsigned)q * 4u : kOut;
        UNFLOW_PIN_VGPR(colterm[j]);
    }
    auto request = [&](float (&w)[DGS][DD], int i0) {
#pragma unroll
        for (int ii = 0; ii < DGS; ++ii) {
            const int i = min(i0 + ii, DD - 1);        // (rows past the last one of the last group: clamped here, skipped by the arithmetic)
            const unsigned rowterm = ((mode == 0) || ((unsigned)(y + i - R) < (unsigned)H)) ? 0u : kOut;
#pragma unroll
            for (int j = 0; j < DD; ++j) {
                const int pl = mode ? (2 * R - i) * DD + (2 * R - j) : i * DD + j;                   // (block-uniform)
                const int uni = (pl * plane + mode * ((i - R) * W + (j - R))) * 4;                    // (block-uniform: the load's scalar offset)
                w[ii][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grs, (int)(colterm[j] + rowterm), uni, 0));
            }
        }
    };
    float wcur[DGS][DD], wnext[DGS][DD];
    request(wcur, 0);                                  // in flight behind the staging below

    // stage the chunk's planes (zero padded; channels past the last one as zeros): F[c][yy][xx] lands at planes[c][yy + R][xx + R]
#pragma unroll 4
    for (int e = threadIdx.x; e < cch * pplane; e += 256) {
        const int c = e / pplane, r = e - c * pplane;
        const int yy = r / PW - R, xx = r - (r / PW) * PW - R;
        float v = 0.f;
        if (c < nch && yy >= 0 && yy < H && xx >= 0 && xx < W) v = F[((size_t)(b * C + c0 + c)) * plane + yy * W + xx];
        planes[e] = v;
    }
    __syncthreads();
    if (!live) return;                                 // (no barrier below)

    float acc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
    // channel cs + k csub of the chunk, tap (i, j) of this pixel: p[k csub pplane + i PW + j]
    const float* p = planes + cs * pplane + y * PW + x;
    const int kstride = csub * pplane;
    // One group of ROWS displacement rows: ROWS x CPL stages, a stage = the 2R + 1 taps of one (row, channel) -- its LDS row segment was requested one
    // stage earlier (two row buffers; sched_barrier keeps the requests where they are written: left alone hipcc reads a segment, waits for it and only
    // then runs its FMAs, every stage exposing the LDS latency)
    auto group = [&](auto rows_tag, int i0) {
        constexpr int ROWS = decltype(rows_tag)::value, T = ROWS * CPL;
        float row[2][DD];
        auto read_row = [&](float (&r)[DD], int t) {
            const float* rp = p + (t % CPL) * kstride + (i0 + t / CPL) * PW;
#pragma unroll
            for (int j = 0; j < DD; ++j) r[j] = rp[j];
        };
        read_row(row[0], 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T) read_row(row[(t + 1) & 1], t + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < DD; ++j) acc[t % CPL] = fmaf(wcur[t / CPL][j], row[t & 1][j], acc[t % CPL]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int ii = 0; ii < DGS; ++ii)
#pragma unroll
        for (int j = 0; j < DD; ++j) wcur[ii][j] *= inv_c;
#pragma unroll 1
    for (int gi = 0; gi < DD / DGS; ++gi) {            // the full groups
        const int i0 = gi * DGS;
        if (i0 + DGS < DD) request(wnext, i0 + DGS);   // the next group's loads fly during this group's arithmetic
        group(std::integral_constant<int, DGS>{}, i0);
        if (i0 + DGS < DD) {
#pragma unroll
            for (int ii = 0; ii < DGS; ++ii)
#pragma unroll
                for (int j = 0; j < DD; ++j) wcur[ii][j] = wnext[ii][j] * inv_c;
        }
    }
    if constexpr (DD % DGS != 0) group(std::integral_constant<int, DD % DGS>{}, (DD / DGS) * DGS);      // the short last group
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = cs + k * csub;
        if (c < nch) out[((size_t)(b * C + c0 + c)) * plane + q] = acc[k];
    }
}

// *launched = false: the shape is not this kernel's (map too large, or a chunk's planes would not fit 64 KB of LDS)
template <int R, int DGS>
int launch_bwd_smallrows(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
                         int B, int C, int H, int W, hipStream_t s, bool* launched) {
    *launched = false;
    const int plane = H * W, pplane = (H + 2 * R) * (W + 2 * R);
    if (plane > 1024 || B > 32767) return 0;
    int pxl = 32;
    while (pxl < plane && pxl < 256) pxl <<= 1;
    const int csub = 256 / pxl;
    // channels per lane: 8 where a workgroup is all pixel lanes (levels 5 of the pyramid: 8 channels per workgroup, 16 x 2B workgroups of two
    // per CU), 4 where four or more channel phases share a workgroup (level 6: 16 channels per workgroup) -- more channels per workgroup means
    // fewer repeats of the gradient gather (its L2 traffic is the gradient x the number of channel chunks), fewer means more workgroups in flight
    const int cpl = csub <= 2 ? 8 : 4;
    const size_t lds = (size_t)cpl * csub * pplane * sizeof(float);
    if (lds > 64 * 1024) return 0;
    dim3 grid(ceil_div(plane, pxl), ceil_div(C, cpl * csub), 2 * B);
    if (cpl == 8) UNFLOW_LAUNCH((corr_bwd_smallrows_kernel<R, DGS, 8>), grid, dim3(256), lds, s, f1, f2, g, gf1, gf2, C, H, W, pxl, 1.0f / C);
    else          UNFLOW_LAUNCH((corr_bwd_smallrows_kernel<R, DGS, 4>), grid, dim3(256), lds, s, f1, f2, g, gf1, gf2, C, H, W, pxl, 1.0f / C);
    *launched = true;
    return unflow_launch_status();
}

}  // namespace
