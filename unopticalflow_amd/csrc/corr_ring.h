// LDS-DMA ring building blocks shared by the cost-volume kernels (corr.hip) and the fused warp + cost-volume
// kernels (warp_corr.hip): XCD-aware tile order, hand-issued ds_read_b64 with counted lgkmcnt waits, counted
// vmcnt waits for global_load_lds stages, and the software-pipelined row-step of the forward accumulation.
#pragma once
#include "common.h"
#include <utility>

namespace {

// (address-space typedefs, zero_line(), the hand-issued ds_read / ds_write forms and the counted waits: device_forms.h through common.h)

// the lane's own f1 pixels of the stage's CC channels (channel stride CH_BYTES)
template <int CH_BYTES, int CC, int... Cs>
__device__ __forceinline__ void own_reads(v2f (&a)[CC], unsigned addr, std::integer_sequence<int, Cs...>) {
    ((a[Cs] = lds_read_b64<Cs * CH_BYTES>(addr)), ...);
}

// The forward accumulators of a lane (DG displacement rows x DD displacements x 2 pixels), laid out for packed FMAs.
// A lane owns pixels x (even) and x + 1 and reads a halo row as R + 1 aligned float pairs row[k] = (f2[x+2k], f2[x+2k+1]).
// cv[i][j] of pixel x needs f2[x + j], of pixel x + 1 f2[x + 1 + j]: pairing pixel x's displacements (0,1)(2,3).. and
// pixel x + 1's (1,2)(3,4).. makes BOTH walk the same aligned pairs, with the own-pixel value broadcast:
//   p0[i][k]   = (cv[i][2k],   cv[i][2k+1]) of pixel x      += a.x * row[k]       k = 0 .. R-1;   s0[i] = cv[i][2R]  += a.x * row[R].x
//   p1[i][k-1] = (cv[i][2k-1], cv[i][2k])   of pixel x + 1  += a.y * row[k]       k = 1 .. R;     s1[i] = cv[i][0]   += a.y * row[0].y
// = 2R v_pk_fma_f32 + 2 v_fma_f32 per row-step instead of 2 (2R + 1) v_fma_f32, no register moves.
// MEASURED: the packed form wins in isolation (level 2: 35.8 -> 34.3 us back-to-back launches) and LOSES inside the
// train step (41.8 vs 36.3 us between MIOpen's MFMA kernels, tools/gpu_r2_j.sh) -- v_pk_fma_f32 draws more power and
// the clock it gets there is lower (tools/probes/valu_rate.hip: same tick count, 1.65x the wall time) -- so the forward
// row pipeline ships with scalar FMAs on this accumulator layout (-DUNFLOW_PK_FWD builds the packed form); the
// backward kernel, with a third of the accumulators and three waves per SIMD, keeps the packed form (93.1 vs 94.7 us).
template <int DG, int R>
struct FwdAcc {
    v2f p0[DG][R];
    float s0[DG];
    v2f p1[DG][R];
    float s1[DG];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < DG; ++i) {
#pragma unroll
            for (int k = 0; k < R; ++k) { p0[i][k] = v2f{0.f, 0.f}; p1[i][k] = v2f{0.f, 0.f}; }
            s0[i] = 0.f; s1[i] = 0.f;
        }
    }
    // constant indices after unrolling: these are register names, not addressing
    __device__ __forceinline__ float get(int i, int j, int p) const {
        if (p == 0) return j == 2 * R ? s0[i] : ((j & 1) ? p0[i][j / 2].y : p0[i][j / 2].x);
        return j == 0 ? s1[i] : (((j + 1) & 1) ? p1[i][(j + 1) / 2 - 1].y : p1[i][(j + 1) / 2 - 1].x);
    }
    __device__ __forceinline__ void add(int i, int j, int p, float v) {
        if (p == 0) {
            if (j == 2 * R) s0[i] += v;
            else if (j & 1) p0[i][j / 2].y += v;
            else p0[i][j / 2].x += v;
        } else {
            if (j == 0) s1[i] += v;
            else if ((j + 1) & 1) p1[i][(j + 1) / 2 - 1].y += v;
            else p1[i][(j + 1) / 2 - 1].x += v;
        }
    }
};

// One row-step of the forward pipeline: ST = c * DG + i  (channel-in-stage, displacement row of the group).
template <int ST, int STEPS, int PF, int DG, int DD, int NCOL, int CH_BYTES, int ROW_BYTES>
struct FwdStep {
    static constexpr int R = (DD - 1) / 2;
    static_assert(NCOL == R + 1, "a halo row is R + 1 float pairs");
    template <int Q, int... Ks>
    static __device__ __forceinline__ void load_cols(v2f (&row)[PF + 1][NCOL], unsigned addr,
                                                     std::integer_sequence<int, Ks...>) {
        constexpr int off = (Q / DG) * CH_BYTES + (Q % DG) * ROW_BYTES;
        ((row[Q % (PF + 1)][Ks] = lds_read_b64<off + 8 * Ks>(addr)), ...);      // 2 px + 2R halo floats = NCOL float2 columns
    }
    template <int Q>
    static __device__ __forceinline__ void load(v2f (&row)[PF + 1][NCOL], unsigned addr) {
        if constexpr (Q < STEPS) load_cols<Q>(row, addr, std::make_integer_sequence<int, NCOL>{});
    }
    template <int CC>
    static __device__ __forceinline__ void run(FwdAcc<DG, R>& acc, v2f (&row)[PF + 1][NCOL],
                                               const v2f (&a)[CC], unsigned addr) {
        if constexpr (ST < STEPS) {
            load<ST + PF>(row, addr);
            constexpr int newer = (STEPS - 1 - ST < PF ? STEPS - 1 - ST : PF) * NCOL;
            lds_wait<newer>();
            constexpr int c = ST / DG, i = ST % DG, rb = ST % (PF + 1);
            const v2f ax = v2f{a[c].x, a[c].x}, ay = v2f{a[c].y, a[c].y};
#pragma unroll
            for (int k = 0; k < R; ++k) {
#ifdef UNFLOW_PK_FWD         // experiment (tools/, UNFLOW_TUNING_EXTRA_FLAGS): packed FMAs on the same accumulators
                acc.p0[i][k] = __builtin_elementwise_fma(ax, row[rb][k], acc.p0[i][k]);
                acc.p1[i][k] = __builtin_elementwise_fma(ay, row[rb][k + 1], acc.p1[i][k]);
#else
                acc.p0[i][k].x = fmaf(a[c].x, row[rb][k].x, acc.p0[i][k].x); acc.p0[i][k].y = fmaf(a[c].x, row[rb][k].y, acc.p0[i][k].y);
                acc.p1[i][k].x = fmaf(a[c].y, row[rb][k + 1].x, acc.p1[i][k].x); acc.p1[i][k].y = fmaf(a[c].y, row[rb][k + 1].y, acc.p1[i][k].y);
#endif
            }
            acc.s0[i] = fmaf(a[c].x, row[rb][R].x, acc.s0[i]);
            acc.s1[i] = fmaf(a[c].y, row[rb][0].y, acc.s1[i]);
            __builtin_amdgcn_sched_barrier(0);
            FwdStep<ST + 1, STEPS, PF, DG, DD, NCOL, CH_BYTES, ROW_BYTES>::template run<CC>(acc, row, a, addr);
        }
    }
};

}  // namespace
