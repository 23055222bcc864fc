// LDS-DMA ring building blocks shared by the cost-volume kernels (corr.hip) and the fused warp + cost-volume
// kernels (warp_corr.hip): XCD-aware tile order, hand-issued ds_read_b64 with counted lgkmcnt waits, counted
// vmcnt waits for global_load_lds stages, and the software-pipelined row-step of the forward accumulation.
#pragma once
#include "common.h"
#include <utility>

namespace {

// Bijective remap of the linear workgroup id so that consecutive tiles share an XCD (workgroups
// are dealt round-robin over the 8 XCDs): speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7;          // XCD x gets q (+1 if x < r) workgroups
    const int xcd = lin & 7, k = lin >> 3;
    return xcd * q + (xcd < r ? xcd : r) + k;
}

typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const v2f lds_cfloat2;

__device__ __attribute__((aligned(16))) float kZeroLine[4] = {0.f, 0.f, 0.f, 0.f};    // source of out-of-image DMA slots

// Its address, fetched ONCE per wave into a scalar register pair.  Naming kZeroLine at the point of use makes hipcc
// re-derive it there every time (s_getpc + s_load through the GOT + s_waitcnt lgkmcnt(0)) and, since that is expensive,
// wrap each `ok ? p : kZeroLine` in an EXEC-masked branch: 58 scalar-memory round trips and 56 branches in the
// gradient gather of the group-split backward alone.
typedef __attribute__((address_space(1))) const float gfloat;       // explicit global address space: a laundered generic
__device__ __forceinline__ gfloat* zero_line() {                     // pointer would turn every load behind it into flat_load
    gfloat* z = (gfloat*)kZeroLine;
    asm volatile("" : "+s"(z));
    return z;
}

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

// Hand-issued LDS reads with hand-counted waits.  hipcc tracks its own ds_reads with lgkmcnt(0)
// whenever a register buffer is re-used, which exposes the full LDS latency every few rows at two
// waves per SIMD; here the row stream is software-pipelined PF row-steps deep and every step waits
// only for its own rows (lgkmcnt is a 4-bit in-order counter: <= 15 reads are kept in flight).
template <int OFF>
__device__ __forceinline__ v2f lds_read_b64(unsigned addr) {
    v2f v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // nothing that consumes the rows may move above the wait
}

template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the lane's own f1 pixels of the stage's CC channels (channel stride CH_BYTES)
template <int CH_BYTES, int CC, int... Cs>
__device__ __forceinline__ void own_reads(v2f (&a)[CC], unsigned addr, std::integer_sequence<int, Cs...>) {
    ((a[Cs] = lds_read_b64<Cs * CH_BYTES>(addr)), ...);
}

// One row-step of the forward pipeline: ST = c * DG + i  (channel-in-stage, displacement row of the group).
template <int ST, int STEPS, int PF, int DG, int DD, int NCOL, int CH_BYTES, int ROW_BYTES>
struct FwdStep {
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
    static __device__ __forceinline__ void run(float (&acc)[DG][DD][2], v2f (&row)[PF + 1][NCOL],
                                               const v2f (&a)[CC], unsigned addr) {
        if constexpr (ST < STEPS) {
            load<ST + PF>(row, addr);
            constexpr int newer = (STEPS - 1 - ST < PF ? STEPS - 1 - ST : PF) * NCOL;
            lds_wait<newer>();
            constexpr int c = ST / DG, i = ST % DG, rb = ST % (PF + 1);
            float r[2 * NCOL];
#pragma unroll
            for (int k = 0; k < NCOL; ++k) { r[2 * k] = row[rb][k].x; r[2 * k + 1] = row[rb][k].y; }
#pragma unroll
            for (int j = 0; j < DD; ++j) {
                acc[i][j][0] = fmaf(a[c].x, r[j], acc[i][j][0]);
                acc[i][j][1] = fmaf(a[c].y, r[j + 1], acc[i][j][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            FwdStep<ST + 1, STEPS, PF, DG, DD, NCOL, CH_BYTES, ROW_BYTES>::template run<CC>(acc, row, a, addr);
        }
    }
};

}  // namespace
