// The gfx950 forms of everything the kernel sources name that has a second, host-executed form in the test build (tests/host_check/hip_on_host.h:
// lanes as fibers of the build host; -DUNFLOW_HOST_CHECK makes common.h include that header INSTEAD of this one).  The product is built from this
// file alone; the only other mention of the test build in csrc/ is the include switch of common.h (and the include guards of multiscale.h /
// warp_taps.h, which a g++-only check program compiles without common.h).
#pragma once

// a kernel's dynamic LDS array
#define UNFLOW_DYNAMIC_LDS(T, name) extern __shared__ T name[]

// hipcc idioms: keep a value in a vector / scalar register where it is (an empty asm the optimiser cannot see through), drain this wave's
// LDS / scalar-memory counter resp. its vector-memory counter
#define UNFLOW_PIN_VGPR(x) asm volatile("" : "+v"(x))
#define UNFLOW_PIN_SGPR(x) asm volatile("" : "+s"(x))
#define UNFLOW_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define UNFLOW_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define UNFLOW_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))

// A wave executes its LDS accesses in program order for all 64 lanes at once: the table writes of a round (corr_mfma.h) are complete before any
// lane reads its fragment, with no barrier.  (Lanes that run one after the other -- the host check -- need the point marked.)
#define UNFLOW_WAVE_LOCKSTEP()

typedef __attribute__((address_space(1))) const void* wgas_ptr;      // (warp.hip: source / destination of an LDS-DMA piece)
typedef __attribute__((address_space(3))) void* wlds_ptr;

namespace {

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
    UNFLOW_PIN_SGPR(z);
    return z;
}

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

// Hand-issued LDS accesses with hand-counted waits.  hipcc tracks its own ds_reads with lgkmcnt(0)
// whenever a register buffer is re-used, which exposes the full LDS latency every few rows at two
// waves per SIMD; the kernels software-pipeline their row streams and every step waits
// only for its own rows (lgkmcnt is a 4-bit in-order counter: <= 15 reads are kept in flight).
template <int OFF>
__device__ __forceinline__ v2f lds_read_b64(unsigned addr) {
    v2f v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF0, int OFF1>
__device__ __forceinline__ v2f lds_read2_b32(unsigned addr) {        // the two dwords at addr + 4 OFF0 and addr + 4 OFF1
    v2f v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(OFF0), "n"(OFF1));
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_write_b64(unsigned addr, v2f v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_write_b32(unsigned addr, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // nothing that consumes the rows may move above the wait
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// wave shifts by one lane (ssim.hip): DPP row shifts across the whole wave, not hipcc's __shfl (= ds_bpermute)
__device__ __forceinline__ float from_lane_below(float v) {       // lane i <- lane i-1 (0 into lane 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_lane_above(float v) {       // lane i <- lane i+1 (0 into lane 63)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

}  // namespace
