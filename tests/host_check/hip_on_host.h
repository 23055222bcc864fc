// A workgroup of a HIP kernel as real threads of the build host (TEST INFRASTRUCTURE: csrc/common.h includes this file instead of the HIP
// headers when it is compiled with -DUNFLOW_HOST_CHECK -I tests/host_check).
//
// What the loss kernels (csrc/photo.hip) use of the HIP dialect is small: threadIdx / blockIdx / blockDim / gridDim, __shared__ arrays,
// __syncthreads(), the 64-lane butterfly __shfl_xor of wave_sum(), and the launch macros.  Here a lane is a fiber with a stack of its own, a workgroup's lanes
// run interleaved and meet at a real barrier, __shared__ is `static`, and __shfl_xor exchanges
// through a per-workgroup array between two barriers -- every lane of a workgroup calls it at the same point, exactly as the kernels
// already require of __syncthreads().  Workgroups run one after the other.  (Lanes are FIBERS of one OS thread -- a stack each, a 15-instruction
// register switch -- run in lane order from barrier to barrier: deterministic, and a barrier costs a switch, not a system call.)  The kernels' own reduction helpers (common.h: wave_sum,
// block_sum_256, sum_partials) are compiled UNCHANGED on top of this, so even the order of their additions is the device's.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
using std::max;
using std::min;

inline dim3 threadIdx;                                            // the running lane's (set on every switch)
inline dim3 blockIdx, blockDim, gridDim;                       // one workgroup at a time: shared by its lanes

namespace hip_on_host {
inline float xchg[1024];
inline unsigned lane_id = 0;                                  // the running lane's linear id inside the workgroup
}  // namespace hip_on_host

namespace hip_on_host { inline void sync(); }
static inline void __syncthreads() { hip_on_host::sync(); }
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    static_assert(sizeof(T) == 4, "32-bit values");
    (void)width;
    memcpy(&hip_on_host::xchg[hip_on_host::lane_id], &v, 4);
    __syncthreads();
    T r; memcpy(&r, &hip_on_host::xchg[hip_on_host::lane_id ^ (unsigned)mask], 4);   // (mask < 64: the partner is in the same wave)
    __syncthreads();
    return r;
}
// barrier + OR over the workgroup's predicates
namespace hip_on_host { inline int or_acc = 0; }
static inline int __syncthreads_or(int p) {
    __syncthreads();                                              // (nobody still reads the previous result)
    hip_on_host::or_acc = 0;
    __syncthreads();
    if (p) hip_on_host::or_acc = 1;
    __syncthreads();
    return hip_on_host::or_acc;
}
// the explicitly rounded operations of warp_taps.h (built with -ffp-contract=off: no fused forms sneak in)
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }

// lane i <- lane i -/+ delta of the same 64-lane wave; a lane without a source keeps its own value (HIP's __shfl_up / __shfl_down)
template <class T> static inline T hip_on_host_shift(T v, int delta) {
    static_assert(sizeof(T) == 4, "32-bit values");
    float f; memcpy(&f, &v, 4);
    hip_on_host::xchg[hip_on_host::lane_id] = f;
    __syncthreads();
    const int lane = (int)(hip_on_host::lane_id & 63u), src = lane + delta;
    const float r = (src >= 0 && src < 64) ? hip_on_host::xchg[(hip_on_host::lane_id & ~63u) + (unsigned)src] : f;
    __syncthreads();
    T out; memcpy(&out, &r, 4);
    return out;
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64) { (void)width; return hip_on_host_shift(v, -(int)delta); }
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) { (void)width; return hip_on_host_shift(v, (int)delta); }
// the two scalar built-ins the SSIM kernels name: a wave-uniform value is itself; the hardware reciprocal (1 ulp) is the IEEE one here
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

// scheduling fences mean nothing here; LDS-DMA (global_load_lds_dwordx4: lane l of a wave moves `size` bytes from ITS global address to the
// wave's LDS base + l * size) is a memcpy; a float atomic is a plain add (fibers never run concurrently)
#define __builtin_amdgcn_sched_barrier(x)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) memcpy((char*)(l) + (hip_on_host::lane_id & 63u) * (size), (const void*)(g), (size))
static inline float atomicAdd(float* p, float v) { const float old = *p; *p = old + v; return old; }
// the raw barrier is the barrier; a 32-bit LDS address (what the hand-issued ds_read / ds_write of the cost-volume kernels take) is the low
// half of the host address of a static array: lds_at() puts the high half back (the library's statics lie within 2 GB of the anchor)
#define __builtin_amdgcn_s_barrier() __syncthreads()
namespace hip_on_host {
inline char lds_anchor;
static inline char* lds_at(unsigned addr, int /*bytes*/) {
    const uintptr_t base = (uintptr_t)&lds_anchor;
    return (char*)(base + (intptr_t)(int32_t)(addr - (uint32_t)base));
}
}  // namespace hip_on_host

// ---- what the matrix-core cost-volume backward (csrc/corr_mfma.h) names
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
// a buffer descriptor: base + size in bytes; an access that does not lie inside [0, size) reads zeros / writes nothing (the hardware's range check,
// which the kernels use as their predicate: offset 0x40000000 = "outside")
namespace hip_on_host { struct Rsrc { char* base; unsigned bytes; }; }
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) (hip_on_host::Rsrc{(char*)(p), (unsigned)(num)})
namespace hip_on_host {
template <int N> static inline void buf_load(const Rsrc& r, int voff, int soff, void* dst) {
    const unsigned long long o = (unsigned long long)(unsigned)voff + (unsigned)soff;
    if (o + N <= r.bytes) memcpy(dst, r.base + o, N); else memset(dst, 0, N);
}
template <int N> static inline void buf_store(const Rsrc& r, int voff, int soff, const void* src) {
    const unsigned long long o = (unsigned long long)(unsigned)voff + (unsigned)soff;
    if (o + N <= r.bytes) memcpy(r.base + o, src, N);
}
}  // namespace hip_on_host
#if defined(__clang__)
typedef unsigned hip_on_host_v4u __attribute__((ext_vector_type(4)));
typedef unsigned hip_on_host_v2u __attribute__((ext_vector_type(2)));
typedef float hip_on_host_v4f __attribute__((ext_vector_type(4)));
typedef __bf16 hip_on_host_bf16x8 __attribute__((ext_vector_type(8)));
static inline unsigned hip_on_host_load_b32(const hip_on_host::Rsrc& r, int v, int s) { unsigned x; hip_on_host::buf_load<4>(r, v, s, &x); return x; }
static inline hip_on_host_v2u hip_on_host_load_b64(const hip_on_host::Rsrc& r, int v, int s) { hip_on_host_v2u x; hip_on_host::buf_load<8>(r, v, s, &x); return x; }
static inline hip_on_host_v4u hip_on_host_load_b128(const hip_on_host::Rsrc& r, int v, int s) { hip_on_host_v4u x; hip_on_host::buf_load<16>(r, v, s, &x); return x; }
static inline void hip_on_host_store_b128(hip_on_host_v4u d, const hip_on_host::Rsrc& r, int v, int s) { hip_on_host::buf_store<16>(r, v, s, &d); }
#define __builtin_amdgcn_raw_buffer_load_b32(r, v, s, aux) hip_on_host_load_b32(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b64(r, v, s, aux) hip_on_host_load_b64(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b128(r, v, s, aux) hip_on_host_load_b128(r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b128(d, r, v, s, aux) hip_on_host_store_b128(d, r, v, s)
// LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... lds): lane l of a wave moves `size` bytes from the descriptor's range (zeros
// from outside it) to the wave's LDS base + l * size
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, l, size, v, s, ioff, aux) \
    hip_on_host::buf_load<(size)>(r, v, (s) + (ioff), (char*)(l) + (hip_on_host::lane_id & 63u) * (size))
// v_mfma_f32_16x16x32_bf16, one wave: D[row][col] = sum_k A[row][k] B[k][col] + C[row][col] with the lane layouts of the CDNA4 ISA --
// A: lane l holds row l & 15, k = 8 (l >> 4) .. + 7;  B: lane l holds column l & 15, the same k;  C / D: lane l holds column l & 15, rows
// 4 (l >> 4) .. + 3.  Products of bf16 values are exact in fp32; they are added here in k order, in fp32 (the hardware's order is its own:
// results can differ from the device's in the last bits, not in what is multiplied with what).
namespace hip_on_host { inline float mf_a[256][8], mf_b[256][8]; }
static inline hip_on_host_v4f hip_on_host_mfma_16x16x32_bf16(hip_on_host_bf16x8 a, hip_on_host_bf16x8 b, hip_on_host_v4f c) {
    const unsigned me = hip_on_host::lane_id, wave0 = me & ~63u, l = me & 63u;
    for (int e = 0; e < 8; ++e) { hip_on_host::mf_a[me][e] = (float)a[e]; hip_on_host::mf_b[me][e] = (float)b[e]; }
    __syncthreads();
    hip_on_host_v4f d = c;
    const unsigned col = l & 15u;
    for (int reg = 0; reg < 4; ++reg) {
        const unsigned row = 4u * (l >> 4) + (unsigned)reg;
        float acc = c[reg];
        for (int k = 0; k < 32; ++k) acc += hip_on_host::mf_a[wave0 + row + 16u * (unsigned)(k >> 3)][k & 7] * hip_on_host::mf_b[wave0 + col + 16u * (unsigned)(k >> 3)][k & 7];
        d[reg] = acc;
    }
    __syncthreads();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hip_on_host_mfma_16x16x32_bf16(a, b, c)
#endif

// ---- the host forms of csrc/device_forms.h (the product's one header of gfx950-only forms: common.h includes THIS file instead of it in the test
// build, so every name defined there is defined here).  Register pins and counted waits mean nothing -- every load has landed when its call
// returns --, the occupancy attribute is dropped, address-space pointers are plain pointers, and a wave's lock step needs marking because the
// lanes of a wave run one after the other.
#define UNFLOW_PIN_VGPR(x) ((void)(x))
#define UNFLOW_PIN_SGPR(x) ((void)(x))
#define UNFLOW_WAIT_LGKMCNT0()
#define UNFLOW_WAIT_VMCNT0()
#define UNFLOW_WAVES_PER_EU(n)
#define UNFLOW_WAVE_LOCKSTEP() __syncthreads()
typedef const void* wgas_ptr;
typedef void* wlds_ptr;
namespace {
typedef const float lds_cfloat;
typedef const float gfloat;
typedef const void* gas_ptr;
typedef void* lds_ptr;
inline __attribute__((aligned(16))) float kZeroLine[4] = {0.f, 0.f, 0.f, 0.f};
static inline gfloat* zero_line() { return kZeroLine; }
template <int N> static inline void lds_wait() {}
template <int N> static inline void vm_wait() {}
// the wave shift by one lane is an exchange between the fibers (0 enters at the wave's ends)
static inline float from_lane_below(float v) { const float r = __shfl_up(v, 1, 64); return (threadIdx.x & 63) == 0 ? 0.f : r; }
static inline float from_lane_above(float v) { const float r = __shfl_down(v, 1, 64); return (threadIdx.x & 63) == 63 ? 0.f : r; }
// a 32-bit "LDS address" is the low half of the host address (lds_at() finds the object again): the hand-issued reads and writes are plain
// loads and stores at the same addresses
static inline void lds_write_b32(unsigned addr, float v) { memcpy(hip_on_host::lds_at(addr, 4), &v, 4); }
#if defined(__clang__)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const v2f lds_cfloat2;
template <int OFF>
static inline v2f lds_read_b64(unsigned addr) { v2f v; memcpy(&v, hip_on_host::lds_at(addr + OFF, 8), 8); return v; }
template <int OFF0, int OFF1>
static inline v2f lds_read2_b32(unsigned addr) {
    float lo, hi;
    memcpy(&lo, hip_on_host::lds_at(addr + 4 * OFF0, 4), 4);
    memcpy(&hi, hip_on_host::lds_at(addr + 4 * OFF1, 4), 4);
    return v2f{lo, hi};
}
template <int OFF>
static inline void lds_write_b64(unsigned addr, v2f v) { memcpy(hip_on_host::lds_at(addr + OFF, 8), &v, 8); }
#endif
}  // namespace

// dynamic LDS (common.h: UNFLOW_DYNAMIC_LDS): one block of the size of a CU's LDS; raising a kernel's dynamic-LDS limit is a no-op
namespace hip_on_host { alignas(16) inline unsigned char dynamic_lds[160 * 1024]; }
#define UNFLOW_DYNAMIC_LDS(T, name) T* name = reinterpret_cast<T*>(hip_on_host::dynamic_lds)
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
static inline int hipFuncSetAttribute(const void*, int, int) { return 0; }

// ---- the runtime names the entry points use
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
#define hipSuccess 0
static inline int hipGetLastError() { return 0; }
static inline int hipEventCreate(hipEvent_t* e) { static char token; *e = &token; return 0; }      // (non-null: an armed timing slot counts its launches; the time is a made-up 1 us)
static inline int hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return 0; }      // (1 us: time means nothing here, but a rate computed from it must not divide by zero)

// a fiber switch without system calls (ucontext's swapcontext saves the signal mask: a syscall per switch): callee-saved registers on the
// old stack, stack pointers exchanged (x86-64 System V)
extern "C" void hip_on_host_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.weak hip_on_host_switch
.type hip_on_host_switch,@function
hip_on_host_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace hip_on_host {
struct Lane { void* sp = nullptr; std::vector<char> stack; dim3 tid; bool done = false, at_barrier = false; };
inline std::vector<Lane> lanes;
inline void* scheduler_sp = nullptr;
inline std::function<void()> body;
inline void lane_entry() {
    body();
    lanes[lane_id].done = true;
    hip_on_host_switch(&lanes[lane_id].sp, scheduler_sp);
    abort();                                                   // (a finished lane is never resumed)
}

template <class K, class... A>
static void launch(K kernel, dim3 grid, dim3 block, A... args) {
    const unsigned n = block.x * block.y * block.z;
    gridDim = grid; blockDim = block;
    body = [&] { kernel(args...); };
    lanes.resize(n);
    for (unsigned t = 0; t < n; ++t) {
        if (lanes[t].stack.size() < 192 * 1024) lanes[t].stack.resize(192 * 1024);
        lanes[t].tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    }
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned t = 0; t < n; ++t) {
            Lane& l = lanes[t];
            l.done = l.at_barrier = false;
            // a fresh stack whose first `ret` enters lane_entry with the alignment of a call: [r15 r14 r13 r12 rbx rbp][lane_entry] below a 16-byte boundary + 8
            uintptr_t top = ((uintptr_t)(l.stack.data() + l.stack.size()) & ~(uintptr_t)15) - 8;
            void** sp = (void**)top;
            *--sp = (void*)&lane_entry;
            for (int r = 0; r < 6; ++r) *--sp = nullptr;
            l.sp = sp;
        }
        // lanes in order, each until its next barrier (or its end); when every lane still alive waits at the barrier, all go on
        for (bool any = true; any;) {
            any = false;
            for (unsigned t = 0; t < n; ++t) {
                Lane& l = lanes[t];
                if (l.done || l.at_barrier) continue;
                lane_id = t; threadIdx = l.tid;
                hip_on_host_switch(&scheduler_sp, l.sp);
                any = true;
            }
            bool waiting = false;
            for (auto& l : lanes) waiting |= (!l.done && l.at_barrier);
            if (waiting) { for (auto& l : lanes) l.at_barrier = false; any = true; }
        }
    }
}
inline void sync() { const unsigned me = lane_id; lanes[me].at_barrier = true; hip_on_host_switch(&lanes[me].sp, scheduler_sp); lane_id = me; threadIdx = lanes[me].tid; }
}  // namespace hip_on_host
// HIP_ON_HOST_TRACE=1: one line per launch on stderr -- the kernel as the launch site names it (template arguments as written there) and its grid
namespace hip_on_host {
inline void trace(const char* kernel, dim3 grid, dim3 block) {
    static const bool on = getenv("HIP_ON_HOST_TRACE") != nullptr;
    if (on) fprintf(stderr, "launch %s grid %u %u %u block %u\n", kernel, grid.x, grid.y, grid.z, block.x * block.y * block.z);
}
}  // namespace hip_on_host
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) (hip_on_host::trace(#kernel, grid, block), hip_on_host::launch(kernel, grid, block, __VA_ARGS__))
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, start, stop, flags, ...) (hip_on_host::trace(#kernel, grid, block), hip_on_host::launch(kernel, grid, block, __VA_ARGS__))
