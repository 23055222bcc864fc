// Cost volume (correlation) forward / backward for gfx950.
//
// Replaces PWC_tf.corr_naive (reference core/networks/structures/pwc_tf.py:97-106) and its
// autograd: 81 (mul, mean, cat) launches become one kernel that reads f1 and f2 once.
//
//   cv[b, i*DD+j, y, x] = (1/C) sum_c f1[b,c,y,x] * f2[b,c,y+i-R,x+j-R]     DD = 2R+1
//
// Layout: NCHW fp32, x contiguous.  Kernel families (chosen per shape by pick_variant(), measured):
//   * tile kernels  corr_fwd_kernel / corr_bwd_kernel<R,PX,DG,CC>: a 256-thread workgroup owns a
//     (32*PX) x 8 pixel tile; CC channels of the f2 halo tile (and of f1) are staged in LDS through
//     registers; a lane keeps PX pixels x DG x DD accumulators (forward) or upstream-gradient values
//     (backward) in VGPRs and walks the LDS rows.  Any W; radii 1, 2, 4, 8.
//   * ring kernels  corr_fwd_ring_kernel<R,CC,DG>, corr_bwd_gs_kernel<R,CC>: same arithmetic, but the
//     channels stream through an LDS ring filled by global_load_lds_dwordx4 (LDS-DMA) several stages
//     ahead, with hand-counted s_waitcnt vmcnt / lgkmcnt and an XCD-aware tile order.  W % 4 == 0.
//   * generic kernels: one lane per output element (tiny maps with many channels; any radius).
// The backward is the same gather twice: gf1 = sum_ij g[ij] * f2(shifted), and
// gf2 = sum_ij g[ij](shifted back) * f1(shifted back), written as a gather -- no float atomics
// (except in the tile kernel when the displacement rows are split over workgroups, R > 4).
#include "common.h"
#include "corr_ring.h"
#include "corr_mfma.h"
#include "corr_small_rows.h"
#include <stdlib.h>
#include <utility>

namespace {

constexpr int TXT = 32;   // lanes along x
constexpr int TY = 8;     // rows per workgroup
#ifndef BWD_PIN
#define BWD_PIN 0
#endif

template <int R, int PX, int DG, int CC>
struct CorrCfg {
    static constexpr int DD = 2 * R + 1;
    static constexpr int TW = TXT * PX;
    static constexpr int LW = TW + 2 * R + ((TW + 2 * R) & 1);   // even: rows stay 8-byte aligned
    static constexpr int LH = TY + DG - 1;
    static constexpr int NG = (DD + DG - 1) / DG;                // displacement-row groups
};

// Stage CC channels of an (LH x LW) tile of `src` (sample b, channels c0..) whose top-left pixel is
// (ytop, xleft); zero outside the image / channel range.  When rows are 16-byte aligned
// (W % 4 == 0, xleft % 4 == 0, LW % 4 == 0) the copy runs as global_load_dwordx4 -> ds_write_b128.
template <int LH, int LW, int CC>
__device__ __forceinline__ void stage_tile(float (*tile)[LH][LW], const float* __restrict__ src,
                                           int b, int C, int H, int W, int c0, int ytop, int xleft) {
    constexpr int PLANE = LH * LW;
    if ((LW % 4 == 0) && ((W & 3) == 0) && ((xleft & 3) == 0)) {
        constexpr int LW4 = LW / 4, PLANE4 = LH * LW4;
        for (int idx = threadIdx.x; idx < CC * PLANE4; idx += 256) {
            const int c = idx / PLANE4;
            const int r = idx - c * PLANE4;
            const int ly = r / LW4;
            const int lx = (r - ly * LW4) * 4;
            const int gy = ytop + ly, gx = xleft + lx, gc = c0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gc < C && gy >= 0 && gy < H && gx >= 0 && gx < W)      // gx % 4 == 0 and W % 4 == 0: all or nothing
                v = *reinterpret_cast<const float4*>(src + ((size_t)(b * C + gc) * H + gy) * W + gx);
            *reinterpret_cast<float4*>(&tile[c][ly][lx]) = v;
        }
        return;
    }
    for (int idx = threadIdx.x; idx < CC * PLANE; idx += 256) {
        const int c = idx / PLANE;
        const int r = idx - c * PLANE;
        const int ly = r / LW;
        const int lx = r - ly * LW;
        const int gy = ytop + ly, gx = xleft + lx, gc = c0 + c;
        float v = 0.f;
        if (gc < C && gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = src[((size_t)(b * C + gc) * H + gy) * W + gx];
        tile[c][ly][lx] = v;
    }
}


// 3-D grid -> XCD-remapped (bx, by, bz): neighbouring tiles of one sample share an XCD's L2 (halo reuse).
__device__ __forceinline__ void remapped_block(int& bx, int& by, int& bz) {
    const int total = gridDim.x * gridDim.y * gridDim.z;
    int t = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), total);
    bx = t % gridDim.x; t /= gridDim.x;
    by = t % gridDim.y;
    bz = t / gridDim.y;
}

template <int N, int PX>
__device__ __forceinline__ void load_row(float (&v)[N], const float* row) {
    if constexpr (PX == 2) {
        const float2* r2 = reinterpret_cast<const float2*>(row);
#pragma unroll
        for (int k = 0; k < N / 2; ++k) { float2 t = r2[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = row[k];
    }
}


// A lane's window into one LDS row: NROW consecutive floats starting at its first pixel.
// PX == 2: the window is only 8-byte aligned for odd lanes, so it must be read as NROW/2
// separate ds_read_b64.  Left to itself hipcc fuses neighbouring reads into ds_read2_b64 /
// ds_read2_b32 (half the LDS rate); laundering one base pointer per float2 column through an
// empty asm hides the adjacency, and the row / channel displacement folds into the offset field.

template <int NROW, int PX>
struct RowReader {
    static constexpr int NP = (PX == 2) ? NROW / 2 : NROW;
    lds_cfloat* col[NP];
    __device__ __forceinline__ explicit RowReader(const float* first) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            lds_cfloat* p = (lds_cfloat*)(first + (PX == 2 ? 2 * k : k));   // generic -> LDS address space
            UNFLOW_PIN_VGPR(p);
            col[k] = p;
        }
    }
    // floats [off, off + NROW) relative to `first`
    __device__ __forceinline__ void read(float (&v)[NROW], int off) const {
        if constexpr (PX == 2) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const v2f t = *(lds_cfloat2*)(col[k] + off);
                v[2 * k] = t.x; v[2 * k + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) v[k] = col[k][off];
        }
    }
};

template <int R, int PX, int DG, int CC>
__global__ __launch_bounds__(256, 2) void corr_fwd_kernel(const float* __restrict__ f1,
                                                       const float* __restrict__ f2,
                                                       float* __restrict__ cv, int C, int H, int W,
                                                       float inv_c) {
    using K = CorrCfg<R, PX, DG, CC>;
    constexpr int DD = K::DD, LW = K::LW, LH = K::LH, NG = K::NG, TW = K::TW;
    constexpr int NROW = PX + 2 * R;
    __shared__ __attribute__((aligned(16))) float tile[CC][LH][LW];     // f2 with halo
    __shared__ __attribute__((aligned(16))) float own[CC][TY][TW];      // f1

    const int tx = threadIdx.x & (TXT - 1), ty = threadIdx.x >> 5;
    int bx, by, bz;
    remapped_block(bx, by, bz);
    const int b = bz / NG, grp = bz - b * NG;
    const int i0 = grp * DG;
    const int x0 = bx * TW, y0 = by * TY;
    const int px = x0 + tx * PX, py = y0 + ty;

    float acc[DG][DD][PX];
#pragma unroll
    for (int i = 0; i < DG; ++i)
#pragma unroll
        for (int j = 0; j < DD; ++j)
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[i][j][p] = 0.f;

    const RowReader<NROW, PX> rows(&tile[0][ty][tx * PX]);
    for (int c0 = 0; c0 < C; c0 += CC) {
        stage_tile<LH, LW, CC>(tile, f2, b, C, H, W, c0, y0 - R + i0, x0 - R);
        stage_tile<TY, TW, CC>(own, f1, b, C, H, W, c0, y0, x0);
        __syncthreads();
        // One channel per trip keeps the loop body (DG*DD*PX FMAs + DG row reads) inside the
        // instruction cache; fully unrolled over CC it ran fetch-bound.
#pragma unroll 1
        for (int c = 0; c < CC; ++c) {
            float a[PX];
            load_row<PX, PX>(a, &own[c][ty][tx * PX]);
            // rows are double-buffered by hand and pinned with sched_barrier: left alone the
            // scheduler hoists all DG row reads (DG*NROW VGPRs) and drops to one wave per SIMD
            float row[2][NROW];
            rows.read(row[0], c * (LH * LW));
#pragma unroll
            for (int i = 0; i < DG; ++i) {
                if (i + 1 < DG) rows.read(row[(i + 1) & 1], c * (LH * LW) + (i + 1) * LW);
#pragma unroll
                for (int j = 0; j < DD; ++j)
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[i][j][p] = fmaf(a[p], row[i & 1][j + p], acc[i][j][p]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    if (py >= H) return;
    const size_t plane = (size_t)H * W;
    float* out = cv + ((size_t)b * DD * DD) * plane + (size_t)py * W + px;
#pragma unroll
    for (int i = 0; i < DG; ++i) {
        if (i0 + i >= DD) break;
#pragma unroll
        for (int j = 0; j < DD; ++j) {
            float* o = out + (size_t)((i0 + i) * DD + j) * plane;
            if (PX == 2 && px + 1 < W && (W & 1) == 0) {
                *reinterpret_cast<float2*>(o) = make_float2(acc[i][j][0] * inv_c, acc[i][j][PX - 1] * inv_c);
            } else {
#pragma unroll
                for (int p = 0; p < PX; ++p)
                    if (px + p < W) o[p] = acc[i][j][p] * inv_c;
            }
        }
    }
}

// MODE 0: gf1[b,c,q] = (1/C) sum_ij g[ij][q]          * f2[c][q + (i-R, j-R)]
// MODE 1: gf2[b,c,q] = (1/C) sum_ij g[ij][q-(i-R,j-R)] * f1[c][q - (i-R, j-R)]
//   rewritten with i' = 2R-i, j' = 2R-j so both read F at q + (i'-R, j'-R).
template <int R, int PX, int DG, int CC, int MODE>
__global__ __launch_bounds__(256, 2) void corr_bwd_kernel(const float* __restrict__ F,
                                                       const float* __restrict__ g,
                                                       float* __restrict__ out, int C, int H, int W,
                                                       float inv_c) {
    using K = CorrCfg<R, PX, DG, CC>;
    constexpr int DD = K::DD, LW = K::LW, LH = K::LH, NG = K::NG;
    constexpr int NROW = PX + 2 * R;
    __shared__ __attribute__((aligned(16))) float tile[CC][LH][LW];

    const int tx = threadIdx.x & (TXT - 1), ty = threadIdx.x >> 5;
    int bx, by, bz;
    remapped_block(bx, by, bz);
    const int b = bz / NG, grp = bz - b * NG;
    const int i0 = grp * DG;
    const int x0 = bx * K::TW, y0 = by * TY;
    const int px = x0 + tx * PX, py = y0 + ty;
    const size_t plane = (size_t)H * W;

    float wr[DG][DD][PX];
#pragma unroll
    for (int i = 0; i < DG; ++i)
#pragma unroll
        for (int j = 0; j < DD; ++j)
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                const int ii = i0 + i;
                float v = 0.f;
                if (ii < DD && py < H && px + p < W) {
                    if (MODE == 0) {
                        v = g[((size_t)b * DD * DD + ii * DD + j) * plane + (size_t)py * W + px + p];
                    } else {
                        const int sy = py + ii - R, sx = px + p + j - R;
                        if (sy >= 0 && sy < H && sx >= 0 && sx < W)
                            v = g[((size_t)b * DD * DD + (2 * R - ii) * DD + (2 * R - j)) * plane +
                                  (size_t)sy * W + sx];
                    }
                }
                wr[i][j][p] = v;
            }

    const RowReader<NROW, PX> rows(&tile[0][ty][tx * PX]);
    for (int c0 = 0; c0 < C; c0 += CC) {
        stage_tile<LH, LW, CC>(tile, F, b, C, H, W, c0, y0 - R + i0, x0 - R);
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < CC; ++c) {
            float acc[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[p] = 0.f;
            float row[2][NROW];
            rows.read(row[0], c * (LH * LW));
#pragma unroll
            for (int i = 0; i < DG; ++i) {
                if (i + 1 < DG) rows.read(row[(i + 1) & 1], c * (LH * LW) + (i + 1) * LW);
#pragma unroll
                for (int j = 0; j < DD; ++j)
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[p] = fmaf(wr[i][j][p], row[i & 1][j + p], acc[p]);
                if (BWD_PIN) __builtin_amdgcn_sched_barrier(0);
            }
            if (c0 + c < C && py < H) {
                float* o = out + ((size_t)(b * C + c0 + c) * H + py) * W + px;
                if (NG == 1 && PX == 2 && px + 1 < W && (W & 1) == 0) {
                    *reinterpret_cast<float2*>(o) = make_float2(acc[0] * inv_c, acc[PX - 1] * inv_c);
                } else {
#pragma unroll
                    for (int p = 0; p < PX; ++p)
                        if (px + p < W) {
                            if (NG == 1) o[p] = acc[p] * inv_c;
                            else atomicAdd(o + p, acc[p] * inv_c);
                        }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Large-map forward, LDS-DMA ring: 64x8 pixel tile, 2 px per lane, DG of the DD displacement rows
// per workgroup (blockIdx.y picks the group; DG = DD: all 81 displacements in one workgroup).  The
// channels stream through a 4-slot LDS ring filled by global_load_lds_dwordx4 (no staging VGPRs, no
// ds_write) with THREE stages in flight behind a counted s_waitcnt vmcnt, one raw s_barrier per
// stage; rows are read with hand-issued ds_read_b64 two row-steps ahead of their FMAs; the tile
// order is XCD-aware so the halos shared by neighbouring tiles are served by one XCD's L2.
// Requires W % 4 == 0.
// ---------------------------------------------------------------------------------------------
template <int R, int CC, int DG, int TYB_ = 8>
struct RingCfg {
    static constexpr int DD = 2 * R + 1;
    static constexpr int NG = (DD + DG - 1) / DG;
    static constexpr int TYB = TYB_, TWL = 256 / TYB, TW = 2 * TWL, NS = 4;     // 64 x 8 tiles, or 16 x 32 (TYB 32: the narrow last column of a map)
    static_assert(TWL == 32 || TWL == 8, "lane mappings of the kernel");
    static constexpr int LW = TW + 2 * R, LH = TYB + DG - 1;
    // float4 slots per channel: the f2 halo tile (padded to whole 64-slot pieces, so that every wave-instruction of the
    // DMA reads ONE tensor and can go through that tensor's buffer descriptor), then the f1 tile
    static constexpr int S2 = LH * LW / 4, S2P = (S2 + 63) / 64 * 64, S1 = TYB * TW / 4, SC = S2P + S1;
    static_assert(S1 % 64 == 0, "the f1 tile is whole DMA pieces");
    static constexpr int ITER = (CC * SC + 255) / 256;
    static constexpr int STAGE = ITER * 256 * 4;                                // floats per ring slot
    static constexpr int WAVES = (DG * DD * 2 <= 96 && CC <= 2) ? 3 : 2;        // occupancy the registers allow
};

// Phase-ablation switches of the ring kernel (loads / compute / stores only) exist in tuning builds
// (-DUNFLOW_TUNING, tools/) alone: the shipped library has no environment-dependent behaviour.
#ifdef UNFLOW_TUNING
#define RING_DBG(bit) (dbg & (bit))
#else
#define RING_DBG(bit) 0
#endif

// In-kernel stamps (tuning builds, UNFLOW_STAMP_PTR = device address of 1024 x 3 x 8 u64): cycles per named segment,
// summed per wave with s_memtime; slot 6 / 7 of a row hold the wave's first and last stamp (the timeline across
// workgroups).  32-bit counters; never place a stamp inside a hand-counted lgkmcnt / vmcnt section (its own
// s_waitcnt lgkmcnt(0) drains the LDS reads).
#ifdef UNFLOW_TUNING
#define STAMP_DECL() unsigned seg[6] = {0, 0, 0, 0, 0, 0}, tlast; unsigned long long t_first, t_now; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_first) :: "memory"); tlast = (unsigned)t_first; t_now = t_first
#define STAMP(IDX) do { __builtin_amdgcn_sched_barrier(0); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
        seg[IDX] += (unsigned)t_now - tlast; tlast = (unsigned)t_now; } while (0)
#define STAMP_WRITE(buf, row, lane0) do { if ((buf) && (lane0) && (row) < 1024 * 3) { \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) (buf)[(size_t)(row) * 8 + q_] = seg[q_]; \
        (buf)[(size_t)(row) * 8 + 6] = t_first; (buf)[(size_t)(row) * 8 + 7] = t_now; } } while (0)
#else
#define STAMP_DECL() do { } while (0)
#define STAMP(IDX) do { } while (0)
#define STAMP_WRITE(buf, row, lane0) do { } while (0)
#endif
static inline unsigned long long* stamp_buffer() {
#ifdef UNFLOW_TUNING
    if (const char* sp = getenv("UNFLOW_STAMP_PTR")) return (unsigned long long*)strtoull(sp, nullptr, 0);
#endif
    return nullptr;
}

// The DMA slots of one ring stage (forward kernels): slot s = it * 256 + tid covers 16 bytes; within a channel the
// first S2P slots are the f2 halo tile (S2 real ones), the rest the f1 tile.  Per lane: the byte offset of its slot inside
// the stage's first CC planes of a sample, or kOutOfRange (outside the image / padding / beyond CC) -- the buffer
// descriptor's range check turns those, and channels past the last one, into zeros.  Per wave-instruction: which tensor.
constexpr unsigned kOutOfRange = 0x40000000u;       // both tensors' per-sample extents stay below this (fwd_offsets_fit)

template <class K, int R, int CC>
struct FwdSlots {
    unsigned off[K::ITER];
    bool from_f2[K::ITER];                          // wave-uniform
    __device__ __forceinline__ void setup(int tid, int wave, int x0, int y0, int i0, int H, int W, unsigned plane) {
        constexpr int LW4 = K::LW / 4, TW4 = K::TW / 4;
#pragma unroll
        for (int it = 0; it < K::ITER; ++it) {
            const int s = it * 256 + tid;
            const int c = s / K::SC;
            int r = s - c * K::SC;
            int gy, gx;
            bool real = c < CC;
            if (r < K::S2P) { const int ly = r / LW4; gy = y0 - R + i0 + ly; gx = x0 - R + (r - ly * LW4) * 4; real = real && r < K::S2; }
            else { r -= K::S2P; const int ly = r / TW4; gy = y0 + ly; gx = x0 + (r - ly * TW4) * 4; }
            const bool in = real & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            off[it] = in ? ((unsigned)c * plane + (unsigned)(gy * W + gx)) * 4u : kOutOfRange;
            UNFLOW_PIN_VGPR(off[it]);       // materialise once; do not re-derive in the loop
            from_f2[it] = ((it * 256 + wave * 64) % K::SC) < K::S2P;
        }
    }
    template <class RS>
    __device__ __forceinline__ void issue(const RS& rs1, const RS& rs2, float* dst, int wave, unsigned stageB) const {
#pragma unroll
        for (int it = 0; it < K::ITER; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(from_f2[it] ? rs2 : rs1, (lds_ptr)(dst + (it * 256 + wave * 64) * 4), 16,
                                                     (int)(off[it] + stageB), 0, 0, 0);
    }
};

template <int R, int CC, int DG, int TYB>
__device__ __forceinline__ void corr_fwd_ring_body(float* __restrict__ ring, int t, int x_origin,
    const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ cv, int C, int H, int W,
    int tiles_x, int tiles_y, float inv_c, int dbg, unsigned long long* stamps) {
    using K = RingCfg<R, CC, DG, TYB>;
    constexpr int DD = K::DD, LW = K::LW, NROW = 2 + 2 * R;
    // stamp segments: 0 set-up + prologue DMA issue | 1 vmcnt wait + barrier | 2 DMA issue | 3 row pipeline | 4 store issue
    STAMP_DECL();

    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int i0 = blockIdx.y * DG;                          // first displacement row of this workgroup
    // lane -> (tx, ty): 64-wide tiles: a row of 32 lanes per 32-lane LDS group; 16-wide tiles (row stride 24 floats): four rows of
    // 8 lanes per group, the even rows of a wave's eight in lanes 0-31 and the odd ones in lanes 32-63 (conflict-free banks,
    // as in the row-streamed backward)
    const int l_ = (int)threadIdx.x;
    const int tx = l_ & (K::TWL - 1);
    const int ty = K::TWL == 32 ? (l_ >> 5) : ((l_ >> 6) * 8 + 2 * ((l_ >> 3) & 3) + ((l_ >> 5) & 1));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: the DMA descriptor choice is scalar
    const int x0 = x_origin + bx * K::TW, y0 = by * K::TYB;
    const int px = x0 + tx * 2, py = y0 + ty;
    const size_t plane = (size_t)H * W;
    const int nchunk = (C + CC - 1) / CC;

    FwdSlots<K, R, CC> slots;
    slots.setup(threadIdx.x, wave, x0, y0, i0, H, W, (unsigned)plane);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f1 + (size_t)b * C * plane), 0, (int)((size_t)C * plane * 4), 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f2 + (size_t)b * C * plane), 0, (int)((size_t)C * plane * 4), 0x00020000);
    auto issue = [&](int stage_idx) {       // stages beyond the last channel fall outside num_records: zeros
        slots.issue(rs1, rs2, ring + (stage_idx & (K::NS - 1)) * K::STAGE, wave, (unsigned)(stage_idx * CC) * (unsigned)plane * 4u);
    };

    FwdAcc<DG, R> acc;
    acc.zero();

    const unsigned rows_addr = (unsigned)(size_t)(lds_cfloat*)(ring + ty * LW + tx * 2);
    const unsigned own_addr = (unsigned)(size_t)(lds_cfloat*)(ring + K::S2P * 4 + ty * K::TW + tx * 2);

#pragma unroll
    for (int st = 0; st < K::NS - 1; ++st) if (!RING_DBG(4)) issue(st);
    STAMP(0);

    for (int k = 0; k < nchunk; ++k) {
        // all but the newest (NS-2) stages have landed -> stage k is complete for this wave ...
        vm_wait<K::ITER * (K::NS - 2)>();
        __builtin_amdgcn_s_barrier();                        // ... and for every wave; slot (k-1) is free
        STAMP(1);
        if (!RING_DBG(4)) issue(k + K::NS - 1);
        STAMP(2);
        const int sbase = (k & (K::NS - 1)) * K::STAGE;
        if (RING_DBG(2)) continue;
        // CC*DG row-steps per stage as one software pipeline (FwdStep): reads of step s+PF are in
        // flight behind the FMAs of step s.
        constexpr int NCOL = NROW / 2, STEPS = CC * DG;
        constexpr int PF = (2 * NCOL <= 15) ? 2 : 1;         // lgkmcnt is a 4-bit counter: <= 15 row reads in flight
        const unsigned abase = rows_addr + (unsigned)sbase * 4u;
        v2f a[CC];
        own_reads<K::SC * 16>(a, own_addr + (unsigned)sbase * 4u, std::make_integer_sequence<int, CC>{});
        static_assert(CC * K::SC * 16 + 64 * 1024 > 0 && CC <= 8, "ds offsets are 16-bit: CC * SC * 16 must stay below 64 KiB");
        v2f row[PF + 1][NCOL];
        using Step0 = FwdStep<0, STEPS, PF, DG, DD, NCOL, K::SC * 16, LW * 4>;
        Step0::template load<0>(row, abase);
        if constexpr (PF > 1) Step0::template load<1>(row, abase);
        Step0::template run<CC>(acc, row, a, abase);
        STAMP(3);
    }
    vm_wait<0>();                                            // drain the tail stages (all zeros)
    STAMP(1);

    if (py >= H || px >= W) return;
    float* out = cv + ((size_t)b * DD * DD) * plane + (size_t)py * W + px;
    if (RING_DBG(1)) {      // timing experiment: keep the accumulators alive, store one plane
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < DG; ++i)
#pragma unroll
            for (int j = 0; j < DD; ++j) sum += acc.get(i, j, 0) + acc.get(i, j, 1);
        out[0] = sum;
        return;
    }
#pragma unroll
    for (int i = 0; i < DG; ++i) {
        if (i0 + i >= DD) break;
#pragma unroll
        for (int j = 0; j < DD; ++j)
            *reinterpret_cast<float2*>(out + (size_t)((i0 + i) * DD + j) * plane) =
                make_float2(acc.get(i, j, 0) * inv_c, acc.get(i, j, 1) * inv_c);
    }
    STAMP(4);
    STAMP_WRITE(stamps, blockIdx.y == 0 ? (int)blockIdx.x * 3 + (wave < 3 ? wave : 2) : 1 << 30, (threadIdx.x & 63) == 0 && wave < 3);
}

template <int R, int CC, int DG>
__global__ __launch_bounds__(256, (RingCfg<R, CC, DG>::WAVES)) void corr_fwd_ring_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ cv, int C, int H, int W,
    int tiles_x, int tiles_y, float inv_c, int dbg, unsigned long long* stamps) {
    using K = RingCfg<R, CC, DG>;
    __shared__ __attribute__((aligned(16))) float ring[K::NS * K::STAGE];
    corr_fwd_ring_body<R, CC, DG, 8>(ring, xcd_remap(blockIdx.x, gridDim.x), 0, f1, f2, cv, C, H, W, tiles_x, tiles_y, inv_c, dbg, stamps);
}

// 64x8 tiles over the full 64-pixel columns and 16x32 tiles over a last column of <= 16 pixels, in one launch (the forward twin of
// corr_bwd_rs_mixed_kernel: 208 = 3 x 64 + 16 leaves 19 % of the lanes of 64-wide tiles idle)
template <int R, int CC, int DG>
__global__ __launch_bounds__(256, (RingCfg<R, CC, DG>::WAVES)) void corr_fwd_ring_mixed_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ cv, int C, int H, int W,
    int wide_tiles_x, int wide_tiles_y, int narrow_tiles_y, int n_narrow, int stride, float inv_c) {
    using KW = RingCfg<R, CC, DG, 8>;
    using KN = RingCfg<R, CC, DG, 32>;
    __shared__ __attribute__((aligned(16))) float ring[KW::NS * (KW::STAGE > KN::STAGE ? KW::STAGE : KN::STAGE)];
    const int u = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = min(n_narrow, (u + 1) / stride);
    const bool narrow = (u % stride == stride - 1) && (u / stride < n_narrow);      // (workgroup-uniform)
    if (narrow)
        corr_fwd_ring_body<R, CC, DG, 32>(ring, nb - 1, wide_tiles_x * 64, f1, f2, cv, C, H, W, 1, narrow_tiles_y, inv_c, 0, nullptr);
    else
        corr_fwd_ring_body<R, CC, DG, 8>(ring, u - nb, 0, f1, f2, cv, C, H, W, wide_tiles_x, wide_tiles_y, inv_c, 0, nullptr);
}

template <int R, int CC, int DG>
int launch_fwd_ring_mixed(const float* f1, const float* f2, float* cv, int B, int C, int H, int W, hipStream_t s) {
    static_assert(RingCfg<R, CC, DG>::NG == 1, "all displacement rows in one workgroup");
    const int wide_x = W / 64, wide_y = ceil_div(H, 8), narrow_y = ceil_div(H, 32);
    const int n_wide = wide_x * wide_y * B, n_narrow = narrow_y * B;
    const int total = n_wide + n_narrow, stride = total / n_narrow;
    UNFLOW_LAUNCH((corr_fwd_ring_mixed_kernel<R, CC, DG>), dim3(total), dim3(256), 0, s, f1, f2, cv, C, H, W,
                       wide_x, wide_y, narrow_y, n_narrow, stride, 1.0f / C);
    return unflow_launch_status();
}

template <int R, int CC, int DG>
int launch_fwd_ring(const float* f1, const float* f2, float* cv, int B, int C, int H, int W, hipStream_t s) {
    using K = RingCfg<R, CC, DG>;
    const int tx = ceil_div(W, K::TW), ty = ceil_div(H, K::TYB);
#ifdef UNFLOW_TUNING
    const char* e = getenv("UNFLOW_CORR_DEBUG");               // timing experiments only
    const int dbg = e ? atoi(e) : 0;
#else
    const int dbg = 0;
#endif
    UNFLOW_LAUNCH((corr_fwd_ring_kernel<R, CC, DG>), dim3(tx * ty * B, K::NG), dim3(256), 0, s, f1, f2, cv, C, H, W,
                       tx, ty, 1.0f / C, dbg, stamp_buffer());
    return unflow_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Mid-size maps (pyramid levels 3, 4), forward: the ring kernel above with the CHANNELS split over PH phases inside
// one workgroup.  A 64x8 tile times 3 displacement rows is a workgroup of the plain ring kernel; level 4 has only 96
// of them, each grinding through all channels on four waves (one wave per SIMD issues an FMA every 4 cycles): 31 us
// for 4 MB.  Here a workgroup has PH x 4 waves; phase p streams its own C/PH channels through its own LDS ring and
// accumulates its own partial cost volume; the partials meet in phase 0 through LDS (fixed order, no atomics).
// Requires W % 4 == 0.
// ---------------------------------------------------------------------------------------------
template <int R, int CC, int DG, int PH>
struct RingPCfg {
    static constexpr int DD = 2 * R + 1;
    static constexpr int NG = (DD + DG - 1) / DG;
    static constexpr int TW = 64, TYB = 8, NS = 3;
    static constexpr int LW = TW + 2 * R, LH = TYB + DG - 1;
    static constexpr int S2 = LH * LW / 4, S2P = (S2 + 63) / 64 * 64, S1 = TYB * TW / 4, SC = S2P + S1;      // as RingCfg
    static constexpr int ITER = (CC * SC + 255) / 256;
    static constexpr int STAGE = ITER * 256 * 4;
    static constexpr int RING = NS * STAGE;                                     // floats per phase
    static constexpr int THREADS = 256 * PH;
    static_assert(DG * DD * 2 * 256 <= PH * RING, "the partial sums of one phase must fit the (then idle) rings");
};

template <int R, int CC, int DG, int PH>
__global__ __launch_bounds__((RingPCfg<R, CC, DG, PH>::THREADS)) void corr_fwd_ringp_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ cv, int C, int H, int W,
    int tiles_x, int tiles_y, float inv_c) {
    using K = RingPCfg<R, CC, DG, PH>;
    constexpr int DD = K::DD, LW = K::LW, NROW = 2 + 2 * R;
    __shared__ __attribute__((aligned(16))) float lds[PH * K::RING];

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int i0 = blockIdx.y * DG;
    const int phase = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);      // wave-uniform
    const int tid = threadIdx.x & 255;
    const int tx = tid & 31, ty = tid >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = bx * K::TW, y0 = by * K::TYB;
    const int px = x0 + tx * 2, py = y0 + ty;
    const size_t plane = (size_t)H * W;
    // this phase's channels [cb, cb + cn): whole stages per phase, the same number of stages in every phase
    const int cpp = ceil_div(ceil_div(C, PH), CC) * CC;
    const int nchunk = cpp / CC;
    const int cb = phase * cpp;
    const int cn = max(0, min(cpp, C - cb));
    float* ring = lds + phase * K::RING;

    FwdSlots<K, R, CC> slots;
    slots.setup(tid, wave, x0, y0, i0, H, W, (unsigned)plane);
    // this phase's channels only: the rest of the sample lies beyond num_records and arrives as zeros
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f1 + ((size_t)b * C + cb) * plane), 0, (int)((size_t)cn * plane * 4), 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f2 + ((size_t)b * C + cb) * plane), 0, (int)((size_t)cn * plane * 4), 0x00020000);
    auto issue = [&](int stage_idx) {
        slots.issue(rs1, rs2, ring + (stage_idx % K::NS) * K::STAGE, wave, (unsigned)(stage_idx * CC) * (unsigned)plane * 4u);
    };

    FwdAcc<DG, R> acc;
    acc.zero();

    const unsigned rows_addr = (unsigned)(size_t)(lds_cfloat*)(ring + ty * LW + tx * 2);
    const unsigned own_addr = (unsigned)(size_t)(lds_cfloat*)(ring + K::S2P * 4 + ty * K::TW + tx * 2);

#pragma unroll
    for (int st = 0; st < K::NS - 1; ++st) issue(st);

    for (int k = 0; k < nchunk; ++k) {
        vm_wait<K::ITER * (K::NS - 2)>();
        __builtin_amdgcn_s_barrier();
        issue(k + K::NS - 1);
        const unsigned sb = (unsigned)((k % K::NS) * K::STAGE) * 4u;
        constexpr int NCOL = NROW / 2, STEPS = CC * DG;
        constexpr int PF = (2 * NCOL <= 15) ? 2 : 1;
        v2f a[CC];
        own_reads<K::SC * 16>(a, own_addr + sb, std::make_integer_sequence<int, CC>{});
        v2f row[PF + 1][NCOL];
        using Step0 = FwdStep<0, STEPS, PF, DG, DD, NCOL, K::SC * 16, LW * 4>;
        Step0::template load<0>(row, rows_addr + sb);
        if constexpr (PF > 1) Step0::template load<1>(row, rows_addr + sb);
        Step0::template run<CC>(acc, row, a, rows_addr + sb);
    }
    vm_wait<0>();                                            // the all-zero tail stages: no LDS-DMA in flight beyond here
    UNFLOW_WAIT_LGKMCNT0();
    __syncthreads();

    // partial cost volumes of phases 1 .. PH-1 -> phase 0, one phase at a time through the idle rings
    float* red = lds;
#pragma unroll 1
    for (int p = 1; p < PH; ++p) {
        if (phase == p) {
#pragma unroll
            for (int i = 0; i < DG; ++i)
#pragma unroll
                for (int j = 0; j < DD; ++j)
                    *reinterpret_cast<float2*>(red + ((i * DD + j) * 256 + tid) * 2) = make_float2(acc.get(i, j, 0), acc.get(i, j, 1));
        }
        __syncthreads();
        if (phase == 0) {
#pragma unroll
            for (int i = 0; i < DG; ++i)
#pragma unroll
                for (int j = 0; j < DD; ++j) {
                    const float2 v = *reinterpret_cast<const float2*>(red + ((i * DD + j) * 256 + tid) * 2);
                    acc.add(i, j, 0, v.x); acc.add(i, j, 1, v.y);
                }
        }
        __syncthreads();
    }
    if (phase != 0 || py >= H || px >= W) return;
    float* out = cv + ((size_t)b * DD * DD) * plane + (size_t)py * W + px;
#pragma unroll
    for (int i = 0; i < DG; ++i) {
        if (i0 + i >= DD) break;
#pragma unroll
        for (int j = 0; j < DD; ++j)
            *reinterpret_cast<float2*>(out + (size_t)((i0 + i) * DD + j) * plane) =
                make_float2(acc.get(i, j, 0) * inv_c, acc.get(i, j, 1) * inv_c);
    }
}

template <int R, int CC, int DG, int PH>
int launch_fwd_ringp(const float* f1, const float* f2, float* cv, int B, int C, int H, int W, hipStream_t s) {
    using K = RingPCfg<R, CC, DG, PH>;
    const int tx = ceil_div(W, K::TW), ty = ceil_div(H, K::TYB);
    UNFLOW_LAUNCH((corr_fwd_ringp_kernel<R, CC, DG, PH>), dim3(tx * ty * B, K::NG), dim3(K::THREADS), 0, s,
                       f1, f2, cv, C, H, W, tx, ty, 1.0f / C);
    return unflow_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Large-map backward, "group split" ring kernel.  blockIdx.y selects the gradient:
//   0: gf1[c][q] = (1/C) sum_ij g[ij][q]             * f2[c][q + (i-R, j-R)]
//   1: gf2[c][q] = (1/C) sum_ij g[ij][q - (i-R,j-R)] * f1[c][q - (i-R, j-R)]   (a gather: no atomics)
// Holding all DD*DD*2 upstream-gradient values of a lane's two pixels costs 162 VGPRs and leaves no
// room for a software pipeline (2 waves/SIMD, spills).  Here a 384-thread workgroup owns a 64x4
// pixel tile THREE times: wave pair g keeps only displacement rows [3g, 3g+3) (54 VGPRs) and
// produces a partial dot product per channel; pairs 1 and 2 hand theirs to pair 0 through a
// double-buffered LDS slab one stage later (no extra barrier), pair 0 adds them in a fixed order
// and issues the 8-byte stores.  ~100 VGPRs -> 4-5 waves per SIMD, so LDS latency, DMA latency and
// stores overlap across waves.  F streams through the same global_load_lds ring as the forward kernel.
// R = 8: six wave pairs (768 threads), the last one with two live rows.  Requires W % 4 == 0.
// ---------------------------------------------------------------------------------------------
template <int R, int CC, int TYB_ = 4>
struct BwdGsCfg {
    static constexpr int DD = 2 * R + 1, DG = 3, NGRP = (DD + DG - 1) / DG;    // R=4: 3 wave groups; R=8: 6 (the last one has 2 live rows)
    static constexpr int TW = 64, TYB = TYB_, GL = 32 * TYB;                   // GL lanes (2 px each) per wave group
    static constexpr int NS = 3, THREADS = GL * NGRP;                          // 3 slots measured best: 6 slots halve the residency
    static constexpr int LW = TW + 2 * R, LH = TYB + DG * NGRP - 1;            // = TYB + 2R when DD % 3 == 0; else one spare row
    static constexpr int SC = LH * LW / 4;                                      // float4 slots per channel
    static constexpr int ITER = (CC * SC + THREADS - 1) / THREADS;
    static constexpr int STAGE = ITER * THREADS * 4;                            // floats per ring slot
    static constexpr int RED = (NGRP - 1) * CC * GL * 2;                        // floats per hand-off buffer (groups 1 .. NGRP-1)
};

// Hand-off slab of the group-split backward: entry e = (group - 1) * CC + c, GL lanes x 8 bytes each.
template <int GL, int N, int... Es>
__device__ __forceinline__ void slab_read(v2f (&q)[N], unsigned addr, std::integer_sequence<int, Es...>) {
    ((q[Es] = lds_read_b64<Es * GL * 8>(addr)), ...);
}
template <int GL, int CC, int... Cs>
__device__ __forceinline__ void slab_write(unsigned addr, const v2f (&v)[CC], std::integer_sequence<int, Cs...>) {
    (lds_write_b64<Cs * GL * 8>(addr, v[Cs]), ...);
}

// A lane's share of the upstream gradient, laid out for packed FMAs.  A lane owns pixels x (even) and x + 1 and reads a
// halo row as R + 1 aligned float pairs row[k] = (f[x + 2k], f[x + 2k + 1]).  Pixel x needs f[x + j], pixel x + 1 needs
// f[x + 1 + j]: both walk the SAME aligned pairs if pixel x pairs its displacements (0,1)(2,3).. and pixel x + 1 pairs
// (1,2)(3,4)..; the two leftovers (j = 2R of pixel x, j = 0 of pixel x + 1) are single FMAs.  So a row-step is 2R
// v_pk_fma_f32 + 2 v_fma_f32 instead of 2 (2R + 1) v_fma_f32, with no register moves: the pairs are pairs as loaded.
// (v_pk_fma_f32 is what the fp32 VALU peak is quoted on; a plain v_fma_f32 is half of it.  Letting hipcc's SLP
// vectoriser pack the old per-pixel FMAs cost a v_mov per pair and lost.)
template <int R>
struct GsWeights {
    v2f p0[3][R];      // pixel x:     (g[2k], g[2k+1])      x row[k],     k = 0 .. R-1
    float s0[3];       //              g[2R]                 x row[R].x
    v2f p1[3][R];      // pixel x + 1: (g[2k-1], g[2k])      x row[k],     k = 1 .. R   (stored at k - 1)
    float s1[3];       //              g[0]                  x row[0].y
    __device__ __forceinline__ void set(int ii, int j, int p, float v) {      // constant indices after unrolling
        if (p == 0) {
            if (j == 2 * R) s0[ii] = v;
            else if (j & 1) p0[ii][j / 2].y = v;
            else p0[ii][j / 2].x = v;
        } else {
            if (j == 0) s1[ii] = v;
            else if ((j + 1) & 1) p1[ii][(j + 1) / 2 - 1].y = v;
            else p1[ii][(j + 1) / 2 - 1].x = v;
        }
    }
    __device__ __forceinline__ void pin() {          // consumers stay where this is called (see the kernel)
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
#pragma unroll
            for (int k = 0; k < R; ++k) { UNFLOW_PIN_VGPR(p0[ii][k]); UNFLOW_PIN_VGPR(p1[ii][k]); }
            UNFLOW_PIN_VGPR(s0[ii]);
            UNFLOW_PIN_VGPR(s1[ii]);
        }
    }
};

// One row-step of the backward pipeline: ST = c * 3 + i.
template <int ST, int STEPS, int PF, int R, int NCOL, int CH_BYTES, int ROW_BYTES>
struct GsStep {
    template <int Q, int... Ks>
    static __device__ __forceinline__ void load_cols(v2f (&row)[PF + 1][NCOL], unsigned addr,
                                                     std::integer_sequence<int, Ks...>) {
        constexpr int off = (Q / 3) * CH_BYTES + (Q % 3) * ROW_BYTES;
        ((row[Q % (PF + 1)][Ks] = lds_read_b64<off + 8 * Ks>(addr)), ...);
    }
    template <int Q>
    static __device__ __forceinline__ void load(v2f (&row)[PF + 1][NCOL], unsigned addr) {
        if constexpr (Q < STEPS) load_cols<Q>(row, addr, std::make_integer_sequence<int, NCOL>{});
    }
    template <int CC>
    static __device__ __forceinline__ void run(const GsWeights<R>& w, v2f (&acc)[CC][2],
                                               v2f (&row)[PF + 1][NCOL], unsigned addr) {
        static_assert(NCOL == R + 1, "a halo row is R + 1 float pairs");
        if constexpr (ST < STEPS) {
            load<ST + PF>(row, addr);
            constexpr int newer = (STEPS - 1 - ST < PF ? STEPS - 1 - ST : PF) * NCOL;
            lds_wait<newer>();
            constexpr int c = ST / 3, i = ST % 3, rb = ST % (PF + 1);
#pragma unroll
            for (int k = 0; k < R; ++k) {
#ifdef UNFLOW_NO_PK_BWD      // experiment (tools/): the same accumulators with scalar FMAs
                acc[c][0].x = fmaf(w.p0[i][k].x, row[rb][k].x, acc[c][0].x); acc[c][0].y = fmaf(w.p0[i][k].y, row[rb][k].y, acc[c][0].y);
                acc[c][1].x = fmaf(w.p1[i][k].x, row[rb][k + 1].x, acc[c][1].x); acc[c][1].y = fmaf(w.p1[i][k].y, row[rb][k + 1].y, acc[c][1].y);
#else
                acc[c][0] = __builtin_elementwise_fma(w.p0[i][k], row[rb][k], acc[c][0]);
                acc[c][1] = __builtin_elementwise_fma(w.p1[i][k], row[rb][k + 1], acc[c][1]);
#endif
            }
            acc[c][0].x = fmaf(w.s0[i], row[rb][R].x, acc[c][0].x);
            acc[c][1].y = fmaf(w.s1[i], row[rb][0].y, acc[c][1].y);
            __builtin_amdgcn_sched_barrier(0);
            GsStep<ST + 1, STEPS, PF, R, NCOL, CH_BYTES, ROW_BYTES>::template run<CC>(w, acc, row, addr);
        }
    }
};

template <int R, int CC, int TYB>
__global__ __launch_bounds__((BwdGsCfg<R, CC, TYB>::THREADS)) void corr_bwd_gs_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         const float* __restrict__ g, float* __restrict__ gf1,
                                                         float* __restrict__ gf2, int Ctot, int cpg, int H, int W,
                                                         int tiles_x, int tiles_y, float inv_c, int dbg, unsigned long long* stamps) {
    using K = BwdGsCfg<R, CC, TYB>;
    // stamp segments: 0 gather + prologue | 1 vmcnt wait + barrier | 2 finish (group 0: slab read, add, store) | 3 DMA issue |
    //                 4 row pipeline (FMAs) + partial hand-off | 5 tail
    STAMP_DECL();
    constexpr int DD = K::DD, LW = K::LW, NROW = 2 + 2 * R, NCOL = NROW / 2, GL = K::GL;
    static_assert(K::NS >= 3 && K::NS <= 6, "ring depth (the counted vmcnt waits cover NS-2 <= 4 store groups)");
    __shared__ __attribute__((aligned(16))) float lds[K::NS * K::STAGE + 2 * K::RED];
    float* ring = lds;
    float* red = lds + K::NS * K::STAGE;

    // work item = (tile, gradient): the two gradients of a tile are neighbours in the XCD-local order, so the
    // upstream-gradient planes both of them read are fetched from HBM once and served from that XCD's L2 the second time
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int mode = t & 1; t >>= 1;
    const float* __restrict__ F = mode ? f1 : f2;
    float* __restrict__ out = mode ? gf2 : gf1;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    // blockIdx.y: channel group [c_begin, c_begin + C) of the Ctot channels (small maps: more, shorter workgroups)
    const int c_begin = blockIdx.y * cpg;
    const int C = min(cpg, Ctot - c_begin);
    const int l = threadIdx.x % GL, wave = threadIdx.x >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / GL);     // wave-uniform: branches on it are scalar
    const int tx = l & 31, ty = l >> 5;
    const int x0 = bx * K::TW, y0 = by * K::TYB;
    const int px = x0 + tx * 2, py = y0 + ty;
    const size_t plane = (size_t)H * W;
    const int nchunk = (C + CC - 1) / CC;
    const bool live = (py < H && px < W);

    // this wave pair's 3 displacement rows of the upstream gradient -> registers
    // (mode 1: displacement-flipped and gathered from q + (i'-R, j'-R))
    // Both streams of this kernel go through buffer descriptors: the hardware range check returns 0 for an offset past
    // num_records, so "outside the image" is an offset with kOut added instead of a branch or a select on a 64-bit
    // address.  (The first version selected between the real address and a zero line per value: hipcc re-derived the
    // zero line's address through the GOT each time -- s_getpc + s_load + lgkmcnt(0) -- and wrapped every load in an
    // EXEC-masked branch; ~30 instructions per gathered value, and the gather was 30-45 % of a workgroup's lifetime.)
    // Requires 81 * plane * 4 and Ctot * plane * 4 < kOut (checked by the launcher).
    constexpr unsigned kOut = 0x40000000u;

    const float* baseF = F + ((size_t)b * Ctot + c_begin) * plane;
    const auto frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(baseF), 0, (int)((size_t)C * plane * 4), 0x00020000);
    // per-lane DMA slot descriptor, computed once: byte offset of the slot's 16 bytes within the first stage's CC planes,
    // or kOut outside the image / in the padding.  A stage adds its (uniform) channel offset; channels past C fall
    // outside num_records and arrive as zeros.
    unsigned slotB[K::ITER];
#pragma unroll
    for (int it = 0; it < K::ITER; ++it) {
        const int s = it * K::THREADS + (int)threadIdx.x;
        const int c = s / K::SC;
        const int r = s - c * K::SC;
        const int ly = r / (LW / 4);
        const int gy = y0 - R + ly, gx = x0 - R + (r - ly * (LW / 4)) * 4;
        const bool in = (c < CC) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        slotB[it] = in ? ((unsigned)c * (unsigned)plane + (unsigned)(gy * W + gx)) * 4u : kOut;
        UNFLOW_PIN_VGPR(slotB[it]);      // materialise once; do not re-derive in the loop
    }
    auto issue = [&](int stage_idx) {
        float* dst = ring + (stage_idx % K::NS) * K::STAGE;
        const unsigned stageB = (unsigned)(stage_idx * CC) * (unsigned)plane * 4u;
#pragma unroll
        for (int it = 0; it < K::ITER; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(frs, (lds_ptr)(dst + (it * K::THREADS + wave * 64) * 4), 16,
                                                 (int)(slotB[it] + stageB), 0, 0, 0);
    };

    // this wave pair's 3 displacement rows of the upstream gradient -> registers
    // (mode 1: displacement-flipped and gathered from q + (i'-R, j'-R))
    GsWeights<R> wr;
    {
        const float* gb = g + (size_t)b * DD * DD * plane;
        const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gb), 0, (int)(DD * DD * plane * 4), 0x00020000);
        unsigned colB[DD][2];
#pragma unroll
        for (int j = 0; j < DD; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int sx = px + p + mode * (j - R);
                colB[j][p] = ((unsigned)sx < (unsigned)W) ? (unsigned)sx * 4u : kOut;
            }
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
            const int gi = grp * 3 + ii;                  // rows past DD (last pair when DD % 3 != 0) read as zero
            const int i = min(gi, DD - 1);
            const int sy = py + mode * (i - R);
            const unsigned rowB = (gi < DD && (unsigned)sy < (unsigned)H) ? (unsigned)(sy * W) * 4u : kOut;
#pragma unroll
            for (int j = 0; j < DD; ++j) {
                const int e = i * DD + j;
                const int pl = e + mode * ((DD * DD - 1) - 2 * e);            // gf2: plane (2R-i, 2R-j)
                const unsigned planeB = (unsigned)pl * (unsigned)plane * 4u;  // wave-uniform
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (RING_DBG(8)) { wr.set(ii, j, p, (float)(ii + j + p) * inv_c); continue; }      // ablation (tuning builds): no gather
                    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(grs, (int)(rowB + colB[j][p] + planeB), 0, 0);
                    wr.set(ii, j, p, __uint_as_float(v) * inv_c);
                }
            }
        }
    }

    // rows of this pair start 3*grp below the tile's first halo row
    const unsigned rows_addr = (unsigned)(size_t)(lds_cfloat*)(ring + (ty + 3 * grp) * LW + tx * 2);
    float* op = out + (((size_t)b * Ctot + c_begin) * H + py) * W + px;

    // pair 0: add the partials pairs 1 and 2 left for stage `st` and store its CC channels
    // (the slab is read and written with hand-issued ds ops: compiler-visible LDS accesses next to an
    // in-flight LDS-DMA make hipcc drain vmcnt(0))
    const unsigned red_addr = (unsigned)(size_t)(lds_cfloat*)(red + l * 2);
    auto finish = [&](int st, const float (&mine)[CC][2]) {
        const unsigned ra = red_addr + (unsigned)((st & 1) * K::RED) * 4u;
        v2f q[(K::NGRP - 1) * CC];                       // partial (pair g+1, channel c) at q[g * CC + c]
        slab_read<GL>(q, ra, std::make_integer_sequence<int, (K::NGRP - 1) * CC>{});
        lds_wait<0>();
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            float sx = mine[c][0], sy = mine[c][1];
#pragma unroll
            for (int gq = 0; gq < K::NGRP - 1; ++gq) { sx += q[gq * CC + c].x; sy += q[gq * CC + c].y; }   // fixed order
            const int gc = st * CC + c;
            // exactly one store instruction per channel per wave (lanes masked by EXEC): the vmcnt counts below rely on it
            if (live && gc < C)
                *reinterpret_cast<float2*>(op + (size_t)gc * plane) = make_float2(sx, sy);
        }
    };

    vm_wait<0>();                                    // the gradient loads are done before the first DMA goes out
    wr.pin();                                        // pin their consumers here too (else hipcc sinks them into the
                                                     // loop and drains vmcnt(0) with the DMA in flight)
    // (measured: issuing the first ring stages BEFORE the gather, to hide their latency behind it, costs 4 us at level 2 --
    // the gather's loads queue behind 36 KB of DMA -- and gains < 1 us at levels 3 / 4)
#pragma unroll
    for (int st = 0; st < K::NS - 1; ++st) issue(st);
    STAMP(0);

    float keep[CC][2];                               // pair 0: its own partials of the previous stage
#pragma unroll
    for (int c = 0; c < CC; ++c) { keep[c][0] = 0.f; keep[c][1] = 0.f; }

    for (int k = 0; k < nchunk; ++k) {
        // vmcnt is in-order over DMA loads AND stores.  Newer than stage k's DMA when we get here:
        // NS-2 later DMA stages, plus (pair 0 only) the stores of the last min(NS-2, k-1) finished stages.
        {
            constexpr int D = (K::NS - 2) * K::ITER;
            const int nst = (grp != 0) ? 0 : min(K::NS - 2, max(0, k - 1));     // wave-uniform
            if (nst == 0) vm_wait<D>();
            else if (nst == 1) vm_wait<D + CC>();
            else if (nst == 2) vm_wait<D + 2 * CC>();
            else if (nst == 3) vm_wait<D + 3 * CC>();
            else vm_wait<D + 4 * CC>();
        }
        __builtin_amdgcn_s_barrier();                // stage k landed for all; slot k-1 and red[(k-1)&1] are complete
        STAMP(1);
        if (grp == 0 && k > 0) finish(k - 1, keep);
        STAMP(2);
        issue(k + K::NS - 1);
        STAMP(3);

        // row reads in flight ahead of their FMAs: 2 row-steps at R=4, 1 at R=8 (lgkmcnt allows <= 15 reads, and 768
        // threads must fit 170 VGPRs each)
        constexpr int PF = (2 * NCOL <= 15) ? 2 : 1, STEPS = CC * 3;
        const unsigned abase = rows_addr + (unsigned)((k % K::NS) * K::STAGE) * 4u;
        v2f acc[CC][2];                              // [channel][pixel]: two partial sums each (the halves of the packed FMAs)
#pragma unroll
        for (int c = 0; c < CC; ++c) { acc[c][0] = v2f{0.f, 0.f}; acc[c][1] = v2f{0.f, 0.f}; }
        v2f row[PF + 1][NCOL];
        using Step0 = GsStep<0, STEPS, PF, R, NCOL, K::SC * 16, LW * 4>;
        if constexpr (PF > 0) Step0::template load<0>(row, abase);
        if constexpr (PF > 1) Step0::template load<1>(row, abase);
        Step0::template run<CC>(wr, acc, row, abase);

        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < CC; ++c) { keep[c][0] = acc[c][0].x + acc[c][0].y; keep[c][1] = acc[c][1].x + acc[c][1].y; }
        } else {
            const unsigned wa = red_addr + (unsigned)((k & 1) * K::RED + (grp - 1) * CC * GL * 2) * 4u;
            v2f pv[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                pv[c].x = acc[c][0].x + acc[c][0].y;
                pv[c].y = acc[c][1].x + acc[c][1].y;
            }
            slab_write<GL, CC>(wa, pv, std::make_integer_sequence<int, CC>{});
            UNFLOW_WAIT_LGKMCNT0();      // slab written before the next barrier
        }
        STAMP(4);
    }
    vm_wait<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 0) finish(nchunk - 1, keep);
    STAMP(5);
    STAMP_WRITE(stamps, blockIdx.y == 0 ? (int)blockIdx.x * K::NGRP + grp : 1 << 30, (threadIdx.x % GL) == 0);
}

// The group-split backward addresses one sample's gradient planes and one sample's feature planes through 32-bit
// buffer offsets, with bit 30 marking "outside": both extents must stay below 1 GiB.
static inline bool gs_offsets_fit(int C, int H, int W, int R) {
    const size_t plane = (size_t)H * W * 4, dd = (size_t)(2 * R + 1) * (2 * R + 1);
    return dd * plane < (1u << 30) && (size_t)C * plane < (1u << 30);
}

template <int R, int CC, int TYB>
int launch_bwd_gs(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
                  int B, int C, int H, int W, int groups, hipStream_t s) {
    using K = BwdGsCfg<R, CC, TYB>;
    const int tx = ceil_div(W, K::TW), ty = ceil_div(H, K::TYB);
    groups = groups < 1 ? 1 : groups;
    const int cpg = ceil_div(ceil_div(C, groups), CC) * CC;      // whole ring stages per channel group
#ifdef UNFLOW_TUNING
    const char* e = getenv("UNFLOW_CORR_DEBUG");
    const int dbg = e ? atoi(e) : 0;
#else
    const int dbg = 0;
#endif
    unsigned long long* stamps = stamp_buffer();
    UNFLOW_LAUNCH((corr_bwd_gs_kernel<R, CC, TYB>), dim3(tx * ty * B * 2, ceil_div(C, cpg)), dim3(K::THREADS), 0, s,
                       f1, f2, g, gf1, gf2, C, cpg, H, W, tx, ty, 1.0f / C, dbg, stamps);
    return unflow_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Large-map backward, "row streamed" kernel (round 3).  The group-split kernel above keeps a third of a lane's upstream
// gradient in registers for ALL channels and pays for it with a serial gather in front of every tile (22-36 % of a
// workgroup's life with one 768-thread workgroup per CU), a barrier per ring stage and a hand-off of partial sums.
// Here the roles are turned around: the CHANNELS of a work item are resident -- the F halo tile of CH channels in LDS
// (staged once by LDS-DMA), one accumulator pair per channel in registers -- and the upstream gradient STREAMS through
// the registers one displacement row at a time (DD values x 2 pixels per lane, the next row requested while the current
// one is being used).  A lane computes all DD x DD taps of its two pixels itself: no partial sums, no hand-off, and after
// the one barrier behind the staging the four waves of a workgroup never meet again, so the gather of one wave, the LDS
// reads of another and the FMAs of a third overlap, within a workgroup and across the two (R = 4, CH = 16: 72 KB) that
// share a CU.  A work item = (tile, gradient, group of CH channels); the items that read the same gradient planes are
// neighbours in the XCD-local order (HBM delivers the planes once, the re-reads come out of that XCD's L2).
//   gf1[c][q] = (1/C) sum_ij g[ij][q]                         * f2[c][q + (i-R, j-R)]
//   gf2[c][q] = (1/C) sum_ij g[(2R-i, 2R-j)][q + (i-R, j-R)]  * f1[c][q + (i-R, j-R)]
// Both modes multiply the weight of tap (i, j) with F at q + (i-R, j-R), which the halo tile holds as an exact 0 outside the
// image; in mode 1 the weight is read at that same position, so a weight fetched for a position outside the image (some
// other finite gradient value, or 0 from the buffer range check) meets a 0 and needs no validity test of its own.
// Requires W % 4 == 0 and 16-byte aligned feature tensors (LDS-DMA pieces), 32-bit buffer offsets (gs_offsets_fit).
// ---------------------------------------------------------------------------------------------
// Tile shapes (TYB rows of 256 / TYB lanes, 2 px per lane): 64 x 8, or 16 x 32 for widths like 13 * 2^k (208 = 13 * 16:
// 64-wide tiles leave the fourth tile column three quarters empty, 19 % of all lanes idle).
template <int R, int CH, int TYB_ = 8>
struct BwdRsCfg {
    static constexpr int DD = 2 * R + 1, TYB = TYB_, THREADS = 256, TWL = THREADS / TYB, TW = 2 * TWL;
    static_assert(TWL == 32 || TWL == 8, "lane mappings below");
    static constexpr int LW = TW + 2 * R, LH = TYB + 2 * R;
    static constexpr int SC = LH * LW / 4;                                      // float4 slots per channel
    static constexpr int ITER = (CH * SC + THREADS - 1) / THREADS;
    static constexpr int TILE = ITER * THREADS * 4;                             // floats
    static constexpr int HALF = (CH + 1) / 2;                                   // channels per ds_read base address (16-bit offsets)
};

template <int R>
struct RowWeights {        // one displacement row of GsWeights
    v2f p0[R]; float s0; v2f p1[R]; float s1;
    __device__ __forceinline__ void set(int j, float vx, float vy) {           // (vx, vy) = tap j of pixels x, x + 1
        if (j == 2 * R) s0 = vx; else if (j & 1) p0[j / 2].y = vx; else p0[j / 2].x = vx;
        if (j == 0) s1 = vy; else if ((j + 1) & 1) p1[(j + 1) / 2 - 1].y = vy; else p1[(j + 1) / 2 - 1].x = vy;
    }
};

// One row-step of the row-streamed backward: ST = channel of the group.
template <int ST, int STEPS, int PF, int R, int NCOL, int CH_BYTES, int HALF>
struct RsStep {
    template <int Q, int... Ks>
    static __device__ __forceinline__ void load_cols(v2f (&row)[PF + 1][NCOL], unsigned a_lo, unsigned a_hi,
                                                     std::integer_sequence<int, Ks...>) {
        constexpr int off = (Q % HALF) * CH_BYTES;
        ((row[Q % (PF + 1)][Ks] = lds_read_b64<off + 8 * Ks>(Q < HALF ? a_lo : a_hi)), ...);
    }
    template <int Q>
    static __device__ __forceinline__ void load(v2f (&row)[PF + 1][NCOL], unsigned a_lo, unsigned a_hi) {
        if constexpr (Q < STEPS) load_cols<Q>(row, a_lo, a_hi, std::make_integer_sequence<int, NCOL>{});
    }
    static __device__ __forceinline__ void run(const RowWeights<R>& w, v2f (&acc)[STEPS][2], v2f (&row)[PF + 1][NCOL],
                                               unsigned a_lo, unsigned a_hi) {
        static_assert(NCOL == R + 1, "a halo row is R + 1 float pairs");
        if constexpr (ST < STEPS) {
            load<ST + PF>(row, a_lo, a_hi);
            constexpr int newer = (STEPS - 1 - ST < PF ? STEPS - 1 - ST : PF) * NCOL;
            lds_wait<newer>();
            constexpr int rb = ST % (PF + 1);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                acc[ST][0] = __builtin_elementwise_fma(w.p0[k], row[rb][k], acc[ST][0]);
                acc[ST][1] = __builtin_elementwise_fma(w.p1[k], row[rb][k + 1], acc[ST][1]);
            }
            acc[ST][0].x = fmaf(w.s0, row[rb][R].x, acc[ST][0].x);
            acc[ST][1].y = fmaf(w.s1, row[rb][0].y, acc[ST][1].y);
            __builtin_amdgcn_sched_barrier(0);
            RsStep<ST + 1, STEPS, PF, R, NCOL, CH_BYTES, HALF>::run(w, acc, row, a_lo, a_hi);
        }
    }
};

typedef unsigned v2u __attribute__((ext_vector_type(2)));

// (Variants measured and removed in round 5 -- wave sets sharing a tile's channels or its displacement rows, 64 x 16 tiles, register
// staging, the two-half phase-shifted persistent form: none beat this one; profiles/r3_experiments.md has their numbers.)
template <int R, int CH, int TYB, int AHEAD>
__device__ __forceinline__ void corr_bwd_rs_body(float* __restrict__ tile, int t, int x_origin,
                                                 const float* __restrict__ f1, const float* __restrict__ f2,
                                                 const float* __restrict__ g, float* __restrict__ gf1,
                                                 float* __restrict__ gf2, int Ctot, int H, int W,
                                                 int tiles_x, int tiles_y, int ngrp, float inv_c) {
    using K = BwdRsCfg<R, CH, TYB>;
    constexpr int DD = K::DD, LW = K::LW, NCOL = R + 1, NT = K::THREADS;
    constexpr int ITER = (CH * K::SC + NT - 1) / NT;

    const int cg = t % ngrp; t /= ngrp;              // the channel groups of a (tile, gradient) share its gradient planes
    const int mode = t & 1; t >>= 1;                 // ... and so do the two gradients of a tile
    const float* __restrict__ F = mode ? f1 : f2;
    float* __restrict__ out = mode ? gf2 : gf1;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int c_begin = cg * CH;
    const int C = min(CH, Ctot - c_begin);
    // (`l` and `ws` are written as round 4 wrote them -- ws is 0: one wave set -- because with exactly these expressions hipcc emits round 4's
    // instruction stream for all four instantiations, byte for byte: the kernels that passed GPUTEST_r04 are the kernels that ship
    // (tests/test_abi.py::test_fp32_cost_volume_kernels_are_the_validated_ones, tools/isa_hashes.py))
    const int l = threadIdx.x % K::THREADS, wave = threadIdx.x >> 6;
    const int ws = __builtin_amdgcn_readfirstlane((int)threadIdx.x / K::THREADS);
    // lane -> (tx, ty).  A ds_read_b64 is serviced in two groups of 32 lanes; the 32 lanes of a group must fall into 64
    // different banks.  64-wide tiles: a group is one row of 32 lanes (64 consecutive floats).  16-wide tiles (row stride
    // LW = 24 floats): a group takes 4 rows of 8 lanes; rows 2 apart start 48 = -16 (mod 64) banks apart, so lanes 0-31 take
    // the even rows of the wave's eight and lanes 32-63 the odd ones (banks 0-15, 48-63, 32-47, 16-31).
    const int tx = l & (K::TWL - 1);
    const int ty = K::TWL == 32 ? (l >> 5) : ((l >> 6) * 8 + 2 * ((l >> 3) & 3) + ((l >> 5) & 1));
    const int x0 = x_origin + bx * K::TW, y0 = by * K::TYB;
    const int px = x0 + tx * 2, py = y0 + ty;
    const unsigned plane = (unsigned)(H * W);
    constexpr unsigned kOut = 0x40000000u;

    // the F halo tile of this channel group -> LDS (LDS-DMA through a buffer descriptor: outside the image, in the padding
    // and past the group's last channel the hardware range check delivers zeros)
    {
        const float* baseF = F + ((size_t)b * Ctot + c_begin) * plane;
        const auto frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(baseF), 0, (int)((size_t)C * plane * 4), 0x00020000);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int s = it * NT + (int)threadIdx.x;
            const int c = s / K::SC;
            const int r = s - c * K::SC;
            const int ly = r / (LW / 4);
            const int gy = y0 - R + ly, gx = x0 - R + (r - ly * (LW / 4)) * 4;
            const bool in = (c < CH) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);      // (no short circuit: no branches)
            const unsigned off = in ? ((unsigned)c * plane + (unsigned)(gy * W + gx)) * 4u : kOut;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(frs, (lds_ptr)(tile + (it * NT + wave * 64) * 4), 16, (int)off, 0, 0, 0);
        }
    }

    // upstream gradient of this sample: 81 planes behind one descriptor; a lane's pair (x, x + 1) of tap (i, j) is one 8-byte load
    const float* gb = g + (size_t)b * DD * DD * plane;
    const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gb), 0, (int)((size_t)DD * DD * plane * 4), 0x00020000);
    const unsigned lane_off = (unsigned)(py * W + px) * 4u;
    // a displacement row of weights travels in two halves: the loads (issued one row AHEAD, raw bits parked in registers)
    // and, behind that row's arithmetic, the scaling into the packed-FMA layout -- consuming a value where it is loaded
    // would make hipcc wait for it there
    auto request_row = [&](v2u (&raw)[DD], int i) {
#pragma unroll
        for (int j = 0; j < DD; ++j) {
            const int pl = mode ? (2 * R - i) * DD + (2 * R - j) : i * DD + j;             // wave-uniform
            const unsigned uni = (unsigned)pl * plane * 4u + (unsigned)(mode * ((i - R) * W + (j - R)) * 4);
            raw[j] = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(grs, (int)(lane_off + uni), 0, 0));
        }
    };
    auto take_row = [&](RowWeights<R>& w, const v2u (&raw)[DD]) {
#pragma unroll
        for (int j = 0; j < DD; ++j) w.set(j, __uint_as_float(raw[j].x) * inv_c, __uint_as_float(raw[j].y) * inv_c);
    };
    // AHEAD rows of weights are in flight at any time: a row of arithmetic (CH x 15 instructions, ~0.45 us) is much shorter
    // than a first-touch HBM access (1-2 us under load); with one row of lead every row waited for its weights (level 2:
    // 67 us, 9 x ~1 us of exposed latency per workgroup).  Occupancy is set by LDS (two workgroups per CU), so the registers are free.
    RowWeights<R> w;
    v2u raw[AHEAD][DD];
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) request_row(raw[k], min(k, DD - 1));

    // this wave's pieces of the tile have landed (the weights may still fly)
    vm_wait<(AHEAD * DD > 63 ? 63 : AHEAD * DD)>();
    __builtin_amdgcn_s_barrier();                    // ... and everyone else's.  The waves do not meet again.

    v2f acc[CH][2];
#pragma unroll
    for (int c = 0; c < CH; ++c) { acc[c][0] = v2f{0.f, 0.f}; acc[c][1] = v2f{0.f, 0.f}; }
    const unsigned rows_addr = (unsigned)(size_t)(lds_cfloat*)(tile + ws * CH * K::SC * 4 + ty * LW + tx * 2);
    constexpr int PF = (2 * NCOL <= 15) ? 2 : 1;
    constexpr int HALF = (CH + 1) / 2;
    static_assert((HALF - 1) * K::SC * 16 + (LW + 2 * R) * 4 < 65536, "ds_read offset field");
    using Step0 = RsStep<0, CH, PF, R, NCOL, K::SC * 16, HALF>;
#pragma unroll 1
    for (int i0 = 0; i0 < DD; i0 += AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const int i = i0 + k;
            if (DD % AHEAD != 0 && i >= DD) break;   // (wave-uniform)
            take_row(w, raw[k]);                     // (the only vmcnt wait of the row: it was requested AHEAD rows ago)
            __builtin_amdgcn_sched_barrier(0);
            request_row(raw[k], min(i + AHEAD, DD - 1));
            __builtin_amdgcn_sched_barrier(0);
            const unsigned a_lo = rows_addr + (unsigned)(i * LW * 4);
            const unsigned a_hi = a_lo + (unsigned)(HALF * K::SC * 16);
            v2f row[PF + 1][NCOL];
            Step0::template load<0>(row, a_lo, a_hi);
            if constexpr (PF > 1) Step0::template load<1>(row, a_lo, a_hi);
            Step0::run(w, acc, row, a_lo, a_hi);
        }
    }
    if (py < H && px < W) {
        float* op = out + (((size_t)b * Ctot + c_begin + ws * CH) * H + py) * W + px;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (ws * CH + c < C)
                *reinterpret_cast<float2*>(op + (size_t)c * plane) = make_float2(acc[c][0].x + acc[c][0].y, acc[c][1].x + acc[c][1].y);
    }
}

template <int R, int CH, int TYB, int AHEAD>
__global__ __launch_bounds__((BwdRsCfg<R, CH, TYB>::THREADS)) void corr_bwd_rs_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         const float* __restrict__ g, float* __restrict__ gf1,
                                                         float* __restrict__ gf2, int Ctot, int H, int W,
                                                         int tiles_x, int tiles_y, int ngrp, float inv_c) {
    using K = BwdRsCfg<R, CH, TYB>;
    constexpr int NT = K::THREADS, ITER = (CH * K::SC + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float tile[ITER * NT * 4];
    corr_bwd_rs_body<R, CH, TYB, AHEAD>(tile, xcd_remap(blockIdx.x, gridDim.x), 0, f1, f2, g, gf1, gf2, Ctot, H, W,
                                        tiles_x, tiles_y, ngrp, inv_c);
}

// Mixed tile shapes in one launch: 64x8 tiles over the whole 64-pixel columns of the map and 16x32 tiles over the remaining
// <= 48 pixels, in columns of 16 (208 = 3 x 64 + 16: with 64-wide tiles only, the fourth tile column runs three quarters empty -- 19 % of all
// lanes idle, and the arithmetic of this kernel is what bounds it).  Both shapes cover 512 pixels with 256 lanes, so the items
// take the same time; the narrow ones are spread evenly through the (XCD-local) item order.
template <int R, int CH, int AHEAD>
__global__ __launch_bounds__(256) void corr_bwd_rs_mixed_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                const float* __restrict__ g, float* __restrict__ gf1,
                                                                float* __restrict__ gf2, int Ctot, int H, int W,
                                                                int wide_tiles_x, int wide_tiles_y, int narrow_tiles_x, int narrow_tiles_y,
                                                                int n_narrow, int stride, int ngrp, float inv_c) {
    using KW = BwdRsCfg<R, CH, 8>;
    using KN = BwdRsCfg<R, CH, 32>;
    constexpr int TILE_W = ((CH * KW::SC + 255) / 256) * 256 * 4, TILE_N = ((CH * KN::SC + 255) / 256) * 256 * 4;
    __shared__ __attribute__((aligned(16))) float tile[TILE_W > TILE_N ? TILE_W : TILE_N];
    const int u = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = min(n_narrow, (u + 1) / stride);                 // narrow items at positions stride - 1, 2 stride - 1, ... up to and including u
    const bool narrow = (u % stride == stride - 1) && (u / stride < n_narrow);      // (workgroup-uniform)
    if (narrow)
        corr_bwd_rs_body<R, CH, 32, AHEAD>(tile, nb - 1, wide_tiles_x * 64, f1, f2, g, gf1, gf2, Ctot, H, W, narrow_tiles_x, narrow_tiles_y, ngrp, inv_c);
    else
        corr_bwd_rs_body<R, CH, 8, AHEAD>(tile, u - nb, 0, f1, f2, g, gf1, gf2, Ctot, H, W, wide_tiles_x, wide_tiles_y, ngrp, inv_c);
}

template <int R, int CH, int AHEAD>
int launch_bwd_rs_mixed(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
                        int B, int C, int H, int W, hipStream_t s) {
    const int ngrp = ceil_div(C, CH);
    const int wide_x = W / 64, wide_y = ceil_div(H, 8), narrow_x = ceil_div(W % 64, 16), narrow_y = ceil_div(H, 32);
    const int n_wide = wide_x * wide_y * B * 2 * ngrp, n_narrow = narrow_x * narrow_y * B * 2 * ngrp;
    const int total = n_wide + n_narrow, stride = total / n_narrow;
    UNFLOW_LAUNCH((corr_bwd_rs_mixed_kernel<R, CH, AHEAD>), dim3(total), dim3(256), 0, s,
                       f1, f2, g, gf1, gf2, C, H, W, wide_x, wide_y, narrow_x, narrow_y, n_narrow, stride, ngrp, 1.0f / C);
    return unflow_launch_status();
}

template <int R, int CH, int TYB, int AHEAD>
int launch_bwd_rs(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
                  int B, int C, int H, int W, hipStream_t s) {
    using K = BwdRsCfg<R, CH, TYB>;
    const int tx = ceil_div(W, K::TW), ty = ceil_div(H, K::TYB), ngrp = ceil_div(C, CH);
    UNFLOW_LAUNCH((corr_bwd_rs_kernel<R, CH, TYB, AHEAD>), dim3(tx * ty * B * 2 * ngrp), dim3(K::THREADS), 0, s,
                       f1, f2, g, gf1, gf2, C, H, W, tx, ty, ngrp, 1.0f / C);
    return unflow_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Small-map backward (pyramid levels 5, 6: 8x26 and 4x13 pixels, 128-196 channels).  Too few pixels
// for the tile kernels (16 samples x 1 tile), and the per-element kernel re-reads 2*DD*DD gradient
// values per output.  Here a lane owns one pixel of the WHOLE map and keeps its DD*DD upstream
// gradients in registers (mode 1: gathered at the displaced positions, planes flipped); a workgroup =
// PXL pixels x CSUB channel phases of one (sample, channel chunk, mode); the chunk's F planes sit
// zero-padded in LDS, so the DD*DD taps of a lane are stride-1 LDS reads and need no bounds checks.
// grid (pixel blocks, channel chunks, 2*B).
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256, 3) void corr_bwd_small_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                             const float* __restrict__ g, float* __restrict__ gf1,
                                                             float* __restrict__ gf2, int C, int H, int W, int pxl,
                                                             int cch, float inv_c) {
    constexpr int DD = 2 * R + 1;
    UNFLOW_DYNAMIC_LDS(float, planes);                 // cch x (H+2R) x (W+2R), zero padded
    const int mode = blockIdx.z & 1, b = blockIdx.z >> 1;
    const float* __restrict__ F = mode ? f1 : f2;
    float* __restrict__ out = mode ? gf2 : gf1;
    const int PH = H + 2 * R, PW = W + 2 * R, plane = H * W, pplane = PH * PW;
    const int csub = 256 / pxl;                        // channel phases per workgroup
    const int lp = threadIdx.x % pxl, cs = threadIdx.x / pxl;
    const int q = blockIdx.x * pxl + lp;               // this lane's pixel
    const bool live = q < plane;
    const int y = live ? q / W : 0, x = live ? q - y * W : 0;
    const int c0 = blockIdx.y * cch;

    gfloat* zline = zero_line();
    float wr[DD][DD];
    const float* gb = g + (size_t)b * DD * DD * plane;
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
        for (int j = 0; j < DD; ++j) {
            const int sy = mode ? y + i - R : y, sx = mode ? x + j - R : x;
            const int pl = mode ? (2 * R - i) * DD + (2 * R - j) : i * DD + j;
            const bool ok = live && sy >= 0 && sy < H && sx >= 0 && sx < W;
            const int cy = min(max(sy, 0), H - 1), cx = min(max(sx, 0), W - 1);
            gfloat* real = (gfloat*)(gb + (size_t)pl * plane + cy * W + cx);         // computed unconditionally: a select, no branch
            gfloat* src = ok ? real : zline;                                         // validity folded into the address
            wr[i][j] = *src * inv_c;
        }

    // stage the chunk's planes (zero padded); F[c][yy][xx] lands at planes[c][yy+R][xx+R]
    const int nch = min(cch, C - c0);
#pragma unroll 4
    for (int e = threadIdx.x; e < cch * pplane; e += 256) {
        const int c = e / pplane, r = e - c * pplane;
        const int yy = r / PW - R, xx = r - (r / PW) * PW - R;
        float v = 0.f;
        if (c < nch && yy >= 0 && yy < H && xx >= 0 && xx < W) v = F[((size_t)(b * C + c0 + c)) * plane + yy * W + xx];
        planes[e] = v;
    }
    __syncthreads();
    if (!live) return;
    for (int c = cs; c < nch; c += csub) {
        const float* p = planes + c * pplane + y * PW + x;       // tap (i, j) at p[i * PW + j]
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < DD; ++i) {
#pragma unroll
            for (int j = 0; j < DD; ++j) {
                if ((i * DD + j) & 1) a1 = fmaf(wr[i][j], p[i * PW + j], a1);
                else a0 = fmaf(wr[i][j], p[i * PW + j], a0);
            }
            __builtin_amdgcn_sched_barrier(0);           // one tap row in flight at a time: the DD*DD gradients own the registers
        }
        out[((size_t)(b * C + c0 + c)) * plane + q] = a0 + a1;
    }
}

// 0 = not applicable (map too large / LDS would not fit a useful chunk)
template <int R>
int launch_bwd_small(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
                     int B, int C, int H, int W, hipStream_t s, bool* launched) {
    *launched = false;
    const int plane = H * W, pplane = (H + 2 * R) * (W + 2 * R);
    if (plane > 1024 || B > 32767) return 0;
    int pxl = 32;
    while (pxl < plane && pxl < 256) pxl <<= 1;
    const int csub = 256 / pxl;
    // few channels per workgroup: these maps are tiny, so the launch wants many short workgroups (the per-lane
    // gradient gather is repeated per chunk, out of L2) rather than a long serial channel loop
    // aim at ~1024 workgroups (4 per CU): levels 5 / 6 of the 832x256 step get 4 / 8 channels per workgroup
    int cch = ceil_div(ceil_div(C * 2 * B * ceil_div(plane, pxl), 1024), csub) * csub;
    if ((size_t)cch * pplane * sizeof(float) > 40 * 1024) cch = (int)(40 * 1024 / sizeof(float) / pplane) / csub * csub;
    if (cch < csub || cch < 1) return 0;
    dim3 grid(ceil_div(plane, pxl), ceil_div(C, cch), 2 * B);
    UNFLOW_LAUNCH((corr_bwd_small_kernel<R>), grid, dim3(256), (size_t)cch * pplane * sizeof(float), s,
                       f1, f2, g, gf1, gf2, C, H, W, pxl, cch, 1.0f / C);
    *launched = true;
    return unflow_launch_status();
}

// ---- any-radius fallback (one thread per output element, direct global reads) ----
#include "corr_generic.h"

// Small maps (pyramid levels 5, 6: a few hundred pixels, 128-196 channels): the per-element kernel above is one serial chain of
// C dependent-latency load pairs per lane (~15 us for 1.5 MB at level 6).  Here a workgroup = 64 consecutive outputs x CS channel
// slices (one wave per slice, the same coalescing along x); the slices' partial sums meet in LDS and are added in a fixed order.
template <int CS>
__global__ __launch_bounds__(64 * CS) void corr_fwd_split_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                 float* __restrict__ cv, int B, int C, int H, int W, int R, float inv_c) {
    __shared__ float red[CS][64];
    const int DD = 2 * R + 1;
    const size_t n = (size_t)B * DD * DD * H * W, plane = (size_t)H * W;
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t t = (size_t)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (t < n) {
        const int x = t % W, y = (t / W) % H, ij = (t / plane) % (DD * DD);
        const int b = t / (plane * DD * DD);
        const int sy = y + ij / DD - R, sx = x + ij % DD - R;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            const int per = (C + CS - 1) / CS, c0 = slice * per, c1 = min(C, c0 + per);
            const float* p1 = f1 + ((size_t)b * C * H + y) * W + x;
            const float* p2 = f2 + ((size_t)b * C * H + sy) * W + sx;
            float a0 = 0.f, a1 = 0.f;
            int c = c0;
#pragma unroll 4
            for (; c + 1 < c1; c += 2) {
                a0 = fmaf(p1[c * plane], p2[c * plane], a0);
                a1 = fmaf(p1[(c + 1) * plane], p2[(c + 1) * plane], a1);
            }
            if (c < c1) a0 = fmaf(p1[c * plane], p2[c * plane], a0);
            s = a0 + a1;
        }
    }
    red[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && t < n) {
        float tot = red[0][lane];
#pragma unroll
        for (int k = 1; k < CS; ++k) tot += red[k][lane];
        cv[t] = tot * inv_c;
    }
}


template <int R, int PX, int DG, int CC>
int launch_fwd(const float* f1, const float* f2, float* cv, int B, int C, int H, int W, hipStream_t s) {
    using K = CorrCfg<R, PX, DG, CC>;
    dim3 grid(ceil_div(W, K::TW), ceil_div(H, TY), B * K::NG);
    UNFLOW_LAUNCH((corr_fwd_kernel<R, PX, DG, CC>), grid, dim3(256), 0, s, f1, f2, cv, C, H, W, 1.0f / C);
    return unflow_launch_status();
}

template <int R, int PX, int DG, int CC>
int launch_bwd(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
               int B, int C, int H, int W, hipStream_t s) {
    using K = CorrCfg<R, PX, DG, CC>;
    dim3 grid(ceil_div(W, K::TW), ceil_div(H, TY), B * K::NG);
    if (K::NG > 1) {                                      // row groups add their partial gradients atomically
        unflow_zero_async(gf1, (size_t)B * C * H * W, s);
        unflow_zero_async(gf2, (size_t)B * C * H * W, s);
    }
    UNFLOW_LAUNCH((corr_bwd_kernel<R, PX, DG, CC, 0>), grid, dim3(256), 0, s, f2, g, gf1, C, H, W, 1.0f / C);
    UNFLOW_LAUNCH((corr_bwd_kernel<R, PX, DG, CC, 1>), grid, dim3(256), 0, s, f1, g, gf2, C, H, W, 1.0f / C);
    return unflow_launch_status();
}

}  // namespace

// Tuning knobs (tools/ builds with -DUNFLOW_TUNING only; the shipped library never reads the environment):
// UNFLOW_CORR_VARIANT forces a d=4 forward code path
//   1: 64x8 tiles, 2 px/lane, all 81 displacements per lane, register-staged LDS tiles
//   2: 32x8 tiles, 1 px/lane, all 81 displacements per lane
//   3: 32x8 tiles, 1 px/lane, displacement rows split over 3 workgroups
//   4: one lane per output element, direct (cached) global reads    (tiny maps, many channels)
//   7, 9: LDS-DMA ring kernel with 9 / 3 displacement rows per workgroup
// UNFLOW_CORR_BWD forces a d=4 backward path (1, 3, 4, 5: group-split ring kernel with 64x4 / 64x8 tiles and 2 / 4
// channels per stage; 2: per element; 6: tile kernel), UNFLOW_CORR_GROUPS its number of channel groups.
#ifdef UNFLOW_TUNING
static int env_int(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
static int forced_variant() { return env_int("UNFLOW_CORR_VARIANT"); }      // re-read per call: one process sweeps them
static int forced_split() { return env_int("UNFLOW_CORR_SPLIT"); }
static int forced_bwd() { return env_int("UNFLOW_CORR_BWD"); }
static int forced_groups() { return env_int("UNFLOW_CORR_GROUPS"); }
#else
static inline int forced_variant() { return 0; }
static inline int forced_split() { return 0; }
static inline int forced_bwd() { return 0; }
static inline int forced_groups() { return 0; }
#endif

static inline bool mid_size(int variant) { return variant == 9 || variant == 12 || variant == 13; }   // levels 3, 4

// The backward on the matrix cores (corr_mfma.h) is chosen PER CALL (unflow_corr_bwd_ex's `arithmetic`, include/unflow_hip.h): the library keeps
// no mode of its own.  UNFLOW_CORR_BWD_AUTO = what passed a whole `-m gpu` run on an MI355X for the shape.
static bool mfma_served(const float* f1, const float* f2, const float* g, const float* gf1, const float* gf2, int B, int C, int H, int W, int R) {
    return (W & 3) == 0 && (C & 15) == 0 && (long)B * H * W >= 8192 && W >= 16 && H >= 4 * R      // (level 4 at d = 8, 16 rows: 50 us against 47 for the fp32 kernel)
           && ((((size_t)f1 | (size_t)f2 | (size_t)g | (size_t)gf1 | (size_t)gf2) & 15) == 0) && mf_offsets_fit(C, H, W, R);
}
// rows per wave (tools/microbench.py corr_bwd_mf, rows sweep on the tuning library): the halo steps of a chunk cost loads but no row
// pairs, and the fastest launch is the one with about one wave per SIMD at d = 8 (412 registers: one wave per SIMD fits) and one and
// a half at d = 4 -- level 2 / 3 of 832x256 at d = 8: 32 / 16 rows (150 / 85 us; 194 / 130 with the next size), d = 4: 16 / 8 rows
static int mfma_rows(int B, int C, int H, int W, int NCG, int R) {
#ifdef UNFLOW_TUNING
    if (getenv("UNFLOW_CORR_MF_ROWS")) return atoi(getenv("UNFLOW_CORR_MF_ROWS"));
#endif
    const long cols = (long)B * 2 * ceil_div(W, 16) * ceil_div(C, 16 * NCG);
    const long want = R > 4 ? 768 : 1536;
    int rows = 32;
    while (rows > 8 && cols * ceil_div(H, rows) < want) rows >>= 1;
    return rows;
}

static int pick_variant(int B, int C, int H, int W) {
    const int f = forced_variant();
    if (f) return f;
    // measured on MI355X at the 832x256 pyramid shapes (tools/microbench.py corr):
    //   level 2 [16,32,64,208]: ring, all 81 displacements per workgroup (7)   37 us
    //   level 3 [16,64,32,104], level 4 [16,96,16,52]: ring, 3 displacement rows per workgroup (9)  29 / 30 us
    //   levels 5, 6: one lane per output element (4)   14 / 12 us  (a whole-map LDS kernel -- one workgroup per sample and
    //   displacement row, channels streamed through LDS -- measured 31 / 26 us: staging 2 x 106 KB per workgroup as dword
    //   LDS-DMA pieces (W = 26 / 13 rows are not 16-byte aligned) costs more than the cached global reads it replaces; a
    //   row-per-lane kernel -- a lane keeps the 9 accumulators of one displacement row, 16 channel phases per workgroup,
    //   1 + 9 cached loads per 9 FMAs instead of 18 -- measured 23 / 14 us: fewer, fatter lanes lose the memory-level
    //   parallelism that 67 k thin lanes have)
    // d=8 (tools/microbench.py corr8): ring with 3 of the 17 rows per workgroup 125 / 52 / 46 us at levels 2 / 3 / 4
    // (tile kernel 234 / 128 / 85), one lane per element 29 / 15 us at levels 5 / 6 (tile kernel 158 / 226)
    const long px = (long)B * H * W;
    const bool dma_ok = ((W & 3) == 0);
    if (W >= 96 && px >= 131072) return dma_ok ? 7 : 1;
    if (px >= 32768 && dma_ok) return 12;      // level 3: two channel phases per workgroup (24.5 us; plain ring 28-29)
    if (px >= 8192 && dma_ok) return 13;       // level 4: four channel phases (19.4 us; plain ring 29-30)
    if (px >= 32768) return 3;
    return 4;
}

extern "C" int unflow_corr_fwd(const float* f1, const float* f2, float* cv, int B, int C, int H, int W,
                               int d, void* stream) {
    UNFLOW_REQUIRE(f1 && f2 && cv && B > 0 && C > 0 && H > 0 && W > 0 && d >= 0);
    hipStream_t s = (hipStream_t)stream;
    int variant = 4;
    const bool ring_ok = (W & 3) == 0 && ((((size_t)f1 | (size_t)f2) & 15) == 0)      // LDS-DMA moves aligned 16-byte pieces
                         && (size_t)C * H * W * 4 < kOutOfRange;                        // 32-bit buffer offsets, bit 30 = "outside"
    switch (d) {
        case 1: return launch_fwd<1, 2, 3, 8>(f1, f2, cv, B, C, H, W, s);
        case 2: return launch_fwd<2, 2, 5, 8>(f1, f2, cv, B, C, H, W, s);
        case 4: variant = pick_variant(B, C, H, W);
                if (variant == 1) return launch_fwd<4, 2, 9, 8>(f1, f2, cv, B, C, H, W, s);
                if (variant == 2) return launch_fwd<4, 1, 9, 8>(f1, f2, cv, B, C, H, W, s);
                if (variant == 3) return launch_fwd<4, 1, 3, 8>(f1, f2, cv, B, C, H, W, s);
                if (variant == 7 && ring_ok) {
#ifdef UNFLOW_TUNING
                    // (208 = 3 x 64 + 16: the remainder column as 16 x 32 tiles in the same launch, as the backward does -- measured for
                    // the forward, whose time is stores and DMA rather than arithmetic: 37.6 vs 36.4 us back to back, 34.7 vs 32.7 in the
                    // step: not taken; UNFLOW_CORR_SPLIT=65 selects it in tuning builds)
                    if (forced_split() == 65 && W % 64 > 0 && W % 64 <= 16 && W >= 64 && H >= 32) return launch_fwd_ring_mixed<4, 2, 9>(f1, f2, cv, B, C, H, W, s);
#endif
                    return launch_fwd_ring<4, 2, 9>(f1, f2, cv, B, C, H, W, s);
                }
                if (variant == 9 && ring_ok) return launch_fwd_ring<4, 2, 3>(f1, f2, cv, B, C, H, W, s);
                if (variant == 12 && ring_ok) return launch_fwd_ringp<4, 2, 3, 2>(f1, f2, cv, B, C, H, W, s);
                if (variant == 13 && ring_ok) return launch_fwd_ringp<4, 2, 3, 4>(f1, f2, cv, B, C, H, W, s);
                if (variant >= 7) return launch_fwd<4, 2, 9, 8>(f1, f2, cv, B, C, H, W, s);
                break;
        case 8: variant = pick_variant(B, C, H, W);
                if ((variant == 7 || mid_size(variant)) && ring_ok) return launch_fwd_ring<8, 2, 3>(f1, f2, cv, B, C, H, W, s);
                if (variant == 4 || (variant == 3 && (long)B * H * W < 8192)) break;
                return launch_fwd<8, 1, 6, 8>(f1, f2, cv, B, C, H, W, s);
        default: break;
    }
    const size_t n = (size_t)B * (2 * d + 1) * (2 * d + 1) * H * W;
    // few outputs, many channels: split the channels over the waves of a workgroup (tools/microbench.py corr_split_sweep; level 6,
    // 67 k outputs x 196 channels: 11.7 / 9.5 / 7.7 / 9.0 us with 1 / 2 / 4 / 8 slices; at level 5, 269 k outputs already fill the
    // chip: 14.5 / 16.0 / 16.4 / 20.8 us)
    const int split = forced_split() ? forced_split() : (C >= 64 && n < 98304 ? 4 : 0);
    if (split == 8 || split == 4 || split == 2) {
        const dim3 grid((unsigned)((n + 63) / 64));
        if (split == 8) UNFLOW_LAUNCH((corr_fwd_split_kernel<8>), grid, dim3(512), 0, s, f1, f2, cv, B, C, H, W, d, 1.0f / C);
        else if (split == 4) UNFLOW_LAUNCH((corr_fwd_split_kernel<4>), grid, dim3(256), 0, s, f1, f2, cv, B, C, H, W, d, 1.0f / C);
        else UNFLOW_LAUNCH((corr_fwd_split_kernel<2>), grid, dim3(128), 0, s, f1, f2, cv, B, C, H, W, d, 1.0f / C);
        return unflow_launch_status();
    }
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    UNFLOW_LAUNCH(corr_fwd_generic, dim3(blocks), dim3(256), 0, s, f1, f2, cv, B, C, H, W, d, 1.0f / C);
    return unflow_launch_status();
}

// d = 8 under UNFLOW_CORR_BWD_AUTO: the matrix-core kernel where it is served (round 5: 150-158 us against 245-272 at level 2).  false until the
// kernel has passed a complete `-m gpu` run at the reference-generated fixture (tests/golden/g6_corr_d8.npz); UNFLOW_CORR_BWD_MFMA asks for it.
static constexpr bool kAutoMatrixCoresAtD8 = false;

extern "C" int unflow_corr_bwd_ex(const float* f1, const float* f2, const float* gcv, float* gf1, float* gf2,
                                  int B, int C, int H, int W, int d, int arithmetic, void* stream) {
    UNFLOW_REQUIRE(f1 && f2 && gcv && gf1 && gf2 && B > 0 && C > 0 && H > 0 && W > 0 && d >= 0);
    UNFLOW_REQUIRE(arithmetic >= UNFLOW_CORR_BWD_AUTO && arithmetic <= UNFLOW_CORR_BWD_FP32_NEXT);
    hipStream_t s = (hipStream_t)stream;
    int variant = 4;
    switch (d) {
        case 1: return launch_bwd<1, 2, 3, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
        case 2: return launch_bwd<2, 2, 5, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
        case 4: {
                variant = pick_variant(B, C, H, W);
                // matrix-core form (corr_mfma.h): at d = 4 level 2 it measures level with the fp32 row-streamed kernel back to back
                // (59 vs 57 us), so it is taken only on request (UNFLOW_CORR_BWD_MFMA)
                if (arithmetic == UNFLOW_CORR_BWD_MFMA && mfma_served(f1, f2, gcv, gf1, gf2, B, C, H, W, 4))
                    return launch_bwd_mf<4, 2, 1, 1>(f1, f2, gcv, gf1, gf2, B, C, H, W, mfma_rows(B, C, H, W, 2, 4), s);
                const bool ring_ok = (W & 3) == 0 && ((((size_t)f1 | (size_t)f2) & 15) == 0)     // LDS-DMA moves aligned 16-byte pieces
                                     && gs_offsets_fit(C, H, W, 4);
                const int fb = forced_bwd();
                // work items (64x8 tile, gradient) of the group-split ring kernel; smaller maps split the channels over
                // workgroups until every CU has one (measured: level 3 = 256 items, 1 group: 35 us; level 4 = 64 items,
                // 4 groups: 20 us; more groups than that only repeat the upstream-gradient gather)
                const int items = ceil_div(W, 64) * ceil_div(H, 8) * B * 2;
                int groups = forced_groups() ? forced_groups() : (items >= 256 ? 1 : 256 / items);
                // group-split ring kernel, 64x8 tiles, 4 channels per stage: levels 2-4 (83 / 35 / 20 us; the tile kernel
                // with all 81 gradients per lane takes 107 us at level 2)
                // (measured and dropped: a persistent form -- 256 workgroups walking 4 items each, the next tile's gradient
                // planes touched into L2 during the ring stages -- 109 us at level 2 against 95 for this one: the exposed
                // gather is not what limits the kernel, and 1024 independently scheduled workgroups balance better)
                // round 3: the row-streamed kernel for levels 2-4 (back to back 63-67 / 35 / 17 us against 86 / 38-41 / 18-19.6 for
                // the group-split ring kernel, which stays selectable in tuning builds: UNFLOW_CORR_BWD=4).  16 channels per work
                // item (8 on small maps: more, shorter workgroups); 16 x 32 tiles where 64-wide ones would leave a mostly empty
                // last tile column and the map is tall enough (level 3: 104 = 6.5 x 16)
                // In the train step (in-step A/B, tools/gpu_r3_experiments.sh instep): level 2 73.6 vs 91.8 us, level 3 37.5 (64 x 8 tiles; 39.5 with
                // 16 x 32) vs 35.0, level 4 18.6 vs 20.1 -- so level 3 (32768 <= pixels < 131072, 256 items: one round of the
                // group-split kernel) stays on the group-split ring kernel
                if (ring_ok && fb == 0 && (variant == 7 || mid_size(variant))) {
                    const long px = (long)B * H * W;
                    if (px < 32768) return launch_bwd_rs<4, 8, 8, 2>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
                    if (px >= 131072) {
                        // a remainder of <= 16 pixels beside the 64-pixel columns (208 = 3 x 64 + 16) as 16 x 32 tiles in the same launch:
                        // 65.9 -> 59.3 us back to back, 74-75 -> 69.2 us in the step.  (Wider remainders -- two or three 16-pixel columns,
                        // UNFLOW_CORR_BWD=33 in tuning builds -- measured at level 3, 104 = 64 + 40: 37.8 vs 33.3 us, no gain.)
                        const int rem = W % 64;
                        if (forced_groups() != 64 && rem > 0 && rem <= 16 && W >= 64 && H >= 32)
                            return launch_bwd_rs_mixed<4, 16, 2>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
                        return launch_bwd_rs<4, 16, 8, 2>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
                    }
                    return launch_bwd_gs<4, 4, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, groups, s);
                }
                if (ring_ok && fb == 4)
                    return launch_bwd_gs<4, 4, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, groups, s);
                if (fb == 6 || variant == 1 || variant == 7) return launch_bwd<4, 2, 9, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
                if (variant == 2 || variant == 3 || (mid_size(variant) && (long)B * H * W >= 32768))
                    return launch_bwd<4, 1, 9, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
                if (arithmetic == UNFLOW_CORR_BWD_FP32_NEXT && fb == 0) {      // round 6, not measured: gradient rows through registers, 4 / 8 channels per lane
                    bool launched = false;
                    const int rc = launch_bwd_smallrows<4, 3>(f1, f2, gcv, gf1, gf2, B, C, H, W, s, &launched);
                    if (launched) return rc;
                }
                if (fb != 2) {                           // small maps (levels 5, 6): whole-map kernel; UNFLOW_CORR_BWD=2: per-element
                    bool launched = false;
                    const int rc = launch_bwd_small<4>(f1, f2, gcv, gf1, gf2, B, C, H, W, s, &launched);
                    if (launched) return rc;
                }
                break;
        }
        case 8: variant = pick_variant(B, C, H, W);
                // round 5: banded bf16x3 products on the matrix cores: level 2 149 us against 271 for the fp32 row-streamed kernel
                // (tools/proto/corr_bwd_mfma.hip); ~4e-6 of the largest gradient away from the fp32 sums
                if ((arithmetic == UNFLOW_CORR_BWD_MFMA || (arithmetic == UNFLOW_CORR_BWD_AUTO && kAutoMatrixCoresAtD8)) && mfma_served(f1, f2, gcv, gf1, gf2, B, C, H, W, 8))
                    return launch_bwd_mf<8, 2, 2, 1>(f1, f2, gcv, gf1, gf2, B, C, H, W, mfma_rows(B, C, H, W, 2, 8), s);
                if (arithmetic == UNFLOW_CORR_BWD_FP32_NEXT && (variant == 4 || (long)B * H * W < 8192)) {      // round 6, not measured: levels 5 / 6 at d = 8
                    bool launched = false;
                    const int rc = launch_bwd_smallrows<8, 3>(f1, f2, gcv, gf1, gf2, B, C, H, W, s, &launched);
                    if (launched) return rc;
                }
                if (variant == 4 || (long)B * H * W < 8192) break;      // small maps: one lane per element (117 / 57 us at levels 5 / 6, tile kernel 283 / 378)
                if ((variant == 7 || mid_size(variant)) && (W & 3) == 0 && ((((size_t)f1 | (size_t)f2) & 15) == 0) && gs_offsets_fit(C, H, W, 8)) {
                    // round 3: row-streamed, 8 channels per item: 290 / 117 / 49 us at levels 2 / 3 / 4 (group-split ring kernel 343 / 129 / 85)
                    return launch_bwd_rs<8, 8, 8, 1>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
                }
                return launch_bwd<8, 1, 6, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
        default: break;
    }
    const size_t n = (size_t)B * C * H * W;
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    UNFLOW_LAUNCH(corr_bwd_generic, dim3(blocks), dim3(256), 0, s, f1, f2, gcv, gf1, gf2, B, C, H, W, d, 1.0f / C);
    return unflow_launch_status();
}

extern "C" int unflow_corr_bwd(const float* f1, const float* f2, const float* gcv, float* gf1, float* gf2,
                               int B, int C, int H, int W, int d, void* stream) {
    return unflow_corr_bwd_ex(f1, f2, gcv, gf1, gf2, B, C, H, W, d, UNFLOW_CORR_BWD_AUTO, stream);
}
