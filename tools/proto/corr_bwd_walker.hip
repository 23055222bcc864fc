// PROTOTYPE (not part of libunflow_hip.so): the level-2 cost-volume backward as a persistent "column walker".
//
// What the shipped row-streamed kernel (csrc/corr.hip, corr_bwd_rs_kernel) loses: staging (27 us), arithmetic (36 us) and the weight
// stream (8 us) ADD UP, because its two workgroups per CU run in lock-step and LDS (72 KB per 16-channel 64x8 tile with a 2.25x
// halo) leaves no room for a second buffer.  Here ONE 512-lane workgroup per CU walks a 64-pixel column strip of one (sample,
// gradient, 16-channel group) from top to bottom as 64x8 tiles t = 0, 1, 2, ...:
//   * the F rows live in an LDS ring of 32 halo rows (4 blocks of 8 rows x 16 channels x 72 floats = 147 456 B); going from tile
//     t to t + 1 only 8 NEW rows are staged (row halo re-used: staged volume 1.125x instead of 2.25x),
//   * two wave sets (waves 0-3 / 4-7) take the even / odd tiles, set 1 half a tile late.  A tile is two half-slots of nine
//     half-rows (a half-row = 8 of the 16 channels of one displacement row; the row's weights are loaded once), every wave meets
//     every other one at ONE s_barrier per half-slot, and the 8 new rows of tile k + 1 are DMA'd by all eight waves at the start of
//     half-slot k (the block they replace was last read in half-slot k - 1) -- so both sets compute all the time (two waves per
//     SIMD, as today) while the next rows arrive,
//   * the weights of the NEXT tile's first two rows are requested during the last two rows of the current one.
// Arithmetic order per output value is the row-streamed kernel's (same accumulators, same row order): results must be bit-identical.
//
// 16 samples x 2 gradients x 2 channel groups x 4 strips = 256 strips = one per CU at level 2; the 16-pixel fourth strip wastes
// 19 % of the lanes exactly like the 64-wide-only row-streamed kernel (launch_bwd_rs<4,16,8,2>, the apples-to-apples baseline
// below: 65.9-67.9 us back to back); the shipped mixed-tile launch is at 59.3-60.5.  Expected: 4.5 tile-pair times (~11 us) + the
// first 16 rows (~3 us) ~ 53 us.
//
// Build + run (the harness compares with unflow_corr_bwd of the same translation unit and times all three):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude tools/proto/corr_bwd_walker.hip -o tools/proto/corr_bwd_walker && tools/proto/corr_bwd_walker
#include "../../unopticalflow_amd/csrc/corr.hip"
#include <cstdio>
#include <cstring>
#include <vector>

UnflowTimingArm& unflow_timing_arm() { static thread_local UnflowTimingArm arm{nullptr, nullptr, false}; return arm; }

namespace {

constexpr int WK_R = 4, WK_DD = 9, WK_CH = 16, WK_LW = 64 + 2 * WK_R;
constexpr int WK_ROWB = WK_LW * 4;                 // bytes per halo row of one channel
constexpr int WK_CHB = 8 * WK_ROWB;                // bytes per channel of one ring block (8 rows)
constexpr int WK_BLKB = WK_CH * WK_CHB;            // bytes per ring block: 36 864
constexpr int WK_PIECES = WK_BLKB / 1024;          // 64-lane x 16-byte DMA pieces per block: 36
constexpr int WK_NCOL = WK_R + 1, WK_PF = 2;

// row-step over channels [ST, END) of the 16 (RsStep of corr.hip with a begin, one base address for all channels)
template <int ST, int END>
struct WkStep {
    template <int Q, int... Ks>
    static __device__ __forceinline__ void load_cols(v2f (&row)[WK_PF + 1][WK_NCOL], unsigned a, std::integer_sequence<int, Ks...>) {
        ((row[Q % (WK_PF + 1)][Ks] = lds_read_b64<Q * WK_CHB + 8 * Ks>(a)), ...);
    }
    template <int Q>
    static __device__ __forceinline__ void load(v2f (&row)[WK_PF + 1][WK_NCOL], unsigned a) {
        if constexpr (Q < END) load_cols<Q>(row, a, std::make_integer_sequence<int, WK_NCOL>{});
    }
    static __device__ __forceinline__ void run(const RowWeights<WK_R>& w, v2f (&acc)[WK_CH][2], v2f (&row)[WK_PF + 1][WK_NCOL], unsigned a) {
        if constexpr (ST < END) {
            load<ST + WK_PF>(row, a);
            constexpr int newer = (END - 1 - ST < WK_PF ? END - 1 - ST : WK_PF) * WK_NCOL;
            lds_wait<newer>();
            constexpr int rb = ST % (WK_PF + 1);
#pragma unroll
            for (int k = 0; k < WK_R; ++k) {
                acc[ST][0] = __builtin_elementwise_fma(w.p0[k], row[rb][k], acc[ST][0]);
                acc[ST][1] = __builtin_elementwise_fma(w.p1[k], row[rb][k + 1], acc[ST][1]);
            }
            acc[ST][0].x = fmaf(w.s0, row[rb][WK_R].x, acc[ST][0].x);
            acc[ST][1].y = fmaf(w.s1, row[rb][0].y, acc[ST][1].y);
            __builtin_amdgcn_sched_barrier(0);
            WkStep<ST + 1, END>::run(w, acc, row, a);
        }
    }
};
static_assert((WK_CH - 1) * WK_CHB + 8 * WK_NCOL < 65536, "ds_read offset field");

// ABL (timing only, wrong results): 1 no s_barrier in the tile loop, 2 no DMA in the tile loop, 4 no weight requests in the tile loop,
// 8 no LDS reads + FMAs (rows are empty)
// AH: displacement rows of weights in flight (2 or 3; row i sits in buffer i % AH, and because 9 % 3 == 0 the three rows requested
// during rows 6, 7, 8 are rows 0, 1, 2 of the set's next tile in their natural order)
template <int ABL, int AH = 2>
__global__ __launch_bounds__(512) void corr_bwd_walker_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                             const float* __restrict__ g, float* __restrict__ gf1,
                                                             float* __restrict__ gf2, int Ctot, int H, int W,
                                                             int strips_x, int ngrp, float inv_c) {
    constexpr int R = WK_R, DD = WK_DD;
    __shared__ __attribute__((aligned(16))) float ring[4 * WK_BLKB / 4];
    int u = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = u % ngrp; u /= ngrp;              // the strips that read the same gradient planes are neighbours
    const int mode = u & 1; u >>= 1;
    const int bx = u % strips_x;
    const int b = u / strips_x;
    const int ntiles = ceil_div(H, 8);
    const float* __restrict__ F = mode ? f1 : f2;
    float* __restrict__ out = mode ? gf2 : gf1;
    const int c_begin = cg * WK_CH;
    const int C = min(WK_CH, Ctot - c_begin);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int set = wave >> 2;                       // wave set 0: even tiles, 1: odd tiles
    const int x0 = bx * 64;                           // lane l of a set: pixels (x0 + 2 (l & 31), + 1) of row l >> 5 of its tile
    const unsigned plane = (unsigned)(H * W);
    constexpr unsigned kOut = 0x40000000u;
    // lane-derived values (tx, ty, offsets) are REcomputed where they are used: kept in registers across the tile loop they are
    // spilled, and a scratch reload costs an s_waitcnt vmcnt(0) -- i.e. the latency of the DMA pieces and weight rows just requested
    auto fresh_lane = []() __attribute__((always_inline)) {
        int id;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(id));
        return id;
    };
    const int wave_in_set = wave & 3;

    const float* baseF = F + ((size_t)b * Ctot + c_begin) * plane;
    const auto frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(baseF), 0, (int)((size_t)C * plane * 4), 0x00020000);
    float* baseO = out + ((size_t)b * Ctot + c_begin) * plane;
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(baseO, 0, (int)((size_t)C * plane * 4), 0x00020000);
    const float* gb = g + (size_t)b * DD * DD * plane;
    const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gb), 0, (int)((size_t)DD * DD * plane * 4), 0x00020000);

    // halo rows h0 .. h0 + 7 (h = y + R) of the 16 channels -> ring block (h0 / 8) % 4; the eight waves share its 36 pieces
    auto stage_block = [&](int h0) __attribute__((always_inline)) {
        float* blk = ring + ((h0 >> 3) & 3) * (WK_BLKB / 4);
        const int ln = fresh_lane();
#pragma unroll
        for (int it = 0; it < (WK_PIECES + 7) / 8; ++it) {
            const int p = wave + 8 * it;             // (wave-uniform)
            if (p < WK_PIECES) {
                const int s = p * 64 + ln;           // float4 slot in the block: [channel][row][18]
                const int c = s / 144;
                const int r = s - c * 144;
                const int ly = r / 18;
                const int gy = h0 - R + ly, gx = x0 - R + (r - ly * 18) * 4;
                const bool in = (c < C) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                const unsigned off = in ? ((unsigned)c * plane + (unsigned)(gy * W + gx)) * 4u : kOut;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(frs, (lds_ptr)(blk + p * 256), 16, (int)off, 0, 0, 0);
            }
        }
    };
    auto request_row = [&](v2u (&rw)[DD], int i, int t) __attribute__((always_inline)) {
        const int id = fresh_lane();
        const unsigned lane_off = (unsigned)((t * 8 + wave_in_set * 2 + (id >> 5)) * W + x0 + (id & 31) * 2) * 4u;
#pragma unroll
        for (int j = 0; j < DD; ++j) {
            const int pl = mode ? (2 * R - i) * DD + (2 * R - j) : i * DD + j;
            const unsigned uni = (unsigned)pl * plane * 4u + (unsigned)(mode * ((i - R) * W + (j - R)) * 4);
            rw[j] = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(grs, (int)(lane_off + uni), 0, 0));
        }
    };
    RowWeights<R> w;
    v2u raw[AH][DD];
    v2f acc[WK_CH][2];
    auto take_row = [&](const v2u (&rw)[DD]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < DD; ++j) w.set(j, __uint_as_float(rw[j].x) * inv_c, __uint_as_float(rw[j].y) * inv_c);
    };
    const unsigned ring_addr = (unsigned)(size_t)(lds_cfloat*)ring;
    auto row_addr = [&](int t, int i) __attribute__((always_inline)) {
        const int id = fresh_lane();
        const int h = t * 8 + wave_in_set * 2 + (id >> 5) + i;
        return ring_addr + (unsigned)((id & 31) * 8) + (unsigned)(((h >> 3) & 3) * WK_BLKB + (h & 7) * WK_ROWB);
    };
    // displacement row i of tile t, channels [C0, C1); its weights were requested two rows ago into raw[i & 1]
    auto channels = [&](int t, int i, auto c0, auto c1) __attribute__((always_inline)) {
        constexpr int C0 = decltype(c0)::value, C1 = decltype(c1)::value;
        if constexpr (ABL & 8) { acc[C0][0] += w.p0[0] * (float)i; return; }
        const unsigned a = row_addr(t, i);
        v2f row[WK_PF + 1][WK_NCOL];
        WkStep<C0, C1>::template load<C0>(row, a);
        WkStep<C0, C1>::template load<C0 + 1>(row, a);
        WkStep<C0, C1>::run(w, acc, row, a);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, AH == 3 ? 2 : 0>;
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;
    // take the weights of row i (buffer KB = i & 1, a register name) and put the next request into the freed buffer: row i + 2 of this
    // tile, or (rows 7, 8) rows 1, 0 of this set's next tile.  i is a run-time value (as in the row-streamed kernel: the 81 plane
    // offsets of a fully unrolled tile would all be kept in scalar registers)
    auto begin_row = [&](int t, int i, auto kb) __attribute__((always_inline)) {
        constexpr int KB = decltype(kb)::value;
        take_row(raw[KB]);
        __builtin_amdgcn_sched_barrier(0);
        const bool same = i + AH < DD;                // (wave-uniform)
        const bool more = t + 2 < ntiles;
        // no branch inside a row (the FMAs of a row must stay in its basic block): past the last tile of the set the two requests
        // re-read the last row
        const int ri = same ? i + AH : (more ? KB : DD - 1), rt = (same | !more) ? t : t + 2;
        if constexpr (!(ABL & 4)) request_row(raw[KB], ri, rt);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto full_row = [&](int t, int i, auto kb) __attribute__((always_inline)) { begin_row(t, i, kb); channels(t, i, I0{}, I16{}); };
    // a row index the optimiser cannot see through (a constant one turns its nine plane offsets into nine live scalar registers)
    auto rowi = [](int i) __attribute__((always_inline)) { asm volatile("" : "+s"(i)); return i; };

    stage_block(0);
    stage_block(8);                                   // tile 0: halo rows 0 .. 15
    __builtin_amdgcn_sched_barrier(0);                // (the weight requests stay behind the DMA: vm_wait counts on it)
#pragma unroll
    for (int k = 0; k < AH; ++k) request_row(raw[k], k, set);
    vm_wait<AH * WK_DD>();                            // this wave's pieces have landed (the weights behind them may still fly)
    __builtin_amdgcn_s_barrier();

    // Half-slot k: the first half of tile k (wave set k & 1) and the second half of tile k - 1 (the other set); all eight waves
    // first DMA the 8 new rows of tile k + 1.  Written per set as a loop over ITS tiles (two half-slots per iteration), so that the
    // accumulators are not carried around a loop: set 1 idles through half-slot 0, set 0 through the last one(s).
    auto slot_dma = [&](int k) __attribute__((always_inline)) {
        if constexpr (!(ABL & 2)) { if (k + 1 < ntiles) stage_block(8 * (k + 1) + 8); }         // (their block was last read in half-slot k - 1)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto slot_end = [&]() __attribute__((always_inline)) { if constexpr (!(ABL & 1)) __builtin_amdgcn_s_barrier(); };
    int slots_done = 0;
    if (set == 1) {
        slot_dma(0);
        vm_wait<0>();
        __builtin_amdgcn_s_barrier();
        slots_done = 1;
    }
#pragma unroll 1
    for (int t = set; t < ntiles; t += 2) {
        // row 0's weights are the YOUNGEST loads in flight here (requested at row 8 of the previous tile, after row 1's): taking them
        // is an s_waitcnt vmcnt(0), so it comes before this half-slot's DMA pieces go out, not behind them
        take_row(raw[0]);
#pragma unroll
        for (int k = 0; k < R; ++k) asm volatile("" : "+v"(w.p0[k]), "+v"(w.p1[k]));      // (pinned here: hipcc sinks the scaling, and with it the wait, behind the DMA)
        asm volatile("" : "+v"(w.s0), "+v"(w.s1));
        __builtin_amdgcn_sched_barrier(0);
        slot_dma(t);
        if constexpr (!(ABL & 4)) request_row(raw[0], rowi(AH), t);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < WK_CH; ++c) { acc[c][0] = v2f{0.f, 0.f}; acc[c][1] = v2f{0.f, 0.f}; }
        channels(t, rowi(0), I0{}, I16{});
        // (the rows sit in opaque loops whose body is one period of the AH weight buffers: straight-line, hipcc pools the rows of a
        // half-slot and spills ~600 registers; with a loop body that is not a whole period its s_waitcnt pass drains every row to vmcnt(0))
        if constexpr (AH == 2) {
            full_row(t, rowi(1), I1{});
#pragma unroll 1
            for (int i0 = rowi(2); i0 < 4; i0 += 2) { full_row(t, i0, I0{}); full_row(t, i0 + 1, I1{}); }
            { const int i4 = rowi(4); begin_row(t, i4, I0{}); channels(t, i4, I0{}, I8{}); }
        } else {
#pragma unroll 1
            for (int i0 = rowi(1); i0 < 4; i0 += 3) { full_row(t, i0, I1{}); full_row(t, i0 + 1, I2{}); full_row(t, i0 + 2, I0{}); }
            { const int i4 = rowi(4); begin_row(t, i4, I1{}); channels(t, i4, I0{}, I8{}); }
        }
        vm_wait<AH * WK_DD>();                        // everything older than the AH rows of weights in flight: this half-slot's DMA pieces
        slot_end();

        slot_dma(t + 1);
        channels(t, rowi(4), I8{}, I16{});
        if constexpr (AH == 2) {
            full_row(t, rowi(5), I1{});
#pragma unroll 1
            for (int i0 = rowi(6); i0 < 8; i0 += 2) { full_row(t, i0, I0{}); full_row(t, i0 + 1, I1{}); }
            full_row(t, rowi(8), I0{});
        } else {
            full_row(t, rowi(5), I2{});
#pragma unroll 1
            for (int i0 = rowi(6); i0 < 9; i0 += 3) { full_row(t, i0, I0{}); full_row(t, i0 + 1, I1{}); full_row(t, i0 + 2, I2{}); }
        }
        vm_wait<AH * WK_DD>();                        // (before the stores: loads return in order among themselves only)
        const int sid = fresh_lane();
        const int py = t * 8 + wave_in_set * 2 + (sid >> 5), px = x0 + (sid & 31) * 2;
        // the sums are pinned in front of the store branches: hipcc otherwise sinks the last row's FMAs into the per-channel store
        // blocks and spills the LDS rows they read (scratch reloads = s_waitcnt vmcnt(0) in front of every barrier)
        float2 o[WK_CH];
#pragma unroll
        for (int c = 0; c < WK_CH; ++c) {
            o[c] = make_float2(acc[c][0].x + acc[c][0].y, acc[c][1].x + acc[c][1].y);
            asm volatile("" : "+v"(o[c].x), "+v"(o[c].y));
        }
        // 16 stores per lane, always (a lane outside the image or a channel past the group's last is dropped by the range check): behind
        // a branch their number is unknown to hipcc's s_waitcnt pass, and the first weights of the next tile then wait for vmcnt(0)
        const unsigned ooff = (py < H && px < W) ? (unsigned)(py * W + px) * 4u : kOut;
#pragma unroll
        for (int c = 0; c < WK_CH; ++c)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o[c]), ors, (int)(ooff + (unsigned)c * plane * 4u), 0, 0);
        slot_end();
        slots_done += 2;
    }
#pragma unroll 1
    for (int k = slots_done; k <= ntiles; ++k) {      // idle half-slots at the bottom: the other set is still working
        slot_dma(k);
        vm_wait<0>();
        __builtin_amdgcn_s_barrier();
    }
}

template <int ABL = 0, int AH = 2>
int launch_bwd_walker(const float* f1, const float* f2, const float* g, float* gf1, float* gf2, int B, int C, int H, int W, hipStream_t s) {
    const int strips = ceil_div(W, 64), ngrp = ceil_div(C, WK_CH);
    hipLaunchKernelGGL((corr_bwd_walker_kernel<ABL, AH>), dim3(strips * B * 2 * ngrp), dim3(512), 0, s, f1, f2, g, gf1, gf2, C, H, W, strips, ngrp, 1.0f / C);
    return (int)hipGetLastError();
}

}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float vv(size_t i) { return (float)((i * 2654435761ull) % 2001ull) / 1000.f - 1.f; }

int main(int argc, char** argv) {
    int B = 16, C = 32, H = 64, W = 208, iters = 50;
    if (argc >= 5) { B = atoi(argv[1]); C = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); }
    if (argc >= 6) iters = atoi(argv[5]);
    const int d = 4, DD = 81;
    const size_t nf = (size_t)B * C * H * W, nc = (size_t)B * DD * H * W, nmax = nf > nc ? nf : nc;
    std::vector<float> h(nmax + 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = vv(i);
    float *f1, *f2, *g, *out[4][2];
    CK(hipMalloc(&f1, nf * 4)); CK(hipMalloc(&f2, nf * 4)); CK(hipMalloc(&g, nc * 4));
    for (int v = 0; v < 4; ++v) for (int k = 0; k < 2; ++k) { CK(hipMalloc(&out[v][k], nf * 4)); CK(hipMemset(out[v][k], 0xff, nf * 4)); }
    CK(hipMemcpy(f1, h.data(), nf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(f2, h.data() + 7, nf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, h.data() + 3, nc * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"shipped entry (unflow_corr_bwd)", "row-streamed, 64-wide tiles only", "column walker, 2 rows in flight", "column walker, 3 rows in flight"};
    auto run = [&](int v) -> int {
        if (v == 0) return unflow_corr_bwd(f1, f2, g, out[0][0], out[0][1], B, C, H, W, d, s);
        if (v == 1) return launch_bwd_rs<4, 16, 8, 2>(f1, f2, g, out[1][0], out[1][1], B, C, H, W, s);
        if (v == 2) return launch_bwd_walker<0, 2>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
        return launch_bwd_walker<0, 3>(f1, f2, g, out[3][0], out[3][1], B, C, H, W, s);
    };
    for (int round = 0; round < 2; ++round)
        for (int v = 0; v < 4; ++v) {
            for (int i = 0; i < 3; ++i) { int rc = run(v); if (rc) { fprintf(stderr, "%s: launch failed %d\n", names[v], rc); return 1; } }
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) run(v);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d  %-36s %.2f us/launch\n", round, names[v], ms * 1e3 / iters);
        }
    std::vector<float> ref(nf), got(nf);
    for (int v = 1; v < 4; ++v)
        for (int k = 0; k < 2; ++k) {
            CK(hipMemcpy(ref.data(), out[0][k], nf * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(got.data(), out[v][k], nf * 4, hipMemcpyDeviceToHost));
            size_t bad = 0, first_bad = 0; double worst = 0;
            for (size_t i = 0; i < nf; ++i) {
                if (memcmp(&ref[i], &got[i], 4) != 0) {
                    if (!bad) first_bad = i;
                    ++bad;
                    const double e = fabs((double)ref[i] - (double)got[i]);
                    if (!(e <= worst)) worst = e;          // (NaN sticks)
                }
            }
            printf("%s gf%d vs shipped: %zu of %zu values differ in bits, worst |diff| %.3g", names[v], k + 1, bad, nf, worst);
            if (bad) {
                const size_t x = first_bad % W, y = (first_bad / W) % H, c = (first_bad / ((size_t)W * H)) % C, b = first_bad / ((size_t)W * H * C);
                printf(" (first at b %zu c %zu y %zu x %zu: %.9g vs %.9g)", b, c, y, x, got[first_bad], ref[first_bad]);
            }
            printf("\n");
        }
    // where the time goes (wrong results by construction)
    const char* an[8] = {"no barrier in the tile loop", "no DMA in the tile loop", "no weight requests in the tile loop", "rows empty (no LDS reads, no FMAs)",
                         "rows empty + no weight requests", "rows empty + no requests + no DMA",
                         "3 rows in flight: no weight requests", "3 rows in flight: rows empty"};
    for (int v = 0; v < 8; ++v) {
        auto ab = [&]() -> int {
            switch (v) {
                case 0: return launch_bwd_walker<1>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
                case 1: return launch_bwd_walker<2>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
                case 2: return launch_bwd_walker<4>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
                case 3: return launch_bwd_walker<8>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
                case 4: return launch_bwd_walker<12>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
                case 5: return launch_bwd_walker<14>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
                case 6: return launch_bwd_walker<4, 3>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
                default: return launch_bwd_walker<8, 3>(f1, f2, g, out[2][0], out[2][1], B, C, H, W, s);
            }
        };
        for (int i = 0; i < 3; ++i) ab();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) ab();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("ablation: %-40s %.2f us/launch\n", an[v], ms * 1e3 / iters);
    }
    return 0;
}
