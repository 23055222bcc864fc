// Fused flow-warp + cost volume for gfx950.
//
// Replaces the pair  cv = corr(f1, warp(f2, flow))  of every decoder level of the reference
// (core/networks/structures/pwc_tf.py:121-122, 134-135, 146-147, 159-160: `feat2_warped = self.warp(feature, flow)`
// followed by `self.corr(feat1, feat2_warped)`): the warped feature map is never written to HBM.  A workgroup owns a
// 64x8 pixel tile and DG displacement rows, exactly like the LDS-DMA ring kernel of corr.hip, but what streams through
// the ring is the SOURCE WINDOW of the tile's halo (the bounding box of the bilinear taps of its (8+DG-1) x 72 halo
// positions, zero cells outside the image), CC channels per stage.  Per stage: the 256 lanes bilinear-sample the
// halo positions out of the window (LDS reads) into a warped tile in LDS, then the cost-volume rows are read from
// that tile with the same software-pipelined ds_read_b64 stream as the unfused kernel.  The tap set-up is the
// bit-exact one of warp.hip (warp_taps.h); window cells outside the image are zero, which is grid_sample's zero
// padding.  A tile whose flow spreads its taps beyond the window capacity gathers from global memory instead (per
// tile, always correct).
//
// Backward (unflow_warp_corr_bwd): gf1 needs the warped f2 -- it is recomputed in LDS by the same warp stage in front
// of the group-split backward pipeline; the gradient w.r.t. the warped map is an ordinary cost-volume backward (it
// reads f1 and gcv only) written to caller scratch, and goes through the LDS-tile warp backward of warp.hip.
#include "common.h"
#include "corr_ring.h"
#include "warp_taps.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

template <int R, int CC, int DG, int NS_, int SWX_, int SWH_>
struct FusedCfg {
    static constexpr int DD = 2 * R + 1;
    static constexpr int NG = (DD + DG - 1) / DG;
    static constexpr int TW = 64, TYB = 8, NS = NS_;
    static constexpr int LW = TW + 2 * R, LH = TYB + DG - 1;                     // halo tile of warped positions
    static constexpr int NPOS = (LH * LW + 255) / 256;                           // halo positions per lane
    static constexpr int SWX = SWX_, SWH = SWH_;                                 // source window capacity (floats x rows)
    static constexpr int S2 = SWH * SWX / 4, S1 = TYB * TW / 4, SC = S2 + S1;    // float4 slots per channel: window | f1 tile
    static constexpr int ITER = (CC * SC + 255) / 256;
    static constexpr int STAGE = ITER * 256 * 4;                                 // floats per ring slot
    static constexpr int WT = CC * LH * LW;                                      // floats of the warped tile
    static constexpr int WAVES = 2;                                              // (134 VGPRs with 3 displacement rows, 256 with 9)
    static_assert(SWX % 4 == 0 && SWX + 1 <= 255, "ds_read2_b32 offsets are 8-bit dword counts");
    static_assert(SWX >= LW + 4 && SWH >= LH + 1, "the window must hold the halo plus one tap row / column and the alignment slack");
};

// What a lane keeps per halo position: the byte offset of its nw tap inside a channel's window and the two
// fractional weights; bit q of `live` = the position samples at least one tap inside the image.
template <int NPOS>
struct HaloTaps {
    unsigned base[NPOS];
    float wx[NPOS], wy[NPOS];
    unsigned live;
};

// Tap set-up of halo position p of the tile (recomputed by the rare fallback path instead of being kept in registers).
template <int R, int LW, int LH>
__device__ __forceinline__ bool halo_taps(Taps& tp, const float* __restrict__ fl, int p, int x0t, int y0t, int i0,
                                          int H, int W, int ac) {
    const int ly = p / LW, lx = p - ly * LW;
    const int gx = x0t - R + lx, gy = y0t - R + i0 + ly;
    const bool inimg = p < LH * LW && gx >= 0 && gx < W && gy >= 0 && gy < H;      // the warped map is zero outside the image
    const int sx = inimg ? gx : 0, sy = inimg ? gy : 0;
    const float u = inimg ? fl[sy * W + sx] : 0.f;
    const float v = inimg ? fl[H * W + sy * W + sx] : 0.f;
    tp = make_taps(u, v, sx, sy, H, W, ac);
    return inimg && (tp.v_nw || tp.v_ne || tp.v_sw || tp.v_se);
}

template <int R, int CC, int DG, int NS, int SWX, int SWH>
__global__ __launch_bounds__(256, (FusedCfg<R, CC, DG, NS, SWX, SWH>::WAVES)) void warp_corr_fwd_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ flow, float* __restrict__ cv,
    int C, int H, int W, int tiles_x, int tiles_y, float inv_c, int ac) {
    using K = FusedCfg<R, CC, DG, NS, SWX, SWH>;
    constexpr int DD = K::DD, LW = K::LW, LH = K::LH, NROW = 2 + 2 * R, NPOS = K::NPOS;
    __shared__ __attribute__((aligned(16))) float lds[K::NS * K::STAGE + K::WT];
    float* ring = lds;
    float* wt = lds + K::NS * K::STAGE;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int b = t / tiles_y;
    const int i0 = blockIdx.y * DG;                          // first displacement row of this workgroup
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0t = bx * K::TW, y0t = by * K::TYB;
    const int px = x0t + tx * 2, py = y0t + ty;
    const int plane_i = H * W;
    const size_t plane = (size_t)plane_i;
    const int nchunk = (C + CC - 1) / CC;

    // ---- bilinear taps of this lane's halo positions, and the bounding box of all taps of the tile
    HaloTaps<NPOS> hp;
    hp.live = 0u;
    int hx0[NPOS], hy0[NPOS];
    int bx0 = 0x3fffffff, bx1 = -0x3fffffff, by0 = 0x3fffffff, by1 = -0x3fffffff;
    const float* fl = flow + (size_t)b * 2 * plane;
#pragma unroll
    for (int q = 0; q < NPOS; ++q) {
        Taps tp;
        const bool any = halo_taps<R, LW, LH>(tp, fl, q * 256 + (int)threadIdx.x, x0t, y0t, i0, H, W, ac);
        hx0[q] = tp.x0; hy0[q] = tp.y0;
        hp.wx[q] = tp.w; hp.wy[q] = tp.n;
        if (any) {                                           // x0 in [-1, W-1], y0 in [-1, H-1]
            hp.live |= 1u << q;
            bx0 = min(bx0, tp.x0); bx1 = max(bx1, tp.x0 + 1);
            by0 = min(by0, tp.y0); by1 = max(by1, tp.y0 + 1);
        }
    }
    int* s_box = reinterpret_cast<int*>(wt);
    bx0 = wave_min_i32(bx0); by0 = wave_min_i32(by0); bx1 = wave_max_i32(bx1); by1 = wave_max_i32(by1);
    if (lane == 0) { s_box[wave * 4 + 0] = bx0; s_box[wave * 4 + 1] = by0; s_box[wave * 4 + 2] = bx1; s_box[wave * 4 + 3] = by1; }
    __syncthreads();
    bx0 = min(min(s_box[0], s_box[4]), min(s_box[8], s_box[12]));
    by0 = min(min(s_box[1], s_box[5]), min(s_box[9], s_box[13]));
    bx1 = max(max(s_box[2], s_box[6]), max(s_box[10], s_box[14]));
    by1 = max(max(s_box[3], s_box[7]), max(s_box[11], s_box[15]));
    __syncthreads();                                         // (wt is written again in the first stage)
    const bool empty = bx1 < bx0;                            // no tap of the tile inside the image: the cost volume is zero here
    const int wxa = empty ? 0 : (bx0 & ~3);                  // window origin, x rounded down to a float4 boundary (may be -4)
    const int wy0 = empty ? 0 : by0;
    const bool fits = empty || (bx1 - wxa + 1 <= K::SWX && by1 - wy0 + 1 <= K::SWH);

#pragma unroll
    for (int q = 0; q < NPOS; ++q)
        hp.base[q] = ((hp.live >> q) & 1u) && fits ? (unsigned)(((hy0[q] - wy0) * K::SWX + (hx0[q] - wxa)) * 4) : 0u;

    // ---- per-lane DMA slot descriptors (loop invariant): plane offset or -1 (zero line)
    int soff[K::ITER];
#pragma unroll
    for (int it = 0; it < K::ITER; ++it) {
        const int s = it * 256 + (int)threadIdx.x;
        const int c = s / K::SC;
        int r = s - c * K::SC;
        int gy, gx;
        bool win = r < K::S2;
        if (win) { const int ly = r / (K::SWX / 4); gy = wy0 + ly; gx = wxa + (r - ly * (K::SWX / 4)) * 4; }
        else { r -= K::S2; const int ly = r / (K::TW / 4); gy = y0t + ly; gx = x0t + (r - ly * (K::TW / 4)) * 4; }
        const bool in = (c < CC) && gy >= 0 && gy < H && gx >= 0 && gx < W && (!win || (fits && !empty));
        soff[it] = in ? gy * W + gx : -1;
        UNFLOW_PIN_VGPR(soff[it]);                   // materialise once; the channel / source are re-derived per stage
    }
    const float* base1 = f1 + (size_t)b * C * plane;
    const float* base2 = f2 + (size_t)b * C * plane;
    gfloat* zline = zero_line();
    auto issue = [&](int stage_idx) {
        float* dst = ring + (stage_idx % K::NS) * K::STAGE;
        const int c0 = stage_idx * CC;
#pragma unroll
        for (int it = 0; it < K::ITER; ++it) {
            const int s = it * 256 + (int)threadIdx.x;
            const int c = s / K::SC;                         // (constant divisor: a multiply and a shift)
            const bool win = s - c * K::SC < K::S2;
            const int gc = c0 + c;
            const bool in = soff[it] >= 0 && gc < C;
            gfloat* g = in ? (gfloat*)((win ? base2 : base1) + (size_t)gc * plane + soff[it]) : zline;
            __builtin_amdgcn_global_load_lds((gas_ptr)g, (lds_ptr)(dst + (it * 256 + wave * 64) * 4), 16, 0, 0);
        }
    };

    FwdAcc<DG, R> acc;
    acc.zero();

    const unsigned ring_addr = (unsigned)(size_t)(lds_cfloat*)ring;
    const unsigned wt_addr = (unsigned)(size_t)(lds_cfloat*)wt;
    const unsigned rows_addr = wt_addr + (unsigned)(ty * LW + tx * 2) * 4u;
    const unsigned own_addr = ring_addr + (unsigned)(K::S2 * 4 + ty * K::TW + tx * 2) * 4u;

#pragma unroll
    for (int st = 0; st < K::NS - 1; ++st) issue(st);        // stages beyond nchunk read the zero line

    for (int k = 0; k < nchunk; ++k) {
        vm_wait<K::ITER * (K::NS - 2)>();                    // stage k has landed for this wave ...
        __builtin_amdgcn_s_barrier();                        // ... and for every wave; the previous stage's tiles are free
        issue(k + K::NS - 1);
        const unsigned sbytes = (unsigned)((k % K::NS) * K::STAGE) * 4u;

        // ---- warp stage: window (ring slot k) -> warped halo tile
        if (fits) {
            // positions in groups of PG: PG * CC * 2 tap reads in flight, then the blends (all positions at once when
            // the accumulators leave room, two at a time with all 81 displacements per lane)
            constexpr int PG = (DG * DD * 2 <= 96) ? NPOS : 2;
#pragma unroll
            for (int q0 = 0; q0 < NPOS; q0 += PG) {
                v2f top[PG][CC], bot[PG][CC];
#pragma unroll
                for (int u = 0; u < PG; ++u)
#pragma unroll
                    for (int c = 0; c < CC; ++c)
                        if (q0 + u < NPOS) {
                            const unsigned a = ring_addr + sbytes + (unsigned)(c * K::SC * 16) + hp.base[q0 + u];
                            top[u][c] = lds_read2_b32<0, 1>(a);
                            bot[u][c] = lds_read2_b32<K::SWX, K::SWX + 1>(a);
                        }
                lds_wait<0>();
#pragma unroll
                for (int u = 0; u < PG; ++u) {
                    if (q0 + u >= NPOS) continue;
                    const int q = q0 + u;
                    const int p = q * 256 + (int)threadIdx.x;
                    const float e = 1.0f - hp.wx[q], s = 1.0f - hp.wy[q];
                    const float nw = s * e, ne = s * hp.wx[q], sw = hp.wy[q] * e, se = hp.wy[q] * hp.wx[q];
#pragma unroll
                    for (int c = 0; c < CC; ++c) {
                        // same weights and accumulation order as warp.hip / ATen: nw, ne, sw, se
                        float r = top[u][c].x * nw;
                        r = fmaf(top[u][c].y, ne, r);
                        r = fmaf(bot[u][c].x, sw, r);
                        r = fmaf(bot[u][c].y, se, r);
                        r = ((hp.live >> q) & 1u) ? r : 0.f;
                        if (p < LH * LW) lds_write_b32(wt_addr + (unsigned)(c * LH * LW + p) * 4u, r);
                    }
                }
            }
        } else {
            // the flow spreads this tile's taps beyond the window: per-tap gathers from global memory (tap set-up
            // recomputed from the flow every stage: the rare path pays, the common one keeps its registers)
#pragma unroll 1
            for (int q = 0; q < NPOS; ++q) {
                const int p = q * 256 + (int)threadIdx.x;
                Taps tp;
                const bool any = halo_taps<R, LW, LH>(tp, fl, p, x0t, y0t, i0, H, W, ac);
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    const int gc = k * CC + c;
                    float r = 0.f;
                    if (any && gc < C) {
                        const float* pl = base2 + (size_t)gc * plane;
                        r = pl[tp.o_nw] * tp.nw;
                        r = fmaf(pl[tp.o_ne], tp.ne, r);
                        r = fmaf(pl[tp.o_sw], tp.sw, r);
                        r = fmaf(pl[tp.o_se], tp.se, r);
                    }
                    if (p < LH * LW) lds_write_b32(wt_addr + (unsigned)(c * LH * LW + p) * 4u, r);
                }
            }
        }
        UNFLOW_WAIT_LGKMCNT0();   // this wave's part of the warped tile is written ...
        __builtin_amdgcn_s_barrier();                        // ... and everyone else's

        // ---- cost-volume stage: CC*DG row-steps as one software pipeline over the warped tile
        constexpr int NCOL = NROW / 2, STEPS = CC * DG;
        constexpr int PF = (2 * NCOL <= 15 && DG * DD * 2 <= 96) ? 2 : 1;     // all 81 displacements per lane: one row ahead (registers)
        v2f a[CC];
        own_reads<K::SC * 16>(a, own_addr + sbytes, std::make_integer_sequence<int, CC>{});
        v2f row[PF + 1][NCOL];
        using Step0 = FwdStep<0, STEPS, PF, DG, DD, NCOL, LH * LW * 4, LW * 4>;
        Step0::template load<0>(row, rows_addr);
        if constexpr (PF > 1) Step0::template load<1>(row, rows_addr);
        Step0::template run<CC>(acc, row, a, rows_addr);
    }
    vm_wait<0>();                                            // drain the zero-line tail loads

    if (py >= H || px >= W) return;
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

template <int R, int CC, int DG, int NS, int SWX, int SWH>
int launch_fused_fwd(const float* f1, const float* f2, const float* flow, float* cv, int B, int C, int H, int W, int ac,
                     hipStream_t s) {
    using K = FusedCfg<R, CC, DG, NS, SWX, SWH>;
    const int tx = ceil_div(W, K::TW), ty = ceil_div(H, K::TYB);
    UNFLOW_LAUNCH((warp_corr_fwd_kernel<R, CC, DG, NS, SWX, SWH>), dim3(tx * ty * B, K::NG), dim3(256), 0, s,
                       f1, f2, flow, cv, C, H, W, tx, ty, 1.0f / C, ac);
    return unflow_launch_status();
}

}  // namespace

// 1 when the fused kernels cover this shape (d = 4, rows of whole float4s); otherwise the caller runs
// unflow_warp_fwd + unflow_corr_fwd.
extern "C" int unflow_warp_corr_supported(int C, int H, int W, int d) {
    return (d == 4 && C > 0 && H > 0 && W >= 8 && (W & 3) == 0) ? 1 : 0;
}

extern "C" int unflow_warp_corr_fwd(const float* f1, const float* f2, const float* flow, float* cv,
                                    int B, int C, int H, int W, int d, int align_corners, void* stream) {
    UNFLOW_REQUIRE(f1 && f2 && flow && cv && B > 0 && C > 0 && H > 0 && W > 0);
    UNFLOW_REQUIRE(unflow_warp_corr_supported(C, H, W, d) && ((((size_t)f1 | (size_t)f2) & 15) == 0));
    hipStream_t s = (hipStream_t)stream;
    const int ac = align_corners ? 1 : 0;
#ifdef UNFLOW_TUNING
    const char* e = getenv("UNFLOW_FUSED_DG");
    const int dg = e ? atoi(e) : 0;
#else
    const int dg = 0;
#endif
    // large maps (level 2): all 81 displacements per workgroup; smaller maps: 3 displacement rows per workgroup
    if (dg == 9 || (dg == 0 && W >= 96 && (long)B * H * W >= 131072))
        return launch_fused_fwd<4, 2, 9, 3, 84, 24>(f1, f2, flow, cv, B, C, H, W, ac, s);
    return launch_fused_fwd<4, 2, 3, 3, 96, 16>(f1, f2, flow, cv, B, C, H, W, ac, s);
}

// Backward of the fused operator.  Nothing of the forward pass is kept: the warped map is recomputed into caller
// scratch (it is needed by the f1 gradient), the gradient w.r.t. the warped map goes to the second half of the
// scratch and is scattered back through the LDS-tile warp backward.  scratch: 2 * B*C*H*W floats.
//   gf1 = dcv/df1,  gf2 = d/df2 (zeroed + accumulated; may be NULL),  gflow = d/dflow.
extern "C" int unflow_warp_corr_bwd(const float* f1, const float* f2, const float* flow, const float* gcv,
                                    float* gf1, float* gf2, float* gflow, float* scratch,
                                    int B, int C, int H, int W, int d, int align_corners, void* stream) {
    UNFLOW_REQUIRE(f1 && f2 && flow && gcv && gf1 && gflow && scratch && B > 0 && C > 0 && H > 0 && W > 0 && d >= 0);
    const size_t n = (size_t)B * C * H * W;
    float* warped = scratch;
    float* gw = scratch + n;
    int rc = unflow_warp_fwd(f2, flow, warped, nullptr, B, C, H, W, align_corners, stream);
    if (rc) return rc;
    rc = unflow_corr_bwd(f1, warped, gcv, gf1, gw, B, C, H, W, d, stream);
    if (rc) return rc;
    return unflow_warp_bwd(f2, flow, gw, nullptr, gf2, gflow, B, C, H, W, align_corners, stream);
}
