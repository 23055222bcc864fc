// PROTOTYPE, NEVER RUN ON A GPU (round 6 moved it out of libunflow_hip.so and its C ABI; tools/proto/corr_bwd_mfma.hip is its harness): the matrix-core cost-volume backward of csrc/corr_mfma.h with TWO PIXELS PER LANE
// in the upstream-gradient stage.  What the counters of the shipped form say (profiles/r5_corr_bwd_mfma.md): at d = 4 the
// vector-memory path carries 27 dword-load instructions per source-row step and wave for 81 x 16 gradients -- four 64-byte pieces per
// instruction -- and a second request set in flight does not help: throughput, not latency.  Here a lane is (pixel PAIR p, displacement
// row 8 rho + q8): one 8-byte load per displacement fetches a pair, one round covers EIGHT displacement rows (2R + 1 = 8 NR8 + 1 for
// R = 4 and 8), and the one row left over (i = 2R) is loaded with the displacements spread over the lane's upper three bits:
// 9 + 2 = 11 load instructions per step at d = 4 (27), 34 + 3 = 37 at d = 8 (85), each moving 128-byte runs.  Everything else is the
// shipped kernel: banded 16 x 32 A tables in per-wave LDS (eight of them per part now: 16.9 KB per wave at a 64-byte row stride),
// B operand from global memory in the MFMA layout, accumulator slots that slide through C != D, no workgroup barrier.
// Written after the GPU lease closed in round 5: it has NOT run on a GPU; compiled for the host and executed lane by lane it holds the oracle at the
// GPU test's bar (tests/test_kernels_on_host.py::test_matrix_core_backward_runs_on_the_host, also under ASan / UBSan).
#pragma once
#include "corr_mfma.h"

namespace {

template <int R>
struct BwdMf2Cfg {
    static_assert(R == 4 || R == 8, "2R + 1 = 8 NR8 + 1");
    static constexpr int DD = 2 * R + 1, SH = 8 - R, NR8 = (DD - 1) / 8;
    static constexpr int RSB = 64, TAB = 16 * RSB + 32;     // (2-way conflicts on the fragment reads; 96-byte rows would cost 25.6 KB per wave)
    static constexpr int WAVE_LDS = 2 * 8 * TAB;            // hi and lo parts of eight displacement rows
};

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

template <int R, int NCG, int MODE, int SKIP>
__device__ __forceinline__ void corr_bwd_mf2_body(lds_byte* __restrict__ tab, const float* __restrict__ F, const float* __restrict__ g,
                                                  float* __restrict__ out, int b, int c_begin, int S, int ya, int ybp,
                                                  int Ctot, int H, int W, float inv_c) {
    using K = BwdMf2Cfg<R>;
    constexpr int DD = K::DD, SH = K::SH, NR8 = K::NR8, TAB = K::TAB, RSB = K::RSB, NB = NR8 + 1;
    constexpr unsigned kOut = 0x40000000u;
    const int lane = (int)(threadIdx.x & 63);
    const int xl = lane & 15, q = lane >> 4;               // A-read / feature / store roles: as in corr_mfma.h
    const int p = lane & 7, q8 = lane >> 3;                // gradient role: pixel pair p (pixels 2p, 2p + 1), displacement row 8 rho + q8 (round B: displacement q8 + 8 t)
    const unsigned plane = (unsigned)(H * W);
    const int C = min(NCG * 16, Ctot - c_begin);

    const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g + (size_t)b * DD * DD * plane), 0, (int)((size_t)DD * DD * plane * 4), 0x00020000);
    const auto frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(F + ((size_t)b * Ctot + c_begin) * plane), 0, (int)((size_t)C * plane * 4), 0x00020000);
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(out + ((size_t)b * Ctot + c_begin) * plane, 0, (int)((size_t)C * plane * 4), 0x00020000);

    for (int o = lane * 16; o < K::WAVE_LDS; o += 64 * 16) *(__attribute__((address_space(3))) v4u_t*)(tab + o) = v4u_t{0u, 0u, 0u, 0u};
    UNFLOW_WAVE_LOCKSTEP();                                   // (corr_mfma.h: nothing on the device)

    const int r_begin = max(ya - R, 0), r_end = min(ybp - 1 + R, H - 1);
    const int xg0 = 16 * S + 2 * p;
    const bool x_ok = xg0 + 1 < W;
    const unsigned jstep = MODE ? (plane - 1u) * 4u : plane * 4u;
    // byte offset of (displacement row i, source row r, the pair's first pixel), displacement n = 0; advances by one row per step
    auto g_base = [&](int i, int r) -> unsigned {
        return MODE ? (((unsigned)((2 * R - i) * DD)) * plane + (unsigned)(r * W + xg0 + R)) * 4u
                    : (((unsigned)(i * DD)) * plane + (unsigned)((r + R - i) * W + xg0)) * 4u;
    };
    unsigned gbA[NR8], gbB = g_base(2 * R, r_begin) + (unsigned)q8 * jstep;      // (round B: the lane's own displacement q8 is in its offset)
#pragma unroll
    for (int rho = 0; rho < NR8; ++rho) gbA[rho] = g_base(8 * rho + q8, r_begin);
    auto a_request = [&](v2u_t (&raw)[DD], unsigned& base, int rho, int r) __attribute__((always_inline)) {
        const int y = r + R - (8 * rho + q8);
        const bool ok = x_ok & (y >= ya) & (y < ybp) & (r <= r_end);
        const unsigned vo = ok ? base : kOut;
#pragma unroll
        for (int n = 0; n < DD; ++n) raw[n] = __builtin_bit_cast(v2u_t, __builtin_amdgcn_raw_buffer_load_b64(grs, (int)vo, (int)(n * jstep), 0));
        base += (unsigned)(W * 4);
    };
    auto b_request = [&](v2u_t (&raw)[NB], unsigned& base, int r) __attribute__((always_inline)) {
        const int y = r - R;                                // displacement row 2R
        const bool ok = x_ok & (y >= ya) & (y < ybp) & (r <= r_end);
#pragma unroll
        for (int t = 0; t < NR8; ++t)                       // displacements q8 + 8 t
            raw[t] = __builtin_bit_cast(v2u_t, __builtin_amdgcn_raw_buffer_load_b64(grs, (int)(ok ? base : kOut), (int)(8 * t * jstep), 0));
        // displacement 2R = 8 NR8: the lanes with q8 == 0 (base already holds q8 jstep = 0 for them)
        raw[NR8] = __builtin_bit_cast(v2u_t, __builtin_amdgcn_raw_buffer_load_b64(grs, (int)((ok & (q8 == 0)) ? base : kOut), (int)(8 * NR8 * jstep), 0));
        base += (unsigned)(W * 4);
    };
    const int fx = 16 * S - 8 + 8 * q;
    const bool f_ok0 = (fx >= 0) & (fx + 3 < W), f_ok1 = (fx + 4 >= 0) & (fx + 7 < W);
    unsigned fo = ((unsigned)xl * plane + (unsigned)(r_begin * W + fx)) * 4u;
    auto f_request = [&](v4u_t (&raw)[NCG][2], int r) __attribute__((always_inline)) {
        const bool rok = r <= r_end;
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const bool cok = 16 * cg + xl < C;
            raw[cg][0] = __builtin_bit_cast(v4u_t, __builtin_amdgcn_raw_buffer_load_b128(frs, (int)((f_ok0 & rok & cok) ? fo : kOut), (int)(cg * 16 * plane * 4), 0));
            raw[cg][1] = __builtin_bit_cast(v4u_t, __builtin_amdgcn_raw_buffer_load_b128(frs, (int)((f_ok1 & rok & cok) ? fo + 16u : kOut), (int)(cg * 16 * plane * 4), 0));
        }
        fo += (unsigned)(W * 4);
    };

    v2u_t rawA[NR8][DD], rawB[NB];
    v4u_t fraw[NCG][2];
    __builtin_amdgcn_sched_barrier(0);
    f_request(fraw, r_begin);
    __builtin_amdgcn_sched_barrier(0);
    b_request(rawB, gbB, r_begin);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rho = NR8 - 1; rho >= 0; --rho) { a_request(rawA[rho], gbA[rho], rho, r_begin); __builtin_amdgcn_sched_barrier(0); }

    v4f acc[DD + 1][NCG];
#pragma unroll
    for (int s = 0; s <= DD; ++s)
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) acc[s][cg] = v4f{0.f, 0.f, 0.f, 0.f};

    typedef __attribute__((address_space(3))) unsigned short lds_u16;
    // cell (row 2p + e, column 2p + e + j + SH) of table t:  wrow[e] + t TAB + 2 j   (lo part: + 8 TAB)
    lds_byte* const wrow0 = tab + (2 * p) * RSB + 2 * (2 * p + SH);
    lds_byte* const wrow1 = wrow0 + RSB + 2;
    lds_byte* const rbase = tab + xl * RSB + 16 * q;
    const unsigned so = ((unsigned)xl * plane + (unsigned)(16 * S + 4 * q)) * 4u;
    const bool s_ok = (16 * S + 4 * q + 3 < W);
    auto store_row = [&](const v4f (&a)[NCG], int y, bool wanted) __attribute__((always_inline)) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const bool ok = s_ok & wanted & (16 * cg + xl < C);
            const v4f v = a[cg] * inv_c;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), ors,
                                                   (int)(ok ? so + (unsigned)(y * W * 4) + (unsigned)cg * 16u * plane * 4u : kOut), 0, 0);
        }
    };
    const unsigned span = (unsigned)(ybp - ya);
    // one displacement row of a pair of pixels -> its two table rows (both parts): x components to row 2p, y components to row 2p + 1
    auto write_pair = [&](lds_byte* w0, lds_byte* w1, float ax, float bx, float ay, float by, int ja, int jb, bool both) __attribute__((always_inline)) {
        unsigned hi, lo;
        mf_split2(ax, bx, hi, lo);
        *(lds_u16*)(w0 + 2 * ja) = (unsigned short)hi;
        *(lds_u16*)(w0 + 8 * TAB + 2 * ja) = (unsigned short)lo;
        if (both) { *(lds_u16*)(w0 + 2 * jb) = (unsigned short)(hi >> 16); *(lds_u16*)(w0 + 8 * TAB + 2 * jb) = (unsigned short)(lo >> 16); }
        mf_split2(ay, by, hi, lo);
        *(lds_u16*)(w1 + 2 * ja) = (unsigned short)hi;
        *(lds_u16*)(w1 + 8 * TAB + 2 * ja) = (unsigned short)lo;
        if (both) { *(lds_u16*)(w1 + 2 * jb) = (unsigned short)(hi >> 16); *(lds_u16*)(w1 + 8 * TAB + 2 * jb) = (unsigned short)(lo >> 16); }
    };
    auto unit = [&](int r, int i, int t, const v4u_t (&bh)[NCG], const v4u_t (&bl)[NCG], bool round_on) __attribute__((always_inline)) {
        if (SKIP == 0 || (SKIP == 2 ? round_on : (unsigned)(r + R - i - ya) < span)) {
            const v4u_t ah = *(__attribute__((address_space(3))) const v4u_t*)(rbase + t * TAB);
            const v4u_t al = *(__attribute__((address_space(3))) const v4u_t*)(rbase + 8 * TAB + t * TAB);
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                v4f a = i ? acc[i][cg] : v4f{0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[cg]), a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[cg]), a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[cg]), a, 0, 0, 0);
                acc[i + 1][cg] = a;
            }
        }
    };

#pragma unroll 1
    for (int r = r_begin; r <= r_end; ++r) {
        v4u_t bh[NCG], bl[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const v4u_t w = fraw[cg][e >> 1];
                unsigned hi, lo;
                mf_split2(__uint_as_float(w[2 * (e & 1)]), __uint_as_float(w[2 * (e & 1) + 1]), hi, lo);
                bh[cg][e] = hi; bl[cg][e] = lo;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f_request(fraw, r + 1);
        __builtin_amdgcn_sched_barrier(0);
        // round B: displacement row 2R into table 0 -- a lane holds displacements q8 + 8 t (and 2R where q8 == 0) of its pixel pair
        {
            const bool on = !SKIP || ((unsigned)(r - R - ya) < span);
            if (on) {
#pragma unroll
                for (int t = 0; t < NR8; ++t) {
                    const int n = q8 + 8 * t, j = MODE ? 2 * R - n : n;
                    write_pair(wrow0, wrow1, __uint_as_float(rawB[t].x), 0.f, __uint_as_float(rawB[t].y), 0.f, j, 0, false);
                }
                if (q8 == 0) write_pair(wrow0, wrow1, __uint_as_float(rawB[NR8].x), 0.f, __uint_as_float(rawB[NR8].y), 0.f, MODE ? 0 : 2 * R, 0, false);
            }
            UNFLOW_WAVE_LOCKSTEP();
            __builtin_amdgcn_sched_barrier(0);
            b_request(rawB, gbB, r + 1);
            __builtin_amdgcn_sched_barrier(0);
            unit(r, 2 * R, 0, bh, bl, on);
        }
#pragma unroll
        for (int rho = NR8 - 1; rho >= 0; --rho) {
            const bool on = !SKIP || ((r + R - 8 * rho >= ya) & (r + R - (8 * rho + 7) < ybp));
            if (on) {
                lds_byte* const w0 = wrow0 + q8 * TAB;
                lds_byte* const w1 = wrow1 + q8 * TAB;
#pragma unroll
                for (int n = 0; n < DD; n += 2) {
                    const int j0 = MODE ? 2 * R - n : n, j1 = MODE ? 2 * R - n - 1 : n + 1;
                    const bool both = n + 1 < DD;
                    write_pair(w0, w1, __uint_as_float(rawA[rho][n].x), both ? __uint_as_float(rawA[rho][n + 1 < DD ? n + 1 : n].x) : 0.f,
                               __uint_as_float(rawA[rho][n].y), both ? __uint_as_float(rawA[rho][n + 1 < DD ? n + 1 : n].y) : 0.f, j0, j1, both);
                }
            }
            UNFLOW_WAVE_LOCKSTEP();
            __builtin_amdgcn_sched_barrier(0);
            a_request(rawA[rho], gbA[rho], rho, r + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 7; t >= 0; --t) unit(r, 8 * rho + t, t, bh, bl, on);
        }
        store_row(acc[DD], r - R, (r - R >= ya) & (r - R < ybp));
    }
#pragma unroll
    for (int s = 1; s <= 2 * R; ++s) {
        const int y = r_end + R + 1 - s;
        store_row(acc[s], y, (y < ybp) & (y >= ya));
    }
}

template <int R, int NCG, int SKIP>
__global__ __launch_bounds__(256, 1) void corr_bwd_mf2_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              const float* __restrict__ g, float* __restrict__ gf1,
                                                              float* __restrict__ gf2, int Ctot, int H, int W,
                                                              int nseg, int nsb, int rows, int nchunk, int ngrp, float inv_c) {
    using K = BwdMf2Cfg<R>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * K::WAVE_LDS];
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = t % ngrp; t /= ngrp;
    const int mode = t & 1; t >>= 1;
    const int sb = t % nsb; t /= nsb;
    const int chunk = t % nchunk;
    const int b = t / nchunk;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int S = sb * 4 + wave;
    if (S >= nseg) return;
    const int ya = chunk * rows, ybp = min(ya + rows, H);
    lds_byte* tab = (lds_byte*)lds + wave * K::WAVE_LDS;
    if (mode) corr_bwd_mf2_body<R, NCG, 1, SKIP>(tab, f1, g, gf2, b, cg * NCG * 16, S, ya, ybp, Ctot, H, W, inv_c);
    else corr_bwd_mf2_body<R, NCG, 0, SKIP>(tab, f2, g, gf1, b, cg * NCG * 16, S, ya, ybp, Ctot, H, W, inv_c);
}

template <int R, int NCG, int SKIP = 1>
int launch_bwd_mf2(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
                   int B, int C, int H, int W, int rows, hipStream_t s) {
    const int nseg = ceil_div(W, 16), nsb = ceil_div(nseg, 4), nchunk = ceil_div(H, rows), ngrp = ceil_div(C, NCG * 16);
    UNFLOW_LAUNCH((corr_bwd_mf2_kernel<R, NCG, SKIP>), dim3(nsb * nchunk * B * 2 * ngrp), dim3(256), 0, s,
                  f1, f2, g, gf1, gf2, C, H, W, nseg, nsb, rows, nchunk, ngrp, 1.0f / C);
    return unflow_launch_status();
}

}  // namespace
