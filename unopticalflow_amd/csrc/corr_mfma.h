// Cost-volume backward on the matrix cores (round 5).  Autograd of PWC_tf.corr_naive, /root/reference
// core/networks/structures/pwc_tf.py:97-106:
//     gf1[c, y, x]   = 1/C  sum_{i, j} g[i (2R+1) + j, y, x]                 f2[c, y + i - R, x + j - R]
//     gf2[c, y', x'] = 1/C  sum_{i, j} g[i (2R+1) + j, y' - i + R, x' - j + R] f1[c, y' - i + R, x' - j + R]
// The fp32 kernels of corr.hip spend their time in packed fp32 FMAs (~75 TFLOP/s; profiles/r4_corr_bwd_walker.md).  Here the
// same sums run as banded matrix products on v_mfma_f32_16x16x32_bf16 with both operands split into bf16 hi + lo parts
// (hi = RNE(v), lo = RNE(v - hi); hi*hi + lo*hi + hi*lo, fp32 accumulation: ~1e-5 relative, no lo*lo term):
//   * a WAVE owns one 16-pixel segment of a chunk of rows of one (sample, gradient, group of NCG x 16 channels) and walks the
//     SOURCE rows r of F top to bottom.  Source row r feeds the output rows y = r + R - i, i = 0 .. 2R, with displacement row i:
//         out[y][x, c] += sum_k  Wt_i[x, k] F_r[k, c],     k = 0 .. 31  <->  source column 16 S - 8 + k,
//         Wt_i[x, k] = g[i, j][y][x] at k = xl + j + 8 - R (a band of 2R + 1 <= 17 of the 32 columns), 0 elsewhere
//     = one 16x16x32 product per (displacement row, 16 channels), three with the split.  The B operand (F_r, 32 source columns
//     x 16 channels) is loaded straight from global memory into the MFMA's lane layout (lane (c, g): 8 consecutive pixels of
//     channel c = 32 bytes), split once per source row and used by all 2R + 1 displacement rows: no feature tile in LDS, no
//     workgroup barrier -- the four waves of a workgroup never meet.
//   * the 2R + 1 output rows in flight live in accumulator slots; the MFMA writes its result to the NEXT slot
//     (acc[i + 1] = A B + acc[i]), so the window slides with the loop and the row that leaves slot 2R + 1 is complete.
//   * the banded A operand is built through a small per-wave LDS table: a lane (pixel xl, displacement row i = 4 rho + q)
//     loads its 2R + 1 upstream gradients with dword loads (64 contiguous bytes per 16 lanes; the displacement j is the
//     scalar offset of the load), splits them and writes each at column xl + j + 8 - R of row xl of table q -- the skew of
//     the band is in the write address; the zeros around the band are written once.  A lane (xl, g) then reads its fragment
//     (row xl, columns 8 g .. 8 g + 7) with one ds_read_b128 per part.
// For gf2 the lane of output pixel x' takes the weight of tap (i, j) from plane (2R - i, 2R - j) at the DISPLACED pixel
// (r, x' + j - R), as the fp32 kernels do; outside the image F is an exact 0, so those weights need no validity test.
// Results are deterministic (no atomics) but not bit-identical to the fp32 kernels.
#pragma once
#define UNFLOW_CORR_MFMA_INCLUDED 1
#include "corr_ring.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

template <int R>
struct BwdMfCfg {
    static_assert(R >= 1 && R <= 8, "the band xl + j + 8 - R must fit the 32 columns of one MFMA");
    static constexpr int DD = 2 * R + 1, SH = 8 - R, ROUNDS = (DD + 3) / 4;
    // a table = 16 rows (pixels) x 32 bf16 columns; row stride 96 bytes: the ds_read_b128 of the 16 lanes of a service group
    // fall into 64 different banks (64 or 80 bytes: 2-way conflicts); + 64 bytes between tables: the 2-byte writes of two
    // lane quarters (two tables) interleave without more than 2-way conflicts, which a ds_write does not pay for
    static constexpr int RSB = 96, TAB = 16 * RSB + 64;
    static constexpr int WAVE_LDS = 2 * 4 * TAB;           // hi and lo parts of four displacement rows
};

// (a, b) -> packed bf16 hi parts and packed bf16 lo parts (element 0 in the low half)
__device__ __forceinline__ void mf_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const bf16x2 h = {(__bf16)a, (__bf16)b};               // v_cvt_pk_bf16_f32: round to nearest even
    hi = __builtin_bit_cast(unsigned, h);
    const float ha = __uint_as_float(hi << 16), hb = __uint_as_float(hi & 0xffff0000u);
    const bf16x2 l = {(__bf16)(a - ha), (__bf16)(b - hb)};  // (the differences are exact in fp32)
    lo = __builtin_bit_cast(unsigned, l);
}

template <int R, int NCG, int MODE, int DEPTH, int SKIP>
__device__ __forceinline__ void corr_bwd_mf_body(lds_byte* __restrict__ tab, const float* __restrict__ F, const float* __restrict__ g,
                                                 float* __restrict__ out, int b, int c_begin, int S, int ya, int ybp,
                                                 int Ctot, int H, int W, float inv_c) {
    using K = BwdMfCfg<R>;
    constexpr int DD = K::DD, SH = K::SH, ROUNDS = K::ROUNDS, TAB = K::TAB, RSB = K::RSB;
    constexpr unsigned kOut = 0x40000000u;
    const int lane = (int)(threadIdx.x & 63);
    const int xl = lane & 15, q = lane >> 4;               // A-build role: pixel xl, displacement row 4 rho + q;  A-read role: row xl, columns 8 q ..
    const unsigned plane = (unsigned)(H * W);
    const int C = min(NCG * 16, Ctot - c_begin);

    const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g + (size_t)b * DD * DD * plane), 0, (int)((size_t)DD * DD * plane * 4), 0x00020000);
    const auto frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(F + ((size_t)b * Ctot + c_begin) * plane), 0, (int)((size_t)C * plane * 4), 0x00020000);
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(out + ((size_t)b * Ctot + c_begin) * plane, 0, (int)((size_t)C * plane * 4), 0x00020000);

    // zeros around the band, once (the band's own cells are rewritten by every round)
    for (int o = lane * 16; o < K::WAVE_LDS; o += 64 * 16) *(__attribute__((address_space(3))) v4u_t*)(tab + o) = v4u_t{0u, 0u, 0u, 0u};
    UNFLOW_WAVE_LOCKSTEP();

    const int r_begin = max(ya - R, 0), r_end = min(ybp - 1 + R, H - 1);      // source rows that feed this chunk
    const int xg = 16 * S + xl;
    // upstream gradient: load n = 0 .. 2R of a lane is displacement j = n (gf1) resp. 2R - n (gf2: plane and pixel move
    // together, + (plane - 1) per n), as the scalar offset of the load
    const unsigned jstep = MODE ? (plane - 1u) * 4u : plane * 4u;
    // DEPTH register sets of requests in flight: set k holds source rows r_begin + k, + k + DEPTH, ... (a step of ~1400 cycles is
    // shorter than a first-touch HBM access under load; with one step of lead every step waited for its loads: 62 us at level 2)
    const int nsteps = ((r_end - r_begin + 1 + DEPTH - 1) / DEPTH) * DEPTH;     // (steps past r_end: zero features, zero weights -- the window keeps sliding)
    const int r_last = r_begin + nsteps - 1;
    unsigned gb[DEPTH][ROUNDS];                             // byte offset of (round's displacement row, the set's next source row, pixel), n = 0
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
#pragma unroll
        for (int rho = 0; rho < ROUNDS; ++rho) {
            const int i = 4 * rho + q, r = r_begin + k;
            gb[k][rho] = MODE ? (((unsigned)((2 * R - i) * DD)) * plane + (unsigned)(r * W + xg + R)) * 4u
                              : (((unsigned)(i * DD)) * plane + (unsigned)((r + R - i) * W + xg)) * 4u;
        }
    const bool x_ok = xg < W;
    auto g_request = [&](unsigned (&raw)[DD], unsigned& base, int rho, int r) __attribute__((always_inline)) {
        const int i = 4 * rho + q, y = r + R - i;
        const bool ok = x_ok & (i < DD) & (y >= ya) & (y < ybp) & (r <= r_end);
        const unsigned vo = ok ? base : kOut;
#pragma unroll
        for (int n = 0; n < DD; ++n) raw[n] = __builtin_amdgcn_raw_buffer_load_b32(grs, (int)vo, (int)(n * jstep), 0);
        base += (unsigned)(DEPTH * W * 4);
    };
    // features: lane (c, gq) = 8 consecutive source pixels 16 S - 8 + 8 gq .. of channel c of each 16-channel group
    const int fx = 16 * S - 8 + 8 * q;
    const bool f_ok0 = (fx >= 0) & (fx + 3 < W), f_ok1 = (fx + 4 >= 0) & (fx + 7 < W);
    unsigned fo[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) fo[k] = ((unsigned)xl * plane + (unsigned)((r_begin + k) * W + fx)) * 4u;
    auto f_request = [&](v4u_t (&raw)[NCG][2], unsigned& base, int r) __attribute__((always_inline)) {
        const bool rok = r <= r_end;
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const bool cok = 16 * cg + xl < C;
            raw[cg][0] = __builtin_bit_cast(v4u_t, __builtin_amdgcn_raw_buffer_load_b128(frs, (int)((f_ok0 & rok & cok) ? base : kOut), (int)(cg * 16 * plane * 4), 0));
            raw[cg][1] = __builtin_bit_cast(v4u_t, __builtin_amdgcn_raw_buffer_load_b128(frs, (int)((f_ok1 & rok & cok) ? base + 16u : kOut), (int)(cg * 16 * plane * 4), 0));
        }
        base += (unsigned)(DEPTH * W * 4);
    };

    unsigned graw[DEPTH][ROUNDS][DD];
    v4u_t fraw[DEPTH][NCG][2];
    // (every group of requests is pinned where it stands, here and in the loop: the waits at the top of the loop body are the
    // weaker of what this prologue and the previous iteration leave in flight -- hipcc otherwise issues the prologue's feature loads
    // last and sinks the loop's loads to the bottom of the body, and the first use then waits for vmcnt(0) in every iteration)
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
        __builtin_amdgcn_sched_barrier(0);
        f_request(fraw[k], fo[k], r_begin + k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rho = ROUNDS - 1; rho >= 0; --rho) { g_request(graw[k][rho], gb[k][rho], rho, r_begin + k); __builtin_amdgcn_sched_barrier(0); }
    }

    v4f acc[DD + 1][NCG];
#pragma unroll
    for (int s = 0; s <= DD; ++s)
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) acc[s][cg] = v4f{0.f, 0.f, 0.f, 0.f};

    lds_byte* const wbase = tab + q * TAB + xl * RSB + 2 * (xl + SH);     // + 2 j: cell (row xl, column xl + j + SH) of table q
    lds_byte* const rbase = tab + xl * RSB + 16 * q;                      // + t TAB: row xl, columns 8 q .. 8 q + 7 of table t
    const unsigned so = ((unsigned)xl * plane + (unsigned)(16 * S + 4 * q)) * 4u;
    const bool s_ok = (16 * S + 4 * q + 3 < W);
    // (always issued, a row that must not be written goes out of range: behind a branch the number of stores in flight is unknown
    // to hipcc's s_waitcnt pass, and every load of the loop then waits for vmcnt(0).  The channel group's offset goes into the VECTOR
    // offset, not into the scalar-offset field: hipcc's hazard recogniser leaves out the wait states between a > 8-byte buffer store
    // and a VALU write to its data registers when the store names an SGPR offset -- and on gfx950 the store then sends whatever the
    // next instructions put there (measured: the second channel group's epilogue rows carried a byte offset / the out-of-range marker
    // in one element, lanes as selected by the v_cndmask that followed the store))
    auto store_row = [&](const v4f (&a)[NCG], int y, bool wanted) __attribute__((always_inline)) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const bool ok = s_ok & wanted & (16 * cg + xl < C);
            const v4f v = a[cg] * inv_c;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), ors,
                                                   (int)(ok ? so + (unsigned)(y * W * 4) + (unsigned)cg * 16u * plane * 4u : kOut), 0, 0);
        }
    };

    // one source row: set K's requests of it are consumed and re-issued for row r + DEPTH
    auto step = [&](int r, auto kk) __attribute__((always_inline)) {
        constexpr int K_ = decltype(kk)::value;
        // B operand of source row r
        v4u_t bh[NCG], bl[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const v4u_t w = fraw[K_][cg][e >> 1];
                unsigned hi, lo;
                mf_split2(__uint_as_float(w[2 * (e & 1)]), __uint_as_float(w[2 * (e & 1) + 1]), hi, lo);
                bh[cg][e] = hi; bl[cg][e] = lo;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f_request(fraw[K_], fo[K_], r + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        // SKIP: a displacement row whose output row r + R - i lies outside the chunk is left out (its weights are zeros; the halo steps
        // of a 16-row chunk are a third of all row pairs).  Its slot keeps what it held -- a value of a row outside the chunk, which
        // only rows outside the chunk inherit: a row of the chunk enters with C = 0 and meets only its own sums.  (wave-uniform)
        const unsigned span = (unsigned)(ybp - ya);
#pragma unroll
        for (int rho = ROUNDS - 1; rho >= 0; --rho) {
            const int i_lo = 4 * rho, i_hi = (4 * rho + 3 < DD ? 4 * rho + 3 : DD - 1);
            // rows of this round: r + R - i_hi .. r + R - i_lo; it has work when that range meets [ya, ybp)
            const bool round_on = !SKIP || ((r + R - i_lo >= ya) & (r + R - i_hi < ybp));
            // this round's four displacement rows -> tables (every lane writes: a lane without a valid row loaded zeros)
            if (round_on) {
#pragma unroll
            for (int n = 0; n < DD; n += 2) {
                const int j0 = MODE ? 2 * R - n : n, j1 = MODE ? 2 * R - n - 1 : n + 1;
                unsigned hi, lo;
                mf_split2(__uint_as_float(graw[K_][rho][n]), n + 1 < DD ? __uint_as_float(graw[K_][rho][n + 1]) : 0.f, hi, lo);
                *(__attribute__((address_space(3))) unsigned short*)(wbase + 2 * j0) = (unsigned short)hi;
                *(__attribute__((address_space(3))) unsigned short*)(wbase + 4 * TAB + 2 * j0) = (unsigned short)lo;
                if (n + 1 < DD) {
                    *(__attribute__((address_space(3))) unsigned short*)(wbase + 2 * j1) = (unsigned short)(hi >> 16);
                    *(__attribute__((address_space(3))) unsigned short*)(wbase + 4 * TAB + 2 * j1) = (unsigned short)(lo >> 16);
                }
            }
            }
            UNFLOW_WAVE_LOCKSTEP();
            __builtin_amdgcn_sched_barrier(0);
            g_request(graw[K_][rho], gb[K_][rho], rho, r + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 3; t >= 0; --t) {
                const int i = 4 * rho + t;
                // (SKIP 1: per row pair, every pair its own basic block; 2: per round of four -- the pairs of a round stay in one block,
                // so hipcc can interleave their MFMA chains, at the price of zero-weight pairs at the two ends of a chunk)
                if (i < DD && (SKIP == 0 || (SKIP == 2 ? round_on : (unsigned)(r + R - i - ya) < span))) {
                    const v4u_t ah = *(__attribute__((address_space(3))) const v4u_t*)(rbase + t * TAB);
                    const v4u_t al = *(__attribute__((address_space(3))) const v4u_t*)(rbase + 4 * TAB + t * TAB);
#pragma unroll
                    for (int cg = 0; cg < NCG; ++cg) {
                        v4f a = i ? acc[i][cg] : v4f{0.f, 0.f, 0.f, 0.f};
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[cg]), a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[cg]), a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[cg]), a, 0, 0, 0);
                        acc[i + 1][cg] = a;
                    }
                }
            }
        }
        store_row(acc[DD], r - R, (r - R >= ya) & (r - R < ybp));      // (row r - R has seen its last source row)
    };
#pragma unroll 1
    for (int r = r_begin; r <= r_last; r += DEPTH) {
        step(r, std::integral_constant<int, 0>{});
        if constexpr (DEPTH > 1) step(r + 1, std::integral_constant<int, 1>{});
        if constexpr (DEPTH > 2) step(r + 2, std::integral_constant<int, 2>{});
    }
    // the map ends below this chunk's last rows: they are complete as they stand (row y sits in slot r_last + R + 1 - y)
#pragma unroll
    for (int s = 1; s <= 2 * R; ++s) {
        const int y = r_last + R + 1 - s;
        store_row(acc[s], y, (y < ybp) & (y >= ya));
    }
}

// grid: one workgroup of four waves = four neighbouring segments of (sample, row chunk, gradient, channel group); the items
// that read the same gradient planes (channel groups, both gradients) are neighbours in the XCD-local order
template <int R, int NCG, int DEPTH, int SKIP>
__global__ __launch_bounds__(256, 1) void corr_bwd_mf_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                             const float* __restrict__ g, float* __restrict__ gf1,
                                                             float* __restrict__ gf2, int Ctot, int H, int W,
                                                             int nseg, int nsb, int rows, int nchunk, int ngrp, float inv_c) {
    using K = BwdMfCfg<R>;
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
    if (mode) corr_bwd_mf_body<R, NCG, 1, DEPTH, SKIP>(tab, f1, g, gf2, b, cg * NCG * 16, S, ya, ybp, Ctot, H, W, inv_c);
    else corr_bwd_mf_body<R, NCG, 0, DEPTH, SKIP>(tab, f2, g, gf1, b, cg * NCG * 16, S, ya, ybp, Ctot, H, W, inv_c);
}

static inline bool mf_offsets_fit(int C, int H, int W, int R) {
    const size_t plane = (size_t)H * W * 4, dd = 2 * R + 1;
    return (dd * dd + 2 * R + 1) * plane < 0x20000000u && (size_t)C * plane < 0x20000000u;
}

template <int R, int NCG, int DEPTH = 2, int SKIP = 1>
int launch_bwd_mf(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
                  int B, int C, int H, int W, int rows, hipStream_t s) {
    const int nseg = ceil_div(W, 16), nsb = ceil_div(nseg, 4), nchunk = ceil_div(H, rows), ngrp = ceil_div(C, NCG * 16);
    UNFLOW_LAUNCH((corr_bwd_mf_kernel<R, NCG, DEPTH, SKIP>), dim3(nsb * nchunk * B * 2 * ngrp), dim3(256), 0, s,
                  f1, f2, g, gf1, gf2, C, H, W, nseg, nsb, rows, nchunk, ngrp, 1.0f / C);
    return unflow_launch_status();
}

}  // namespace
