// Cost volume (correlation) forward / backward for gfx950.
//
// Replaces PWC_tf.corr_naive (reference core/networks/structures/pwc_tf.py:97-106) and its
// autograd: 81 (mul, mean, cat) launches become one kernel that reads f1 and f2 once.
//
//   cv[b, i*DD+j, y, x] = (1/C) sum_c f1[b,c,y,x] * f2[b,c,y+i-R,x+j-R]     DD = 2R+1
//
// Layout: NCHW fp32, x contiguous.  A 256-thread workgroup owns a TW x 8 pixel tile of one
// sample; the f2 halo tile of CC channels is staged in LDS with coalesced row reads, f1 goes
// straight to registers (no reuse across lanes), every lane keeps PX pixels x DG x DD
// displacement accumulators in VGPRs and walks the LDS rows (ds_read_b64 when PX == 2).
// The backward is the same gather twice: gf1 = sum_ij g[ij] * f2(shifted), and
// gf2 = sum_ij g[ij](shifted back) * f1(shifted back), written as a gather -- no atomics
// (except when the displacement rows are split over workgroups, R > 4).
#include "common.h"

namespace {

constexpr int TXT = 32;   // lanes along x
constexpr int TY = 8;     // rows per workgroup

template <int R, int PX, int DG, int CC>
struct CorrCfg {
    static constexpr int DD = 2 * R + 1;
    static constexpr int TW = TXT * PX;
    static constexpr int LW = TW + 2 * R + ((TW + 2 * R) & 1);   // even: rows stay 8-byte aligned
    static constexpr int LH = TY + DG - 1;
    static constexpr int NG = (DD + DG - 1) / DG;                // displacement-row groups
};

// Stage CC channels of the (LH x LW) halo tile of `src` (sample b, channels c0..) whose top-left
// pixel is (ytop, xleft); zero outside the image / channel range.
template <int LH, int LW, int CC>
__device__ __forceinline__ void stage_tile(float (*tile)[LH][LW], const float* __restrict__ src,
                                           int b, int C, int H, int W, int c0, int ytop, int xleft) {
    constexpr int PLANE = LH * LW;
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

template <int R, int PX, int DG, int CC>
__global__ __launch_bounds__(256) void corr_fwd_kernel(const float* __restrict__ f1,
                                                       const float* __restrict__ f2,
                                                       float* __restrict__ cv, int C, int H, int W,
                                                       float inv_c) {
    using K = CorrCfg<R, PX, DG, CC>;
    constexpr int DD = K::DD, LW = K::LW, LH = K::LH, NG = K::NG;
    constexpr int NROW = PX + 2 * R + ((PX + 2 * R) & (PX == 2 ? 1 : 0));
    __shared__ __attribute__((aligned(16))) float tile[CC][LH][LW];

    const int tx = threadIdx.x & (TXT - 1), ty = threadIdx.x >> 5;
    const int b = blockIdx.z / NG, grp = blockIdx.z - b * NG;
    const int i0 = grp * DG;
    const int x0 = blockIdx.x * K::TW, y0 = blockIdx.y * TY;
    const int px = x0 + tx * PX, py = y0 + ty;

    float acc[DG][DD][PX];
#pragma unroll
    for (int i = 0; i < DG; ++i)
#pragma unroll
        for (int j = 0; j < DD; ++j)
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[i][j][p] = 0.f;

    for (int c0 = 0; c0 < C; c0 += CC) {
        float a[CC][PX];
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int p = 0; p < PX; ++p)
                a[c][p] = (c0 + c < C && py < H && px + p < W)
                              ? f1[((size_t)(b * C + c0 + c) * H + py) * W + px + p] : 0.f;
        stage_tile<LH, LW, CC>(tile, f2, b, C, H, W, c0, y0 - R + i0, x0 - R);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CC; ++c) {
#pragma unroll
            for (int i = 0; i < DG; ++i) {
                float row[NROW];
                load_row<NROW, PX>(row, &tile[c][ty + i][tx * PX]);
#pragma unroll
                for (int j = 0; j < DD; ++j)
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[i][j][p] = fmaf(a[c][p], row[j + p], acc[i][j][p]);
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
__global__ __launch_bounds__(256) void corr_bwd_kernel(const float* __restrict__ F,
                                                       const float* __restrict__ g,
                                                       float* __restrict__ out, int C, int H, int W,
                                                       float inv_c) {
    using K = CorrCfg<R, PX, DG, CC>;
    constexpr int DD = K::DD, LW = K::LW, LH = K::LH, NG = K::NG;
    constexpr int NROW = PX + 2 * R + ((PX + 2 * R) & (PX == 2 ? 1 : 0));
    __shared__ __attribute__((aligned(16))) float tile[CC][LH][LW];

    const int tx = threadIdx.x & (TXT - 1), ty = threadIdx.x >> 5;
    const int b = blockIdx.z / NG, grp = blockIdx.z - b * NG;
    const int i0 = grp * DG;
    const int x0 = blockIdx.x * K::TW, y0 = blockIdx.y * TY;
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

    for (int c0 = 0; c0 < C; c0 += CC) {
        stage_tile<LH, LW, CC>(tile, F, b, C, H, W, c0, y0 - R + i0, x0 - R);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            float acc[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[p] = 0.f;
#pragma unroll
            for (int i = 0; i < DG; ++i) {
                float row[NROW];
                load_row<NROW, PX>(row, &tile[c][ty + i][tx * PX]);
#pragma unroll
                for (int j = 0; j < DD; ++j)
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[p] = fmaf(wr[i][j][p], row[j + p], acc[p]);
            }
            if (c0 + c < C && py < H) {
                float* o = out + ((size_t)(b * C + c0 + c) * H + py) * W + px;
#pragma unroll
                for (int p = 0; p < PX; ++p)
                    if (px + p < W) {
                        if (NG == 1) o[p] = acc[p] * inv_c;
                        else atomicAdd(o + p, acc[p] * inv_c);
                    }
            }
        }
        __syncthreads();
    }
}

// ---- any-radius fallback (one thread per output element, direct global reads) ----
__global__ void corr_fwd_generic(const float* __restrict__ f1, const float* __restrict__ f2,
                                 float* __restrict__ cv, int B, int C, int H, int W, int R, float inv_c) {
    const int DD = 2 * R + 1;
    const size_t n = (size_t)B * DD * DD * H * W;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int x = t % W, y = (t / W) % H, ij = (t / ((size_t)W * H)) % (DD * DD);
        const int b = t / ((size_t)W * H * DD * DD);
        const int sy = y + ij / DD - R, sx = x + ij % DD - R;
        float s = 0.f;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W)
            for (int c = 0; c < C; ++c)
                s = fmaf(f1[((size_t)(b * C + c) * H + y) * W + x], f2[((size_t)(b * C + c) * H + sy) * W + sx], s);
        cv[t] = s * inv_c;
    }
}

__global__ void corr_bwd_generic(const float* __restrict__ f1, const float* __restrict__ f2,
                                 const float* __restrict__ g, float* __restrict__ gf1,
                                 float* __restrict__ gf2, int B, int C, int H, int W, int R, float inv_c) {
    const int DD = 2 * R + 1;
    const size_t n = (size_t)B * C * H * W, plane = (size_t)H * W;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int x = t % W, y = (t / W) % H, c = (t / plane) % C;
        const int b = t / (plane * C);
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < DD; ++i)
            for (int j = 0; j < DD; ++j) {
                const int sy = y + i - R, sx = x + j - R;       // f2 position seen from (y,x)
                if (sy >= 0 && sy < H && sx >= 0 && sx < W)
                    s1 = fmaf(g[((size_t)b * DD * DD + i * DD + j) * plane + (size_t)y * W + x],
                              f2[((size_t)(b * C + c) * H + sy) * W + sx], s1);
                const int qy = y - (i - R), qx = x - (j - R);   // f1 position that looked at (y,x)
                if (qy >= 0 && qy < H && qx >= 0 && qx < W)
                    s2 = fmaf(g[((size_t)b * DD * DD + i * DD + j) * plane + (size_t)qy * W + qx],
                              f1[((size_t)(b * C + c) * H + qy) * W + qx], s2);
            }
        gf1[t] = s1 * inv_c;
        gf2[t] = s2 * inv_c;
    }
}

template <int R, int PX, int DG, int CC>
int launch_fwd(const float* f1, const float* f2, float* cv, int B, int C, int H, int W, hipStream_t s) {
    using K = CorrCfg<R, PX, DG, CC>;
    dim3 grid(ceil_div(W, K::TW), ceil_div(H, TY), B * K::NG);
    hipLaunchKernelGGL((corr_fwd_kernel<R, PX, DG, CC>), grid, dim3(256), 0, s, f1, f2, cv, C, H, W, 1.0f / C);
    return unflow_launch_status();
}

template <int R, int PX, int DG, int CC>
int launch_bwd(const float* f1, const float* f2, const float* g, float* gf1, float* gf2,
               int B, int C, int H, int W, hipStream_t s) {
    using K = CorrCfg<R, PX, DG, CC>;
    dim3 grid(ceil_div(W, K::TW), ceil_div(H, TY), B * K::NG);
    if (K::NG > 1) {
        const size_t bytes = (size_t)B * C * H * W * sizeof(float);
        hipError_t e = hipMemsetAsync(gf1, 0, bytes, s);
        if (e != hipSuccess) return (int)e;
        e = hipMemsetAsync(gf2, 0, bytes, s);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((corr_bwd_kernel<R, PX, DG, CC, 0>), grid, dim3(256), 0, s, f2, g, gf1, C, H, W, 1.0f / C);
    hipLaunchKernelGGL((corr_bwd_kernel<R, PX, DG, CC, 1>), grid, dim3(256), 0, s, f1, g, gf2, C, H, W, 1.0f / C);
    return unflow_launch_status();
}

}  // namespace

extern "C" int unflow_corr_fwd(const float* f1, const float* f2, float* cv, int B, int C, int H, int W,
                               int d, void* stream) {
    UNFLOW_REQUIRE(f1 && f2 && cv && B > 0 && C > 0 && H > 0 && W > 0 && d >= 0);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (W >= 96);          // 64-pixel tiles only where they fill
    switch (d) {
        case 1: return launch_fwd<1, 2, 3, 8>(f1, f2, cv, B, C, H, W, s);
        case 2: return launch_fwd<2, 2, 5, 8>(f1, f2, cv, B, C, H, W, s);
        case 4: return wide ? launch_fwd<4, 2, 9, 8>(f1, f2, cv, B, C, H, W, s)
                            : launch_fwd<4, 1, 9, 8>(f1, f2, cv, B, C, H, W, s);
        case 8: return launch_fwd<8, 1, 6, 8>(f1, f2, cv, B, C, H, W, s);
        default: break;
    }
    const size_t n = (size_t)B * (2 * d + 1) * (2 * d + 1) * H * W;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(corr_fwd_generic, dim3(blocks), dim3(256), 0, s, f1, f2, cv, B, C, H, W, d, 1.0f / C);
    return unflow_launch_status();
}

extern "C" int unflow_corr_bwd(const float* f1, const float* f2, const float* gcv, float* gf1, float* gf2,
                               int B, int C, int H, int W, int d, void* stream) {
    UNFLOW_REQUIRE(f1 && f2 && gcv && gf1 && gf2 && B > 0 && C > 0 && H > 0 && W > 0 && d >= 0);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (W >= 96);
    switch (d) {
        case 1: return launch_bwd<1, 2, 3, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
        case 2: return launch_bwd<2, 2, 5, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
        case 4: return wide ? launch_bwd<4, 2, 9, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s)
                            : launch_bwd<4, 1, 9, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
        case 8: return launch_bwd<8, 1, 6, 8>(f1, f2, gcv, gf1, gf2, B, C, H, W, s);
        default: break;
    }
    const size_t n = (size_t)B * C * H * W;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(corr_bwd_generic, dim3(blocks), dim3(256), 0, s, f1, f2, gcv, gf1, gf2, B, C, H, W, d, 1.0f / C);
    return unflow_launch_status();
}
