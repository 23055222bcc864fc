// The any-radius fallback of the cost volume (pwc_tf.py:97-106 and its autograd): one lane per output element, direct global reads.  In a
// header of its own because it is what tests/host_check/hostexec_corr.cpp -- the cost-volume entry points of the host-executed test library --
// compiles next to corr_mfma*.h: the fast fp32 kernels of corr.hip (LDS-DMA rings, hand-issued ds_read, packed-FMA register layouts) have
// no host form.  Included by corr.hip inside its anonymous namespace.
#pragma once

__global__ void corr_fwd_generic(const float* __restrict__ f1, const float* __restrict__ f2,
                                 float* __restrict__ cv, int B, int C, int H, int W, int R, float inv_c) {
    const int DD = 2 * R + 1;
    const size_t n = (size_t)B * DD * DD * H * W;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int x = t % W, y = (t / W) % H, ij = (t / ((size_t)W * H)) % (DD * DD);
        const int b = t / ((size_t)W * H * DD * DD);
        const int sy = y + ij / DD - R, sx = x + ij % DD - R;
        float s = 0.f;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            const float* p1 = f1 + ((size_t)b * C * H + y) * W + x;
            const float* p2 = f2 + ((size_t)b * C * H + sy) * W + sx;
            const size_t plane = (size_t)H * W;
#pragma unroll 8      // (32 in flight: 15.0 -> 18.1 us at level 5 -- the kernel is bound by its L1 traffic, not by a latency chain)
            for (int c = 0; c < C; ++c) s = fmaf(p1[c * plane], p2[c * plane], s);
        }
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
