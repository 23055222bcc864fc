// The input-stage kernel (csrc/prepare.hip: KITTI_Prepared.__getitem__ after the PNG decode, core/dataset/kitti_prepared.py:63-90,145-148 --
// split, cv2's 8-bit INTER_LINEAR resize in fixed point, flip, / 255, HWC -> CHW) EXECUTED on the build host: this program includes the
// shipped source file itself, compiled with g++, and runs every workgroup and lane of the launch unflow_prepare_triplets() makes
// (test infrastructure; tests/test_kernels_on_host.py compares the result with oracle/prepare_cpu.py byte for byte).
//
// The kernel's one barrier separates the fill of a 256-entry table in LDS from its use; lanes run one after the other here, so each
// workgroup is run twice -- the second pass finds the table complete and rewrites every output (the kernel is idempotent).
//
//   g++ -O1 -std=c++17 -ffp-contract=off -DUNFLOW_HOST_CHECK -I unopticalflow_amd/csrc tests/host_check/prepare_check.cpp -o prepare_check
//   prepare_check in.bin out.bin      in: int32 B, H, W, swap_rb; per image int32 rows, cols, flip; then the images' bytes back to back
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
static inline void __syncthreads() {}
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static dim3 threadIdx, blockIdx, blockDim, gridDim;
using std::max;
using std::min;
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
#define UNFLOW_EINVAL (-22)
typedef void* hipStream_t;
static inline int hipGetLastError() { return 0; }
template <class K, class... A>
static void launch(K kernel, dim3 grid, dim3 block, A... args) {
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (int pass = 0; pass < 2; ++pass)                                  // (see above: the table in LDS is complete for the second pass)
            for (unsigned tx = 0; tx < block.x; ++tx) { threadIdx = dim3(tx, 0, 0); kernel(args...); }
    }
}
#define UNFLOW_LAUNCH(kernel, grid, block, shmem, stream, ...) launch(kernel, grid, block, __VA_ARGS__)

#include "prepare.hip"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[4];
    if (fread(hdr, 4, 4, f) != 4) return 2;
    const int B = hdr[0], H = hdr[1], W = hdr[2], swap_rb = hdr[3];
    std::vector<int> dims(2 * B);
    std::vector<unsigned char> flip(B);
    std::vector<long long> offsets(B);
    long long total = 0;
    for (int b = 0; b < B; ++b) {
        int t[3];
        if (fread(t, 4, 3, f) != 3) return 2;
        dims[2 * b] = t[0]; dims[2 * b + 1] = t[1]; flip[b] = (unsigned char)t[2];
        offsets[b] = total; total += (long long)t[0] * t[1] * 3;
    }
    std::vector<unsigned char> src((size_t)total);
    if (fread(src.data(), 1, src.size(), f) != src.size()) return 2;
    fclose(f);
    std::vector<float> dst((size_t)B * 3 * 3 * H * W, -1.f);
    const int rc = unflow_prepare_triplets(src.data(), offsets.data(), dims.data(), flip.data(), dst.data(), B, H, W, swap_rb, nullptr);
    if (rc != 0) { printf("rc %d\n", rc); return 1; }
    if (unflow_prepare_triplets(src.data(), offsets.data(), dims.data(), flip.data(), dst.data(), B, H, W + 2, swap_rb, nullptr) != UNFLOW_EINVAL) return 1;   // W % 4
    f = fopen(argv[2], "wb");
    fwrite(dst.data(), 4, dst.size(), f);
    fclose(f);
    printf("OK\n");
    return 0;
}
