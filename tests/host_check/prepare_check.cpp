// The input-stage kernel (csrc/prepare.hip: KITTI_Prepared.__getitem__ after the PNG decode, core/dataset/kitti_prepared.py:63-90,145-148 --
// split, cv2's 8-bit INTER_LINEAR resize in fixed point, flip, / 255, HWC -> CHW) EXECUTED on the build host: this program includes the
// shipped source file itself, compiled with g++, and runs every workgroup and lane of the launch unflow_prepare_triplets() makes
// (test infrastructure; tests/test_kernels_on_host.py compares the result with oracle/prepare_cpu.py byte for byte).
//
// Lanes are fibers (tests/host_check/hip_on_host.h): the kernel's one barrier -- between the fill of a 256-entry table in LDS and its use -- is real.
//
//   g++ -O1 -std=c++20 -ffp-contract=off -DUNFLOW_HOST_CHECK -I tests/host_check -I unopticalflow_amd/csrc tests/host_check/prepare_check.cpp -o prepare_check
//   prepare_check in.bin out.bin      in: int32 B, H, W, swap_rb; per image int32 rows, cols, flip; then the images' bytes back to back
#include "prepare.hip"

UnflowTimingArm& unflow_timing_arm() { static UnflowTimingArm arm = {nullptr, nullptr, false}; return arm; }      // photo.hip's: never armed here

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
