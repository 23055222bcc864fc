// csrc/warp.hip -- every warp kernel of the library (row-segment kernels, LDS-tile forward, tile / cell scatter backward, the one-pass gather
// backward with its displacement table, the `_ms` image warps) with its C entries -- compiled with g++ and EXECUTED on the build host, lanes as
// fibers (tests/host_check/hip_on_host.h; LDS-DMA is a memcpy, float atomics plain adds: TEST INFRASTRUCTURE, tests/test_kernels_on_host.py).
//
//   warp_check in.bin out.bin
// in : int32 ncases; per case int32 B, C, H, W, masked, align_corners; float src[B,C,H,W], flow[B,2,H,W], gout[B,C,H,W]
// out: per case out[B,C,H,W], (masked: uint8 mask[B,H,W]) and the gradients of the entry point ops.warp_flow would pick:
//      feature maps: gsrc, gflow of unflow_warp_bwd AND -- where unflow_warp_bwd_fused_supported() != 0 -- of the one-pass unflow_warp_fwd_table +
//      unflow_warp_bwd_fused(table_ready = 1) (flag + out + gsrc + gflow); masked image warps: gflow only.
// The masked cases (three scales of one pyramid, in order) are also run through unflow_warp_fwd_ms / unflow_warp_bwd_ms and compared bit for bit.
#include "warp.hip"

UnflowTimingArm& unflow_timing_arm() { static UnflowTimingArm arm = {nullptr, nullptr, false}; return arm; }      // photo.hip's: never armed here

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    FILE* o = fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int ncases;
    if (fread(&ncases, 4, 1, f) != 1) return 2;
    int failures = 0;
    struct Img { int H, W; std::vector<float> src, flow, gout, out, gflow; std::vector<uint8_t> mask; };
    std::vector<Img> pyramid;
    int pyrB = 0, pyrC = 0, pyrAc = 0;
    for (int k = 0; k < ncases; ++k) {
        int h[6];
        if (fread(h, 4, 6, f) != 6) return 2;
        const int B = h[0], C = h[1], H = h[2], W = h[3], masked = h[4], ac = h[5];
        const size_t n = (size_t)B * C * H * W, nf = (size_t)B * 2 * H * W, np = (size_t)B * H * W;
        std::vector<float> src(n), flow(nf), gout(n), out(n, -7.f), gsrc(n, -7.f), gflow(nf, -7.f);
        if (fread(src.data(), 4, n, f) != n || fread(flow.data(), 4, nf, f) != nf || fread(gout.data(), 4, n, f) != n) return 2;
        std::vector<uint8_t> mask(np, 9);
        if (unflow_warp_fwd(src.data(), flow.data(), out.data(), masked ? mask.data() : nullptr, B, C, H, W, ac, nullptr)) { printf("fwd rc, case %d\n", k); return 1; }
        fwrite(out.data(), 4, n, o);
        if (masked) {
            fwrite(mask.data(), 1, np, o);
            if (unflow_warp_bwd(src.data(), flow.data(), gout.data(), mask.data(), nullptr, gflow.data(), B, C, H, W, ac, nullptr)) return 1;
            fwrite(gflow.data(), 4, nf, o);
            pyramid.push_back(Img{H, W, src, flow, gout, out, gflow, mask});
            pyrB = B; pyrC = C; pyrAc = ac;
            continue;
        }
        if (unflow_warp_bwd(src.data(), flow.data(), gout.data(), nullptr, gsrc.data(), gflow.data(), B, C, H, W, ac, nullptr)) return 1;
        fwrite(gsrc.data(), 4, n, o); fwrite(gflow.data(), 4, nf, o);
        const int fused = unflow_warp_bwd_fused_supported(B, C, H, W);
        fwrite(&fused, 4, 1, o);
        if (fused) {
            std::vector<uint8_t> table((size_t)unflow_warp_bwd_table_bytes(B, C, H, W));
            std::vector<float> out2(n, -7.f), gsrc2(n, -7.f), gflow2(nf, -7.f);
            int rc;
            if (fused == 2) {                                       // what ops.warp_flow picks by itself: the forward leaves the table, the backward is ONE launch
                rc = unflow_warp_fwd_table(src.data(), flow.data(), out2.data(), table.data(), B, C, H, W, ac, nullptr);
                if (!rc) rc = unflow_warp_bwd_fused(src.data(), flow.data(), gout.data(), gsrc2.data(), gflow2.data(), table.data(), 1, B, C, H, W, ac, nullptr);
            } else {                                                // served, not picked: the backward computes the table itself
                out2 = out;
                rc = unflow_warp_bwd_fused(src.data(), flow.data(), gout.data(), gsrc2.data(), gflow2.data(), table.data(), 0, B, C, H, W, ac, nullptr);
            }
            if (rc) { printf("fused rc %d, case %d\n", rc, k); return 1; }
            fwrite(out2.data(), 4, n, o); fwrite(gsrc2.data(), 4, n, o); fwrite(gflow2.data(), 4, nf, o);
        }
    }
    if (!pyramid.empty()) {                                         // the pyramid's scales in one launch each way
        const int n = (int)pyramid.size();
        std::vector<const float*> src(n), flow(n), gout(n);
        std::vector<const uint8_t*> maskc(n);
        std::vector<float*> out(n), gflow(n);
        std::vector<uint8_t*> mask(n);
        std::vector<int> Hs(n), Ws(n);
        std::vector<std::vector<float>> o2(n), g2(n);
        std::vector<std::vector<uint8_t>> m2(n);
        for (int s = 0; s < n; ++s) {
            Img& q = pyramid[s];
            o2[s].assign(q.out.size(), -7.f); g2[s].assign(q.gflow.size(), -7.f); m2[s].assign(q.mask.size(), 9);
            src[s] = q.src.data(); flow[s] = q.flow.data(); gout[s] = q.gout.data(); out[s] = o2[s].data(); gflow[s] = g2[s].data(); mask[s] = m2[s].data(); maskc[s] = m2[s].data();
            Hs[s] = q.H; Ws[s] = q.W;
        }
        if (unflow_warp_fwd_ms(n, src.data(), flow.data(), out.data(), mask.data(), Hs.data(), Ws.data(), pyrB, pyrC, pyrAc, nullptr)) return 3;
        if (unflow_warp_bwd_ms(n, src.data(), flow.data(), gout.data(), maskc.data(), gflow.data(), Hs.data(), Ws.data(), pyrB, pyrC, pyrAc, nullptr)) return 3;
        for (int s = 0; s < n; ++s) {
            Img& q = pyramid[s];
            if (o2[s] != q.out || m2[s] != q.mask || g2[s] != q.gflow) { printf("MISMATCH _ms image warp, scale %d\n", s); ++failures; }
        }
    }
    fclose(f); fclose(o);
    printf("%s: %d mismatches\n", failures ? "FAILED" : "OK", failures);
    return failures ? 1 : 0;
}
