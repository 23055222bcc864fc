// Host-side helper of the evaluation path (row N1): undo the PNG row filters in place.
// KITTI flow ground truth is 16-bit RGB PNG, which needs a byte-exact decoder; the Average / Paeth
// filters are sequential per byte, so this one loop is native instead of Python.
#include <stdint.h>
#include <stdlib.h>
#include "../../include/unflow_hip.h"

extern "C" int unflow_png_unfilter(uint8_t* rows, int height, int stride, int bpp) {
    // rows: height x (1 + stride) bytes (filter type byte + filtered scanline), rewritten to raw bytes
    if (!rows || height <= 0 || stride <= 0 || bpp <= 0) return UNFLOW_EINVAL;
    const uint8_t* prev = nullptr;
    for (int y = 0; y < height; ++y) {
        uint8_t* line = rows + (size_t)y * (stride + 1);
        const int ft = line[0];
        uint8_t* cur = line + 1;
        for (int x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0;
            const int b = prev ? prev[x] : 0;
            const int c = (prev && x >= bpp) ? prev[x - bpp] : 0;
            int pred;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return UNFLOW_EINVAL;
            }
            cur[x] = (uint8_t)(cur[x] + pred);
        }
        prev = cur;
    }
    return 0;
}
