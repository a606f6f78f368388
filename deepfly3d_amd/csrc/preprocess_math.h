// The arithmetic of the input front-end (uint8 camera frame -> one pixel of the network input), shared by
// preprocess_kernel (csrc/preprocess.hip) and by the stem kernels when they sample the camera frames themselves
// (df3d_hg_forward_u8): optional left-right flip, bilinear down-scale with half-pixel centres, grey -> 3 channels,
// (v / 255 - mean) * (1 / std).  Multiply-add fusion is off inside, so both users round identically (and like the numpy oracle).
#pragma once
#include <hip/hip_runtime.h>

namespace df3d_pre {

struct Norm {
    float mean[3];
    float inv_std[3];
};

// frame: the view's [H][W][C] uint8 image (C = 1 or 3); (oy, ox) a pixel of the OH x OW network input
__device__ __forceinline__ void pixel(const unsigned char* __restrict__ frame, int H, int W, int C, bool flip, int OH, int OW, int oy, int ox,
                                      const Norm& nm, float res[3]) {
#pragma clang fp contract(off)
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
    fy = fminf(fmaxf(fy, 0.0f), (float)(H - 1));
    fx = fminf(fmaxf(fx, 0.0f), (float)(W - 1));
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const int xa = flip ? W - 1 - x0 : x0, xb = flip ? W - 1 - x1 : x1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cc = C == 1 ? 0 : c;
        const float p00 = frame[((size_t)y0 * W + xa) * C + cc], p01 = frame[((size_t)y0 * W + xb) * C + cc];
        const float p10 = frame[((size_t)y1 * W + xa) * C + cc], p11 = frame[((size_t)y1 * W + xb) * C + cc];
        const float top = p00 + (p01 - p00) * wx, bot = p10 + (p11 - p10) * wx;
        res[c] = ((top + (bot - top) * wy) * (1.0f / 255.0f) - nm.mean[c]) * nm.inv_std[c];
    }
}

}  // namespace df3d_pre
