// The arithmetic of the input front-end (uint8 camera frame -> one pixel of the network input), shared by
// preprocess_kernel (csrc/preprocess.hip) and by the stem kernels when they sample the camera frames themselves
// (df3d_hg_forward_u8): optional left-right flip, down-scale, grey -> 3 channels, (v / 255 - mean) * (1 / std).
// Multiply-add fusion is off inside, so both users round identically (and like the numpy oracle).
//
// df2d's resize rule is not in the reference checkout, so it is DATA (Norm::resize, DF3D_RESIZE_* of df3d_hip.h):
//   0  bilinear, half-pixel centres, no antialias   (cv2.INTER_LINEAR / torch interpolate(align_corners=False))
//   1  bilinear, corner-aligned                     (torch interpolate(align_corners=True))
//   2  area: every output pixel is the mean of the source rectangle it covers, source pixels weighted by their overlap
//      (cv2.INTER_AREA for a down-scale)
#pragma once
#include <hip/hip_runtime.h>

namespace df3d_pre {

struct Norm {
    float mean[3];
    float inv_std[3];
    int resize;   // DF3D_RESIZE_BILINEAR (0) | DF3D_RESIZE_BILINEAR_ALIGN_CORNERS (1) | DF3D_RESIZE_AREA (2)
};

// frame: the view's [H][W][C] uint8 image (C = 1 or 3); (oy, ox) a pixel of the OH x OW network input
__device__ __forceinline__ void pixel(const unsigned char* __restrict__ frame, int H, int W, int C, bool flip, int OH, int OW, int oy, int ox,
                                      const Norm& nm, float res[3]) {
#pragma clang fp contract(off)
    if (nm.resize == 2) {
        // source interval [o * s, (o + 1) * s) per axis; weights = overlap lengths / s
        const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
        const float ya = oy * sy, yb = fminf((oy + 1) * sy, (float)H);
        const float xa = ox * sx, xb = fminf((ox + 1) * sx, (float)W);
        const int y0 = (int)ya, y1 = min((int)ceilf(yb), H);
        const int x0 = (int)xa, x1 = min((int)ceilf(xb), W);
        const float inv = 1.0f / ((yb - ya) * (xb - xa));
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (int y = y0; y < y1; ++y) {
            const float wy = fminf(yb, (float)(y + 1)) - fmaxf(ya, (float)y);
            float row[3] = {0.0f, 0.0f, 0.0f};
            for (int x = x0; x < x1; ++x) {
                const float wx = fminf(xb, (float)(x + 1)) - fmaxf(xa, (float)x);
                const int xs = flip ? W - 1 - x : x;
#pragma unroll
                for (int c = 0; c < 3; ++c) row[c] = row[c] + wx * (float)frame[((size_t)y * W + xs) * C + (C == 1 ? 0 : c)];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + wy * row[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) res[c] = ((acc[c] * inv) * (1.0f / 255.0f) - nm.mean[c]) * nm.inv_std[c];
        return;
    }
    float fy, fx;
    if (nm.resize == 1) {
        fy = OH > 1 ? oy * ((float)(H - 1) / (float)(OH - 1)) : 0.0f;
        fx = OW > 1 ? ox * ((float)(W - 1) / (float)(OW - 1)) : 0.0f;
    } else {
        const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
        fy = (oy + 0.5f) * sy - 0.5f;
        fx = (ox + 0.5f) * sx - 0.5f;
    }
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
