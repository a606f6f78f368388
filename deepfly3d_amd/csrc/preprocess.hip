// Input front-end of inference_folder: uint8 camera frames -> network input.
// One thread per output pixel: optional left-right flip (cameras facing left, reference df3d/core.py:179),
// down-scale (rule = DF3D_RESIZE_*), grey -> 3 channels, (v/255 - mean) / std.  HBM-bound.
// df2d's exact resize/normalisation is not in the reference checkout ("parity unpinned"), so mean/std and the rule are data.
#include "common.h"
#include "preprocess_math.h"

namespace {

using df3d_pre::Norm;

__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ flip,
                                                         int n, int H, int W, int C, float* __restrict__ out, int OH, int OW, Norm nm) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n * OH * OW;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int v = (int)(idx / ((long long)OW * OH));
    float res[3];
    df3d_pre::pixel(img + (size_t)v * H * W * C, H, W, C, flip && flip[v], OH, OW, oy, ox, nm, res);
    float* o = out + idx * 3;
    o[0] = res[0];
    o[1] = res[1];
    o[2] = res[2];
}

}  // namespace

extern "C" int df3d_preprocess_u8(const unsigned char* img_dev, const unsigned char* flip_dev, int n, int H, int W, int C,
                                  float* out_dev, int OH, int OW, const float* mean3_host, const float* std3_host, int resize, void* stream) {
    DF3D_CHECK_ARG(resize >= DF3D_RESIZE_BILINEAR && resize <= DF3D_RESIZE_AREA, "resize must be one of DF3D_RESIZE_*");
    DF3D_CHECK_ARG(n >= 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "bad shape");
    DF3D_CHECK_ARG(C == 1 || C == 3, "C must be 1 or 3");
    if (n == 0) return DF3D_OK;
    DF3D_CHECK_ARG(img_dev && out_dev && mean3_host && std3_host, "null pointer");
    Norm nm;
    nm.resize = resize;
    for (int c = 0; c < 3; ++c) {
        DF3D_CHECK_ARG(std3_host[c] != 0.0f, "std must be non-zero");
        nm.mean[c] = mean3_host[c];
        nm.inv_std[c] = 1.0f / std3_host[c];
    }
    const long long total = (long long)n * OH * OW;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, df3d::as_stream(stream), img_dev,
                       flip_dev, n, H, W, C, out_dev, OH, OW, nm);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}
