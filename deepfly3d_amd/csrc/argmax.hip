// a3: per-(view, joint) heat-map arg-max + peak confidence.   HBM-bound, read-once.
//
// One 64-lane wavefront owns one H*W plane (64*128 = 8192 floats = 32 KiB): every iteration the wave
// reads 1 KiB contiguous (float4 per lane), four iterations are kept in flight.  Each lane scans its
// elements in increasing flat index with a strict '>' so the first occurrence wins inside the lane;
// the cross-lane butterfly keeps the larger value and, on equal values, the smaller index, which
// reproduces numpy's first-index tie-break (oracle/geometry.py:heatmap_argmax).
#include "common.h"

namespace {

struct Best {
    float v;
    int i;
};

__device__ __forceinline__ Best better(Best a, Best b) {
    // NaN never wins (comparisons with NaN are false); -inf planes resolve to index 0 via the seed
    bool take_b = (b.v > a.v) || (b.v == a.v && b.i < a.i);
    return take_b ? b : a;
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ hm, int planes, int hw, int w,
                                                     float inv_h, float inv_w, float* __restrict__ pts,
                                                     float* __restrict__ conf, int* __restrict__ nonfinite_planes) {
    const int lane = threadIdx.x & 63;
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float4* src = reinterpret_cast<const float4*>(hm + (size_t)plane * hw);
    const int nvec = hw >> 2;

    Best best{-__builtin_inff(), 0x7fffffff};
    unsigned mag = 0;   // max of |x| as an integer: >= 0x7f800000 <=> an infinity or a NaN was among the values (the overflow guard of the 16-bit / f32s engines)
    auto seen = [&](const float4& q) {
        mag = max(max(mag, __float_as_uint(q.x) & 0x7fffffffu), max(__float_as_uint(q.y) & 0x7fffffffu, max(__float_as_uint(q.z) & 0x7fffffffu, __float_as_uint(q.w) & 0x7fffffffu)));
    };
    int i = lane;
    // 4 independent 16-B loads in flight per lane
    for (; i + 192 < nvec; i += 256) {
        float4 a = src[i], b = src[i + 64], c = src[i + 128], d = src[i + 192];
        const float4 q[4] = {a, b, c, d};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            seen(q[u]);
            const int base = (i + 64 * u) * 4;
            if (q[u].x > best.v) best = Best{q[u].x, base};
            if (q[u].y > best.v) best = Best{q[u].y, base + 1};
            if (q[u].z > best.v) best = Best{q[u].z, base + 2};
            if (q[u].w > best.v) best = Best{q[u].w, base + 3};
        }
    }
    for (; i < nvec; i += 64) {
        float4 a = src[i];
        seen(a);
        const int base = i * 4;
        if (a.x > best.v) best = Best{a.x, base};
        if (a.y > best.v) best = Best{a.y, base + 1};
        if (a.z > best.v) best = Best{a.z, base + 2};
        if (a.w > best.v) best = Best{a.w, base + 3};
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Best o;
        o.v = __shfl_xor(best.v, off, 64);
        o.i = __shfl_xor(best.i, off, 64);
        best = better(best, o);
        mag = max(mag, (unsigned)__shfl_xor((int)mag, off, 64));
    }
    if (lane == 0) {
        if (nonfinite_planes && mag >= 0x7f800000u) atomicAdd(nonfinite_planes, 1);
        int idx = best.i == 0x7fffffff ? 0 : best.i;
        float v = best.i == 0x7fffffff ? hm[(size_t)plane * hw] : best.v;
        pts[2 * (size_t)plane + 0] = (float)(idx / w) * inv_h;
        pts[2 * (size_t)plane + 1] = (float)(idx % w) * inv_w;
        conf[plane] = v;
    }
}

}  // namespace

extern "C" int df3d_heatmap_argmax_checked(const float* hm_dev, int n, int joints, int h, int w, float* pts_dev,
                                           float* conf_dev, int* nonfinite_planes_dev, void* stream) {
    DF3D_CHECK_ARG(n >= 0 && joints > 0 && h > 0 && w > 0, "bad shape");
    DF3D_CHECK_ARG(((h * w) & 3) == 0, "h*w must be a multiple of 4");
    if (n == 0) return DF3D_OK;
    DF3D_CHECK_ARG(hm_dev && pts_dev && conf_dev, "null pointer");
    DF3D_CHECK_ARG((reinterpret_cast<uintptr_t>(hm_dev) & 15) == 0, "heat-maps must be 16-byte aligned");
    const long long planes = (long long)n * joints;
    DF3D_CHECK_ARG(planes < (1ll << 31), "too many planes");
    const int blocks = (int)((planes + 3) / 4);
    // (row / h) computed as row * (1/h): exact for the power-of-two grids of the reference (64, 128);
    // for other sizes divide exactly instead
    const bool pow2 = ((h & (h - 1)) == 0) && ((w & (w - 1)) == 0);
    DF3D_CHECK_ARG(pow2, "h and w must be powers of two (reference heat-maps are 64 x 128)");
    hipLaunchKernelGGL(argmax_kernel, dim3(blocks), dim3(256), 0, df3d::as_stream(stream), hm_dev, (int)planes,
                       h * w, w, 1.0f / (float)h, 1.0f / (float)w, pts_dev, conf_dev, nonfinite_planes_dev);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

extern "C" int df3d_heatmap_argmax(const float* hm_dev, int n, int joints, int h, int w, float* pts_dev,
                                   float* conf_dev, void* stream) {
    return df3d_heatmap_argmax_checked(hm_dev, n, joints, h, w, pts_dev, conf_dev, nullptr, stream);
}
