// a4 (19 -> 38 joint re-layout) and a6 (multi-view DLT triangulation), float64.
//
// Triangulation: one thread per (frame, joint).  The thread accumulates the 4x4 normal matrix
// M = A^T A of the DLT system (rows x*P2 - P0, y*P2 - P1 of every camera that sees the joint) in
// registers, diagonalises it with cyclic Jacobi rotations (fully unrolled, no scratch) and returns
// the eigenvector of the smallest eigenvalue = the last right-singular vector of A
// (oracle/geometry.py:triangulate_dlt).  M is pre-scaled by 1/trace so the
// rotations work on O(1) numbers; scaling does not move eigenvectors.  Latency-bound: 5 KB/frame.
#include "common.h"

namespace {

constexpr int MAX_CAM = 8;

struct CamP {
    double p[MAX_CAM][12];
};

template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&a)[4][4], double (&v)[4][4]) {
    const double apq = a[P][Q];
    if (apq == 0.0) return;
    const double app = a[P][P], aqq = a[Q][Q];
    // tiny off-diagonal relative to the diagonal: nothing to do
    if (fabs(apq) <= 1e-300) return;
    const double theta = (aqq - app) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0);
    const double s = t * c;
    a[P][P] = app - t * apq;
    a[Q][Q] = aqq + t * apq;
    a[P][Q] = 0.0;
    a[Q][P] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k != P && k != Q) {
            const double akp = a[k][P], akq = a[k][Q];
            a[k][P] = c * akp - s * akq;
            a[P][k] = a[k][P];
            a[k][Q] = s * akp + c * akq;
            a[Q][k] = a[k][Q];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double vkp = v[k][P], vkq = v[k][Q];
        v[k][P] = c * vkp - s * vkq;
        v[k][Q] = s * vkp + c * vkq;
    }
}

__global__ __launch_bounds__(256) void triangulate_kernel(CamP cams, const double* __restrict__ pts, int ncam,
                                                          long long TJ, double row_scale, double col_scale,
                                                          double* __restrict__ X) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= TJ) return;

    double a[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = 0.0;

    int nviews = 0;
    for (int c = 0; c < ncam; ++c) {
        const double2 rc = *reinterpret_cast<const double2*>(pts + ((size_t)c * TJ + idx) * 2);
        const double row = rc.x * row_scale, col = rc.y * col_scale;
        if (row != 0.0 && col != 0.0) {
            ++nviews;
            double r0[4], r1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r0[k] = col * cams.p[c][8 + k] - cams.p[c][k];      // x * P[2] - P[0],  x = col_px
                r1[k] = row * cams.p[c][8 + k] - cams.p[c][4 + k];  // y * P[2] - P[1],  y = row_px
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = i; j < 4; ++j) a[i][j] += r0[i] * r0[j] + r1[i] * r1[j];
        }
    }
    double out0 = 0.0, out1 = 0.0, out2 = 0.0;
    if (nviews >= 2) {
        const double tr = a[0][0] + a[1][1] + a[2][2] + a[3][3];
        const double inv = tr > 0.0 ? 1.0 / tr : 1.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) {
                a[i][j] *= inv;
                a[j][i] = a[i][j];
            }
        double v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;

        for (int sweep = 0; sweep < 16; ++sweep) {
            const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[0][3]) + fabs(a[1][2]) + fabs(a[1][3]) +
                               fabs(a[2][3]);
            if (off < 1e-40) break;
            jacobi_rotate<0, 1>(a, v);
            jacobi_rotate<0, 2>(a, v);
            jacobi_rotate<0, 3>(a, v);
            jacobi_rotate<1, 2>(a, v);
            jacobi_rotate<1, 3>(a, v);
            jacobi_rotate<2, 3>(a, v);
        }
        // eigenvector of the smallest eigenvalue (select with a compare chain: no dynamic indexing)
        double best = a[0][0];
        double e0 = v[0][0], e1 = v[1][0], e2 = v[2][0], e3 = v[3][0];
        if (a[1][1] < best) { best = a[1][1]; e0 = v[0][1]; e1 = v[1][1]; e2 = v[2][1]; e3 = v[3][1]; }
        if (a[2][2] < best) { best = a[2][2]; e0 = v[0][2]; e1 = v[1][2]; e2 = v[2][2]; e3 = v[3][2]; }
        if (a[3][3] < best) { best = a[3][3]; e0 = v[0][3]; e1 = v[1][3]; e2 = v[2][3]; e3 = v[3][3]; }
        out0 = e0 / e3;
        out1 = e1 / e3;
        out2 = e2 / e3;
    }
    X[idx * 3 + 0] = out0;
    X[idx * 3 + 1] = out1;
    X[idx * 3 + 2] = out2;
}

struct Order {
    int o[7];
};

// out[7, T, 38, 2] from pts19[7, T, 19, 2]; one thread per output (cam, t, joint)
__global__ __launch_bounds__(256) void relayout_kernel(const float* __restrict__ in, Order ord, int T,
                                                       double* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = 7ll * T * 38;
    if (idx >= total) return;
    const int j = (int)(idx % 38);
    const long long ct = idx / 38;
    const int t = (int)(ct % T);
    const int cam = (int)(ct / T);
    // position of this physical camera in the ordering (ordering is a permutation of 0..6)
    int pos = -1;
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (ord.o[k] == cam) pos = k;
    double row = 0.0, col = 0.0;
    const bool right = pos >= 0 && pos < 3;   // ordering[0:3] -> joints 0..18
    const bool left = pos >= 4 && pos < 7;    // ordering[4:7] -> joints 19..37
    int src = -1;
    if (right && j < 19) src = j;
    if (left && j >= 19) src = j - 19;
    if (pos == 2 && j >= 15) src = -1;       // ordering[2] cannot see antenna / stripes
    if (pos == 4 && j >= 19 + 15) src = -1;  // ordering[4] neither
    if (src >= 0) {
        const float2 p = *reinterpret_cast<const float2*>(in + (((size_t)cam * T + t) * 19 + src) * 2);
        row = (double)p.x;
        col = (double)p.y;
    }
    if (left) col = 1.0 - col;  // un-flip: applied to ALL 38 joints of the three left-side cameras
    out[idx * 2 + 0] = row;
    out[idx * 2 + 1] = col;
}

}  // namespace

static int triangulate_impl(const double* P_dev_or_host, const double* pts_px_dev, double row_scale, double col_scale,
                            int ncam, int T, int J, double* X_dev, void* stream) {
    DF3D_CHECK_ARG(ncam >= 1 && ncam <= MAX_CAM, "ncam must be in [1, 8]");
    DF3D_CHECK_ARG(T >= 0 && J > 0, "bad shape");
    if (T == 0) return DF3D_OK;
    DF3D_CHECK_ARG(P_dev_or_host && pts_px_dev && X_dev, "null pointer");
    // the 7 x 12 projection entries travel as a kernel argument (scalar registers / constant loads)
    CamP cams;
    memset(&cams, 0, sizeof(cams));
    hipPointerAttribute_t attr;
    bool on_device = false;
    if (hipPointerGetAttributes(&attr, P_dev_or_host) == hipSuccess)
        on_device = (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
    else
        (void)hipGetLastError();
    if (on_device) {
        DF3D_HIP(hipMemcpyAsync(cams.p, P_dev_or_host, sizeof(double) * 12 * ncam, hipMemcpyDeviceToHost,
                                df3d::as_stream(stream)));
        DF3D_HIP(hipStreamSynchronize(df3d::as_stream(stream)));
    } else {
        memcpy(cams.p, P_dev_or_host, sizeof(double) * 12 * ncam);
    }
    const long long TJ = (long long)T * J;
    const int blocks = (int)((TJ + 255) / 256);
    hipLaunchKernelGGL(triangulate_kernel, dim3(blocks), dim3(256), 0, df3d::as_stream(stream), cams, pts_px_dev,
                       ncam, TJ, row_scale, col_scale, X_dev);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

extern "C" int df3d_triangulate(const double* P, const double* pts_px_dev, int ncam, int T, int J, double* X_dev,
                                void* stream) {
    return triangulate_impl(P, pts_px_dev, 1.0, 1.0, ncam, T, J, X_dev, stream);
}

extern "C" int df3d_triangulate_scaled(const double* P, const double* pts_norm_dev, double row_scale, double col_scale,
                                       int ncam, int T, int J, double* X_dev, void* stream) {
    DF3D_CHECK_ARG(row_scale > 0 && col_scale > 0, "scales must be positive");
    return triangulate_impl(P, pts_norm_dev, row_scale, col_scale, ncam, T, J, X_dev, stream);
}

extern "C" int df3d_relayout_19_to_38(const float* pts19_dev, const int* ordering_host, int T, double* out_dev,
                                      void* stream) {
    DF3D_CHECK_ARG(T >= 0, "bad T");
    if (T == 0) return DF3D_OK;
    DF3D_CHECK_ARG(pts19_dev && ordering_host && out_dev, "null pointer");
    Order ord;
    int seen = 0;
    for (int k = 0; k < 7; ++k) {
        DF3D_CHECK_ARG(ordering_host[k] >= 0 && ordering_host[k] < 7, "camera ordering entries must be in [0, 6]");
        seen |= 1 << ordering_host[k];
        ord.o[k] = ordering_host[k];
    }
    DF3D_CHECK_ARG(seen == 0x7f, "camera ordering must be a permutation of 0..6");
    const long long total = 7ll * T * 38;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(relayout_kernel, dim3(blocks), dim3(256), 0, df3d::as_stream(stream), pts19_dev, ord, T,
                       out_dev);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}
