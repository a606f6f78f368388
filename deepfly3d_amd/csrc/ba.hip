// a7: bundle-adjustment arithmetic on the device (float64).
//
//   * ba_eval_kernel     one thread per observation: pinhole residual + analytic 2x6 / 2x3 Jacobian blocks.
//                        The <= 8 camera rotations R(rvec) and the Rodrigues derivative factor M are built
//                        once per workgroup in LDS (oracle/trf_lsmr.py:rotation_and_dfactor).
//   * matvec / rmatvec   J v and J^T u on the block form.  J^T u needs a reduction over all observations of a
//                        camera: observations are pre-grouped by camera (cam_perm / cam_start), every
//                        (camera, chunk) workgroup reduces its slice in a fixed order, a second kernel sums the
//                        DF3D chunks in order -> bit-reproducible (no floating-point atomics).
//   * df3d_ba_lsmr       Fong & Saunders LSMR on A = J diag(d) with damping, the inner solver of scipy's
//                        trust-region-reflective step (oracle/trf_lsmr.py:lsmr).  Vectors stay on the device;
//                        three scalars (beta, alpha, |x|) come back to the host per iteration.
// All kernels are latency/launch-bound at the reference's sizes (1e5 observations): ~15 MB per Jacobian pass.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "ba_lsmr.h"

namespace {

constexpr int MAX_CAM = 8;
constexpr int NCHUNK = 32;      // chunks per camera in the J^T u / column-norm reductions
constexpr int RED_BLOCKS = 256;  // partial blocks of the generic sum-of-squares / dot reductions

struct CamLds {
    double R[MAX_CAM][9];
    double M[MAX_CAM][9];
    double t[MAX_CAM][3];
    double k[MAX_CAM][4];
};

__device__ void cam_prep(const double* __restrict__ x, const double* __restrict__ intr4, int ncam, CamLds& L) {
    const int c = threadIdx.x;
    if (c < ncam) {
        const double r0 = x[c * 6 + 0], r1 = x[c * 6 + 1], r2 = x[c * 6 + 2];
        const double th2 = r0 * r0 + r1 * r1 + r2 * r2;
        double R[9], M[9];
        if (th2 < 1e-24) {
            // first-order: R = I + [r]x, M = I
            R[0] = 1; R[1] = -r2; R[2] = r1;
            R[3] = r2; R[4] = 1; R[5] = -r0;
            R[6] = -r1; R[7] = r0; R[8] = 1;
            M[0] = 1; M[1] = 0; M[2] = 0; M[3] = 0; M[4] = 1; M[5] = 0; M[6] = 0; M[7] = 0; M[8] = 1;
        } else {
            const double th = sqrt(th2);
            const double k0 = r0 / th, k1 = r1 / th, k2 = r2 / th;
            const double s = sin(th), cm = 1.0 - cos(th);
            // R = I + s K + (1-c) K^2
            const double K[9] = {0, -k2, k1, k2, 0, -k0, -k1, k0, 0};
            double K2[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double acc = 0;
                    for (int l = 0; l < 3; ++l) acc += K[i * 3 + l] * K[l * 3 + j];
                    K2[i * 3 + j] = acc;
                }
            for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s * K[i] + cm * K2[i];
            // M = (r r^T + (R^T - I) [r]x) / |r|^2
            const double S[9] = {0, -r2, r1, r2, 0, -r0, -r1, r0, 0};
            const double rv[3] = {r0, r1, r2};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double acc = rv[i] * rv[j];
                    for (int l = 0; l < 3; ++l) acc += (R[l * 3 + i] - (l == i ? 1.0 : 0.0)) * S[l * 3 + j];
                    M[i * 3 + j] = acc / th2;
                }
        }
        for (int i = 0; i < 9; ++i) {
            L.R[c][i] = R[i];
            L.M[c][i] = M[i];
        }
        for (int i = 0; i < 3; ++i) L.t[c][i] = x[c * 6 + 3 + i];
        for (int i = 0; i < 4; ++i) L.k[c][i] = intr4[c * 4 + i];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void ba_eval_kernel(df3d_ba_problem p, const double* __restrict__ x,
                                                      double* __restrict__ r, double* __restrict__ Jc,
                                                      double* __restrict__ Jp) {
    __shared__ CamLds L;
    cam_prep(x, p.intr4, p.ncam, L);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.nobs) return;
    const int c = p.cam_idx[i];
    const int q = p.pt_idx[i];
    const double* Xp = x + 6 * p.ncam + 3 * (size_t)q;
    const double X0 = Xp[0], X1 = Xp[1], X2 = Xp[2];
    const double* R = L.R[c];
    const double xc = R[0] * X0 + R[1] * X1 + R[2] * X2 + L.t[c][0];
    const double yc = R[3] * X0 + R[4] * X1 + R[5] * X2 + L.t[c][1];
    const double zc = R[6] * X0 + R[7] * X1 + R[8] * X2 + L.t[c][2];
    const double fx = L.k[c][0], fy = L.k[c][1], cx = L.k[c][2], cy = L.k[c][3];
    const double iz = 1.0 / zc;
    if (r) {
        const double2 o = *reinterpret_cast<const double2*>(p.obs_xy + 2 * (size_t)i);
        double2 res;
        res.x = fx * xc * iz + cx - o.x;
        res.y = fy * yc * iz + cy - o.y;
        *reinterpret_cast<double2*>(r + 2 * (size_t)i) = res;
    }
    if (Jc && Jp) {
        // d pi / d Xc
        const double a00 = fx * iz, a02 = -fx * xc * iz * iz;
        const double a11 = fy * iz, a12 = -fy * yc * iz * iz;
        // point block: dpi * R
        const size_t n = (size_t)p.nobs;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            Jp[(0 * 3 + k) * n + i] = a00 * R[0 + k] + a02 * R[6 + k];
            Jp[(1 * 3 + k) * n + i] = a11 * R[3 + k] + a12 * R[6 + k];
        }
        // rotation block: dpi * (-R [X]x M)
        // G = [X]x M
        const double* M = L.M[c];
        double G[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            G[0 + k] = -X2 * M[3 + k] + X1 * M[6 + k];
            G[3 + k] = X2 * M[0 + k] - X0 * M[6 + k];
            G[6 + k] = -X1 * M[0 + k] + X0 * M[3 + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double d0 = -(R[0] * G[0 + k] + R[1] * G[3 + k] + R[2] * G[6 + k]);
            const double d1 = -(R[3] * G[0 + k] + R[4] * G[3 + k] + R[5] * G[6 + k]);
            const double d2 = -(R[6] * G[0 + k] + R[7] * G[3 + k] + R[8] * G[6 + k]);
            Jc[(0 * 6 + k) * n + i] = a00 * d0 + a02 * d2;
            Jc[(1 * 6 + k) * n + i] = a11 * d1 + a12 * d2;
        }
        // translation block: dpi
        Jc[(0 * 6 + 3) * n + i] = a00;
        Jc[(0 * 6 + 4) * n + i] = 0.0;
        Jc[(0 * 6 + 5) * n + i] = a02;
        Jc[(1 * 6 + 3) * n + i] = 0.0;
        Jc[(1 * 6 + 4) * n + i] = a11;
        Jc[(1 * 6 + 5) * n + i] = a12;
    }
}

__global__ __launch_bounds__(256) void ba_matvec_kernel(df3d_ba_problem p, const double* __restrict__ Jc,
                                                        const double* __restrict__ Jp,
                                                        const double* __restrict__ d,
                                                        const double* __restrict__ v, double* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.nobs) return;
    const size_t n = (size_t)p.nobs;
    const int c = p.cam_idx[i];
    const int q = p.pt_idx[i];
    double y0 = 0.0, y1 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int col = c * 6 + k;
        const double vk = d ? d[col] * v[col] : v[col];
        y0 += Jc[(0 * 6 + k) * n + i] * vk;
        y1 += Jc[(1 * 6 + k) * n + i] * vk;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t col = 6 * (size_t)p.ncam + 3 * (size_t)q + k;
        const double vk = d ? d[col] * v[col] : v[col];
        y0 += Jp[(0 * 3 + k) * n + i] * vk;
        y1 += Jp[(1 * 3 + k) * n + i] * vk;
    }
    double2 out;
    out.x = y0;
    out.y = y1;
    *reinterpret_cast<double2*>(y + 2 * (size_t)i) = out;
}

using df3d_lsmr::block_reduce_256;

// stage 1 of J^T u (SQUARE = false) or of the column norms (SQUARE = true), camera columns only:
// partial[(c * NCHUNK + chunk) * 6 + k]
template <bool SQUARE>
__global__ __launch_bounds__(256) void ba_cam_partial_kernel(df3d_ba_problem p, const double* __restrict__ Jc,
                                                             const double* __restrict__ u,
                                                             double* __restrict__ partial) {
    __shared__ double lds4[4];
    const int chunk = blockIdx.x, c = blockIdx.y;
    const int lo = p.cam_start[c], hi = p.cam_start[c + 1];
    const int cnt = hi - lo;
    const int per = (cnt + NCHUNK - 1) / NCHUNK;
    const int a = lo + chunk * per;
    const int b = min(a + per, hi);
    const size_t n = (size_t)p.nobs;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int s = a + (int)threadIdx.x; s < b; s += 256) {
        const int i = p.cam_perm[s];
        double u0 = 1.0, u1 = 1.0;
        if (!SQUARE) {
            const double2 uu = *reinterpret_cast<const double2*>(u + 2 * (size_t)i);
            u0 = uu.x;
            u1 = uu.y;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double j0 = Jc[(0 * 6 + k) * n + i], j1 = Jc[(1 * 6 + k) * n + i];
            acc[k] += SQUARE ? (j0 * j0 + j1 * j1) : (j0 * u0 + j1 * u1);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double tot = block_reduce_256(acc[k], lds4);
        if (threadIdx.x == 0) partial[((size_t)c * NCHUNK + chunk) * 6 + k] = tot;
    }
}

// stage 2: one thread per unknown
template <bool SQUARE>
__global__ __launch_bounds__(256) void ba_rmatvec_final_kernel(df3d_ba_problem p, const double* __restrict__ Jp,
                                                               const double* __restrict__ d,
                                                               const double* __restrict__ u,
                                                               const double* __restrict__ partial,
                                                               double* __restrict__ w) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long ncols = 6ll * p.ncam + 3ll * p.npts;
    if (k >= ncols) return;
    double acc = 0.0;
    if (k < 6 * p.ncam) {
        const int c = (int)(k / 6), col = (int)(k % 6);
        for (int ch = 0; ch < NCHUNK; ++ch) acc += partial[((size_t)c * NCHUNK + ch) * 6 + col];
    } else {
        const long long kk = k - 6 * p.ncam;
        const int q = (int)(kk / 3), col = (int)(kk % 3);
        const size_t n = (size_t)p.nobs;
        for (int i = p.pt_start[q]; i < p.pt_start[q + 1]; ++i) {
            const double j0 = Jp[(0 * 3 + col) * n + i], j1 = Jp[(1 * 3 + col) * n + i];
            if (SQUARE) {
                acc += j0 * j0 + j1 * j1;
            } else {
                acc += j0 * u[2 * (size_t)i] + j1 * u[2 * (size_t)i + 1];
            }
        }
    }
    w[k] = (d && !SQUARE) ? d[k] * acc : acc;
}

// ---- generic float64 vector helpers -------------------------------------------------------------
// out = a*x + b*y (y may be null); optionally accumulates sum(out^2) partials (fixed RED_BLOCKS blocks)
template <bool SUMSQ>
// (x, y and out may alias element-wise: no __restrict__ on them)
__global__ __launch_bounds__(256) void axpby_kernel(double a, const double* x, double b, const double* y,
                                                    double* out, size_t n, double* __restrict__ partial) {
    __shared__ double lds4[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = a * x[i];
        if (y) v += b * y[i];
        out[i] = v;
        if (SUMSQ) acc += v * v;
    }
    if (SUMSQ) {
        const double tot = block_reduce_256(acc, lds4);
        if (threadIdx.x == 0) partial[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(256) void dot_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                  size_t n, double* __restrict__ partial) {
    __shared__ double lds4[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i] * b[i];
    const double tot = block_reduce_256(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// up to 8 dot products in one launch (blockIdx.y = product), each with dot_kernel's grid and summation order
struct DotBatch {
    const double* a[8];
    const double* b[8];
    unsigned long long n[8];
    int g[8];
};
__global__ __launch_bounds__(256) void dots_kernel(DotBatch q, double* __restrict__ partial) {
    __shared__ double lds4[4];
    const int j = blockIdx.y;
    const int g = q.g[j];
    if ((int)blockIdx.x >= g) return;
    const double* __restrict__ a = q.a[j];
    const double* __restrict__ b = q.b[j];
    const size_t n = (size_t)q.n[j];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)g * 256) acc += a[i] * b[i];
    const double tot = block_reduce_256(acc, lds4);
    if (threadIdx.x == 0) partial[(size_t)j * RED_BLOCKS + blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void dots_final_kernel(DotBatch q, const double* __restrict__ partial, double* __restrict__ result) {
    __shared__ double lds4[4];
    const int j = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < q.g[j]; i += 256) acc += partial[(size_t)j * RED_BLOCKS + i];
    const double tot = block_reduce_256(acc, lds4);
    if (threadIdx.x == 0) result[j] = tot;
}

// partial sums of the Euclidean norms of n (x, y) pairs (reprojection error: mean pixel distance)
__global__ __launch_bounds__(256) void pairnorm_kernel(const double* __restrict__ r, size_t n, double* __restrict__ partial) {
    __shared__ double lds4[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 v = *reinterpret_cast<const double2*>(r + 2 * i);
        acc += sqrt(v.x * v.x + v.y * v.y);
    }
    const double tot = block_reduce_256(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// sums `count` partials in index order into result[0]
__global__ __launch_bounds__(256) void final_sum_kernel(const double* __restrict__ partial, int count,
                                                        double* __restrict__ result) {
    __shared__ double lds4[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial[i];
    const double tot = block_reduce_256(acc, lds4);
    if (threadIdx.x == 0) result[0] = tot;
}

__global__ __launch_bounds__(256) void mul_kernel(const double* x, const double* y, double* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = x[i] * y[i];
}

// LSMR vector update:  hbar = h - c1*hbar ; x += c2*hbar ; h = v - c3*h ; partial sum(x^2)
__global__ __launch_bounds__(256) void lsmr_update_kernel(double c1, double c2, double c3, double* __restrict__ hbar,
                                                          double* __restrict__ h, double* __restrict__ x,
                                                          const double* __restrict__ v, size_t n,
                                                          double* __restrict__ partial) {
    __shared__ double lds4[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double hi = h[i];
        const double hb = hi - c1 * hbar[i];
        const double xi = x[i] + c2 * hb;
        hbar[i] = hb;
        x[i] = xi;
        h[i] = v[i] - c3 * hi;
        acc += xi * xi;
    }
    const double tot = block_reduce_256(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// x_scale='jac' bookkeeping: scale_inv = sqrt(colsq) (zeros -> 1 on the first call, running max later)
__global__ __launch_bounds__(256) void update_scale_kernel(const double* __restrict__ colsq, double* __restrict__ scale_inv,
                                                           double* __restrict__ scale, size_t n, int first) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double si = sqrt(colsq[i]);
        if (first) {
            if (si == 0.0) si = 1.0;
        } else {
            si = fmax(si, scale_inv[i]);
        }
        scale_inv[i] = si;
        scale[i] = 1.0 / si;
    }
}

__global__ __launch_bounds__(256) void absmax_kernel(const double* __restrict__ a, size_t n, double* __restrict__ partial) {
    __shared__ double lds[256];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc = fmax(acc, fabs(a[i]));
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) lds[threadIdx.x] = fmax(lds[threadIdx.x], lds[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = lds[0];
}

// ---- LSMR vector kernels that take their coefficients from the device-resident state (no-ops once it has stopped) ----
// WHICH = 0:  u = t - alpha * u   (t = A v);   WHICH = 1:  v = t - beta * v   (t = A^T u, only when beta > 0)
template <int WHICH>
__global__ __launch_bounds__(256) void lsmr_bidiag_kernel(const df3d_lsmr::State* __restrict__ st, const double* __restrict__ t,
                                                          double* __restrict__ y, size_t n, double* __restrict__ partial) {
    __shared__ double lds4[4];
    if (st->istop || (WHICH == 1 && !st->beta_pos)) return;
    const double b = WHICH == 0 ? -st->alpha : -st->beta;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = 1.0 * t[i];
        v += b * y[i];
        y[i] = v;
        acc += v * v;
    }
    const double tot = block_reduce_256(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// WHICH = 0:  u *= 1/beta;   WHICH = 1:  v *= 1/alpha   (both only when beta > 0)
template <int WHICH>
__global__ __launch_bounds__(256) void lsmr_scale_kernel(const df3d_lsmr::State* __restrict__ st, double* __restrict__ x, size_t n) {
    if (st->istop || !st->beta_pos) return;
    const double a = WHICH == 0 ? st->inv_beta : st->inv_alpha;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = a * x[i];
}

// hbar = h - c1*hbar ; x += c2*hbar ; h = v - c3*h ; partial sum(x^2)
__global__ __launch_bounds__(256) void lsmr_update_dev_kernel(const df3d_lsmr::State* __restrict__ st, double* __restrict__ hbar,
                                                              double* __restrict__ h, double* __restrict__ x,
                                                              const double* __restrict__ v, size_t n, double* __restrict__ partial) {
    __shared__ double lds4[4];
    if (st->istop) return;
    const double c1 = st->c1, c2 = st->c2, c3 = st->c3;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double hi = h[i];
        const double hb = hi - c1 * hbar[i];
        const double xi = x[i] + c2 * hb;
        hbar[i] = hb;
        x[i] = xi;
        h[i] = v[i] - c3 * hi;
        acc += xi * xi;
    }
    const double tot = block_reduce_256(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > RED_BLOCKS ? RED_BLOCKS : b));
}

int check_problem(const df3d_ba_problem* p) {
    DF3D_CHECK_ARG(p != nullptr, "null problem");
    DF3D_CHECK_ARG(p->ncam >= 1 && p->ncam <= MAX_CAM, "ncam must be in [1, 8]");
    DF3D_CHECK_ARG(p->nobs > 0 && p->npts > 0, "empty problem");
    DF3D_CHECK_ARG(p->intr4 && p->obs_xy && p->cam_idx && p->pt_idx && p->pt_start && p->cam_perm && p->cam_start,
                   "null array in problem");
    return DF3D_OK;
}

// host-visible reduction result: sum of squares / dot -> host double (synchronous)
int read_back(const double* dev_scalar, double* host, hipStream_t s) {
    DF3D_HIP(hipMemcpyAsync(host, dev_scalar, sizeof(double), hipMemcpyDeviceToHost, s));
    DF3D_HIP(hipStreamSynchronize(s));
    return DF3D_OK;
}

int launch_rmatvec(const df3d_ba_problem* p, const double* Jc, const double* Jp, const double* d, const double* u,
                   double* w, double* scratch, hipStream_t s) {
    hipLaunchKernelGGL(ba_cam_partial_kernel<false>, dim3(NCHUNK, p->ncam), dim3(256), 0, s, *p, Jc, u, scratch);
    const long long ncols = 6ll * p->ncam + 3ll * p->npts;
    hipLaunchKernelGGL(ba_rmatvec_final_kernel<false>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, *p, Jp,
                       d, u, scratch, w);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int launch_matvec(const df3d_ba_problem* p, const double* Jc, const double* Jp, const double* d, const double* v,
                  double* y, hipStream_t s) {
    hipLaunchKernelGGL(ba_matvec_kernel, dim3((p->nobs + 255) / 256), dim3(256), 0, s, *p, Jc, Jp, d, v, y);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

void sym_ortho(double a, double b, double& c, double& s, double& r) {
    auto sgn = [](double v) { return (v > 0) - (v < 0); };
    if (b == 0) {
        c = sgn(a);
        s = 0;
        r = std::fabs(a);
    } else if (a == 0) {
        c = 0;
        s = sgn(b);
        r = std::fabs(b);
    } else if (std::fabs(b) > std::fabs(a)) {
        const double tau = a / b;
        s = sgn(b) / std::sqrt(1 + tau * tau);
        c = s * tau;
        r = b / s;
    } else {
        const double tau = b / a;
        c = sgn(a) / std::sqrt(1 + tau * tau);
        s = c * tau;
        r = a / c;
    }
}

}  // namespace

extern "C" {

int df3d_ba_eval(const df3d_ba_problem* p, const double* x_dev, double* r_dev, double* Jc_dev, double* Jp_dev,
                 void* stream) {
    if (int rc = check_problem(p)) return rc;
    DF3D_CHECK_ARG(x_dev != nullptr, "null x");
    DF3D_CHECK_ARG((Jc_dev == nullptr) == (Jp_dev == nullptr), "Jc and Jp must be given together");
    hipLaunchKernelGGL(ba_eval_kernel, dim3((p->nobs + 255) / 256), dim3(256), 0, df3d::as_stream(stream), *p, x_dev,
                       r_dev, Jc_dev, Jp_dev);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int df3d_ba_colsq(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, double* colsq_dev,
                  double* scratch_dev, void* stream) {
    if (int rc = check_problem(p)) return rc;
    DF3D_CHECK_ARG(Jc_dev && Jp_dev && colsq_dev && scratch_dev, "null pointer");
    hipStream_t s = df3d::as_stream(stream);
    hipLaunchKernelGGL(ba_cam_partial_kernel<true>, dim3(NCHUNK, p->ncam), dim3(256), 0, s, *p, Jc_dev, nullptr,
                       scratch_dev);
    const long long ncols = 6ll * p->ncam + 3ll * p->npts;
    hipLaunchKernelGGL(ba_rmatvec_final_kernel<true>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, *p,
                       Jp_dev, nullptr, nullptr, scratch_dev, colsq_dev);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int df3d_ba_matvec(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, const double* d_dev,
                   const double* v_dev, double* y_dev, void* stream) {
    if (int rc = check_problem(p)) return rc;
    DF3D_CHECK_ARG(Jc_dev && Jp_dev && v_dev && y_dev, "null pointer");
    return launch_matvec(p, Jc_dev, Jp_dev, d_dev, v_dev, y_dev, df3d::as_stream(stream));
}

int df3d_ba_rmatvec(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, const double* d_dev,
                    const double* u_dev, double* w_dev, double* scratch_dev, void* stream) {
    if (int rc = check_problem(p)) return rc;
    DF3D_CHECK_ARG(Jc_dev && Jp_dev && u_dev && w_dev && scratch_dev, "null pointer");
    return launch_rmatvec(p, Jc_dev, Jp_dev, d_dev, u_dev, w_dev, scratch_dev, df3d::as_stream(stream));
}

int df3d_vec_dot(const double* a_dev, const double* b_dev, size_t n, double* result_host, double* scratch_dev,
                 void* stream) {
    DF3D_CHECK_ARG(a_dev && b_dev && result_host && scratch_dev, "null pointer");
    hipStream_t s = df3d::as_stream(stream);
    const int g = grid_for(n);
    hipLaunchKernelGGL(dot_kernel, dim3(g), dim3(256), 0, s, a_dev, b_dev, n, scratch_dev);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, scratch_dev, g, scratch_dev + RED_BLOCKS);
    DF3D_LAUNCH_CHECK();
    return read_back(scratch_dev + RED_BLOCKS, result_host, s);
}

int df3d_vec_dots(int count, const double* const* a_dev, const double* const* b_dev, const size_t* n, double* results_host, double* scratch_dev,
                  void* stream) {
    DF3D_CHECK_ARG(count >= 1 && count <= 8 && a_dev && b_dev && n && results_host && scratch_dev, "1..8 products, no null pointer");
    hipStream_t s = df3d::as_stream(stream);
    DotBatch q{};
    int gmax = 1;
    for (int j = 0; j < count; ++j) {
        DF3D_CHECK_ARG(a_dev[j] && b_dev[j], "null vector");
        q.a[j] = a_dev[j];
        q.b[j] = b_dev[j];
        q.n[j] = n[j];
        q.g[j] = grid_for(n[j]);
        gmax = q.g[j] > gmax ? q.g[j] : gmax;
    }
    double* const result = scratch_dev + 8 * RED_BLOCKS;
    hipLaunchKernelGGL(dots_kernel, dim3(gmax, count), dim3(256), 0, s, q, scratch_dev);
    hipLaunchKernelGGL(dots_final_kernel, dim3(count), dim3(256), 0, s, q, scratch_dev, result);
    DF3D_LAUNCH_CHECK();
    DF3D_HIP(hipMemcpyAsync(results_host, result, count * sizeof(double), hipMemcpyDeviceToHost, s));
    DF3D_HIP(hipStreamSynchronize(s));
    return DF3D_OK;
}

int df3d_vec_pairnorm_sum(const double* r_dev, size_t npairs, double* result_host, double* scratch_dev, void* stream) {
    DF3D_CHECK_ARG(r_dev && result_host && scratch_dev, "null pointer");
    hipStream_t s = df3d::as_stream(stream);
    const int g = grid_for(npairs);
    hipLaunchKernelGGL(pairnorm_kernel, dim3(g), dim3(256), 0, s, r_dev, npairs, scratch_dev);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, scratch_dev, g, scratch_dev + RED_BLOCKS);
    DF3D_LAUNCH_CHECK();
    return read_back(scratch_dev + RED_BLOCKS, result_host, s);
}

int df3d_vec_axpby(double a, const double* x_dev, double b, const double* y_dev, double* out_dev, size_t n,
                   void* stream) {
    DF3D_CHECK_ARG(x_dev && out_dev, "null pointer");
    hipLaunchKernelGGL(axpby_kernel<false>, dim3(grid_for(n)), dim3(256), 0, df3d::as_stream(stream), a, x_dev, b,
                       y_dev, out_dev, n, nullptr);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int df3d_vec_mul(const double* x_dev, const double* y_dev, double* out_dev, size_t n, void* stream) {
    DF3D_CHECK_ARG(x_dev && y_dev && out_dev, "null pointer");
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, df3d::as_stream(stream), x_dev, y_dev, out_dev, n);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int df3d_vec_absmax(const double* a_dev, size_t n, double* result_host, double* scratch_dev, void* stream) {
    DF3D_CHECK_ARG(a_dev && result_host && scratch_dev, "null pointer");
    hipStream_t s = df3d::as_stream(stream);
    const int g = grid_for(n);
    hipLaunchKernelGGL(absmax_kernel, dim3(g), dim3(256), 0, s, a_dev, n, scratch_dev);
    hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(256), 0, s, scratch_dev, (size_t)g, scratch_dev + RED_BLOCKS);
    DF3D_LAUNCH_CHECK();
    return read_back(scratch_dev + RED_BLOCKS, result_host, s);
}

int df3d_ba_update_scale(const double* colsq_dev, double* scale_inv_dev, double* scale_dev, size_t n, int first,
                         void* stream) {
    DF3D_CHECK_ARG(colsq_dev && scale_inv_dev && scale_dev, "null pointer");
    hipLaunchKernelGGL(update_scale_kernel, dim3(grid_for(n)), dim3(256), 0, df3d::as_stream(stream), colsq_dev,
                       scale_inv_dev, scale_dev, n, first);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

size_t df3d_ba_lsmr_work_doubles(const df3d_ba_problem* p) {
    if (!p) return 0;
    const size_t m = 2 * (size_t)p->nobs, n = 6 * (size_t)p->ncam + 3 * (size_t)p->npts;
    // u, tmp_m (m each); v, h, hbar, tmp_n (n each); scratch; round 4 (fused iteration): two state slots, |u|^2 / |x|^2 partials, |v|^2 partials
    // round 5 (persistent run): 8 doubles for the grid barrier's two words
    //          + the data-local form's ranges, granules and state
    return 2 * m + 4 * n + DF3D_BA_SCRATCH_DOUBLES + 64 + 2 * df3d_lsmr::FUSED_DOUBLES + 3 * df3d_lsmr::FUSED_RED + 8 + (df3d_lsmr::local_scratch_bytes() + 7) / 8 + 64;
}

}  // extern "C"

namespace {
// form: 0 = one persistent kernel per run (round 5, the default), 2 = two kernels per iteration (round 4), 11 = round 3's eleven
int lsmr_run(const df3d_ba_problem* p, const double* Jc, const double* Jp, const double* d_dev,
             const double* b_dev, double damp, double atol, double btol, double conlim, int maxiter,
             double* x_dev, double* work_dev, double* info_host, void* stream, int form) {
    if (int rc = check_problem(p)) return rc;
    DF3D_CHECK_ARG(Jc && Jp && b_dev && x_dev && work_dev && info_host, "null pointer");
    hipStream_t s = df3d::as_stream(stream);
    const size_t m = 2 * (size_t)p->nobs, n = 6 * (size_t)p->ncam + 3 * (size_t)p->npts;
    if (maxiter <= 0) maxiter = (int)(m < n ? m : n);
    if (form == 3) {
        // round 5: the data-local run (ba_lsmr.hip: lsmr_local_kernel) -- the whole solve, its set-up included, in ONE launch and ONE read-back
        double* const lscratch = work_dev + (2 * m + 4 * n + DF3D_BA_SCRATCH_DOUBLES + 64 + 2 * df3d_lsmr::FUSED_DOUBLES + 3 * df3d_lsmr::FUSED_RED + 8);
        double* const lstate = lscratch + (df3d_lsmr::local_scratch_bytes() + 7) / 8;
        const int rc = df3d_lsmr::launch_local(*p, Jc, Jp, d_dev, b_dev, x_dev, damp, atol, btol, conlim > 0 ? 1.0 / conlim : 0.0, maxiter, lscratch, lstate, s);
        if (rc < 0) {
            info_host[0] = -2;   // does not fit this form: the caller takes another
            return DF3D_OK;
        }
        DF3D_LAUNCH_CHECK();
        df3d_lsmr::State now{};
        DF3D_HIP(hipMemcpyAsync(&now, lstate, sizeof(now), hipMemcpyDeviceToHost, s));
        DF3D_HIP(hipStreamSynchronize(s));
        info_host[0] = now.istop;
        info_host[1] = now.itn;
        info_host[2] = now.normr;
        info_host[3] = now.normar;
        info_host[4] = now.normA;
        info_host[5] = now.condA;
        info_host[6] = now.normx;
        info_host[7] = 0;
        return DF3D_OK;
    }
    double* u = work_dev;
    double* tmp_m = u + m;
    double* v = tmp_m + m;
    double* h = v + n;
    double* hbar = h + n;
    double* tmp_n = hbar + n;
    double* scratch = tmp_n + n;                  // DF3D_BA_SCRATCH_DOUBLES: camera partials
    double* red = scratch + 2048;                 // RED_BLOCKS partials + result slot
    double* result = red + RED_BLOCKS;
    const int gm = grid_for(m), gn = grid_for(n);

    auto sumsq_axpby = [&](double a, const double* x, double bb, const double* y, double* out, size_t len, int g,
                           double* host) -> int {
        hipLaunchKernelGGL(axpby_kernel<true>, dim3(g), dim3(256), 0, s, a, x, bb, y, out, len, red);
        hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, red, g, result);
        DF3D_LAUNCH_CHECK();
        return read_back(result, host, s);
    };

    double ss = 0.0;
    // u = b ; normb
    if (int rc = sumsq_axpby(1.0, b_dev, 0.0, nullptr, u, m, gm, &ss)) return rc;
    const double normb = std::sqrt(ss);
    double beta = normb, alpha = 0.0;
    DF3D_HIP(hipMemsetAsync(x_dev, 0, n * sizeof(double), s));
    DF3D_HIP(hipMemsetAsync(hbar, 0, n * sizeof(double), s));
    if (beta > 0) {
        if (int rc = df3d_vec_axpby(1.0 / beta, u, 0.0, nullptr, u, m, stream)) return rc;
        if (int rc = launch_rmatvec(p, Jc, Jp, d_dev, u, tmp_n, scratch, s)) return rc;
        if (int rc = sumsq_axpby(1.0, tmp_n, 0.0, nullptr, v, n, gn, &ss)) return rc;
        alpha = std::sqrt(ss);
    } else {
        DF3D_HIP(hipMemsetAsync(v, 0, n * sizeof(double), s));
    }
    if (alpha > 0)
        if (int rc = df3d_vec_axpby(1.0 / alpha, v, 0.0, nullptr, v, n, stream)) return rc;

    // ---- iterations: every scalar of the recurrence lives in `st` on the device; the host only enqueues kernels and
    // looks at (istop, itn) once per CHUNK iterations (kernels of iterations past the stop are no-ops), instead of three
    // synchronous read-backs per iteration
    df3d_lsmr::State init{};
    init.alpha = alpha;
    init.beta = beta;
    init.rho = init.rhobar = init.cbar = 1;
    init.sbar = 0;
    init.zeta = 0;
    init.zetabar = alpha * beta;
    init.alphabar = alpha;
    init.betadd = beta;
    init.betad = 0;
    init.rhodold = 1;
    init.tautildeold = init.thetatilde = init.dd = 0;
    init.normA2 = alpha * alpha;
    init.maxrbar = 0;
    init.minrbar = 1e100;
    init.normA = std::sqrt(init.normA2);
    init.condA = 1;
    init.normx = 0;
    init.normr = beta;
    init.normar = alpha * beta;
    init.normb = normb;
    init.damp = damp;
    init.atol = atol;
    init.btol = btol;
    init.ctol = conlim > 0 ? 1.0 / conlim : 0.0;
    init.inv_beta = init.inv_alpha = 1.0;
    init.itn = 0;
    init.istop = 0;
    init.maxiter = maxiter;
    init.beta_pos = 1;
    DF3D_HIP(hipMemcpyAsync(h, v, n * sizeof(double), hipMemcpyDeviceToDevice, s));

    auto finish = [&](const df3d_lsmr::State& f) {
        info_host[0] = f.istop;
        info_host[1] = f.itn;
        info_host[2] = f.normr;
        info_host[3] = f.normar;
        info_host[4] = f.normA;
        info_host[5] = f.condA;
        info_host[6] = f.normx;
        info_host[7] = 0;
        return DF3D_OK;
    };
    if (init.normar == 0 || normb == 0) {
        DF3D_HIP(hipStreamSynchronize(s));
        return finish(init);
    }
    // round 4: the iteration is TWO kernels (ba_lsmr.hip: fused_ka / fused_kb; the scalar steps run in every workgroup's prologue, the
    // state alternates between two slots, u and v stay un-normalised with 1 / beta, 1 / alpha in the state) instead of eleven
    double* const fbase = work_dev + (2 * m + 4 * n + DF3D_BA_SCRATCH_DOUBLES + 64);
    df3d_lsmr::FusedArgs fa{};
    fa.Jc = Jc; fa.Jp = Jp; fa.d = d_dev;
    fa.u = u; fa.v = v; fa.h = h; fa.hbar = hbar; fa.x = x_dev;
    fa.cam_partial = scratch;
    fa.st = fbase;
    fa.red1 = fbase + 2 * df3d_lsmr::FUSED_DOUBLES;
    fa.red3 = fa.red1 + df3d_lsmr::FUSED_RED;
    fa.red2 = fa.red3 + df3d_lsmr::FUSED_RED;
    fa.nchunk = NCHUNK;
    fa.g1 = gm;    // round 3's grids: the grouping of the sums of squares is part of the arithmetic
    fa.g2p = gn;
    fa.g3 = gn;
    df3d_lsmr::Fused finit{};
    finit.s = init;
    finit.pending_c = 0;
    finit.pending_b = 0;
    DF3D_HIP(hipMemcpyAsync(fa.st, &finit, sizeof(finit), hipMemcpyHostToDevice, s));
    DF3D_HIP(hipMemcpyAsync(reinterpret_cast<unsigned char*>(fa.st) + offsetof(df3d_lsmr::Fused, vcam), v, 6 * (size_t)p->ncam * sizeof(double),
                            hipMemcpyDeviceToDevice, s));
    DF3D_HIP(hipStreamSynchronize(s));  // `finit` is on the stack

    // round 3's form (eleven kernels per iteration, one state, u and v normalised in place) stays selectable for A/B runs and as the
    // arithmetic reference: DF3D_LSMR_KERNELS=11 in the environment
    const bool r3_form = form == 11;
    static_assert(sizeof(df3d_lsmr::State) <= 64 * sizeof(double), "state must fit behind the reduction scratch");
    df3d_lsmr::State* st = reinterpret_cast<df3d_lsmr::State*>(result + 8);
    if (r3_form) {
        DF3D_HIP(hipMemcpyAsync(st, &init, sizeof(init), hipMemcpyHostToDevice, s));
        DF3D_HIP(hipStreamSynchronize(s));
    }

    if (form == 0) {
        // round 5: the whole run in ONE launch (ba_lsmr.hip: fused_persistent) -- grid-wide barriers where the two-kernel form has kernel
        // boundaries, ONE read-back.  Grid: a quarter of the largest phase's virtual workgroups, 8..128 (DF3D_LSMR_GRID overrides): the
        // barrier's price grows with the arrivals, the phases' time shrinks with them; measured in profiles/r05_ba_timings.txt
        const char* const ge = getenv("DF3D_LSMR_GRID");
        const int grid_env = ge ? atoi(ge) : 0;
        const int vmax = std::max(std::max(fa.g1, fa.g3), p->ncam * fa.nchunk + fa.g2p);
        int grid = grid_env > 0 ? grid_env : std::min(128, std::max(8, (vmax + 3) / 4));
        grid = std::min(grid, 256);
        unsigned* const bar = reinterpret_cast<unsigned*>(fa.red2 + df3d_lsmr::FUSED_RED);
        df3d_lsmr::launch_fused_persistent(*p, fa, bar, maxiter, grid, s);
        DF3D_LAUNCH_CHECK();
        df3d_lsmr::State now = init;
        DF3D_HIP(hipMemcpyAsync(&now, fa.st, sizeof(now), hipMemcpyDeviceToHost, s));
        DF3D_HIP(hipStreamSynchronize(s));
        return finish(now);
    }

    constexpr int CHUNK = 16;
    df3d_lsmr::State now = init;
    auto enqueue_iteration = [&]() -> int {
        if (r3_form) {
            // u = A v - alpha u ; beta = |u| ; u /= beta
            if (int rc = launch_matvec(p, Jc, Jp, d_dev, v, tmp_m, s)) return rc;
            hipLaunchKernelGGL(lsmr_bidiag_kernel<0>, dim3(gm), dim3(256), 0, s, st, tmp_m, u, m, red);
            df3d_lsmr::launch_step_a(st, red, gm, s);
            hipLaunchKernelGGL(lsmr_scale_kernel<0>, dim3(gm), dim3(256), 0, s, st, u, m);
            // v = A^T u - beta v ; alpha = |v| ; rotations ; v /= alpha
            if (int rc = launch_rmatvec(p, Jc, Jp, d_dev, u, tmp_n, scratch, s)) return rc;
            hipLaunchKernelGGL(lsmr_bidiag_kernel<1>, dim3(gn), dim3(256), 0, s, st, tmp_n, v, n, red);
            df3d_lsmr::launch_step_b(st, red, gn, s);
            hipLaunchKernelGGL(lsmr_scale_kernel<1>, dim3(gn), dim3(256), 0, s, st, v, n);
            // hbar, x, h ; |x| ; stopping tests
            hipLaunchKernelGGL(lsmr_update_dev_kernel, dim3(gn), dim3(256), 0, s, st, hbar, h, x_dev, v, n, red);
            df3d_lsmr::launch_step_c(st, red, gn, s);
            return DF3D_OK;
        }
        df3d_lsmr::launch_fused_iteration(*p, fa, s);   // (ka: state slot 0 -> 1, kb: 1 -> 0)
        DF3D_LAUNCH_CHECK();
        return DF3D_OK;
    };
    // One chunk of CHUNK iterations = 2 x CHUNK small dependent kernels (round 3: 11 x CHUNK): launch-bound when enqueued one by one.  On a
    // capturable stream (not the legacy default stream) the chunk is recorded once into a HIP graph and replayed -- every
    // scalar the kernels need lives in `st`, so the recording is valid for every chunk of every LSMR run on the same problem
    // and buffers (the three or four runs of one trust-region solve); otherwise the kernels are enqueued directly.
    // The recorded kernels take *p BY VALUE, so the key is every field of the problem (all index tables and the ncam / nobs /
    // npts split, not only their sum) plus every buffer the chunk touches: a recording is replayed only against exactly the
    // arguments it was made with.
    constexpr int NKEY = 12;
    struct ChunkGraph {
        hipGraphExec_t exec = nullptr;
        const void* key[NKEY] = {};
        long long dims[5] = {0, 0, 0, 0, 0};
    };
    static thread_local ChunkGraph cache;
    const void* key[NKEY] = {p->obs_xy, Jc, Jp, d_dev, x_dev, work_dev, p->cam_idx, p->pt_idx, p->intr4, p->pt_start, p->cam_perm, p->cam_start};
    const long long dims[5] = {(long long)m, (long long)n, p->ncam, p->nobs, p->npts};
    bool use_graph = s != nullptr && maxiter >= CHUNK;
    if (use_graph) {
        bool same = cache.exec != nullptr;
        for (int k = 0; k < 5 && same; ++k) same = cache.dims[k] == dims[k];
        for (int k = 0; k < NKEY && same; ++k) same = cache.key[k] == key[k];
        if (!same) {
            if (cache.exec) (void)hipGraphExecDestroy(cache.exec);
            cache.exec = nullptr;
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                int rc = DF3D_OK;
                for (int it = 0; it < CHUNK && rc == DF3D_OK; ++it) rc = enqueue_iteration();
                const hipError_t e = hipStreamEndCapture(s, &graph);
                if (rc == DF3D_OK && e == hipSuccess && graph && hipGraphInstantiate(&cache.exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    for (int k = 0; k < NKEY; ++k) cache.key[k] = key[k];
                    for (int k = 0; k < 5; ++k) cache.dims[k] = dims[k];
                } else {
                    cache.exec = nullptr;
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();   // a refused capture is not an error of this call: the direct path below takes over
            if (!cache.exec) use_graph = false;
        }
    }
    for (int done = 0; done < maxiter && now.istop == 0; done += CHUNK) {
        const int todo = maxiter - done < CHUNK ? maxiter - done : CHUNK;
        if (use_graph && todo == CHUNK) {
            DF3D_HIP(hipGraphLaunch(cache.exec, s));
        } else {
            for (int it = 0; it < todo; ++it)
                if (int rc = enqueue_iteration()) return rc;
            DF3D_LAUNCH_CHECK();
        }
        DF3D_HIP(hipMemcpyAsync(&now, r3_form ? reinterpret_cast<const double*>(st) : fa.st, sizeof(now), hipMemcpyDeviceToHost,
                                s));   // (two-kernel form: an iteration leaves the state in slot 0; State leads Fused)
        DF3D_HIP(hipStreamSynchronize(s));
    }
    if (now.istop == 0 && !r3_form) {
        // maxiter reached: the last iteration's steps B and C (and its update of x) are still pending -- one more ka and the scalar half of
        // kb take them; a run that stopped earlier was settled by the kernels behind its last iteration
        df3d_lsmr::launch_fused_flush(*p, fa, s);
        DF3D_HIP(hipMemcpyAsync(&now, fa.st, sizeof(now), hipMemcpyDeviceToHost, s));
        DF3D_HIP(hipStreamSynchronize(s));
    }
    return finish(now);
}
}  // namespace

extern "C" {

int df3d_ba_lsmr_form(const df3d_ba_problem* p, const double* Jc, const double* Jp, const double* d_dev,
                      const double* b_dev, double damp, double atol, double btol, double conlim, int maxiter,
                      double* x_dev, double* work_dev, double* info_host, void* stream, int form_arg) {
    DF3D_CHECK_ARG(form_arg == DF3D_LSMR_AUTO || form_arg == DF3D_LSMR_BARRIERS || form_arg == DF3D_LSMR_LAUNCHES || form_arg == DF3D_LSMR_LOCAL ||
                       form_arg == DF3D_LSMR_ELEVEN, "unknown LSMR form");
    // AUTO: the environment may name a form (A/B runs; tests/test_gpu_ba.py compares them): DF3D_LSMR_KERNELS = 0 (AUTO) | 1 | 2 | 11
    int want = form_arg;
    if (want == DF3D_LSMR_AUTO) {
        const char* e = getenv("DF3D_LSMR_KERNELS");
        const int k = e ? atoi(e) : 0;
        want = k == 11 ? DF3D_LSMR_ELEVEN : k == 2 ? DF3D_LSMR_LAUNCHES : k == 1 ? DF3D_LSMR_BARRIERS : DF3D_LSMR_LOCAL;
    }
    const int form = want == DF3D_LSMR_ELEVEN ? 11 : want == DF3D_LSMR_LAUNCHES ? 2 : want == DF3D_LSMR_BARRIERS ? 0 : 3;   // (lsmr_run's numbering)
    int rc = lsmr_run(p, Jc, Jp, d_dev, b_dev, damp, atol, btol, conlim, maxiter, x_dev, work_dev, info_host, stream, form);
    if (rc == DF3D_OK && (form == 0 || form == 3) && info_host[0] < 0) {
        // the problem does not fit the data-local form, or a persistent kernel timed out waiting for its peers (its workgroups were not
        // all resident: a device full of other persistent work): the run is repeated from its inputs in the two-kernel form, which
        // needs no co-residency
        const bool timeout = info_host[0] == -1;
        rc = lsmr_run(p, Jc, Jp, d_dev, b_dev, damp, atol, btol, conlim, maxiter, x_dev, work_dev, info_host, stream, 2);
        if (rc == DF3D_OK) info_host[7] = timeout ? 1 : 2;   // (reported: the fallback was taken, and why)
    }
    return rc;
}

int df3d_ba_lsmr(const df3d_ba_problem* p, const double* Jc, const double* Jp, const double* d_dev,
                 const double* b_dev, double damp, double atol, double btol, double conlim, int maxiter,
                 double* x_dev, double* work_dev, double* info_host, void* stream) {
    return df3d_ba_lsmr_form(p, Jc, Jp, d_dev, b_dev, damp, atol, btol, conlim, maxiter, x_dev, work_dev, info_host, stream, DF3D_LSMR_AUTO);
}

}  // extern "C"

// ---- round 6: the trust-region driver's scalars stay on the device ------------------------------------------------------------------
// One outer iteration of bundle_adjust.py:solve_trf used to read back seven groups of scalars (|g|_inf; the Cauchy model's two dots; the
// LSMR state; r01; |s1|; the 2x2 model's five dots; the trial step's six): each a stream synchronisation with the device idle behind it.
// df3d_ba_trf_subspace enqueues everything between "g is known" and "the 2-D model is known" -- the damping, the LSMR solve that reads it
// from device memory, the QR of [g_h, gn_h], J_h S and the model's dots -- and reads ONE block back; df3d_ba_trf_trial is a trial step with
// its one read-back; df3d_ba_trf_linearize the re-linearisation (no read-back).  The arithmetic is the Python driver's, operation for
// operation (same kernels for every vector operation and reduction; the scalar expressions below are written without contraction), so
// the iterate sequence -- nfev, LSMR counts, every bit of x -- is unchanged (tests/test_gpu_ba.py compares the two drivers).
namespace {
enum { TR_GNORM = 0, TR_JG2, TR_GH2, TR_DAMP, TR_CS0, TR_R01, TR_MR01, TR_N1SQ, TR_CS1, TR_B00, TR_B01, TR_B11, TR_GS0, TR_GS1, TR_ONE, TR_SLOTS };
constexpr int TR_OFFSET = 8 * RED_BLOCKS + 16;   // the block's place in scratch_dev (behind the dots' partials and results)
static_assert(TR_OFFSET + TR_SLOTS <= DF3D_BA_SCRATCH_DOUBLES, "scratch");

// out = (*ca) x + (*cb) y (y / cb may be null): axpby_kernel with its coefficients in device memory
__global__ __launch_bounds__(256) void axpby_dev_kernel(const double* __restrict__ ca, const double* x, const double* __restrict__ cb, const double* y, double* out, size_t n) {
    const double a = *ca, b = y ? *cb : 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = a * x[i];
        if (y) v += b * y[i];
        out[i] = v;
    }
}
// WHICH 0: the Tikhonov term of the 1-D Cauchy model along -g_h (solve_trf: `damp`) and the scale of s0;  1: -r01;  2: the scale of s1
template <int WHICH>
__global__ void trf_scalar_kernel(double* __restrict__ tr, double Delta) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (WHICH == 0) {
        const double jg2 = tr[TR_JG2], gh2 = tr[TR_GH2];
        const double a = 0.5 * jg2, b = -gh2;
        const double n0 = sqrt(gh2);
        const double to_tr = Delta / n0;
        double ag = fmin(0.0 * (a * 0.0 + b), to_tr * (a * to_tr + b));
        if (a != 0) {
            const double ext = -0.5 * b / a;
            if (0 < ext && ext < to_tr) ag = fmin(ag, ext * (a * ext + b));
        }
        const double reg = -ag / (Delta * Delta);
        tr[TR_DAMP] = n0 > 0 ? sqrt(reg) : 0.0;   // (g == 0: the caller stops on |g|_inf before it looks at anything else)
        tr[TR_CS0] = n0 > 0 ? -1.0 / n0 : 0.0;
        tr[TR_ONE] = 1.0;
    } else if (WHICH == 1) {
        tr[TR_MR01] = -tr[TR_R01];
    } else {
        tr[TR_CS1] = -1.0 / sqrt(tr[TR_N1SQ]);
    }
}

int enqueue_dots(int count, const double* const* a, const double* const* b, const size_t* n, double* scratch, double* result, hipStream_t s) {
    DotBatch q{};
    int gmax = 1;
    for (int j = 0; j < count; ++j) {
        q.a[j] = a[j];
        q.b[j] = b[j];
        q.n[j] = n[j];
        q.g[j] = grid_for(n[j]);
        gmax = q.g[j] > gmax ? q.g[j] : gmax;
    }
    hipLaunchKernelGGL(dots_kernel, dim3(gmax, count), dim3(256), 0, s, q, scratch);
    hipLaunchKernelGGL(dots_final_kernel, dim3(count), dim3(256), 0, s, q, scratch, result);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}
int resolve_form(int form_arg) {   // df3d_ba_lsmr_form's rule: lsmr_run's numbering
    int want = form_arg;
    if (want == DF3D_LSMR_AUTO) {
        const char* e = getenv("DF3D_LSMR_KERNELS");
        const int k = e ? atoi(e) : 0;
        want = k == 11 ? DF3D_LSMR_ELEVEN : k == 2 ? DF3D_LSMR_LAUNCHES : k == 1 ? DF3D_LSMR_BARRIERS : DF3D_LSMR_LOCAL;
    }
    return want == DF3D_LSMR_ELEVEN ? 11 : want == DF3D_LSMR_LAUNCHES ? 2 : want == DF3D_LSMR_BARRIERS ? 0 : 3;
}
}  // namespace

extern "C" {

int df3d_ba_trf_subspace(const df3d_ba_problem* p, const double* Jc, const double* Jp, const double* scale_dev, const double* g_dev, const double* f_dev,
                         double Delta, double* g_h, double* gn_h, double* s0, double* s1, double* Js0, double* Js1, double* tmp_m, double* work_dev,
                         double* scratch_dev, double* out_host, void* stream, int form_arg) {
    if (int rc = check_problem(p)) return rc;
    DF3D_CHECK_ARG(Jc && Jp && scale_dev && g_dev && f_dev && g_h && gn_h && s0 && s1 && Js0 && Js1 && tmp_m && work_dev && scratch_dev && out_host, "null pointer");
    DF3D_CHECK_ARG(form_arg == DF3D_LSMR_AUTO || form_arg == DF3D_LSMR_BARRIERS || form_arg == DF3D_LSMR_LAUNCHES || form_arg == DF3D_LSMR_LOCAL ||
                       form_arg == DF3D_LSMR_ELEVEN, "unknown LSMR form");
    DF3D_CHECK_ARG(Delta > 0, "the trust radius must be positive");
    hipStream_t s = df3d::as_stream(stream);
    const size_t m = 2 * (size_t)p->nobs, n = 6 * (size_t)p->ncam + 3 * (size_t)p->npts;
    double* const tr = scratch_dev + TR_OFFSET;
    const int gn = grid_for(n);
    // |g|_inf; g_h = D g; the Cauchy model's two dots; the damping
    hipLaunchKernelGGL(absmax_kernel, dim3(gn), dim3(256), 0, s, g_dev, n, scratch_dev);
    hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(256), 0, s, scratch_dev, (size_t)gn, tr + TR_GNORM);
    hipLaunchKernelGGL(mul_kernel, dim3(gn), dim3(256), 0, s, scale_dev, g_dev, g_h, n);
    if (int rc = launch_matvec(p, Jc, Jp, scale_dev, g_h, tmp_m, s)) return rc;
    {
        const double* a[2] = {tmp_m, g_h};
        const size_t len[2] = {m, n};
        if (int rc = enqueue_dots(2, a, a, len, scratch_dev, tr + TR_JG2, s)) return rc;
    }
    hipLaunchKernelGGL(trf_scalar_kernel<0>, dim3(1), dim3(64), 0, s, tr, Delta);
    DF3D_LAUNCH_CHECK();

    double info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int maxiter = (int)(m < n ? m : n);
    const double atol = 1e-6, btol = 1e-6, conlim = 1e8;   // solve_trf's (scipy's tr_options defaults for lsmr inside least_squares)
    int form = resolve_form(form_arg);
    bool local = form == 3 && df3d_lsmr::local_fits(*p);
    double* const lscratch = work_dev + (2 * m + 4 * n + DF3D_BA_SCRATCH_DOUBLES + 64 + 2 * df3d_lsmr::FUSED_DOUBLES + 3 * df3d_lsmr::FUSED_RED + 8);
    double* const lstate = lscratch + (df3d_lsmr::local_scratch_bytes() + 7) / 8;
    auto lsmr_on_host_damp = [&](int f) -> int {   // the forms whose set-up runs on the host: one more read-back for the damping
        double damp = 0.0;
        if (int rc = read_back(tr + TR_DAMP, &damp, s)) return rc;
        int rc = lsmr_run(p, Jc, Jp, scale_dev, f_dev, damp, atol, btol, conlim, maxiter, gn_h, work_dev, info, stream, f);
        if (rc == DF3D_OK && f == 0 && info[0] < 0) {
            rc = lsmr_run(p, Jc, Jp, scale_dev, f_dev, damp, atol, btol, conlim, maxiter, gn_h, work_dev, info, stream, 2);
            if (rc == DF3D_OK) info[7] = 1;
        }
        return rc;
    };
    if (local) {
        if (df3d_lsmr::launch_local(*p, Jc, Jp, scale_dev, f_dev, gn_h, 0.0, atol, btol, 1.0 / conlim, maxiter, lscratch, lstate, s, tr + TR_DAMP) < 0) local = false;
        DF3D_LAUNCH_CHECK();
    }
    if (!local) {
        if (int rc = lsmr_on_host_damp(form == 3 ? 2 : form)) return rc;
        if (form == 3) info[7] = 2;
    }
    // S = qr([g_h, gn_h]) with LAPACK's signs, J_h S, the 2x2 model
    auto subspace = [&]() -> int {
        hipLaunchKernelGGL(axpby_dev_kernel, dim3(gn), dim3(256), 0, s, tr + TR_CS0, g_h, nullptr, nullptr, s0, n);
        {
            const double* a[1] = {s0};
            const double* b[1] = {gn_h};
            const size_t len[1] = {n};
            if (int rc = enqueue_dots(1, a, b, len, scratch_dev, tr + TR_R01, s)) return rc;
        }
        hipLaunchKernelGGL(trf_scalar_kernel<1>, dim3(1), dim3(64), 0, s, tr, Delta);
        hipLaunchKernelGGL(axpby_dev_kernel, dim3(gn), dim3(256), 0, s, tr + TR_ONE, gn_h, tr + TR_MR01, s0, s1, n);
        {
            const double* a[1] = {s1};
            const size_t len[1] = {n};
            if (int rc = enqueue_dots(1, a, a, len, scratch_dev, tr + TR_N1SQ, s)) return rc;
        }
        hipLaunchKernelGGL(trf_scalar_kernel<2>, dim3(1), dim3(64), 0, s, tr, Delta);
        hipLaunchKernelGGL(axpby_dev_kernel, dim3(gn), dim3(256), 0, s, tr + TR_CS1, s1, nullptr, nullptr, s1, n);
        if (int rc = launch_matvec(p, Jc, Jp, scale_dev, s0, Js0, s)) return rc;
        if (int rc = launch_matvec(p, Jc, Jp, scale_dev, s1, Js1, s)) return rc;
        const double* a[5] = {Js0, Js0, Js1, s0, s1};
        const double* b[5] = {Js0, Js1, Js1, g_h, g_h};
        const size_t len[5] = {m, m, m, n, n};
        return enqueue_dots(5, a, b, len, scratch_dev, tr + TR_B00, s);
    };
    if (int rc = subspace()) return rc;
    double host[TR_SLOTS];
    df3d_lsmr::State now{};
    DF3D_HIP(hipMemcpyAsync(host, tr, sizeof(host), hipMemcpyDeviceToHost, s));
    if (local) DF3D_HIP(hipMemcpyAsync(&now, lstate, sizeof(now), hipMemcpyDeviceToHost, s));
    DF3D_HIP(hipStreamSynchronize(s));
    if (local) {
        if (now.istop < 0) {
            // the layout check on the device refused the problem, or the persistent kernel timed out waiting for its peers: the solve is
            // repeated from its inputs in the two-kernel form (df3d_ba_lsmr_form's rule), and what was built on its result with it
            const bool timeout = now.istop == -1;
            if (int rc = lsmr_on_host_damp(2)) return rc;
            info[7] = timeout ? 1 : 2;
            if (int rc = subspace()) return rc;
            DF3D_HIP(hipMemcpyAsync(host, tr, sizeof(host), hipMemcpyDeviceToHost, s));
            DF3D_HIP(hipStreamSynchronize(s));
        } else {
            info[0] = now.istop;
            info[1] = now.itn;
            info[2] = now.normr;
            info[3] = now.normar;
            info[4] = now.normA;
            info[5] = now.condA;
            info[6] = now.normx;
            info[7] = 0;
        }
    }
    out_host[0] = host[TR_GNORM];
    out_host[1] = host[TR_JG2];
    out_host[2] = host[TR_GH2];
    out_host[3] = host[TR_DAMP];
    out_host[4] = host[TR_R01];
    out_host[5] = host[TR_N1SQ];
    for (int k = 0; k < 5; ++k) out_host[6 + k] = host[TR_B00 + k];
    for (int k = 0; k < 8; ++k) out_host[11 + k] = info[k];
    return DF3D_OK;
}

int df3d_ba_trf_trial(const df3d_ba_problem* p, double p0, double p1, const double* s0, const double* s1, const double* Js0, const double* Js1,
                      const double* scale_dev, const double* x_dev, const double* g_h, double* step_h, double* tmp_m, double* step, double* x_new,
                      double* f_new, double* scratch_dev, double* out_host, void* stream) {
    if (int rc = check_problem(p)) return rc;
    DF3D_CHECK_ARG(s0 && s1 && Js0 && Js1 && scale_dev && x_dev && g_h && step_h && tmp_m && step && x_new && f_new && scratch_dev && out_host, "null pointer");
    hipStream_t s = df3d::as_stream(stream);
    const size_t m = 2 * (size_t)p->nobs, n = 6 * (size_t)p->ncam + 3 * (size_t)p->npts;
    const int gm = grid_for(m), gn = grid_for(n);
    hipLaunchKernelGGL(axpby_kernel<false>, dim3(gn), dim3(256), 0, s, p0, s0, p1, s1, step_h, n, nullptr);
    hipLaunchKernelGGL(axpby_kernel<false>, dim3(gm), dim3(256), 0, s, p0, Js0, p1, Js1, tmp_m, m, nullptr);   // J_h step = p0 J_h s0 + p1 J_h s1
    hipLaunchKernelGGL(mul_kernel, dim3(gn), dim3(256), 0, s, scale_dev, step_h, step, n);
    hipLaunchKernelGGL(axpby_kernel<false>, dim3(gn), dim3(256), 0, s, 1.0, x_dev, 1.0, step, x_new, n, nullptr);
    hipLaunchKernelGGL(ba_eval_kernel, dim3((p->nobs + 255) / 256), dim3(256), 0, s, *p, x_new, f_new, nullptr, nullptr);
    const double* a[6] = {tmp_m, step_h, step_h, f_new, step, x_dev};
    const double* b[6] = {tmp_m, g_h, step_h, f_new, step, x_dev};
    const size_t len[6] = {m, n, n, m, n, n};
    double* const result = scratch_dev + 8 * RED_BLOCKS;
    if (int rc = enqueue_dots(6, a, b, len, scratch_dev, result, s)) return rc;
    DF3D_HIP(hipMemcpyAsync(out_host, result, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    DF3D_HIP(hipStreamSynchronize(s));
    return DF3D_OK;
}

int df3d_ba_trf_linearize(const df3d_ba_problem* p, const double* x_dev, double* f_dev, int eval_f, double* Jc, double* Jp, double* g_dev, double* colsq_dev,
                          double* scale_inv_dev, double* scale_dev, int first, double* scratch_dev, void* stream) {
    if (int rc = check_problem(p)) return rc;
    DF3D_CHECK_ARG(x_dev && f_dev && Jc && Jp && g_dev && colsq_dev && scale_inv_dev && scale_dev && scratch_dev, "null pointer");
    hipStream_t s = df3d::as_stream(stream);
    const size_t n = 6 * (size_t)p->ncam + 3 * (size_t)p->npts;
    hipLaunchKernelGGL(ba_eval_kernel, dim3((p->nobs + 255) / 256), dim3(256), 0, s, *p, x_dev, eval_f ? f_dev : nullptr, Jc, Jp);
    if (int rc = launch_rmatvec(p, Jc, Jp, nullptr, f_dev, g_dev, scratch_dev, s)) return rc;   // g = J^T f
    if (int rc = df3d_ba_colsq(p, Jc, Jp, colsq_dev, scratch_dev, stream)) return rc;
    hipLaunchKernelGGL(update_scale_kernel, dim3(grid_for(n)), dim3(256), 0, s, colsq_dev, scale_inv_dev, scale_dev, n, first);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

}  // extern "C"
