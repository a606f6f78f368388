// a9 (per-side Procrustes registration) and the `Core.get_points3d` chain (median-centre / axis swap / One-Euro
// temporal filter), float64, sequence-global: runs once on the gathered T x 38 x 3 pose.
//
//   median_kernel      exact order statistics by MSB-first radix select on the order-preserving uint64 image of
//                      the doubles: 8 passes x 256-bin LDS histogram, one workgroup per column, both middle ranks
//                      of an even-length column in the same passes.  (a+b)/2 for even n = numpy.median.
//   procrustes chain   segment lengths -> medians -> scale / centre -> fit-joint medians -> 3x3 rigid fit
//                      (one-sided Jacobi SVD) -> apply.  7 launches for both sides together.
//   oneeuro_kernel     one thread per (joint, axis) channel, the reference's scalar recurrence in its exact
//                      operation order (no FMA contraction: bit-identical to the CPython float arithmetic).
//
// Reference: df3d/procrustes.py:51-151,154-263; df3d/plot_util.py:10-18,85-91; df3d/signal_util.py:5-100.
// Everything here is latency-bound (912 B per frame); the point of having it on the device is that the whole
// a1..a10 chain stays in HBM between the hourglass and the result pickle.
#include "common.h"

namespace {

constexpr int SIDE_JOINTS = 19;
constexpr int NSEG = 12;  // 3 legs x 4 segments
constexpr int NFIT = 6;   // body-coxa + coxa-femur of the 3 legs: side joints 0,1,5,6,10,11

__device__ __forceinline__ unsigned long long order_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// Column (blockIdx.x, blockIdx.y) starts at data + x*col_stride_x + y*col_stride_y; its element i lives at
// (i / inner_n) * outer_stride + (i % inner_n) * inner_stride.  out[y*gridDim.x + x] = median.
__global__ __launch_bounds__(256) void median_kernel(const double* __restrict__ data, long long n, long long inner_n,
                                                     long long inner_stride, long long outer_stride,
                                                     long long col_stride_x, long long col_stride_y,
                                                     double* __restrict__ out) {
    __shared__ unsigned int hist[2][256];
    __shared__ unsigned long long s_prefix[2];
    __shared__ long long s_rank[2];
    const double* col = data + blockIdx.x * col_stride_x + blockIdx.y * col_stride_y;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_prefix[0] = s_prefix[1] = 0;
        s_rank[0] = (n - 1) / 2;
        s_rank[1] = n / 2;
    }
    const bool dense = inner_n >= n;
    unsigned long long mask = 0;
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = 8 * pass;
        hist[0][tid] = 0;
        hist[1][tid] = 0;
        __syncthreads();
        const unsigned long long p0 = s_prefix[0], p1 = s_prefix[1];
        const bool same = p0 == p1;
        for (long long i = tid; i < n; i += 256) {
            long long o, r;
            if (dense) {  // one run of n elements: no index arithmetic
                o = 0;
                r = i;
            } else {      // 32-bit division (n < 2^31 is checked by the callers): a 64-bit one costs more than the load
                const unsigned q = (unsigned)i / (unsigned)inner_n;
                o = q;
                r = (long long)((unsigned)i - q * (unsigned)inner_n);
            }
            const unsigned long long k = order_key(col[o * outer_stride + r * inner_stride]);
            const unsigned int bin = (unsigned int)(k >> shift) & 255u;
            if ((k & mask) == p0) atomicAdd(&hist[0][bin], 1u);
            if (!same && (k & mask) == p1) atomicAdd(&hist[1][bin], 1u);
        }
        __syncthreads();
        if (tid < 2) {
            const int w = (same && tid == 1) ? 0 : tid;
            long long rank = s_rank[tid];
            int b = 0;
            for (; b < 255; ++b) {
                const long long c = hist[w][b];
                if (rank < c) break;
                rank -= c;
            }
            s_rank[tid] = rank;
            s_prefix[tid] |= (unsigned long long)b << shift;
        }
        mask |= 255ull << shift;
        __syncthreads();
    }
    if (tid == 0) {
#pragma clang fp contract(off)
        const double a = key_value(s_prefix[0]), b = key_value(s_prefix[1]);
        out[blockIdx.y * gridDim.x + blockIdx.x] = (n & 1) ? a : (a + b) / 2.0;
    }
}

// ---- long columns: the same radix select spread over many workgroups -------------------------------------------------------
// One workgroup per column walks all n values eight times: fine for the 60 columns of 1e5 segment lengths, 44 ms for the three
// columns of 3.8e6 coordinates of a 100 000-frame recording (two workgroups' worth of loads on a 256-CU device).  With a little
// scratch memory a pass becomes a histogram kernel over all the values (LDS histograms, one global atomic per bin and
// workgroup) and a one-wave kernel that picks the bin -- the same counts, the same medians.
constexpr int MED_SCRATCH_DOUBLES = 272;   // per column: hist [2][256] u32 (256 doubles) + prefix [2] + rank [2] (+ pad)
constexpr long long MED_LONG = 1 << 16;    // columns at least this long take the multi-workgroup route when scratch is there
struct MedCol {
    unsigned int hist[2][256];
    unsigned long long prefix[2];
    long long rank[2];
};
static_assert(sizeof(MedCol) <= MED_SCRATCH_DOUBLES * 8, "scratch per column");

__global__ __launch_bounds__(256) void median_hist_kernel(const double* __restrict__ data, long long n, long long inner_n, long long inner_stride,
                                                          long long outer_stride, long long col_stride_x, long long col_stride_y, int shift,
                                                          unsigned long long mask, MedCol* __restrict__ cols) {
    __shared__ unsigned int hist[2][256];
    const int c = blockIdx.z * gridDim.y + blockIdx.y;
    const double* col = data + blockIdx.y * col_stride_x + blockIdx.z * col_stride_y;
    MedCol* const mc = cols + c;
    const int tid = threadIdx.x;
    hist[0][tid] = 0;
    hist[1][tid] = 0;
    __syncthreads();
    const unsigned long long p0 = mc->prefix[0], p1 = mc->prefix[1];
    const bool same = p0 == p1;
    const bool dense = inner_n >= n;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < n; i += (long long)gridDim.x * 256) {
        long long o = 0, r = i;
        if (!dense) {
            const unsigned q = (unsigned)i / (unsigned)inner_n;
            o = q;
            r = (long long)((unsigned)i - q * (unsigned)inner_n);
        }
        const unsigned long long k = order_key(col[o * outer_stride + r * inner_stride]);
        const unsigned int bin = (unsigned int)(k >> shift) & 255u;
        if ((k & mask) == p0) atomicAdd(&hist[0][bin], 1u);
        if (!same && (k & mask) == p1) atomicAdd(&hist[1][bin], 1u);
    }
    __syncthreads();
    if (hist[0][tid]) atomicAdd(&mc->hist[0][tid], hist[0][tid]);
    if (hist[1][tid]) atomicAdd(&mc->hist[1][tid], hist[1][tid]);
}

// shift = 64: initialise the column's state; otherwise pick the bin of this pass, clear the histograms; shift = 0: write the median
__global__ __launch_bounds__(256) void median_select_kernel(MedCol* __restrict__ cols, long long n, int shift, double* __restrict__ out) {
    MedCol* const mc = cols + blockIdx.x;
    const int tid = threadIdx.x;
    if (shift == 64) {
        mc->hist[0][tid] = 0;
        mc->hist[1][tid] = 0;
        if (tid == 0) {
            mc->prefix[0] = mc->prefix[1] = 0;
            mc->rank[0] = (n - 1) / 2;
            mc->rank[1] = n / 2;
        }
        return;
    }
    __shared__ unsigned int hist[2][256];
    hist[0][tid] = mc->hist[0][tid];
    hist[1][tid] = mc->hist[1][tid];
    const bool same = mc->prefix[0] == mc->prefix[1];
    __syncthreads();
    mc->hist[0][tid] = 0;
    mc->hist[1][tid] = 0;
    if (tid < 2) {
        const int w = (same && tid == 1) ? 0 : tid;
        long long rank = mc->rank[tid];
        int b = 0;
        for (; b < 255; ++b) {
            const long long c = hist[w][b];
            if (rank < c) break;
            rank -= c;
        }
        mc->rank[tid] = rank;
        mc->prefix[tid] |= (unsigned long long)b << shift;
    }
    if (shift == 0) {
        __syncthreads();
        if (tid == 0) {
#pragma clang fp contract(off)
            const double a = key_value(mc->prefix[0]), b = key_value(mc->prefix[1]);
            out[blockIdx.x] = (n & 1) ? a : (a + b) / 2.0;
        }
    }
}

// medians of gx x gy columns (column (x, y) -> out[y * gx + x]) through the multi-workgroup route; scratch: gx * gy MedCol
void launch_long_median(const double* data, long long n, long long inner_n, long long inner_stride, long long outer_stride, long long col_stride_x,
                        long long col_stride_y, int gx, int gy, double* out, double* scratch, hipStream_t s) {
    MedCol* const cols = reinterpret_cast<MedCol*>(scratch);
    const int ncols = gx * gy;
    const int nb = (int)((n + 256 * 16 - 1) / (256 * 16) < 1024 ? (n + 256 * 16 - 1) / (256 * 16) : 1024);
    hipLaunchKernelGGL(median_select_kernel, dim3(ncols), dim3(256), 0, s, cols, n, 64, out);
    unsigned long long mask = 0;
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = 8 * pass;
        hipLaunchKernelGGL(median_hist_kernel, dim3(nb, gx, gy), dim3(256), 0, s, data, n, inner_n, inner_stride, outer_stride, col_stride_x, col_stride_y, shift,
                           mask, cols);
        hipLaunchKernelGGL(median_select_kernel, dim3(ncols), dim3(256), 0, s, cols, n, shift, out);
        mask |= 255ull << shift;
    }
}

// work-buffer layout (doubles), T = frames:
//   seg   [2][12][T]      segment lengths per side
//   fit   [2][18][T]      scaled, centred fit-joint coordinates per side
//   med_seg [2][12], med_all [2][3], scale [2], med_fit [2][18], xf [2][12] (rot 3x3 row-major, off 3)
struct Work {
    double *seg, *fit, *med_seg, *med_all, *scale, *med_fit, *xf;
};
__host__ __device__ inline long long work_doubles(long long T) { return 2 * (NSEG + 3 * NFIT) * T + 128 + 6 * MED_SCRATCH_DOUBLES; }
inline Work carve(double* w, long long T) {
    Work k;
    k.seg = w;
    k.fit = k.seg + 2 * NSEG * T;
    k.med_seg = k.fit + 2 * 3 * NFIT * T;
    k.med_all = k.med_seg + 2 * NSEG;
    k.scale = k.med_all + 6;
    k.med_fit = k.scale + 2;
    k.xf = k.med_fit + 2 * 3 * NFIT;
    return k;
}

struct Template {
    double seg_med[2][NSEG];     // median over the template's frames of each segment length
    double fit_med[2][NFIT][3];  // median over the template's frames of each fit joint
};

__global__ __launch_bounds__(256) void seglen_kernel(const double* __restrict__ pts, long long T, double* __restrict__ seg) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    for (int side = 0; side < 2; ++side) {
        const double* p = pts + (t * 38 + side * SIDE_JOINTS) * 3;
        for (int leg = 0; leg < 3; ++leg) {
            for (int s = 0; s < 4; ++s) {
                const double* a = p + (leg * 5 + s) * 3;
                const double dx = a[3] - a[0], dy = a[4] - a[1], dz = a[5] - a[2];
                seg[((long long)side * NSEG + leg * 4 + s) * T + t] = sqrt(dx * dx + dy * dy + dz * dz);
            }
        }
    }
}

// scale[side] = median over the 12 segments of (template median / sequence median)
__global__ void scale_kernel(Template tm, const double* __restrict__ med_seg, double* __restrict__ scale) {
#pragma clang fp contract(off)
    const int side = threadIdx.x;
    if (side >= 2) return;
    double r[NSEG];
    for (int i = 0; i < NSEG; ++i) r[i] = tm.seg_med[side][i] / med_seg[side * NSEG + i];
    for (int i = 1; i < NSEG; ++i) {  // insertion sort of 12 numbers
        const double v = r[i];
        int j = i - 1;
        while (j >= 0 && r[j] > v) {
            r[j + 1] = r[j];
            --j;
        }
        r[j + 1] = v;
    }
    scale[side] = (r[NSEG / 2 - 1] + r[NSEG / 2]) / 2.0;
}

__device__ __forceinline__ int fit_joint(int f) { return (f >> 1) * 5 + (f & 1); }

__global__ __launch_bounds__(256) void fit_cols_kernel(const double* __restrict__ pts, long long T, const double* __restrict__ med_all,
                                                       const double* __restrict__ scale, double* __restrict__ fit) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    for (int side = 0; side < 2; ++side) {
        const double s = scale[side];
        for (int f = 0; f < NFIT; ++f) {
            const double* p = pts + (t * 38 + side * SIDE_JOINTS + fit_joint(f)) * 3;
            for (int a = 0; a < 3; ++a) fit[((long long)side * 18 + f * 3 + a) * T + t] = (p[a] - med_all[side * 3 + a]) * s;
        }
    }
}

// One thread per side: rigid fit (rotation or reflection, no scaling) of the 6 median fit joints to the template's.
__global__ void rigid_fit_kernel(Template tm, const double* __restrict__ med_fit, double* __restrict__ xf) {
    const int side = threadIdx.x;
    if (side >= 2) return;
    double tg[NFIT][3], sr[NFIT][3], mu_t[3] = {0, 0, 0}, mu_s[3] = {0, 0, 0};
    for (int f = 0; f < NFIT; ++f)
        for (int a = 0; a < 3; ++a) {
            tg[f][a] = tm.fit_med[side][f][a];
            sr[f][a] = med_fit[side * 18 + f * 3 + a];
            mu_t[a] += tg[f][a];
            mu_s[a] += sr[f][a];
        }
    double nt = 0, ns = 0;
    for (int a = 0; a < 3; ++a) {
        mu_t[a] /= NFIT;
        mu_s[a] /= NFIT;
    }
    for (int f = 0; f < NFIT; ++f)
        for (int a = 0; a < 3; ++a) {
            tg[f][a] -= mu_t[a];
            sr[f][a] -= mu_s[a];
            nt += tg[f][a] * tg[f][a];
            ns += sr[f][a] * sr[f][a];
        }
    nt = sqrt(nt);
    ns = sqrt(ns);
    // G = target^T source (3x3), normalised; one-sided Jacobi: G V = U S
    double G[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int f = 0; f < NFIT; ++f) acc += (tg[f][i] / nt) * (sr[f][j] / ns);
            G[i][j] = acc;
            V[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int k = 0; k < 3; ++k) {
                    al += G[k][p] * G[k][p];
                    be += G[k][q] * G[k][q];
                    ga += G[k][p] * G[k][q];
                }
                if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                off = fmax(off, fabs(ga) / sqrt(al * be));
                const double zeta = (be - al) / (2.0 * ga);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
                for (int k = 0; k < 3; ++k) {
                    const double gp = G[k][p], gq = G[k][q];
                    G[k][p] = c * gp - s * gq;
                    G[k][q] = s * gp + c * gq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - s * vq;
                    V[k][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-16) break;
    }
    double U[3][3], sv[3], smax = 0;
    for (int j = 0; j < 3; ++j) {
        sv[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
        smax = fmax(smax, sv[j]);
    }
    int weak = -1;
    for (int j = 0; j < 3; ++j) {
        if (sv[j] > 1e-13 * smax) {
            for (int k = 0; k < 3; ++k) U[k][j] = G[k][j] / sv[j];
        } else {
            weak = j;
        }
    }
    if (weak >= 0) {  // coplanar fit joints: complete the frame (the reference's SVD is arbitrary here as well)
        const int a = (weak + 1) % 3, b = (weak + 2) % 3;
        U[0][weak] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
        U[1][weak] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
        U[2][weak] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
    }
    // rot = V U^T ; off = mu_t - mu_s @ rot
    double* o = xf + side * 12;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + V[i][2] * U[j][2];
    for (int j = 0; j < 3; ++j) o[9 + j] = mu_t[j] - (mu_s[0] * o[0 * 3 + j] + mu_s[1] * o[1 * 3 + j] + mu_s[2] * o[2 * 3 + j]);
}

__global__ __launch_bounds__(256) void procrustes_apply_kernel(const double* __restrict__ pts, long long TJ, const double* __restrict__ med_all,
                                                               const double* __restrict__ scale, const double* __restrict__ xf,
                                                               double* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (t, joint)
    if (idx >= TJ) return;
    const int side = (int)(idx % 38) >= SIDE_JOINTS ? 1 : 0;
    const double s = scale[side];
    const double* x = xf + side * 12;
    double v[3];
    for (int a = 0; a < 3; ++a) v[a] = (pts[idx * 3 + a] - med_all[side * 3 + a]) * s;
    for (int b = 0; b < 3; ++b) out[idx * 3 + b] = v[0] * x[b] + v[1] * x[3 + b] + v[2] * x[6 + b] + x[9 + b];
}

__global__ __launch_bounds__(256) void normalize_kernel(const double* __restrict__ in, long long TJ, const double* __restrict__ med,
                                                        int rotate, double* __restrict__ out) {
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= TJ) return;
    const double x = in[idx * 3] - med[0], y = in[idx * 3 + 1] - med[1], z = in[idx * 3 + 2] - med[2];
    out[idx * 3] = x;
    out[idx * 3 + 1] = rotate ? -z : y;
    out[idx * 3 + 2] = rotate ? -y : z;
}

struct EuroCfg {
    double freq, mincutoff, beta, dcutoff, dt;
    long long i0;
};

__device__ __forceinline__ double euro_alpha(double freq, double cutoff) {
#pragma clang fp contract(off)
    const double te = 1.0 / freq;
    const double tau = 1.0 / (2 * 3.141592653589793 * cutoff);
    return 1.0 / (1.0 + tau / te);
}

__global__ __launch_bounds__(64) void oneeuro_kernel(const double* __restrict__ in, long long T, int nch, EuroCfg cfg, double* __restrict__ out) {
#pragma clang fp contract(off)
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nch) return;
    double f = cfg.freq, last_t = 0.0, prev = 0.0, sx = 0.0, sdx = 0.0;
    double nxt = T > 0 ? in[ch] : 0.0;
    for (long long i = 0; i < T; ++i) {
        const double v = nxt;
        if (i + 1 < T) nxt = in[(i + 1) * nch + ch];  // next sample in flight during this step's divisions
        const double t = (double)(i + cfg.i0) * cfg.dt;
        if (last_t != 0.0 && t != 0.0) f = 1.0 / (t - last_t);
        last_t = t;
        const double dx = i == 0 ? 0.0 : (v - prev) * f;
        const double ad = euro_alpha(f, cfg.dcutoff);
        sdx = i == 0 ? dx : ad * dx + (1.0 - ad) * sdx;
        const double cutoff = cfg.mincutoff + cfg.beta * fabs(sdx);
        const double a = euro_alpha(f, cutoff);
        sx = i == 0 ? v : a * v + (1.0 - a) * sx;
        prev = v;
        out[i * nch + ch] = sx;
    }
}

inline unsigned grid_for(long long n, int block) { return (unsigned)((n + block - 1) / block); }

}  // namespace

extern "C" {

int df3d_column_median(const double* cols, int ncols, long long n, long long col_stride, double* out, void* stream) {
    DF3D_CHECK_ARG(cols && out, "null pointer");
    DF3D_CHECK_ARG(ncols >= 0 && n >= 1 && n < (1LL << 31), "need ncols >= 0 and 1 <= n < 2^31");
    if (ncols == 0) return DF3D_OK;
    hipLaunchKernelGGL(median_kernel, dim3(ncols, 1), dim3(256), 0, df3d::as_stream(stream), cols, n, n, 1LL, 0LL, col_stride, 0LL, out);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

long long df3d_procrustes_work_doubles(long long T) { return T < 0 ? 0 : work_doubles(T); }

int df3d_procrustes(const double* pts, long long T, const double* tmpl_seg_med, const double* tmpl_fit_med, double* out,
                    double* work, long long work_len, void* stream) {
    DF3D_CHECK_ARG(pts && out && work && tmpl_seg_med && tmpl_fit_med, "null pointer");
    DF3D_CHECK_ARG(T >= 1 && T * 38 < (1LL << 31), "need 1 <= T < 2^31 / 38 frames");
    DF3D_CHECK_ARG(work_len >= work_doubles(T), "work buffer too small (df3d_procrustes_work_doubles)");
    hipStream_t s = df3d::as_stream(stream);
    Template tm;
    memcpy(tm.seg_med, tmpl_seg_med, sizeof(tm.seg_med));
    memcpy(tm.fit_med, tmpl_fit_med, sizeof(tm.fit_med));
    const Work w = carve(work, T);
    hipLaunchKernelGGL(seglen_kernel, dim3(grid_for(T, 256)), dim3(256), 0, s, pts, T, w.seg);
    // medians: 24 segment-length columns over T; per side and axis all T*19 points (strided in place)
    hipLaunchKernelGGL(median_kernel, dim3(2 * NSEG, 1), dim3(256), 0, s, w.seg, T, T, 1LL, 0LL, T, 0LL, w.med_seg);
    if (T * SIDE_JOINTS >= MED_LONG)
        launch_long_median(pts, T * SIDE_JOINTS, (long long)SIDE_JOINTS, 3LL, 114LL, 1LL, (long long)SIDE_JOINTS * 3, 3, 2, w.med_all,
                           work + work_doubles(T) - 6 * MED_SCRATCH_DOUBLES, s);
    else
        hipLaunchKernelGGL(median_kernel, dim3(3, 2), dim3(256), 0, s, pts, T * SIDE_JOINTS, (long long)SIDE_JOINTS, 3LL, 114LL, 1LL,
                           (long long)SIDE_JOINTS * 3, w.med_all);
    hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(64), 0, s, tm, w.med_seg, w.scale);
    hipLaunchKernelGGL(fit_cols_kernel, dim3(grid_for(T, 256)), dim3(256), 0, s, pts, T, w.med_all, w.scale, w.fit);
    hipLaunchKernelGGL(median_kernel, dim3(2 * 3 * NFIT, 1), dim3(256), 0, s, w.fit, T, T, 1LL, 0LL, T, 0LL, w.med_fit);
    hipLaunchKernelGGL(rigid_fit_kernel, dim3(1), dim3(64), 0, s, tm, w.med_fit, w.xf);
    hipLaunchKernelGGL(procrustes_apply_kernel, dim3(grid_for(T * 38, 256)), dim3(256), 0, s, pts, T * 38, w.med_all, w.scale, w.xf, out);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int df3d_pose_normalize(const double* in, long long T, int njoints, int rotate, double* out, double* work, long long work_len,
                        void* stream) {
    DF3D_CHECK_ARG(in && out && work, "null pointer");
    DF3D_CHECK_ARG(T >= 1 && njoints >= 1 && T * njoints < (1LL << 31), "need T >= 1, njoints >= 1 and T * njoints < 2^31");
    DF3D_CHECK_ARG(work_len >= 3, "work buffer needs 3 doubles");
    hipStream_t s = df3d::as_stream(stream);
    const long long TJ = T * njoints;
    if (TJ >= MED_LONG && work_len >= 8 + 3 * MED_SCRATCH_DOUBLES)   // (a caller of round 1 passes 3 doubles: one workgroup per column)
        launch_long_median(in, TJ, TJ, 3LL, 0LL, 1LL, 0LL, 3, 1, work, work + 8, s);
    else
        hipLaunchKernelGGL(median_kernel, dim3(3, 1), dim3(256), 0, s, in, TJ, TJ, 3LL, 0LL, 1LL, 0LL, work);
    hipLaunchKernelGGL(normalize_kernel, dim3(grid_for(TJ, 256)), dim3(256), 0, s, in, TJ, work, rotate, out);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int df3d_oneeuro_filter(const double* in, long long T, int nch, double freq, double mincutoff, double beta, double dcutoff,
                        long long first_stamp, double stamp_step, double* out, void* stream) {
    DF3D_CHECK_ARG(T >= 0 && nch >= 1, "need T >= 0 and nch >= 1");
    DF3D_CHECK_ARG(freq > 0 && mincutoff > 0 && dcutoff > 0, "freq, mincutoff and dcutoff must be > 0");
    if (T == 0) return DF3D_OK;
    DF3D_CHECK_ARG(in && out, "null pointer");
    EuroCfg cfg{freq, mincutoff, beta, dcutoff, stamp_step, first_stamp};
    hipLaunchKernelGGL(oneeuro_kernel, dim3(grid_for(nch, 64)), dim3(64), 0, df3d::as_stream(stream), in, T, nch, cfg, out);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

}  // extern "C"
