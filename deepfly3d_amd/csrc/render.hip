// f4: frames of the pose videos (reference df3d/video.py:21-108, called from df3d/cli.py:308-321).
//
// The reference stacks six `Core.plot_2d` images (cameras 0, 1, 2 over 4, 5, 6: matplotlib / cv2 drawing on the host, one image at a
// time) into a 2 x 3 grid per frame and, for the 3-D video, adds a row of three matplotlib 3-D plots.  Here a frame is ONE kernel launch:
// every output pixel decides for itself whether it lies on a joint disc, on a bone segment or shows the camera image, from the 38
// joints of its camera in LDS -- no host drawing, no per-image round trip.  The rasterisation rule is this file's own (distance to
// the joint / to the segment against radius / half width, later entries of the tables win, joints over bones) and is restated in
// oracle/render.py, which the tests compare with bit for bit; float64 throughout and compiled without multiply-add fusion (build.py)
// so that the restatement's numpy arithmetic rounds the same way.  Visualisation only: nothing downstream reads these pixels.
#include <cmath>

#include "common.h"

namespace {

constexpr int MAXJ = 64;     // joints per camera the kernels hold in LDS
constexpr int MAXB = 96;     // bones

struct Skeleton {
    int nj, nb;
    short bones[MAXB][2];
    unsigned char rgb[MAXJ][3];   // colour of a joint = colour of its limb; a bone takes the colour of its first joint
};

// squared distance from p to the segment a-b
__device__ __forceinline__ double seg_dist2(double px, double py, double ax, double ay, double bx, double by) {
    const double dx = bx - ax, dy = by - ay;
    const double len2 = dx * dx + dy * dy;
    double t = len2 > 0.0 ? ((px - ax) * dx + (py - ay) * dy) / len2 : 0.0;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    const double qx = ax + t * dx, qy = ay + t * dy;
    return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

// grid [2 H, 3 W, 3]: block (x tile, y, camera slot)
__global__ __launch_bounds__(256) void render_pose2d_kernel(const unsigned char* __restrict__ luma, int H, int W,
                                                            const double* __restrict__ pts, Skeleton sk, double radius, double half_width,
                                                            unsigned char* __restrict__ out) {
    __shared__ double jx[MAXJ], jy[MAXJ];
    __shared__ int seen[MAXJ];
    const int slot = blockIdx.z;
    if (threadIdx.x < (unsigned)sk.nj) {
        const double r = pts[((size_t)slot * sk.nj + threadIdx.x) * 2 + 0], c = pts[((size_t)slot * sk.nj + threadIdx.x) * 2 + 1];
        jy[threadIdx.x] = r;
        jx[threadIdx.x] = c;
        seen[threadIdx.x] = (r != 0.0 && c != 0.0) ? 1 : 0;
    }
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const unsigned char g = luma[((size_t)slot * H + y) * W + x];
    unsigned char r = g, gg = g, b = g;
    const double px = (double)x, py = (double)y;
    const double hw2 = half_width * half_width, rad2 = radius * radius;
    for (int k = 0; k < sk.nb; ++k) {
        const int a = sk.bones[k][0], bb = sk.bones[k][1];
        if (!seen[a] || !seen[bb]) continue;
        if (seg_dist2(px, py, jx[a], jy[a], jx[bb], jy[bb]) <= hw2) {
            r = sk.rgb[a][0];
            gg = sk.rgb[a][1];
            b = sk.rgb[a][2];
        }
    }
    for (int j = 0; j < sk.nj; ++j) {
        if (!seen[j]) continue;
        const double dx = px - jx[j], dy = py - jy[j];
        if (dx * dx + dy * dy <= rad2) {
            r = sk.rgb[j][0];
            gg = sk.rgb[j][1];
            b = sk.rgb[j][2];
        }
    }
    const int row = slot / 3, col = slot % 3;
    unsigned char* o = out + (((size_t)(row * H + y)) * (3 * W) + (size_t)col * W + x) * 3;
    o[0] = r;
    o[1] = gg;
    o[2] = b;
}

// three square panels side by side [S, 3 S, 3]: the 3-D skeleton seen from azimuth az[panel], elevation el (orthographic), +-lim
// mapped onto the panel, black background
struct Views {
    double ca[3], sa[3], ce, se;   // cosines / sines of the three azimuths and of the elevation, taken on the host (libm: what the restatement uses)
};
__global__ __launch_bounds__(256) void render_pose3d_kernel(const double* __restrict__ p3, Skeleton sk, Views vw,
                                                            double lim, int S, double half_width, unsigned char* __restrict__ out) {
    __shared__ double sx[MAXJ], sy[MAXJ];
    const int panel = blockIdx.z;
    if (threadIdx.x < (unsigned)sk.nj) {
        const double X = p3[threadIdx.x * 3 + 0], Y = p3[threadIdx.x * 3 + 1], Z = p3[threadIdx.x * 3 + 2];
        const double ca = vw.ca[panel], sa = vw.sa[panel], ce = vw.ce, se = vw.se;
        const double u = -sa * X + ca * Y;                       // screen right
        const double v = -se * ca * X - se * sa * Y + ce * Z;    // screen up
        sx[threadIdx.x] = (u / lim * 0.5 + 0.5) * (double)(S - 1);
        sy[threadIdx.x] = (0.5 - v / lim * 0.5) * (double)(S - 1);
    }
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= S) return;
    unsigned char r = 0, g = 0, b = 0;
    const double hw2 = half_width * half_width;
    for (int k = 0; k < sk.nb; ++k) {
        const int a = sk.bones[k][0], bb = sk.bones[k][1];
        if (seg_dist2((double)x, (double)y, sx[a], sy[a], sx[bb], sy[bb]) <= hw2) {
            r = sk.rgb[a][0];
            g = sk.rgb[a][1];
            b = sk.rgb[a][2];
        }
    }
    unsigned char* o = out + ((size_t)y * (3 * S) + (size_t)panel * S + x) * 3;
    o[0] = r;
    o[1] = g;
    o[2] = b;
}

// bilinear resize of an RGB image, pixel centres at half integers (cv2.INTER_LINEAR's / torch align_corners=False geometry), clamped
__global__ __launch_bounds__(256) void resize_rgb_kernel(const unsigned char* __restrict__ in, int ih, int iw, int in_pitch_px,
                                                         unsigned char* __restrict__ out, int oh, int ow, int out_pitch_px) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    const double fx = ((double)x + 0.5) * (double)iw / (double)ow - 0.5, fy = ((double)y + 0.5) * (double)ih / (double)oh - 0.5;
    const double cx = fx < 0.0 ? 0.0 : (fx > (double)(iw - 1) ? (double)(iw - 1) : fx);
    const double cy = fy < 0.0 ? 0.0 : (fy > (double)(ih - 1) ? (double)(ih - 1) : fy);
    const int x0 = (int)floor(cx), y0 = (int)floor(cy);
    const int x1 = x0 + 1 < iw ? x0 + 1 : iw - 1, y1 = y0 + 1 < ih ? y0 + 1 : ih - 1;
    const double wx = cx - (double)x0, wy = cy - (double)y0;
    for (int c = 0; c < 3; ++c) {
        const double v00 = in[((size_t)y0 * in_pitch_px + x0) * 3 + c], v01 = in[((size_t)y0 * in_pitch_px + x1) * 3 + c];
        const double v10 = in[((size_t)y1 * in_pitch_px + x0) * 3 + c], v11 = in[((size_t)y1 * in_pitch_px + x1) * 3 + c];
        const double top = v00 + wx * (v01 - v00), bot = v10 + wx * (v11 - v10);
        const double v = top + wy * (bot - top);
        out[((size_t)y * out_pitch_px + x) * 3 + c] = (unsigned char)floor(v + 0.5);
    }
}

int fill_skeleton(Skeleton& sk, int nj, const int* bones, int nb, const unsigned char* joint_rgb) {
    DF3D_CHECK_ARG(nj >= 1 && nj <= MAXJ && nb >= 0 && nb <= MAXB && (nb == 0 || bones) && joint_rgb, "1..64 joints, at most 96 bones, host tables");
    sk.nj = nj;
    sk.nb = nb;
    for (int k = 0; k < nb; ++k) {
        DF3D_CHECK_ARG(bones[2 * k] >= 0 && bones[2 * k] < nj && bones[2 * k + 1] >= 0 && bones[2 * k + 1] < nj, "bone joint out of range");
        sk.bones[k][0] = (short)bones[2 * k];
        sk.bones[k][1] = (short)bones[2 * k + 1];
    }
    for (int j = 0; j < nj; ++j)
        for (int c = 0; c < 3; ++c) sk.rgb[j][c] = joint_rgb[3 * j + c];
    return DF3D_OK;
}

}  // namespace

extern "C" {

int df3d_render_pose2d_grid(const unsigned char* luma_dev, int height, int width, const double* points_px_dev, int num_joints, const int* bones_host,
                            int num_bones, const unsigned char* joint_rgb_host, double radius, double line_width, unsigned char* out_rgb_dev, void* stream) {
    DF3D_CHECK_ARG(luma_dev && points_px_dev && out_rgb_dev && height > 0 && width > 0 && radius >= 0 && line_width >= 0, "null pointer or empty image");
    Skeleton sk;
    if (int rc = fill_skeleton(sk, num_joints, bones_host, num_bones, joint_rgb_host)) return rc;
    hipLaunchKernelGGL(render_pose2d_kernel, dim3((width + 255) / 256, height, 6), dim3(256), 0, df3d::as_stream(stream), luma_dev, height, width,
                       points_px_dev, sk, radius, 0.5 * line_width, out_rgb_dev);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int df3d_render_pose3d_panels(const double* points3d_dev, int num_joints, const int* bones_host, int num_bones, const unsigned char* joint_rgb_host,
                              const double* azimuth_deg3, double elevation_deg, double lim, int size, double line_width, unsigned char* out_rgb_dev,
                              void* stream) {
    DF3D_CHECK_ARG(points3d_dev && azimuth_deg3 && out_rgb_dev && lim > 0 && size > 1 && line_width >= 0, "null pointer or bad panel geometry");
    Skeleton sk;
    if (int rc = fill_skeleton(sk, num_joints, bones_host, num_bones, joint_rgb_host)) return rc;
    const double k = 3.14159265358979323846 / 180.0;
    Views vw;
    for (int i = 0; i < 3; ++i) {
        vw.ca[i] = std::cos(azimuth_deg3[i] * k);
        vw.sa[i] = std::sin(azimuth_deg3[i] * k);
    }
    vw.ce = std::cos(elevation_deg * k);
    vw.se = std::sin(elevation_deg * k);
    hipLaunchKernelGGL(render_pose3d_kernel, dim3((size + 255) / 256, size, 3), dim3(256), 0, df3d::as_stream(stream), points3d_dev, sk, vw, lim, size,
                       0.5 * line_width, out_rgb_dev);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

int df3d_resize_rgb(const unsigned char* in_dev, int in_h, int in_w, int in_pitch_px, unsigned char* out_dev, int out_h, int out_w, int out_pitch_px,
                    void* stream) {
    DF3D_CHECK_ARG(in_dev && out_dev && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && in_pitch_px >= in_w && out_pitch_px >= out_w, "null pointer or bad size");
    hipLaunchKernelGGL(resize_rgb_kernel, dim3((out_w + 255) / 256, out_h), dim3(256), 0, df3d::as_stream(stream), in_dev, in_h, in_w, in_pitch_px, out_dev,
                       out_h, out_w, out_pitch_px);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

}  // extern "C"
