/*
 * df3d_hip.h -- C ABI of libdf3d_hip.so, the MI355X (gfx950) back-end of the DeepFly3D per-frame hot
 * path.  Plain C: pointers, sizes, no torch / C++ types.  This is the drop-in boundary: every entry
 * point replaces one piece of work the reference reaches through its two Python imports
 *     from df2d.inference import inference_folder        (reference df3d/core.py:11)
 *     from pyba.CameraNetwork import CameraNetwork       (reference df3d/core.py:12)
 * The reference has no FFI of its own; INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every pointer named *_dev is DEVICE memory owned by the caller (a torch tensor's data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous
 *     on that stream unless the comment says "synchronous".
 *   - return value: 0 = DF3D_OK, negative = error; df3d_last_error() returns a thread-local message.
 *   - no exceptions cross the ABI; handles are not thread-safe, distinct handles are independent.
 */
#ifndef DF3D_HIP_H
#define DF3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DF3D_OK 0
#define DF3D_EINVAL (-1)  /* bad argument                       */
#define DF3D_EHIP (-2)    /* a HIP runtime call failed          */
#define DF3D_ESTATE (-3)  /* handle not ready (weights missing) */
#define DF3D_ENOGPU (-4)  /* no gfx950 device visible           */
#define DF3D_ENOSPC (-5)  /* the caller's buffer is too small (the size needed was reported) */
#define DF3D_EIO (-6)     /* a file could not be opened or read */

#define DF3D_DTYPE_F32 0
#define DF3D_DTYPE_BF16 1
#define DF3D_DTYPE_F16 2  /* IEEE half activations + weights, fp32 accumulate: the bf16 engine's kernels on the other 16-bit format */
#define DF3D_DTYPE_F32S 3 /* float32 tensors, weights and accumulation -- the F32 engine's plan, workspace and kernels -- with every product formed on the
                             half-precision matrix pipe from a two-way IEEE-half split of both operands (x = hi + lo to 2^-22 |x|): three MFMAs per
                             K step (hi hi + lo hi + hi lo, the 2^-22 lo lo term dropped) where exact fp32 takes eight four times as long.  Operands
                             must lie inside the half range (|x| < 65 504), which batch-normalised activations and their weights do.  Not
                             bit-identical to F32: heat-maps differ by ~1.5e-6 of their range (the same 5e-5 test tolerance, the same arg-max cells).
                             Needs the df3d_hg_lowp_bytes() buffer (the pre-split copy of the weights and the streams packed from it) */

const char* df3d_last_error(void);
/* ABI revision of the library: DF3D_ABI_VERSION of the header it was built from.  It changes whenever an entry point's signature or
 * a struct layout does (round 3 inserted `resize` into df3d_preprocess_u8 / df3d_hg_forward_u8 and `bytes_m1` into
 * df3d_hg_profile_read: 300; round 4: 400; round 5 added df3d_ba_lsmr_form and grew df3d_ba_lsmr_work_doubles: 500; round 6 added df3d_hg_profile_executed_flops and df3d_heatmap_argmax_checked: 600, and the df3d_ba_trf_* entries: 610); a caller compares it with the header it compiled against before its first call
 * (deepfly3d_amd/_native.py:load does). */
#define DF3D_ABI_VERSION 610
int df3d_version(void);
/* number of visible HIP devices (<0 on error); name of device `dev` copied to buf */
int df3d_device_count(void);
int df3d_device_name(int dev, char* buf, int buflen);

/* ------------------------------------------------------------------------------------------------
 * a1  input front-end of df2d's inference_folder (call site reference df3d/core.py:177-185): uint8 frames
 *     [n, H, W, C] (C = 1 or 3) -> float32 NHWC [n, OH, OW, 3]: optional left-right flip per view
 *     (flip_dev[n] uint8, may be NULL; the reference flips cameras ordering[4:]), resize, (v/255 - mean[c]) / std[c].
 *     mean3/std3 are HOST float[3].  df2d's resize rule is not in the reference checkout, so it is an argument:
 *     DF3D_RESIZE_BILINEAR (half-pixel centres, no antialias: cv2.INTER_LINEAR), DF3D_RESIZE_BILINEAR_ALIGN_CORNERS,
 *     DF3D_RESIZE_AREA (overlap-weighted mean of the covered source rectangle: cv2.INTER_AREA).
 *     (Round 3 added the `resize` argument to this function and to df3d_hg_forward_u8.)
 * ---------------------------------------------------------------------------------------------- */
#define DF3D_RESIZE_BILINEAR 0
#define DF3D_RESIZE_BILINEAR_ALIGN_CORNERS 1
#define DF3D_RESIZE_AREA 2
int df3d_preprocess_u8(const unsigned char* img_dev, const unsigned char* flip_dev, int n, int H, int W, int C,
                       float* out_dev, int OH, int OW, const float* mean3_host, const float* std3_host, int resize, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a1  JPEG front-end: n baseline JPEG files -> their luma planes, on the device.
 *     Replaces the libjpeg decode inside df2d's DataLoader (call site reference df3d/core.py:177-185; the
 *     frames are ffmpeg-written JPEGs, df3d/core.py:446-459).  Baseline / extended sequential Huffman, 8 bit,
 *     1-4 components in one interleaved scan, any sampling factors, restart intervals; libjpeg's default
 *     "islow" inverse DCT, bit-exact.  Only the first (luma) component is reconstructed.
 * files_dev   the files' bytes in one 16-byte aligned buffer of total_file_bytes (+ 16 readable bytes behind it);
 *             file i = [offsets[i], offsets[i] + sizes[i]), every offset a multiple of 16, files in
 *             ascending, non-overlapping order
 * offsets_dev, sizes_dev  [n] uint32 (device);  max_file_bytes = the largest size (host value, 0 = unknown; a hint kept
 *             for callers of round 1, whose parallel decoder staged small streams in LDS: the decoder now keeps each lane's
 *             next stream dwords in registers and ignores it)
 * luma_dev    [n, height, width] uint8;   every file must be width x height
 * status_dev  [n] int32: 0 ok, 1 truncated, 2 not a JPEG, 3 unsupported (progressive / arithmetic / 12 bit /
 *             multi-scan), 4 corrupt, 5 size differs.  Planes of failed files are left untouched.
 * path_dev    optional [n] int32 (may be NULL): > 0 = decoded by the parallel Huffman kernel (the value is its number of
 *             synchronisation passes); <= 0 = by the sequential one (0: restart intervals / sequential requested / table
 *             problem, -1: no convergence, -2: stream ends early, -3: invalid symbols)
 * work_dev    >= df3d_jpeg_work_bytes(n, width, height, total_file_bytes) bytes, 256-byte aligned
 * flags       0, or DF3D_JPEG_SEQUENTIAL: skip the parallel (self-synchronising) Huffman kernel and decode every
 *             file with the sequential wave-per-file kernel (the exact fall-back the parallel one defers to for
 *             restart-interval, truncated or invalid streams); results are identical either way
 * ---------------------------------------------------------------------------------------------- */
#define DF3D_JPEG_SEQUENTIAL 1
size_t df3d_jpeg_work_bytes(int n, int width, int height, size_t total_file_bytes);
int df3d_jpeg_decode_luma(const unsigned char* files_dev, const unsigned* offsets_dev, const unsigned* sizes_dev, int n,
                          size_t total_file_bytes, unsigned max_file_bytes, int width, int height, unsigned char* luma_dev, int* status_dev,
                          int* path_dev, void* work_dev, size_t work_bytes, int flags, void* stream);

/* Host helper of the JPEG front-end (no device work): read n files with `threads` native threads into one staging buffer
 * (pinned host memory, typically) in the layout df3d_jpeg_decode_luma takes: file i at starts[i] (a multiple of 16, path
 * order, padding zeroed), sizes[i] bytes; *total_bytes = the bytes the layout needs (16 readable bytes follow it).  Returns
 * DF3D_ENOSPC, with *total_bytes set and nothing read, when dst is NULL or dst_bytes < *total_bytes + 16; DF3D_EIO (and the
 * path in df3d_last_error()) when a file cannot be opened or read.  The reference reads its frames one per DataLoader
 * worker call (df2d, behind reference df3d/core.py:177-185). */
int df3d_read_files(const char* const* paths, int n, unsigned char* dst, size_t dst_bytes, unsigned* starts, unsigned* sizes,
                    size_t* total_bytes, int threads);

/* ------------------------------------------------------------------------------------------------
 * a3  heat-map -> point + confidence.   Replaces df2d's heatmap2points / confidence extraction behind
 *     inference_folder(..., return_confidence=True)          (reference df3d/core.py:177-185,
 *     semantics reference README.md:404: arg-max over (h, w), confidence = the max value).
 * hm_dev   [n, joints, h, w] float32 planes (NCHW, the reference's heat-map layout)
 * pts_dev  [n, joints, 2]    float32   (row / h, col / w)   -- the reference's normalised (row, col)
 * conf_dev [n, joints]       float32   the peak value
 * Ties resolve to the first index in row-major order.  h*w must be a multiple of 4.
 * ---------------------------------------------------------------------------------------------- */
int df3d_heatmap_argmax(const float* hm_dev, int n, int joints, int h, int w, float* pts_dev, float* conf_dev,
                        void* stream);
/* (round 6) the same, plus the overflow guard of the reduced-precision engines: *nonfinite_planes_dev (device int, zeroed by the caller, may be
 * NULL) is incremented once for every plane that holds an infinity or a NaN anywhere -- the kernel reads every value anyway.  The F16 and F32S
 * hourglass engines need every activation inside the IEEE-half range (|x| < 65 504); a trained checkpoint that leaves it turns into inf / NaN
 * heat-maps and, without this flag, into a silently wrong points2d / heatmap_confidence (the bar it protects: reference tests/test_df3d.py:167-178).
 * The host reads the counter once per recording (deepfly3d_amd/inference.py) and refuses the result, naming the dtype to rerun with. */
int df3d_heatmap_argmax_checked(const float* hm_dev, int n, int joints, int h, int w, float* pts_dev, float* conf_dev,
                                int* nonfinite_planes_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a4  19 -> 38 joint re-layout and un-flip.   Replaces reference df3d/core.py:187-203.
 * pts19_dev [7, T, 19, 2] float32 (network order), ordering[7] HOST ints (camera_ordering),
 * out_dev   [7, T, 38, 2] float64 normalised (row, col).
 * ---------------------------------------------------------------------------------------------- */
int df3d_relayout_19_to_38(const float* pts19_dev, const int* ordering_host, int T, double* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a6  multi-view DLT triangulation.   Replaces pyba CameraNetwork.triangulate()
 *     (call site reference df3d/core.py:355).
 * P          [ncam, 3, 4] float64   K[R|t], pixel units (device OR host pointer; 84 doubles travel as
 *            a kernel argument)
 * pts_px_dev [ncam, T, J, 2] float64 (row_px, col_px); a camera sees (t, j) iff both are != 0
 * X_dev      [T, J, 3] float64; 0 where fewer than 2 cameras see the joint.      ncam <= 8.
 * ---------------------------------------------------------------------------------------------- */
int df3d_triangulate(const double* P, const double* pts_px_dev, int ncam, int T, int J, double* X_dev,
                     void* stream);
/* same, taking NORMALISED (row, col) detections and multiplying by (row_scale, col_scale) = (H, W) in the kernel,
 * i.e. the reference's `points2d * image_shape[::-1]` (df3d/core.py:247) fused into the triangulation */
int df3d_triangulate_scaled(const double* P, const double* pts_norm_dev, double row_scale, double col_scale, int ncam,
                            int T, int J, double* X_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a9  per-side Procrustes registration to the template pose + the `Core.get_points3d` chain.
 *     Replaces reference df3d/procrustes.py:51-151 (`procrustes_seperate`, call site df3d/core.py:358,340),
 *     df3d/plot_util.py:85-91 (`normalize_pose_3d`, call site core.py:341) and df3d/signal_util.py:69-100
 *     (`filter_batch`, call site core.py:342).  All float64, SEQUENCE-GLOBAL (medians over every frame): call
 *     once on the gathered sequence.  Several small launches on `stream`; `work` is caller-owned scratch.
 *
 * df3d_column_median : cols_dev [ncols] columns of n doubles, column c at cols_dev + c*col_stride;
 *                      out_dev[c] = median (exact order statistic; mean of the two middle values for even n).
 * df3d_procrustes    : pts_dev / out_dev [T, 38, 3]; tmpl_seg_med HOST [2][12] = per side, median over the
 *                      template's frames of the 12 leg-segment lengths; tmpl_fit_med HOST [2][6][3] = median of
 *                      the template's 6 fit joints (side joints 0,1,5,6,10,11).  work_dev >=
 *                      df3d_procrustes_work_doubles(T) doubles.
 * df3d_pose_normalize: out = in - median over all T*njoints points (per axis); rotate != 0 additionally maps
 *                      (x, y, z) -> (x, -z, -y).  work_dev >= 3 doubles; with >= 1024 doubles the medians of sequences of
 *                      more than 65 536 values per axis are taken by many workgroups (same values).  in == out allowed.
 * df3d_oneeuro_filter: in/out [T, nch]; one independent One-Euro filter per channel; sample i carries the time
 *                      stamp (i + first_stamp) * stamp_step and, like the reference, the filter re-derives its
 *                      sampling frequency from consecutive stamps.  The reference's `filter_batch` is
 *                      (freq 100, mincutoff 0.1, beta 2.0, dcutoff 1.0, first_stamp 1, stamp_step 0.1).
 *                      Bit-identical to the reference's float arithmetic.
 * ---------------------------------------------------------------------------------------------- */
int df3d_column_median(const double* cols_dev, int ncols, long long n, long long col_stride, double* out_dev, void* stream);
long long df3d_procrustes_work_doubles(long long T);
int df3d_procrustes(const double* pts_dev, long long T, const double* tmpl_seg_med, const double* tmpl_fit_med,
                    double* out_dev, double* work_dev, long long work_doubles, void* stream);
int df3d_pose_normalize(const double* in_dev, long long T, int njoints, int rotate, double* out_dev, double* work_dev,
                        long long work_doubles, void* stream);
int df3d_oneeuro_filter(const double* in_dev, long long T, int nch, double freq, double mincutoff, double beta,
                        double dcutoff, long long first_stamp, double stamp_step, double* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a7  bundle adjustment building blocks.   Replaces the arithmetic under pyba
 *     CameraNetwork.bundle_adjust(update_intrinsic=False, update_distort=False)
 *     (call site reference df3d/core.py:249).  Unknowns x = [ncam x (rvec, tvec)] ++ [npts x XYZ];
 *     observation i links camera cam_idx[i] to point pt_idx[i]; observations of one point are
 *     contiguous: point p owns observations [pt_start[p], pt_start[p+1]).
 * All arrays are device float64 / int32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct df3d_ba_problem {
    int ncam;              /* <= 8                                                        */
    int nobs;              /* number of 2-D observations (residual vector has 2*nobs)     */
    int npts;              /* number of 3-D points                                        */
    const double* intr4;   /* [ncam, 4]  fx, fy, cx, cy                                   */
    const double* obs_xy;  /* [nobs, 2]  x = col_px, y = row_px                           */
    const int* cam_idx;    /* [nobs]                                                      */
    const int* pt_idx;     /* [nobs]                                                      */
    const int* pt_start;   /* [npts + 1]  observations of point p: [pt_start[p], pt_start[p+1]) */
    const int* cam_perm;   /* [nobs]      observation ids grouped by camera (stable order)      */
    const int* cam_start;  /* [ncam + 1]  camera c owns cam_perm[cam_start[c] : cam_start[c+1]] */
} df3d_ba_problem;

/* Jacobian storage is component-major so per-observation threads read/write coalesced:
 *   Jc_dev [12, nobs]  entry (row*6 + col, i) = d r[2i+row] / d cam(cam_idx[i])[col]   (rvec 0..2, tvec 3..5)
 *   Jp_dev [ 6, nobs]  entry (row*3 + col, i) = d r[2i+row] / d point(pt_idx[i])[col]
 * scratch_dev arguments need DF3D_BA_SCRATCH_DOUBLES doubles; reductions run in a fixed order, so
 * results are bit-reproducible run to run. */
#define DF3D_BA_SCRATCH_DOUBLES 4096

/* residuals r[2*nobs] (interleaved x0,y0,x1,y1,...) and analytic Jacobian blocks at x; any of
 * r_dev / Jc_dev / Jp_dev may be NULL to skip it (Jc and Jp go together). */
int df3d_ba_eval(const df3d_ba_problem* p, const double* x_dev, double* r_dev, double* Jc_dev, double* Jp_dev,
                 void* stream);
/* colsq[n] = sum over rows of J[:, k]^2  (for scipy's x_scale='jac').  n = 6*ncam + 3*npts */
int df3d_ba_colsq(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, double* colsq_dev,
                  double* scratch_dev, void* stream);
/* y[2*nobs] = J * (d .* v)      (d_dev may be NULL = ones) */
int df3d_ba_matvec(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, const double* d_dev,
                   const double* v_dev, double* y_dev, void* stream);
/* w[n] = d .* (J^T u)           (d_dev may be NULL = ones) */
int df3d_ba_rmatvec(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, const double* d_dev,
                    const double* u_dev, double* w_dev, double* scratch_dev, void* stream);

/* LSMR on A = J*diag(d) with Tikhonov `damp`, the inner solve of scipy's TRF (tr_solver='lsmr').
 * Synchronous: returns when the solve has stopped; x_dev[n] receives the solution.  Every scalar of the recurrence lives on the device.
 * work_dev: at least df3d_ba_lsmr_work_doubles(p) doubles (includes the scratch).  info_host[8] =
 * {istop, itn, normr, normar, normA, condA, normx, fallback} (fallback: 0, or why a persistent form handed the run to the launch-based
 * one: 1 its workgroups never became co-resident, 2 the problem does not fit it).
 * Forms (df3d_ba_lsmr_form; df3d_ba_lsmr = DF3D_LSMR_AUTO, the environment variable DF3D_LSMR_KERNELS = 0 | 1 | 2 | 11 overrides AUTO):
 *   DF3D_LSMR_LOCAL     ONE persistent kernel per solve: every workgroup owns a range of points with their observations, Jacobian slice and
 *                       vectors in registers, two small all-reduces per iteration, one read-back per solve (round 5).  Up to 128 ranges of
 *                       1 024 observations (every window of <= 1 000 frames); wants the device to itself: up to 128 workgroups of 512
 *                       threads and 96 KB of LDS have to be resident at once.  Sums grouped per range: last-bit differences from the others.
 *   DF3D_LSMR_LAUNCHES  two kernels per iteration, 16 iterations per host read-back (round 4): needs no co-residency -- the form to use
 *                       when the solve runs BESIDE other work on the device (a re-calibration next to the frame pipeline).
 *   DF3D_LSMR_BARRIERS  the launch-based arithmetic in one persistent kernel with grid-wide barriers (round 5; measured slower than
 *                       LAUNCHES: profiles/r05_ba_timings.txt);  DF3D_LSMR_ELEVEN  round 3's eleven kernels per iteration.
 *                       LAUNCHES, BARRIERS and ELEVEN produce the same bits.
 *   DF3D_LSMR_AUTO      LOCAL where the problem fits, else LAUNCHES. */
#define DF3D_LSMR_AUTO 0
#define DF3D_LSMR_BARRIERS 1
#define DF3D_LSMR_LAUNCHES 2
#define DF3D_LSMR_LOCAL 3
#define DF3D_LSMR_ELEVEN 11
size_t df3d_ba_lsmr_work_doubles(const df3d_ba_problem* p);
int df3d_ba_lsmr(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, const double* d_dev,
                 const double* b_dev, double damp, double atol, double btol, double conlim, int maxiter,
                 double* x_dev, double* work_dev, double* info_host, void* stream);
int df3d_ba_lsmr_form(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, const double* d_dev,
                      const double* b_dev, double damp, double atol, double btol, double conlim, int maxiter,
                      double* x_dev, double* work_dev, double* info_host, void* stream, int form);

/* small device-vector helpers used by the host TRF driver (all float64, asynchronous except dot) */
int df3d_vec_dot(const double* a_dev, const double* b_dev, size_t n, double* result_host, double* scratch_dev,
                 void* stream); /* synchronous; scratch_dev >= 1024 doubles */
/* `count` (1..8) dot products a_dev[j] . b_dev[j] of lengths n[j] in ONE launch and ONE synchronising read-back (results_host[count]); every
 * product is summed exactly as df3d_vec_dot sums it (same grid, same order: the same bits).  The trust-region driver of a7 needs its
 * scalars in groups (the 2 x 2 subspace system: five products) -- round 4, reference call site df3d/core.py:249.
 * scratch_dev: DF3D_BA_SCRATCH_DOUBLES doubles. */
int df3d_vec_dots(int count, const double* const* a_dev, const double* const* b_dev, const size_t* n, double* results_host,
                  double* scratch_dev, void* stream); /* synchronous; scratch_dev >= DF3D_BA_SCRATCH_DOUBLES (uses 8*256+8) */
int df3d_vec_axpby(double a, const double* x_dev, double b, const double* y_dev, double* out_dev, size_t n,
                   void* stream); /* out = a*x + b*y (y may be NULL)        */
int df3d_vec_mul(const double* x_dev, const double* y_dev, double* out_dev, size_t n, void* stream);
int df3d_vec_absmax(const double* a_dev, size_t n, double* result_host, double* scratch_dev, void* stream); /* sync */
/* sum over npairs (x, y) pairs of sqrt(x^2 + y^2): with the residuals of df3d_ba_eval this is nobs x the mean
 * re-projection distance in pixels that CameraNetwork.reprojection_error() reports (call site reference
 * df3d/core.py:250); fixed summation order; synchronises like df3d_vec_dot */
int df3d_vec_pairnorm_sum(const double* r_dev, size_t npairs, double* result_host, double* scratch_dev, void* stream);
/* scipy's x_scale='jac': scale_inv = sqrt(colsq) (0 -> 1 when first != 0, else max with the previous value),
 * scale = 1 / scale_inv */
int df3d_ba_update_scale(const double* colsq_dev, double* scale_inv_dev, double* scale_dev, size_t n, int first,
                         void* stream);

/* The trust-region driver with its scalars on the device (round 6): what deepfly3d_amd/bundle_adjust.py:solve_trf does between two
 * evaluations, as three calls with ONE stream synchronisation each instead of one per scalar group.  They replace the solver loop of
 * scipy.optimize.least_squares(method='trf', tr_solver='lsmr', x_scale='jac') that pyba runs (reference call site df3d/core.py:249);
 * the arithmetic is that of the separate calls above, operation for operation (same kernels, same fixed summation orders).
 * All vectors are device float64: n = 6 ncam + 3 npts columns, m = 2 nobs rows; scratch_dev: DF3D_BA_SCRATCH_DOUBLES; work_dev:
 * df3d_ba_lsmr_work_doubles(p).
 * df3d_ba_trf_subspace: from (J, scale, g = J^T f, f, Delta): |g|_inf, g_h = scale g, the damping of the 1-D Cauchy model, the LSMR step
 *   gn_h (damping read from device memory by the data-local form; the other forms take one more read-back), the orthonormal basis
 *   S = [s0 s1] of span{g_h, gn_h} (LAPACK's QR signs), J_h S and the 2x2 model.  out_host[19]: |g|_inf, |J_h g_h|^2, |g_h|^2, damp, r01,
 *   |s1|^2 (before normalisation), B_S[0][0], B_S[0][1], B_S[1][1], g_S[0], g_S[1], then df3d_ba_lsmr's info[8].
 * df3d_ba_trf_trial: step_h = p0 s0 + p1 s1, x_new = x + scale step_h, f_new = residuals(x_new); out_host[6]: |J_h step_h|^2, step_h . g_h,
 *   |step_h|^2, |f_new|^2, |step|^2, |x|^2.
 * df3d_ba_trf_linearize: J (and f when eval_f != 0) at x, g = J^T f, x_scale='jac' bookkeeping (first != 0: the first call); no read-back. */
int df3d_ba_trf_subspace(const df3d_ba_problem* p, const double* Jc_dev, const double* Jp_dev, const double* scale_dev, const double* g_dev,
                         const double* f_dev, double Delta, double* g_h_dev, double* gn_h_dev, double* s0_dev, double* s1_dev, double* Js0_dev,
                         double* Js1_dev, double* tmp_m_dev, double* work_dev, double* scratch_dev, double* out_host, void* stream, int form);
int df3d_ba_trf_trial(const df3d_ba_problem* p, double p0, double p1, const double* s0_dev, const double* s1_dev, const double* Js0_dev,
                      const double* Js1_dev, const double* scale_dev, const double* x_dev, const double* g_h_dev, double* step_h_dev,
                      double* tmp_m_dev, double* step_dev, double* x_new_dev, double* f_new_dev, double* scratch_dev, double* out_host, void* stream);
int df3d_ba_trf_linearize(const df3d_ba_problem* p, const double* x_dev, double* f_dev, int eval_f, double* Jc_dev, double* Jp_dev, double* g_dev,
                          double* colsq_dev, double* scale_inv_dev, double* scale_dev, int first, double* scratch_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a2  stacked-hourglass forward.   Replaces the network forward inside df2d's inference_folder
 *     (call site reference df3d/core.py:177-185; 2 stacks / 19 maps / 64x128 heat-maps:
 *     reference df3d/config.py:18,33,36).
 *
 * The engine owns the layer plan (which convolution follows which); the caller owns all memory:
 *   1. df3d_hg_create(dtype, num_stacks, &h)
 *   2. for i in [0, df3d_hg_num_params(h)): df3d_hg_param_desc(h, i, &d) names one packed tensor
 *      (e.g. "layer1.0.conv1") with its shape; the caller fills a float32 host blob of
 *      df3d_hg_blob_floats(h) floats at d.offset and uploads it:  df3d_hg_set_weights(h, blob_dev, ...).
 *      Packed layout per convolution: weight [taps][cout_pad][cin_pad] (k contiguous), bias[cout_pad],
 *      and for pre-activated convs in_scale[cin_pad], in_shift[cin_pad] (eval-mode BN as y = x*s + t).
 *      The stem ("conv1", taps = 49) is packed [148][64]: row k = ky*21 + kx*3 + c, 64 outputs contiguous (its slot
 *      is 184*64 floats: bf16 engines re-lay the low-precision copy as a [64][184] tile; leave the tail zero).
 *      BatchNorms that FOLLOW a convolution are folded into its weight/bias by the caller (the Python packer
 *      deepfly3d_amd/hourglass.py does this in float64).
 *   3. df3d_hg_forward(h, images_dev, n, heatmaps_dev, workspace_dev, workspace_bytes, stream)
 *      images_dev   [n, 256, 512, 3] float32 NHWC      (any H, W multiple of 64 via df3d_hg_set_input)
 *      heatmaps_dev [n, 19, H/4, W/4] float32 NCHW
 *      workspace    >= df3d_hg_workspace_bytes(h, n)
 * ---------------------------------------------------------------------------------------------- */
typedef struct df3d_hg df3d_hg;

typedef struct df3d_hg_param {
    char name[64];   /* e.g. "hg.0.hg.3.0.0.conv2"                                    */
    int kind;        /* 0 weight, 1 bias, 2 in_scale, 3 in_shift                       */
    int taps;        /* 1, 9 or 49                                                     */
    int cin, cout;   /* logical sizes                                                  */
    int cin_pad, cout_pad;
    size_t offset;   /* in floats, into the blob                                       */
    size_t count;    /* in floats                                                      */
    int kperm;       /* 1: weight K (cin) order is permuted inside every 32-channel group: packed position
                        8*(2q + h) + e holds channel 16q + 8*(e>>2) + 4h + (e&3)  (q, h in {0,1}, e in 0..7);
                        used by the bf16 fused bottleneck, whose conv3 consumes MFMA accumulators directly */
    int reserved;
} df3d_hg_param;

int df3d_hg_create(int dtype, int num_stacks, df3d_hg** out);
void df3d_hg_destroy(df3d_hg* h);
int df3d_hg_set_input(df3d_hg* h, int height, int width);
int df3d_hg_num_params(const df3d_hg* h);
int df3d_hg_param_desc(const df3d_hg* h, int i, df3d_hg_param* out);
size_t df3d_hg_blob_floats(const df3d_hg* h);
/* Engine-private device copies of the weights, made by df3d_hg_set_weights in a caller-owned buffer of
 * df3d_hg_lowp_bytes(h) bytes (256-byte aligned; NULL is accepted when that size is 0): the bf16 copy of the blob (bf16
 * engines) and, for both dtypes, the "weight streams" of the 256 -> 128 -> 128 -> 256 bottlenecks -- their weights repacked
 * as the sequence of 8 KB LDS images the kernel pulls through its LDS-DMA ring (option "ring", default 1); bf16 engines also
 * keep the heads' fc / fc_ / score_ weights in that form and layer1's whole weight set as one LDS image (option "l1").
 * DF3D_DTYPE_F32S: a float32-sized copy of the blob with every 16-float K step of every weight row stored as its IEEE-half hi / lo
 * parts in MFMA operand order (and the stem's weights as two half-precision tiles), followed by the streams packed from that copy;
 * lowp_dev must not be NULL. */
size_t df3d_hg_lowp_bytes(const df3d_hg* h);
int df3d_hg_set_weights(df3d_hg* h, const float* blob_dev, void* lowp_dev, void* stream);
/* knobs: "fuse" = 1 (default) | 0: run 256->128->128->256 bottlenecks as one fused kernel -- must be set before
 * the weights (it changes the manifest);  "fuse_upadd" = 1 (default) | 0: the hourglass'
 * nearest-upsample + add is folded into the input load of the bottleneck that consumes the sum (same results bit
 * for bit; set before the weights);  "row_bytes" = 0 (auto) | 64 | 128 bytes staged per operand row per K-step;
 * "ring" = 1 (default) | 0: weights through LDS-DMA stage rings (see df3d_hg_lowp_bytes) or register-staged;  "l1" = 1
 * (default) | 0: bf16 layer1 as the LDS-resident-weights kernel that writes only the pooled tensor -- both set before the
 * weights, both bit-identical to their 0 form;  "split1" = 1 (default) | 0 (fp32): conv1 of every bottleneck (layer1 / layer2 included) once per
 * pixel in a kernel of its own, the rest in tail kernels (17 MB more workspace per view: query df3d_hg_workspace_bytes);  "w2d" = 1 (default) | 0 (16-bit): the 3x3's weights of the ring bottlenecks as per-wave MFMA
 * fragments loaded straight from global memory (288 KB more stream space per bottleneck) -- both before the weights, both
 * bit-identical to their 0 form;  "chain_views" = 0 (default) | n: chains of full-resolution steps in chunks of n views;
 * "no_reuse" = 0 (default) | 1: the alias-free workspace plan (no tensor is ever given memory another tensor has released: ~5x the workspace) --
 * the reference form the aliasing tests compare the default plan with, bit for bit; before the weights;
 * "wino" = 1 (default) | 0 (exact fp32 with "split1"): the 3x3 of the 256->128->128->256 identity blocks as Winograd F(2x2, 3x3) -- 16 instead of 36 multiplies per
 * 2x2 output patch and (cin, cout) pair, 1 MB more stream space per block; the SAME float32 tolerance against the reference arithmetic, but NOT bit-identical
 * to the direct form ("wino" = 0: the bit-identity reference of the other options); before the weights;
 * "c1res" = 1 (default) | 0 (with "wino"): conv1 of the plain identity blocks with its 128 KB of weights resident in LDS (128 KB more stream space per block,
 * allocated with "wino") instead of streamed through the LDS ring -- bit-identical either way; may be set between forwards */
int df3d_hg_set_option(df3d_hg* h, const char* key, int value);
size_t df3d_hg_workspace_bytes(const df3d_hg* h, int n);
int df3d_hg_forward(df3d_hg* h, const float* images_dev, int n, float* heatmaps_dev, void* workspace_dev,
                    size_t workspace_bytes, void* stream);
/* The same forward pass fed with the camera frames themselves: frames_dev [n][frame_h][frame_w][frame_c] uint8 (frame_c = 1 or
 * 3), flip_dev [n] uint8 or NULL (non-zero: mirror the frame left-right).  The stem samples its input patches from the frames
 * with the arithmetic of df3d_preprocess_u8 (`resize` rule to the engine's input size, (v/255 - mean) / std), so the result is bit
 * for bit df3d_hg_forward(df3d_preprocess_u8(frames)) without the float image in between (the call df2d's dataset +
 * network make per batch, behind reference df3d/core.py:177-185). */
int df3d_hg_forward_u8(df3d_hg* h, const unsigned char* frames_dev, const unsigned char* flip_dev, int n, int frame_h, int frame_w,
                       int frame_c, const float* mean3_host, const float* std3_host, int resize, float* heatmaps_dev, void* workspace_dev,
                       size_t workspace_bytes, void* stream);
/* algorithmic work of one forward over n views: FLOPs and activation bytes (fusion model M1 of
 * SURVEY.md 8d evaluated on this engine's own plan) */
int df3d_hg_work(const df3d_hg* h, int n, double* flops, double* bytes);
/* per-kernel timing with HIP events recorded on the launch stream around every launch (small overhead: enable it
 * for a measurement pass only).  Launches are grouped by kernel instantiation, named as rocprofv3 prints them
 * (e.g. "bottleneck_kernel<float, 256, 128, false>").  df3d_hg_profile(h, 1) clears previous samples;
 * df3d_hg_profile_count() = number of distinct kernels seen; df3d_hg_profile_read() waits for the events and
 * returns name, summed duration (ms), algorithmic FLOPs, two byte figures and the launch count of kernel `index`:
 * bytes = the minimum the launches can move (their inputs read once, their outputs written once, fused intermediates on
 * chip), bytes_m1 = what the fusion model M1 of SURVEY.md 8(d) charges for the same plan steps (every convolution's input
 * and output once, pooling and upsample passes) -- the convention df3d_hg_work() sums.  (Round 3 added bytes_m1.) */
int df3d_hg_profile(df3d_hg* h, int enable);
int df3d_hg_profile_count(const df3d_hg* h);
int df3d_hg_profile_read(df3d_hg* h, int index, char* name_buf, int buflen, double* ms, double* flops, double* bytes, double* bytes_m1,
                         int* launches);
/* (round 6) FLOPs the launches of kernel `index` EXECUTED on the matrix pipe, summed like df3d_hg_profile_read's `flops`.  Equal to `flops` (the
 * direct-convolution count of the plan steps, the reference's arithmetic) for every kernel except the Winograd tail of option "wino", which does a
 * 3x3 with 16/36 of the direct form's multiplies: a roofline fraction is executed FLOPs over peak, never direct-equivalent FLOPs over peak. */
int df3d_hg_profile_executed_flops(df3d_hg* h, int index, double* flops_executed);
/* debugging / layer-wise parity: number of plan steps, and run only steps [0, upto) then copy the
 * tensor produced by step upto-1 (NHWC, engine dtype widened to float32) into out_dev */
int df3d_hg_num_steps(const df3d_hg* h);
/* model-M1 bytes of plan step `step` over n views (the steps' values sum to df3d_hg_work()'s bytes) */
double df3d_hg_step_m1_bytes(const df3d_hg* h, int step, int n);
int df3d_hg_step_desc(const df3d_hg* h, int step, char* name_buf, int buflen, int* n_h_w_c /*[3]: h, w, c*/);
int df3d_hg_forward_upto(df3d_hg* h, const float* images_dev, int n, int upto, float* out_dev, void* workspace_dev,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * f4  frames of the pose videos (reference df3d/video.py:21-108, called from df3d/cli.py:308-321).  One launch draws one frame:
 *     every output pixel tests itself against the joints and bones of its camera (rule restated in oracle/render.py).
 * df3d_render_pose2d_grid: luma_dev [6, height, width] uint8 = the images of cameras (0, 1, 2 / 4, 5, 6) in that order;
 *     points_px_dev [6, num_joints, 2] float64 (row_px, col_px), a joint with a 0 coordinate is unseen; bones_host [num_bones, 2] and
 *     joint_rgb_host [num_joints, 3] (colour of a joint; a bone takes its first joint's) are HOST tables (<= 64 joints, <= 96 bones);
 *     out_rgb_dev [2 height, 3 width, 3] uint8: grey image, bone segments of thickness line_width, joint discs of `radius` on top.
 * df3d_render_pose3d_panels: points3d_dev [num_joints, 3] float64 (the pose Core.get_points3d returns for one image), three square
 *     panels [size, 3 size, 3] uint8, orthographic views from azimuth_deg3[k] / elevation_deg (matplotlib's view_init angles,
 *     reference df3d/plot_util.py:48-51), +-lim mapped onto the panel, black background.
 * df3d_resize_rgb: bilinear, pixel centres at half integers (cv2.INTER_LINEAR's geometry); pitches in pixels.
 * All asynchronous on `stream`. */
int df3d_render_pose2d_grid(const unsigned char* luma_dev, int height, int width, const double* points_px_dev, int num_joints,
                            const int* bones_host, int num_bones, const unsigned char* joint_rgb_host, double radius, double line_width,
                            unsigned char* out_rgb_dev, void* stream);
int df3d_render_pose3d_panels(const double* points3d_dev, int num_joints, const int* bones_host, int num_bones,
                              const unsigned char* joint_rgb_host, const double* azimuth_deg3, double elevation_deg, double lim, int size,
                              double line_width, unsigned char* out_rgb_dev, void* stream);
int df3d_resize_rgb(const unsigned char* in_dev, int in_h, int in_w, int in_pitch_px, unsigned char* out_dev, int out_h, int out_w,
                    int out_pitch_px, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DF3D_HIP_H */
