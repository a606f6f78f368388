// LSMR (Fong & Saunders) with its scalar recurrences kept ON THE DEVICE: shared between ba.hip (vector kernels,
// host driver) and ba_lsmr.hip (the three scalar steps of an iteration, compiled without multiply-add fusion so that
// they round exactly like the numpy/scipy scalar code the oracle runs).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/df3d_hip.h"

namespace df3d_lsmr {

struct State {
    // Golub-Kahan bidiagonalisation and the two rotations
    double alpha, beta, rho, rhobar, cbar, sbar, zeta, zetabar, alphabar;
    // residual-norm estimate
    double betadd, betad, rhodold, tautildeold, thetatilde, dd;
    // norms and condition estimate
    double normA2, maxrbar, minrbar, normA, condA, normx, normr, normar, normb;
    // parameters
    double damp, atol, btol, ctol;
    // carried from step B to step C of the same iteration
    double rhobarold, zetaold, thetabar, rhotemp, chat, shat, c, sn;
    // coefficients the vector kernels read
    double inv_beta, inv_alpha, c1, c2, c3;
    int itn, istop, maxiter, beta_pos;
};

// fixed-order block reduction of one double per thread (256 threads): wave butterflies then 4 -> 1 in LDS
__device__ __forceinline__ double block_reduce_256(double v, double* lds4) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[wave] = v;
    __syncthreads();
    return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// ---- round 4: the iteration as TWO kernels (ba_lsmr.hip: fused_ka / fused_kb) instead of eleven -----------------------------------------
// Each kernel starts by doing, in every workgroup, the scalar steps that used to be kernels of their own: it sums the previous kernel's
// per-workgroup partials in the fixed order of the single-workgroup kernels and runs the same scalar code, so all workgroups hold the same
// scalars without a grid-wide hand-off; workgroup 0 writes the state for the next kernel into the OTHER of two state slots (a workgroup
// that starts late must still find the state its kernel was launched against).  u and v are kept UN-normalised with 1 / beta, 1 / alpha
// in the state (the normalised element is formed where it is used, rounded like the stored one was), which removes the two scaling
// kernels; the 6 x ncam camera entries of v live in the state (`vcam`), because their J^T u sums are finished by the kernel AFTER the one
// that forms the partial sums.  The kernel boundaries sit where a grid-wide sum is needed and nowhere else:
//     ka_i = [step B of i - 1 (alpha, rotations); hbar, x, h of i - 1 -> partial |x|^2] + [u <- A v - alpha u of i -> partial |u|^2]
//     kb_i = [step C of i - 1 (stopping tests; a stopped run ends here); step A of i (beta)] + [v <- A^T u - beta v of i -> partial |v|^2]
// (a run's last matrix-vector product is computed in vain: it touches u only).
struct Fused {
    State s;
    double vcam[48];   // camera entries of v (un-normalised), 6 per camera
    int pending_c;     // step C of the previous iteration is still to be taken (0 in front of the first iteration)
    int pending_b;     // step B and the vector update of the previous iteration are still to be taken (likewise)
};
constexpr int FUSED_DOUBLES = 128;        // one state slot, in doubles (sizeof(Fused) rounded up)
constexpr int FUSED_RED = 256;            // partial sums per reduction array (= RED_BLOCKS of ba.hip)
static_assert(sizeof(Fused) <= FUSED_DOUBLES * sizeof(double), "state slot too small");
struct FusedArgs {
    const double *Jc, *Jp, *d;            // Jacobian blocks, column scaling
    double *u, *v, *h, *hbar, *x;         // u [m]; v, h, hbar, x [n] (v[0 .. 6 ncam) unused: see Fused::vcam)
    double *cam_partial;                  // [ncam][nchunk][6]
    double *red1, *red2, *red3;           // FUSED_RED partials each: |u|^2, |v|^2 (point entries), |x|^2
    double* st;                           // two state slots, FUSED_DOUBLES doubles apart
    int nchunk;                           // chunks per camera of the J^T u partial sums
    int g1, g2p, g3;                      // round 3's grids of the |u|^2, |v|^2, |x|^2 sums (= partials in red1 / red2 / red3): part of the arithmetic
};
// one iteration = ka (state slot 0 -> 1), kb (1 -> 0)
void launch_fused_iteration(const df3d_ba_problem& p, const FusedArgs& a, hipStream_t s);
// ka + the scalar half of kb: takes the pending steps B and C of the last iteration of a run that ends on maxiter
void launch_fused_flush(const df3d_ba_problem& p, const FusedArgs& a, hipStream_t s);

// round 5: the whole run (up to `maxiter` iterations + the flush) as ONE persistent kernel of `grid` workgroups with grid-wide barriers in
// the kernel boundaries' places; state in and out through slot 0; `bar`: two 32-bit words (arrival counter, failure flag), zeroed by the
// launch.  Bit-identical to launch_fused_iteration x maxiter (+ launch_fused_flush).  A run whose barrier timed out leaves istop = -1.
void launch_fused_persistent(const df3d_ba_problem& p, const FusedArgs& a, unsigned* bar, int maxiter, int grid, hipStream_t s);

// round 5: the DATA-LOCAL run (ba_lsmr.hip: lsmr_local_kernel): every workgroup owns a range of points with their observations for the
// whole run, Jacobian slice and vectors in registers, two small all-reduces per iteration.  Returns 0, or < 0 when the problem does not
// fit (more than local_max_workgroups() ranges: the caller takes another form).  The final State lands in state_out (istop = -1: timeout).
int local_workgroups_for(int nobs);
int local_max_workgroups();
size_t local_scratch_bytes();
int launch_local(const df3d_ba_problem& p, const double* Jc, const double* Jp, const double* d, const double* b, double* x, double damp, double atol,
                 double btol, double ctol, int maxiter, void* scratch, double* state_out, hipStream_t s, const double* damp_dev = nullptr);
// whether launch_local can take the problem at all (the layout itself is checked on the device: istop -2)
bool local_fits(const df3d_ba_problem& p);

// step A: beta = |u| from `count` partials;  step B: alpha = |v| and the rotations;  step C: |x| and the stopping tests.
// One workgroup each; no-ops once st->istop != 0.
void launch_step_a(State* st, const double* partial, int count, hipStream_t s);
void launch_step_b(State* st, const double* partial, int count, hipStream_t s);
void launch_step_c(State* st, const double* partial, int count, hipStream_t s);

}  // namespace df3d_lsmr
