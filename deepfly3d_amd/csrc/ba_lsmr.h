// LSMR (Fong & Saunders) with its scalar recurrences kept ON THE DEVICE: shared between ba.hip (vector kernels,
// host driver) and ba_lsmr.hip (the three scalar steps of an iteration, compiled without multiply-add fusion so that
// they round exactly like the numpy/scipy scalar code the oracle runs).
#pragma once
#include <hip/hip_runtime.h>

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

// step A: beta = |u| from `count` partials;  step B: alpha = |v| and the rotations;  step C: |x| and the stopping tests.
// One workgroup each; no-ops once st->istop != 0.
void launch_step_a(State* st, const double* partial, int count, hipStream_t s);
void launch_step_b(State* st, const double* partial, int count, hipStream_t s);
void launch_step_c(State* st, const double* partial, int count, hipStream_t s);

}  // namespace df3d_lsmr
