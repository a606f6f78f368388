// The scalar side of one LSMR iteration as three single-workgroup kernels (see ba_lsmr.h).  The arithmetic follows
// scipy.sparse.linalg.lsmr (the solver under the reference's bundle adjustment) statement by
// statement, as restated in oracle/trf_lsmr.py:lsmr; this file is compiled with -ffp-contract=off (build.py), so every
// product and sum rounds on its own, exactly as the CPython / numpy float arithmetic of the oracle does.
#include <cmath>

#include "ba_lsmr.h"

namespace df3d_lsmr {
namespace {

__device__ void sym_ortho(double a, double b, double& c, double& s, double& r) {
    auto sgn = [](double v) { return (double)((v > 0) - (v < 0)); };
    if (b == 0) {
        c = sgn(a);
        s = 0;
        r = fabs(a);
    } else if (a == 0) {
        c = 0;
        s = sgn(b);
        r = fabs(b);
    } else if (fabs(b) > fabs(a)) {
        const double tau = a / b;
        s = sgn(b) / sqrt(1 + tau * tau);
        c = s * tau;
        r = b / s;
    } else {
        const double tau = b / a;
        c = sgn(a) / sqrt(1 + tau * tau);
        s = c * tau;
        r = a / c;
    }
}

__device__ __forceinline__ double sum_partials(const double* __restrict__ partial, int count, double* lds4) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial[i];
    return block_reduce_256(acc, lds4);
}

__global__ __launch_bounds__(256) void step_a_kernel(State* st, const double* __restrict__ partial, int count) {
    __shared__ double lds4[4];
    if (st->istop) return;
    const double ss = sum_partials(partial, count, lds4);
    if (threadIdx.x) return;
    ++st->itn;
    const double beta = sqrt(ss);
    st->beta = beta;
    st->beta_pos = beta > 0;
    st->inv_beta = beta > 0 ? 1.0 / beta : 0.0;
}

__global__ __launch_bounds__(256) void step_b_kernel(State* st, const double* __restrict__ partial, int count) {
    __shared__ double lds4[4];
    if (st->istop) return;
    const double ss = sum_partials(partial, count, lds4);
    if (threadIdx.x) return;
    State S = *st;
    if (S.beta_pos) {
        S.alpha = sqrt(ss);
        S.inv_alpha = S.alpha > 0 ? 1.0 / S.alpha : 1.0;  // alpha == 0: v stays as it is
    } else {
        S.inv_alpha = 1.0;
    }
    double alphahat;
    sym_ortho(S.alphabar, S.damp, S.chat, S.shat, alphahat);
    const double rhoold = S.rho;
    sym_ortho(alphahat, S.beta, S.c, S.sn, S.rho);
    const double thetanew = S.sn * S.alpha;
    S.alphabar = S.c * S.alpha;
    S.rhobarold = S.rhobar;
    S.zetaold = S.zeta;
    S.thetabar = S.sbar * S.rho;
    S.rhotemp = S.cbar * S.rho;
    sym_ortho(S.cbar * S.rho, thetanew, S.cbar, S.sbar, S.rhobar);
    S.zeta = S.cbar * S.zetabar;
    S.zetabar = -S.sbar * S.zetabar;
    S.c1 = S.thetabar * S.rho / (rhoold * S.rhobarold);
    S.c2 = S.zeta / (S.rho * S.rhobar);
    S.c3 = thetanew / S.rho;
    *st = S;
}

__global__ __launch_bounds__(256) void step_c_kernel(State* st, const double* __restrict__ partial, int count) {
    __shared__ double lds4[4];
    if (st->istop) return;
    const double ss = sum_partials(partial, count, lds4);
    if (threadIdx.x) return;
    State S = *st;
    S.normx = sqrt(ss);
    const double betaacute = S.chat * S.betadd;
    const double betacheck = -S.shat * S.betadd;
    const double betahat = S.c * betaacute;
    S.betadd = -S.sn * betaacute;
    const double thetatildeold = S.thetatilde;
    double ctildeold, stildeold, rhotildeold;
    sym_ortho(S.rhodold, S.thetabar, ctildeold, stildeold, rhotildeold);
    S.thetatilde = stildeold * S.rhobar;
    S.rhodold = ctildeold * S.rhobar;
    S.betad = -stildeold * S.betad + ctildeold * betahat;
    S.tautildeold = (S.zetaold - thetatildeold * S.tautildeold) / rhotildeold;
    const double taud = (S.zeta - S.thetatilde * S.tautildeold) / S.rhodold;
    S.dd += betacheck * betacheck;
    S.normr = sqrt(S.dd + (S.betad - taud) * (S.betad - taud) + S.betadd * S.betadd);
    S.normA2 += S.beta * S.beta;
    S.normA = sqrt(S.normA2);
    S.normA2 += S.alpha * S.alpha;
    S.maxrbar = fmax(S.maxrbar, S.rhobarold);
    if (S.itn > 1) S.minrbar = fmin(S.minrbar, S.rhobarold);
    S.condA = fmax(S.maxrbar, S.rhotemp) / fmin(S.minrbar, S.rhotemp);
    S.normar = fabs(S.zetabar);
    const double test1 = S.normr / S.normb;
    const double test2 = (S.normA * S.normr) != 0 ? S.normar / (S.normA * S.normr) : INFINITY;
    const double test3 = 1.0 / S.condA;
    const double t1 = test1 / (1 + S.normA * S.normx / S.normb);
    const double rtol = S.btol + S.atol * S.normA * S.normx / S.normb;
    int istop = 0;
    if (S.itn >= S.maxiter) istop = 7;
    if (1 + test3 <= 1) istop = 6;
    if (1 + test2 <= 1) istop = 5;
    if (1 + t1 <= 1) istop = 4;
    if (test3 <= S.ctol) istop = 3;
    if (test2 <= S.atol) istop = 2;
    if (test1 <= rtol) istop = 1;
    S.istop = istop;
    *st = S;
}

}  // namespace

void launch_step_a(State* st, const double* partial, int count, hipStream_t s) {
    hipLaunchKernelGGL(step_a_kernel, dim3(1), dim3(256), 0, s, st, partial, count);
}
void launch_step_b(State* st, const double* partial, int count, hipStream_t s) {
    hipLaunchKernelGGL(step_b_kernel, dim3(1), dim3(256), 0, s, st, partial, count);
}
void launch_step_c(State* st, const double* partial, int count, hipStream_t s) {
    hipLaunchKernelGGL(step_c_kernel, dim3(1), dim3(256), 0, s, st, partial, count);
}

}  // namespace df3d_lsmr
