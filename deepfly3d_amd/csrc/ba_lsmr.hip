// The scalar side of one LSMR iteration as three single-workgroup kernels (see ba_lsmr.h).  The arithmetic follows
// scipy.sparse.linalg.lsmr (the solver under the reference's bundle adjustment) statement by
// statement, as restated in oracle/trf_lsmr.py:lsmr; this file is compiled with -ffp-contract=off (build.py), so every
// product and sum rounds on its own, exactly as the CPython / numpy float arithmetic of the oracle does.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

// the three scalar steps on a state held by ONE thread (the single-workgroup kernels below and, round 4, the prologues of the fused
// kernels): `ss` = the sum of squares the step needs
__device__ void apply_step_a(State& S, double ss) {
    ++S.itn;
    const double beta = sqrt(ss);
    S.beta = beta;
    S.beta_pos = beta > 0;
    S.inv_beta = beta > 0 ? 1.0 / beta : 0.0;
}

__device__ void apply_step_b(State& S, double ss) {
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
}

__device__ void apply_step_c(State& S, double ss) {
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
}

__global__ __launch_bounds__(256) void step_a_kernel(State* st, const double* __restrict__ partial, int count) {
    __shared__ double lds4[4];
    if (st->istop) return;
    const double ss = sum_partials(partial, count, lds4);
    if (threadIdx.x) return;
    State S = *st;
    apply_step_a(S, ss);
    *st = S;
}

__global__ __launch_bounds__(256) void step_b_kernel(State* st, const double* __restrict__ partial, int count) {
    __shared__ double lds4[4];
    if (st->istop) return;
    const double ss = sum_partials(partial, count, lds4);
    if (threadIdx.x) return;
    State S = *st;
    apply_step_b(S, ss);
    *st = S;
}

__global__ __launch_bounds__(256) void step_c_kernel(State* st, const double* __restrict__ partial, int count) {
    __shared__ double lds4[4];
    if (st->istop) return;
    const double ss = sum_partials(partial, count, lds4);
    if (threadIdx.x) return;
    State S = *st;
    apply_step_c(S, ss);
    *st = S;
}

// ---- round 4: the iteration as two kernels (ba_lsmr.h) --------------------------------------------------------------------------
constexpr int FWORDS = (int)(sizeof(Fused) / sizeof(double));
static_assert(sizeof(Fused) % sizeof(double) == 0, "the state is copied as doubles");

// every workgroup: a private copy of the state slot in LDS (ends with a barrier)
__device__ __forceinline__ void load_state(Fused& F, const double* __restrict__ slot) {
    if (threadIdx.x < FWORDS) reinterpret_cast<double*>(&F)[threadIdx.x] = slot[threadIdx.x];
    __syncthreads();
}
// workgroup 0 publishes its copy for the next kernel (call with the LDS copy final and a barrier behind its last write)
__device__ __forceinline__ void store_state(double* __restrict__ slot, const Fused& F) {
    if (blockIdx.x == 0 && threadIdx.x < FWORDS) slot[threadIdx.x] = reinterpret_cast<const double*>(&F)[threadIdx.x];
}

// The vector loops below reproduce, element for element and partial sum for partial sum, what the eleven kernels of round 3 computed
// (ba.hip: matvec, bidiag, scale, cam partial, rmatvec final, update -- compiled WITH multiply-add fusion, hence the pragma inside each
// loop's function; this file's default is no fusion, for the scalar steps): same products, same fused operations, same grid-stride
// grouping of the sums of squares, so alpha, beta, |x| and with them the iteration counts are those of round 3.

// a product that rounds on its own and can never be fused into a later add or subtract: what round 3 stored to memory in one kernel and
// loaded in the next (w = d * acc, the normalised u and v) is formed in registers here, and a multiply-add fusion across that former
// store would change the bits
__device__ __forceinline__ double rounded_product(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}

// row e of u <- (A (D v))_e - alpha u_e for this thread's rows e = first, first + stride, ... ; returns the thread's sum of squares
__device__ double k1_rows(const df3d_ba_problem& p, const FusedArgs& a, const Fused& F, size_t first, size_t stride) {
#pragma clang fp contract(fast)
    const double alpha = F.s.alpha, inv_alpha = F.s.inv_alpha, inv_beta = F.s.inv_beta;
    const size_t n = (size_t)p.nobs, m = 2 * n;
    const size_t pbase = 6 * (size_t)p.ncam;
    double acc = 0.0;
    for (size_t e = first; e < m; e += stride) {
        const size_t i = e >> 1;
        const int row = (int)(e & 1);
        const int c = p.cam_idx[i];
        const int q = p.pt_idx[i];
        double y = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int col = c * 6 + k;
            const double vn = rounded_product(inv_alpha, F.vcam[col]);
            const double vk = a.d[col] * vn;
            y += a.Jc[(row * 6 + k) * n + i] * vk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t col = pbase + 3 * (size_t)q + k;
            const double vn = rounded_product(inv_alpha, a.v[col]);
            const double vk = a.d[col] * vn;
            y += a.Jp[(row * 3 + k) * n + i] * vk;
        }
        const double un = rounded_product(inv_beta, a.u[e]);
        double v = 1.0 * y;
        v += (-alpha) * un;
        a.u[e] = v;
        acc += v * v;
    }
    return acc;
}

__device__ void k2_cam_block(const df3d_ba_problem& p, const FusedArgs& a, double inv_beta, int c, int chunk, double* lds4) {
#pragma clang fp contract(fast)
    const size_t n = (size_t)p.nobs;
    const int lo = p.cam_start[c], hi = p.cam_start[c + 1];
    const int per = (hi - lo + a.nchunk - 1) / a.nchunk;
    const int s0 = lo + chunk * per;
    const int s1 = min(s0 + per, hi);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int s = s0 + (int)threadIdx.x; s < s1; s += 256) {
        const int i = p.cam_perm[s];
        const double2 uu = *reinterpret_cast<const double2*>(a.u + 2 * (size_t)i);
        const double u0 = rounded_product(inv_beta, uu.x), u1 = rounded_product(inv_beta, uu.y);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double j0 = a.Jc[(0 * 6 + k) * n + i], j1 = a.Jc[(1 * 6 + k) * n + i];
            acc[k] += (j0 * u0 + j1 * u1);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double tot = block_reduce_256(acc[k], lds4);
        if (threadIdx.x == 0) a.cam_partial[((size_t)c * a.nchunk + chunk) * 6 + k] = tot;
    }
}

// the point entries k >= 6 ncam of v <- D J^T u - beta v for this thread's k = first, first + stride, ...; returns its sum of squares
// (the camera entries are finished by k3, which also re-forms workgroup 0's partial sum with them in place)
__device__ double k2_points(const df3d_ba_problem& p, const FusedArgs& a, const Fused& F, size_t first, size_t stride) {
#pragma clang fp contract(fast)
    const double beta = F.s.beta, inv_beta = F.s.inv_beta, inv_alpha = F.s.inv_alpha;
    const size_t n = (size_t)p.nobs;
    const size_t ncols = 6 * (size_t)p.ncam + 3 * (size_t)p.npts;
    double acc2 = 0.0;
    for (size_t k = first; k < ncols; k += stride) {
        if (k < 6 * (size_t)p.ncam) continue;
        const size_t kk = k - 6 * (size_t)p.ncam;
        const int q = (int)(kk / 3), col = (int)(kk % 3);
        double acc = 0.0;
        for (int i = p.pt_start[q]; i < p.pt_start[q + 1]; ++i) {
            const double j0 = a.Jp[(0 * 3 + col) * n + i], j1 = a.Jp[(1 * 3 + col) * n + i];
            const double u0 = rounded_product(inv_beta, a.u[2 * (size_t)i]), u1 = rounded_product(inv_beta, a.u[2 * (size_t)i + 1]);
            acc += j0 * u0 + j1 * u1;
        }
        const double w = rounded_product(a.d[k], acc);
        const double vn = rounded_product(inv_alpha, a.v[k]);
        double v = 1.0 * w;
        v += (-beta) * vn;
        a.v[k] = v;
        acc2 += v * v;
    }
    return acc2;
}

// kb: [step C of the previous iteration: a stopped run ends here] ; step A (beta) ; camera workgroups: partial sums of J_c^T u ; the others:
// the point entries of v, partial |v|^2 in round 3's grouping.  flush: the scalar half only.
__global__ __launch_bounds__(256) void fused_kb(df3d_ba_problem p, FusedArgs a, int flush) {
    __shared__ double lds4[4];
    __shared__ Fused F;
    load_state(F, a.st + FUSED_DOUBLES);
    double* const out = a.st;
    if (F.s.istop) {
        store_state(out, F);
        return;
    }
    if (F.pending_c) {
        const double ss = sum_partials(a.red3, a.g3, lds4);
        if (threadIdx.x == 0) {
            apply_step_c(F.s, ss);
            F.pending_c = 0;
        }
        __syncthreads();
        if (F.s.istop) {
            store_state(out, F);
            return;
        }
    }
    if (flush) {
        store_state(out, F);
        return;
    }
    {
        const double ss = sum_partials(a.red1, a.g1, lds4);
        if (threadIdx.x == 0) {
            apply_step_a(F.s, ss);
            F.pending_b = 1;
        }
        __syncthreads();
    }
    const int ncamblk = p.ncam * a.nchunk;
    if ((int)blockIdx.x < ncamblk) {
        k2_cam_block(p, a, F.s.inv_beta, blockIdx.x / a.nchunk, blockIdx.x % a.nchunk, lds4);
    } else if (F.s.beta_pos) {
        const int pb = (int)blockIdx.x - ncamblk;   // workgroup pb of a.g2p: elements pb * 256 + t, + g2p * 256, ... (round 3's bidiag grid)
        const double acc = k2_points(p, a, F, (size_t)pb * 256 + threadIdx.x, (size_t)a.g2p * 256);
        const double tot = block_reduce_256(acc, lds4);
        if (threadIdx.x == 0) a.red2[pb] = tot;
    }
    store_state(out, F);
}

__device__ double k3_update(const df3d_ba_problem& p, const FusedArgs& a, const Fused& F, size_t first, size_t stride) {
#pragma clang fp contract(fast)
    const double c1 = F.s.c1, c2 = F.s.c2, c3 = F.s.c3, inv_alpha = F.s.inv_alpha;
    const size_t ncc = 6 * (size_t)p.ncam;
    const size_t n = ncc + 3 * (size_t)p.npts;
    double acc = 0.0;
    for (size_t i = first; i < n; i += stride) {
        const double vraw = i < ncc ? F.vcam[i] : a.v[i];
        const double vn = rounded_product(inv_alpha, vraw);
        const double hi = a.h[i];
        const double hb = hi - c1 * a.hbar[i];
        const double xi = a.x[i] + c2 * hb;
        a.hbar[i] = hb;
        a.x[i] = xi;
        a.h[i] = vn - c3 * hi;
        acc += xi * xi;
    }
    return acc;
}

// workgroup 0's partial sum of |v|^2 as round 3's bidiag kernel formed it: elements t, t + g * 256, ... with the camera entries in place
__device__ double k3_partial0(const df3d_ba_problem& p, const FusedArgs& a, const double* vnew, size_t stride) {
#pragma clang fp contract(fast)
    const size_t ncc = 6 * (size_t)p.ncam;
    const size_t n = ncc + 3 * (size_t)p.npts;
    double acc = 0.0;
    for (size_t k = threadIdx.x; k < n; k += stride) {
        const double v = k < ncc ? vnew[k] : a.v[k];
        acc += v * v;
    }
    return acc;
}

__device__ double k3_cam_entry(const FusedArgs& a, const Fused& F, int k) {
#pragma clang fp contract(fast)
    const int c = k / 6, col = k % 6;
    double acc = 0.0;
    for (int ch = 0; ch < a.nchunk; ++ch) acc += a.cam_partial[((size_t)c * a.nchunk + ch) * 6 + col];
    const double w = rounded_product(a.d[k], acc);
    const double vn = rounded_product(F.s.inv_alpha, F.vcam[k]);
    double v = 1.0 * w;
    v += (-F.s.beta) * vn;
    return v;
}

// ka: [the camera entries of v from kb's partial sums ; step B (alpha, rotations) ; hbar, x, h ; partial |x|^2] of the previous iteration, then
// u <- A (D v) - alpha u ; partial |u|^2 of this one
__global__ __launch_bounds__(256) void fused_ka(df3d_ba_problem p, FusedArgs a) {
    __shared__ double lds4[4];
    __shared__ double vnew[48];
    __shared__ Fused F;
    load_state(F, a.st);
    double* const out = a.st + FUSED_DOUBLES;
    if (F.s.istop) {
        store_state(out, F);
        return;
    }
    if (F.pending_b) {
        const int ncc = 6 * p.ncam;
        if ((int)threadIdx.x < ncc) vnew[threadIdx.x] = F.s.beta_pos ? k3_cam_entry(a, F, threadIdx.x) : F.vcam[threadIdx.x];
        __syncthreads();
        double ss = 0.0;
        if (F.s.beta_pos) {   // (step B ignores the sum otherwise, as round 3's skipped kernels left it stale)
            const double p0 = block_reduce_256(k3_partial0(p, a, vnew, (size_t)a.g2p * 256), lds4);
            double acc = 0.0;
            for (int i = threadIdx.x; i < a.g2p; i += 256) acc += i == 0 ? p0 : a.red2[i];
            ss = block_reduce_256(acc, lds4);
        }
        if (threadIdx.x == 0) {
            apply_step_b(F.s, ss);
            F.pending_c = 1;
            F.pending_b = 0;
        }
        __syncthreads();
        if ((int)threadIdx.x < ncc) F.vcam[threadIdx.x] = vnew[threadIdx.x];
        __syncthreads();
        if ((int)blockIdx.x < a.g3) {
            const double acc = k3_update(p, a, F, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)a.g3 * 256);
            const double tot = block_reduce_256(acc, lds4);
            if (threadIdx.x == 0) a.red3[blockIdx.x] = tot;
        }
    }
    if ((int)blockIdx.x < a.g1) {
        const double acc = k1_rows(p, a, F, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)a.g1 * 256);
        const double tot = block_reduce_256(acc, lds4);
        if (threadIdx.x == 0) a.red1[blockIdx.x] = tot;
    }
    store_state(out, F);
}

// ---- round 5: the whole LSMR run as ONE persistent kernel ------------------------------------------------------------------------------
// The two kernels of an iteration become the two phases of a loop inside one launch; the kernel boundary -- which sat exactly where a
// grid-wide sum is needed -- becomes a grid-wide barrier.  Everything else is unchanged ON PURPOSE: the same leaf functions run over the
// same VIRTUAL workgroups (round 3's grids g1 / g2p / g3 and the ncam x nchunk camera blocks are part of the arithmetic: they fix how the
// sums of squares are grouped), workgroup w of the G resident ones taking virtual blocks w, w + G, ...; every workgroup keeps the
// state in its LDS and runs the scalar steps itself on the same partial sums, so no state is handed from workgroup to workgroup.  Same
// products, same sums, same order: the iterates, alpha, beta, |x| and the iteration counts are bit for bit those of the two-kernel form
// (tests/test_gpu_ba.py), at 2 barriers instead of 2 launches per iteration and ONE host synchronisation per solve instead of one per
// 16 iterations.
//
// The barrier (guide: cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "barrier-counter"): every wave drains its stores,
// __syncthreads, ONE lane: agent-scope release (writes back this XCD's L2) -> s_waitcnt vmcnt(0) (in asm: the compiler may drop its own)
// -> relaxed agent-scope add on a monotonic counter -> relaxed polling with s_sleep until the counter reaches G x epoch -> ONE agent-scope
// acquire (drops this CU's stale L1 lines) -> __syncthreads.  The counter is zeroed by a memset in front of every launch; the spin is bounded
// (a workgroup that never becomes resident -- the device full of somebody else's persistent kernels -- ends the run with istop = -1, and
// the host falls back to the two-kernel form); G <= 128 workgroups of 256 threads and ~2 KB of LDS are resident on an idle device by
// construction.
constexpr unsigned BAR_SPIN_LIMIT = 1u << 21;   // x (poll + s_sleep) ~ 1 us each: seconds

__device__ __forceinline__ bool grid_barrier(unsigned* __restrict__ bar, unsigned target, int* ok_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (gridDim.x == 1) return true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > BAR_SPIN_LIMIT || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everybody else leaves with us
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}

__global__ __launch_bounds__(256) void fused_persistent(df3d_ba_problem p, FusedArgs a, unsigned* bar, int maxiter) {
    __shared__ double lds4[4];
    __shared__ double vnew[48];
    __shared__ Fused F;
    __shared__ int bar_ok;
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    unsigned epoch = 0;
    bool failed = false;
    load_state(F, a.st);
    for (int it = 0; it <= maxiter; ++it) {
        const bool flush = it == maxiter;   // maxiter reached: the last iteration's steps B and C are still to be taken (launch_fused_flush)
        if (F.s.istop) break;
        // ---- phase A (fused_ka)
        if (F.pending_b) {
            const int ncc = 6 * p.ncam;
            if ((int)threadIdx.x < ncc) vnew[threadIdx.x] = F.s.beta_pos ? k3_cam_entry(a, F, threadIdx.x) : F.vcam[threadIdx.x];
            __syncthreads();
            double ss = 0.0;
            if (F.s.beta_pos) {
                const double p0 = block_reduce_256(k3_partial0(p, a, vnew, (size_t)a.g2p * 256), lds4);
                double acc = 0.0;
                for (int i = threadIdx.x; i < a.g2p; i += 256) acc += i == 0 ? p0 : a.red2[i];
                ss = block_reduce_256(acc, lds4);
            }
            if (threadIdx.x == 0) {
                apply_step_b(F.s, ss);
                F.pending_c = 1;
                F.pending_b = 0;
            }
            __syncthreads();
            if ((int)threadIdx.x < ncc) F.vcam[threadIdx.x] = vnew[threadIdx.x];
            __syncthreads();
            for (int vb = wg; vb < a.g3; vb += G) {
                const double acc = k3_update(p, a, F, (size_t)vb * 256 + threadIdx.x, (size_t)a.g3 * 256);
                const double tot = block_reduce_256(acc, lds4);
                if (threadIdx.x == 0) a.red3[vb] = tot;
            }
        }
        for (int vb = wg; vb < a.g1; vb += G) {
            const double acc = k1_rows(p, a, F, (size_t)vb * 256 + threadIdx.x, (size_t)a.g1 * 256);
            const double tot = block_reduce_256(acc, lds4);
            if (threadIdx.x == 0) a.red1[vb] = tot;
        }
        ++epoch;
        if (!grid_barrier(bar, epoch * (unsigned)G, &bar_ok)) {
            failed = true;
            break;
        }
        // ---- phase B (fused_kb)
        if (F.pending_c) {
            const double ss = sum_partials(a.red3, a.g3, lds4);
            if (threadIdx.x == 0) {
                apply_step_c(F.s, ss);
                F.pending_c = 0;
            }
            __syncthreads();
            if (F.s.istop) break;
        }
        if (flush) break;
        {
            const double ss = sum_partials(a.red1, a.g1, lds4);
            if (threadIdx.x == 0) {
                apply_step_a(F.s, ss);
                F.pending_b = 1;
            }
            __syncthreads();
        }
        const int ncamblk = p.ncam * a.nchunk;
        for (int vb = wg; vb < ncamblk + a.g2p; vb += G) {
            if (vb < ncamblk) {
                k2_cam_block(p, a, F.s.inv_beta, vb / a.nchunk, vb % a.nchunk, lds4);
            } else if (F.s.beta_pos) {
                const int pb = vb - ncamblk;
                const double acc = k2_points(p, a, F, (size_t)pb * 256 + threadIdx.x, (size_t)a.g2p * 256);
                const double tot = block_reduce_256(acc, lds4);
                if (threadIdx.x == 0) a.red2[pb] = tot;
            }
        }
        ++epoch;
        if (!grid_barrier(bar, epoch * (unsigned)G, &bar_ok)) {
            failed = true;
            break;
        }
    }
    __syncthreads();
    if (failed && threadIdx.x == 0) F.s.istop = -1;
    __syncthreads();
    store_state(a.st, F);
}

// ---- round 5: the DATA-LOCAL run ------------------------------------------------------------------------------------------------------
// What the persistent form above shows (profiles/r05_ba_timings.txt): replacing the launches by barriers is not enough -- an iteration is
// two latency chains (index loads -> Jacobian loads -> gathers through L2, block reductions, partial sums re-read by every workgroup),
// and a grid barrier with its release / acquire fences costs more than the launch boundary it replaces.  This form removes the chains:
//
//   * the observations are cut into G contiguous ranges at POINT boundaries (<= 1 024 observations, hence <= 512 points each); a
//     workgroup of 512 threads owns one range for the whole run and keeps its slice of the Jacobian (18 doubles per observation), of u
//     (2) and of the point entries of v, h, hbar, x (12 per point) in REGISTERS, read once; J v and J^T u for the point block touch only
//     the workgroup's own data (a point's observations are consecutive), through LDS;
//   * what is global is small: |u|^2 (+ the previous iteration's |x|^2) after the first half, the 6 x ncam camera sums of J^T u and
//     |v|^2 after the second -- two ALL-REDUCES of 2 and 43 doubles per iteration, done with the guide's tagged granules (cdna_hip_programming.md
//     Guideline 16 R2: 8-byte {epoch, 32-bit value} agent-scope atomic stores, every workgroup sweeps all G x N x 2 granules until the
//     tags match, no counter, no fence) and summed by every workgroup in the same fixed order, so all of them hold the same scalars and
//     run the scalar steps themselves (the step functions are the ones of the other forms);
//   * the camera entries (6 x ncam of every vector) are kept by every workgroup redundantly in LDS.
// The sums are grouped differently from the launch-based forms (per workgroup range instead of grid-strided), so the iterates differ
// from theirs in the last bits; every reduction has a fixed order, so a run reproduces itself bit for bit.  Parity: the oracle's
// iteration counts, stop reasons and solution (tests/test_gpu_ba.py).
constexpr int LT = 512;             // threads per workgroup: 8 waves per CU = 256 registers per thread (1 024 threads x 1 observation: 128
                                    // registers, 49-65 of them spilled beside the inlined scalar steps)
#ifndef DF3D_LSMR_LK
#define DF3D_LSMR_LK 2
#endif
constexpr int LK = DF3D_LSMR_LK;    // observations per thread
constexpr int LOBS = LT * LK;       // observations per workgroup
constexpr int LNAR = 43;            // doubles of the larger all-reduce: |v_points|^2 + 6 x 7 camera sums (MAX_CAM = 8 -> 49 would be needed: checked by the host)
constexpr int LMAXG = 128;          // workgroups (<= 130 000 observations: every window of 1 000 frames)
constexpr int LRED = 16;            // partial sums per (camera, column) of the camera reduction (8 cameras x 3 columns x 16 <= LT threads)
constexpr unsigned LSWEEP_LIMIT = 1u << 20;

struct LocalLds {
    double contrib[3][LOBS];        // J^T u contributions of the observations, one column triple at a time
    double dv[3 * LT];              // d * v of the workgroup's points
    int camlist[LOBS];              // local observation indices grouped by camera (ascending inside a camera)
    double part[8 * 3 * LRED];      // partial camera sums
    double gathered[LMAXG * 49];    // an all-reduce's values of every workgroup
    double own[49], res[49];
    double dcam[48], vcam[48], hcam[48], hbarcam[48], xcam[48], dvcam[48];
    double wred[16];
    int camstart[9];
    int flag;
    State S;
    int pending_c;
    unsigned long long tacc[10];    // development (DF3D_LSMR_DEBUG): cycles per segment, summed over the iterations (workgroup 0, thread 0)
};

__device__ __forceinline__ double block_reduce_wg(double v, double* wred) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < LT / 64; ++w) t += wred[w];
    return t;
}

typedef __attribute__((address_space(1))) unsigned long long gu64;   // every shared word: a GLOBAL agent-scope access, never flat (Guideline 16)
// all-reduce of L.own[0 .. n) over the G workgroups -> L.res[0 .. n), every workgroup summing everybody's values in workgroup order.
// Two transports (cdna_hip_programming.md Guideline 16):
//   n <= 2 (R2, the data is the flag): 2 n tagged granules per workgroup, every workgroup sweeps all of them until the tags match;
//   larger (R1): the payload as 8-byte write-through (sc1) stores from ONE wave, that wave's s_waitcnt vmcnt(0), then ONE tagged flag per
//     workgroup; readers poll the G flags, then read the payload once with sc1 loads.  With 105 workgroups the granule form of the
//     43-double reduction had every workgroup re-reading 72 KB per sweep: 12.6 us per call; flags: G x 8 bytes per sweep.
// Returns false on a timeout (a workgroup that never became resident).
constexpr int LDBG_CALLS_FWD = 64;
// The granules / flags are double-buffered by epoch parity (LAR1_STRIDE / LAR2_STRIDE apart): a workgroup can only publish epoch e + 2 after every
// workgroup published e + 1, i.e. finished sweeping e -- two all-reduces over the same granules may then follow each other directly (the
// beta == 0 branch skips the second one in between).
constexpr size_t LAR1_STRIDE = (size_t)LMAXG * 4, LAR2_STRIDE = (size_t)LMAXG * (16 + 49);
__device__ bool all_reduce(LocalLds& L, unsigned long long* gran_, int n, unsigned epoch, double* dbg = nullptr, int call = 0) {
    constexpr int LDBG_CALLS = LDBG_CALLS_FWD;
    gu64* const gran = (gu64*)gran_ + (epoch & 1u) * (n <= 2 ? LAR1_STRIDE : LAR2_STRIDE);
    const int G = (int)gridDim.x, wg = (int)blockIdx.x, tid = (int)threadIdx.x;
    __syncthreads();   // L.own is complete
    if (G == 1) {
        if (tid < n) L.res[tid] = L.own[tid];
        __syncthreads();
        return true;
    }
    unsigned spins = 0;
    if (n <= 2) {
        if (tid < 2 * n) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(L.own[tid >> 1]);
            const unsigned half = (unsigned)(tid & 1 ? bits >> 32 : bits & 0xffffffffull);
            __hip_atomic_store(gran + (size_t)wg * 2 * n + tid, ((unsigned long long)epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int total = G * 2 * n;
        unsigned* const words = reinterpret_cast<unsigned*>(L.gathered);
        for (;;) {
            int ok = 1;
            for (int idx = tid; idx < total; idx += LT) {
                const unsigned long long g = __hip_atomic_load(gran + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok &= (unsigned)(g >> 32) == epoch;
                words[idx] = (unsigned)g;
            }
            if (__syncthreads_and(ok)) break;
            if (++spins > LSWEEP_LIMIT) return false;   // (uniform: every thread counts the same sweeps)
            __builtin_amdgcn_s_sleep(1);
        }
    } else {
        // layout: [G flags, one per 128-byte line][G x n payload doubles]
        gu64* const flags = gran;
        gu64* const payload = gran + (size_t)gridDim.x * 16;
        if (tid < 64) {   // wave 0 alone publishes (n <= 49 < 64 lanes), so its own vmcnt(0) orders payload before flag
            if (tid < n) __hip_atomic_store(payload + (size_t)wg * n + tid, (unsigned long long)__double_as_longlong(L.own[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_store(flags + (size_t)wg * 16, (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (;;) {
            int ok = 1;
            for (int w = tid; w < G; w += LT) ok &= (unsigned)__hip_atomic_load(flags + (size_t)w * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
            if (__syncthreads_and(ok)) break;
            if (++spins > LSWEEP_LIMIT) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        unsigned long long* const q = reinterpret_cast<unsigned long long*>(L.gathered);
        for (int idx = tid; idx < G * n; idx += LT) q[idx] = __hip_atomic_load(payload + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    if (tid < n) {
        double t = 0.0;
        for (int w = 0; w < G; ++w) t += L.gathered[w * n + tid];
        L.res[tid] = t;
    }
    __syncthreads();
    if (dbg != nullptr && call < LDBG_CALLS && tid < n) {
        double* const row = dbg + ((size_t)call * G + wg) * 98;
        row[tid] = L.own[tid];
        row[49 + tid] = L.res[tid];
    }
    return true;
}

// the scalar steps on the state in LDS
__device__ __forceinline__ void local_step_a(State* S, double ss) { apply_step_a(*S, ss); }
__device__ __forceinline__ void local_step_b(State* S, double ss) { apply_step_b(*S, ss); }
__device__ __forceinline__ void local_step_c(State* S, double ss) { apply_step_c(*S, ss); }
__device__ __forceinline__ void local_init_state(State* S, double alpha, double beta, double normb, double damp, double atol, double btol, double ctol, int maxiter) {
    S->alpha = alpha;
    S->beta = beta;
    S->rho = S->rhobar = S->cbar = 1;
    S->sbar = 0;
    S->zeta = 0;
    S->zetabar = alpha * beta;
    S->alphabar = alpha;
    S->betadd = beta;
    S->betad = 0;
    S->rhodold = 1;
    S->tautildeold = S->thetatilde = S->dd = 0;
    S->normA2 = alpha * alpha;
    S->maxrbar = 0;
    S->minrbar = 1e100;
    S->normA = sqrt(S->normA2);
    S->condA = 1;
    S->normx = 0;
    S->normr = beta;
    S->normar = alpha * beta;
    S->normb = normb;
    S->damp = damp;
    S->atol = atol;
    S->btol = btol;
    S->ctol = ctol;
    S->rhobarold = S->zetaold = S->thetabar = S->rhotemp = S->chat = S->shat = S->c = S->sn = 0;
    S->c1 = S->c2 = S->c3 = 0;
    S->inv_beta = S->inv_alpha = 1.0;
    S->itn = 0;
    S->istop = 0;
    S->maxiter = maxiter;
    S->beta_pos = 1;
}

struct LocalArgs {
    const double *Jc, *Jp, *d, *b;
    double* x;                         // [n] out
    const int* wg_obs;                 // [G + 1] observation ranges, cut at point boundaries
    unsigned long long *gr1, *gr2;     // granules of the two all-reduces: G x 4, G x 2 (1 + 6 ncam); zeroed in front of the launch
    double* state_out;                 // the final State (workgroup 0)
    double damp, atol, btol, ctol;
    const double* damp_dev;            // when not null: the damping is read HERE (a scalar an earlier kernel of the stream left on the device)
    int maxiter;
    double* dbg;                       // development (DF3D_LSMR_DEBUG=1): per all-reduce call [call][workgroup][own 49 | res 49], or nullptr
};
constexpr int LDBG_CALLS = 64;

__global__ __launch_bounds__(LT, 2) void lsmr_local_kernel(df3d_ba_problem p, LocalArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    LocalLds& L = *reinterpret_cast<LocalLds*>(lds_raw);
    const int tid = (int)threadIdx.x, wg = (int)blockIdx.x;
    const int ncc = 6 * p.ncam, nar2 = 1 + ncc;
    const size_t nobs = (size_t)p.nobs;
    if (a.wg_obs[gridDim.x] != p.nobs) {   // the partition does not cover the problem (see lsmr_local_partition_kernel): every workgroup leaves, before any all-reduce
        if (wg == 0 && tid == 0) {
            State bad{};
            bad.istop = -2;
            *reinterpret_cast<State*>(a.state_out) = bad;
        }
        return;
    }
    const int o0 = a.wg_obs[wg], o1 = a.wg_obs[wg + 1], nob = o1 - o0;
    const int q0 = nob > 0 ? p.pt_idx[o0] : 0, npt = nob > 0 ? p.pt_idx[o1 - 1] + 1 - q0 : 0;

    // ---- this thread's observations (local index tid + k LT) and point (local index tid)
    double Jc[LK][12], Jp[LK][6], u[LK][2];
    int cam[LK], ql[LK];
    bool have[LK];
#pragma unroll
    for (int k = 0; k < LK; ++k) {
        const int jl = tid + k * LT;
        have[k] = jl < nob;
        const size_t i = have[k] ? (size_t)(o0 + jl) : 0;
        cam[k] = p.cam_idx[i];
        ql[k] = p.pt_idx[i] - q0;
#pragma unroll
        for (int e = 0; e < 12; ++e) Jc[k][e] = have[k] ? a.Jc[(size_t)e * nobs + i] : 0.0;
#pragma unroll
        for (int e = 0; e < 6; ++e) Jp[k][e] = have[k] ? a.Jp[(size_t)e * nobs + i] : 0.0;
        u[k][0] = have[k] ? a.b[2 * i] : 0.0;
        u[k][1] = have[k] ? a.b[2 * i + 1] : 0.0;
    }
    const bool owner = tid < npt;
    const size_t col0 = (size_t)ncc + 3 * (size_t)(q0 + (owner ? tid : 0));
    double dp[3], v[3] = {0, 0, 0}, h[3], hbar[3] = {0, 0, 0}, x[3] = {0, 0, 0};
#pragma unroll
    for (int e = 0; e < 3; ++e) dp[e] = owner ? a.d[col0 + e] : 0.0;
    const int ps = owner ? p.pt_start[q0 + tid] - o0 : 0, pe = owner ? p.pt_start[q0 + tid + 1] - o0 : 0;
    if (tid < 48) {
        L.dcam[tid] = tid < ncc ? a.d[tid] : 0.0;
        L.vcam[tid] = L.hcam[tid] = L.hbarcam[tid] = L.xcam[tid] = L.dvcam[tid] = 0.0;
    }
    // the workgroup's observations grouped by camera: its part of camera c's (ascending) list in cam_perm is contiguous
    if (tid <= p.ncam) {
        int pos = 0;
        for (int c = 0; c < tid; ++c) {   // (entries of the cameras in front of camera tid that fall into [o0, o1))
            int lo = p.cam_start[c], hi = p.cam_start[c + 1];
            auto lower = [&](int key) {
                int l = lo, r = hi;
                while (l < r) {
                    const int mid = (l + r) >> 1;
                    if (p.cam_perm[mid] < key) l = mid + 1;
                    else r = mid;
                }
                return l;
            };
            pos += lower(o1) - lower(o0);
        }
        L.camstart[tid] = pos;
    }
    if (tid == 0) L.pending_c = 0;
    __syncthreads();
    for (int c = 0; c < p.ncam; ++c) {
        const int cnt = L.camstart[c + 1] - L.camstart[c];
        if (cnt > 0) {
            int l = p.cam_start[c], r = p.cam_start[c + 1];   // first entry of camera c >= o0
            while (l < r) {
                const int mid = (l + r) >> 1;
                if (p.cam_perm[mid] < o0) l = mid + 1;
                else r = mid;
            }
            for (int j = tid; j < cnt; j += LT) L.camlist[L.camstart[c] + j] = p.cam_perm[l + j] - o0;
        }
    }
    __syncthreads();

    unsigned e1 = 0, e2 = 0;
    bool failed = false;
    unsigned long long tlast = 0;
    const bool timing = a.dbg != nullptr && tid == 0;
    if (tid < 10) L.tacc[tid] = 0;
    auto stamp = [&](int k) {   // (development builds of the run only: a.dbg set)
        if (timing) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            L.tacc[k] += now - tlast;
            tlast = now;
        }
    };

    // u <- u * s (this thread's rows)
    auto scale_u = [&](double sc) {
#pragma unroll
        for (int k = 0; k < LK; ++k) {
            u[k][0] *= sc;
            u[k][1] *= sc;
        }
    };
    // second half of a bidiagonalisation step: v <- D J^T u - beta v for the workgroup's points, the camera sums and |v_points|^2 into
    // L.own, all-reduce, the camera entries of v and |v|^2 (returned) in every workgroup
    auto half_b = [&](double beta, double& ss_v) -> bool {
        double accv = 0.0;
        for (int pass = 0; pass < 3; ++pass) {   // 0: point columns; 1, 2: camera columns 0..2, 3..5
#pragma unroll
            for (int k = 0; k < LK; ++k) {
                const int jl = tid + k * LT;
                if (have[k]) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const double j0 = pass == 0 ? Jp[k][e] : Jc[k][3 * (pass - 1) + e];
                        const double j1 = pass == 0 ? Jp[k][3 + e] : Jc[k][6 + 3 * (pass - 1) + e];
                        L.contrib[e][jl] = j0 * u[k][0] + j1 * u[k][1];
                    }
                }
            }
            __syncthreads();
            if (pass == 0) {
                if (owner) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        double w = 0.0;
                        for (int j = ps; j < pe; ++j) w += L.contrib[e][j];
                        const double vn = dp[e] * w - beta * v[e];
                        v[e] = vn;
                        accv += vn * vn;
                    }
                }
            } else {
                if (tid < p.ncam * 3 * LRED) {
                    const int c = tid / (3 * LRED), e = (tid / LRED) % 3, r = tid % LRED;
                    double t = 0.0;
                    for (int j = L.camstart[c] + r; j < L.camstart[c + 1]; j += LRED) t += L.contrib[e][L.camlist[j]];
                    L.part[tid] = t;
                }
                __syncthreads();
                if (tid < p.ncam * 3) {
                    const int c = tid / 3, e = tid % 3;
                    double t = 0.0;
                    for (int r = 0; r < LRED; ++r) t += L.part[(c * 3 + e) * LRED + r];
                    L.own[1 + c * 6 + 3 * (pass - 1) + e] = t;
                }
            }
            __syncthreads();
        }
        const double tot = block_reduce_wg(accv, L.wred);
        if (tid == 0) L.own[0] = tot;
        stamp(3);
        if (!all_reduce(L, a.gr2, nar2, e2 + 1, a.dbg, (int)(e1 + e2))) return false;
        ++e2;
        stamp(4);
        if (tid < ncc) L.vcam[tid] = L.dcam[tid] * L.res[1 + tid] - beta * L.vcam[tid];
        __syncthreads();
        double t = L.res[0];
        for (int j = 0; j < ncc; ++j) t += L.vcam[j] * L.vcam[j];   // (every thread: the same sum in the same order)
        ss_v = t;
        __syncthreads();   // every thread has read L.vcam: the caller rescales it next (without this barrier a fast wave's scaling reached a slow
                           // wave's sum: 1e-8 off in 18 % of the runs at 105 workgroups -- found with DF3D_LSMR_DEBUG and tests/perf/lsmr_stress.py)
        return true;
    };
    auto scale_v = [&](double sc) {
#pragma unroll
        for (int e = 0; e < 3; ++e) v[e] *= sc;
        if (tid < ncc) L.vcam[tid] *= sc;
        __syncthreads();
    };

    // ---- u = b, beta = |b|; v = D J^T u, alpha = |v| (scipy's lsmr, restated in oracle/trf_lsmr.py:lsmr)
    double normb, alpha = 0.0, beta;
    {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < LK; ++k) acc += u[k][0] * u[k][0] + u[k][1] * u[k][1];
        const double tot = block_reduce_wg(acc, L.wred);
        if (tid == 0) {
            L.own[0] = tot;
            L.own[1] = 0.0;
        }
        if (!all_reduce(L, a.gr1, 2, e1 + 1, a.dbg, (int)(e1 + e2))) failed = true;
        ++e1;
        normb = beta = failed ? 0.0 : sqrt(L.res[0]);
    }
    if (!failed && beta > 0) {
        scale_u(1.0 / beta);
        double ss;
        if (!half_b(0.0, ss)) failed = true;
        else {
            alpha = sqrt(ss);
            if (alpha > 0) scale_v(1.0 / alpha);
        }
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) h[e] = v[e];
    if (tid < 48) L.hcam[tid] = L.vcam[tid];
    if (tid == 0) local_init_state(&L.S, alpha, beta, normb, a.damp_dev ? *a.damp_dev : a.damp, a.atol, a.btol, a.ctol, a.maxiter);
    __syncthreads();
    const bool trivial = L.S.normar == 0 || L.S.normb == 0;

    // ---- iterations
    double ssx_points = 0.0;
    if (timing) tlast = __builtin_amdgcn_s_memtime();
    while (!failed && !trivial) {
        const bool last = L.S.itn >= a.maxiter;   // maxiter reached: only the pending stopping tests are left
        double acc = 0.0;
        if (!last) {
            // u <- J (D v) - alpha u
            if (owner) {
#pragma unroll
                for (int e = 0; e < 3; ++e) L.dv[3 * tid + e] = dp[e] * v[e];
            }
            if (tid < ncc) L.dvcam[tid] = L.dcam[tid] * L.vcam[tid];
            __syncthreads();
            const double al = L.S.alpha;
#pragma unroll
            for (int k = 0; k < LK; ++k) {
                if (have[k]) {
#pragma unroll
                    for (int row = 0; row < 2; ++row) {
                        double y = 0.0;
#pragma unroll
                        for (int e = 0; e < 6; ++e) y += Jc[k][row * 6 + e] * L.dvcam[cam[k] * 6 + e];
#pragma unroll
                        for (int e = 0; e < 3; ++e) y += Jp[k][row * 3 + e] * L.dv[3 * ql[k] + e];
                        const double un = y - al * u[k][row];
                        u[k][row] = un;
                        acc += un * un;
                    }
                }
            }
        }
        const double tot = block_reduce_wg(acc, L.wred);
        if (tid == 0) {
            L.own[0] = tot;
            L.own[1] = ssx_points;
        }
        stamp(0);
        if (!all_reduce(L, a.gr1, 2, e1 + 1, a.dbg, (int)(e1 + e2))) {
            failed = true;
            break;
        }
        ++e1;
        stamp(1);
        if (tid == 0) {
            if (L.pending_c) {
                double ssx = L.res[1];
                for (int j = 0; j < ncc; ++j) ssx += L.xcam[j] * L.xcam[j];
                local_step_c(&L.S, ssx);
                L.pending_c = 0;
            }
            if (!L.S.istop && !last) local_step_a(&L.S, L.res[0]);
        }
        __syncthreads();
        stamp(2);
        if (L.S.istop || last) break;
        if (L.S.beta_pos) {
            scale_u(L.S.inv_beta);
            double ss;
            if (!half_b(L.S.beta, ss)) {
                failed = true;
                break;
            }
            if (tid == 0) local_step_b(&L.S, ss);
            __syncthreads();
            stamp(5);
            scale_v(L.S.inv_alpha);
        } else {
            if (tid == 0) local_step_b(&L.S, 0.0);
            __syncthreads();
        }
        // hbar, x, h
        const double c1 = L.S.c1, c2 = L.S.c2, c3 = L.S.c3;
        double accx = 0.0;
        if (owner) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const double hb = h[e] - c1 * hbar[e];
                x[e] += c2 * hb;
                hbar[e] = hb;
                h[e] = v[e] - c3 * h[e];
                accx += x[e] * x[e];
            }
        }
        if (tid < ncc) {
            const double hb = L.hcam[tid] - c1 * L.hbarcam[tid];
            L.xcam[tid] += c2 * hb;
            L.hbarcam[tid] = hb;
            L.hcam[tid] = L.vcam[tid] - c3 * L.hcam[tid];
        }
        ssx_points = block_reduce_wg(accx, L.wred);
        if (tid == 0) L.pending_c = 1;
        __syncthreads();
        stamp(6);
    }
    if (a.dbg != nullptr && wg == 0 && tid < 10) a.dbg[(size_t)LDBG_CALLS * gridDim.x * 98 + tid] = (double)L.tacc[tid];
    if (owner) {
#pragma unroll
        for (int e = 0; e < 3; ++e) a.x[col0 + e] = x[e];
    }
    if (wg == 0) {
        if (tid < ncc) a.x[tid] = L.xcam[tid];
        if (tid == 0) {
            if (failed) L.S.istop = -1;
            *reinterpret_cast<State*>(a.state_out) = L.S;
        }
    }
}

// observation ranges of the data-local form: greedy, as many whole points as fit into LOBS observations AND into the LT point-owner threads of a
// workgroup; one thread (G <= 64 steps of a binary search).  The kernel's layout assumes what bundle_adjust.py guarantees (every point seen by
// >= 2 cameras, <= 8 observations per point) but the C ABI does not: a problem with single-observation points has more than LT points per LOBS
// observations, and a point of more than 8 observations can make the ranges run out before the observations do.  Either way the partition does
// not cover [0, nobs) in gmax ranges: wg_obs[gmax] != nobs, which lsmr_local_kernel reports as "does not fit" (istop -2: the caller falls back
// to the launch-based forms, which handle any problem).
__global__ void lsmr_local_partition_kernel(const int* __restrict__ pt_start, int npts, int nobs, int gmax, int* __restrict__ wg_obs) {
    if (threadIdx.x || blockIdx.x) return;
    int start = 0, q = 0;
    wg_obs[0] = 0;
    for (int w = 0; w < gmax; ++w) {
        if (start < nobs) {
            int lo = q, hi = npts < q + LT ? npts : q + LT;   // the largest point boundary <= start + LOBS, at most LT points on
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pt_start[mid] <= start + LOBS) lo = mid;
                else hi = mid - 1;
            }
            q = lo;
            start = pt_start[q];
        }
        wg_obs[w + 1] = start;
    }
}

}  // namespace

int local_max_workgroups() { return LMAXG; }
int local_workgroups_for(int nobs) { return (nobs + (LOBS - 8) - 1) / (LOBS - 8); }   // a range holds at least LOBS - 7 observations (a point has <= 8)
size_t local_scratch_bytes() { return (size_t)(LMAXG + 1) * sizeof(int) + 64 + 2 * (LAR1_STRIDE + LAR2_STRIDE) * sizeof(unsigned long long); }

bool local_fits(const df3d_ba_problem& p) { return local_workgroups_for(p.nobs) <= LMAXG && 1 + 6 * p.ncam <= 49; }

// the whole run in one launch; `scratch`: local_scratch_bytes() bytes of device memory (zeroed here); state_out: >= sizeof(State)
int launch_local(const df3d_ba_problem& p, const double* Jc, const double* Jp, const double* d, const double* b, double* x, double damp, double atol,
                 double btol, double ctol, int maxiter, void* scratch, double* state_out, hipStream_t s, const double* damp_dev) {
    const int G = local_workgroups_for(p.nobs);
    if (G > LMAXG || 1 + 6 * p.ncam > 49) return -1;
    if (hipMemsetAsync(scratch, 0, local_scratch_bytes(), s) != hipSuccess) return -2;
    int* const wg_obs = reinterpret_cast<int*>(scratch);
    unsigned long long* const gr1 = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(scratch) + (((size_t)(LMAXG + 1) * sizeof(int) + 63) & ~size_t(63)));
    unsigned long long* const gr2 = gr1 + 2 * LAR1_STRIDE;
    hipLaunchKernelGGL(lsmr_local_partition_kernel, dim3(1), dim3(64), 0, s, p.pt_start, p.npts, p.nobs, G, wg_obs);
    static double* dbg = nullptr;
    static const bool want_dbg = getenv("DF3D_LSMR_DEBUG") != nullptr;
    if (want_dbg && !dbg && hipMalloc(&dbg, ((size_t)LDBG_CALLS * LMAXG * 98 + 16) * sizeof(double)) != hipSuccess) return -4;
    if (want_dbg) (void)hipMemsetAsync(dbg, 0, (size_t)LDBG_CALLS * LMAXG * 98 * sizeof(double), s);
    LocalArgs a{Jc, Jp, d, b, x, wg_obs, gr1, gr2, state_out, damp, atol, btol, ctol, damp_dev, maxiter, want_dbg ? dbg : nullptr};
    {   // the attribute is per DEVICE (a process may drive several GPUs) and the guard must be thread-safe: one bit per device ordinal
        static std::atomic<unsigned long long> attr_set{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return -3;
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_set.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(lsmr_local_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LocalLds)) != hipSuccess) return -3;
            attr_set.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(lsmr_local_kernel, dim3(G), dim3(LT), sizeof(LocalLds), s, p, a);
    if (want_dbg) {   // development: every all-reduce's inputs and outputs of every workgroup, checked on the host
        std::vector<double> hbuf((size_t)LDBG_CALLS * G * 98);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hbuf.data(), dbg, hbuf.size() * sizeof(double), hipMemcpyDeviceToHost);
        double tacc[10];
        (void)hipMemcpy(tacc, dbg + (size_t)LDBG_CALLS * G * 98, sizeof(tacc), hipMemcpyDeviceToHost);
        fprintf(stderr, "lsmr dbg: cycles of workgroup 0 summed over the run (100 MHz s_memtime ticks?): A+reduce %.0f | AR1 %.0f | steps C,A %.0f | scale u + J^T u passes %.0f | AR2 %.0f | vcam + step B %.0f | scale v + update %.0f\n",
                tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5], tacc[6]);
        static std::vector<double> first;
        int bad_sum = 0, bad_agree = 0, bad_repro = 0, calls = 0;
        for (int c = 0; c < LDBG_CALLS; ++c) {
            const double* base = hbuf.data() + (size_t)c * G * 98;
            bool any = false;
            for (int k = 0; k < 49 && !any; ++k) any = base[49 + k] != 0.0;
            if (!any) continue;
            ++calls;
            for (int k = 0; k < 49; ++k) {
                double t = 0.0;
                for (int w = 0; w < G; ++w) t += base[(size_t)w * 98 + k];
                for (int w = 0; w < G; ++w) {
                    bad_agree += base[(size_t)w * 98 + 49 + k] != base[49 + k];
                    bad_sum += base[(size_t)w * 98 + 49 + k] != t;
                }
            }
        }
        if (first.empty()) first = hbuf;
        else {
            for (size_t i = 0; i < hbuf.size() && i < first.size(); ++i)
                if (hbuf[i] != first[i]) {
                    if (bad_repro < 6) fprintf(stderr, "lsmr dbg: first difference from run 0: call %zu wg %zu slot %zu (%s): %.17g vs %.17g\n", i / ((size_t)G * 98), (i / 98) % G, i % 98,
                                               (i % 98) < 49 ? "own" : "res", hbuf[i], first[i]);
                    ++bad_repro;
                }
        }
        fprintf(stderr, "lsmr dbg: G %d, %d all-reduces logged: results != ordered sum of inputs: %d, workgroups disagreeing: %d, values differing from the first run: %d\n", G, calls, bad_sum,
                bad_agree, bad_repro);
    }
    return 0;
}

void launch_fused_persistent(const df3d_ba_problem& p, const FusedArgs& a, unsigned* bar, int maxiter, int grid, hipStream_t s) {
    (void)hipMemsetAsync(bar, 0, 2 * sizeof(unsigned), s);
    hipLaunchKernelGGL(fused_persistent, dim3(grid), dim3(256), 0, s, p, a, bar, maxiter);
}

void launch_fused_iteration(const df3d_ba_problem& p, const FusedArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(fused_ka, dim3(a.g1 > a.g3 ? a.g1 : a.g3), dim3(256), 0, s, p, a);
    hipLaunchKernelGGL(fused_kb, dim3(p.ncam * a.nchunk + a.g2p), dim3(256), 0, s, p, a, 0);
}

void launch_fused_flush(const df3d_ba_problem& p, const FusedArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(fused_ka, dim3(a.g1 > a.g3 ? a.g1 : a.g3), dim3(256), 0, s, p, a);
    hipLaunchKernelGGL(fused_kb, dim3(1), dim3(256), 0, s, p, a, 1);
}

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
