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

}  // namespace

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
