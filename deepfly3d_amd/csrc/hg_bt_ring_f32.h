// fp32 fused pre-activation bottleneck 256 -> 128 -> 128 -> 256 (identity skip) with the weights streamed by LDS-DMA:
// the fp32 sibling of hg_bt_ring.h (same 4-slot ring of 8 KB stage images, same swizzles, same barrier discipline).
//
// Same tile, wave -> tile mapping and MFMA K order as hg_kernels.h:bottleneck_kernel<float, 256, 128, false, UP>, whose
// results it reproduces bit for bit.  As there, the t1 tile is built and consumed in two 64-channel halves (kh), so a tile
// walks the stage sequence
//     kh = 0:  8 x W1 (two 16-float K steps of the half's 64 rows per stage), 36 x W2 (tap, 16-float K slice of the half)
//     kh = 1:  the same for the second half
//     16 x W3 (output half nh, 16-float K slice)                                     = 104 stages, 832 KB per tile.
// What the ring buys in fp32 (an MFMA-bound kernel: 64 cycles per v_mfma_f32_32x32x2_f32): no weight registers (the
// register-staged kernel sits at 254 VGPRs), one barrier per 24-32 MFMAs instead of two, the residual values requested a
// whole double-step before they are used, bn1 coefficients / biases in LDS, and 16-byte LDS stores of the transposed t1 tile.
//
// Waits: every ring wait allows exactly the DMA pieces of the stages requested after the awaited one (2 per stage); plain
// loads issued in between only make that wait conservative (steps are 1 500-2 000 cycles long, so that costs nothing).
// gfx950 retires vector-memory operations in issue order (see hg_bt_ring.h).
#pragma once
#include "hg_bt_ring.h"

#ifndef BRF_ABLM
#define BRF_ABLM 0   // development builds (scripts/build_variant.sh): ablation mask of the fp32 / f32s ring kernels, see the uses below
#endif

namespace hgk {

constexpr int BRF_W1_STAGES = 8, BRF_W2_STAGES = 36, BRF_KH_STAGES = BRF_W1_STAGES + BRF_W2_STAGES, BRF_W3_STAGES = 16;
constexpr int BRF_NSTAGE = 2 * BRF_KH_STAGES + BRF_W3_STAGES;   // 104

// fp32 blob -> weight stream of one bottleneck.  One thread per 16-byte chunk (4 floats): 104 stages x 512 chunks.
__global__ __launch_bounds__(256) void bt_ring_pack_f32_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                               const float* __restrict__ w3, unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= BRF_NSTAGE * 512) return;
    const int s = idx >> 9, rem = idx & 511, c = rem & 3;
    const float* src;
    int dst;
    if (s < 2 * BRF_KH_STAGES) {
        const int kh = s / BRF_KH_STAGES, q = s % BRF_KH_STAGES;
        if (q < BRF_W1_STAGES) {   // two K steps (16 floats each) of W1 rows 64 kh .. 64 kh + 63
            const int sub = rem >> 8, n = (rem >> 2) & 63, ks = 2 * q + sub;
            src = w1 + (size_t)(kh * 64 + n) * 256 + 16 * ks + 4 * c;
            dst = sub * 4096 + br_swz(n, c);
        } else {
            const int tap = (q - BRF_W1_STAGES) >> 2, kc = (q - BRF_W1_STAGES) & 3, r = rem >> 2;
            src = w2 + ((size_t)tap * 128 + r) * 128 + kh * 64 + 16 * kc + 4 * c;
            dst = br_swz(r, c);
        }
    } else {
        const int k = s - 2 * BRF_KH_STAGES, nh = k >> 3, k8 = k & 7, r = rem >> 2;
        src = w3 + ((size_t)nh * 128 + r) * 128 + 16 * k8 + 4 * c;
        dst = br_swz(r, c);
    }
    *reinterpret_cast<u32x4*>(stream + (size_t)s * BR_STAGE_BYTES + dst) = *reinterpret_cast<const u32x4*>(src);
}

// max(x, 0) without the NaN-canonicalising v_max hipcc puts in front of fmaxf on MFMA results: as a signed integer a negative
// float is negative, so v_max_i32(bits, 0) is the ReLU (-0 -> +0).  A builtin, not inline assembly -- see br_relu_pk.
__device__ __forceinline__ float br_relu(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// TAIL (hg_c1_f32.h): phase 1 is not computed here -- t1 = relu(W1' relu(bn1 x) + b1') was written to HBM for every pixel by
// conv1_ring_f32_kernel and its 10 x 18 halo tile arrives by LDS-DMA, one 64-channel half at a time (lane -> (halo pixel,
// 16-byte slot), fetching the chunk that belongs in that slot of the swizzled tile; halo pixels outside the image fetch from a
// page of zeros = the 3x3 convolution's padding).  The W1 stages of the stream are skipped: the tail walks 88 of the 104.
template <bool UP, bool ADD2 = false, bool TAIL = false, typename T = float>   // T: float (exact-fp32 MFMA) or F32S (the same kernel, split products)
__global__ __launch_bounds__(256, 2) void bottleneck_ring_f32_kernel(BtRingArgs p) {
    static_assert(!(UP && ADD2), "the fused up-path sum is written by plain blocks");
    constexpr int CIN = 256, CO = 256, NT = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const t1_lds = smem + BR_RING_BYTES;
    float* const coef_lds = reinterpret_cast<float*>(smem + BR_RING_BYTES + BR_T1_BYTES);   // [0..255] bn1 scale, [256..511] shift (later b3), [512..639] b1
    unsigned long long* const valid_lds = reinterpret_cast<unsigned long long*>(smem + BR_RING_BYTES + BR_T1_BYTES + BR_COEF_BYTES);
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DF3D_BT_TIMING
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles_x = p.W / BT_TW, tiles_y = p.H / BT_TH;
    // XCD-aware tile order (speed only): workgroup b runs on XCD b % 8, so XCD x takes the x-th contiguous eighth of the tiles and
    // the 64 workgroups resident on it work on neighbouring tiles, whose halos then meet in that XCD's L2 (bijective for any grid)
    int b;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tx0 = (b % tiles_x) * BT_TW;
    b /= tiles_x;
    const int ty0 = (b % tiles_y) * BT_TH;
    const int view = b / tiles_y;
    const unsigned char* const xin = reinterpret_cast<const unsigned char*>(p.in) + (size_t)view * p.H * p.W * CIN * 4;
    const unsigned char* const xin2 = UP ? reinterpret_cast<const unsigned char*>(p.in2) + (size_t)view * (p.H / 2) * (p.W / 2) * CIN * 4 : nullptr;

    // ---- the weight ring (every stage index below is a compile-time constant: all loops are unrolled) ----------------------
    const unsigned wvoff = (unsigned)wave * 2048u + (unsigned)lane * 16u;
    // stage counter q -> ring slot q % 4; TAIL skips the W1 stages of the stream (q counts W2 kh 0 | W2 kh 1 | W3)
    constexpr int NQ = TAIL ? BRF_NSTAGE - 2 * BRF_W1_STAGES : BRF_NSTAGE;
    auto stream_index = [](int q) { return !TAIL ? q : q < BRF_W2_STAGES ? BRF_W1_STAGES + q : q < 2 * BRF_W2_STAGES ? 2 * BRF_W1_STAGES + q : 2 * BRF_W1_STAGES + q; };
    auto ring_issue = [&](int q) {   // this wave copies pieces 2 wave, 2 wave + 1
        if ((BRF_ABLM & 2) && q >= 3) return;   // (ablation mask 2: no weight DMA after the prologue's stages)
        if (q < NQ) {
            const unsigned dst = ring_addr + (unsigned)(q % BR_RING) * BR_STAGE_BYTES + (unsigned)wave * 2048;
            br_glds_stage(reinterpret_cast<const unsigned char*>(p.wstream) + (size_t)stream_index(q) * BR_STAGE_BYTES, wvoff, dst);
        }
    };
    const unsigned char* const wf0 = ring + br_swz(l31, half);
    const unsigned char* const wf1 = ring + br_swz(l31, 2 + half);

    // coefficients -> LDS, t2 start values (b2) straight into the accumulators, b3 waits in a register until bn1 is dead
    // TAIL: the t1 halo tile of half kh by LDS-DMA.  Piece pc (1 KB) = halo pixels 4 pc .. 4 pc + 3, lane -> (pixel 4 pc + (lane >> 4),
    // slot lane & 15), fetching chunk slot ^ swizzle(pixel) of that pixel's 256-byte half row; this wave copies pieces wave, wave + 4, ...
    const unsigned t1_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)t1_lds;
    auto t1_issue = [&](int kh) {
#ifdef BRF_NO_T1DMA   // development builds: the tail without its t1 halo DMA (what that traffic and its exposed latency cost)
        return;
#endif
        const unsigned char* const tin = reinterpret_cast<const unsigned char*>(p.t1in) + (size_t)view * p.H * p.W * 512;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int pc = wave + 4 * k;
            if (pc < BT_HALO / 4) {
                const int hp = 4 * pc + (lane >> 4);
                const int hy = hp / BT_HW, hx = hp % BT_HW;
                const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
                const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                const unsigned chunk = (unsigned)((lane & 15) ^ br_t1_swz(hp));
                const unsigned char* const src = ok ? tin + ((size_t)y * p.W + x) * 512 + kh * 256 + chunk * 16 : reinterpret_cast<const unsigned char*>(p.zeros) + chunk * 16;
                br_glds_piece64(src, t1_addr + (unsigned)pc * 1024u);
            }
        }
    };
    if constexpr (TAIL) t1_issue(0);   // first of all: its latency runs under the rest of the prologue (nobody else touches the t1 region yet)
    // (loaded now, stored to LDS behind the first DMA requests: a store in front of them would make the wave sit out the
    // coefficients' round trip before it requests anything else)
    float late_b3 = p.b3[tid];
    float pre_s1 = 0.0f, pre_t1 = 0.0f, pre_b1 = 0.0f;
    if constexpr (!TAIL) {
        pre_s1 = p.s1[tid];
        pre_t1 = p.t1[tid];
        pre_b1 = p.b1[tid & 127];
    }
    const float* const b1_lds = coef_lds + 512;
    const float* const b3_lds = coef_lds + 256;
    f32x16 t2[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + 32 * m + 8 * q + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) t2[m][4 * q + e] = bb[e];
        }
    if (tid < BT_HROWS) {
        const int hy = tid / BT_HW, hx = tid % BT_HW;
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
        const bool ok = tid < BT_HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const unsigned long long m = __ballot(ok);
        if (lane == 0) valid_lds[wave] = m;
    }
    ring_issue(0);
    ring_issue(1);
    ring_issue(2);
    if constexpr (TAIL) {
        coef_lds[256 + tid] = late_b3;   // no bn1, no b1 here: b3 sits in its place from the start
    } else {
        coef_lds[tid] = pre_s1;
        coef_lds[256 + tid] = pre_t1;
        if (tid < 128) coef_lds[512 + tid] = pre_b1;
    }

    // x staging: thread -> (row = (tid + 256 i) >> 2, 16-byte chunk = tid & 3) of a 16-float K step
    constexpr int XP = 3;
    const int xchunk = tid & 3;
    const unsigned char* xp[XP];
    const unsigned char* xq[UP ? XP : 1];
    unsigned xkeep[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int hp = (tid >> 2) + 64 * i;
        const int hy = hp / BT_HW, hx = hp % BT_HW;
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
        const bool ok = hp < BT_HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        xkeep[i] = ok ? 0xffffffffu : 0u;
        xp[i] = xin + (((size_t)(ok ? y : 0) * p.W + (ok ? x : 0)) * CIN + xchunk * 4) * 4;
        if constexpr (UP) xq[i] = xin2 + (((size_t)(ok ? (y >> 1) : 0) * (p.W / 2) + (ok ? (x >> 1) : 0)) * CIN + xchunk * 4) * 4;
    }
    constexpr int DX = UP ? 2 : 3;   // K steps of x requested ahead (registers); the LDS x ring has three slots either way
    u32x4 rx[DX][XP];
    u32x4 rb[UP ? DX : 1][XP];
    auto loadx = [&](int s, int slot) {
#pragma unroll
        for (int i = 0; i < XP; ++i) rx[slot][i] = *reinterpret_cast<const u32x4*>(xp[i] + s * 64);
        if constexpr (UP) {
#pragma unroll
            for (int i = 0; i < XP; ++i) rb[slot][i] = *reinterpret_cast<const u32x4*>(xq[i] + s * 64);
        }
    };
    auto storex = [&](int s, int slot) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(coef_lds + s * 16 + xchunk * 4);
        const f32x4 ct_ = *reinterpret_cast<const f32x4*>(coef_lds + 256 + s * 16 + xchunk * 4);
        unsigned char* const sx = t1_lds + (s % 3) * BR_XSTAGE;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            f32x4 v = __builtin_bit_cast(f32x4, rx[slot][i]);
            if constexpr (UP) v += __builtin_bit_cast(f32x4, rb[slot][i]);   // x = in + upsample(in2), what upadd_kernel would have stored
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(br_relu(fmaf(v[e], cs[e], ct_[e]))) & xkeep[i];
            *reinterpret_cast<u32x4*>(sx + ((tid >> 2) + 64 * i) * BR_XPITCH + br_xslot(tid >> 2, xchunk)) = o;
        }
    };

    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;   // this wave's 32 pixels (phases 2, 3)
    const unsigned char* const t1_lane = t1_lds + (py * BT_HW + px) * BR_T1_PITCH;
    unsigned tsw[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) tsw[kx] = (unsigned)((((px + kx) & 15) ^ half) << 4);
    // phase 1: wave -> channel tile ct (of the half's two) and row tiles rt0 .. rt0 + 2
    const int ct = wave & 1, rt0 = (wave >> 1) * 3;

#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int base = kh * BRF_KH_STAGES;
        if constexpr (TAIL) {
            br_barrier();   // kh = 0: masks and b3 visible; kh = 1: every wave has finished reading the first t1 half
            BR_STAMP(kh == 0 ? 0 : 3);
            if (kh == 1) t1_issue(kh);   // (half 0 was requested at the top of the kernel)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces (and the weight stages requested before them) have landed;
                                                                // the barrier of the first double-step below publishes the tile
            BR_STAMP(2);
        } else {
            // ---- phase 1 (half kh): t1^T = relu(W1' relu(bn1 x)^T + b1') on the halo, 16 K steps of 16 floats ---------------
    #pragma unroll
            for (int k = 0; k < DX; ++k) loadx(k, k);
            br_barrier();   // kh = 0: coefficients / masks visible; kh = 1: every wave has finished reading the first t1 half
            BR_STAMP(kh == 0 ? 0 : 3);
            f32x16 acc[3];
    #pragma unroll
            for (int t = 0; t < 4; ++t) {   // register 4 t + e <-> channel 64 kh + 32 ct + 8 t + 4 half + e
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b1_lds + kh * 64 + ct * 32 + 8 * t + 4 * half);
    #pragma unroll
                for (int i = 0; i < 3; ++i)
    #pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][4 * t + e] = bb[e];
            }
    #pragma unroll
            for (int xs = 0; xs < 16; ++xs) {
                storex(xs, xs % DX);
                const int st = base + (xs >> 1);     // W1 stage of this step: two K steps per stage
                // stages requested after `st` so far: st + 1, st + 2 (only st + 1 at the start of the second half, whose first
                // two stages were requested by the last double-step of the first half's phase 2)
                if ((xs & 1) == 0) br_wait_vm(kh == 1 && xs == 0 ? 2 : 4);
                br_barrier();
                if ((xs & 1) == 0) {
                    if (kh == 1 && xs == 0) ring_issue(st + 2);
                    ring_issue(st + 3);
                }
                if (xs + DX < 16) loadx(xs + DX, xs % DX);
                const unsigned char* const sx = t1_lds + (xs % 3) * BR_XSTAGE;
                if constexpr (std::is_same<T, F32S>::value) {   // the K step as a whole: both weight fragments against the split activation pair
                    const u32x4 w0 = *reinterpret_cast<const u32x4*>(wf0 + (st % BR_RING) * BR_STAGE_BYTES + (xs & 1) * 4096 + ct * 2048);
                    const u32x4 w1 = *reinterpret_cast<const u32x4*>(wf1 + (st % BR_RING) * BR_STAGE_BYTES + (xs & 1) * 4096 + ct * 2048);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const unsigned char* const row = sx + ((rt0 + i) * 32 + l31) * BR_XPITCH;
                        const XPair<T> xp2 = make_xpair<T>(*reinterpret_cast<const u32x4*>(row + br_xslot(l31, half)), *reinterpret_cast<const u32x4*>(row + br_xslot(l31, 2 + half)));
                        mfma_pair<T, true>(w0, w1, xp2, acc[i]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const u32x4 wf = *reinterpret_cast<const u32x4*>((j ? wf1 : wf0) + (st % BR_RING) * BR_STAGE_BYTES + (xs & 1) * 4096 + ct * 2048);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const u32x4 xf = *reinterpret_cast<const u32x4*>(sx + ((rt0 + i) * 32 + l31) * BR_XPITCH + br_xslot(l31, 2 * j + half));
                            mfma_chunk<T>(wf, xf, acc[i]);
                        }
                    }
                }
            }
            br_barrier();   // every wave is done with the x ring: the t1 half may overwrite it
            BR_STAMP(1);
    #pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int hp = (rt0 + i) * 32 + l31;
                const unsigned keep = 0u - (unsigned)((valid_lds[(rt0 + i) >> 1] >> (((rt0 + i) & 1) * 32 + l31)) & 1ull);
                unsigned char* const trow = t1_lds + hp * BR_T1_PITCH;
                const int sw = br_t1_swz(hp);
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    u32x4 w;
    #pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(br_relu(acc[i][4 * t + e])) & keep;
                    if (hp < BT_HALO) *reinterpret_cast<u32x4*>(trow + (((ct * 8 + 2 * t + half) ^ sw) << 4)) = w;
                }
            }
            if (kh == 1) coef_lds[256 + tid] = late_b3;   // bn1 is dead: b3 takes the shift vector's place (read in phase 3)
            BR_STAMP(2);
        }

        // ---- phase 2 (half kh): t2^T += W2'[:, half] (*) t1 half, two stages per barrier --------------------------------
#pragma unroll
        for (int d = 0; d < BRF_W2_STAGES / 2; ++d) {
            const int s0 = TAIL ? BRF_W2_STAGES * kh + 2 * d : base + BRF_W1_STAGES + 2 * d;   // stage counter (see ring_issue)
            br_wait_vm(TAIL ? 0 : d == 0 ? 2 : 0);   // d = 0: phase 1 has already requested stage s0 + 2
            br_barrier();                 // (first iteration: also publishes the t1 half)
            if (d > 0 || (TAIL && kh == 1)) ring_issue(s0 + 2);   // (fused: phase 1 requested it; TAIL, kh = 0: the prologue did)
            ring_issue(s0 + 3);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = s0 + u;
                const int q = TAIL ? s - BRF_W2_STAGES * kh : s - base - BRF_W1_STAGES, tap = q >> 2, kc = q & 3;
                const int ky = tap / 3, kx = tap - 3 * ky;
                if constexpr (std::is_same<T, F32S>::value) {   // the K step as a whole: the t1 pair split once, then four tiles of three MFMAs
                    const unsigned char* const trow = t1_lane + (ky * BT_HW + kx) * BR_T1_PITCH;
                    const XPair<T> tp = make_xpair<T>(*reinterpret_cast<const u32x4*>(trow + (tsw[kx] ^ (unsigned)((4 * kc) << 4))),
                                                      *reinterpret_cast<const u32x4*>(trow + (tsw[kx] ^ (unsigned)((4 * kc + 2) << 4))));
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        const u32x4 w0 = *reinterpret_cast<const u32x4*>(wf0 + (s % BR_RING) * BR_STAGE_BYTES + m * 2048);
                        const u32x4 w1 = *reinterpret_cast<const u32x4*>(wf1 + (s % BR_RING) * BR_STAGE_BYTES + m * 2048);
                        if (BRF_ABLM & 1) asm volatile("" ::"v"(w0), "v"(w1), "v"(tp.a), "v"(tp.b));
                        else mfma_pair<T, true>(w0, w1, tp, t2[m]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const u32x4 tf = *reinterpret_cast<const u32x4*>(t1_lane + (ky * BT_HW + kx) * BR_T1_PITCH + (tsw[kx] ^ (unsigned)((4 * kc + 2 * j) << 4)));
#pragma unroll
                        for (int m = 0; m < NT; ++m) {
                            const u32x4 wf = *reinterpret_cast<const u32x4*>((j ? wf1 : wf0) + (s % BR_RING) * BR_STAGE_BYTES + m * 2048);
                            if (BRF_ABLM & 1) asm volatile("" ::"v"(wf), "v"(tf));   // (ablation mask 1: no phase-2 MFMAs)
                            else if ((BRF_ABLM & 4) && m > 0) continue;              // (ablation mask 4: a quarter of the phase-2 MFMAs and weight fragment reads)
                            else mfma_chunk<T>(wf, tf, t2[m]);
                        }
                    }
                }
            }
        }
    }
    BR_STAMP(3);
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) t2[m][r] = br_relu(t2[m][r]);
    // F32S: relu(t2) is phase 3's activation operand and every K step of it meets eight weight fragment pairs: split ONCE (the same 64 registers)
    XPair<T> t2p[NT][2];   // [tile][16-channel K step q2]: registers 8 q2 .. 8 q2 + 7 = channels 32 tile + 16 q2 + {4 half + e, 8 + 4 half + e}
    if constexpr (std::is_same<T, F32S>::value) {
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                t2p[m][q] = make_xpair<T>(t2[m][8 * q], t2[m][8 * q + 1], t2[m][8 * q + 2], t2[m][8 * q + 3], t2[m][8 * q + 4], t2[m][8 * q + 5], t2[m][8 * q + 6], t2[m][8 * q + 7]);
    }

    if constexpr (std::is_same<T, float>::value) {
        // exact fp32 keeps round 3's phase 3 (rows = pixels, 4-byte accesses of 128 contiguous bytes): the transposed form below is bit-identical
        // and measured 0.5-0.7 % SLOWER here, same box (5 542-5 560 against 5 500-5 527 us per average launch) -- the matrix pipe, not the
        // epilogue's memory operations, bounds this instantiation
        // ---- phase 3: out = W3 relu(t2) + b3 + x  (rows = the wave's pixels, columns = channels, as in the register-staged
        //      kernel: its epilogue -- residual add, 4-byte stores of 128 contiguous bytes per pixel, in-lane pooling -- is kept) ----
        unsigned char* const outp = reinterpret_cast<unsigned char*>(p.out) + (size_t)view * p.H * p.W * CO * 4;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            f32x16 acc[4];
            float xr[2][16];   // residual values of channel tiles 0 and 1, requested during the last two double-steps
            auto load_res = [&](int i, float (&dst)[16]) {
                const int n = nh * 128 + i * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pl = (r & 3) + 8 * (r >> 2) + 4 * half;
                    dst[r] = reinterpret_cast<const float*>(xin)[((size_t)(ty0 + 2 * wave + (pl >> 4)) * p.W + (tx0 + (pl & 15))) * CIN + n];
                }
            };
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int s0 = (TAIL ? 2 * BRF_W2_STAGES : 2 * BRF_KH_STAGES) + 8 * nh + 2 * dd;
                br_wait_vm(0);   // the pair was requested a whole double-step ago
                br_barrier();
                ring_issue(s0 + 2);
                ring_issue(s0 + 3);
                if (dd >= 2) load_res(dd - 2, xr[dd - 2]);
                if (dd == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float bias = b3_lds[nh * 128 + i * 32 + l31];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][r] = bias;
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k8 = 2 * dd + u, s = s0 + u, tile = k8 >> 1, q2 = k8 & 1;
                    // registers 8 q2 + 4 jj + e of t2 tile `tile` hold channels 32 tile + 16 q2 + 8 jj + 4 half + e: 16-byte chunk 2 jj + half of the stage
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 wf = *reinterpret_cast<const f32x4*>((jj ? wf1 : wf0) + (s % BR_RING) * BR_STAGE_BYTES + i * 2048);
                            mfma_quad<T>(t2[tile][8 * q2 + 4 * jj], t2[tile][8 * q2 + 4 * jj + 1], t2[tile][8 * q2 + 4 * jj + 2], t2[tile][8 * q2 + 4 * jj + 3], wf, acc[i]);
                        }
                }
            }
            BR_STAMP(4 + 2 * nh);
            // epilogue: D[row = pixel (r&3) + 8(r>>2) + 4 half of the wave][col = channel nh*128 + 32 i + l31]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nh * 128 + i * 32 + l31;
                float xv[16];
                if (i < 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) xv[r] = xr[i][r];
                } else {
#ifdef BRF_NO_LATE_RES   // development builds: what the two residual tiles requested in the epilogue cost
#pragma unroll
                    for (int r = 0; r < 16; ++r) xv[r] = 0.0f;
#else
                    load_res(i, xv);
#endif
                }
                if constexpr (UP) {
                    // the wave's two tile rows share ONE half-resolution row and neighbouring columns one pixel: registers r, r^1, r^8,
                    // r^9 take the same addend -> 4 loads per tile, key = bits 1 and 2 of r
                    float t4[4];
#pragma unroll
                    for (int key = 0; key < 4; ++key)
                        t4[key] = reinterpret_cast<const float*>(xin2)[((size_t)(ty0 / 2 + wave) * (p.W / 2) + tx0 / 2 + (key & 1) + 4 * (key >> 1) + 2 * half) * CIN + n];
#pragma unroll
                    for (int r = 0; r < 16; ++r) xv[r] += t4[((r >> 1) & 1) + 2 * ((r >> 2) & 1)];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] += xv[r];
                if constexpr (ADD2) {   // + nearest-upsample(add2): a second fp32 add, as upadd_kernel would have done on the stored tensor
                    // (registers r, r^1, r^8, r^9 share one half-resolution pixel: key = bits 1 and 2 of r; one value live at a time -- the
                    // kernel has no registers to spare)
                    const float* const lrow = reinterpret_cast<const float*>(p.add2) + ((size_t)view * (p.H / 2) * (p.W / 2) + (size_t)(ty0 / 2 + wave) * (p.W / 2) + tx0 / 2 + 2 * half) * CO + n;
#pragma unroll
                    for (int key = 0; key < 4; ++key) {
                        const float t = lrow[((key & 1) + 4 * (key >> 1)) * CO];
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (((r >> 1) & 1) + 2 * ((r >> 2) & 1) == key) acc[i][r] += t;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pl = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const size_t po = ((size_t)(ty0 + 2 * wave + (pl >> 4)) * p.W + (tx0 + (pl & 15))) * CO + n;
#ifdef BRF_NT_STORES   // development switch: streaming stores of the block output (measured in round 4: see DESIGN.md 8)
                    __builtin_nontemporal_store(acc[i][r], reinterpret_cast<float*>(outp) + po);
#else
                    reinterpret_cast<float*>(outp)[po] = acc[i][r];
#endif
                }
                if constexpr (!UP) if (p.pool_in) {   // 2x2 max-pool of the block's INPUT (the skip values just added), same in-lane geometry as below (the engine asks for it on plain blocks only)
                    float* const pp = reinterpret_cast<float*>(p.pool_in) + (size_t)view * (p.H / 2) * (p.W / 2) * CIN;
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                        for (int b2 = 0; b2 < 2; ++b2) {
                            const int r0 = 2 * a2 + 4 * b2;
                            const float v = fmaxf(fmaxf(xv[r0], xv[r0 + 1]), fmaxf(xv[r0 + 8], xv[r0 + 9]));
                            const int ppx = a2 + 4 * b2 + 2 * half;
                            pp[((size_t)(ty0 / 2 + wave) * (p.W / 2) + (tx0 / 2 + ppx)) * CIN + n] = v;
                        }
                }
                if (p.pool) {
                    // 2x2 max-pool inside the lane: horizontal neighbour = register r^1, vertical neighbour = r^8
                    float* const pp = reinterpret_cast<float*>(p.pool) + (size_t)view * (p.H / 2) * (p.W / 2) * CO;
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                        for (int b2 = 0; b2 < 2; ++b2) {
                            const int r0 = 2 * a2 + 4 * b2;
                            const float v = fmaxf(fmaxf(acc[i][r0], acc[i][r0 + 1]), fmaxf(acc[i][r0 + 8], acc[i][r0 + 9]));
                            const int ppx = a2 + 4 * b2 + 2 * half;
                            pp[((size_t)(ty0 / 2 + wave) * (p.W / 2) + (tx0 / 2 + ppx)) * CO + n] = v;
                        }
                }
            }
            BR_STAMP(5 + 2 * nh);
        }
    } else {
        // ---- phase 3: out^T = W3 relu(t2)^T + b3 + x^T  (round 5, F32S: TRANSPOSED -- rows = channels, columns = the wave's pixels, like t2 itself:
        //      W3's fragment is the MFMA's first operand, the t2 registers its second; every accumulator sees the same products in the same K
        //      order as before (an MFMA's operands commute), so the result is bit-identical.  A lane now owns ONE pixel and, per 32-channel tile,
        //      four groups of four CONSECUTIVE channels (register 4 q + e <-> channel 8 q + 4 half + e): residual, upsampled addends and output
        //      move as 16-byte accesses -- 16 loads and 16 stores per lane and output half instead of 64 + 64 four-byte ones -- and the 2x2
        //      max-pools are taken across lanes (x neighbour = lane ^ 1, y neighbour = lane ^ 16). ----
        unsigned char* const outp = reinterpret_cast<unsigned char*>(p.out) + (size_t)view * p.H * p.W * CO * 4;
        const size_t pix_off = ((size_t)(ty0 + py) * p.W + (tx0 + px)) * (CO * 4);                                      // this lane's pixel (CIN == CO)
        const size_t hpix_off = ((size_t)(ty0 / 2 + wave) * (p.W / 2) + (tx0 / 2 + (px >> 1))) * (CO * 4);               // its half-resolution pixel
        const bool pool_lane = (l31 & 17) == 0;   // even column of the wave's first row: stores the pooled pixel of its 2x2 quad
        auto quad_max = [](float v) {             // max over the lane's 2x2 pixel quad (lanes ^1, ^16, ^17)
            const float h = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));   // quad_perm [1, 0, 3, 2]
            return fmaxf(h, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, h), 0x401F)));                              // xor 16 inside 32 lanes
        };
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            f32x16 acc[4];
            f32x4 xr[2][4];   // residual values of channel tiles 0 and 1, requested during the last two double-steps (all four requested there: no faster)
            auto load_res = [&](int i, f32x4 (&dst)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4*>(xin + pix_off + (nh * 128 + i * 32 + 8 * q + 4 * half) * 4);
            };
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int s0 = (TAIL ? 2 * BRF_W2_STAGES : 2 * BRF_KH_STAGES) + 8 * nh + 2 * dd;
                br_wait_vm(0);   // the pair was requested a whole double-step ago
                br_barrier();
                ring_issue(s0 + 2);
                ring_issue(s0 + 3);
                if (dd >= 2) load_res(dd - 2, xr[dd - 2]);
                if (dd == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(b3_lds + nh * 128 + i * 32 + 8 * q + 4 * half);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = bb[e];
                        }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k8 = 2 * dd + u, s = s0 + u, tile = k8 >> 1, q2 = k8 & 1;
                    // registers 8 q2 + 4 jj + e of t2 tile `tile` hold channels 32 tile + 16 q2 + 8 jj + 4 half + e: 16-byte chunk 2 jj + half of the stage
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        mfma_pair<T, true>(*reinterpret_cast<const u32x4*>(wf0 + (s % BR_RING) * BR_STAGE_BYTES + i * 2048),
                                           *reinterpret_cast<const u32x4*>(wf1 + (s % BR_RING) * BR_STAGE_BYTES + i * 2048), t2p[tile][q2], acc[i]);
                }
            }
            BR_STAMP(4 + 2 * nh);
            // epilogue: D^T[row = channel nh*128 + 32 i + (r&3) + 8 (r>>2) + 4 half][col = this lane's pixel]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 xv[4];
                if (i < 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[q] = xr[i][q];
                } else {
#ifdef BRF_NO_LATE_RES   // development builds: what the two residual tiles requested in the epilogue cost
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#else
                    load_res(i, xv);
#endif
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = (nh * 128 + i * 32 + 8 * q + 4 * half) * 4;   // byte offset of the lane's four channels inside a pixel
                    if constexpr (UP) xv[q] += *reinterpret_cast<const f32x4*>(xin2 + hpix_off + c4);   // x = in + upsample(in2): the pixel's half-resolution parent
                    f32x4 o = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                    o += xv[q];
                    if constexpr (ADD2)   // + nearest-upsample(add2): a second fp32 add, as upadd_kernel would have done on the stored tensor
                        o += *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(p.add2) + (size_t)view * (p.H / 2) * (p.W / 2) * (CO * 4) + hpix_off + c4);
#ifdef BRF_NT_STORES   // development switch: streaming stores of the block output (measured in round 4: see DESIGN.md 8)
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(outp + pix_off + c4));
#else
                    *reinterpret_cast<f32x4*>(outp + pix_off + c4) = o;
#endif
                    if constexpr (!UP) if (p.pool_in) {   // 2x2 max-pool of the block's INPUT (the skip values just added; the engine asks for it on plain blocks only)
                        f32x4 m;
#pragma unroll
                        for (int e = 0; e < 4; ++e) m[e] = quad_max(xv[q][e]);
                        if (pool_lane) *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.pool_in) + (size_t)view * (p.H / 2) * (p.W / 2) * (CIN * 4) + hpix_off + c4) = m;
                    }
                    if (p.pool) {
                        f32x4 m;
#pragma unroll
                        for (int e = 0; e < 4; ++e) m[e] = quad_max(o[e]);
                        if (pool_lane) *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.pool) + (size_t)view * (p.H / 2) * (p.W / 2) * (CO * 4) + hpix_off + c4) = m;
                    }
                }
            }
            BR_STAMP(5 + 2 * nh);
        }
    }
}

}  // namespace hgk
