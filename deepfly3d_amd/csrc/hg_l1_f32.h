// fp32 layer1 (the 64 -> 64 -> 64 -> 128 bottleneck with the 1x1 skip convolution, 128 x 256 pixels per view: 11 % of the fp32 step)
// in the split form of hg_c1_f32.h:
//
//   conv1_ring_f32_kernel<false, 64, 64>   t1 = relu(W1' relu(bn1 x) + b1') for EVERY pixel, once            -> HBM [px][64] f32
//   layer1_tail_f32_kernel                 3x3 (64 -> 64), 1x1 (64 -> 128), the skip convolution Wd x (64 -> 128) and the 2x2
//                                          max-pool on 8 x 16 tiles; the t1 halo tile arrives by LDS-DMA, the weights through the
//                                          4-slot LDS-DMA ring, the skip convolution's x operand straight from global memory
//
// It replaces hg_kernels.h:bottleneck_kernel<float, 64, 64, true> (register-staged weights, two barriers per K step, conv1
// recomputed on the halo: 122 TFLOP/s) and reproduces it bit for bit: same products in the same K order in every accumulator
// (t2: bias, then tap-major, 8-float chunks ascending; out: b3 + bd, W3 over t2's channels ascending, then Wd over x's).
//
// Weight stream (26 stages of 8 KB, bt_l1f_pack_kernel): 18 x W2' -- per tap two stages, each holding TWO 16-float K slices of the
// 64 output rows (rows 0..63 of the image: slice 2 u, rows 64..127: slice 2 u + 1) --, 4 x W3 (128 rows, K slice k), 4 x Wd.
// A double-step (one barrier) consumes two stages = 64 MFMAs per wave; the next pair is requested right behind the barrier and
// awaited with vmcnt(0) a whole double-step (4 096 MFMA cycles) later.
#pragma once
#include "hg_c1_f32.h"

namespace hgk {

constexpr int L1F_W2_STAGES = 18, L1F_W3_STAGES = 4, L1F_WD_STAGES = 4;
constexpr int L1F_NSTAGE = L1F_W2_STAGES + L1F_W3_STAGES + L1F_WD_STAGES;   // 26
constexpr int L1F_LDS_BYTES = BR_RING_BYTES + BR_T1_BYTES + 128 * 4;        // ring | t1 halo tile (180 rows x 256 B) | b3 + bd

// fp32 blob -> the tail's weight stream.  One thread per 16-byte chunk: 26 stages x 128 rows x 4 chunks.
__global__ __launch_bounds__(256) void bt_l1f_pack_kernel(const float* __restrict__ w2, const float* __restrict__ w3, const float* __restrict__ wd,
                                                          unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L1F_NSTAGE * 512) return;
    const int s = idx >> 9, r = (idx >> 2) & 127, c = idx & 3;
    const float* src;
    if (s < L1F_W2_STAGES) {
        const int tap = s >> 1, kc = 2 * (s & 1) + (r >> 6), n = r & 63;
        src = w2 + ((size_t)tap * 64 + n) * 64 + 16 * kc + 4 * c;                 // W2 [9][64][64]
    } else if (s < L1F_W2_STAGES + L1F_W3_STAGES) {
        src = w3 + (size_t)r * 64 + 16 * (s - L1F_W2_STAGES) + 4 * c;             // W3 [128][64]
    } else {
        src = wd + (size_t)r * 64 + 16 * (s - L1F_W2_STAGES - L1F_W3_STAGES) + 4 * c;   // Wd [128][64]
    }
    *reinterpret_cast<u32x4*>(stream + (size_t)s * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(src);
}

// BtRingArgs: in = x [V, H, W, 64], t1in = [V, H, W, 64] (conv1's output), zeros, out = [V, H, W, 128] or nullptr (round 5: the engine's
// plan asks for the pooled tensor only -- layer1's full-resolution output has no other reader), pool (optional) = [V, H/2, W/2, 128],
// wstream, b2 [64], b3 [128], bd [128].
template <typename T = float>   // float or F32S (split products)
__global__ __launch_bounds__(256, 2) void layer1_tail_f32_kernel(BtRingArgs p) {
    constexpr int CIN = 64, CO = 128, NT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const t1_lds = smem + BR_RING_BYTES;
    float* const b3_lds = reinterpret_cast<float*>(smem + BR_RING_BYTES + BR_T1_BYTES);
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned t1_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)t1_lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles_x = p.W / BT_TW, tiles_y = p.H / BT_TH;
    int b;   // XCD-aware tile order, as in the ring kernels
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tx0 = (b % tiles_x) * BT_TW;
    b /= tiles_x;
    const int ty0 = (b % tiles_y) * BT_TH;
    const int view = b / tiles_y;

    const unsigned wvoff = (unsigned)wave * 2048u + (unsigned)lane * 16u;
    auto ring_issue = [&](int q) {   // stage q -> ring slot q % 4; this wave copies pieces 2 wave, 2 wave + 1
        if (q < L1F_NSTAGE)
            br_glds_stage(reinterpret_cast<const unsigned char*>(p.wstream) + (size_t)q * BR_STAGE_BYTES, wvoff,
                          ring_addr + (unsigned)(q % BR_RING) * BR_STAGE_BYTES + (unsigned)wave * 2048);
    };
    const unsigned char* const wf0 = ring + br_swz(l31, half);
    const unsigned char* const wf1 = ring + br_swz(l31, 2 + half);

    // the t1 halo tile by LDS-DMA: piece pc (1 KB) = halo pixels 4 pc .. 4 pc + 3, lane -> (pixel 4 pc + (lane >> 4), slot lane & 15),
    // fetching the chunk that belongs in that slot of the swizzled tile; pixels outside the image fetch zeros (the 3x3's padding)
    {
        const unsigned char* const tin = reinterpret_cast<const unsigned char*>(p.t1in) + (size_t)view * p.H * p.W * 256;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int pc = wave + 4 * k;
            if (pc < BT_HALO / 4) {
                const int hp = 4 * pc + (lane >> 4);
                const int hy = hp / BT_HW, hx = hp % BT_HW;
                const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
                const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                const unsigned chunk = (unsigned)((lane & 15) ^ br_t1_swz(hp));
                const unsigned char* const src = ok ? tin + ((size_t)y * p.W + x) * 256 + chunk * 16 : reinterpret_cast<const unsigned char*>(p.zeros) + chunk * 16;
                br_glds_piece64(src, t1_addr + (unsigned)pc * 1024u);
            }
        }
    }
    ring_issue(0);
    ring_issue(1);
    ring_issue(2);
    const float pre_b = tid < CO ? p.b3[tid] + p.bd[tid] : 0.0f;   // (the register-staged kernel's expression: one float add)
    f32x16 t2[NT];   // t2^T: rows = channels (register 4 q + e <-> channel 32 m + 8 q + 4 half + e), columns = the wave's 32 pixels
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + 32 * m + 8 * q + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) t2[m][4 * q + e] = bb[e];
        }
    if (tid < CO) b3_lds[tid] = pre_b;

    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;   // this wave's 32 pixels: tile rows 2 wave, 2 wave + 1
    const unsigned char* const t1_lane = t1_lds + (py * BT_HW + px) * BR_T1_PITCH;
    unsigned tsw[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) tsw[kx] = (unsigned)((((px + kx) & 15) ^ half) << 4);
    // the skip convolution's A operand: this lane's pixel, 16-byte chunk 2 jj + half of 16-float K slice k (requested in the last
    // double-step of phase 2)
    const unsigned char* const xpix = reinterpret_cast<const unsigned char*>(p.in) + (((size_t)view * p.H + (ty0 + py)) * p.W + (tx0 + px)) * (CIN * 4) + half * 16;
    f32x4 xfr[4][2];

    // ---- phase 2: t2^T = W2' (*) t1, one tap per double-step ---------------------------------------------------------------
#pragma unroll
    for (int d = 0; d < 9; ++d) {
        const int s0 = 2 * d;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // d = 0: the t1 pieces and stages 0..2; later: the pair requested a double-step ago
        br_barrier();                                       // (d = 0: also publishes b3 + bd)
        if (d > 0) ring_issue(s0 + 2);                      // (d = 0: the prologue requested it)
        ring_issue(s0 + 3);
        if (d == 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) xfr[k][jj] = *reinterpret_cast<const f32x4*>(xpix + k * 64 + jj * 32);
        }
        const int ky = d / 3, kx = d - 3 * ky;
        // eight groups (stage u, K slice sub of the stage, 8-float chunk j) of one t1 fragment, two weight fragments and eight MFMAs;
        // the three fragments of group g + 1 are requested before the MFMAs of group g (hipcc on its own requests them one MFMA
        // ahead and then waits: 24.1 -> 23.4 ms.  With SIXTEEN MFMAs per group -- layer2's tail, the identity-skip tails -- the same
        // hand-pipelining made the kernels 1-4 % slower: there hipcc's own order is the better one)
        if constexpr (std::is_same<T, F32S>::value) {   // whole K steps: the t1 pair split once, then NT tiles of three MFMAs
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int kc = 2 * u + sub;
                    const unsigned char* const trow = t1_lane + (ky * BT_HW + kx) * BR_T1_PITCH;
                    const XPair<T> tp = make_xpair<T>(*reinterpret_cast<const u32x4*>(trow + (tsw[kx] ^ (unsigned)((4 * kc) << 4))),
                                                      *reinterpret_cast<const u32x4*>(trow + (tsw[kx] ^ (unsigned)((4 * kc + 2) << 4))));
#pragma unroll
                    for (int m = 0; m < NT; ++m)
                        mfma_pair<T, true>(*reinterpret_cast<const u32x4*>(wf0 + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + (2 * sub + m) * 2048),
                                           *reinterpret_cast<const u32x4*>(wf1 + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + (2 * sub + m) * 2048), tp, t2[m]);
                }
            continue;
        }
        u32x4 tfv[2], wfv[2][NT];
        auto load_group = [&](int g, int buf) {
            const int u = g >> 2, sub = (g >> 1) & 1, j = g & 1, kc = 2 * u + sub;
            tfv[buf] = *reinterpret_cast<const u32x4*>(t1_lane + (ky * BT_HW + kx) * BR_T1_PITCH + (tsw[kx] ^ (unsigned)((4 * kc + 2 * j) << 4)));
#pragma unroll
            for (int m = 0; m < NT; ++m)
                wfv[buf][m] = *reinterpret_cast<const u32x4*>((j ? wf1 : wf0) + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + (2 * sub + m) * 2048);
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) load_group(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                if constexpr (!std::is_same<T, F32S>::value) mfma_chunk<T>(wfv[g & 1][m], tfv[g & 1], t2[m]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) t2[m][r] = br_relu(t2[m][r]);
    XPair<T> t2p[NT][2];   // F32S: relu(t2) split once per 16-channel K step (registers 8 q .. 8 q + 7 of a tile)
    if constexpr (std::is_same<T, F32S>::value) {
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                t2p[m][q] = make_xpair<T>(t2[m][8 * q], t2[m][8 * q + 1], t2[m][8 * q + 2], t2[m][8 * q + 3], t2[m][8 * q + 4], t2[m][8 * q + 5], t2[m][8 * q + 6], t2[m][8 * q + 7]);
    }

    // ---- phase 3: out = W3 relu(t2) + Wd x + (b3 + bd): rows = the wave's pixels, columns = channels -------------------------
    f32x16 acc[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        const int s0 = L1F_W2_STAGES + 2 * dd;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        br_barrier();
        ring_issue(s0 + 2);
        ring_issue(s0 + 3);
        if (dd == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float bias = b3_lds[i * 32 + l31];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = bias;
            }
        }
        if constexpr (std::is_same<T, F32S>::value) {   // whole K steps: two per double-step, four output tiles of three MFMAs each
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = (2 * dd + u) & 3;
                XPair<T> ap;
                if (dd < 2) ap = t2p[k >> 1][k & 1];
                else ap = make_xpair<T>(__builtin_bit_cast(u32x4, xfr[k][0]), __builtin_bit_cast(u32x4, xfr[k][1]));
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    mfma_pair<T, false>(*reinterpret_cast<const u32x4*>(wf0 + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + i * 2048),
                                        *reinterpret_cast<const u32x4*>(wf1 + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + i * 2048), ap, acc[i]);
            }
            continue;
        }
        // four groups (stage u, chunk jj) of four weight fragments and sixteen MFMAs, the fragments one group ahead
        f32x4 w3v[2][4];
        auto load_w3 = [&](int g, int buf) {
            const int u = g >> 1, jj = g & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) w3v[buf][i] = *reinterpret_cast<const f32x4*>((jj ? wf1 : wf0) + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + i * 2048);
        };
        load_w3(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) load_w3(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int u = g >> 1, jj = g & 1, k = (2 * dd + u) & 3;   // 16-float K slice of t2's (dd < 2) or x's 64 channels
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // t2 tile k >> 1, registers 8 (k & 1) + 4 jj + e <-> channels 16 k + 8 jj + 4 half + e: chunk 2 jj + half of the slice
                float av[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = dd < 2 ? t2[k >> 1][8 * (k & 1) + 4 * jj + e] : xfr[k][jj][e];
                if constexpr (!std::is_same<T, F32S>::value) mfma_quad<T>(av[0], av[1], av[2], av[3], w3v[g & 1][i], acc[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: D[row = pixel (r & 3) + 8 (r >> 2) + 4 half of the wave][col = channel 32 i + l31]; 2x2 max-pool inside the lane
    float* const outp = reinterpret_cast<float*>(p.out) + (size_t)view * p.H * p.W * CO;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = i * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pl = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (p.out) outp[((size_t)(ty0 + 2 * wave + (pl >> 4)) * p.W + (tx0 + (pl & 15))) * CO + n] = acc[i][r];   // (out == nullptr: pooled output only)
        }
        if (p.pool) {   // horizontal neighbour = register r ^ 1, vertical neighbour = r ^ 8
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
}

// =====================================================================================================================
// fp32 layer2 (128 -> 128 -> 128 -> 256 with the 1x1 skip convolution, 64 x 128 pixels per view) in the same split form:
//   conv1_ring_f32_kernel<false, 128, 128>   t1 for every pixel, once                                          -> HBM [px][128] f32
//   layer2_tail_f32_kernel                   as layer1's tail, with the t1 tile built and consumed in two 64-channel halves (kh) and
//                                            the output in two 128-channel halves (nh), like the identity-skip tail (hg_bt_ring_f32.h)
// Replaces bottleneck_kernel<float, 128, 128, true> bit for bit (t2: bias, half kh 0 tap-major, then half kh 1; out: b3 + bd, W3 over
// t2's 128 channels ascending, then Wd over x's 128).
// Weight stream (104 stages, bt_l2f_pack_kernel): per kh 36 x W2' (tap, 16-float K slice of the half; 128 rows), then per nh
// 8 x W3 (K slice k) and 8 x Wd (K slice k).
// =====================================================================================================================
constexpr int L2F_W2_STAGES = 36;                       // per t1 half
constexpr int L2F_NH_STAGES = 16;                       // per output half: 8 x W3, 8 x Wd
constexpr int L2F_NSTAGE = 2 * L2F_W2_STAGES + 2 * L2F_NH_STAGES;   // 104
constexpr int L2F_LDS_BYTES = BR_RING_BYTES + BR_T1_BYTES + 256 * 4;

__global__ __launch_bounds__(256) void bt_l2f_pack_kernel(const float* __restrict__ w2, const float* __restrict__ w3, const float* __restrict__ wd,
                                                          unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L2F_NSTAGE * 512) return;
    const int s = idx >> 9, r = (idx >> 2) & 127, c = idx & 3;
    const float* src;
    if (s < 2 * L2F_W2_STAGES) {
        const int kh = s / L2F_W2_STAGES, q = s % L2F_W2_STAGES, tap = q >> 2, kc = q & 3;
        src = w2 + ((size_t)tap * 128 + r) * 128 + kh * 64 + 16 * kc + 4 * c;     // W2 [9][128][128]
    } else {
        const int k = s - 2 * L2F_W2_STAGES, nh = k / L2F_NH_STAGES, k16 = k % L2F_NH_STAGES;
        if (k16 < 8) src = w3 + ((size_t)nh * 128 + r) * 128 + 16 * k16 + 4 * c;  // W3 [256][128]
        else src = wd + ((size_t)nh * 128 + r) * 128 + 16 * (k16 - 8) + 4 * c;    // Wd [256][128]
    }
    *reinterpret_cast<u32x4*>(stream + (size_t)s * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(src);
}

// BtRingArgs: in = x [V, H, W, 128], t1in = [V, H, W, 128], zeros, out = [V, H, W, 256], wstream, b2 [128], b3 [256], bd [256].
template <typename T = float>   // float or F32S (split products)
__global__ __launch_bounds__(256, 2) void layer2_tail_f32_kernel(BtRingArgs p) {
    constexpr int CIN = 128, CO = 256, NT = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const t1_lds = smem + BR_RING_BYTES;
    float* const b3_lds = reinterpret_cast<float*>(smem + BR_RING_BYTES + BR_T1_BYTES);
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned t1_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)t1_lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles_x = p.W / BT_TW, tiles_y = p.H / BT_TH;
    int b;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tx0 = (b % tiles_x) * BT_TW;
    b /= tiles_x;
    const int ty0 = (b % tiles_y) * BT_TH;
    const int view = b / tiles_y;

    const unsigned wvoff = (unsigned)wave * 2048u + (unsigned)lane * 16u;
    auto ring_issue = [&](int q) {
        if (q < L2F_NSTAGE)
            br_glds_stage(reinterpret_cast<const unsigned char*>(p.wstream) + (size_t)q * BR_STAGE_BYTES, wvoff,
                          ring_addr + (unsigned)(q % BR_RING) * BR_STAGE_BYTES + (unsigned)wave * 2048);
    };
    const unsigned char* const wf0 = ring + br_swz(l31, half);
    const unsigned char* const wf1 = ring + br_swz(l31, 2 + half);
    // the 64-channel half kh of the t1 halo tile by LDS-DMA (see layer1_tail_f32_kernel; a t1 row is 512 B here)
    auto t1_issue = [&](int kh) {
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
    t1_issue(0);
    ring_issue(0);
    ring_issue(1);
    ring_issue(2);
    const float pre_b = p.b3[tid] + p.bd[tid];   // (the register-staged kernel's expression: one float add)
    f32x16 t2[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + 32 * m + 8 * q + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) t2[m][4 * q + e] = bb[e];
        }
    b3_lds[tid] = pre_b;

    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;
    const unsigned char* const t1_lane = t1_lds + (py * BT_HW + px) * BR_T1_PITCH;
    unsigned tsw[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) tsw[kx] = (unsigned)((((px + kx) & 15) ^ half) << 4);
    const unsigned char* const xpix = reinterpret_cast<const unsigned char*>(p.in) + (((size_t)view * p.H + (ty0 + py)) * p.W + (tx0 + px)) * (CIN * 4) + half * 16;

    // ---- phase 2, per t1 half: t2^T += W2'[:, half] (*) t1 half, two stages per barrier ---------------------------------------
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        if (kh == 1) {
            br_barrier();   // every wave has finished reading the first t1 half
            t1_issue(1);
        }
#pragma unroll
        for (int d = 0; d < L2F_W2_STAGES / 2; ++d) {
            const int s0 = L2F_W2_STAGES * kh + 2 * d;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // d = 0: the t1 pieces too
            br_barrier();                                       // (very first: also publishes b3 + bd)
            if (d > 0 || kh == 1) ring_issue(s0 + 2);           // (kh = 0, d = 0: the prologue requested it)
            ring_issue(s0 + 3);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = 2 * d + u, tap = q >> 2, kc = q & 3;
                const int ky = tap / 3, kx = tap - 3 * ky;
                if constexpr (std::is_same<T, F32S>::value) {   // the K step as a whole: the t1 pair split once, then four tiles of three MFMAs
                    const unsigned char* const trow = t1_lane + (ky * BT_HW + kx) * BR_T1_PITCH;
                    const XPair<T> tp = make_xpair<T>(*reinterpret_cast<const u32x4*>(trow + (tsw[kx] ^ (unsigned)((4 * kc) << 4))),
                                                      *reinterpret_cast<const u32x4*>(trow + (tsw[kx] ^ (unsigned)((4 * kc + 2) << 4))));
#pragma unroll
                    for (int m = 0; m < NT; ++m)
                        mfma_pair<T, true>(*reinterpret_cast<const u32x4*>(wf0 + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + m * 2048),
                                           *reinterpret_cast<const u32x4*>(wf1 + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + m * 2048), tp, t2[m]);
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const u32x4 tf = *reinterpret_cast<const u32x4*>(t1_lane + (ky * BT_HW + kx) * BR_T1_PITCH + (tsw[kx] ^ (unsigned)((4 * kc + 2 * j) << 4)));
#pragma unroll
                        for (int m = 0; m < NT; ++m) {
                            const u32x4 wf = *reinterpret_cast<const u32x4*>((j ? wf1 : wf0) + ((s0 + u) % BR_RING) * BR_STAGE_BYTES + m * 2048);
                            mfma_chunk<T>(wf, tf, t2[m]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) t2[m][r] = br_relu(t2[m][r]);
    XPair<T> t2p[NT][2];   // F32S: relu(t2) split once per 16-channel K step (registers 8 q .. 8 q + 7 of a tile)
    if constexpr (std::is_same<T, F32S>::value) {
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                t2p[m][q] = make_xpair<T>(t2[m][8 * q], t2[m][8 * q + 1], t2[m][8 * q + 2], t2[m][8 * q + 3], t2[m][8 * q + 4], t2[m][8 * q + 5], t2[m][8 * q + 6], t2[m][8 * q + 7]);
    }

    // ---- phase 3, per output half: out = W3 relu(t2) + Wd x + (b3 + bd) ----------------------------------------------------------
    float* const outp = reinterpret_cast<float*>(p.out) + (size_t)view * p.H * p.W * CO;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
        f32x16 acc[4];
        f32x4 xfr[2][2];   // the skip convolution's A operand for one double-step: K slices 2 dd', 2 dd' + 1, chunks 2 jj + half (requested a double-step ahead)
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) {
            const int s0 = 2 * L2F_W2_STAGES + L2F_NH_STAGES * nh + 2 * dd;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            br_barrier();
            ring_issue(s0 + 2);
            ring_issue(s0 + 3);
            f32x4 xcur[2][2];
            if (dd >= 4) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) xcur[u][jj] = xfr[u][jj];
            }
            if (dd >= 3 && dd < 7) {   // x slices of the NEXT double-step (the vmcnt(0) at its head waits for them)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) xfr[u][jj] = *reinterpret_cast<const f32x4*>(xpix + (2 * (dd - 3) + u) * 64 + jj * 32);
            }
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
                const int s = s0 + u, k = (2 * dd + u) & 7;   // 16-float K slice of t2's (dd < 4) or x's 128 channels
                if constexpr (std::is_same<T, F32S>::value) {
                    XPair<T> ap;
                    if (dd < 4) ap = t2p[k >> 1][k & 1];
                    else ap = make_xpair<T>(__builtin_bit_cast(u32x4, xcur[u][0]), __builtin_bit_cast(u32x4, xcur[u][1]));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        mfma_pair<T, false>(*reinterpret_cast<const u32x4*>(wf0 + (s % BR_RING) * BR_STAGE_BYTES + i * 2048),
                                            *reinterpret_cast<const u32x4*>(wf1 + (s % BR_RING) * BR_STAGE_BYTES + i * 2048), ap, acc[i]);
                } else {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 wf = *reinterpret_cast<const f32x4*>((jj ? wf1 : wf0) + (s % BR_RING) * BR_STAGE_BYTES + i * 2048);
                            float av[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) av[e] = dd < 4 ? t2[k >> 1][8 * (k & 1) + 4 * jj + e] : xcur[u][jj][e];
                            mfma_quad<T>(av[0], av[1], av[2], av[3], wf, acc[i]);
                        }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nh * 128 + i * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pl = (r & 3) + 8 * (r >> 2) + 4 * half;
                outp[((size_t)(ty0 + 2 * wave + (pl >> 4)) * p.W + (tx0 + (pl & 15))) * CO + n] = acc[i][r];
            }
        }
    }
}

}  // namespace hgk
