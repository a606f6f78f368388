// Fused stack head of the hourglass (fc -> score -> fc_ / score_ + skip), moved out of hg_kernels.h because its bf16 form takes
// Wfc through the LDS-DMA stage ring of hg_bt_ring.h.
#pragma once
#include "hg_bt_ring.h"

namespace hgk {

// bf16 blob -> Wfc stage stream: 8 K steps x 2 row halves, every stage the LDS image of 128 rows x 32 K (br_swz order)
constexpr int HD_FC_STAGES = 16;
__global__ __launch_bounds__(256) void bt_fc_pack_kernel(const unsigned short* __restrict__ wfc, unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HD_FC_STAGES * 512) return;
    const int st = idx >> 9, r = (idx >> 2) & 127, c = idx & 3;
    const int s = st >> 1, rh = st & 1;
    const unsigned short* const src = wfc + (size_t)(rh * 128 + r) * 256 + 32 * s + 8 * c;   // Wfc [256][256]
    *reinterpret_cast<u32x4*>(stream + (size_t)st * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(src);
}

// float32 blob -> Wfc stage stream: 16 K steps (16 channels = 64 bytes) x 2 row halves, the same 128-row x 64-byte stage image
constexpr int HD_FC_STAGES_F32 = 32;
__global__ __launch_bounds__(256) void bt_fc_pack_f32_kernel(const float* __restrict__ wfc, unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HD_FC_STAGES_F32 * 512) return;
    const int st = idx >> 9, r = (idx >> 2) & 127, c = idx & 3;
    const int s = st >> 1, rh = st & 1;
    const float* const src = wfc + (size_t)(rh * 128 + r) * 256 + 16 * s + 4 * c;   // Wfc [256][256]
    *reinterpret_cast<u32x4*>(stream + (size_t)st * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(src);
}

// =====================================================================================================
// Fused stack head:   y = relu(Wfc r + bfc)                       (fc, BN folded; 256 -> 256)
//                     score = Wsc y + bsc                          (256 -> 19, padded to 32)
//   not LAST:         x_new = x + (Wfc_ y + bfc_) + (Wsc_ score + bsc_)
//   LAST:             heat-maps (NCHW float32) = score
// All 1x1: a workgroup owns 128 consecutive pixels, a wave 32 of them.  y is computed TRANSPOSED (rows = channels,
// columns = the wave's pixels) so that its accumulators serve directly as the B operand of the score GEMM
// (score^T = Wsc y^T) and as the A operand of the fc_ GEMM; score^T in turn is the A operand of score_.
// r and x are read once, x_new / the heat-maps written once; y and score never leave the registers.
// =====================================================================================================
struct HeadArgs {
    const void* r;       // [M, 256] output of the stack's residual block
    const void* x;       // [M, 256] stack input (not LAST)
    void* out;           // [M, 256] next stack input (not LAST)
    float* heat;         // [V, 19, HW] (LAST)
    const void* wfc;     // [256][256]
    const void* wsc;     // [32][256]   (K permuted for bf16)
    const void* wfc_;    // [256][256]  (K permuted for bf16)
    const void* wsc_;    // [256][32]   (K permuted for bf16)
    const float* bfc;    // [256]
    const float* bsc;    // [32]
    const float* bfc_;   // [256]
    const float* bsc_;   // [256]
    const void* fcstream;  // bf16 only: Wfc as 16 pre-swizzled 8 KB LDS stage images (bt_fc_pack_kernel), or nullptr
    const void* fc2stream; // bf16 only, not LAST: Wfc_ / Wsc_ as 18 stage images (bt_fc2_pack_kernel), or nullptr; needs M % 128 == 0
    long long M;
    int HW;
};

// bf16 blob -> phase-C stage stream: per 128-channel output half nh: Wfc_ rows 128 nh .. as 8 K slices of 32 (the host's K order),
// then the same rows of Wsc_ (K = 32) -> 2 x 9 stages
constexpr int HD_FC2_STAGES = 18;
__global__ __launch_bounds__(256) void bt_fc2_pack_kernel(const unsigned short* __restrict__ wfc2, const unsigned short* __restrict__ wsc2,
                                                          unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HD_FC2_STAGES * 512) return;
    const int st = idx >> 9, r = (idx >> 2) & 127, c = idx & 3;
    const int nh = st / 9, k = st % 9;
    const unsigned short* const src = k < 8 ? wfc2 + (size_t)(nh * 128 + r) * 256 + 32 * k + 8 * c     // Wfc_ [256][256]
                                            : wsc2 + (size_t)(nh * 128 + r) * 32 + 8 * c;              // Wsc_ [256][32]
    *reinterpret_cast<u32x4*>(stream + (size_t)st * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(src);
}

template <typename T, bool LAST>
struct HeadCfg {
    static constexpr int EB = Elem<T>::BYTES;
    static constexpr int RBA = 64;                                   // phase A staged row bytes
    static constexpr int STAGE_A = (256 + 128) * (RBA + 16);         // Wfc rows + r rows
    static constexpr int WSC_PITCH = 256 * EB + 16;
    static constexpr int WSC_BYTES = 32 * WSC_PITCH;
    static constexpr int RBC = 128;                                  // phase C staged row bytes
    static constexpr int STAGE_C = 128 * (RBC + 16);
    static constexpr int S1 = 2 * STAGE_A > WSC_BYTES ? 2 * STAGE_A : WSC_BYTES;
    static constexpr int S2 = S1 > 2 * STAGE_C ? S1 : 2 * STAGE_C;
    static constexpr int RING_SLOTS = 6;                             // 16-bit: LDS-DMA ring of 8 KB weight stages ...
    static constexpr int R_SLOTS = 3;                                // ... behind it three 8 KB K slices of the r tile (phase A: LDS-DMA too) ...
    static constexpr int SLICE_PITCH = 64 * 2 + 16;                  // ... whose space the epilogue re-uses: one 32 px x 64 ch slice per wave
    static constexpr int RING_BYTES = RING_SLOTS * 8192 + (LAST || EB == 4 ? R_SLOTS * 8192 : 4 * 32 * SLICE_PITCH);   // (LAST has no phase C; float32: the r ring in both)
    static constexpr int STAGE_BYTES = S2 > RING_BYTES ? S2 : RING_BYTES;
    static constexpr int MISC = (256 + 32 + 256) * 4;                // bfc | bsc | bfc_ + bsc_
    static constexpr int LDS_BYTES = STAGE_BYTES + MISC;
};

template <typename T, bool LAST>
__global__ __launch_bounds__(256, 2) void head_kernel(HeadArgs p) {
    using C = HeadCfg<T, LAST>;
    constexpr int EB = C::EB;
    constexpr int PER16 = Elem<T>::PER16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const stage = smem;
    float* const bfc_lds = reinterpret_cast<float*>(smem + C::STAGE_BYTES);
    float* const bsc_lds = bfc_lds + 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const long long m0 = (long long)blockIdx.x * 128;
    float* const bout_lds = bsc_lds + 32;   // not LAST: bfc_ + bsc_ (the sum the epilogue adds), staged once instead of 32 global loads per pass
    // the bias vectors: loaded now, stored to LDS behind the first DMA requests (a store in front of them would make the wave sit out
    // this round trip before it requests anything else); published by the first barrier of phase A
    const float pre_bfc = p.bfc[tid], pre_bsc = p.bsc[tid & 31];
    float pre_bout = 0.0f;
    if constexpr (!LAST) pre_bout = p.bfc_[tid] + p.bsc_[tid];
    auto publish_bias = [&]() {
        bfc_lds[tid] = pre_bfc;
        if (tid < 32) bsc_lds[tid] = pre_bsc;
        if constexpr (!LAST) bout_lds[tid] = pre_bout;
    };

    // ================= phase A: y^T = Wfc r^T ==========================================================
    f32x16 y[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[m][r] = 0.0f;
    if (p.fcstream != nullptr) {
        // Wfc arrives as pre-swizzled 8 KB stage images (two per 64-byte K step -- 32 channels in 16-bit, 16 in float32: row halves) through a six-slot LDS-DMA
        // ring, two K steps ahead, one barrier per step.  The r operand:
        //   LAST (RDMA): the same road -- per K step every wave copies the 64-byte K slice of ITS OWN 32 pixels (two 1 KB pieces,
        //     16 pixels each, four lanes per pixel) into a three-slot ring: no registers, 64-byte segments instead of the 32-byte
        //     ones a lane-per-pixel load makes, and only the wave itself reads them back (own vmcnt, no barrier involved).  Which
        //     16-byte chunk of its pixel a lane fetches is chosen so that the fragment reads are bank-conflict-free: slot c of
        //     pixel p holds chunk c ^ ((p >> 2) & 3).  1 650 -> 1 490 us on 896 views.
        //   not LAST: lane (l31, half) loads the 16-byte chunks the MFMAs want from its own pixel -- ALL of them up front, so that no
        //     activation load sits in the in-order queue between two weight stages.  (Measured with the DMA form: 3 460 us against
        //     3 290; with phase C's skip values requested at the top of the kernel as well 4 000 -- 256 registers do not hold them
        //     beside y's 128 accumulators.)
        //   float32: the DMA form in both heads (the register form would need 128 registers of r fragments).
        // Same MFMA K order as the staged form below in every case.
        constexpr bool RDMA = LAST || EB == 4;
        constexpr int NSTEP = 256 * EB / 64;
        constexpr int NSLOT = 6, RSLOT = C::R_SLOTS;
        static_assert(!RDMA || (NSLOT + RSLOT) * BR_STAGE_BYTES <= C::STAGE_BYTES, "the rings live in the stage area");
        const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)stage;
        const unsigned wuni = (unsigned)__builtin_amdgcn_readfirstlane(wave);
        const unsigned wvoff = wuni * 2048u + (unsigned)lane * 16u;
        // r pieces: lane -> (pixel 16 q + (lane >> 2) of the wave, chunk (lane & 3) ^ ((pixel >> 2) & 3)); rows past the end read the last pixel
        const long long mw = m0 + (long long)wuni * 32;
        const long long mb = mw < p.M ? mw : p.M - 1;   // uniform base pixel (a wave wholly past the end reads the last pixel: nothing of it is stored)
        unsigned rvoff[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pl = 16 * q + (lane >> 2);
            long long m = mw + pl;
            if (m >= p.M) m = p.M - 1;
            rvoff[q] = (unsigned)((m - mb) * (256 * EB) + (((lane & 3) ^ ((pl >> 2) & 3)) << 4));
        }
        const unsigned char* const rbase = reinterpret_cast<const unsigned char*>(p.r) + (size_t)mb * (256 * EB);
        auto issue_step = [&](int s) {   // both weight stages of K step s (this wave copies pieces 2 wave, 2 wave + 1 of each) and the r slice
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int st = 2 * s + rh;
                br_glds_stage(reinterpret_cast<const unsigned char*>(p.fcstream) + (size_t)st * BR_STAGE_BYTES, wvoff,
                              ring_addr + (unsigned)(st % NSLOT) * BR_STAGE_BYTES + wuni * 2048u);
            }
            if constexpr (RDMA) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    br_glds_piece(rbase + s * 64, rvoff[q], ring_addr + (unsigned)(NSLOT + s % RSLOT) * BR_STAGE_BYTES + wuni * 2048u + q * 1024u);
            }
        };
        u32x4 rfg[RDMA ? 1 : 8][2];
        if constexpr (!RDMA) {
            const long long m = mw + l31 < p.M ? mw + l31 : p.M - 1;   // rows past the end compute on a valid pixel; nothing of them is stored
            const unsigned char* const src = rbase + ((size_t)(m - mb) * 256 + half * 8) * 2;
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) rfg[s][j] = *reinterpret_cast<const u32x4*>(src + s * 64 + j * 32);
        }
        issue_step(0);
        issue_step(1);
        publish_bias();
        const unsigned char* const wf0 = stage + br_swz(l31, half);
        const unsigned char* const wf1 = stage + br_swz(l31, 2 + half);
        // this lane's B fragments: pixel l31 of the wave, chunk 2 j + half -> slot (2 j + half) ^ ((l31 >> 2) & 3)
        const unsigned char* const rfrag = stage + NSLOT * BR_STAGE_BYTES + wave * 2048 + l31 * 64;
        const unsigned rsw = (unsigned)((l31 >> 2) & 3);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            br_wait_vm(s < NSTEP - 1 ? (RDMA ? 6 : 4) : 0);   // the pieces of step s + 1 (requested one step ago) may still be in flight
            br_barrier();                // (first step: also publishes the bias vectors)
            if (s + 2 < NSTEP) issue_step(s + 2);   // into the slots step s - 1 has released
            if constexpr (RDMA) {
                u32x4 rf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) rf[j] = *reinterpret_cast<const u32x4*>(rfrag + (s % RSLOT) * BR_STAGE_BYTES + ((((unsigned)(2 * j + half)) ^ rsw) << 4));
                if constexpr (std::is_same<T, F32S>::value) {   // the K step as a whole: the r pair split once, eight row tiles of three MFMAs
                    const XPair<T> rp = make_xpair<T>(rf[0], rf[1]);
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            mfma_pair<T, true>(*reinterpret_cast<const u32x4*>(wf0 + ((2 * s + rh) % NSLOT) * BR_STAGE_BYTES + m * 2048),
                                               *reinterpret_cast<const u32x4*>(wf1 + ((2 * s + rh) % NSLOT) * BR_STAGE_BYTES + m * 2048), rp, y[4 * rh + m]);
                    continue;
                }
                // four groups of four MFMAs (K half j, row half of the stage pair), the four weight fragments of group g + 1 requested
                // BEFORE the MFMAs of group g (hipcc otherwise serialises ds_read -> wait -> MFMA on one register quad: hg_bt_ring.h)
                u32x4 wfr[2][4];
                auto load_group = [&](int g, int buf) {
                    const int j = g >> 1, rh = g & 1;
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        wfr[buf][m] = *reinterpret_cast<const u32x4*>((j ? wf1 : wf0) + ((2 * s + rh) % NSLOT) * BR_STAGE_BYTES + m * 2048);
                };
                load_group(0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < 3) load_group(g + 1, (g + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if constexpr (!std::is_same<T, F32S>::value) mfma_chunk<T>(wfr[g & 1][m], rf[g >> 1], y[4 * (g & 1) + m]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // (the hand-pipelined form above measured 4-5 % SLOWER here, same box: with 64 registers of r fragments beside y's
                // 128 accumulators hipcc's own schedule of the eight fragment reads is the better one)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x4 wfr[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m)
                        wfr[m] = *reinterpret_cast<const u32x4*>((j ? wf1 : wf0) + ((2 * s + (m >> 2)) % NSLOT) * BR_STAGE_BYTES + (m & 3) * 2048);
#pragma unroll
                    for (int m = 0; m < 8; ++m) mfma_chunk<T>(wfr[m], rfg[s][j], y[m]);
                }
            }
        }
        __syncthreads();   // every wave is done with the ring before the stage area is reused
    } else {
        publish_bias();
        constexpr int RB = C::RBA, PITCH = RB + 16, CPR = RB / 16, RPP = 256 / CPR;   // 4 chunks per row, 64 rows per pass
        constexpr int KE = RB / EB, NSTEPS = 256 / KE;
        constexpr int WP = 256 / RPP, RP = 128 / RPP;                                  // 4 and 2 passes
        constexpr int W_BYTES = 256 * PITCH;
        const int chunk = tid % CPR, srow = tid / CPR;
        u32x4 rw[WP], rr[RP];
        auto loadA = [&](int s) {
            const int c0 = s * KE + chunk * PER16;
#pragma unroll
            for (int i = 0; i < WP; ++i)
                rw[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.wfc) + ((size_t)(srow + i * RPP) * 256 + c0) * EB);
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                const long long m = m0 + srow + i * RPP;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (m < p.M) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.r) + ((size_t)m * 256 + c0) * EB);
                rr[i] = v;
            }
        };
        auto storeA = [&](int buf) {
            unsigned char* const sw = stage + buf * C::STAGE_A;
            unsigned char* const sr = sw + W_BYTES;
#pragma unroll
            for (int i = 0; i < WP; ++i) *reinterpret_cast<u32x4*>(sw + (srow + i * RPP) * PITCH + chunk * 16) = rw[i];
#pragma unroll
            for (int i = 0; i < RP; ++i) *reinterpret_cast<u32x4*>(sr + (srow + i * RPP) * PITCH + chunk * 16) = rr[i];
        };
        loadA(0);
        storeA(0);
        __syncthreads();
        for (int s = 0; s < NSTEPS; ++s) {
            const unsigned char* const sw = stage + (s & 1) * C::STAGE_A;
            const unsigned char* const sr = sw + W_BYTES;
            if (s + 1 < NSTEPS) loadA(s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RB / 32; j += 2) {   // fragment pairs (j, j + 1): one 64-byte K step (mfma_pair)
                const unsigned char* const rrow = sr + (wave * 32 + l31) * PITCH + half * 16;
                const XPair<T> rp = make_xpair<T>(*reinterpret_cast<const u32x4*>(rrow + j * 32), *reinterpret_cast<const u32x4*>(rrow + (j + 1) * 32));
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const unsigned char* const wrowa = sw + (m * 32 + l31) * PITCH + half * 16;
                    mfma_pair<T, true>(*reinterpret_cast<const u32x4*>(wrowa + j * 32), *reinterpret_cast<const u32x4*>(wrowa + (j + 1) * 32), rp, y[m]);
                }
            }
            if (s + 1 < NSTEPS) storeA((s & 1) ^ 1);
            __syncthreads();
        }
    }
    // bias + ReLU (channel of register r in tile m: 32m + (r&3) + 8(r>>2) + 4*half)
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bfc_lds + 32 * m + 8 * q + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[m][4 * q + e] = fmaxf(y[m][4 * q + e] + bb[e], 0.0f);
        }
    // F32S: y is the activation operand of the score convolution and of phase C: split once per 16-channel K step
    XPair<T> ysp[std::is_same<T, F32S>::value ? 8 : 1][2];
    if constexpr (std::is_same<T, F32S>::value) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
                ysp[m][q2] = make_xpair<T>(y[m][8 * q2], y[m][8 * q2 + 1], y[m][8 * q2 + 2], y[m][8 * q2 + 3], y[m][8 * q2 + 4], y[m][8 * q2 + 5], y[m][8 * q2 + 6], y[m][8 * q2 + 7]);
    }
    // 16-bit engines: pack y once (slot e of group q2 <-> register 8*q2 + e)
    u32x4 ypk[EB == 2 ? 8 : 1][2];
    if constexpr (EB == 2) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
                ypk[m][q2] = lp_pack8<T>(y[m][8 * q2], y[m][8 * q2 + 1], y[m][8 * q2 + 2], y[m][8 * q2 + 3], y[m][8 * q2 + 4], y[m][8 * q2 + 5], y[m][8 * q2 + 6],
                                         y[m][8 * q2 + 7]);
    }

    // ================= phase B: score^T = Wsc y^T  (A = Wsc from LDS, B = y registers) =================
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
    {
        // stage the whole padded Wsc [32][256] (the last barrier of phase A has passed: the stage area is free)
        constexpr int CH = 256 * EB / 16;   // 16-byte chunks per row
        // all of a thread's chunks are requested before the first is stored (rolled, the loop was 4 -- float32: 8 -- load -> store
        // round trips in series in every workgroup)
        constexpr int NW = 32 * CH / 256;
        u32x4 wv[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = tid + 256 * k, row = i / CH, ch = i % CH;
            wv[k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.wsc) + ((size_t)row * 256) * EB + ch * 16);
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = tid + 256 * k, row = i / CH, ch = i % CH;
            *reinterpret_cast<u32x4*>(stage + row * C::WSC_PITCH + ch * 16) = wv[k];
        }
        __syncthreads();
        const unsigned char* const wrow = stage + l31 * C::WSC_PITCH;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if constexpr (EB == 4) {
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    if constexpr (std::is_same<T, F32S>::value) {
                        const unsigned char* const w2p = wrow + (32 * m) * 4 + (4 * q2 + half) * 16;
                        mfma_pair<T, true>(*reinterpret_cast<const u32x4*>(w2p), *reinterpret_cast<const u32x4*>(w2p + 32), ysp[m][q2], sc);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const f32x4 wf = *reinterpret_cast<const f32x4*>(wrow + (32 * m) * 4 + (4 * q2 + 2 * jj + half) * 16);
                            mfma_quad<T, false>(y[m][8 * q2 + 4 * jj], y[m][8 * q2 + 4 * jj + 1], y[m][8 * q2 + 4 * jj + 2], y[m][8 * q2 + 4 * jj + 3], wf, sc);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const u32x4 wf = *reinterpret_cast<const u32x4*>(wrow + (32 * m + (2 * q2 + half) * 8) * 2);
                    sc = Lp<T>::mfma(wf, ypk[m][q2], sc);
                }
            }
        }
        __syncthreads();  // everyone is done reading Wsc before the stage area is reused
    }
    // score^T: lane = pixel, register r = score channel (r&3) + 8(r>>2) + 4*half
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bsc_lds + 8 * q + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[4 * q + e] += bb[e];
    }
    const long long mpix = m0 + wave * 32 + l31;   // this lane's pixel
    if constexpr (LAST) {
        if (mpix < p.M) {
            const long long view = mpix / p.HW;
            const int pix = (int)(mpix - view * p.HW);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c < 19) p.heat[((size_t)view * 19 + c) * p.HW + pix] = sc[r];
            }
        }
        return;
    } else {
        // ================= phase C: x_new = x + Wfc_ y + Wsc_ score + biases ==============================
        constexpr int RB = C::RBC, PITCH = RB + 16, CPR = RB / 16, RPP = 256 / CPR;   // 8 chunks per row, 32 rows per pass
        constexpr int KE = RB / EB;
        // output channels per pass: fp32 64 (two accumulator tiles: with y's 128 registers the kernel then fits 256 VGPRs and
        // two workgroups share a CU), bf16 128
        constexpr int NI = EB == 4 ? 2 : 4;
        constexpr int CH = NI * 32, NPASS = 256 / CH;
        constexpr int WPASS = CH / RPP;
        constexpr int YSTEPS = 256 / KE;          // K-steps over y (8 for f32, 4 for bf16); one more step for score
        const int chunk = tid % CPR, srow = tid / CPR;
        u32x4 rw[WPASS];
        // step s < YSTEPS: rows n of Wfc_ (row stride 256), K offset s*KE; step YSTEPS: rows of Wsc_ (row stride 32), first 32 K
        auto loadC = [&](int nh, int s) {
            const bool is_sc = s == YSTEPS;
            const unsigned char* base = reinterpret_cast<const unsigned char*>(is_sc ? p.wsc_ : p.wfc_);
            const size_t stride = is_sc ? 32 : 256;
            const size_t koff = is_sc ? 0 : (size_t)s * KE;
#pragma unroll
            for (int i = 0; i < WPASS; ++i) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (!is_sc || chunk * PER16 < 32)
                    v = *reinterpret_cast<const u32x4*>(base + ((size_t)(nh * CH + srow + i * RPP) * stride + koff + chunk * PER16) * EB);
                rw[i] = v;
            }
        };
        auto storeC = [&](int buf) {
            unsigned char* const sw = stage + buf * C::STAGE_C;
#pragma unroll
            for (int i = 0; i < WPASS; ++i) *reinterpret_cast<u32x4*>(sw + (srow + i * RPP) * PITCH + chunk * 16) = rw[i];
        };
        XPair<T> scsp[2];   // F32S: the 32 (padded) score channels as phase C's last K steps, split once
        if constexpr (std::is_same<T, F32S>::value) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
                scsp[q2] = make_xpair<T>(sc[8 * q2], sc[8 * q2 + 1], sc[8 * q2 + 2], sc[8 * q2 + 3], sc[8 * q2 + 4], sc[8 * q2 + 5], sc[8 * q2 + 6], sc[8 * q2 + 7]);
        }
        u32x4 scpk[2];
        if constexpr (EB == 2) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
                scpk[q2] = lp_pack8<T>(sc[8 * q2], sc[8 * q2 + 1], sc[8 * q2 + 2], sc[8 * q2 + 3], sc[8 * q2 + 4], sc[8 * q2 + 5], sc[8 * q2 + 6], sc[8 * q2 + 7]);
        }
        if constexpr (EB == 2) {
            if (p.fc2stream != nullptr) {
                // bf16: the 2 x 9 weight stages of phase C through the same six-slot LDS-DMA ring as phase A (two steps ahead, one
                // barrier per step); the skip values of BOTH passes are requested before the first stage (no activation load
                // between two weight stages in the in-order queue).  Operations one wave issues, in order: 16 skip loads |
                // steps 0, 1 | per step t: wait, barrier, step t + 2 | after step 4: 8 stores | ... | after step 9: 8 stores;
                // a step is two stages (four pieces per wave), the score steps 4 and 9 one.
                constexpr int NSLOT = C::RING_SLOTS, SP = C::SLICE_PITCH;
                const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)stage;
                const unsigned wpiece = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 2048u;
                auto issue_step = [&](int t) {
                    const int base = 9 * (t / 5) + 2 * (t % 5), cnt = (t % 5 == 4) ? 1 : 2;
#pragma unroll
                    for (int k = 0; k < cnt; ++k)
                        br_glds_stage(reinterpret_cast<const unsigned char*>(p.fc2stream) + (size_t)(base + k) * BR_STAGE_BYTES, wpiece + (unsigned)lane * 16u,
                                      ring_addr + (unsigned)((base + k) % NSLOT) * BR_STAGE_BYTES + wpiece);
                };
                u32x4 xch[2][2][4];
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int hc = 0; hc < 2; ++hc)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const long long m = m0 + wave * 32 + 8 * c + (lane >> 3);   // (M is a multiple of 128 here)
                            xch[nh][hc][c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.x) + (size_t)m * 256 + nh * 128 + 64 * hc + (lane & 7) * 8);
                        }
                issue_step(0);
                issue_step(1);
                const unsigned char* const wf0 = stage + br_swz(l31, half);
                const unsigned char* const wf1 = stage + br_swz(l31, 2 + half);
                unsigned char* const slice = stage + NSLOT * BR_STAGE_BYTES + wave * (32 * SP);
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    f32x16 acc[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
                    for (int s5 = 0; s5 < 5; ++s5) {
                        const int t = 5 * nh + s5;
                        // operations issued after step t's pieces: step t + 1's (4, or 2 for a score step), and for steps 5 and 6
                        // also the first pass' 8 stores
                        br_wait_vm(t == 9 ? 0 : ((t + 1) % 5 == 4 ? 2 : 4) + (t == 5 || t == 6 ? 8 : 0));
                        br_barrier();
                        if (t + 2 < 10) issue_step(t + 2);
                        const int base = 9 * nh + 2 * s5;
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm) {
                            if (s5 == 4 && mm >= 1) break;
#pragma unroll
                            for (int q2 = 0; q2 < 2; ++q2) {
                                const u32x4 af = s5 < 4 ? ypk[2 * s5 + mm][q2] : scpk[q2];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const u32x4 wf = *reinterpret_cast<const u32x4*>((q2 ? wf1 : wf0) + ((base + mm) % NSLOT) * BR_STAGE_BYTES + i * 2048);
                                    acc[i] = Lp<T>::mfma(wf, af, acc[i]);  // transposed: rows = channels
                                }
                            }
                        }
                    }
                    // epilogue (see the staged form below): skip chunks -> slice -> accumulator layout; x_new the other way
#pragma unroll
                    for (int hc = 0; hc < 2; ++hc) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(slice + (8 * c + (lane >> 3)) * SP + (lane & 7) * 16) = xch[nh][hc][c];
                        uint2 xv[8];
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int q = 0; q < 4; ++q) xv[4 * ii + q] = *reinterpret_cast<const uint2*>(slice + l31 * SP + (32 * ii + 8 * q + 4 * half) * 2);
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int i = 2 * hc + ii;
                                const int n = nh * 128 + 32 * i + 8 * q + 4 * half;
                                const f32x4 bb = *reinterpret_cast<const f32x4*>(bout_lds + n);   // bfc_ + bsc_
                                const uint2 xx = xv[4 * ii + q];
                                const float x0 = Lp<T>::to_f32((unsigned short)(xx.x & 0xffffu)), x1 = Lp<T>::to_f32((unsigned short)(xx.x >> 16));
                                const float x2 = Lp<T>::to_f32((unsigned short)(xx.y & 0xffffu)), x3 = Lp<T>::to_f32((unsigned short)(xx.y >> 16));
                                const float v0 = acc[i][4 * q + 0] + bb[0] + x0, v1 = acc[i][4 * q + 1] + bb[1] + x1;
                                const float v2 = acc[i][4 * q + 2] + bb[2] + x2, v3 = acc[i][4 * q + 3] + bb[3] + x3;
                                uint2 o;
                                o.x = (unsigned)Lp<T>::from_f32(v0) | ((unsigned)Lp<T>::from_f32(v1) << 16);
                                o.y = (unsigned)Lp<T>::from_f32(v2) | ((unsigned)Lp<T>::from_f32(v3) << 16);
                                *reinterpret_cast<uint2*>(slice + l31 * SP + (32 * ii + 8 * q + 4 * half) * 2) = o;
                            }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const long long m = m0 + wave * 32 + 8 * c + (lane >> 3);
                            const u32x4 v = *reinterpret_cast<const u32x4*>(slice + (8 * c + (lane >> 3)) * SP + (lane & 7) * 16);
                            hg_store16(reinterpret_cast<unsigned short*>(p.out) + (size_t)m * 256 + nh * 128 + 64 * hc + (lane & 7) * 8, v);
                        }
                    }
                }
                return;
            }
        }
#pragma unroll 1
        for (int nh = 0; nh < NPASS; ++nh) {
            f32x16 acc[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
            // bf16: the skip values of this pass are requested NOW as coalesced 16-byte chunks (lane owns chunk (lane & 7) of the
            // wave's pixel 8 c + (lane >> 3), per 64-channel half hc): their latency hides behind the K loop, and the epilogue
            // turns them into the accumulators' layout through a wave-private LDS slice
            u32x4 xch[EB == 2 ? 2 : 1][4];
            if constexpr (EB == 2) {
#pragma unroll
                for (int hc = 0; hc < 2; ++hc)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const long long m = m0 + wave * 32 + 8 * c + (lane >> 3);
                        u32x4 v = {0u, 0u, 0u, 0u};
                        if (m < p.M) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.x) + (size_t)m * 256 + nh * CH + 64 * hc + (lane & 7) * 8);
                        xch[hc][c] = v;
                    }
            }
            // float32: the skip values of this pass (4-byte pieces in the accumulators' layout: lane = channel, register = pixel) are requested
            // NOW too -- round 5; they used to be requested in the epilogue, their latency exposed four times per workgroup
            float xr4[EB == 4 ? NI : 1][16];
            if constexpr (EB == 4) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        xr4[i][r] = m < p.M ? reinterpret_cast<const float*>(p.x)[(size_t)m * 256 + nh * CH + i * 32 + l31] : 0.0f;
                    }
            }
            loadC(nh, 0);
            storeC(0);
            __syncthreads();
#pragma unroll
            for (int s = 0; s <= YSTEPS; ++s) {
                const unsigned char* const sw = stage + (s & 1) * C::STAGE_C;
                if (s + 1 <= YSTEPS) loadC(nh, s + 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (EB == 4) {
                    // y steps: KE = 32 channels = tile s; score step: the 32 score channels
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        if constexpr (std::is_same<T, F32S>::value) {
                            const XPair<T> ap = s < YSTEPS ? ysp[s < YSTEPS ? s : 0][q2] : scsp[q2];
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                const unsigned char* const wcp = sw + (i * 32 + l31) * PITCH + (4 * q2 + half) * 16;
                                mfma_pair<T, false>(*reinterpret_cast<const u32x4*>(wcp), *reinterpret_cast<const u32x4*>(wcp + 32), ap, acc[i]);
                            }
                        } else {
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                                for (int i = 0; i < NI; ++i) {
                                    const f32x4 wf = *reinterpret_cast<const f32x4*>(sw + (i * 32 + l31) * PITCH + (4 * q2 + 2 * jj + half) * 16);
                                    const f32x16& src = s < YSTEPS ? y[s < YSTEPS ? s : 0] : sc;
                                    mfma_quad<T>(src[8 * q2 + 4 * jj], src[8 * q2 + 4 * jj + 1], src[8 * q2 + 4 * jj + 2], src[8 * q2 + 4 * jj + 3], wf, acc[i]);
                                }
                        }
                    }
                } else {
                    // y steps: KE channels = tiles s*(KE/32) .. +KE/32-1; score step: one 32-channel group
#pragma unroll
                    for (int mm = 0; mm < KE / 32; ++mm) {
                        if (s == YSTEPS && mm >= 1) break;
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2) {
                            const u32x4 af = s < YSTEPS ? ypk[s < YSTEPS ? s * (KE / 32) + mm : 0][q2] : scpk[q2];
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                const u32x4 wf = *reinterpret_cast<const u32x4*>(sw + (i * 32 + l31) * PITCH + (mm * 32 + (2 * q2 + half) * 8) * 2);
                                acc[i] = Lp<T>::mfma(wf, af, acc[i]);  // transposed: rows = channels
                            }
                        }
                    }
                }
                if (s + 1 <= YSTEPS) storeC((s & 1) ^ 1);
                __syncthreads();
            }
            if constexpr (EB == 4) {
                // epilogue: D[row = pixel (r&3) + 8(r>>2) + 4*half of the wave][col = channel nh*128 + 32 i + l31]
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int n = nh * CH + i * 32 + l31;
                    const float bias = bout_lds[n];   // bfc_ + bsc_, staged once (the same float32 sum the two global loads gave)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (m < p.M) reinterpret_cast<float*>(p.out)[(size_t)m * 256 + n] = acc[i][r] + bias + xr4[i][r];
                    }
                }
            } else {
                // bf16 epilogue on the TRANSPOSED product: lane = pixel m0 + 32 wave + l31, registers 4q..4q+3 of tile i =
                // channels nh*128 + 32 i + 8 q + 4 half + {0..3}.  Global memory is touched in whole 128-byte pixel-half rows only
                // (16-byte chunks, eight lanes per row): x arrives as chunks, is parked in the wave's LDS slice and read back as the
                // 8-byte pieces the accumulator layout wants; x_new takes the same road in the other direction.  Same fp32
                // arithmetic and the same single rounding as the direct 8-byte accesses it replaces.
                constexpr int SP = 64 * 2 + 16;   // slice pitch: 64 channels + pad
                unsigned char* const slice = stage + 2 * C::STAGE_C + wave * (32 * SP);
                static_assert(2 * C::STAGE_C + 4 * 32 * SP <= C::STAGE_BYTES, "the epilogue slices sit behind the phase-C stages");
#pragma unroll
                for (int hc = 0; hc < 2; ++hc) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(slice + (8 * c + (lane >> 3)) * SP + (lane & 7) * 16) = xch[hc][c];
                    uint2 xv[8];
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int q = 0; q < 4; ++q) xv[4 * ii + q] = *reinterpret_cast<const uint2*>(slice + l31 * SP + (32 * ii + 8 * q + 4 * half) * 2);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int i = 2 * hc + ii;
                            const int n = nh * CH + 32 * i + 8 * q + 4 * half;
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(bout_lds + n);   // bfc_ + bsc_
                            const uint2 xx = xv[4 * ii + q];
                            const float x0 = Lp<T>::to_f32((unsigned short)(xx.x & 0xffffu)), x1 = Lp<T>::to_f32((unsigned short)(xx.x >> 16));
                            const float x2 = Lp<T>::to_f32((unsigned short)(xx.y & 0xffffu)), x3 = Lp<T>::to_f32((unsigned short)(xx.y >> 16));
                            const float v0 = acc[i][4 * q + 0] + bb[0] + x0, v1 = acc[i][4 * q + 1] + bb[1] + x1;
                            const float v2 = acc[i][4 * q + 2] + bb[2] + x2, v3 = acc[i][4 * q + 3] + bb[3] + x3;
                            uint2 o;
                            o.x = (unsigned)Lp<T>::from_f32(v0) | ((unsigned)Lp<T>::from_f32(v1) << 16);
                            o.y = (unsigned)Lp<T>::from_f32(v2) | ((unsigned)Lp<T>::from_f32(v3) << 16);
                            *reinterpret_cast<uint2*>(slice + l31 * SP + (32 * ii + 8 * q + 4 * half) * 2) = o;
                        }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const long long m = m0 + wave * 32 + 8 * c + (lane >> 3);
                        const u32x4 v = *reinterpret_cast<const u32x4*>(slice + (8 * c + (lane >> 3)) * SP + (lane & 7) * 16);
                        if (m < p.M) hg_store16(reinterpret_cast<unsigned short*>(p.out) + (size_t)m * 256 + nh * CH + 64 * hc + (lane & 7) * 8, v);
                    }
                }
            }
        }
    }
}

}  // namespace hgk
