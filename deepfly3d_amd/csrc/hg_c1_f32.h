// fp32: the pre-activation bottleneck 256 -> 128 -> 128 -> 256 as TWO launches ("split" form of hg_bt_ring_f32.h):
//
//   conv1_ring_f32_kernel      t1 = relu(W1' relu(bn1 x) + b1')  for EVERY pixel of the level, once      -> HBM [px][128] f32
//   bottleneck_tail_f32_kernel the 3x3, the last 1x1 and the skip on 8 x 16 tiles; the t1 halo tile arrives by LDS-DMA
//
// Why: the fused kernel is matrix-pipe bound (0.89 busy) and computes conv1 on the 10 x 18 halo of every 8 x 16 tile:
// 192 MFMA rows for 128 pixels, 7.7 % of all its MFMAs are recomputation.  conv1 is pixel-wise, so computing it once per
// pixel and reading it back with the halo trades those MFMAs for HBM bytes the fp32 path has to spare (it runs at 1.4 of
// 8 TB/s): +0.5 KB/px written, the tail reads 0.72 KB/px of t1 instead of 1.4 KB/px of x.  MFMA work 27.3 instead of
// 29.4 MMAC per tile.  Same MFMA K order in every accumulator as the fused kernel (the bias is the start value, K
// ascending), so t1 -- and with it the block's output -- is BIT-IDENTICAL to the fused form.
//
// conv1: a workgroup owns 128 consecutive pixels (the op is pixel-wise: no tiles, no halo), wave w the pixel row tiles
// 2 (w >> 1), +1 and the channel tiles 2 (w & 1), +1: 64 accumulators, four LDS fragment reads per four MFMA chunks.
// W1' streams through the 4-slot LDS-DMA ring as 16 stages (K slice of 16 floats x 128 rows); x is staged global -> registers ->
// bn1 + ReLU -> LDS three K steps ahead, exactly like phase 1 of the fused kernel.
#pragma once
#include "hg_bt_ring_f32.h"

namespace hgk {

struct Conv1Args {
    const void* in;       // NHWC f32 [M, 256]
    const void* in2;      // UP: NHWC f32 [V, H/2, W/2, 256]: the block's input is in + nearest-upsample(in2) (one fp32 add, as upadd_kernel)
    int H, W;             // UP: the level's size (M = V * H * W)
    void* t1;             // [M, 128] f32: relu(W1' relu(bn1 x) + b1')
    const void* wstream;  // C1_NSTAGE x BR_STAGE_BYTES (bt_c1_pack_f32_kernel)
    const float* b1;      // [128] (bn2 folded)
    const float* s1;      // [256] bn1 scale
    const float* t1c;     // [256] bn1 shift
    long long M;          // pixels
};

constexpr int C1_NSTAGE = 16;
#ifndef C1_ABLM
#define C1_ABLM 0   // development builds: ablation mask (1 no MFMAs, 2 no weight DMA, 4 no x loads, 8 no t1 stores)
#endif
constexpr int C1_XPITCH = 64;                                                 // unpadded, chunks swizzled (br_xslot)
constexpr int C1_XSTAGE = 128 * C1_XPITCH;                                   // 8 192
constexpr int C1_XSLOTS = 2;   // x ring: step s is staged into slot s % 2 while slot (s - 1) % 2 may still be read; the slot was last read in
                               // step s - 2, which every wave has left before anyone passes the barrier of step s - 1
constexpr int C1_LDS_BYTES = BR_RING_BYTES + C1_XSLOTS * C1_XSTAGE + 512 * 4 + 128 * 4;   // ring | x ring | bn1 scale, shift | b1: 50.5 KB

// fp32 blob -> conv1 weight stream: stage s = K slice [16 s, 16 s + 16) of the cout (128, or layer1's 64: the upper half of the
// stage image stays unused) rows of W1' [cout][cin]
__global__ __launch_bounds__(256) void bt_c1_pack_f32_kernel(const float* __restrict__ w1, unsigned char* __restrict__ stream, int cin = 256, int cout = 128) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (cin / 16) * 512) return;
    const int s = idx >> 9, r = (idx >> 2) & 127, c = idx & 3;
    if (r >= cout) return;
    *reinterpret_cast<u32x4*>(stream + (size_t)s * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(w1 + (size_t)r * cin + 16 * s + 4 * c);
}

// PERSISTENT: a workgroup walks pixel tiles tile, tile + gridDim.x, ... with the pipeline running across tile boundaries -- the
// coefficients are staged once, stage (s + 3) % 16 and the x loads of step s + DX belong to the NEXT tile during a tile's last
// steps (the weights are the same for every tile), so no tile after the first has a prologue.  Every tile issues exactly the same
// vector-memory operations in the same order (the last tile re-reads its own rows and re-requests stages it will not use), which
// keeps the counted waits valid; the kernel drains the queue before it ends.
// CIN / COUT: 256 -> 128 (the identity-skip bottlenecks) or 64 -> 64 (layer1: four K steps, one channel tile per wave, and only
// the lower half of every stage image travels)
template <bool UP, int CIN = 256, int COUT = 128, typename T = float>   // T: float or F32S (split products)
__global__ __launch_bounds__(256, 2) void conv1_ring_f32_kernel(Conv1Args p) {
    static_assert((CIN == 256 && COUT == 128) || (CIN == 128 && COUT == 128 && !UP) || (CIN == 64 && COUT == 64 && !UP), "instantiated shapes");
    constexpr int NST = CIN / 16;       // K steps = weight stages
    constexpr int NCT = COUT / 64;      // channel tiles per wave   // (three workgroups per CU fit -- 50.5 KB -- and measured no faster)
    // Also measured, same box, none of them faster for the plain form (1 043 us per average launch; matrix pipe 0.83 busy at 2.30 GHz,
    // the lowest clock of the fp32 kernels): x requested 2 / 4 / 8 K steps ahead (1 057-1 063 us: not latency); the step's eight
    // fragments requested right behind the barrier, the previous step's second half multiplied first and the next step's x staged
    // under the MFMAs (1 051-1 057; UP form 4 410 -> 4 200); transposed accumulators with sixteen 16-byte stores per lane
    // instead of 64 four-byte ones (1 071).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const xr = smem + BR_RING_BYTES;
    float* const coef_lds = reinterpret_cast<float*>(smem + BR_RING_BYTES + C1_XSLOTS * C1_XSTAGE);   // [0..255] scale, [256..511] shift, [512..639] b1
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const long long ntiles = (p.M + 127) / 128;
    long long tile = blockIdx.x;
    if (tile >= ntiles) return;

    const unsigned wvoff = (unsigned)wave * 2048u + (unsigned)lane * 16u;
    auto ring_issue = [&](int s) {   // stage s (the same for every tile) -> ring slot s % 4
        if (C1_ABLM & 2) return;
        if constexpr (COUT == 128)
            br_glds_stage(reinterpret_cast<const unsigned char*>(p.wstream) + (size_t)s * BR_STAGE_BYTES, wvoff,
                          ring_addr + (unsigned)(s % BR_RING) * BR_STAGE_BYTES + (unsigned)wave * 2048);
        else   // 64 rows = the stage image's lower 4 KB: one 1 KB piece per wave
            br_glds_piece(reinterpret_cast<const unsigned char*>(p.wstream) + (size_t)s * BR_STAGE_BYTES, (unsigned)wave * 1024u + (unsigned)lane * 16u,
                          ring_addr + (unsigned)(s % BR_RING) * BR_STAGE_BYTES + (unsigned)wave * 1024);
    };
    constexpr int NPC = COUT == 128 ? 2 : 1;   // DMA pieces per wave and stage
    const unsigned char* const wf0 = ring + br_swz(l31, half);
    const unsigned char* const wf1 = ring + br_swz(l31, 2 + half);

    if (tid < CIN) {
        coef_lds[tid] = p.s1[tid];
        coef_lds[256 + tid] = p.t1c[tid];
    }
    if (tid < COUT) coef_lds[512 + tid] = p.b1[tid];
    ring_issue(0);
    ring_issue(1);
    ring_issue(2);

    // x staging: thread -> (row = (tid >> 2) + 64 i, 16-byte chunk = tid & 3) of a 16-float K step; rows past the end read the last pixel
    constexpr int XP = 2, DX = UP ? 2 : 4, LX = UP ? 2 * XP : XP;   // DX divides 16: the register slots repeat from tile to tile
    static_assert(NST % DX == 0 && NST % C1_XSLOTS == 0 && NST % BR_RING == 0, "slot patterns repeat per tile");
    const int xchunk = tid & 3;
    const unsigned char* xp[XP];            // this tile's rows
    const unsigned char* xq[UP ? XP : 1];
    const unsigned char* xn[XP];            // the next tile's rows
    const unsigned char* xqn[UP ? XP : 1];
    auto rows_of = [&](long long t, const unsigned char* (&px)[XP], const unsigned char* (&pq)[UP ? XP : 1]) {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            long long m = t * 128 + (tid >> 2) + 64 * i;
            if (m >= p.M) m = p.M - 1;
            px[i] = reinterpret_cast<const unsigned char*>(p.in) + ((size_t)m * CIN + xchunk * 4) * 4;
            if constexpr (UP) {
                const long long hw = (long long)p.H * p.W, view = m / hw;
                const int pix = (int)(m - view * hw), y = pix / p.W, x = pix - y * p.W;
                pq[i] = reinterpret_cast<const unsigned char*>(p.in2) + ((((size_t)view * (p.H / 2) + (y >> 1)) * (p.W / 2) + (x >> 1)) * 256 + xchunk * 4) * 4;
            }
        }
    };
    rows_of(tile, xp, xq);
    u32x4 rx[DX][XP];
    u32x4 rb[UP ? DX : 1][XP];
    auto loadx = [&](bool next_tile, int s, int slot) {
#pragma unroll
        for (int i = 0; i < XP; ++i) rx[slot][i] = (C1_ABLM & 4) ? u32x4{(unsigned)s, (unsigned)tid, 0u, 0u} : *reinterpret_cast<const u32x4*>((next_tile ? xn[i] : xp[i]) + s * 64);
        if constexpr (UP) {
#pragma unroll
            for (int i = 0; i < XP; ++i) rb[slot][i] = *reinterpret_cast<const u32x4*>((next_tile ? xqn[i] : xq[i]) + s * 64);
        }
    };
    auto storex = [&](int s, int slot) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(coef_lds + s * 16 + xchunk * 4);
        const f32x4 ct_ = *reinterpret_cast<const f32x4*>(coef_lds + 256 + s * 16 + xchunk * 4);
        unsigned char* const sx = xr + (s % C1_XSLOTS) * C1_XSTAGE;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            f32x4 v = __builtin_bit_cast(f32x4, rx[slot][i]);
            if constexpr (UP) v += __builtin_bit_cast(f32x4, rb[slot][i]);   // x = in + upsample(in2), what upadd_kernel would have stored
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(br_relu(fmaf(v[e], cs[e], ct_[e])));
            *reinterpret_cast<u32x4*>(sx + ((tid >> 2) + 64 * i) * C1_XPITCH + br_xslot(tid >> 2, xchunk)) = o;
        }
    };
#pragma unroll
    for (int k = 0; k < DX; ++k) loadx(false, k, k);

    const int rt0 = 2 * (wave >> 1), ct0 = NCT * (wave & 1);
    float* const out = reinterpret_cast<float*>(p.t1);
    br_barrier();   // coefficients visible
    bool first = true;
    for (;;) {
        const long long nxt = tile + gridDim.x < ntiles ? tile + gridDim.x : tile;   // (last tile: its own rows again, see above)
        rows_of(nxt, xn, xqn);
        f32x16 acc[2][NCT];   // [pixel row tile][channel tile]; register r <-> pixel (r & 3) + 8 (r >> 2) + 4 half, lane l31 <-> channel
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const float bias = coef_lds[512 + (ct0 + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = bias;
        }
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            storex(s, s % DX);
            // Operations issued after stage s's two DMA pieces.  Steady state: the x loads of the step that requested it (LX), then
            // two more steps' pieces and loads: 4 + 3 LX.  First tile: the prologue's three stages and DX x steps come first.  Later
            // tiles, s < 3: the previous tile's 64 (COUT = 64: 32) stores sit in between as well -- 64 are more than the counter's six bits hold: wait for 63.
            const int steady = 2 * NPC + 3 * LX;
            const int n_first = s == 0 ? 2 * NPC + DX * LX : s == 1 ? 2 * NPC + DX * LX + LX : s == 2 ? 2 * NPC + DX * LX + 2 * LX : steady;
            const int n_later = s < 3 ? (steady + 32 * NCT < 63 ? steady + 32 * NCT : 63) : steady;   // + the previous tile's stores (a smaller count is safe, a larger one is not)
            if (n_first == n_later) {
                br_wait_vm(n_first);
            } else if (first) {
                br_wait_vm(n_first);
            } else {
                br_wait_vm(n_later);
            }
            br_barrier();
            ring_issue((s + 3) % NST);
            if (s + DX < NST) loadx(false, s + DX, s % DX);
            else loadx(true, s + DX - NST, s % DX);
            const unsigned char* const sx = xr + (s % C1_XSLOTS) * C1_XSTAGE;
            if constexpr (std::is_same<T, F32S>::value) {   // the K step as a whole: two split pixel-row pairs against NCT weight fragment pairs
                XPair<T> xp2[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned char* const row = sx + ((rt0 + i) * 32 + l31) * C1_XPITCH;
                    xp2[i] = make_xpair<T>(*reinterpret_cast<const u32x4*>(row + br_xslot(l31, half)), *reinterpret_cast<const u32x4*>(row + br_xslot(l31, 2 + half)));
                }
#pragma unroll
                for (int j = 0; j < NCT; ++j) {
                    const u32x4 w0 = *reinterpret_cast<const u32x4*>(wf0 + (s % BR_RING) * BR_STAGE_BYTES + (ct0 + j) * 2048);
                    const u32x4 w1 = *reinterpret_cast<const u32x4*>(wf1 + (s % BR_RING) * BR_STAGE_BYTES + (ct0 + j) * 2048);
#pragma unroll
                    for (int i = 0; i < 2; ++i) mfma_pair<T, false>(w0, w1, xp2[i], acc[i][j]);
                }
            } else {
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {   // the K step's two 8-float halves
                    u32x4 wf[NCT], xf[2];
#pragma unroll
                    for (int j = 0; j < NCT; ++j) wf[j] = *reinterpret_cast<const u32x4*>((j2 ? wf1 : wf0) + (s % BR_RING) * BR_STAGE_BYTES + (ct0 + j) * 2048);
#pragma unroll
                    for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const u32x4*>(sx + ((rt0 + i) * 32 + l31) * C1_XPITCH + br_xslot(l31, 2 * j2 + half));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NCT; ++j) {
                            if (C1_ABLM & 1) asm volatile("" ::"v"(xf[i]), "v"(wf[j]));
                            else mfma_chunk<T>(xf[i], wf[j], acc[i][j]);
                        }
                }
            }
        }
        // epilogue: ReLU, 4-byte stores of 128 contiguous bytes per (pixel, channel tile): 64 per lane, unconditional (M is a multiple
        // of 128: the launcher checks it -- every level the fused kernels take has whole 8 x 16 tiles)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = tile * 128 + (rt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                for (int j = 0; j < NCT; ++j) {
                    if ((C1_ABLM & 8) && acc[i][j][r] != 12345.678f) continue;
                    out[(size_t)m * COUT + (ct0 + j) * 32 + l31] = br_relu(acc[i][j][r]);
                }
            }
        if (nxt == tile) break;
        tile = nxt;
        first = false;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            xp[i] = xn[i];
            if constexpr (UP) xq[i] = xqn[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-requested stages and rows of the last tile have landed: nothing of this
                                                        // workgroup writes LDS after it ends
}

}  // namespace hgk
