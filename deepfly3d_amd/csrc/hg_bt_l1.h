// 16-bit (T = __hip_bfloat16 or _Float16; "bf16" below stands for either) fused pre-activation bottleneck 64 -> 64 -> 64 -> 128 with the 1x1 skip convolution (layer1 of the hourglass stem),
// ALL WEIGHTS RESIDENT IN LDS, persistent workgroups, 16 x 16 output tiles, optional "pooled output only".
//
// layer1 runs at half the image resolution (128 x 256 for the reference's 256 x 512 views): 29 M pixels per 896 views, only
// 115 kFLOP but 384 algorithmic bytes per pixel -- an HBM-bound block (tests/perf/ubench/README.md: a CU's fair share of HBM
// is ~10 bytes per cycle).  Its 57 k weights (W1 64x64, W2 9x64x64, W3 128x64, Wd 128x64 = 112 KB in bf16) fit the 160 KB
// LDS of a CU next to a 16 x 16 tile's t1 halo (324 px x 64 ch = 41 KB), so:
//
//   * a workgroup (EIGHT waves: the block needs few registers, and one wave sustains only one LDS fragment read per ~31
//     cycles -- two waves per SIMD keep the matrix pipe fed where a four-wave version was LDS-latency-bound) copies the
//     pre-swizzled weight image (bt_l1_pack_kernel) into LDS ONCE (LDS-DMA) and then walks a strided list of tiles of its
//     XCD's share -- no weight traffic, no counted waits, three barriers per tile;
//   * the x operand of phase 1 and of the skip convolution comes straight from global memory in MFMA operand layout
//     (lane = pixel, 16 bytes of its 128-byte channel vector per load) and is requested ONE TILE AHEAD (32 registers);
//   * the only consumer of layer1's output is the 2x2 max-pool in front of layer2: with out == nullptr the kernel writes the
//     pooled tensor only (the full-resolution tensor, 2/3 of the block's traffic, never exists).
//
// Same arithmetic as hg_kernels.h:bottleneck_kernel<bf16, 64, 64, true>: same K order in every accumulator (phase 1: 16-channel
// chunks ascending; phase 2: tap-major, chunks ascending; phase 3: bias b3 + bd, then W3 t2 in the host's K order, then Wd x),
// products formed transposed where that only swaps the MFMA operands -- bit-identical results.
//
// LDS: weights 114 688 B | t1 tile 41 472 B (128 B per halo pixel, 16-byte chunk c of pixel hp in slot c ^ ((hp >> 1) & 7):
// the 16 lanes of a fragment read hit 16 different slots of the 256-byte bank row) | coefficients | masks.  The epilogue's
// transposition slices (one per wave, 32 px x 64 channels at a time) reuse the t1 region.
#pragma once
#include "hg_bt_ring.h"

namespace hgk {

struct BtL1Args {
    const void* in;       // NHWC bf16 [V, H, W, 64]
    void* out;            // NHWC bf16 [V, H, W, 128] or nullptr (pooled output only)
    void* pool;           // NHWC bf16 [V, H/2, W/2, 128] or nullptr
    const void* wimage;   // L1_W_BYTES: the LDS image of the weights (bt_l1_pack_kernel)
    const float* b1;      // [64] (bn2 folded)
    const float* b2;      // [64] (bn3 folded)
    const float* b3;      // [128]
    const float* bd;      // [128] skip convolution bias
    const float* s1;      // [64] bn1 scale
    const float* t1;      // [64] bn1 shift
    int V, H, W;
};

constexpr int L1_TH = 16;
constexpr int L1_HALO = (L1_TH + 2) * BT_HW;        // 324
constexpr int L1_WAVES = 8;
constexpr int L1_RT = (L1_HALO + 31) / 32;          // 11 halo row tiles: wave w owns w and, w < 3, w + 8
constexpr int L1_W1_OFF = 0;                        // 2 K-slices x 64 rows x 64 B
constexpr int L1_W2_OFF = 8192;                     // 9 taps x 2 K-slices x 4096
constexpr int L1_W3_OFF = L1_W2_OFF + 9 * 2 * 4096; // 81 920: 2 K-slices x 128 rows x 64 B
constexpr int L1_WD_OFF = L1_W3_OFF + 2 * 8192;     // 98 304
constexpr int L1_W_BYTES = L1_WD_OFF + 2 * 8192;    // 114 688
constexpr int L1_T1_PITCH = 128;
constexpr int L1_T1_BYTES = L1_HALO * L1_T1_PITCH;  // 41 472
constexpr int L1_COEF_FLOATS = 64 * 4 + 128;        // bn1 scale | shift | b1 | b2 | b3 + bd
constexpr int L1_LDS_BYTES = L1_W_BYTES + L1_T1_BYTES + L1_COEF_FLOATS * 4 + 64;
constexpr int L1_OP = 64 * 2 + 16;                  // epilogue slice pitch (64 channels at a time)
static_assert(L1_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(L1_WAVES * 32 * L1_OP <= L1_T1_BYTES, "the epilogue slices live inside the t1 region");
static_assert(L1_W_BYTES % (L1_WAVES * 2048) == 0, "every wave copies whole 2 KB piece pairs");

// bf16 blob -> LDS image.  One thread per 16-byte chunk (8 K values of one row).
__global__ __launch_bounds__(256) void bt_l1_pack_kernel(const unsigned short* __restrict__ w1, const unsigned short* __restrict__ w2,
                                                         const unsigned short* __restrict__ w3, const unsigned short* __restrict__ wd,
                                                         unsigned char* __restrict__ image) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L1_W_BYTES / 16) return;
    const unsigned short* src;
    int off, r, c;
    if (idx < L1_W2_OFF / 16) {                      // W1 [64][64]
        const int sl = idx >> 8;
        r = (idx >> 2) & 63, c = idx & 3;
        src = w1 + (size_t)r * 64 + 32 * sl + 8 * c;
        off = L1_W1_OFF + sl * 4096;
    } else if (idx < L1_W3_OFF / 16) {               // W2 [9][64][64]
        const int q = idx - L1_W2_OFF / 16, ts = q >> 8, tap = ts >> 1, sl = ts & 1;
        r = (q >> 2) & 63, c = q & 3;
        src = w2 + ((size_t)tap * 64 + r) * 64 + 32 * sl + 8 * c;
        off = L1_W2_OFF + ts * 4096;
    } else {                                         // W3 [128][64] (K already permuted by the host packer), Wd [128][64]
        const int q = idx - L1_W3_OFF / 16, which = q >> 10, sl = (q >> 9) & 1;
        r = (q >> 2) & 127, c = q & 3;
        src = (which ? wd : w3) + (size_t)r * 64 + 32 * sl + 8 * c;
        off = (which ? L1_WD_OFF : L1_W3_OFF) + sl * 8192;
    }
    *reinterpret_cast<u32x4*>(image + off + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(src);
}

template <typename T>
__global__ __launch_bounds__(512, 1) void bottleneck_l1_kernel(BtL1Args p) {
    static_assert(sizeof(T) == 2, "16-bit storage formats only");
    constexpr int CIN = 64, CO = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const w_lds = smem;
    unsigned char* const t1_lds = smem + L1_W_BYTES;
    float* const coef_lds = reinterpret_cast<float*>(smem + L1_W_BYTES + L1_T1_BYTES);   // [0..63] s1 [64..127] t1 [128..191] b1 [192..255] b2 [256..383] b3 + bd
    unsigned long long* const valid_lds = reinterpret_cast<unsigned long long*>(smem + L1_W_BYTES + L1_T1_BYTES + L1_COEF_FLOATS * 4);
    const unsigned w_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)w_lds;

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DF3D_BT_TIMING
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31, px = l31 & 15;
    const bool two = wave + 8 < L1_RT;   // this wave owns a second halo row tile

    // ---- this workgroup's tiles: XCD x (workgroup b runs on XCD b % 8) owns the x-th contiguous eighth of the tiles, its
    // workgroups walk it with their count as the stride -> at any time an XCD works on neighbouring tiles
    const int tiles_x = p.W / BT_TW, tiles_y = p.H / L1_TH;
    int cnt, wx, cur;
    {
        const int ntiles = p.V * tiles_x * tiles_y, nwg = gridDim.x, xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        cnt = q + (xcd < r ? 1 : 0);
        wx = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
        cur = xcd * q + (xcd < r ? xcd : r);
    }
    int local = blockIdx.x >> 3;
    if (local >= cnt) return;
    cur += local;

    // ---- once per workgroup: the weight image (LDS-DMA, 14 KB per wave) and the coefficient vectors
#pragma unroll
    for (int i = 0; i < L1_W_BYTES / (L1_WAVES * 2048); ++i) {
        const unsigned piece = (unsigned)(wave * (L1_W_BYTES / (L1_WAVES * 2048)) + i) * 2048u;
        br_glds_stage(p.wimage, piece + (unsigned)lane * 16u, w_addr + piece);
    }
    if (tid < 64) {
        coef_lds[tid] = p.s1[tid];
        coef_lds[64 + tid] = p.t1[tid];
        coef_lds[128 + tid] = p.b1[tid];
        coef_lds[192 + tid] = p.b2[tid];
    }
    if (tid < 128) coef_lds[256 + tid] = p.b3[tid] + p.bd[tid];

    // ---- x operand: lane (l31, half) of halo row tile rt <-> halo pixel 32 rt + l31, 16-byte chunks 2 kc + half
    auto tile_origin = [&](int tile, int& tx0, int& ty0, int& view) {
        tx0 = (tile % tiles_x) * BT_TW;
        const int t2_ = tile / tiles_x;
        ty0 = (t2_ % tiles_y) * L1_TH;
        view = t2_ / tiles_y;
    };
    u32x4 rx[2][4];
    auto loadx = [&](int tile) {   // out-of-image (and past-the-halo) pixels read pixel (0, 0); the t1 epilogue zeroes them
        int tx0, ty0, view;
        tile_origin(tile, tx0, ty0, view);
        const unsigned char* const xin = reinterpret_cast<const unsigned char*>(p.in) + (size_t)view * p.H * p.W * CIN * 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int hp = (wave + 8 * k) * 32 + l31;
            const int hy = hp / BT_HW, hx = hp % BT_HW;
            const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
            const bool ok = hp < L1_HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            const unsigned char* const src = xin + (((size_t)(ok ? y : 0) * p.W + (ok ? x : 0)) * CIN + half * 8) * 2;
            if (k == 0 || two) {
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) rx[k][kc] = *reinterpret_cast<const u32x4*>(src + kc * 32);
            }
        }
    };
    loadx(cur);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the weight image have landed (and its first x operand)
    br_barrier();

    // fragment bases inside the weight image: rows 32 m + l31, chunk 2 (kc & 1) + half of K-slice kc >> 1
    const unsigned char* const wfe = w_lds + br_swz(l31, half);       // even K chunks
    const unsigned char* const wfo = w_lds + br_swz(l31, 2 + half);   // odd K chunks
    auto wfrag = [&](int base, int slice_bytes, int kc, int m) -> const unsigned char* {
        return ((kc & 1) ? wfo : wfe) + base + (kc >> 1) * slice_bytes + m * 2048;
    };

    for (;;) {
        int tx0, ty0, view;
        tile_origin(cur, tx0, ty0, view);
        const bool has_next = local + wx < cnt;
        const int nxt = has_next ? cur + wx : cur;
        // opaque per-iteration copies of the lane coordinates (hipcc would otherwise hoist the epilogues' address arithmetic
        // out of the tile loop and keep its registers live throughout)
        int l31v = l31, halfv = half, lanev = lane, tidv = tid;
        asm volatile("" : "+v"(l31v), "+v"(halfv), "+v"(lanev), "+v"(tidv));
        const unsigned char* const xin = reinterpret_cast<const unsigned char*>(p.in) + (size_t)view * p.H * p.W * CIN * 2;

        // halo validity masks (read in the t1 epilogue, behind the next barrier)
        if (tidv < 12 * 32) {
            const int hy = tidv / BT_HW, hx = tidv % BT_HW;
            const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
            const bool ok = tidv < L1_HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            const unsigned long long m = __ballot(ok);
            if (lanev == 0) valid_lds[tidv >> 6] = m;
        }
        BR_STAMP(0);

        // ---- phase 1: t1^T = relu(W1' relu(bn1 x)^T + b1') on the halo (transposed: D[channel][halo pixel]) ------------
        {
            f32x16 acc[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int t = 0; t < 4; ++t) {   // register 4 t + e of channel tile ct <-> channel 32 ct + 8 t + 4 half + e
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(coef_lds + 128 + ct * 32 + 8 * t + 4 * half);
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[k][ct][4 * t + e] = bb[e];
                }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                PreactCoef<T> coef;
                const int c0 = kc * 16 + half * 8;
                coef.s[0] = *reinterpret_cast<const f32x4*>(coef_lds + c0);
                coef.s[1] = *reinterpret_cast<const f32x4*>(coef_lds + c0 + 4);
                coef.t[0] = *reinterpret_cast<const f32x4*>(coef_lds + 64 + c0);
                coef.t[1] = *reinterpret_cast<const f32x4*>(coef_lds + 64 + c0 + 4);
                u32x4 wfr[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) wfr[ct] = *reinterpret_cast<const u32x4*>(wfrag(L1_W1_OFF, 4096, kc, ct));
                {
                    const u32x4 xa = br_preact<T>(rx[0][kc], coef);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) mfma_chunk<T>(wfr[ct], xa, acc[0][ct]);
                }
                if (two) {
                    const u32x4 xa = br_preact<T>(rx[1][kc], coef);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) mfma_chunk<T>(wfr[ct], xa, acc[1][ct]);
                }
            }
            BR_STAMP(1);
            br_barrier();   // B3: every wave is past the previous tile's epilogue slices (they share the t1 region); masks visible
            // epilogue: ReLU (the bias was the start value), zero outside the image, 8-byte stores into the swizzled t1 tile
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k == 1 && !two) break;
                const int rt = wave + 8 * k;
                const int hp = rt * 32 + l31v;
                const unsigned keep = 0u - (unsigned)((valid_lds[rt >> 1] >> ((rt & 1) * 32 + l31v)) & 1ull);
                unsigned char* const trow = t1_lds + hp * L1_T1_PITCH + halfv * 8;
                const int sw = (hp >> 1) & 7;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        uint2 w;
                        w.x = br_relu_pk(Lp<T>::pack2(acc[k][ct][4 * t + 0], acc[k][ct][4 * t + 1])) & keep;
                        w.y = br_relu_pk(Lp<T>::pack2(acc[k][ct][4 * t + 2], acc[k][ct][4 * t + 3])) & keep;
                        if (hp < L1_HALO) *reinterpret_cast<uint2*>(trow + (((ct * 4 + t) ^ sw) << 4)) = w;
                    }
            }
        }
        BR_STAMP(2);
        br_barrier();   // B1: the t1 tile is complete
        BR_STAMP(3);

        // the next tile's x operand (one tile ahead) and this tile's centre pixels for the skip convolution: lane (l31, half)
        // <-> tile pixel (2 wave + (l31 >> 4), l31 & 15), chunks 2 kc + half
        loadx(nxt);
        u32x4 xc[4];
        {
            const unsigned char* const src = xin + (((size_t)(ty0 + 2 * wave + (l31 >> 4)) * p.W + (tx0 + px)) * CIN + half * 8) * 2;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) xc[kc] = *reinterpret_cast<const u32x4*>(src + kc * 32);
        }

        // ---- phase 2: t2^T = W2' (*) t1 (A = W2 tap rows, B = t1 at the tap-shifted pixel); this wave = pixel tile `wave` ----
        u32x4 t2f[2][2];   // relu(t2) as bf16 [channel tile m][register half q2]: the B operands of phase 3
        {
            f32x16 t2[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {   // channel of register r in tile m: 32 m + (r & 3) + 8 (r >> 2) + 4 half
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(coef_lds + 192 + 32 * m + 8 * q + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t2[m][4 * q + e] = bb[e];
                }
            const int hp0 = (2 * wave + (l31 >> 4)) * BT_HW + px;
            // 36 (tap, K chunk) groups, the fragments of group g + 2 requested before the MFMAs of group g
            u32x4 tfr[3], wfr[3][2];
            auto load_group = [&](int g, int buf) {
                const int tap = g >> 2, kc = g & 3;
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int hp = hp0 + ky * BT_HW + kx;
                tfr[buf] = *reinterpret_cast<const u32x4*>(t1_lds + hp * L1_T1_PITCH + (((2 * kc + half) ^ ((hp >> 1) & 7)) << 4));
#pragma unroll
                for (int m = 0; m < 2; ++m) wfr[buf][m] = *reinterpret_cast<const u32x4*>(wfrag(L1_W2_OFF + tap * 8192, 4096, kc, m));
            };
            load_group(0, 0);
            load_group(1, 1);
#pragma unroll
            for (int g = 0; g < 36; ++g) {
                if (g + 2 < 36) load_group(g + 2, (g + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m) mfma_chunk<T>(wfr[g % 3][m], tfr[g % 3], t2[m]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        t2f[m][q2][e] = br_relu_pk(Lp<T>::pack2(t2[m][8 * q2 + 2 * e], t2[m][8 * q2 + 2 * e + 1]));
        }
        BR_STAMP(4);
        br_barrier();   // B2: every wave is done with the t1 tile -> the epilogue slices may overwrite it
        BR_STAMP(5);

        // ---- phase 3: out^T = W3 t2^T + Wd x^T + (b3 + bd) ------------------------------------------------------------------
        // accumulator register 4 t + e of channel tile i holds, for pixel l31 of the pixel tile, output channel 32 i + 8 t + 4 half + e
        {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(coef_lds + 256 + i * 32 + 8 * t + 4 * halfv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][4 * t + e] = bb[e];
                }
            // W3: t2 tile m, registers 8 q2 .. 8 q2 + 7 <-> packed W3 K positions 32 m + 16 q2 + 8 half .. (host K order): K chunk 2 m + q2
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                u32x4 wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const u32x4*>(wfrag(L1_W3_OFF, 8192, kc, i));
                const u32x4 tf = t2f[kc >> 1][kc & 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = Lp<T>::mfma(wf[i], tf, acc[i]);
            }
            // skip convolution on the raw input
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                u32x4 wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const u32x4*>(wfrag(L1_WD_OFF, 8192, kc, i));
#pragma unroll
                for (int i = 0; i < 4; ++i) mfma_chunk<T>(wf[i], xc[kc], acc[i]);
            }
            // epilogue through LDS, 64 channels at a time: the wave parks 32 px x 64 ch in its slice (8-byte stores) and reads
            // it back as 16-byte chunks -- rows coalesced for the store, and both pooling partners within reach
            unsigned char* const slice = t1_lds + wave * (32 * L1_OP);
            unsigned short* const outs = p.out ? reinterpret_cast<unsigned short*>(p.out) + (size_t)view * p.H * p.W * CO : nullptr;
            unsigned short* const pp = p.pool ? reinterpret_cast<unsigned short*>(p.pool) + (size_t)view * (p.H / 2) * (p.W / 2) * CO : nullptr;
#pragma unroll
            for (int hc = 0; hc < 2; ++hc) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i = 2 * hc + ii;
                        uint2 w;
                        w.x = Lp<T>::pack2(acc[i][4 * t + 0], acc[i][4 * t + 1]);
                        w.y = Lp<T>::pack2(acc[i][4 * t + 2], acc[i][4 * t + 3]);
                        *reinterpret_cast<uint2*>(slice + l31v * L1_OP + (ii * 32 + 8 * t + 4 * halfv) * 2) = w;
                    }
                u32x4 fin[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {   // lane owns chunk (lane & 7) of the pixel tile's pixel 8 c + (lane >> 3)
                    const int pw = 8 * c + (lanev >> 3);
                    fin[c] = *reinterpret_cast<const u32x4*>(slice + pw * L1_OP + (lanev & 7) * 16);
                    if (outs)
                        hg_store16(outs + ((size_t)(ty0 + 2 * wave + (pw >> 4)) * p.W + (tx0 + (pw & 15))) * CO + 64 * hc + (lanev & 7) * 8, fin[c]);
                }
                if (pp) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {   // pixels 8 c + (lane >> 3) of tile row 0 and the one below; the column partner sits 8 lanes away
                        // in the ordered-integer domain (hg_kernels.h: bf16x2_key); lane ^ 8 = a rotation by 8 inside the row of 16: one DPP move
                        u32x4 m = bf16_key_max_chunk(bf16_key_chunk(fin[c]), bf16_key_chunk(fin[c + 2]));
                        u32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m[e], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
                        m = bf16_key_chunk(bf16_key_max_chunk(m, o));
                        if (((lanev >> 3) & 1) == 0)
                            hg_store16(pp + ((size_t)(ty0 / 2 + wave) * (p.W / 2) + (tx0 / 2 + 4 * c + (lanev >> 4))) * CO + 64 * hc + (lanev & 7) * 8, m);
                    }
                }
            }
        }
        BR_STAMP(6);
        if (!has_next) break;
        local += wx;
        cur = nxt;
    }
}

}  // namespace hgk
