// fp32 conv1 of the identity bottlenecks (t1 = relu(W1' relu(bn1 x) + b1'), 256 -> 128, every pixel once) with the WHOLE weight matrix resident in
// LDS: the round-6 sibling of conv1_ring_f32_kernel (hg_c1_f32.h), built by the rules hg_bt_wino_f32.h measured for the exact-fp32 MFMA.
//
// conv1_ring_f32_kernel streams W1' through a 4-slot LDS ring and stages x global -> registers -> bn1 + ReLU -> LDS, sixteen barriers per 128-pixel
// tile, two workgroups per CU: matrix pipe 0.83 busy, and nothing tried on it in rounds 3-5 moved that.  W1' is 128 x 256 floats = 128 KB: it FITS
// the CU's 160 KB.  So here:
//   * one workgroup per CU, persistent; the prologue copies the sixteen 8 KB stage images by LDS-DMA, once; after its barrier the four waves never
//     synchronise again: a wave owns 32 consecutive pixels x all 128 output channels (64 accumulators) and walks tiles wave, wave + 4 G, ...;
//   * x goes global -> registers -> bn1 + ReLU in place -> MFMA A operand (lane (l31, half): pixel l31, channels 8 q + 4 half .. + 3 per 16-byte load:
//     the K pairs of four MFMAs).  Loads are inline assembly (scalar base + lane offset, issued where they stand) with counted waits: half a tile
//     (16 loads, 64 registers) is in flight while the other half is multiplied -- 16 000 MFMA cycles ahead of its use;
//   * weight fragments come from LDS between the MFMAs (ds_read_b128: all but free there); bn1 + ReLU of four loads is ONE clump of VALU work;
//   * W1's rows are permuted (row 32 i + l <-> channel 4 l + i, as bt_wino_pack_w3_kernel does for W3): a lane's four accumulator tiles are four
//     consecutive channels of one pixel -- sixteen 16-byte stores per lane and tile.
// Arithmetic: the bias is the accumulators' start value and K ascends exactly as in conv1_ring_f32_kernel (stage s, half j2, e): BIT-IDENTICAL t1.
#pragma once
#include "hg_bt_wino_f32.h"
#include "hg_c1_f32.h"

namespace hgk {

#ifndef C1R_ABL
#define C1R_ABL 0   // development builds: 1 no bn1 + ReLU clumps, 2 no x loads (and no waits for them), 4 no stores, 8 no MFMAs
#endif

constexpr int C1R_W_BYTES = C1_NSTAGE * BR_STAGE_BYTES;                 // 131 072
constexpr int C1R_LDS_BYTES = C1R_W_BYTES + 512 * 4 + 128 * 4;         // weights | bn1 scale, shift | b1
static_assert(C1R_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// W1' [128 cout][256 cin] -> sixteen stage images (16-float K slice s; 128 rows x 64 bytes, br_swz), row 32 i + l = channel 4 l + i
__global__ __launch_bounds__(256) void c1r_pack_kernel(const float* __restrict__ w1, unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C1_NSTAGE * 512) return;
    const int s = idx >> 9, rem = idx & 511, c = rem & 3, r = rem >> 2;
    const int ch = 4 * (r & 31) + (r >> 5);
    *reinterpret_cast<u32x4*>(stream + (size_t)s * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(w1 + (size_t)ch * 256 + 16 * s + 4 * c);
}

// Conv1Args: in [M, 256], t1 [M, 128], wstream = c1r_pack_kernel's images, b1 [128], s1 / t1c [256]; M % 32 == 0.  (in2 / H / W unused: the plain form.)
__global__ __launch_bounds__(256, 1) void conv1_res_f32_kernel(Conv1Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const coef_lds = reinterpret_cast<float*>(smem + C1R_W_BYTES);   // [0..255] scale, [256..511] shift, [512..639] b1
    const unsigned w_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    {   // prologue: the sixteen stages (this wave: pieces 2 wave, 2 wave + 1 of each), the coefficients
        const unsigned wvoff = (unsigned)wave * 2048u + (unsigned)lane * 16u;
#pragma unroll
        for (int k = 0; k < C1_NSTAGE; ++k)
            br_glds_stage(reinterpret_cast<const unsigned char*>(p.wstream) + (size_t)k * BR_STAGE_BYTES, wvoff, w_addr + (unsigned)(k * BR_STAGE_BYTES + wave * 2048));
        coef_lds[tid] = p.s1[tid];
        coef_lds[256 + tid] = p.t1c[tid];
        if (tid < 128) coef_lds[512 + tid] = p.b1[tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        br_barrier();
    }
    const long long ntiles = p.M / 32, stride = 4LL * gridDim.x;
    long long t = 4LL * blockIdx.x + wave;
    if (t >= ntiles) return;   // (behind the only barrier)

    const unsigned char* const wf0 = smem + br_swz(l31, half);
    const unsigned char* const wf1 = smem + br_swz(l31, 2 + half);
    const unsigned xlane = (unsigned)((l31 * 256 + 4 * half) * 4);            // pixel l31 of the tile, channels 4 half ..
    const unsigned olane = (unsigned)(((4 * half) * 128 + 4 * l31) * 4);      // pixel 4 half of a register's group, channels 4 l31 ..
    const float* const cs_l = coef_lds + 4 * half;                            // this lane's coefficients of load q: + 8 q
    const unsigned char* const in = reinterpret_cast<const unsigned char*>(p.in);
    unsigned char* const out = reinterpret_cast<unsigned char*>(p.t1);

    // x of half a tile: [statement m][load j]: load q = 16 h + 4 m + j = channels 8 q + 4 half .. + 3 of the lane's pixel (statement m: 128 contiguous bytes per pixel)
    f32x4 xa[4][4], xb[4][4];
    auto issue_half = [&](f32x4 (&x)[4][4], long long tile, int h) {
        const unsigned char* const sb = in + (size_t)tile * (32 * 1024) + h * 512;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (C1R_ABL & 2) asm volatile("" : "+v"(x[m][0]), "+v"(x[m][1]), "+v"(x[m][2]), "+v"(x[m][3]));
            else wn_xload4s<32>(x[m], sb + m * 128, xlane);
        }
    };
    auto bn_relu = [&](f32x4 (&x)[4], int q0) {   // one clump: 16 fused multiply-adds, 16 maxima
        if (C1R_ABL & 1) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 cs = *reinterpret_cast<const f32x4*>(cs_l + 8 * (q0 + j));
            const f32x4 ct = *reinterpret_cast<const f32x4*>(cs_l + 256 + 8 * (q0 + j));
#pragma unroll
            for (int e = 0; e < 4; ++e) x[j][e] = br_relu(fmaf(x[j][e], cs[e], ct[e]));
        }
    };
    f32x16 o[4];   // tile i, register r: channel 4 l31 + i of pixel (r & 3) + 8 (r >> 2) + 4 half
    // A tile is 128 groups of four MFMAs: group n = (load q = n >> 2, accumulator tile i = n & 3) needs ONE 16-byte weight fragment; group n + 1's is
    // requested before group n's MFMAs are issued (two register sets; the tile's last group requests the next tile's first).  Measured and not
    // kept: the sixteen MFMAs of a load rotating over the four accumulators (four in a row on one accumulator are no slower).
    f32x4 wfq[2];
    auto wf_of = [&](int n) {
        const int q = (n >> 2) & 31, i = n & 3;
        return *reinterpret_cast<const f32x4*>(((q & 1) ? wf1 : wf0) + (q >> 1) * BR_STAGE_BYTES + i * 2048);
    };
    auto multiply = [&](const f32x4 (&x)[4], int q0) {   // 64 MFMAs: K = channels 8 q0 .. 8 q0 + 31, stage q >> 1, chunk pair q & 1
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = 4 * (q0 + j) + i;
                wfq[(n + 1) & 1] = wf_of(n + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (C1R_ABL & 8) o[i][0] += x[j][0] * wfq[n & 1][0];
                else mfma_quad<float>(x[j][0], x[j][1], x[j][2], x[j][3], wfq[n & 1], o[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    wfq[0] = wf_of(0);

    issue_half(xa, t, 0);
    issue_half(xb, t, 1);
    bool first = true;
#pragma unroll 1
    for (;;) {
        const long long tn = t + stride < ntiles ? t + stride : t;   // (last tile: its own rows again -- every tile issues the same operations: the counted waits stay valid)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float bias = coef_lds[512 + 4 * l31 + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = bias;
        }
        // first half: behind its sixteen loads stand the second half's sixteen and, from the second tile on, the sixteen stores of the tile before
        if (first) {
#pragma unroll
            for (int m = 0; m < 4; ++m) if (!(C1R_ABL & 2)) wn_uwait<16>(xa[m]);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) if (!(C1R_ABL & 2)) wn_uwait<32>(xa[m]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            bn_relu(xa[m], 4 * m);
            __builtin_amdgcn_sched_barrier(0);
            multiply(xa[m], 4 * m);
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_half(xa, tn, 0);   // the next tile's first half, into the registers just multiplied
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) if (!(C1R_ABL & 2)) wn_uwait<16>(xb[m]);   // (behind them: the sixteen loads just issued)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            bn_relu(xb[m], 16 + 4 * m);
            __builtin_amdgcn_sched_barrier(0);
            multiply(xb[m], 16 + 4 * m);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: ReLU, sixteen 16-byte stores (register r of the four tiles = four consecutive channels of one pixel)
        unsigned char* const ot = out + (size_t)t * (32 * 512);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 v = {br_relu(o[0][r]), br_relu(o[1][r]), br_relu(o[2][r]), br_relu(o[3][r])};
            if (!(C1R_ABL & 4) || v[0] == 12345.678f) *reinterpret_cast<f32x4*>(ot + ((r & 3) + 8 * (r >> 2)) * 512 + olane) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_half(xb, tn, 1);
        if (tn == t) break;
        t = tn;
        first = false;
    }
    // the re-requested rows of the last tile: their registers stay named until they have landed (hg_bt_wino_f32.h: a load into registers the
    // compiler considers dead lands in whatever they hold by then)
#pragma unroll
    for (int m = 0; m < 4; ++m) wn_uwait<0>(xa[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) wn_uwait<0>(xb[m]);
}

}  // namespace hgk
