// fp32 identity-skip bottleneck tail (3x3 -> 1x1 -> + x) with the 3x3 convolution as WINOGRAD F(2x2, 3x3) on the exact-fp32 MFMA
// (round 6).  The exact-fp32 engine sits at the matrix pipe's roof (0.88 of 157.3 TFLOP/s, hg_bt_ring_f32.h), so the only way left to make
// it faster is to issue fewer MFMAs: 82 % of the tail's MFMAs are the 3x3 done as direct implicit GEMM (9 taps x 128 x 128 per pixel);
// Winograd does a 2x2 output patch with 16 instead of 36 multiplies per (cin, cout) pair -- the tail's MFMA count falls to 0.545.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch of t1 (per channel), g: 3x3 weights (per cin, cout), Y: 2x2 outputs
//
// summed over the 128 input channels INSIDE the transformed domain: 16 "positions" p = 4 i + j, each a plain GEMM
//   M_p [128 cout x 32 patches] = U_p [128 cout x 128 cin] . V_p [128 cin x 32 patches]
// over the 32 patches (4 x 8) of an 8 x 16 output tile.  U = G g G^T is transformed once at df3d_hg_set_weights (fp64, rounded once:
// bt_wino_pack_kernel), V = B^T d B costs adds only, Y = A^T M A costs adds only.
//
// Mapping (one workgroup = one 8 x 16 tile as before, but ONE wave per SIMD: a wave holds 16 positions x 16 accumulator registers = 256):
//   * wave w owns output channels 32 w .. 32 w + 31 for all 32 patches and all 16 positions: its output transform is in-lane;
//   * K is walked in 16 chunks of 8 input channels.  Per chunk the four waves build V (16 positions x 8 channels x 32 patches = 16 KB,
//     double-buffered in LDS) from the t1 halo tile -- every wave a quarter, one channel of one patch per lane: 16 ds_read_b32, 32 adds,
//     16 ds_write_b32, under the MFMAs of the chunk before -- and each wave then runs 16 x 4 MFMAs: B = V_p (one ds_read_b128), A = its own
//     1 KB fragment of U_p straight from global memory (L2-resident, 1 MB per bottleneck, prefetched one chunk = 16 fragments ahead:
//     nothing is shared between waves on the weight side, so no LDS ring and no barrier for it);  ONE barrier per chunk (4 096 MFMA cycles);
//   * t2 = relu(Y + b2) (b2 is the start value of position (1,1), which enters all four outputs with weight +1) crosses to the
//     pixel-major mapping of phase 3 through LDS (64 KB in the dead t1 region), and phase 3 is hg_bt_ring_f32.h's, unchanged:
//     W3 through the 4-slot LDS-DMA ring (which takes the dead V buffers' place), residual add, ADD2 / UP / pooled outputs.
//
// LDS: V / ring 32 KB | t1 halo tile 90 KB (both 64-channel halves; t2 later) | b3 1 KB = 125 952 B: one workgroup per CU.
// Not bit-identical to the direct form (different products): the engine option `wino` selects it, the tests hold it to the
// fp32 tolerance against the torch oracle on every plan step, and `wino=0` keeps the direct kernels as the bit-identity reference.
#pragma once
#include "hg_bt_ring_f32.h"

#ifndef WN_ABL
#define WN_ABL 0   // development builds (scripts/build_variant.sh): ablation mask -- phase 2: 1 no U loads, 2 no input transform, 4 no chunk barrier, 8 no V fragment reads; the rest of a tile: 512 no patch reads, 1024 no V stores, 2048 no packed adds of the input transform; 16 no output transform, 32 no halo DMA for the next tile, 64 no residual / operand loads, 128 no output stores, 256 no phase-3 MFMAs
#endif

namespace hgk {

constexpr int WN_CHUNKS = 16;                          // K chunks of 8 input channels
constexpr int WN_U_BYTES = WN_CHUNKS * 16 * 4 * 1024;  // [chunk][pass][row group][wave] x 1 KB MFMA A fragments = 1 MiB per bottleneck
constexpr int WN_W3_BYTES = BRF_W3_STAGES * BR_STAGE_BYTES;   // behind U: W3's 16 stage images with their rows permuted (bt_wino_pack_w3_kernel)
constexpr int WN_STREAM_BYTES = WN_U_BYTES + WN_W3_BYTES;
constexpr int WN_STREAM_BYTES_L2 = WN_U_BYTES + 2 * WN_W3_BYTES;   // layer2: U | W3 | Wd (the skip convolution's weights, rows permuted the same way)
constexpr int WN_V_BYTES = 16 * 1024;                  // one V chunk: [channel 4][row group 4][channel quad 2][patch 32] x 16 bytes
constexpr int WN_T1_OFF = 2 * WN_V_BYTES;              // = BR_RING_BYTES: the W3 ring reuses the V buffers
constexpr int WN_T1_BYTES = 2 * BR_T1_BYTES;           // both 64-channel halves of the 10 x 18 halo tile (92 160)
constexpr int WN_T2_BYTES = BT_TH * BT_TW * 512;       // t2 [128 pixels][128 channels] fp32 (65 536), inside the t1 region
constexpr int WN_B3_OFF = WN_T1_OFF + WN_T1_BYTES;
constexpr int WN_RING2_OFF = WN_B3_OFF + 1024 + 512;   // b3 [256] | b2 [128] | W3 ring slots 4 .. 7
constexpr int WN_LDS_BYTES = WN_RING2_OFF + BR_RING_BYTES;
static_assert(WN_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(WN_T1_OFF == BR_RING_BYTES, "the W3 ring takes the V buffers' place");
static_assert(WN_T2_BYTES <= WN_T1_BYTES, "t2 lives in the t1 region");

// W2' [9][128 cout][128 cin] fp32 (bn3 folded) -> U stream.  One thread per (cout, cin): G g G^T in fp64, each value rounded once.
// Fragment (chunk c, pass e, wave w, column j) = 1 KB: lane (l31, half) holds U'_{i j}[32 w + l31][8 c + 4 half + e], i = 0..3 (pass e of a chunk
// multiplies the K pair (8 c + e, 8 c + 4 + e) at all 16 positions; a wave's four fragments of a pass are 4 KB contiguous: one scalar base,
// immediate offsets).  U' = s_i s_j U with s_2 = -1: the input transform builds row 2 / column 2 of V with the opposite sign (d1 - d2 instead of
// d2 - d1: its packed adds then need no operand swap), and the products U' V' = U V are unchanged.
__global__ __launch_bounds__(256) void bt_wino_pack_kernel(const float* __restrict__ w2, float* __restrict__ ustream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 128 * 128) return;
    const int co = idx >> 7, ci = idx & 127;
    double g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w2[((size_t)(ky * 3 + kx) * 128 + co) * 128 + ci];
    double t[4][3];   // G g
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        t[0][kx] = g[0][kx];
        t[1][kx] = 0.5 * (g[0][kx] + g[1][kx] + g[2][kx]);
        t[2][kx] = 0.5 * (g[0][kx] - g[1][kx] + g[2][kx]);
        t[3][kx] = g[2][kx];
    }
    const int c = ci >> 3, half = (ci >> 2) & 1, e = ci & 3, w = co >> 5, l31 = co & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double u[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};   // (G g) G^T
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ustream[((size_t)(((c * 4 + e) * 4 + w) * 4 + j) * 64 + half * 32 + l31) * 4 + i] = (float)(((i == 2) != (j == 2)) ? -u[j] : u[j]);
    }
}

// W3 [256][128] fp32 -> 16 stage images (hg_bt_ring_f32.h's: output half nh, 16-float K slice k8; 128 rows x 64 bytes, br_swz) with the ROWS PERMUTED:
// row 32 i + l of an image holds output channel 128 nh + 4 l + i.  Phase 3's accumulator tile i, column l31 is then channel 4 l31 + i: the lane's
// four tiles are four CONSECUTIVE channels of one pixel, and residual, addends, output and pooled outputs move as 16-byte accesses (a quarter of the
// vector-memory instructions, each of which this one-wave-per-SIMD kernel pays for in full; and a wave may have only 64 of them in flight).
__global__ __launch_bounds__(256) void bt_wino_pack_w3_kernel(const float* __restrict__ w3, unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= BRF_W3_STAGES * 512) return;
    const int k = idx >> 9, rem = idx & 511, c = rem & 3, r = rem >> 2, nh = k >> 3, k8 = k & 7;
    const int ch = 4 * (r & 31) + (r >> 5);
    *reinterpret_cast<u32x4*>(stream + (size_t)k * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(w3 + ((size_t)nh * 128 + ch) * 128 + 16 * k8 + 4 * c);
}

// What an instruction costs between two of a wave's fp32 MFMAs when the wave is alone on its SIMD (tests/perf/ubench/mfma_f32_shadow.hip, cycles
// added to a 64-cycle v_mfma_f32_32x32x2_f32): one VALU op 14, a clump of n VALU ops ~10 + 4.3 n (v_pk_add_f32 counts as one), global_load_dwordx4
// with a 64-bit vector address 17.6, with a scalar base + 32-bit lane offset 6.6, LDS-DMA the same, ds_read_b128 2.0, ds_write_b128 3.1, SALU / s_nop 0.
// The exact-fp32 MFMA evidently shares the vector ALU: NOTHING vector hides in its shadow.  Hence, in phase 2: scalar-base loads, the input
// transform as sixteen packed adds in ONE clump, no address arithmetic on the vector side.
// The four U fragments of a pass as ONE inline-assembly statement: scalar base (the wave's 4 KB of the pass), the lane's 16-byte offset, immediate
// offsets.  Inline assembly because, left to hipcc, the loads sink behind the MFMAs that still read the registers it wants to reuse for them (issued
// at the END of the pass they belong to, a vmcnt(1) at the top of every chunk: 580 cycles each); with "=&v" outputs and a scheduling fence behind
// the statement they are issued where they stand.  The s_nop: a base restored from a spill lane by v_readlane right in front of the statement is a
// VALU-writes-SGPR -> VMEM-reads-it hazard (5 wait states) that the hazard recogniser cannot see inside inline assembly (measured: a memory fault).
__device__ __forceinline__ void wn_uload4(f32x4 (&d)[4], const void* sbase, unsigned voff) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                 : "v"(voff), "s"(sbase)
                 : "memory");
}
// ... and the counted wait that makes a pass's four fragments valid (operations retire in issue order: N = the loads issued behind them); it names
// the registers as read-write operands, so the MFMAs that consume them depend on it
template <int N>
__device__ __forceinline__ void wn_uwait(f32x4 (&u)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : "n"(N) : "memory");
}
// one 16-byte load, scalar base + lane offset, issued where it stands (the residual tiles: hipcc sinks compiler-visible loads to their first use --
// the epilogue -- and the wave then sits out an HBM round trip with nothing else to do; measured 4.7 % of the kernel)
__device__ __forceinline__ void wn_xload1(f32x4& d, const void* sbase, unsigned voff) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(voff), "s"(sbase) : "memory");
}
// four 16-byte loads STEP bytes apart (scalar base + lane offset), issued where they stand
template <int STEP>
__device__ __forceinline__ void wn_xload4s(f32x4 (&d)[4], const void* sbase, unsigned voff) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:%6\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:%7\n\tglobal_load_dwordx4 %3, %4, %5 offset:%8"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                 : "v"(voff), "s"(sbase), "n"(STEP), "n"(2 * STEP), "n"(3 * STEP)
                 : "memory");
}
// The input transform of one (patch, channel): V' = B'^T d B' as SIXTEEN packed adds in one statement (one VALU clump per chunk).
// In: P[b] = (d[0][b], d[1][b]), Q[b] = (d[2][b], d[3][b]) -- the register pairs the two ds_read2st64_b32 of patch column b deliver.
// Rows first: per column b,  T[b] = (d0 - d2, d1 + d2),  S[b] = (d1 - d2, d1 - d3)   [row 2 with the opposite sign: see bt_wino_pack_kernel];
// then columns, on whole pairs: j = 0: X0 - X2, 1: X1 + X2, 2: X1 - X2 (opposite sign), 3: X1 - X3 for X = T (rows 0, 1) and X = S (rows 2, 3).
// Out: VT[j] = (V'[0][j], V'[1][j]), VS[j] = (V'[2][j], V'[3][j]).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wn_transform(const f32x2 (&P)[4], const f32x2 (&Q)[4], f32x2 (&VT)[4], f32x2 (&VS)[4]) {
    f32x2 T0, T1, T2, T3, S0, S1, S2, S3;
    asm volatile(
        "v_pk_add_f32 %8, %16, %20 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %12, %16, %20 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %9, %17, %21 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %13, %17, %21 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %10, %18, %22 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %14, %18, %22 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %11, %19, %23 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %15, %19, %23 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %8, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %9, %10\n\t"
        "v_pk_add_f32 %2, %9, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %9, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %4, %12, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %5, %13, %14\n\t"
        "v_pk_add_f32 %6, %13, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %7, %13, %15 neg_lo:[0,1] neg_hi:[0,1]"
        : "=&v"(VT[0]), "=&v"(VT[1]), "=&v"(VT[2]), "=&v"(VT[3]), "=&v"(VS[0]), "=&v"(VS[1]), "=&v"(VS[2]), "=&v"(VS[3]),
          "=&v"(T0), "=&v"(T1), "=&v"(T2), "=&v"(T3), "=&v"(S0), "=&v"(S1), "=&v"(S2), "=&v"(S3)
        : "v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]), "v"(Q[0]), "v"(Q[1]), "v"(Q[2]), "v"(Q[3]));
}

// L2: fp32 layer2 (128 -> 128 -> 128 -> 256 with a 1x1 SKIP CONVOLUTION instead of the identity skip): the same phases 1-2 on its t1, and in
// phase 3 the skip convolution accumulated into the same accumulators behind W3 (as layer2_tail_f32_kernel does): per output half eight more ring
// stages (Wd, rows permuted like W3's) whose A operand is the raw x of the lane's pixel, prefetched from global memory in MFMA layout; no residual.
template <bool UP, bool ADD2 = false, bool L2 = false>
__global__ __launch_bounds__(256, 1) void bottleneck_wino_f32_kernel(BtRingArgs p) {
    static_assert(!(UP && ADD2), "the fused up-path sum is written by plain blocks");
    static_assert(!L2 || (!UP && !ADD2), "layer2 is a plain block");
    using T = float;
    constexpr int CIN = L2 ? 128 : 256, CO = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;                      // phase 3 (the V double buffer until then)
    unsigned char* const t1_lds = smem + WN_T1_OFF;
    float* const b3_lds = reinterpret_cast<float*>(smem + WN_B3_OFF);
    float* const b2_lds = b3_lds + 256;
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned t1_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)t1_lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles_x = p.W / BT_TW, tiles_y = p.H / BT_TH;
    const int ntiles = p.V * tiles_y * tiles_x;
    // PERSISTENT: one workgroup per CU walks tiles vb = blockIdx.x, + gridDim.x, ... (a single resident workgroup has nobody to hide its
    // prologue behind: the next tile's t1 halo is requested while this tile's phase 3 runs).  XCD-aware order as in bottleneck_ring_f32_kernel:
    // virtual block vb runs on XCD vb % 8 (the grid is a multiple of 8, or one tile per workgroup) and XCD x takes the x-th contiguous eighth.
    auto tile_of = [&](int vb, int& tx0, int& ty0, int& view) {
        const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
        int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
        tx0 = (b % tiles_x) * BT_TW;
        b /= tiles_x;
        ty0 = (b % tiles_y) * BT_TH;
        view = b / tiles_y;
    };
    // the t1 halo tile by LDS-DMA, both 64-channel halves (hg_bt_ring_f32.h t1_issue; half kh at t1 + kh * BR_T1_BYTES)
    // ONE lane-derived register lives across phase 2 (uoff, the U loads' lane offset, opaque to the compiler); whatever else a tile derives from the
    // lane index is recomputed from it where it is needed -- kept alive, `lane` itself sat in scratch and every reload was a vmcnt(0)
    unsigned uoff = (unsigned)(lane * 16);
    asm volatile("" : "+v"(uoff));
    auto t1_issue = [&](int tx0, int ty0, int view) {
        // the per-piece lane values (halo pixel, swizzled chunk) and the per-piece uniform values (piece index, LDS address) are recomputed at every
        // call: hoisted out of the tile loop they are ~70 registers and ~50 spill lanes alive across phase 2 (the empty asms hide their loop invariance)
        int lane_ = (int)(uoff >> 4);
        asm volatile("" : "+v"(lane_));
        int wave_ = wave;
        asm volatile("" : "+s"(wave_));
        const unsigned char* const tin = reinterpret_cast<const unsigned char*>(p.t1in) + (size_t)view * p.H * p.W * 512;
        const unsigned char* const zer = reinterpret_cast<const unsigned char*>(p.zeros);
        const int q = lane_ >> 4, slot = lane_ & 15;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int pc = wave_ + 4 * k;
            if (pc < BT_HALO / 4) {
                const int hp = 4 * pc + q;
                const int hy = hp / BT_HW, hx = hp - hy * BT_HW;
                const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
                const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                const unsigned c16 = (unsigned)((slot ^ (hx & 15)) << 4);               // br_t1_swz(hp) = hx & 15
                const unsigned in_view = (unsigned)((y * p.W + x) * 512) + c16;         // (a view's t1 is < 4 GB: 32-bit offsets, ONE 64-bit add per piece)
                const unsigned char* const base = ok ? tin : zer;                       // pixels outside the image: the 256 zero bytes
                br_glds_piece64(base + (ok ? in_view : c16), t1_addr + (unsigned)(pc * 1024));
                br_glds_piece64(base + (ok ? in_view + 256u : c16), t1_addr + (unsigned)(BR_T1_BYTES + pc * 1024));
            }
        }
    };

    // ---- U fragments straight from global memory (L2) into the MFMA A registers: the wave's four fragments (columns j = 0..3) of (chunk c, pass e)
    //      are 4 KB at U + ((4 c + e) 4 + wave) 4096; lane (l31, half) of fragment j -> U'_{i j}[32 wave + l31][8 c + 4 half + e], i = 0..3.
    //      Rolling prefetch three passes (3 072 MFMA cycles) ahead: at the start of pass e the registers of the pass before are free and take
    //      (c + 1, e - 1) [pass 0: (c, 3)] -------------------------------------------------------------------------------------------------------
    const unsigned char* const ubase = reinterpret_cast<const unsigned char*>(p.w2d) + (size_t)wave * 4096;
    auto uload = [&](int c, int e, f32x4 (&dst)[4]) { wn_uload4(dst, ubase + (size_t)(c * 4 + e) * 16384, uoff); };

    // ---- input transform: lane -> (patch 8 wave + (lane & 7), channel quad (lane >> 3) & 1, channel lane >> 4) of the chunk -----------
    // t1 element (halo pixel hp, channel 64 kh + 4 kq + e) sits at hp * 256 + ((kq ^ swz(hp)) << 4) + 4 e of half kh (br_t1_swz): the 64 lanes
    // of a read touch 64 different banks; chunk c's quads are kq = 2 (c & 7) + {0, 1}: an XOR of the address with (c & 7) << 5
    const int ptx = lane & 7, pkq = (lane >> 3) & 1, pe = lane >> 4;
    unsigned rd[4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
        const int hx = 2 * ptx + bb, hp = (2 * wave) * BT_HW + hx;
        rd[bb] = (unsigned)(hp * 256 + ((pkq ^ (hx & 15)) << 4) + 4 * pe);
    }
    // V chunk image [channel e 4][column j 4][quad kq 2][patch 32][row i 4] floats: the 16 bytes a lane writes per column j (two 8-byte halves)
    // are V'_{0..3, j} of its (patch, channel), the 16 bytes a lane reads per (e, j) are the B values of MFMAs (4 i + j, e) for its patch; patch n of
    // quad kq sits in slot n ^ 8 kq (a permuted 512-byte row either way)
    const unsigned vwr = (unsigned)(pe * 4096 + pkq * 512 + (((8 * wave + ptx) ^ (pkq << 3)) << 4));
    const unsigned vrd = (unsigned)(half * 512 + ((l31 ^ (half << 3)) << 4));
    // three V buffers (chunk c is read from buffer c % 3 while chunk c + 2 is built into (c + 2) % 3: a chunk's first fragments can then be
    // requested before the barrier that ends the chunk in front of it): 0, 1 where the W3 ring's slots 0 .. 3 will be, 2 where its slots 4 .. 7 will be
    auto vbuf_off = [](int b) { return b < 2 ? b * WN_V_BYTES : WN_RING2_OFF; };
    f32x2 tP[4], tQ[4];
    unsigned ta[4];   // LDS addresses of the next patch reads (computed in the VALU clump of the chunk before: no lone VALU instruction in phase 2)
    auto t_addr = [&](int c) {   // chunk c's patch columns
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) ta[bb] = (rd[bb] ^ (unsigned)((c & 7) << 5)) + (t1_addr + (unsigned)((c >> 3) * BR_T1_BYTES));   // (a complete LDS address: one v_xad_u32)
    };
    auto t_read = [&](int bb) {   // column bb of the 4 x 4 patch: rows (0, 1) and (2, 3) as register pairs
        typedef const __attribute__((address_space(3))) float* lds_f;
        if (WN_ABL & 512) {   // (timing only: no patch reads)
            asm volatile("" : "+v"(tP[bb]), "+v"(tQ[bb]));
            return;
        }
        tP[bb] = f32x2{*(lds_f)(size_t)ta[bb], *(lds_f)(size_t)(ta[bb] + BT_HW * 256)};
        tQ[bb] = f32x2{*(lds_f)(size_t)(ta[bb] + 2 * BT_HW * 256), *(lds_f)(size_t)(ta[bb] + 3 * BT_HW * 256)};
    };
    f32x2 vt[4], vs[4];   // V' of the lane's (patch, channel): columns j = 0 .. 3, rows (0, 1) and (2, 3)
    // sixteen packed adds + the next reads' four addresses in one clump
    auto t_transform = [&](int c_next_addr) {
        wn_transform(tP, tQ, vt, vs);
        t_addr(c_next_addr);
    };
    // ... and column j's 16 bytes into V buffer `buf`: in phase 2 one store per MFMA group (between MFMAs an LDS instruction is all but free; the four
    // stores right behind the clump cost 1.8 % of the kernel)
    auto v_store = [&](int buf, int j) {
        unsigned char* const dst = smem + vbuf_off(buf) + vwr;
        *reinterpret_cast<f32x2*>(dst + j * 1024) = vt[j];
        *reinterpret_cast<f32x2*>(dst + j * 1024 + 8) = vs[j];
    };
    auto t_transform_write = [&](int buf, int c_next_addr) {   // (tile entry: clump, then the stores)
        if (WN_ABL & 2048) {   // (timing only: no packed adds)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vt[j] = tP[j];
                vs[j] = tQ[j];
            }
        } else {
            wn_transform(tP, tQ, vt, vs);
        }
        t_addr(c_next_addr);
        unsigned char* const dst = smem + vbuf_off(buf) + vwr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (WN_ABL & 1024) {   // (timing only: no V stores)
                asm volatile("" ::"v"(vt[j]), "v"(vs[j]));
                continue;
            }
            *reinterpret_cast<f32x2*>(dst + j * 1024) = vt[j];
            *reinterpret_cast<f32x2*>(dst + j * 1024 + 8) = vs[j];
        }
    };
    // W3's 16 stages in two sets of eight (one per 128-channel output half): stage k -> slot k % 8, slots 0 .. 3 where the V buffers were, 4 .. 7
    // in a region of their own.  A whole half is resident before its K loop starts: no wait, no barrier inside the loop (operations retire in
    // issue order, so a wait for a young ring stage would also wait for every older load -- the tile's 128 KB of residual values, the next
    // tile's halo, which all 256 workgroups request within the same microsecond)
    auto ring_slot = [](int k) { return (k & 7) < 4 ? (k & 7) * BR_STAGE_BYTES : WN_RING2_OFF + ((k & 7) - 4) * BR_STAGE_BYTES; };
    auto ring_issue8 = [&](int set) {   // stage set: W3 half 0, W3 half 1 (, L2: Wd half 0, Wd half 1); this wave copies pieces 2 wave, 2 wave + 1 of each stage
        unsigned wvoff = (unsigned)wave * 2048u + uoff;
        asm volatile("" : "+v"(wvoff));   // (per call: not one more register across phase 2)
#pragma unroll
        for (int k = 8 * set; k < 8 * set + 8; ++k)
            br_glds_stage(reinterpret_cast<const unsigned char*>(p.w2d) + WN_U_BYTES + (size_t)k * BR_STAGE_BYTES, wvoff,
                          ring_addr + (unsigned)ring_slot(k) + (unsigned)wave * 2048);
    };

#ifdef DF3D_BT_TIMING
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();   // (timing builds: scripts/probe_wino.py)
#endif
    int vb = blockIdx.x;
    int tx0, ty0, view;
    tile_of(vb, tx0, ty0, view);
    t1_issue(tx0, ty0, view);
    b3_lds[tid] = L2 ? p.b3[tid] + p.bd[tid] : p.b3[tid];   // (L2: b3 + bd as ONE float add, as the direct kernels)
    if (tid < 128) b2_lds[tid] = p.b2[tid];
    bool first = true;

#pragma unroll 1
    for (;;) {
        const int vbn = vb + (int)gridDim.x;
        const bool has_next = vbn < ntiles;
        int ntx0 = 0, nty0 = 0, nview = 0;
        if (has_next) tile_of(vbn, ntx0, nty0, nview);
        // uniform byte offsets of this wave's two tile rows (full resolution) / its one half-resolution row inside a [V][H][W][256] fp32 tensor
        const size_t ftile = (((size_t)view * p.H + ty0 + 2 * wave) * p.W + tx0) * (CO * 4);
        const size_t htile = (((size_t)view * (p.H / 2) + ty0 / 2 + wave) * (p.W / 2) + tx0 / 2) * (CO * 4);
        const unsigned char* const xtile = reinterpret_cast<const unsigned char*>(p.in) + (L2 ? ftile / 2 : ftile);   // (L2: x has 128 channels)

        // accumulators (b2 is added to position (1,1) -- which enters all four outputs of a patch with weight +1 -- in the output transform: as a
        // start value it would be a global load straight into accumulator registers, and the wait hipcc puts in front of the first MFMA that
        // touches them sits inside the chunk loop: a vmcnt(0) per chunk, 580 cycles each)
        f32x16 acc[16];   // (their first MFMAs take a zero addend: chunk 0, pass 0)
        f32x4 ufr[4][4];
        uload(0, 0, ufr[0]);
        uload(0, 1, ufr[1]);
        uload(0, 2, ufr[2]);
        // first tile: both t1 halves of this wave have landed once only the loads issued behind them are outstanding.  Later tiles: the halo was
        // requested during the tile before's phase 3, whose counted waits and barriers have long published it.  The barrier: every wave is past
        // its last reads of the ring (the tile before), V buffer 0 may be written
        if (first) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        first = false;
        br_barrier();
        BR_STAMP(0);
        asm volatile("" : "+v"(rd[0]), "+v"(rd[1]), "+v"(rd[2]), "+v"(rd[3]));   // (or the first chunks' read addresses are hoisted out of the tile loop: 12 registers kept alive in scratch)
        t_addr(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) t_read(g);
        t_transform_write(0, 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) t_read(g);
        t_transform_write(1, 2);
        br_barrier();
        BR_STAMP(1);

        // ---- phase 2: 16 chunks x 4 passes (K pair e) x 4 columns x 4 rows.  Chunk c: MFMAs on V(c) (buffer c % 3) while V(c + 2) is built.
        //      Nothing vector-side hides behind an fp32 MFMA here, so a chunk issues, beside its 64 MFMAs: 16 U loads (scalar base), 16 + 8 LDS
        //      reads, 4 LDS stores, ONE clump of 20 VALU instructions, one barrier -------------------------------------------------------------
        f32x4 vf[2][4];
        // c: the chunk (runtime); BR = c % 3 and FIRST = (c == 0) as compile-time tags
        auto chunk = [&](int c, auto br_tag, auto first_tag) {
            constexpr int BR = decltype(br_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            const int cn = c + 1 < WN_CHUNKS ? c + 1 : c;   // (the last chunks re-request fragments / rebuild buffers nobody reads any more: one code path)
            const int c2 = c + 2 < WN_CHUNKS ? c + 2 : WN_CHUNKS - 1, c3 = c + 3 < WN_CHUNKS ? c + 3 : WN_CHUNKS - 1;
            (void)c2;
            const unsigned char* const vb_ = smem + vbuf_off(BR) + vrd;
            const unsigned char* const vn_ = smem + vbuf_off((BR + 1) % 3) + vrd;
            if (FIRST) {
#pragma unroll
                for (int g = 0; g < 4; ++g) vf[0][g] = *reinterpret_cast<const f32x4*>(vb_ + g * 1024);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __builtin_amdgcn_sched_barrier(0);
                if (!(WN_ABL & 1)) {
                    wn_uwait<8>(ufr[e]);   // this pass's fragments (requested three passes ago) have landed: behind them only two passes' 8 loads
                    if (e == 0) uload(c, 3, ufr[3]);
                    else uload(cn, e - 1, ufr[e - 1]);
                    __builtin_amdgcn_sched_barrier(0);   // (... issued HERE: the scheduler would sink the statement behind the pass's MFMAs)
                }
                if (e == 2 && !(WN_ABL & 2)) {
                    if (WN_ABL & 4096) t_transform_write((BR + 2) % 3, c3);   // (development: the stores right behind the clump, as before)
                    else t_transform(c3);   // V(c + 2) from the patch read in pass 0; addresses for the reads of chunk c + 1's pass 0
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // LDS instructions are all but free between MFMAs: one V fragment read for the next pass (pass 3: the NEXT chunk's pass 0,
                    // whose buffer was complete a barrier ago) and, in pass 0, a column of the patch V(c + 2) is built from
                    if (e < 3) vf[(e + 1) & 1][g] = *reinterpret_cast<const f32x4*>(vb_ + (e + 1) * 4096 + g * 1024);
                    else vf[0][g] = *reinterpret_cast<const f32x4*>(vn_ + g * 1024);
                    if (e == 0 && !(WN_ABL & 2)) t_read(g);
                    if (e == 2 && !(WN_ABL & (2 | 4096 | 1024))) v_store((BR + 2) % 3, g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (FIRST && e == 0)   // a tile's first product into each accumulator starts from zero: no 256 v_accvgpr_write per tile
                            acc[4 * i + g] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[e][g][i], vf[e & 1][g][i], f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
                        else
                            acc[4 * i + g] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[e][g][i], vf[e & 1][g][i], acc[4 * i + g], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!(WN_ABL & 4)) br_barrier();   // V(c) is read by every wave (its buffer is rebuilt two chunks on), V(c + 2) written by every wave
        };
        chunk(0, std::integral_constant<int, 0>{}, std::true_type{});
#pragma unroll 1
        for (int c = 1; c < WN_CHUNKS; c += 3) {
            chunk(c, std::integral_constant<int, 1>{}, std::false_type{});
            chunk(c + 1, std::integral_constant<int, 2>{}, std::false_type{});
            chunk(c + 2, std::integral_constant<int, 0>{}, std::false_type{});
        }
        // the last chunk re-requested fragments nobody multiplies: keep their registers named until they have landed (a load whose destination is
        // dead to the compiler lands, asynchronously, in whatever the register holds by then)
        if (!(WN_ABL & 1)) {
            wn_uwait<4>(ufr[0]);
            wn_uwait<0>(ufr[1]);
            wn_uwait<0>(ufr[2]);
        }

        BR_STAMP(2);
        // Everything phase 3 derives from the lane index is derived HERE, per tile, from a copy the compiler cannot see through: hoisted out of
        // the tile loop these ~60 lane constants would live across phase 2, where the 256 + 256 registers are spoken for (scratch spills, and
        // every reload a vmcnt(0) in the middle of the asynchronous machinery)
        int lane3 = (int)(uoff >> 4);
        asm volatile("" : "+v"(lane3));
        const int half3 = lane3 >> 5, l31_3 = lane3 & 31;
        const unsigned char* const wf0 = ring + br_swz(l31_3, half3);
        const unsigned char* const wf1 = ring + br_swz(l31_3, 2 + half3);
        const int py = 2 * wave + (l31_3 >> 4), px = l31_3 & 15;   // this wave's 32 pixels (phase 3)
        // phase 3's global addresses = (tile, register)-dependent UNIFORM part + one of two per-lane byte offsets (pixel 4 half / half-resolution
        // pixel 2 half of the register's group, channels 4 l31 .. 4 l31 + 3 -- bt_wino_pack_w3_kernel): scalar address arithmetic, two VGPRs -- not one address register pair per access
        const unsigned lane_full = (unsigned)((4 * half3 * CO + 4 * l31_3) * 4);
        const unsigned lane_half = (unsigned)((2 * half3 * CO + 4 * l31_3) * 4);
        // W3's first half into the ring (both V buffers are dead; slots 4 .. 7 were last read a tile ago): it lands under the output transform
        ring_issue8(0);

        // ---- output transform Y = A^T M A, ReLU, t2 -> LDS (pixel-major, 16-byte chunk ch of pixel (y, x) in slot ch ^ (x & 15) ^ ((y >> 1) & 1)
        //      of its 512-byte row: conflict-free for these writes (16 lanes = 8 patch columns x 2 patch rows) and for phase 3's reads).
        //      Address = [pixel (2 ty, 2 tx) | lane part of the slot] ^ [(2 q ^ bb) << 4] + (16 a + bb) * 512: one lane base, XOR constants ----
        if (WN_ABL & 16) {   // (timing only: one value that depends on every accumulator tile, so that phase 2 stays)
            float keep = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) keep += acc[k][0];
            *reinterpret_cast<float*>(t1_lds + lane3 * 4) = keep;
        } else {
            const int ty = l31_3 >> 3, tx = l31_3 & 7;
            const unsigned wbase = (unsigned)((32 * ty + 2 * tx) * 512 + ((((8 * wave + half3) ^ (2 * tx) ^ (ty & 1)) & 31) << 4));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 y[2][2];
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b2_lds + 32 * wave + 8 * q + 4 * half3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    float s[2][4];   // A^T M: rows
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float m1 = j == 1 ? acc[4 + j][r] + bb[e] : acc[4 + j][r];
                        s[0][j] = acc[j][r] + m1 + acc[8 + j][r];
                        s[1][j] = m1 - acc[8 + j][r] - acc[12 + j][r];
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        y[a][0][e] = br_relu(s[a][0] + s[a][1] + s[a][2]);
                        y[a][1][e] = br_relu(s[a][1] - s[a][2] - s[a][3]);
                    }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
                        *reinterpret_cast<f32x4*>(t1_lds + (wbase ^ (unsigned)(((2 * q) ^ bb) << 4)) + (16 * a + bb) * 512) = y[a][bb];
            }
        }
        BR_STAMP(3);
        br_wait_vm(0);   // the ring's first half (and whatever is older: the tile before's stores, a whole tile ago)
        br_barrier();    // t2 and the eight stages are every wave's
        f32x16 t2[4];
        {
            const unsigned rbase = (unsigned)((py * BT_TW + px) * 512 + ((half3 ^ (px & 15) ^ (wave & 1)) << 4));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {   // registers 4 g + e of tile m: channels 32 m + 8 g + 4 half + e = chunk 8 m + 2 g + half, in slot chunk ^ f
                    const f32x4 v = *reinterpret_cast<const f32x4*>(t1_lds + (rbase ^ (unsigned)((8 * m + 2 * g) << 4)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) t2[m][4 * g + e] = v[e];
                }
        }

        br_barrier();   // every wave holds its t2 in registers: the t1 region may take the next tile's halo
        if (has_next && !(WN_ABL & 32)) t1_issue(ntx0, nty0, nview);
        // ---- residual values (and the ADD2 addends) of the whole tile, requested here -- behind the ring's first half (no ring wait has to let them pass), 16 000 MFMA cycles
        //      in front of their first use (one wave per SIMD: nobody hides a load issued in an epilogue; the 512-register file has room) ----
        f32x4 xres[1][16];   // !L2: the CURRENT output half's residual values, [register r <-> pixel (r & 3) + 8 (r >> 2) + 4 half]: channels 128 nh + 4 l31 .. + 3
                             // L2: [0][2 k8 + jj] = the skip convolution's A operand, x[this lane's pixel][16 k8 + 8 jj + 4 half ..]
        f32x4 exv[4];        // ADD2: the addends / UP: the half-resolution values of the current half's four pixel quads
        static_assert(!(UP && ADD2), "one extra operand");
        // (inline assembly: scalar base + lane offset, issued HERE -- half 0 now, 16 000 MFMA cycles in front of its epilogue; half 1 behind the
        // first epilogue, into the same registers)
        auto res_issue = [&](int nh) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // registers 4 g .. 4 g + 3: pixels 8 g .. 8 g + 3 (+ 4 half) = row g >> 1, columns 8 (g & 1) ..
                if (WN_ABL & 64) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xres[0][4 * g + j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                } else {
                    wn_uload4(*reinterpret_cast<f32x4(*)[4]>(&xres[0][4 * g]), xtile + ((size_t)(g >> 1) * p.W + 8 * (g & 1)) * (CIN * 4) + nh * 512, lane_full);
                }
            }
            if constexpr (ADD2 || UP) {
#pragma unroll
                for (int key = 0; key < 4; ++key)
                    wn_xload1(exv[key], reinterpret_cast<const unsigned char*>(ADD2 ? p.add2 : p.in2) + htile + ((key & 1) + 4 * (key >> 1)) * (CO * 4) + nh * 512, lane_half);
            }
        };
        auto res_wait = [&](auto n_tag) {   // the counted wait that makes them valid, naming the registers (the epilogue's reads depend on it)
            constexpr int N = decltype(n_tag)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) wn_uwait<N>(*reinterpret_cast<f32x4(*)[4]>(&xres[0][4 * g]));
            if constexpr (ADD2 || UP) wn_uwait<N>(exv);
        };
        if constexpr (L2) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                xres[0][r] = *reinterpret_cast<const f32x4*>(xtile + (8 * r) * 4 + (unsigned)((((l31_3 >> 4) * p.W + (l31_3 & 15)) * CIN + 4 * half3) * 4));
        } else {
            res_issue(0);
        }
        BR_STAMP(4);
        // ---- phase 3: out = W3 relu(t2) + b3 + x  (bottleneck_ring_f32_kernel's exact-fp32 form: rows = the wave's pixels, columns = channels);
        //      L2: out = W3 relu(t2) + Wd x + (b3 + bd) ----
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            f32x16 o[4];
            if (nh == 1) {
                // the second half's stages were requested behind the first half's K loop(s); younger than them are only the first half's epilogue's
                // operations -- at least its 16 output stores -- which need not have drained (operations retire in issue order)
                // (!L2: and the second half's residual loads: 16 + 4 with an extra operand)
                br_wait_vm(L2 ? 16 : (ADD2 || UP) ? 36 : 32);
                br_barrier();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float bias = b3_lds[nh * 128 + 4 * l31_3 + i];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] = bias;
            }
            // eight resident stages, no wait, no barrier: stage k8 = the 16-float K slice 16 k8 .. of W3 (operand: registers 8 q2 + 4 jj + e of t2
            // tile k8 >> 1 = channels 32 tile + 16 q2 + 8 jj + 4 half + e) or of Wd (operand: x of the lane's pixel, channels 16 k8 + 8 jj + 4 half + e)
            auto kloop = [&](auto skip_tag) {
                constexpr bool SKIP = decltype(skip_tag)::value;
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int tile = k8 >> 1, q2 = k8 & 1;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 wf = *reinterpret_cast<const f32x4*>((jj ? wf1 : wf0) + ring_slot(k8) + i * 2048);
                            if (WN_ABL & 256) {
                                o[i][0] += wf[0];
                                continue;
                            }
                            if constexpr (SKIP) {
                                const f32x4 xa = xres[0][2 * k8 + jj];
                                mfma_quad<T>(xa[0], xa[1], xa[2], xa[3], wf, o[i]);
                            } else {
                                mfma_quad<T>(t2[tile][8 * q2 + 4 * jj], t2[tile][8 * q2 + 4 * jj + 1], t2[tile][8 * q2 + 4 * jj + 2], t2[tile][8 * q2 + 4 * jj + 3], wf, o[i]);
                            }
                        }
                }
            };
            kloop(std::false_type{});
            if constexpr (L2) {
                br_barrier();             // every wave is through W3's eight stages
                ring_issue8(2 + nh);      // Wd of this half (L2-resident: the one exposed round trip of the half)
                br_wait_vm(0);
                br_barrier();
                kloop(std::true_type{});
            }
            if (nh == 0) {
                br_barrier();     // every wave is through the first half's stages
                ring_issue8(1);   // ... the second half's land under the epilogue
            }
            BR_STAMP(5 + 2 * nh);
            if constexpr (!L2) {
                if (!(WN_ABL & 64)) {
                    if (nh == 0) res_wait(std::integral_constant<int, 16>{});   // younger: the sixteen stage pieces just requested
                    else res_wait(std::integral_constant<int, 0>{});
                }
            }
            // epilogue: D[row = pixel (r&3) + 8(r>>2) + 4 half of the wave][col = channel nh*128 + 4 l31 + i]: register r of the four tiles = four
            // consecutive channels of one pixel
            // -- walked by 2x2 pixel quads (registers r0, r0 + 1, r0 + 8, r0 + 9: horizontal neighbour r ^ 1, vertical r ^ 8; one half-resolution pixel),
            // so that only four 16-byte values are alive at a time and the max-pools stay inside the lane
#pragma unroll
            for (int key = 0; key < 4; ++key) {
                const int r0 = 2 * (key & 1) + 4 * (key >> 1);
                const int hoff = ((key & 1) + 4 * (key >> 1)) * (CO * 4) + nh * 512;   // the quad's half-resolution pixel (+ 2 half in the lane offset)
                f32x4 xv[4], ov[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) xv[t] = L2 ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : xres[0][r0 + (t & 1) + 8 * (t >> 1)];
                if constexpr (UP) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) xv[t] += exv[key];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = r0 + (t & 1) + 8 * (t >> 1);
                    ov[t] = f32x4{o[0][r], o[1][r], o[2][r], o[3][r]};
                    if constexpr (!L2) ov[t] += xv[t];
                    if constexpr (ADD2) ov[t] += exv[key];   // a second fp32 add, as upadd_kernel would have done on the stored tensor
                    const int pl0 = (r & 3) + 8 * (r >> 2);
                    if (!(WN_ABL & 128) || ov[t][0] == 12345.678f)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.out) + ftile + ((size_t)(pl0 >> 4) * p.W + (pl0 & 15)) * (CO * 4) + nh * 512 + lane_full) = ov[t];
                }
                auto max4 = [](const f32x4 (&v)[4]) {
                    f32x4 m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(v[0][e], v[1][e]), fmaxf(v[2][e], v[3][e]));
                    return m;
                };
                if constexpr (!UP && !L2) if (p.pool_in)   // 2x2 max-pool of the block's INPUT (the skip values just added)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.pool_in) + htile + hoff + lane_half) = max4(xv);
                if (p.pool) *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.pool) + htile + hoff + lane_half) = max4(ov);
            }
            BR_STAMP(6 + 2 * nh);
            if constexpr (!L2) {
                if (nh == 0) res_issue(1);
            }
        }
        if (!has_next) break;
        vb = vbn;
        tx0 = ntx0;
        ty0 = nty0;
        view = nview;
    }
}

}  // namespace hgk
