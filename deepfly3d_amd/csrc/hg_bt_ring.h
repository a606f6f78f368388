// 16-bit (T = __hip_bfloat16 or _Float16: hg_kernels.h Lp<T>; "bf16" below stands for either) fused pre-activation bottleneck
// 256 -> 128 -> 128 -> 256 (identity skip) with the WEIGHTS streamed by LDS-DMA.
//
// Same tile (8 x 16 output pixels, 10 x 18 halo), same wave -> tile mapping, same MFMA K order as
// hg_kernels.h:bottleneck_kernel<bf16, 256, 128, false, UP> -- the results are bit-identical -- but the 416 KB of
// weights a tile consumes (W1 64 KB, W2 288 KB, W3 64 KB: 2.7 x the activation bytes of the tile) no longer travel
// global -> VGPR -> ds_write behind two barriers per K-step.  Instead:
//
//   * the engine keeps, per bottleneck, a "weight stream": 52 stages x 8 KB, every stage the exact LDS image of one
//     K-slice (128 rows x 32 bf16) in a bank-conflict-free order (bt_ring_pack_kernel, at set_weights time);
//   * the kernel copies the stages into a 4-slot LDS ring with global_load_lds_dwordx4 (asynchronous, no VGPRs, no VALU,
//     no ds_write) while the MFMAs consume earlier ones: phase 1 one stage per barrier, three stages ahead; phases 2 and 3
//     TWO stages per barrier (16 MFMAs per wave between barriers), the next pair requested right after the barrier into the
//     slots just released; counted s_waitcnt vmcnt(N) only, 16-24 KB of weights in flight per workgroup;
//   * x (phase 1) is staged through registers as before (it needs bn1 + ReLU and, UP, the upsample add) but three
//     K-steps ahead, into a 3-deep LDS ring that lives in the not-yet-written t1 region, one barrier per step; the
//     bn1 scale / shift vectors sit in LDS so that no other vector-memory operation sits in the in-order queue.
//
//   * W2D (round 3, the engine's default; template flag): the 36 W2 stages skip the ring -- phase 2 runs channel-split, every wave
//     loading its own 1 KB MFMA fragments straight from global memory (see the template's comment); the ring then carries W1 | W3.
//
// LDS per workgroup: ring 32 KB (4 stages) + t1 45 KB (180 rows x 256 B, XOR-swizzled instead of padded) + coefficients
// 2.5 KB + masks = 81 472 B -> two workgroups per CU.
//
// Stage image (8 KB): row r (0..127), 16-byte chunk c (0..3) at byte (r >> 2) * 256 + ((((r & 3) << 2 | c) ^ ((r >> 3) & 3)) << 4).
// An MFMA fragment read (ds_read_b128; 32 consecutive rows, one chunk per lane half) then touches 16 distinct 16-byte
// slots of the 256-byte bank row in each of the instruction's four lane groups.
//
// vmcnt discipline: gfx950 retires vector-memory operations (loads, stores, LDS-DMA) in issue order, so
// "s_waitcnt vmcnt(N)" with N = the number of operations issued AFTER stage s's two DMA pieces guarantees that stage s
// has landed for this wave; the barrier that follows makes that true for all four waves.  A smaller N is always safe.
#pragma once
#include <type_traits>

#include "hg_kernels.h"
#ifndef BR_ABLM
#define BR_ABLM 0   // development builds: ablation mask (1 no MFMAs, 2 no weight DMA, 4 no x loads, 8 no residual loads, 16 no output stores, 512 no phase-2 MFMAs of the W2D form)
#endif

namespace hgk {

struct BtRingArgs {
    const void* in;       // NHWC bf16 [V, H, W, 256]
    const void* in2;      // UP: NHWC bf16 [V, H/2, W/2, 256]; the block's input is in + nearest-upsample(in2), rounded to bf16
    const void* add2;     // ADD2: NHWC [V, H/2, W/2, 256]; the block WRITES out + nearest-upsample(add2) -- the hourglass' up-path sum,
                          // with upadd_kernel's roundings (the rounded block output plus the low-resolution tensor, rounded again)
    void* out;            // NHWC bf16 [V, H, W, 256]
    void* pool;           // optional NHWC bf16 [V, H/2, W/2, 256]: 2x2 max-pool of `out`
    void* pool_in;        // optional NHWC [V, H/2, W/2, 256]: 2x2 max-pool of the block's INPUT (for the hourglass level whose
                          // input no fused producer has pooled: its skip values pass through the epilogue anyway)
    const void* wstream;  // br_nstage(CIN, DS) x BR_STAGE_BYTES: pre-swizzled stage images (bt_ring_pack_kernel)
    const void* w2d;      // W2D kernels: W2' as 72 x 4 one-KB MFMA A fragments (bt_w2d_pack_kernel): group (tap, kc, K half), wave's 32 rows
    const void* t1in;     // fp32 split form (hg_c1_f32.h): [V, H, W, 128] f32 = relu(W1' relu(bn1 x) + b1'), from conv1_ring_f32_kernel
    const void* zeros;    // ... and >= 256 bytes of zeros (the padding of the 3x3 convolution for halo pixels outside the image)
    const float* b1;      // [128] (bn2 folded)
    const float* b2;      // [128] (bn3 folded)
    const float* b3;      // [256]
    const float* bd;      // [256] skip-convolution bias (DS kernels only)
    const float* s1;      // [CIN] bn1 scale
    const float* t1;      // [CIN] bn1 shift
    int V, H, W;
};

constexpr int BR_STAGE_BYTES = 8192;                     // 128 rows x 64 bytes (32 bf16 of K)
constexpr int BR_RING = 4;
constexpr int BR_W1_STAGES = 8, BR_W2_STAGES = 36, BR_W3_STAGES = 8;
constexpr int BR_NSTAGE = BR_W1_STAGES + BR_W2_STAGES + BR_W3_STAGES;   // 52
constexpr int BR_T1_PITCH = 128 * 2;                    // bytes per halo pixel of the t1 tile: no padding, 16-byte chunks XOR-swizzled
constexpr int BR_T1_BYTES = BT_HALO * BR_T1_PITCH;      // 46 080 (the 12 pad rows of the sixth MFMA row tile are not stored)
constexpr int BR_XPITCH = 64;                           // staged x rows: one 64-byte K step, no padding: 16-byte chunk c of row r sits in slot
                                                        // c ^ ((r >> 2) & 3) (br_xslot), which makes the 32-row fragment reads and the row-pair
                                                        // stores bank-conflict-free (round 3: the padded 80-byte rows cost 13-15 % of the LDS cycles)
constexpr int BR_XSTAGE = BT_HROWS * BR_XPITCH;         // 12 288; three of them inside the t1 region
constexpr int BR_RING_BYTES = BR_RING * BR_STAGE_BYTES;
constexpr int BR_COEF_BYTES = 512 * 4 + 128 * 4;        // phase 1: bn1 scale [256] | shift [256]; afterwards b2 [128] | b3 [256]; then b1 [128]
constexpr int BR_LDS_BYTES = BR_RING_BYTES + BR_T1_BYTES + BR_COEF_BYTES + 64;
// chunk k (8 channels) of halo pixel hp sits in 16-byte slot k ^ ((hp % 18) & 15) of its 256-byte row: the 16 lanes of a
// ds_read_b128 lane group read 16 different tile columns, hence 16 different slots
__device__ __forceinline__ int br_t1_swz(int hp) { return (hp % BT_HW) & 15; }
static_assert(3 * BR_XSTAGE <= BR_T1_BYTES, "the x ring lives inside the t1 region");
static_assert(2 * BR_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ int br_xslot(int row, int chunk) { return ((chunk ^ ((row >> 2) & 3)) << 4); }   // byte offset inside a staged x row
__host__ __device__ constexpr int br_swz(int r, int c) { return (r >> 2) * 256 + (((((r & 3) << 2) | c) ^ ((r >> 3) & 3)) << 4); }

// stages of one bottleneck's stream: W1 (CIN / 32) | W2 (36) | per 128-channel output half: W3 (4) and, with the 1x1 skip
// convolution (DS: CIN != 256), Wd (CIN / 32)
__host__ __device__ constexpr int br_nstage(int cin, bool ds) { return cin / 32 + BR_W2_STAGES + 2 * (4 + (ds ? cin / 32 : 0)); }

// bf16 blob -> weight stream of one bottleneck.  One thread per 16-byte chunk: stages x 128 rows x 4 chunks.
__global__ __launch_bounds__(256) void bt_ring_pack_kernel(const unsigned short* __restrict__ w1, const unsigned short* __restrict__ w2,
                                                           const unsigned short* __restrict__ w3, const unsigned short* __restrict__ wd, int cin,
                                                           unsigned char* __restrict__ stream) {
    const bool ds = wd != nullptr;
    const int ns1 = cin / 32, per_nh = 4 + (ds ? ns1 : 0);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= br_nstage(cin, ds) * 512) return;
    const int s = idx >> 9, r = (idx >> 2) & 127, c = idx & 3;
    const unsigned short* src;
    if (s < ns1) {
        src = w1 + (size_t)r * cin + 32 * s + 8 * c;                          // W1 [128][CIN], K slice s
    } else if (s < ns1 + BR_W2_STAGES) {
        const int tap = (s - ns1) >> 2, kc = (s - ns1) & 3;
        src = w2 + ((size_t)tap * 128 + r) * 128 + 32 * kc + 8 * c;           // W2 [9][128][128]
    } else {
        const int q = s - ns1 - BR_W2_STAGES, nh = q / per_nh, k = q % per_nh;
        if (k < 4) src = w3 + ((size_t)nh * 128 + r) * 128 + 32 * k + 8 * c;  // W3 [256][128] (K already permuted by the host packer)
        else src = wd + ((size_t)nh * 128 + r) * cin + 32 * (k - 4) + 8 * c;  // Wd [256][CIN]
    }
    *reinterpret_cast<u32x4*>(stream + (size_t)s * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(src);
}

// 16-bit blob -> the direct-load form of W2' (W2D kernels): block (g, w) = 1 KB = the MFMA A fragment of rows 32 w .. 32 w + 31 for
// group g = (tap * 4 + kc) * 2 + j: lane (l31, half) holds the 8 K values 32 kc + 8 (2 j + half) .. of row 32 w + l31 -- the very
// 16 bytes the ring form reads from chunk 2 j + half of stage image tap * 4 + kc
constexpr int BR_W2D_GROUPS = 2 * BR_W2_STAGES;           // 72
constexpr int BR_W2D_BYTES = BR_W2D_GROUPS * 4 * 1024;    // 294 912
constexpr int BR_W2D_DEPTH = 12;                          // fragments in flight per wave (one per four MFMAs)
__global__ __launch_bounds__(256) void bt_w2d_pack_kernel(const unsigned short* __restrict__ w2, unsigned char* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= BR_W2D_GROUPS * 256) return;
    const int lane = idx & 63, w = (idx >> 6) & 3, g = idx >> 8;
    const int j = g & 1, q = g >> 1, tap = q >> 2, kc = q & 3;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned short* const src = w2 + ((size_t)tap * 128 + 32 * w + l31) * 128 + 32 * kc + 8 * (2 * j + half);   // W2 [9][128][128]
    *reinterpret_cast<u32x4*>(out + (size_t)idx * 16) = *reinterpret_cast<const u32x4*>(src);
}

// LDS-DMA of one 8 KB stage: this wave's two 1 KB pieces (lane l's 16 bytes land at dst + 16 l; dst wave-uniform, in M0).
// sbase (uniform) + voff (per lane, 32 bit) is the source address.  The instruction's immediate offset is added to the
// global address AND to the LDS address, so the second piece needs no second M0 value.  M0 is saved and restored inside
// the statement.
__device__ __forceinline__ void br_glds_stage(const void* sbase, unsigned voff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(dst)
                 : "memory");
}
// one 1 KB LDS-DMA piece: lane l's 16 bytes, read from sbase + voff (voff per lane, 32 bit), land at dst + 16 l (dst wave-uniform)
__device__ __forceinline__ void br_glds_piece(const void* sbase, unsigned voff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(dst)
                 : "memory");
}
// the same with a full 64-bit address per lane (lanes of one piece may read from unrelated places)
__device__ __forceinline__ void br_glds_piece64(const void* vaddr, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(vaddr), "s"(dst)
                 : "memory");
}
// s_waitcnt vmcnt(n) for a value that is a compile-time constant after unrolling (the switch folds away)
__device__ __forceinline__ void br_wait_vm(int n) {
#define BR_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        BR_W(0) BR_W(1) BR_W(2) BR_W(3) BR_W(4) BR_W(5) BR_W(6) BR_W(7) BR_W(8) BR_W(9) BR_W(10) BR_W(11) BR_W(12) BR_W(13)
        BR_W(14) BR_W(15) BR_W(16) BR_W(17) BR_W(18) BR_W(19) BR_W(20) BR_W(21) BR_W(22) BR_W(23) BR_W(24) BR_W(25) BR_W(26)
        BR_W(27) BR_W(28) BR_W(29) BR_W(30) BR_W(31) BR_W(32) BR_W(33) BR_W(34) BR_W(35) BR_W(36) BR_W(37) BR_W(38) BR_W(39) BR_W(40)
        BR_W(41) BR_W(42) BR_W(43) BR_W(44) BR_W(45) BR_W(46) BR_W(47) BR_W(48) BR_W(49) BR_W(50) BR_W(51) BR_W(52) BR_W(53) BR_W(54)
        BR_W(55) BR_W(56) BR_W(57) BR_W(58) BR_W(59) BR_W(60) BR_W(61) BR_W(62) BR_W(63)
        default:   // (the counter has six bits; a smaller count is always safe)
            if (n > 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            break;
    }
#undef BR_W
}
// 16-byte global store with a cache policy: 0 plain, 1 sc1 (written through, the line dropped from this XCD's L2), 2 nt (streaming), 3 sc0 sc1.
// Measured in round 4 on the identity-skip kernel, same box: sc1 -0.7 %, nt -1.2 % of the kernel's time, HBM-side bytes unchanged; MODE 2 stores
// nt (development switch BR_ST_POLICY >= 0 overrides the choice for every mode)
#ifndef BR_ST_POLICY
#define BR_ST_POLICY -1
#endif
template <int POLICY>
__device__ __forceinline__ void br_store16(void* dst, u32x4 v) {
    if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
    else *reinterpret_cast<u32x4*>(dst) = v;
}
#ifndef BR_RET
#define BR_RET 0   // development switch: the workgroup returns at checkpoint n (1 before phase 2's barrier, 2 after its loop, 3 after the t2 crossing, 4 after the
                   // first output half's K loop, 5 after its epilogue, 6 after the second half's K loop): where an ablated kernel's time goes, by wall clock
#endif
#define BR_CHECKPOINT(n) do { if (BR_RET == n && p.V > 0) return; } while (0)
// workgroup barrier that does NOT drain the vector-memory queue (a __syncthreads() beside pending LDS-DMA waits vmcnt(0))
__device__ __forceinline__ void br_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// max(x, 0) on two packed 16-bit floats (bf16 or half: both are sign-magnitude): as signed 16-bit integers a negative float is a negative integer, so one v_pk_max_i16 does
// both halves (and needs no NaN-canonicalising v_max before it, which hipcc puts in front of every fmaxf on an MFMA result).
// NOT inline assembly: hipcc's hazard recognizer does not look inside an asm statement, and a VALU result that the very next
// instruction, an MFMA, reads as SrcA / SrcB needs a wait state in between on gfx950 (tests/perf/ubench/mfma_war.hip shows the
// stale read) -- with the builtin the compiler sees the producer and keeps its distance.
typedef short br_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned br_relu_pk(unsigned packed) {
    const br_i16x2 zero = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(br_i16x2, packed), zero));
}
// bn1 + ReLU on one 16-byte chunk of 8 values: y = max(x * s + t, 0), rounded to T (rounding and max(., 0) commute)
template <typename T>
__device__ __forceinline__ u32x4 br_preact(u32x4 raw, const PreactCoef<T>& k) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = Lp<T>::to_f32((unsigned short)(raw[i] & 0xffffu));
        const float hi = Lp<T>::to_f32((unsigned short)(raw[i] >> 16));
        const float a = fmaf(lo, k.s[i >> 1][(2 * i) & 3], k.t[i >> 1][(2 * i) & 3]);
        const float b = fmaf(hi, k.s[i >> 1][(2 * i + 1) & 3], k.t[i >> 1][(2 * i + 1) & 3]);
        o[i] = br_relu_pk(Lp<T>::pack2(a, b));
    }
    return o;
}

#ifdef DF3D_BT_TIMING
// development build only (scripts/probe_ring.py): per-phase shader-cycle sums of wave 0 of every workgroup
__device__ unsigned long long br_dbg[12];
#define BR_STAMP(k)                                                              \
    do {                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
        if (tid == 0) atomicAdd(&br_dbg[k], now_ - stamp_);                      \
        stamp_ = now_;                                                           \
    } while (0)
#else
#define BR_STAMP(k) do { } while (0)
#endif

// CIN = 256: the identity-skip block (out = ... + x); CIN = 128 (DS): the skip is a 1x1 convolution of the raw input, accumulated
// into the same MFMA accumulators behind W3 (layer2), the x operand of which comes straight from global memory in MFMA layout
// W2D: the 3x3's weights do not go through the ring: in phase 2 wave w computes t2 channels 32 w .. 32 w + 31 for ALL 128 pixels and
// loads its own A fragments straight from global memory into registers (L2 hits, BR_W2D_DEPTH ahead) -- no DMA into LDS, no
// barrier in the phase, four LDS fragment reads per four MFMAs instead of five; t2 then crosses to the pixel-owning waves
// through the dead t1 region.  Same products in the same K order in every accumulator: bit-identical to the ring form.
// MODE: 0 = every weight through the ring; 1 = W2D (round 3's default); 2 = W2D + round 4: phase 3 without DMA round trips on its path
// (FAST, below) and streaming (nt) output stores -- bit-identical to modes 0 / 1 and to the register-staged kernels.
// (Tried in round 4 and NOT kept, because the kernel is bound by the socket's POWER cap, not by its VALU count -- DESIGN.md: bn1 + ReLU as
// packed half arithmetic (v_pk_fma_f16 on coefficients rounded to half: -1.8 % time, three roundings instead of one: 1.3e-3 per block against
// the float32-coefficient form) and on the mixed-precision fma (v_fma_mixlo/hi_f16: no gain, one ulp off in rare elements).)
template <typename T, bool UP, int CIN = 256, bool ADD2 = false, int MODE = 0>
__global__ __launch_bounds__(256, 2) void bottleneck_ring_kernel(BtRingArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage formats only (the fp32 form is hg_bt_ring_f32.h)");
    constexpr bool W2D = MODE >= 1;
    // FAST (MODE 2, identity skip, either 16-bit format): phase 3 without DMA round trips on its path -- each output half's four W3 stages fill the four ring
    // slots a whole phase before they are multiplied (half 0 during phase 2, half 1 during the first half's epilogue), two barriers
    // instead of four; the residual values are requested a phase ahead as well (half 0 before t2 crosses, half 1 before the first K loop)
    constexpr bool FAST = MODE >= 2 && CIN == 256;
    static_assert(!ADD2 || (!UP && CIN == 256), "the fused up-path sum is written by plain identity-skip blocks");
    constexpr int CO = 256, NT = 4;
    constexpr bool DS = CIN != 256;
    static_assert(!(UP && DS), "the upsample-add input exists for the identity-skip block only");
    constexpr int NS1 = CIN / 32;                        // W1 stages = K steps of phase 1
    constexpr int NSD = DS ? CIN / 32 : 0;               // skip-convolution stages per output half
    constexpr int NSTAGE = br_nstage(CIN, DS) - (W2D ? BR_W2_STAGES : 0);   // stages that go through the ring
    constexpr int LX = UP ? 6 : 3;   // vector-memory loads per thread and x step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const t1_lds = smem + BR_RING_BYTES;
    float* const coef_lds = reinterpret_cast<float*>(smem + BR_RING_BYTES + BR_T1_BYTES);
    unsigned long long* const valid_lds = reinterpret_cast<unsigned long long*>(smem + BR_RING_BYTES + BR_T1_BYTES + BR_COEF_BYTES);
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef DF3D_BT_TIMING
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
#if BR_ABLM & 128   // (ablation: the workgroup does nothing at all: what dispatching the grid costs)
    if (p.V > 0) return;
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
    const unsigned char* const xin = reinterpret_cast<const unsigned char*>(p.in) + (size_t)view * p.H * p.W * CIN * 2;
    const unsigned char* const xin2 = UP ? reinterpret_cast<const unsigned char*>(p.in2) + (size_t)view * (p.H / 2) * (p.W / 2) * CIN * 2
                                    : ADD2 ? reinterpret_cast<const unsigned char*>(p.add2) + (size_t)view * (p.H / 2) * (p.W / 2) * CO * 2 : nullptr;

    // ---- the weight ring ----------------------------------------------------------------------------------
    const unsigned wvoff = (unsigned)wave * 2048u + (unsigned)lane * 16u;
    auto ring_issue = [&](int s) {   // stage s -> ring slot s % 4; this wave copies pieces 2 wave, 2 wave + 1
        if (BR_ABLM & 2) return;   // (ablation mask, development builds only: no weight DMA)
        const unsigned dst = ring_addr + (unsigned)(s % BR_RING) * BR_STAGE_BYTES + (unsigned)wave * 2048;
        const int si = (W2D && s >= NS1) ? s + BR_W2_STAGES : s;   // W2D: the ring sequence is W1 | W3 (| Wd), the stream keeps W2 in between
        br_glds_stage(reinterpret_cast<const unsigned char*>(p.wstream) + (size_t)si * BR_STAGE_BYTES, wvoff, dst);
    };
    // fragment addresses inside a stage: rows 32 m + l31, chunk 2 j + half (+ slot * 8192 + m * 2048 as immediates)
    const unsigned char* const wf0 = ring + br_swz(l31, half);
    const unsigned char* const wf1 = ring + br_swz(l31, 2 + half);

    // bn1 coefficients and b1 -> LDS; b2 / b3 wait in two registers and replace the bn1 coefficients after phase 1
    // (the only plain loads before the ring starts); halo validity masks
    //   coef_lds: phase 1: [0..255] scale, [256..511] shift; afterwards [0..127] b2, [128..383] b3;  [512..639] b1
    // (the loads now, the LDS stores behind the first weight stages and x steps: a store in front of them would make the wave sit
    // out the coefficients' round trip before it requests anything else)
    const float pre_s1 = tid < CIN ? p.s1[tid] : 0.0f, pre_t1 = tid < CIN ? p.t1[tid] : 0.0f, pre_b1 = p.b1[tid & 127];
    const float late_b2 = p.b2[tid & 127], late_b3 = DS ? p.b3[tid] + p.bd[tid] : p.b3[tid];
    const float* const b1_lds = coef_lds + 512;
    const float* const b2_lds = coef_lds;
    const float* const b3_lds = coef_lds + 128;
    if (tid < BT_HROWS) {
        const int hy = tid / BT_HW, hx = tid % BT_HW;
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
        const bool ok = tid < BT_HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const unsigned long long m = __ballot(ok);
        if (lane == 0) valid_lds[wave] = m;
    }
#if BR_ABLM & 256   // (ablation: prologue only -- index arithmetic, coefficient loads, halo masks)
    if (tid < CIN) coef_lds[tid] = pre_s1 + pre_t1 + pre_b1 + late_b2 + late_b3;
    if (p.V > 0) return;
#endif
#if defined(BR_ABL) && BR_ABL == 10   // ablation 10: NO phase 1 at all (the t1 tile keeps whatever the LDS holds): what phases 2-3 cost alone -- the
    // upper bound on a 16-bit "tail" kernel fed with t1 from elsewhere.  The ring starts at the W3 stages.
    ring_issue(NS1);
    ring_issue(NS1 + 1);
    ring_issue(NS1 + 2);
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;
    if (tid < 128) coef_lds[tid] = late_b2;
    coef_lds[128 + tid] = late_b3;
    asm volatile("" ::"v"(pre_s1), "v"(pre_t1), "v"(pre_b1));
#else
    ring_issue(0);
    ring_issue(1);
    ring_issue(2);

    // ---- phase 1: t1^T = relu(W1' relu(bn1 x)^T + b1') on the halo --------------------------------------------
    // x staging: thread -> (row = (tid + 256 i) >> 2, 16-byte chunk = tid & 3) of a 32-channel K step
    constexpr int XP = 3;
    const int xchunk = tid & 3;
    const unsigned char* xp[XP];
    const unsigned char* xq[UP ? XP : 1];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int hp = (tid >> 2) + 64 * i;
        const int hy = hp / BT_HW, hx = hp % BT_HW;
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
        const bool ok = hp < BT_HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        xp[i] = xin + (((size_t)(ok ? y : 0) * p.W + (ok ? x : 0)) * CIN + xchunk * 8) * 2;
        if constexpr (UP) xq[i] = xin2 + (((size_t)(ok ? (y >> 1) : 0) * (p.W / 2) + (ok ? (x >> 1) : 0)) * CIN + xchunk * 8) * 2;
    }
    constexpr int DX = UP ? 2 : 3;   // K steps of x requested ahead (registers); the LDS x ring has three slots either way
    u32x4 rx[DX][XP];
    u32x4 rb[UP ? DX : 1][XP];
    auto loadx = [&](int s, int slot) {   // out-of-image halo rows read pixel (0, 0) (a valid address); the t1 epilogue zeroes what comes of them
#pragma unroll
#if (defined(BR_ABL) && BR_ABL == 5) || (BR_ABLM & 4)   // ablation: no x loads (registers only)
        for (int i = 0; i < XP; ++i) rx[slot][i] = u32x4{(unsigned)s, (unsigned)tid, 0u, 0u};
#else
        for (int i = 0; i < XP; ++i) rx[slot][i] = *reinterpret_cast<const u32x4*>(xp[i] + s * 64);
#endif
        if constexpr (UP) {
#pragma unroll
            for (int i = 0; i < XP; ++i) rb[slot][i] = *reinterpret_cast<const u32x4*>(xq[i] + s * 64);
        }
    };
    auto storex = [&](int s, int slot) {
        PreactCoef<T> coef;
        const int c0 = s * 32 + xchunk * 8;
        coef.s[0] = *reinterpret_cast<const f32x4*>(coef_lds + c0);
        coef.s[1] = *reinterpret_cast<const f32x4*>(coef_lds + c0 + 4);
        coef.t[0] = *reinterpret_cast<const f32x4*>(coef_lds + 256 + c0);
        coef.t[1] = *reinterpret_cast<const f32x4*>(coef_lds + 256 + c0 + 4);
        unsigned char* const sx = t1_lds + (s % 3) * BR_XSTAGE;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            u32x4 v = rx[slot][i];
            if constexpr (UP) v = add_chunk<T>(v, rb[slot][i]);   // x = in + upsample(in2), rounded like upadd_kernel's output
#if !defined(BR_ABL) || BR_ABL != 6   // ablation 6: no bn1 + ReLU arithmetic
            v = br_preact<T>(v, coef);
#endif
            *reinterpret_cast<u32x4*>(sx + ((tid >> 2) + 64 * i) * BR_XPITCH + br_xslot(tid >> 2, xchunk)) = v;
        }
    };
#pragma unroll
    for (int k = 0; k < DX; ++k) loadx(k, k);
    if (tid < CIN) {
        coef_lds[tid] = pre_s1;
        coef_lds[256 + tid] = pre_t1;
    }
    if (tid < 128) coef_lds[512 + tid] = pre_b1;

    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;   // this wave's 32 pixels (phases 2, 3)
    {
        // wave w owns t1 channels 32 w .. 32 w + 31 for all six halo row tiles.  The product is formed TRANSPOSED (A = W1
        // rows, B = x rows: D[channel][halo pixel]) so that a lane ends up with four consecutive channels of ONE pixel per
        // register group -> 8-byte LDS stores into the t1 tile, no cross-lane traffic.  Same products, same K order.
        const int ct = wave;
        f32x16 acc[6];
        br_barrier();   // coefficients, b1 and masks visible
        BR_STAMP(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // register 4 t + e <-> channel 32 ct + 8 t + 4 half + e
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b1_lds + ct * 32 + 8 * t + 4 * half);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][4 * t + e] = bb[e];
        }
#pragma unroll
        for (int s = 0; s < NS1; ++s) {
            storex(s, s % DX);
            // operations issued after stage s's DMA pieces (see the file header): the next two stages' pieces (4) and the x
            // loads requested since -- the prologue's DX steps for s < 3, then one step's worth per K step while any remain
            auto cx = [](int k) { return k < NS1 ? LX : 0; };
            br_wait_vm(s == 0 ? 4 + DX * LX : s == 1 ? 4 + DX * LX + cx(DX) : s == 2 ? 4 + DX * LX + cx(DX) + cx(DX + 1)
                                                                               : 4 + cx(s - 3 + DX) + cx(s - 2 + DX) + cx(s - 1 + DX));
            br_barrier();
            ring_issue(s + 3);
            if (s + DX < NS1) loadx(s + DX, s % DX);
            const unsigned char* const sx = t1_lds + (s % 3) * BR_XSTAGE;
            // both K halves' fragments are requested before the first MFMA (see phase 2: hipcc would serialise read -> MFMA)
            u32x4 wfr[2], xfr[2][6];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wfr[j] = *reinterpret_cast<const u32x4*>((j ? wf1 : wf0) + (s % BR_RING) * BR_STAGE_BYTES + ct * 2048);
#pragma unroll
                for (int i = 0; i < 6; ++i) xfr[j][i] = *reinterpret_cast<const u32x4*>(sx + (i * 32 + l31) * BR_XPITCH + br_xslot(l31, 2 * j + half));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
#if (defined(BR_ABL) && BR_ABL == 9) || (BR_ABLM & 1)   // ablation 9: no MFMAs anywhere (the memory floor of the kernel's access pattern)
                for (int i = 0; i < 6; ++i) asm volatile("" ::"v"(wfr[j]), "v"(xfr[j][i]));
#else
                for (int i = 0; i < 6; ++i) mfma_chunk<T>(wfr[j], xfr[j][i], acc[i]);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        br_barrier();   // every wave is done with the x ring: the t1 tile may overwrite it
        if constexpr (FAST) ring_issue(NS1 + 3);   // W3's first half complete in the ring: the slot held the last W1 stage, which this barrier has released
        BR_STAMP(1);
        // epilogue: ReLU (the bias was the start value), zero outside the image, 8-byte stores into the swizzled t1 tile
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int hp = i * 32 + l31;
            const unsigned keep = 0u - (unsigned)((valid_lds[i >> 1] >> ((i & 1) * 32 + l31)) & 1ull);
            unsigned char* const trow = t1_lds + hp * BR_T1_PITCH + half * 8;
            const int sw = br_t1_swz(hp);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint2 w;
                w.x = br_relu_pk(Lp<T>::pack2(acc[i][4 * t + 0], acc[i][4 * t + 1])) & keep;
                w.y = br_relu_pk(Lp<T>::pack2(acc[i][4 * t + 2], acc[i][4 * t + 3])) & keep;
                if (i < 5 || hp < BT_HALO) *reinterpret_cast<uint2*>(trow + (((ct * 4 + t) ^ sw) << 4)) = w;
            }
        }
        // the bn1 coefficients are dead: b2 / b3 take their place (read after the next barrier)
        if (tid < 128) coef_lds[tid] = late_b2;
        coef_lds[128 + tid] = late_b3;
    }

#endif
    BR_STAMP(2);
    // ---- phase 2: t2^T = W2' (*) t1 (fully unrolled: every LDS address is one register + an immediate) -----------
    f32x16 t2[NT];
    const unsigned char* const t1_lane = t1_lds + (py * BT_HW + px) * BR_T1_PITCH;
    unsigned tsw[3];   // ((tile column + kx) & 15 ^ half) << 4: the swizzle term of this lane's t1 fragment, per kx
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) tsw[kx] = (unsigned)((((px + kx) & 15) ^ half) << 4);
    u32x4 t2f[NT][2];   // ReLU(t2) rounded to 16 bits: the B operands of phase 3 (tile kc, registers 8 q2 .. 8 q2 + 7 -> four dwords)
    // identity skip: the residual values of the two 128-channel output halves: lane owns, for c = 0..7, chunk (lane & 15) of wave pixel 4 c + (lane >> 4)
    unsigned xres[2][DS ? 1 : 32];
    auto load_res = [&](int nh) {
        if constexpr (!DS) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int pw = 4 * c + (lane >> 4);
                const u32x4 v = (BR_ABLM & 8) ? u32x4{(unsigned)pw, (unsigned)tid, 0u, 0u} : *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(xin) +
                    ((size_t)(ty0 + 2 * wave + (pw >> 4)) * p.W + (tx0 + (pw & 15))) * CIN + nh * 128 + (lane & 15) * 8);
                xres[nh][4 * c + 0] = v[0];
                xres[nh][4 * c + 1] = v[1];
                xres[nh][4 * c + 2] = v[2];
                xres[nh][4 * c + 3] = v[3];
            }
        }
    };
    if constexpr (W2D) {
        constexpr int NG = BR_W2D_GROUPS, D = BR_W2D_DEPTH;
        static_assert(NG % D == 0, "the fragment queue's slots repeat");
        const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.w2d) + ((size_t)wave * 64 + lane) * 16;   // group g: + 4096 g
        u32x4 wq[D];
#pragma unroll
        for (int k = 0; k < D; ++k) wq[k] = *reinterpret_cast<const u32x4*>(wsrc + (size_t)k * 4096);
        BR_CHECKPOINT(1);
        br_barrier();   // publishes the t1 tile and b2 / b3 (the ring rests: phase 1 has requested the first three W3 stages)
#pragma unroll
        for (int m = 0; m < NT; ++m)   // accumulator m = pixel tile m (tile rows 2 m, 2 m + 1); register 4 q + e <-> channel 32 wave + 8 q + 4 half + e
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b2_lds + 32 * wave + 8 * q + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) t2[m][4 * q + e] = bb[e];
            }
        const unsigned char* const t1_px = t1_lds + ((l31 >> 4) * BT_HW + px) * BR_T1_PITCH;   // pixel l31 of tile 0; tile m: + m * 2 * BT_HW rows
#ifndef BR_P2_DEPTH
#define BR_P2_DEPTH 1   // development switch: t1 fragment groups requested ahead of the MFMAs that consume them
#endif
        constexpr int TD = BR_P2_DEPTH + 1;
        u32x4 tfr[TD][NT];
        auto load_t = [&](int g, int buf) {
            const int j = g & 1, q = g >> 1, tap = q >> 2, kc = q & 3;
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int m = 0; m < NT; ++m)
                tfr[buf][m] = *reinterpret_cast<const u32x4*>(t1_px + ((2 * m + ky) * BT_HW + kx) * BR_T1_PITCH + (tsw[kx] ^ (unsigned)((4 * kc + 2 * j) << 4)));
        };
#pragma unroll
        for (int k = 0; k < BR_P2_DEPTH; ++k) load_t(k, k);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + BR_P2_DEPTH < NG && !((BR_ABLM & 64) && g >= 2)) load_t(g + BR_P2_DEPTH, (g + BR_P2_DEPTH) % TD);   // (ablation mask 64: no t1 fragment reads after the first groups)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                if (BR_ABLM & 512) asm volatile("" ::"v"(wq[g % D]), "v"(tfr[g % TD][m]));   // (ablation mask 512: no phase-2 MFMAs in the W2D form)
                else mfma_chunk<T>(wq[g % D], tfr[g % TD][m], t2[m]);
            }
            __builtin_amdgcn_sched_barrier(0);
#if !(BR_ABLM & 32)   // (ablation mask 32: the W2 fragments are loaded once, the first BR_W2D_DEPTH of them, and re-used)
            if (g + D < NG) wq[g % D] = *reinterpret_cast<const u32x4*>(wsrc + (size_t)(g + D) * 4096);
#endif
        }
        BR_STAMP(8);
        BR_CHECKPOINT(2);
        if constexpr (FAST) load_res(0);   // (every W2 fragment has been consumed: nothing of this wave's is queued in front of these loads)
        // t2 crosses to the waves that own the pixels in phase 3: 128 rows x 256 B in the dead t1 region; chunk (kc, q2, half) of a
        // row = the 16 bytes t2f[kc][q2] of lane (pixel, half), in slot chunk ^ (row & 15)
        br_barrier();   // every wave is done reading t1
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint2 w;
                w.x = br_relu_pk(Lp<T>::pack2(t2[m][4 * t + 0], t2[m][4 * t + 1]));
                w.y = br_relu_pk(Lp<T>::pack2(t2[m][4 * t + 2], t2[m][4 * t + 3]));
                const int chunk = wave * 4 + (t >> 1) * 2 + half;
                *reinterpret_cast<uint2*>(t1_lds + (32 * m + l31) * 256 + ((chunk ^ (l31 & 15)) << 4) + (t & 1) * 8) = w;
            }
        br_barrier();
#pragma unroll
        for (int kc = 0; kc < NT; ++kc)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
                t2f[kc][q2] = *reinterpret_cast<const u32x4*>(t1_lds + (32 * wave + l31) * 256 + (((kc * 4 + q2 * 2 + half) ^ (l31 & 15)) << 4));
    } else {
    // two stages per barrier: the pair (s0, s0 + 1) was requested one double-step earlier, the pair after it goes into the two
    // slots the previous double-step has just released
#pragma unroll
    for (int d = 0; d < BR_W2_STAGES / 2; ++d) {
        const int s0 = NS1 + 2 * d;
#if defined(BR_ABL) && (BR_ABL == 7 || BR_ABL == 8)   // ablation: every wave waits for its own DMA pieces, no barrier (timing of a private-ring phase 2)
        br_wait_vm(d == 0 ? 2 : 0);
        if (d == 0) br_barrier();
#elif !defined(BR_ABL) || BR_ABL != 4
        br_wait_vm(d == 0 ? 2 : 0);   // d = 0: phase 1 has already requested stage 10
        br_barrier();                 // (first iteration: also publishes the t1 tile and b2 / b3)
#else
        if (d == 0) { br_wait_vm(2); br_barrier(); }
#endif
#if !defined(BR_ABL) || BR_ABL != 3
        if (d > 0) ring_issue(s0 + 2);
        ring_issue(s0 + 3);
#endif
        if (d == 0) {   // t2 accumulators start at b2' (channel of register r in tile m: 32 m + (r & 3) + 8 (r >> 2) + 4 half)
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b2_lds + 32 * m + 8 * q + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t2[m][4 * q + e] = bb[e];
                }
        }
        // The four (stage, K half) groups of the double-step, software-pipelined by hand: the five fragments of group g + 1 are
        // requested BEFORE the four MFMAs of group g (hipcc otherwise re-uses one register quad for every weight fragment and
        // serialises ds_read -> wait -> MFMA, which leaves the LDS latency exposed in front of every MFMA).
        u32x4 tfr[2], wfr[2][NT];
        auto load_group = [&](int g, int buf) {
            const int s = s0 + (g >> 1), j = g & 1;
            const int q = s - NS1, tap = q >> 2, kc = q & 3;
            const int ky = tap / 3, kx = tap - 3 * ky;
            // chunk (4 kc + 2 j + half) ^ swizzle = ((4 kc + 2 j) << 4) ^ tsw[kx]   (4 kc + 2 j is even)
            tfr[buf] = *reinterpret_cast<const u32x4*>(t1_lane + (ky * BT_HW + kx) * BR_T1_PITCH + (tsw[kx] ^ (unsigned)((4 * kc + 2 * j) << 4)));
#pragma unroll
            for (int m = 0; m < NT; ++m)
                wfr[buf][m] = *reinterpret_cast<const u32x4*>((j ? wf1 : wf0) + (s % BR_RING) * BR_STAGE_BYTES + m * 2048);
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) load_group(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#ifdef BR_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int m = 0; m < NT; ++m) {
#if (defined(BR_ABL) && (BR_ABL == 1 || BR_ABL == 9)) || (BR_ABLM & 1)   // ablation: no MFMAs (the fragments stay live)
                asm volatile("" ::"v"(wfr[g & 1][m]), "v"(tfr[g & 1]));
#else
                mfma_chunk<T>(wfr[g & 1][m], tfr[g & 1], t2[m]);
#endif
            }
#ifdef BR_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ReLU + rounding to bf16 once
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
            for (int e = 0; e < 4; ++e) t2f[m][q2][e] = br_relu_pk(Lp<T>::pack2(t2[m][8 * q2 + 2 * e], t2[m][8 * q2 + 2 * e + 1]));
    }

    BR_CHECKPOINT(3);
    BR_STAMP(3);
    // ---- phase 3: out^T = W3 t2^T + b3 (+ x) -------------------------------------------------------------------
    // transposed like phase 1 (A = W3 rows, B = the t2 registers): accumulator register 4 t + e of channel tile i holds, for
    // pixel l31 of the wave, output channel 128 nh + 32 i + 8 t + 4 half + e -> 8-byte stores into the wave's LDS slice
    unsigned char* const outp = reinterpret_cast<unsigned char*>(p.out) + (size_t)view * p.H * p.W * CO * 2;
    constexpr int S3 = W2D ? NS1 : NS1 + BR_W2_STAGES;   // first W3 stage (ring numbering)
    constexpr int PER_NH = 4 + NSD;          // stages per 128-channel output half: W3 (4), then the skip convolution's (DS)
    // DS: the raw input of this lane's pixel as MFMA B operands -- K chunk kc = channels 16 kc + 8 half .. (requested once, in the
    // first double-step, straight from global memory: tile pixel (2 wave + (l31 >> 4), l31 & 15))
    u32x4 xc[DS ? CIN / 16 : 1];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
        f32x16 acc[4];
#pragma unroll
        for (int dd = 0; dd < PER_NH / 2; ++dd) {
            const int s0 = S3 + PER_NH * nh + 2 * dd;
            // operations issued after the pair's DMA pieces: identity skip: the 8 residual loads of this half (requested in its
            // first double-step, after the DMA), or the first half's epilogue (8 stores, UP: 4 loads; the optional pool stores are
            // left out, which only makes the wait conservative); DS: the CIN / 16 input loads of the very first double-step, the
            // first half's 8 stores in front of the second half
            constexpr int E0 = 8 + ((UP || ADD2) ? 4 : 0);
            if constexpr (FAST) {
                // the half's four stages were requested a phase ago.  Half 0: this wave's pieces landed before its last W2 fragment did
                // (returns are in order) and every wave has passed the t2 crossing's barriers since; half 1: requested behind the first
                // K loop, E0 operations of the first epilogue behind them
                if (dd == 0 && nh == 1) {
                    br_wait_vm(E0);
                    br_barrier();
                }
            } else {
            if constexpr (DS) br_wait_vm(dd == 0 ? (nh == 0 ? 0 : 8) : (dd == 1 && nh == 0) ? CIN / 16 : 0);
            else br_wait_vm(dd == 1 ? 8 : nh == 0 ? 0 : E0);
#if defined(BR_ABL) && BR_ABL == 8
            if (dd == 0 && nh == 0) br_barrier();
#else
            br_barrier();
#endif
            if (s0 + 3 < NSTAGE) {
                if (!(W2D && nh == 0 && dd == 0)) ring_issue(s0 + 2);   // (W2D: phase 1, three stages ahead, has requested it)
                ring_issue(s0 + 3);
            }
            }
            if (dd == 0) {
                if constexpr (DS) {
                    if (nh == 0) {
                        const unsigned char* const src = xin + (((size_t)(ty0 + py) * p.W + (tx0 + px)) * CIN + half * 8) * 2;
#pragma unroll
                        for (int kc = 0; kc < CIN / 16; ++kc) xc[kc] = *reinterpret_cast<const u32x4*>(src + kc * 32);
                    }
                } else {
                    if constexpr (FAST) {
                        if (nh == 0) load_res(1);   // (half 0: before t2 crossed)
                    } else {
                        load_res(nh);   // residual values requested now
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(b3_lds + nh * 128 + i * 32 + 8 * t + 4 * half);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][4 * t + e] = bb[e];
                    }
            }
            // four (stage, K half) groups, the weight fragments of group g + 1 requested before the MFMAs of group g.  W3 steps
            // (dd < 2): t2 tile kc, registers 8 q2 .. 8 q2 + 7 <-> packed W3 K positions 32 kc + 16 q2 + 8 half .. (host K order,
            // kperm); skip-convolution steps (dd >= 2): K chunk 2 (stage) + q2 of the raw input
            u32x4 w3r[2][4];
            auto load_w3 = [&](int g, int buf) {
                const int s = s0 + (g >> 1), q2 = g & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    w3r[buf][i] = *reinterpret_cast<const u32x4*>((q2 ? wf1 : wf0) + (s % BR_RING) * BR_STAGE_BYTES + i * 2048);
            };
            load_w3(0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) load_w3(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 tf = dd < 2 ? t2f[2 * dd + (g >> 1)][g & 1] : xc[DS ? 2 * (2 * (dd - 2) + (g >> 1)) + (g & 1) : 0];
#pragma unroll
#if (defined(BR_ABL) && BR_ABL == 9) || (BR_ABLM & 1)
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(w3r[g & 1][i]), "v"(tf));
#else
                for (int i = 0; i < 4; ++i) acc[i] = Lp<T>::mfma(w3r[g & 1][i], tf, acc[i]);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (FAST) {
            if (nh == 0) {   // every wave is done with the four slots: the second half's stages arrive while the first half's epilogue runs
                br_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) ring_issue(S3 + 4 + k);
            }
        }
        BR_STAMP(4 + 2 * nh);
        if (nh == 0) BR_CHECKPOINT(4);
        else BR_CHECKPOINT(6);
        // epilogue through LDS (the t1 region is dead): every wave parks its 32 px x 128 ch tile in its own slice (8-byte stores:
        // a lane owns four consecutive channels of its pixel per register group) and streams it out as 16-byte chunks with the
        // residual added; rows fully coalesced.  Only this wave touches its slice.
        constexpr int OP = 128 * 2 + 16;
        unsigned char* const slice = t1_lds + wave * (32 * OP);
        u32x4 x2[(UP || ADD2) ? 4 : 1];
        if constexpr (UP || ADD2) {   // the low-resolution addend: UP of the residual (the block's input is the sum), ADD2 of the output
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int pw = 4 * c + (lane >> 4);
                x2[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(xin2) +
                    ((size_t)(ty0 / 2 + wave) * (p.W / 2) + ((tx0 + (pw & 15)) >> 1)) * CIN + nh * 128 + (lane & 15) * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint2 w;
                w.x = Lp<T>::pack2(acc[i][4 * t + 0], acc[i][4 * t + 1]);
                w.y = Lp<T>::pack2(acc[i][4 * t + 2], acc[i][4 * t + 3]);
                *reinterpret_cast<uint2*>(slice + l31 * OP + (i * 32 + 8 * t + 4 * half) * 2) = w;
            }
        unsigned short* const outs = reinterpret_cast<unsigned short*>(outp);
        u32x4 fin[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int pw = 4 * c + (lane >> 4);
            u32x4 v = *reinterpret_cast<const u32x4*>(slice + pw * OP + (lane & 15) * 16);
            if constexpr (!DS) {   // identity skip (DS: the skip convolution is already in the accumulators)
                u32x4 x4 = {xres[nh][4 * c], xres[nh][4 * c + 1], xres[nh][4 * c + 2], xres[nh][4 * c + 3]};
                if constexpr (UP) x4 = add_chunk<T>(x4, x2[c & 3]);
                v = add_chunk<T>(v, x4);
                xres[nh][4 * c] = x4[0], xres[nh][4 * c + 1] = x4[1], xres[nh][4 * c + 2] = x4[2], xres[nh][4 * c + 3] = x4[3];   // (the block's input, for pool_in)
            }
            if constexpr (ADD2) v = add_chunk<T>(v, x2[c & 3]);   // the rounded block output + the low-resolution tensor, rounded again
            fin[c] = v;
            if ((BR_ABLM & 16) && v[0] != 0x12345678u) continue;   // (ablation: practically no output stores)
            br_store16<(BR_ST_POLICY >= 0 ? BR_ST_POLICY : MODE >= 2 ? 2 : 0)>(outs + ((size_t)(ty0 + 2 * wave + (pw >> 4)) * p.W + (tx0 + (pw & 15))) * CO + nh * 128 + (lane & 15) * 8, v);
        }
        if (!DS && p.pool_in) {   // same lane geometry as the pooling of `out` below
            unsigned short* const pp = reinterpret_cast<unsigned short*>(p.pool_in) + (size_t)view * (p.H / 2) * (p.W / 2) * CIN;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 xa = {xres[nh][DS ? 0 : 4 * c], xres[nh][DS ? 0 : 4 * c + 1], xres[nh][DS ? 0 : 4 * c + 2], xres[nh][DS ? 0 : 4 * c + 3]};
                const u32x4 xb = {xres[nh][DS ? 0 : 4 * c + 16], xres[nh][DS ? 0 : 4 * c + 17], xres[nh][DS ? 0 : 4 * c + 18], xres[nh][DS ? 0 : 4 * c + 19]};
                u32x4 m = max_chunk<T>(xa, xb);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __shfl_xor(m[e], 16, 64);
                m = max_chunk<T>(m, o);
                if (((lane >> 4) & 1) == 0)
                    *reinterpret_cast<u32x4*>(pp + ((size_t)(ty0 / 2 + wave) * (p.W / 2) + (tx0 / 2 + 2 * c + (lane >> 5))) * CIN + nh * 128 + (lane & 15) * 8) = m;
            }
        }
        if (p.pool) {
            unsigned short* const pp = reinterpret_cast<unsigned short*>(p.pool) + (size_t)view * (p.H / 2) * (p.W / 2) * CO;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32x4 m = max_chunk<T>(fin[c], fin[c + 4]);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __shfl_xor(m[e], 16, 64);
                m = max_chunk<T>(m, o);
                if (((lane >> 4) & 1) == 0)
                    *reinterpret_cast<u32x4*>(pp + ((size_t)(ty0 / 2 + wave) * (p.W / 2) + (tx0 / 2 + 2 * c + (lane >> 5))) * CO + nh * 128 + (lane & 15) * 8) = m;
            }
        }
        BR_STAMP(5 + 2 * nh);
        if (nh == 0) BR_CHECKPOINT(5);
    }
}

}  // namespace hgk
