// Device kernels of the stacked-hourglass engine (a2).  NHWC activations, gfx950 MFMA.
//
//   conv_mfma_kernel<T, TAPS, BN, RB>
//       1x1 (TAPS=1) and 3x3/pad 1 (TAPS=9) convolutions as an implicit GEMM
//           out[m, n] = sum_{tap, c} act(in[pixel(m) + tap, c]) * W[tap][n][c]  (+ bias, + residual, ReLU)
//       with m = (view, y, x) flattened.  Workgroup tile 128 pixels x BN channels, 4 wavefronts, each owning
//       32x32 MFMA tiles (v_mfma_f32_32x32x2_f32 for T=float: exact f32 FMA chains at the 157 TF rate;
//       v_mfma_f32_32x32x16_bf16 / _f16 for the 16-bit engines, T = __hip_bfloat16 / _Float16: see Lp<T>).  Per K-step both operands are staged global -> registers -> LDS
//       as RB-byte row segments (16-byte chunks), double-buffered with one barrier per step; the global loads
//       for step s+1 are issued before the MFMAs of step s.  Rows are padded by 16 bytes in LDS, which makes
//       the 16-byte ds_read of a 32-row fragment conflict-free (row pitch 80 B / 144 B: see DESIGN.md).
//       A 16-byte fragment holds 4 (f32) or 8 (bf16) consecutive K values of one row; lanes 0-31 take the
//       even 16-byte chunk and lanes 32-63 the odd one, which permutes K inside the step identically for
//       both operands (the sum over K is unchanged).
//       Fusions: eval-mode BN + ReLU of the *input* (pre-activation bottlenecks) while staging; bias, residual
//       add and ReLU in the epilogue; optional NCHW plane output for the final heat-maps.
//   stem_kernel       7x7 stride-2 convolution 3 -> 64 (+ folded BN, ReLU) from an LDS-resident input patch.
//   pool2_kernel      2x2 max-pool;  upadd_kernel  out = a + nearest-upsample2(b).   HBM-bound, 16-byte lanes.
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <type_traits>

#include "preprocess_math.h"

namespace hgk {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

struct ConvArgs {
    const void* in;
    void* out;          // NHWC [M, out_pitch] (may be null when only the NCHW output is wanted)
    const void* res;    // NHWC residual [M, res_pitch] or null
    float* out_nchw;    // float32 planes [views, cout_real, H*W] or null
    const void* w;      // [TAPS][cout][cin]  (T)
    const float* bias;  // [cout] f32
    const float* scale; // [cin] f32 or null: input BN as x*scale+shift then ReLU
    const float* shift;
    long long M;        // views * H * W
    int H, W;
    int cin, cout;      // padded: cin multiple of RB/sizeof(T), cout multiple of BN
    int in_pitch, out_pitch, res_pitch;
    int relu;
    int cout_real;
};

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// The 16-bit storage formats of the low-precision engines.  Both carry a sign, so one bit pattern trick serves both (ReLU and
// max as signed 16-bit integers: br_relu_pk, bf16x2_key); both multiply on the matrix cores at the same rate with fp32
// accumulation.  What differs is where the 16 bits go:
//   __hip_bfloat16  8 exponent bits (fp32's range), 8 significant bits: every stored value carries 2^-9 relative rounding
//   _Float16        IEEE half: 11 significant bits (2^-12 relative, eight times finer), range 6.1e-5 .. 65 504 -- ample for
//                   this network's batch-normalised activations and O(1) weights
// Conversions round to nearest even through the hardware converters (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950).
template <typename T>
struct Lp;
template <>
struct Lp<__hip_bfloat16> {
    static __device__ __forceinline__ float to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
    static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    static __device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <>
struct Lp<_Float16> {
    static __device__ __forceinline__ float to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
    static __device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
};
// 16-byte global store of an activation chunk; HG_NT_STORES (development switch, default 0) makes it a streaming (nt) store in the stack
// heads and the 16-bit layer1 kernel, as the ring bottleneck's MODE 2 does (hg_bt_ring.h: -1.2 % there)
#ifndef HG_NT_STORES
#define HG_NT_STORES 0
#endif
using hg_u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ void hg_store16(void* dst, hg_u32x4 v) {
#if HG_NT_STORES
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
#else
    *reinterpret_cast<hg_u32x4*>(dst) = v;
#endif
}

// eight consecutive floats -> one 16-byte MFMA operand chunk
template <typename T>
__device__ __forceinline__ u32x4 lp_pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    return u32x4{Lp<T>::pack2(a0, a1), Lp<T>::pack2(a2, a3), Lp<T>::pack2(a4, a5), Lp<T>::pack2(a6, a7)};
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int BYTES = 4;
    static constexpr int PER16 = 4;  // elements per 16-byte chunk
};
// "f32s" (round 5): float32 STORAGE everywhere -- the fp32 engine's tensors, weights, streams, LDS images and kernels, bit for bit --
// with every product formed on the 16-bit matrix pipe as a two-way IEEE-half split (mfma_chunk<F32S>).  A tag type: sizeof == 4 sends it
// down the float32 path of every helper and kernel; only the MFMA differs.
struct F32S {
    float v;
};
template <>
struct Elem<F32S> {
    static constexpr int BYTES = 4;
    static constexpr int PER16 = 4;
};
template <>
struct Elem<__hip_bfloat16> {
    static constexpr int BYTES = 2;
    static constexpr int PER16 = 8;
};
template <>
struct Elem<_Float16> {
    static constexpr int BYTES = 2;
    static constexpr int PER16 = 8;
};

// Input BatchNorm + ReLU, y = max(x*s + t, 0), on one 16-byte chunk of channels starting at channel c.
// Split in two so the scale/shift loads can be issued with the prefetch (PreactCoef::load) and the arithmetic
// happens after the MFMAs of the current step (preact_apply).
template <typename T>
struct PreactCoef {
    static constexpr int N = Elem<T>::PER16 / 4;  // float4 groups: 1 (f32) or 2 (bf16)
    f32x4 s[N], t[N];
    __device__ __forceinline__ void load(const float* __restrict__ scale, const float* __restrict__ shift, int c) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            s[i] = *reinterpret_cast<const f32x4*>(scale + c + 4 * i);
            t[i] = *reinterpret_cast<const f32x4*>(shift + c + 4 * i);
        }
    }
};

template <typename T>
__device__ __forceinline__ u32x4 preact_apply(u32x4 raw, const PreactCoef<T>& k) {
    if constexpr (sizeof(T) == 4) {
        f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(fmaf(v[i], k.s[0][i], k.t[0][i]), 0.0f);
        return __builtin_bit_cast(u32x4, v);
    } else {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float lo = Lp<T>::to_f32((unsigned short)(raw[i] & 0xffffu));
            const float hi = Lp<T>::to_f32((unsigned short)(raw[i] >> 16));
            const float a = fmaxf(fmaf(lo, k.s[i >> 1][(2 * i) & 3], k.t[i >> 1][(2 * i) & 3]), 0.0f);
            const float b = fmaxf(fmaf(hi, k.s[i >> 1][(2 * i + 1) & 3], k.t[i >> 1][(2 * i + 1) & 3]), 0.0f);
            o[i] = Lp<T>::pack2(a, b);
        }
        return o;
    }
}

// float32 x -> IEEE-half (hi, lo) with x = hi + lo to 2^-22 |x|: hi = rn(x), lo = rn(x - hi) (the difference is exact in float32).  gfx950's
// v_mfma_f32_32x32x16_f16 keeps subnormal inputs (tests/perf/ubench/mfma_f16_denorm.hip), so lo needs no scaling for small |x|; |x| up to the
// half range 65 504 (the hourglass' batch-normalised activations and O(1) weights are nowhere near it).  Two floats per call: packed converters.
__device__ __forceinline__ void f32s_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {x0, x1};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r = {x0 - (float)h[0], x1 - (float)h[1]};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// ---- F32S: float32 products on the half-precision matrix pipe (round 5) ---------------------------------------------------------------
// A lane's share of one 16-float K step is EIGHT floats: the step's 16-byte chunks `half` and `2 + half` (every float32 kernel reads them as
// fragment j = 0 and j = 1 of the step).  With x = hi + lo (f32s_split2: two IEEE halves, x to 2^-22 |x|) the step's product is three K = 16
// half-precision MFMAs -- w_hi x_hi + w_lo x_hi + w_hi x_lo, float32 accumulation; the w_lo x_lo term (2^-22 of the product) is dropped --
// where the exact-fp32 v_mfma_f32_32x32x2_f32 takes eight: 96 matrix-pipe cycles instead of 512.
//   * WEIGHTS are stored pre-split (f32s_presplit_kernel at df3d_hg_set_weights: the blob's copy, from which every stream / LDS image is then
//     packed -- the packers move whole chunks and keep a chunk's index inside its step): per 64-byte step of a row, chunk 0 = hi(f0..3, f8..11),
//     chunk 1 = hi(f4..7, f12..15), chunk 2 = lo(f0..3, f8..11), chunk 3 = lo(f4..7, f12..15) -- so the fragment a lane reads at j = 0 IS the
//     MFMA operand w_hi of its eight K values and the one at j = 1 is w_lo: no arithmetic, no register moves.
//   * ACTIVATIONS are split where they are used, once per fragment pair (make_xpair), however many weight fragments the pair then meets.
__global__ __launch_bounds__(256) void f32s_presplit_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nsteps) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nsteps; i += (size_t)gridDim.x * 256) {
        f32x4 f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) f[c] = __builtin_bit_cast(f32x4, src[4 * i + c]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // lane half h owns floats 4 h .. 4 h + 3 and 8 + 4 h .. 8 + 4 h + 3 of the step
            unsigned hi[4], lo[4];
            f32s_split2(f[h][0], f[h][1], hi[0], lo[0]);
            f32s_split2(f[h][2], f[h][3], hi[1], lo[1]);
            f32s_split2(f[2 + h][0], f[2 + h][1], hi[2], lo[2]);
            f32s_split2(f[2 + h][2], f[2 + h][3], hi[3], lo[3]);
            dst[4 * i + h] = u32x4{hi[0], hi[1], hi[2], hi[3]};
            dst[4 * i + 2 + h] = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
    }
}

// the activation operand of one K step: float / 16-bit kernels keep the two chunks as loaded; F32S holds a = the hi halves, b = the lo halves
template <typename T>
struct XPair {
    u32x4 a, b;
};
template <typename T>
__device__ __forceinline__ XPair<T> make_xpair(const u32x4& x0, const u32x4& x1) {
    if constexpr (std::is_same<T, F32S>::value) {
        const f32x4 f0 = __builtin_bit_cast(f32x4, x0), f1 = __builtin_bit_cast(f32x4, x1);
        unsigned hi[4], lo[4];
        f32s_split2(f0[0], f0[1], hi[0], lo[0]);
        f32s_split2(f0[2], f0[3], hi[1], lo[1]);
        f32s_split2(f1[0], f1[1], hi[2], lo[2]);
        f32s_split2(f1[2], f1[3], hi[3], lo[3]);
        return XPair<T>{u32x4{hi[0], hi[1], hi[2], hi[3]}, u32x4{lo[0], lo[1], lo[2], lo[3]}};
    } else {
        return XPair<T>{x0, x1};
    }
}
template <typename T>
__device__ __forceinline__ XPair<T> make_xpair(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    return make_xpair<T>(__builtin_bit_cast(u32x4, f32x4{a0, a1, a2, a3}), __builtin_bit_cast(u32x4, f32x4{a4, a5, a6, a7}));
}

// one 16-byte A fragment x one 16-byte B fragment -> accumulate into a 32x32 tile (float32: K = 8, 16-bit: K = 16)
template <typename T>
__device__ __forceinline__ void mfma_chunk(const u32x4& a, const u32x4& b, f32x16& acc) {
    static_assert(!std::is_same<T, F32S>::value, "F32S kernels multiply whole K steps (mfma_pair): a single pre-split weight chunk is half an operand");
    if constexpr (sizeof(T) == 4) {
        const f32x4 af = __builtin_bit_cast(f32x4, a);
        const f32x4 bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc, 0, 0, 0);
    } else {
        acc = Lp<T>::mfma(a, b, acc);
    }
}

// one K step: the weight fragments w0 (j = 0), w1 (j = 1) against the activation pair.  W_FIRST: the weights are the MFMA's first operand
// (rows of the result = weight rows).  float / 16-bit: chunk 0 then chunk 1, as the kernels did before round 5.
template <typename T, bool W_FIRST = true>
__device__ __forceinline__ void mfma_pair(const u32x4& w0, const u32x4& w1, const XPair<T>& x, f32x16& acc) {
    if constexpr (std::is_same<T, F32S>::value) {
        const f16x8 WH = __builtin_bit_cast(f16x8, w0), WL = __builtin_bit_cast(f16x8, w1), XH = __builtin_bit_cast(f16x8, x.a), XL = __builtin_bit_cast(f16x8, x.b);
        if constexpr (W_FIRST) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH, XH, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(WL, XH, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH, XL, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(XH, WH, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(XH, WL, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(XL, WH, acc, 0, 0, 0);
        }
    } else if constexpr (W_FIRST) {
        mfma_chunk<T>(w0, x.a, acc);
        mfma_chunk<T>(w1, x.b, acc);
    } else {
        mfma_chunk<T>(x.a, w0, acc);
        mfma_chunk<T>(x.b, w1, acc);
    }
}

// four K = 2 steps of the exact-fp32 engine whose operands sit in registers as scalars (an accumulator row used as the next product's operand):
// acc += sum_e a_e x w[e] with the a's as the MFMA's A operand (A_FIRST) or its B operand.  (F32S sites build an XPair from the eight registers
// of a K step and call mfma_pair.)
template <typename T, bool A_FIRST = true>
__device__ __forceinline__ void mfma_quad(float a0, float a1, float a2, float a3, const f32x4& w, f32x16& acc) {
    static_assert(std::is_same<T, float>::value, "exact-fp32 form");
    const f32x4 a = {a0, a1, a2, a3};
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = A_FIRST ? __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], w[e], acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], a[e], acc, 0, 0, 0);
}

// Tile geometry for a given BN
template <int BN>
struct Geo;
template <>
struct Geo<128> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; };
template <>
struct Geo<64> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 1; };
template <>
struct Geo<32> { static constexpr int WM = 4, WN = 1, TM = 1, TN = 1; };

constexpr int BM = 128;

template <typename T, int TAPS, int BN, int RB>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs p) {
    using G = Geo<BN>;
    constexpr int PITCH = RB + 16;                 // LDS row pitch in bytes
    constexpr int CPR = RB / 16;                   // 16-byte chunks per row per step
    constexpr int ROWS_PER_PASS = 256 / CPR;       // rows covered by one pass of the 256 threads
    constexpr int A_PASSES = BM / ROWS_PER_PASS;
    constexpr int B_PASSES = (BN + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
    constexpr int KE = RB / Elem<T>::BYTES;        // K elements per step
    constexpr int A_BYTES = BM * PITCH, B_BYTES = BN * PITCH;
    static_assert(BN % 32 == 0 && A_PASSES >= 1, "tile");

    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // one pipeline stage: A tile then B tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- staging assignment: thread -> (row, 16-byte chunk) ------------------------------------------
    const int chunk = tid % CPR;
    const int srow = tid / CPR;
    const unsigned char* a_ptr[A_PASSES];  // pointer to in[pixel][0] (bytes) for each staged row
    int a_yx[A_PASSES];                    // (y << 16) | x, or -1 when the row is beyond M
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const long long m = m0 + srow + i * ROWS_PER_PASS;
        if (m < p.M) {
            const int hw = p.H * p.W;
            const int pix = (int)(m % hw);
            a_yx[i] = ((pix / p.W) << 16) | (pix % p.W);
            a_ptr[i] = reinterpret_cast<const unsigned char*>(p.in) + (size_t)m * p.in_pitch * Elem<T>::BYTES;
        } else {
            a_yx[i] = -1;
            a_ptr[i] = reinterpret_cast<const unsigned char*>(p.in);
        }
    }
    const int ksteps_per_tap = p.cin / KE;
    const int nsteps = TAPS * ksteps_per_tap;

    u32x4 ra[A_PASSES], rb[B_PASSES];
    bool ra_ok[A_PASSES];
    PreactCoef<T> coef;

    auto load_step = [&](int s) {
        const int tap = TAPS == 1 ? 0 : s / ksteps_per_tap;
        const int kc = TAPS == 1 ? s : s - tap * ksteps_per_tap;
        const int c0 = kc * KE + chunk * Elem<T>::PER16;  // first channel of this thread's chunk
        int dy = 0, dx = 0;
        if (TAPS == 9) {
            dy = tap / 3 - 1;
            dx = tap % 3 - 1;
        }
        const long long tap_off = ((long long)dy * p.W + dx) * p.in_pitch * Elem<T>::BYTES;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            bool ok = a_yx[i] >= 0;
            if (TAPS == 9 && ok) {
                const int y = (a_yx[i] >> 16) + dy, x = (a_yx[i] & 0xffff) + dx;
                ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            }
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(a_ptr[i] + tap_off + (size_t)c0 * Elem<T>::BYTES);
            ra[i] = v;
            ra_ok[i] = ok;
        }
        if (TAPS == 1 && p.scale) coef.load(p.scale, p.shift, c0);
        const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w) +
                                     ((size_t)tap * p.cout * p.cin + (size_t)c0) * Elem<T>::BYTES;
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int n = srow + i * ROWS_PER_PASS;
            if (BN % ROWS_PER_PASS == 0 || n < BN)
                rb[i] = *reinterpret_cast<const u32x4*>(wbase + (size_t)(n0 + n) * p.cin * Elem<T>::BYTES);
        }
    };
    auto store_step = [&](int buf) {
        unsigned char* const sa = smem + buf * STAGE_BYTES;
        unsigned char* const sb = sa + A_BYTES;
        // the input BN + ReLU is applied here, AFTER the MFMAs of the current step, so the global loads issued by
        // load_step() stay in flight across the whole compute phase
        if (TAPS == 1 && p.scale) {
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i)
                if (ra_ok[i]) ra[i] = preact_apply<T>(ra[i], coef);
        }
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i)
            *reinterpret_cast<u32x4*>(sa + (srow + i * ROWS_PER_PASS) * PITCH + chunk * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int n = srow + i * ROWS_PER_PASS;
            if (BN % ROWS_PER_PASS == 0 || n < BN) *reinterpret_cast<u32x4*>(sb + n * PITCH + chunk * 16) = rb[i];
        }
    };

    f32x16 acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets (bytes) inside a tile
    const int frag_row = lane & 31;
    const int frag_half = (lane >> 5) * 16;
    const int a_off = (wm * (G::TM * 32) + frag_row) * PITCH + frag_half;
    const int b_off = (wn * (G::TN * 32) + frag_row) * PITCH + frag_half;

    load_step(0);
    store_step(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const unsigned char* const sa = smem + buf * STAGE_BYTES;
        const unsigned char* const sb = sa + A_BYTES;
        if (s + 1 < nsteps) load_step(s + 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch loads ABOVE the MFMA block (hipcc sinks them otherwise)
#pragma unroll
        for (int j = 0; j < RB / 32; j += 2) {   // fragment pairs (j, j + 1) = one 64-byte step of K (float32: 16 values) per lane pair, see mfma_pair
            XPair<T> af[G::TM];
            u32x4 bf[G::TN][2];
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
                af[i] = make_xpair<T>(*reinterpret_cast<const u32x4*>(sa + a_off + i * 32 * PITCH + j * 32), *reinterpret_cast<const u32x4*>(sa + a_off + i * 32 * PITCH + (j + 1) * 32));
#pragma unroll
            for (int i = 0; i < G::TN; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) bf[i][jj] = *reinterpret_cast<const u32x4*>(sb + b_off + i * 32 * PITCH + (j + jj) * 32);
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
#pragma unroll
                for (int k = 0; k < G::TN; ++k) mfma_pair<T, false>(bf[k][0], bf[k][1], af[i], acc[i][k]);
        }
        if (s + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------------
    // C layout of the 32x32 MFMA: column (channel) = lane & 31, row (pixel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int hw = p.H * p.W;
#pragma unroll
    for (int j = 0; j < G::TN; ++j) {
        const int n = n0 + wn * (G::TN * 32) + j * 32 + (lane & 31);
        const float bias = p.bias[n];
#pragma unroll
        for (int i = 0; i < G::TM; ++i) {
            const long long mbase = m0 + wm * (G::TM * 32) + i * 32 + 4 * (lane >> 5);
            // all 16 residual loads of the tile are issued before the first store (loads may not pass stores
            // to possibly aliasing memory, so interleaving them would serialise 16 round trips)
            float resv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = mbase + (r & 3) + 8 * (r >> 2);
                resv[r] = 0.0f;
                if (p.res && m < p.M) {
                    if constexpr (sizeof(T) == 4)
                        resv[r] = reinterpret_cast<const float*>(p.res)[(size_t)m * p.res_pitch + n];
                    else
                        resv[r] = Lp<T>::to_f32(reinterpret_cast<const unsigned short*>(p.res)[(size_t)m * p.res_pitch + n]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = mbase + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] + bias;
                if (m < p.M) {
                    v += resv[r];
                    if (p.relu) v = fmaxf(v, 0.0f);
                    if (p.out) {
                        if constexpr (sizeof(T) == 4)
                            reinterpret_cast<float*>(p.out)[(size_t)m * p.out_pitch + n] = v;
                        else
                            reinterpret_cast<unsigned short*>(p.out)[(size_t)m * p.out_pitch + n] = Lp<T>::from_f32(v);
                    }
                }
                acc[i][j][r] = v;
            }
            if (p.out_nchw && n < p.cout_real) {
                // 4 consecutive registers = 4 consecutive pixels of one plane
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long m = mbase + 8 * q;
                    if (m + 3 < p.M) {
                        const long long view = m / hw;
                        const int pix = (int)(m - view * hw);
                        f32x4 o = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(p.out_nchw + ((size_t)view * p.cout_real + n) * hw + pix) = o;
                    }
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// stem: 7x7 / stride 2 / pad 3 convolution 3 -> 64 with folded BN + ReLU.  images f32 NHWC [V, H, W, 3].
// Workgroup = 8 x 16 output pixels x 64 channels; input patch 21 x 37 x 3 and the whole 147 x 64 weight
// matrix live in LDS.  K index k = ky*21 + kx*3 + c, so a patch row is contiguous in k.
// -----------------------------------------------------------------------------------------------------
// camera frames as the stem's input (df3d_hg_forward_u8): the patch values are sampled from the uint8 frames with the front-end's
// arithmetic (preprocess_math.h) instead of being read from a float image
struct StemU8 {
    const unsigned char* frames;   // [V][FH][FW][FC] uint8, or nullptr: read StemArgs::img
    const unsigned char* flip;     // [V] or nullptr
    int FH, FW, FC;
    df3d_pre::Norm nm;
};

struct StemArgs {
    const float* img;
    void* out;           // NHWC [V, H/2, W/2, 64] (T)
    const float* w;      // [148][64] f32, k-major (row 147 = 0)
    const void* w_bf16;  // bf16 engine: [64][184] bf16, k' = ky*24 + kx*3 + c (stem_relayout_kernel)
    const float* bias;   // [64]
    int V, H, W;         // input size
    StemU8 u8;
};

// one 1 KB LDS-DMA piece (the helper of hg_bt_ring.h, needed here before that header): lane l's 16 bytes from sbase + voff -> dst + 16 l
__device__ __forceinline__ void stem_glds_piece(const void* sbase, unsigned voff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(dst)
                 : "memory");
}

// PERSISTENT (round 3): a workgroup walks tiles b, b + gridDim.x, ...: the 37 KB weight matrix enters LDS ONCE per workgroup (it was
// 1.2 x the tile's output bytes, per tile), and the next tile's patch is gathered into registers before the K loop and written to the
// other patch buffer behind it, so that no tile after the first waits for its input.  Same MFMA order: bit-identical.
template <typename T>
__global__ __launch_bounds__(256, 3) void stem_kernel(StemArgs p) {
    constexpr int PR = 21, PC = 37, PROW = 112;  // patch rows, cols, floats per patch row (111 padded)
    constexpr int KTOT = 148;
    constexpr int NIT = (PR * PC + 255) / 256;   // patch pixels per thread (4; the last round is partial)
    __shared__ float patch2[1][PR * PROW];   // ONE buffer: 47 KB per workgroup keep three workgroups on a CU (two buffers, two workgroups: 8 % slower)
    __shared__ float wl[KTOT * 64];
    const int OH = p.H / 2, OW = p.W / 2;
    const int tiles_x = OW / 16, tiles_y = OH / 8;
    const int ntiles = p.V * tiles_y * tiles_x;
    const int tid = threadIdx.x;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;

    {   // the [148][64] fp32 weight matrix as it lies: 37 one-KB pieces by LDS-DMA (no registers, no ds_write; runs under the patch staging)
        const unsigned wl_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)wl;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        static_assert(KTOT * 64 * 4 == 37 * 1024, "37 whole pieces");
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int pc = wv + 4 * k;
            if (pc < 37) stem_glds_piece(p.w, (unsigned)pc * 1024u + (unsigned)(tid & 63) * 16u, wl_addr + (unsigned)pc * 1024u);
        }
    }
    // one item = one patch pixel (three contiguous floats): index arithmetic and bounds test per pixel, not per value
    float pv[NIT][3];
    auto gather = [&](int t) {
        int b = t;
        const int tx0 = (b % tiles_x) * 16;
        b /= tiles_x;
        const int ty0 = (b % tiles_y) * 8;
        const int view = b / tiles_y;
        const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
        const float* img = p.img + (size_t)view * p.H * p.W * 3;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int i = tid + 256 * j;
            const int r = i / PC, pxl = i - r * PC;
            const int y = iy0 + r, x = ix0 + pxl;
            float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
            if (i < PR * PC && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
                if (p.u8.frames) {
                    float res[3];
                    df3d_pre::pixel(p.u8.frames + (size_t)view * p.u8.FH * p.u8.FW * p.u8.FC, p.u8.FH, p.u8.FW, p.u8.FC, p.u8.flip && p.u8.flip[view], p.H, p.W, y,
                                    x, p.u8.nm, res);
                    v0 = res[0];
                    v1 = res[1];
                    v2 = res[2];
                } else {
                    const float* const src = img + ((size_t)y * p.W + x) * 3;
                    v0 = src[0];
                    v1 = src[1];
                    v2 = src[2];
                }
            }
            pv[j][0] = v0;
            pv[j][1] = v1;
            pv[j][2] = v2;
        }
    };
    auto scatter = [&](float* patch) {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int i = tid + 256 * j;
            if (i < PR * PC) {
                const int r = i / PC, pxl = i - r * PC;
                float* const dst = patch + r * PROW + 3 * pxl;
                dst[0] = pv[j][0];
                dst[1] = pv[j][1];
                dst[2] = pv[j][2];
            }
        }
        if (tid < PR) patch[tid * PROW + PC * 3] = 0.0f;   // the pad cell behind the 111 values of a row
    };
    gather(tile);
    scatter(patch2[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's weight pieces have landed
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31;                 // pixel inside the wave's 2 x 16 sub-tile
    const int py = wave * 2 + (m >> 4), px = m & 15;
    const int a_base = (2 * py) * PROW + 6 * px;
    const int khalf = lane >> 5;
    const int n = lane & 31;
    const float bias0 = p.bias[n], bias1 = p.bias[32 + n];
    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) gather(next);   // requested now, consumed behind the K loop
        const float* const patch = patch2[0];
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
        // K step s multiplies k = 2 s (lanes 0..31) and k = 2 s + 1 (lanes 32..63); tap k lies at patch offset (k / 21) * PROW + k % 21.
        // Fully unrolled with every LDS address a register + an immediate: the offset of the ODD tap is the even one's + 1, or + PROW - 20
        // where the pair straddles two patch rows -- two base registers.  Reads run one group of four steps ahead of the MFMAs (the
        // rolled loop spent ten VALU instructions on k / 21 and waited for each step's three reads in front of its two MFMAs: 0.59
        // matrix-pipe busy).  Same products, same order: bit-identical.
        const float* const pa = patch + a_base + khalf;                  // odd tap = even tap + 1
        const float* const pb = patch + a_base + khalf * (PROW - 20);    // ... or first tap of the next patch row
        const float* const wb = wl + khalf * 64 + (lane & 31);
        constexpr int G = 4, NG = (KTOT / 2 + G - 1) / G;                // 74 steps in 19 groups (the last has two)
        float fa[2][G], fb0[2][G], fb1[2][G];
        auto load_group = [&](int g, int buf) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = g * G + i;
                if (s >= KTOT / 2) break;
                const int k0 = 2 * s, off0 = (k0 / 21) * PROW + k0 % 21;
                const bool straddle = k0 % 21 == 20;
                float a = straddle ? pb[off0] : pa[off0];
                if (k0 + 1 >= 147) a = khalf ? 0.0f : a;   // the 148th tap is padding
                fa[buf][i] = a;
                fb0[buf][i] = wb[k0 * 64];
                fb1[buf][i] = wb[k0 * 64 + 32];
            }
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                if (g * G + i >= KTOT / 2) break;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i], fb0[g & 1][i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i], fb1[g & 1][i], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        int b = tile;
        const int tx0 = (b % tiles_x) * 16;
        b /= tiles_x;
        const int ty0 = (b % tiles_y) * 8;
        const int view = b / tiles_y;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int oy = ty0 + wave * 2 + (mm >> 4), ox = tx0 + (mm & 15);
            const size_t o = (((size_t)view * OH + oy) * OW + ox) * 64;
            const float v0 = fmaxf(acc0[r] + bias0, 0.0f), v1 = fmaxf(acc1[r] + bias1, 0.0f);
            if constexpr (sizeof(T) == 4) {
                reinterpret_cast<float*>(p.out)[o + n] = v0;
                reinterpret_cast<float*>(p.out)[o + 32 + n] = v1;
            } else {
                reinterpret_cast<unsigned short*>(p.out)[o + n] = Lp<T>::from_f32(v0);
                reinterpret_cast<unsigned short*>(p.out)[o + 32 + n] = Lp<T>::from_f32(v1);
            }
        }
        if (!has_next) break;
        __syncthreads();   // every wave is done reading the patch
        scatter(patch2[0]);
        __syncthreads();
        tile = next;
    }
}

// -----------------------------------------------------------------------------------------------------
// stem for the 16-bit engines (T = __hip_bfloat16 / _Float16): the same 7x7/2 convolution on v_mfma_f32_32x32x16_bf16 / _f16.  K is laid out ky-major
// with every ky row padded from 21 to 24 taps (k' = ky*24 + kx*3 + c; 7*24 = 168, padded to 176 = 11 MFMA steps),
// so the 8 K-slots a lane feeds to one MFMA are 8 CONSECUTIVE bf16 values of one patch row (4 ds_read_b32).
// The f32 image patch is converted to bf16 while it is staged; weights are re-laid [64][176] bf16 in LDS from the
// same f32 blob the f32 stem uses.  22 MFMAs per wave instead of 148: the kernel becomes load/store-bound.
// -----------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_lp_kernel(StemArgs p) {
    constexpr int PR = 21, PC = 37, PROW = 120;   // patch row pitch in bf16 elements (111 used; 240 B, a multiple of 16)
    constexpr int KP = 176, WPITCH = KP + 8;      // weight row: 176 k' + 8 pad (368 B: conflict-free 16-byte reads)
    __shared__ __attribute__((aligned(16))) unsigned short patch[PR * PROW + 64];
    __shared__ __attribute__((aligned(16))) unsigned short wl[64 * WPITCH];
    const int OH = p.H / 2, OW = p.W / 2;
    const int tiles_x = OW / 16, tiles_y = OH / 8;
    const int ntiles = p.V * tiles_y * tiles_x;
    const int tid = threadIdx.x;
    // PERSISTENT (round 5, as stem_kernel<float> since round 3): a workgroup walks tiles b, b + gridDim.x, ...: the 23 KB weight tile enters
    // LDS ONCE per workgroup (per tile it was 1.4 x the tile's output bytes), and the next tile's patch is gathered into registers before the
    // K loop and written behind it.  Same MFMA order: bit-identical.
    int tile = blockIdx.x;
    if (tile >= ntiles) return;

    // weights: the [64][184] 16-bit tile was laid out once by stem_relayout_kernel (wl[n][ky*24 + kk] = w[ky*21 + kk][n]); it is
    // copied as it lies, 23 one-KB pieces, by LDS-DMA (no registers, no ds_write: the copy runs under the patch staging below)
    {
        const unsigned wl_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)wl;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        static_assert(64 * WPITCH * 2 == 23 * 1024, "23 whole pieces");
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int pc = wv + 4 * k;
            if (pc < 23) stem_glds_piece(p.w_bf16, (unsigned)pc * 1024u + (unsigned)(tid & 63) * 16u, wl_addr + (unsigned)pc * 1024u);
        }
    }
    // one item = one patch pixel (three contiguous floats): the index arithmetic and the bounds test are per pixel, not per value.
    // ALL of a thread's items are requested before the first is stored (as a rolled loop the staging was four load -> store round
    // trips in series)
    constexpr int NIT = (PR * PC + 255) / 256;
    float pv[NIT][3];
    auto gather = [&](int t) {
        int b = t;
        const int tx0 = (b % tiles_x) * 16;
        b /= tiles_x;
        const int ty0 = (b % tiles_y) * 8;
        const int view = b / tiles_y;
        const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
        const float* img = p.img + (size_t)view * p.H * p.W * 3;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int i = tid + 256 * j;
            const int r = i / PC, pxl = i - r * PC;
            const int y = iy0 + r, x = ix0 + pxl;
            float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
            if (i < PR * PC && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
                if (p.u8.frames) {
                    float res[3];
                    df3d_pre::pixel(p.u8.frames + (size_t)view * p.u8.FH * p.u8.FW * p.u8.FC, p.u8.FH, p.u8.FW, p.u8.FC, p.u8.flip && p.u8.flip[view], p.H, p.W, y, x,
                                    p.u8.nm, res);
                    v0 = res[0];
                    v1 = res[1];
                    v2 = res[2];
                } else {
                    const float* const src = img + ((size_t)y * p.W + x) * 3;
                    v0 = src[0];
                    v1 = src[1];
                    v2 = src[2];
                }
            }
            pv[j][0] = v0;
            pv[j][1] = v1;
            pv[j][2] = v2;
        }
    };
    auto scatter = [&]() {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int i = tid + 256 * j;
            if (i < PR * PC) {
                const int r = i / PC, pxl = i - r * PC;
                unsigned short* const dst = patch + r * PROW + 3 * pxl;
                dst[0] = Lp<T>::from_f32(pv[j][0]);
                dst[1] = Lp<T>::from_f32(pv[j][1]);
                dst[2] = Lp<T>::from_f32(pv[j][2]);
            }
        }
    };
    gather(tile);
    scatter();
    // the pad cells behind the 111 values of a row and behind the last row (read by the last K slots against zero weights) are zero; no tile writes them
    for (int i = tid; i < PR * (PROW - PC * 3) + 64; i += 256) {
        const int r = i / (PROW - PC * 3), c = i - r * (PROW - PC * 3);
        patch[i < PR * (PROW - PC * 3) ? r * PROW + PC * 3 + c : PR * PROW + (i - PR * (PROW - PC * 3))] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's weight pieces have landed
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, half = lane >> 5;
    const int py = wave * 2 + (m >> 4), px = m & 15;
    const unsigned short* const abase = patch + (2 * py) * PROW + 6 * px;   // 12*px bytes: 4-byte aligned
    const unsigned short* const wrow0 = wl + m * WPITCH;
    const unsigned short* const wrow1 = wl + (32 + m) * WPITCH;
    const int n = lane & 31;
    const int odd = lane & 1;
    const float bias0 = p.bias[n], bias1 = p.bias[32 + n];
    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) gather(next);   // requested now, consumed behind the K loop
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
#pragma unroll
        for (int g = 0; g < KP / 16; ++g) {
            const int kp = 16 * g + 8 * half;          // first k' of this lane's 8 slots
            const int ky = kp / 24, kk = kp % 24;       // 8 consecutive taps of patch row 2*py + ky (kk in {0, 8, 16})
            u32x4 av = {0u, 0u, 0u, 0u};
            if (kp < 168) {
                const unsigned* ap = reinterpret_cast<const unsigned*>(abase + ky * PROW + kk);
                av[0] = ap[0];
                av[1] = ap[1];
                av[2] = ap[2];
                av[3] = ap[3];   // taps 21..23 of the row multiply zero weights
            }
            const u32x4 b0 = *reinterpret_cast<const u32x4*>(wrow0 + kp);
            const u32x4 b1 = *reinterpret_cast<const u32x4*>(wrow1 + kp);
            acc0 = Lp<T>::mfma(av, b0, acc0);
            acc1 = Lp<T>::mfma(av, b1, acc1);
        }
        // epilogue: adjacent lanes hold adjacent channels of the same pixel; exchanging one register between lane pairs
        // lets every lane store TWO channels (4 bytes) of one pixel: even lanes take pixel-register r, odd lanes r + 1
        int b = tile;
        const int tx0 = (b % tiles_x) * 16;
        b /= tiles_x;
        const int ty0 = (b % tiles_y) * 8;
        const int view = b / tiles_y;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float v0a = fmaxf(acc0[r] + bias0, 0.0f), v0b = fmaxf(acc0[r + 1] + bias0, 0.0f);
            const float v1a = fmaxf(acc1[r] + bias1, 0.0f), v1b = fmaxf(acc1[r + 1] + bias1, 0.0f);
            const float g0 = __shfl_xor(odd ? v0a : v0b, 1, 64);   // even gets partner's value for register r, odd for r + 1
            const float g1 = __shfl_xor(odd ? v1a : v1b, 1, 64);
            const int rr = r + odd;
            const int mm = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
            const int oy = ty0 + wave * 2 + (mm >> 4), ox = tx0 + (mm & 15);
            const size_t o = (((size_t)view * OH + oy) * OW + ox) * 64 + (n & ~1);
            const unsigned w0 = odd ? Lp<T>::pack2(g0, v0b) : Lp<T>::pack2(v0a, g0);
            const unsigned w1 = odd ? Lp<T>::pack2(g1, v1b) : Lp<T>::pack2(v1a, g1);
            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(p.out) + o) = w0;
            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(p.out) + o + 32) = w1;
        }
        if (!has_next) break;
        __syncthreads();   // every wave is done reading the patch
        scatter();
        __syncthreads();
        tile = next;
    }
}

// one-time re-layout of the stem weights for stem_lp_kernel: f32 [148][64] (k = ky*21 + kk) -> 16-bit [64][184]
template <typename T>
__global__ __launch_bounds__(256) void stem_relayout_kernel(const float* __restrict__ w, unsigned short* __restrict__ out) {
    constexpr int WPITCH = 184;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * WPITCH) return;
    const int n = i / WPITCH, kp = i % WPITCH;
    const int ky = kp / 24, kk = kp % 24;
    float v = 0.0f;
    if (kp < 168 && kk < 21) v = w[(ky * 21 + kk) * 64 + n];
    out[i] = Lp<T>::from_f32(v);
}

// -----------------------------------------------------------------------------------------------------
// stem of the f32s engine: stem_lp_kernel's layout (K ky-major, 24 slots per patch row, 11 MFMA steps) with BOTH operands as IEEE-half
// hi / lo pairs -- the image patch is split while it is staged (two 16-bit patches), the weights were split and re-laid once by
// stem_relayout_f32s_kernel (two [64][184] tiles, hi at byte 0 and lo at byte 23 552 of the stem's slot in the pre-split blob copy) --
// three MFMAs per step and output tile (x_hi w_hi + x_lo w_hi + x_hi w_lo), float32 accumulation, float32 output: 66 MFMAs per wave
// and tile where the exact-fp32 stem_kernel issues 148 at four times the cycles each.
// -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_f32s_kernel(StemArgs p) {
    constexpr int PR = 21, PC = 37, PROW = 120;
    constexpr int KP = 176, WPITCH = KP + 8;
    constexpr int PATCH = PR * PROW + 64;
    __shared__ __attribute__((aligned(16))) unsigned short patch[2][PATCH];       // [hi | lo]
    __shared__ __attribute__((aligned(16))) unsigned short wl[2][64 * WPITCH];    // [hi | lo]
    const int OH = p.H / 2, OW = p.W / 2;
    const int tiles_x = OW / 16, tiles_y = OH / 8;
    const int ntiles = p.V * tiles_y * tiles_x;
    const int tid = threadIdx.x;
    int tile = blockIdx.x;   // PERSISTENT, as the other stems: the 46 KB of weight tiles enter LDS once per workgroup
    if (tile >= ntiles) return;
    {   // both weight tiles as they lie: 46 one-KB pieces by LDS-DMA, under the patch staging
        const unsigned wl_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)&wl[0][0];
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        static_assert(64 * WPITCH * 2 == 23 * 1024, "23 whole pieces per tile");
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int pc = wv + 4 * k;
            if (pc < 46) stem_glds_piece(p.w_bf16, (unsigned)pc * 1024u + (unsigned)(tid & 63) * 16u, wl_addr + (unsigned)pc * 1024u);
        }
    }
    constexpr int NIT = (PR * PC + 255) / 256;
    float pv[NIT][3];
    auto gather = [&](int t) {
        int b = t;
        const int tx0 = (b % tiles_x) * 16;
        b /= tiles_x;
        const int ty0 = (b % tiles_y) * 8;
        const int view = b / tiles_y;
        const int iy0 = 2 * ty0 - 3, ix0 = 2 * tx0 - 3;
        const float* img = p.img + (size_t)view * p.H * p.W * 3;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int i = tid + 256 * j;
            const int r = i / PC, pxl = i - r * PC;
            const int y = iy0 + r, x = ix0 + pxl;
            float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
            if (i < PR * PC && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
                if (p.u8.frames) {
                    float res[3];
                    df3d_pre::pixel(p.u8.frames + (size_t)view * p.u8.FH * p.u8.FW * p.u8.FC, p.u8.FH, p.u8.FW, p.u8.FC, p.u8.flip && p.u8.flip[view], p.H, p.W, y, x,
                                    p.u8.nm, res);
                    v0 = res[0];
                    v1 = res[1];
                    v2 = res[2];
                } else {
                    const float* const src = img + ((size_t)y * p.W + x) * 3;
                    v0 = src[0];
                    v1 = src[1];
                    v2 = src[2];
                }
            }
            pv[j][0] = v0;
            pv[j][1] = v1;
            pv[j][2] = v2;
        }
    };
    auto scatter = [&]() {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int i = tid + 256 * j;
            if (i < PR * PC) {
                const int r = i / PC, pxl = i - r * PC;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const _Float16 h = (_Float16)pv[j][c];
                    patch[0][r * PROW + 3 * pxl + c] = __builtin_bit_cast(unsigned short, h);
                    patch[1][r * PROW + 3 * pxl + c] = __builtin_bit_cast(unsigned short, (_Float16)(pv[j][c] - (float)h));
                }
            }
        }
    };
    gather(tile);
    scatter();
    for (int i = tid; i < PR * (PROW - PC * 3) + 64; i += 256) {   // the pad cells (read against zero weights) are zero; no tile writes them
        const int r = i / (PROW - PC * 3), c = i - r * (PROW - PC * 3);
        const int at = i < PR * (PROW - PC * 3) ? r * PROW + PC * 3 + c : PR * PROW + (i - PR * (PROW - PC * 3));
        patch[0][at] = 0;
        patch[1][at] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's weight pieces have landed
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, half = lane >> 5;
    const int py = wave * 2 + (m >> 4), px = m & 15;
    const int aoff = (2 * py) * PROW + 6 * px;
    const int n = lane & 31;
    const float bias0 = p.bias[n], bias1 = p.bias[32 + n];
    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) gather(next);   // requested now, consumed behind the K loop
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
#pragma unroll
        for (int g = 0; g < KP / 16; ++g) {
            const int kp = 16 * g + 8 * half;
            const int ky = kp / 24, kk = kp % 24;
            u32x4 ah = {0u, 0u, 0u, 0u}, al = {0u, 0u, 0u, 0u};
            if (kp < 168) {
                const unsigned* const aph = reinterpret_cast<const unsigned*>(&patch[0][aoff + ky * PROW + kk]);
                const unsigned* const apl = reinterpret_cast<const unsigned*>(&patch[1][aoff + ky * PROW + kk]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ah[e] = aph[e];
                    al[e] = apl[e];
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f16x8 wh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(&wl[0][(32 * t + m) * WPITCH + kp]));
                const f16x8 wlo = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(&wl[1][(32 * t + m) * WPITCH + kp]));
                f32x16& acc = t ? acc1 : acc0;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), wh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), wh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), wlo, acc, 0, 0, 0);
            }
        }
        // epilogue as stem_kernel<float>: lane = channel, register = pixel; 4-byte stores of 128 contiguous bytes per pixel and tile
        int b = tile;
        const int tx0 = (b % tiles_x) * 16;
        b /= tiles_x;
        const int ty0 = (b % tiles_y) * 8;
        const int view = b / tiles_y;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int oy = ty0 + wave * 2 + (mm >> 4), ox = tx0 + (mm & 15);
            const size_t o = (((size_t)view * OH + oy) * OW + ox) * 64;
            reinterpret_cast<float*>(p.out)[o + n] = fmaxf(acc0[r] + bias0, 0.0f);
            reinterpret_cast<float*>(p.out)[o + 32 + n] = fmaxf(acc1[r] + bias1, 0.0f);
        }
        if (!has_next) break;
        __syncthreads();   // every wave is done reading the patches
        scatter();
        __syncthreads();
        tile = next;
    }
}

// one-time re-layout of the stem weights for stem_f32s_kernel: f32 [148][64] -> two IEEE-half [64][184] tiles, hi = rn(w) and lo = rn(w - hi)
__global__ __launch_bounds__(256) void stem_relayout_f32s_kernel(const float* __restrict__ w, unsigned short* __restrict__ out) {
    constexpr int WPITCH = 184;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * WPITCH) return;
    const int n = i / WPITCH, kp = i % WPITCH;
    const int ky = kp / 24, kk = kp % 24;
    float v = 0.0f;
    if (kp < 168 && kk < 21) v = w[(ky * 21 + kk) * 64 + n];
    const _Float16 h = (_Float16)v;
    out[i] = __builtin_bit_cast(unsigned short, h);
    out[64 * WPITCH + i] = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
}

// -----------------------------------------------------------------------------------------------------
// 2x2 max-pool and nearest-upsample + add: one thread per 16-byte channel chunk
// -----------------------------------------------------------------------------------------------------
// bf16 as an ORDERED 16-bit integer and back (the map is its own inverse): a negative float's magnitude bits are flipped, so
// that signed integer comparison orders the keys like the floats (-0 just below +0).  Lets max() of packed bf16 run as one
// v_pk_max_i16 per two values instead of unpack + three v_max_f32 (the NaN-quieting hipcc adds) + repack.
typedef short hg_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16x2_key(unsigned v) {
    const hg_i16x2 x = __builtin_bit_cast(hg_i16x2, v);
    const hg_i16x2 m = (x >> 15) & (short)0x7fff;
    return __builtin_bit_cast(unsigned, x ^ m);
}
__device__ __forceinline__ unsigned bf16x2_key_max(unsigned ka, unsigned kb) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(hg_i16x2, ka), __builtin_bit_cast(hg_i16x2, kb)));
}
__device__ __forceinline__ u32x4 bf16_key_chunk(u32x4 v) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = bf16x2_key(v[i]);
    return o;
}
__device__ __forceinline__ u32x4 bf16_key_max_chunk(u32x4 a, u32x4 b) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = bf16x2_key_max(a[i], b[i]);
    return o;
}

// element-wise maximum of one 16-byte chunk (finite values; of +0 and -0 the bf16 form returns +0)
template <typename T>
__device__ __forceinline__ u32x4 max_chunk(u32x4 a, u32x4 b) {
    if constexpr (sizeof(T) == 4) {
        f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = fmaxf(x[i], y[i]);
        return __builtin_bit_cast(u32x4, x);
    } else if constexpr (std::is_same<T, _Float16>::value) {
        // IEEE half has a packed maximum of its own: one instruction per two values (the ordered-integer detour of the bf16 form
        // costs seven).  Inline assembly on purpose: the builtin brings a NaN-quieting v_pk_max_f16 per operand; pooled values only
        // ever go to stores and lane shuffles, never straight into an MFMA (see br_relu_pk on that hazard)
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) asm("v_pk_max_f16 %0, %1, %2" : "=v"(o[i]) : "v"(a[i]), "v"(b[i]));
        return o;
    } else {
        return bf16_key_chunk(bf16_key_max_chunk(bf16_key_chunk(a), bf16_key_chunk(b)));
    }
}
template <typename T>
__device__ __forceinline__ u32x4 add_chunk(u32x4 a, u32x4 b) {
    if constexpr (sizeof(T) == 4) {
        f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
        x += y;
        return __builtin_bit_cast(u32x4, x);
    } else {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float al = Lp<T>::to_f32((unsigned short)(a[i] & 0xffffu)), ah = Lp<T>::to_f32((unsigned short)(a[i] >> 16));
            const float bl = Lp<T>::to_f32((unsigned short)(b[i] & 0xffffu)), bh = Lp<T>::to_f32((unsigned short)(b[i] >> 16));
            o[i] = Lp<T>::pack2(al + bl, ah + bh);
        }
        return o;
    }
}

// in [V, H, W, C] -> out [V, H/2, W/2, C];  chunks = C * sizeof(T) / 16 per pixel
template <typename T>
__global__ __launch_bounds__(256) void pool2_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long total,
                                                    int OH, int OW, int chunks) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % chunks);
    long long pidx = idx / chunks;
    const int ox = (int)(pidx % OW);
    pidx /= OW;
    const int oy = (int)(pidx % OH);
    const long long v = pidx / OH;
    const int W = OW * 2;
    const size_t base = (((size_t)v * OH * 2 + 2 * oy) * W + 2 * ox) * chunks + ch;
    const u32x4 a = in[base], b = in[base + chunks], c = in[base + (size_t)W * chunks], d = in[base + (size_t)W * chunks + chunks];
    out[idx] = max_chunk<T>(max_chunk<T>(a, b), max_chunk<T>(c, d));
}

// out[v, y, x, :] = a[v, y, x, :] + b[v, y/2, x/2, :]   (out may alias a)
template <typename T>
__global__ __launch_bounds__(256) void upadd_kernel(const u32x4* a, const u32x4* __restrict__ b, u32x4* out, long long total,
                                                    int H, int W, int chunks) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % chunks);
    long long pidx = idx / chunks;
    const int x = (int)(pidx % W);
    pidx /= W;
    const int y = (int)(pidx % H);
    const long long v = pidx / H;
    const size_t bidx = (((size_t)v * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * chunks + ch;
    out[idx] = add_chunk<T>(a[idx], b[bidx]);
}

// debug / parity export: NHWC with channel pitch -> dense float32 NHWC with c channels
template <typename T>
__global__ __launch_bounds__(256) void export_kernel(const void* __restrict__ in, float* __restrict__ out, long long pixels,
                                                     int c, int pitch) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= pixels * c) return;
    const long long pix = idx / c;
    const int ch = (int)(idx % c);
    if constexpr (sizeof(T) == 4)
        out[idx] = reinterpret_cast<const float*>(in)[(size_t)pix * pitch + ch];
    else
        out[idx] = Lp<T>::to_f32(reinterpret_cast<const unsigned short*>(in)[(size_t)pix * pitch + ch]);
}

// weights f32 -> 16-bit storage format (round to nearest even)
template <typename T>
__global__ __launch_bounds__(256) void f32_to_lp_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = Lp<T>::from_f32(in[i]);
}


// =====================================================================================================
// Fused pre-activation bottleneck  (256 -> 128 -> 128 -> 256, the block that makes up 29 of the network's
// 32 bottlenecks):    out = W3 relu(bn3(W2 (*) relu(bn2(W1 relu(bn1 x))))) + x
// One workgroup = one 8 x 16 output tile.  x is read once (10 x 18 halo tile), the output written once; the two
// 128-channel intermediates never leave the CU:
//   phase 1  t1 = relu(W1' relu(bn1 x) + b1')  on the 180 halo pixels (GEMM 192 x 128 x 256) -> LDS [180][128],
//            out-of-image halo pixels forced to 0 (= the zero padding of the 3x3 convolution).
//            Wave w owns output channels 32w..32w+31 for all 6 pixel tiles (x fragments are shared through LDS).
//   phase 2  t2^T = W2' (*) t1: rows = output channels (A = W2 taps, staged), columns = the wave's 32 pixels
//            (B = t1 read straight from the LDS tile at the tap-shifted pixel) -> 4 accumulator tiles per wave.
//   phase 3  out = W3 relu(t2 + b2') + b3 + x.  The transposed phase-2 accumulators ARE valid MFMA A operands:
//            accumulator register r of channel tile m holds, for pixel lane&31, channel 32m + (r&3) + 8(r>>2) +
//            4(lane>>5) -- a permutation of K that is matched on the W3 side by which 16-byte chunk a lane reads
//            (f32) / by the host's K order of the packed W3 (bf16).  So t2 never goes through LDS.
// LDS (fp32): t1 180 x 528 B = 95 KB + staging 51 KB (phase 1) / 37 KB (phases 2, 3) -> one workgroup per CU.
// =====================================================================================================
struct BottleneckArgs {
    const void* in;     // NHWC [V, H, W, CIN]
    const void* in2;    // UP: NHWC [V, H/2, W/2, CIN]; the block's input is in + nearest-upsample(in2), rounded to T
    const void* add2;   // ADD2: NHWC [V, H/2, W/2, 2*PL]; the block WRITES out + nearest-upsample(add2) (the hourglass' up-path sum,
                        // what upadd_kernel would have made of `out`: same roundings, in the same order)
    void* out;          // NHWC [V, H, W, 2*PL]
    void* pool;         // optional NHWC [V, H/2, W/2, 2*PL]: 2x2 max-pool of `out`, written by the same epilogue
    const void* w1;     // [PL][CIN]
    const void* w2;     // [9][PL][PL]
    const void* w3;     // [2*PL][PL]      (K permuted for bf16)
    const void* wd;     // [2*PL][CIN]     downsample (skip) convolution, DS only
    const float* b1;    // [PL]   (bn2 folded)
    const float* b2;    // [PL]   (bn3 folded)
    const float* b3;    // [2*PL] (DS: conv3 bias + downsample bias, summed by the engine at set_weights time)
    const float* bd;    // [2*PL] DS only
    const float* s1;    // [CIN] bn1 scale
    const float* t1;    // [CIN] bn1 shift
    int V, H, W;
};

constexpr int BT_TH = 8, BT_TW = 16;              // output tile
constexpr int BT_HW = BT_TW + 2;                  // halo tile width (18)
constexpr int BT_HALO = (BT_TH + 2) * BT_HW;      // 180 halo pixels
constexpr int BT_HROWS = 192;                     // padded to 6 MFMA row tiles

// CIN -> PL -> PL -> 2*PL; DS: the skip path is a 1x1 convolution of the raw input (CIN != 2*PL)
template <typename T, int CIN, int PL, bool DS>
struct BtCfg {
    static constexpr int EB = Elem<T>::BYTES;
    static constexpr int CO = 2 * PL;
    // fp32, PL = 128: the t1 tile is built and consumed in TWO halves of 64 channels (phase 1 twice over x, phase 2
    // accumulates tap x channel-half), which halves its LDS footprint; with a single staging buffer the workgroup
    // then needs 73 KB instead of 153 KB and two workgroups share a CU, as in bf16.
    static constexpr int KSPLIT = (EB == 4 && PL == 128) ? 2 : 1;
    static constexpr int T1W = PL / KSPLIT;                        // channels of t1 resident at a time
    static constexpr int T1_PITCH = T1W * EB + 16;                 // bytes per halo pixel in the t1 tile
    static constexpr int T1_BYTES = BT_HROWS * T1_PITCH;           // 192 rows: the 12 pad rows make the phase-1 epilogue branch-free
    static constexpr int RB1 = 64;                                 // staged row bytes, phase 1
    static constexpr int RB2 = 128;                                // phases 2 and 3 (K = PL)
    // ONE staging buffer (two barriers per K-step) so that the workgroup needs < 80 KB of LDS and two workgroups
    // share a CU -- the second one's MFMAs cover the first one's barrier / LDS / memory stalls.
    static constexpr int RBD = 64;                                 // downsample steps of phase 3 (K = CIN)
    static constexpr int STAGE1 = (BT_HROWS + T1W) * (RB1 + 16);   // x rows + W1 rows
    static constexpr int STAGE2 = 128 * (RB2 + 16);                // W2 (PL rows) / W3 (128 rows) 
    static constexpr int STAGED = DS ? (128 + 128) * (RBD + 16) : 0;  // x centre rows + Wd rows
    static constexpr int SMAX = STAGE1 > STAGE2 ? (STAGE1 > STAGED ? STAGE1 : STAGED) : (STAGE2 > STAGED ? STAGE2 : STAGED);
    static constexpr int STAGE_BYTES = SMAX;
    static constexpr int MISC = 64;                                // halo validity masks (3 x 64 bit)
    static constexpr int LDS_BYTES = T1_BYTES + STAGE_BYTES + MISC;
    static constexpr int NT = PL / 32;                             // channel tiles of the intermediates
};

template <typename T, int CIN, int PL, bool DS, bool UP = false, bool ADD2 = false>
__global__ __launch_bounds__(256, 2) void bottleneck_kernel(BottleneckArgs p) {
    using C = BtCfg<T, CIN, PL, DS>;
    static_assert(!UP || !DS, "the upsample-add input exists for the identity-skip block only");
    static_assert(!ADD2 || (!DS && !UP), "the fused up-path sum is written by plain identity-skip blocks");
    constexpr int EB = C::EB;
    constexpr int CO = C::CO;
    constexpr int NT = C::NT;
    constexpr int PER16 = Elem<T>::PER16;
    static_assert(PL == 128 || PL == 64, "planes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const t1_lds = smem;
    unsigned char* const stage = smem + C::T1_BYTES;
    unsigned long long* const valid_lds = reinterpret_cast<unsigned long long*>(smem + C::T1_BYTES + C::STAGE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5;          // which 16-byte chunk of a 32-byte K group this lane reads
    const int l31 = lane & 31;
    const int tiles_x = p.W / BT_TW, tiles_y = p.H / BT_TH;
    int b = blockIdx.x;
    const int tx0 = (b % tiles_x) * BT_TW;
    b /= tiles_x;
    const int ty0 = (b % tiles_y) * BT_TH;
    const int view = b / tiles_y;
    const unsigned char* const xin = reinterpret_cast<const unsigned char*>(p.in) + (size_t)view * p.H * p.W * CIN * EB;
    // UP: the low-resolution addend (the hourglass' up-path: x = in + upsample(in2), what upadd_kernel would have written)
    const unsigned char* const xin2 = UP ? reinterpret_cast<const unsigned char*>(p.in2) + (size_t)view * (p.H / 2) * (p.W / 2) * CIN * EB : nullptr;
    const unsigned char* const lo2 = ADD2 ? reinterpret_cast<const unsigned char*>(p.add2) + (size_t)view * (p.H / 2) * (p.W / 2) * CO * EB : nullptr;

    // validity of the 192 halo rows (inside the image?) as three 64-bit masks
    if (tid < BT_HROWS) {
        const int hy = tid / BT_HW, hx = tid % BT_HW;
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
        const bool ok = tid < BT_HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const unsigned long long m = __ballot(ok);
        if (lane == 0) valid_lds[wave] = m;
    }

    // =========================== phase 2: t2^T = W2' (*) t1 ==============================================
    // (declarations first: phase 1 and phase 2 alternate when the t1 tile is built in channel parts)
    constexpr int RB = C::RB2, PITCH = RB + 16, CPR = RB / 16, RPP = 256 / CPR;
    constexpr int KE = RB / EB;
    const int chunk = tid % CPR, srow = tid / CPR;
    constexpr int WPASS = 128 / RPP;   // passes for 128 rows (W3 half); W2 has PL rows
    u32x4 rw[WPASS];
    auto load_w = [&](const void* wbase, int rows, size_t row_stride_elems, size_t elem_off) {
#pragma unroll
        for (int i = 0; i < WPASS; ++i)
            if (PL == 128 || srow + i * RPP < rows)  // compile-time true for PL = 128 (all tiles have 128 rows)
                rw[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wbase) +
                                                        ((size_t)(srow + i * RPP) * row_stride_elems + elem_off + chunk * PER16) * EB);
    };
    auto store_w = [&](int buf, int rows) {
        unsigned char* const sw = stage + buf * C::SMAX;
#pragma unroll
        for (int i = 0; i < WPASS; ++i)
            if (PL == 128 || srow + i * RPP < rows) *reinterpret_cast<u32x4*>(sw + (srow + i * RPP) * PITCH + chunk * 16) = rw[i];
    };

    // this wave's 32 pixels: tile rows 2*wave, 2*wave + 1
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;
    // =========================== phase 1: t1 = relu(W1' relu(bn1 x) + b1') on the halo ===================
    // (kh: which T1W-channel part of t1 is produced: rows kh*T1W.. of W1)
    auto phase1 = [&](int kh) {
        constexpr int RB = C::RB1, PITCH = RB + 16, CPR = RB / 16, RPP = 256 / CPR;   // 4 chunks/row, 64 rows/pass
        constexpr int KE = RB / EB;
        constexpr int T1W = C::T1W, NT1 = T1W / 32;
        constexpr int XP = BT_HROWS / RPP, WP = T1W / RPP;                             // 3 and 2 (1) passes
        constexpr int X_BYTES = BT_HROWS * PITCH;
        constexpr int NSTEPS = CIN / KE;
        // wave -> (channel tile ct, row tiles rt0 .. rt0 + RT - 1)
        constexpr int RT = 6 * NT1 / 4;
        const int ct = wave % NT1, rt0 = (wave / NT1) * RT;
        const int chunk = tid % CPR, srow = tid / CPR;
        const unsigned char* xp[XP];
        const unsigned char* xq[UP ? XP : 1];
        bool xok[XP];
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int hp = srow + i * RPP;
            const int hy = hp / BT_HW, hx = hp % BT_HW;
            const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
            xok[i] = hp < BT_HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            xp[i] = xin + ((size_t)(xok[i] ? y : 0) * p.W + (xok[i] ? x : 0)) * CIN * EB;
            if constexpr (UP) xq[i] = xin2 + ((size_t)(xok[i] ? (y >> 1) : 0) * (p.W / 2) + (xok[i] ? (x >> 1) : 0)) * CIN * EB;
        }
        u32x4 rx[XP], rw[WP];
        u32x4 rb[(UP && EB == 2) ? XP : 1];
        int c0_late = 0;
        PreactCoef<T> coef;
        auto load1 = [&](int s) {
            const int c0 = s * KE + chunk * PER16;
            coef.load(p.s1, p.t1, c0);
            // out-of-image halo rows read pixel (0,0) (a valid address) and are masked to zero in store1: no branches
#pragma unroll
            for (int i = 0; i < XP; ++i) rx[i] = *reinterpret_cast<const u32x4*>(xp[i] + (size_t)c0 * EB);
            if constexpr (UP && EB == 2) {
#pragma unroll
                for (int i = 0; i < XP; ++i) rb[i] = *reinterpret_cast<const u32x4*>(xq[i] + (size_t)c0 * EB);
            }
            if constexpr (UP && EB == 4) c0_late = c0;
#pragma unroll
            for (int i = 0; i < WP; ++i)
                rw[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.w1) + ((size_t)(kh * T1W + srow + i * RPP) * CIN + c0) * EB);
        };
        auto store1 = [&](int buf) {
            unsigned char* const sx = stage + buf * C::STAGE1;
            unsigned char* const sw = sx + X_BYTES;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                u32x4 v = rx[i];
                // x = in + upsample(in2), rounded like upadd_kernel's output.  bf16 prefetched the addend with x; fp32 (no
                // registers to spare beside its 254) fetches it here: a 4x re-used, L2-resident tensor
                if constexpr (UP && EB == 2) v = add_chunk<T>(v, rb[i]);
                if constexpr (UP && EB == 4) v = add_chunk<T>(v, *reinterpret_cast<const u32x4*>(xq[i] + (size_t)c0_late * EB));
                v = preact_apply<T>(v, coef);  // bn1 + ReLU, deferred past the MFMAs
                const unsigned keep = xok[i] ? 0xffffffffu : 0u;
                v &= keep;
                *reinterpret_cast<u32x4*>(sx + (srow + i * RPP) * PITCH + chunk * 16) = v;
            }
#pragma unroll
            for (int i = 0; i < WP; ++i) *reinterpret_cast<u32x4*>(sw + (srow + i * RPP) * PITCH + chunk * 16) = rw[i];
        };
        f32x16 acc[RT];
        {
            const float bias1 = p.b1[kh * T1W + ct * 32 + l31];   // bias folded into the accumulator start value
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = bias1;
        }
        load1(0);
        store1(0);
        __syncthreads();
        for (int s = 0; s < NSTEPS; ++s) {
            const unsigned char* const sx = stage + 0 * C::STAGE1;
            const unsigned char* const sw = sx + X_BYTES;
            if (s + 1 < NSTEPS) load1(s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RB / 32; j += 2) {   // fragment pairs (j, j + 1): one 64-byte K step (mfma_pair)
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(sw + (ct * 32 + l31) * PITCH + j * 32 + half * 16);
                const u32x4 w1 = *reinterpret_cast<const u32x4*>(sw + (ct * 32 + l31) * PITCH + (j + 1) * 32 + half * 16);
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const unsigned char* const xrow = sx + ((rt0 + i) * 32 + l31) * PITCH + half * 16;
                    mfma_pair<T, false>(w0, w1, make_xpair<T>(*reinterpret_cast<const u32x4*>(xrow + j * 32), *reinterpret_cast<const u32x4*>(xrow + (j + 1) * 32)), acc[i]);
                }
            }
            if (s + 1 < NSTEPS) __syncthreads();  // every wave is done reading the only buffer
            if (s + 1 < NSTEPS) store1(0);
            __syncthreads();
        }
        // epilogue: bias + ReLU, zero outside the image, into the t1 tile (rows = halo pixels, PL channels).
        // Branch-free: every row (also the 12 pad rows, whose validity bit is 0) is written.
        const int n = ct * 32 + l31;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const unsigned vmh = (unsigned)(valid_lds[(rt0 + i) >> 1] >> (((rt0 + i) & 1) * 32 + 4 * half));
            unsigned char* const trow = t1_lds + ((rt0 + i) * 32 + 4 * half) * C::T1_PITCH + n * EB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                // select by bit mask (a ?: here makes hipcc emit one branch per register)
                const unsigned keep = 0u - ((vmh >> ro) & 1u);
                const float v = __uint_as_float(__float_as_uint(fmaxf(acc[i][r], 0.0f)) & keep);
                if constexpr (EB == 4)
                    *reinterpret_cast<float*>(trow + ro * C::T1_PITCH) = v;
                else
                    *reinterpret_cast<unsigned short*>(trow + ro * C::T1_PITCH) = Lp<T>::from_f32(v);
            }
        }
    };

    phase1(0);
    __syncthreads();

    // t2 accumulators start at b2' (channel of register r in tile m: 32m + (r&3) + 8(r>>2) + 4*half)
    f32x16 t2[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + 32 * m + 8 * q + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) t2[m][4 * q + e] = bb[e];
        }

    {
        // one staging buffer, two barriers per K-step (two workgroups per CU); kh selects the resident t1 channel part
        auto phase2 = [&](int kh) {
            constexpr int T1W = C::T1W;
            constexpr int KSTEPS = T1W / KE;         // K-steps per tap
            constexpr int NSTEPS = 9 * KSTEPS;
            load_w(p.w2, PL, PL, (size_t)kh * T1W);
            store_w(0, PL);
            __syncthreads();
            for (int s = 0; s < NSTEPS; ++s) {
                const int tap = s / KSTEPS, kc = s - tap * KSTEPS;
                const unsigned char* const sw = stage;
                if (s + 1 < NSTEPS) {
                    const int tap1 = (s + 1) / KSTEPS, kc1 = (s + 1) - tap1 * KSTEPS;
                    load_w(p.w2, PL, PL, (size_t)tap1 * PL * PL + (size_t)kh * T1W + (size_t)kc1 * KE);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int ky = tap / 3, kx = tap - 3 * ky;
                const unsigned char* const tb = t1_lds + ((py + ky) * BT_HW + (px + kx)) * C::T1_PITCH + kc * RB + half * 16;
                const unsigned char* const wrow = sw + l31 * PITCH + half * 16;
                // (requesting the fragments of two K chunks before their eight MFMA groups, pinned with sched_barrier, made the fp32
                // layer1 form 2-3 % SLOWER -- at three workgroups per CU hipcc's own interleaving is the better one)
#pragma unroll
                for (int j = 0; j < RB / 32; j += 2) {   // fragment pairs (j, j + 1): one 64-byte K step (mfma_pair)
                    const XPair<T> tp = make_xpair<T>(*reinterpret_cast<const u32x4*>(tb + j * 32), *reinterpret_cast<const u32x4*>(tb + (j + 1) * 32));
#pragma unroll
                    for (int m = 0; m < NT; ++m)
                        mfma_pair<T, true>(*reinterpret_cast<const u32x4*>(wrow + m * 32 * PITCH + j * 32), *reinterpret_cast<const u32x4*>(wrow + m * 32 * PITCH + (j + 1) * 32), tp, t2[m]);
                }
                if (s + 1 < NSTEPS) __syncthreads();
                if (s + 1 < NSTEPS) store_w(0, PL);
                __syncthreads();
            }
        };
#pragma unroll 1
        for (int kh = 0; kh < C::KSPLIT; ++kh) {
            if (kh > 0) phase1(kh);
            if (kh > 0) __syncthreads();
            phase2(kh);
        }
    }

    // ReLU on t2 (the bias was the accumulator start value)
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) t2[m][r] = fmaxf(t2[m][r], 0.0f);

    // =========================== phase 3: out = W3 t2 (+ Wd x) + b + (x) ==================================
    // halves of 128 output channels; K-step = KE channels of t2 = accumulator registers of one/two tiles
#pragma unroll 1
    for (int nh = 0; nh < CO / 128; ++nh) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nh * 128 + i * 32 + l31;
            const float bias = DS ? p.b3[n] + p.bd[n] : p.b3[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bias;
        }
        // identity skip, bf16: the residual values are requested NOW (their latency hides behind the K loop); fp32 at two
        // workgroups per CU has no registers to spare and loads them in the epilogue
        constexpr int NRES = (!DS && EB == 2) ? 32 : 1;
        unsigned xres[NRES];
        if constexpr (!DS) {
            if constexpr (EB == 2) {
                // bf16: the epilogue goes through LDS (see below): lane owns, for c = 0..7, the 16-byte chunk (lane & 15)
                // of wave pixel 4c + (lane >> 4) -> eight 16-byte residual loads
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int pw = 4 * c + (lane >> 4);
                    const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(xin) +
                        ((size_t)(ty0 + 2 * wave + (pw >> 4)) * p.W + (tx0 + (pw & 15))) * CIN + nh * 128 + (lane & 15) * 8);
                    xres[4 * c + 0] = v[0];
                    xres[4 * c + 1] = v[1];
                    xres[4 * c + 2] = v[2];
                    xres[4 * c + 3] = v[3];
                }
            }
        }
        constexpr int NSTEPS = PL / KE;
        const void* w3h = reinterpret_cast<const unsigned char*>(p.w3) + (size_t)nh * 128 * PL * EB;
        load_w(w3h, 128, PL, 0);
        store_w(0, 128);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            const unsigned char* const sw = stage + 0 * C::SMAX;
            if (s + 1 < NSTEPS) load_w(w3h, 128, PL, (size_t)(s + 1) * KE);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (EB == 4) {
                // one 32-channel tile per 128 staged bytes: tile index = s * (KE / 32) + mm
#pragma unroll
                for (int mm = 0; mm < KE / 32; ++mm)
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
                    {
                        // registers 8*q2 + 4*jj + e hold channels 16*q2 + 8*jj + 4*half + e  -> 16-byte chunk (4*q2 + 2*jj + half): jj = 0, 1 = one K step
                        const f32x16& tt = t2[s * (KE / 32) + mm];
                        const XPair<T> tp = make_xpair<T>(tt[8 * q2], tt[8 * q2 + 1], tt[8 * q2 + 2], tt[8 * q2 + 3], tt[8 * q2 + 4], tt[8 * q2 + 5], tt[8 * q2 + 6], tt[8 * q2 + 7]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned char* const wrow3 = sw + (i * 32 + l31) * PITCH + mm * 128 + (4 * q2 + half) * 16;
                            mfma_pair<T, false>(*reinterpret_cast<const u32x4*>(wrow3), *reinterpret_cast<const u32x4*>(wrow3 + 32), tp, acc[i]);
                        }
                    }
            } else {
                // per 32-channel tile two MFMAs (registers 0-7 and 8-15).  Packed W3 K order (host): position
                // 8*(2*q + half) + e  <->  channel 16*q + 8*(e>>2) + 4*half + (e&3) within the tile
#pragma unroll
                for (int mm = 0; mm < KE / 32; ++mm)
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        const f32x16& tt = t2[s * (KE / 32) + mm];
                        const u32x4 af = lp_pack8<T>(tt[8 * q2], tt[8 * q2 + 1], tt[8 * q2 + 2], tt[8 * q2 + 3], tt[8 * q2 + 4], tt[8 * q2 + 5], tt[8 * q2 + 6], tt[8 * q2 + 7]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u32x4 wf = *reinterpret_cast<const u32x4*>(sw + (i * 32 + l31) * PITCH + (mm * 32 + (2 * q2 + half) * 8) * 2);
                            acc[i] = Lp<T>::mfma(af, wf, acc[i]);
                        }
                    }
            }
            if (s + 1 < NSTEPS) __syncthreads();
            if (s + 1 < NSTEPS) store_w(0, 128);
            __syncthreads();
        }
        if constexpr (DS) {
            // skip path: acc += x_centre[32 px][CIN] * Wd[nh half][CIN]^T, both operands staged (standard orientation)
            constexpr int RBd = C::RBD, PITCHd = RBd + 16, CPRd = RBd / 16, RPPd = 256 / CPRd;   // 4 chunks/row, 64 rows/pass
            constexpr int KEd = RBd / EB, NSTEPSd = CIN / KEd;
            constexpr int XBYTES = 128 * PITCHd;
            const int chunkd = tid % CPRd, srowd = tid / CPRd;
            u32x4 rxd[2], rwd[2];
            auto loadd = [&](int s) {
                const int c0 = s * KEd + chunkd * PER16;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int pr = srowd + i * RPPd;   // pixel row 0..127 of the 8 x 16 tile
                    rxd[i] = *reinterpret_cast<const u32x4*>(xin + (((size_t)(ty0 + (pr >> 4)) * p.W + (tx0 + (pr & 15))) * CIN + c0) * EB);
                    rwd[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.wd) + ((size_t)(nh * 128 + pr) * CIN + c0) * EB);
                }
            };
            auto stored = [&](int buf) {
                unsigned char* const sx = stage + buf * C::SMAX;
                unsigned char* const sw = sx + XBYTES;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    *reinterpret_cast<u32x4*>(sx + (srowd + i * RPPd) * PITCHd + chunkd * 16) = rxd[i];
                    *reinterpret_cast<u32x4*>(sw + (srowd + i * RPPd) * PITCHd + chunkd * 16) = rwd[i];
                }
            };
            loadd(0);
            stored(0);
            __syncthreads();
            for (int s = 0; s < NSTEPSd; ++s) {
                const unsigned char* const sx = stage + 0 * C::SMAX;
                const unsigned char* const sw = sx + XBYTES;
                if (s + 1 < NSTEPSd) loadd(s + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < RBd / 32; j += 2) {   // fragment pairs (j, j + 1): one 64-byte K step (mfma_pair)
                    const unsigned char* const xrow = sx + (wave * 32 + l31) * PITCHd + half * 16;
                    const XPair<T> xp2 = make_xpair<T>(*reinterpret_cast<const u32x4*>(xrow + j * 32), *reinterpret_cast<const u32x4*>(xrow + (j + 1) * 32));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned char* const wrowd = sw + (i * 32 + l31) * PITCHd + half * 16;
                        mfma_pair<T, false>(*reinterpret_cast<const u32x4*>(wrowd + j * 32), *reinterpret_cast<const u32x4*>(wrowd + (j + 1) * 32), xp2, acc[i]);
                    }
                }
                if (s + 1 < NSTEPSd) __syncthreads();
                if (s + 1 < NSTEPSd) stored(0);
                __syncthreads();
            }
        }
        // epilogue: D[row = pixel (r&3) + 8(r>>2) + 4*half of the wave][col = channel nh*128 + 32 i + l31]
        unsigned char* const outp = reinterpret_cast<unsigned char*>(p.out) + (size_t)view * p.H * p.W * CO * EB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nh * 128 + i * 32 + l31;
            if constexpr (EB == 4) {
                if constexpr (!DS) {   // all 16 residual loads of the tile before the first store
                    float xr[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pl = (r & 3) + 8 * (r >> 2) + 4 * half;
                        xr[r] = reinterpret_cast<const float*>(xin)[((size_t)(ty0 + 2 * wave + (pl >> 4)) * p.W + (tx0 + (pl & 15))) * CIN + n];
                    }
                    if constexpr (UP) {
                        // the wave's two tile rows share ONE half-resolution row and neighbouring columns one pixel: registers
                        // r, r^1, r^8, r^9 take the same addend -> 4 loads per tile, key = bits 1 and 2 of r
                        float t4[4];
#pragma unroll
                        for (int key = 0; key < 4; ++key)
                            t4[key] = reinterpret_cast<const float*>(xin2)[((size_t)(ty0 / 2 + wave) * (p.W / 2) + tx0 / 2 + (key & 1) + 4 * (key >> 1) + 2 * half) * CIN + n];
#pragma unroll
                        for (int r = 0; r < 16; ++r) xr[r] += t4[((r >> 1) & 1) + 2 * ((r >> 2) & 1)];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] += xr[r];
                }
                if constexpr (ADD2) {   // + nearest-upsample(add2): a second fp32 add, as upadd_kernel would have done on the stored tensor
                    float t4[4];
#pragma unroll
                    for (int key = 0; key < 4; ++key)
                        t4[key] = reinterpret_cast<const float*>(lo2)[((size_t)(ty0 / 2 + wave) * (p.W / 2) + tx0 / 2 + (key & 1) + 4 * (key >> 1) + 2 * half) * CO + n];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] += t4[((r >> 1) & 1) + 2 * ((r >> 2) & 1)];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pl = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const size_t po = ((size_t)(ty0 + 2 * wave + (pl >> 4)) * p.W + (tx0 + (pl & 15))) * CO + n;
                    reinterpret_cast<float*>(outp)[po] = acc[i][r];
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
        }
        if constexpr (EB == 2) {
            // bf16 epilogue through LDS: 4-byte-per-lane global stores cost 17 % of the kernel; instead every wave
            // parks its 32 px x 128 ch tile (bf16) in its own slice of the dead t1 region and streams it out as
            // 16-byte chunks: per lane 8 x (ds_read_b128 + residual add + global_store_dwordx4), rows fully coalesced.
            // Only this wave touches its slice, so no barrier is needed (LDS operations of a wave complete in order).
            constexpr int OP = 128 * 2 + 16;                    // slice row pitch (bytes)
            unsigned char* const slice = t1_lds + wave * (32 * OP);
            const int odd = lane & 1;
            u32x4 x2[(UP || ADD2) ? 4 : 1];
            if constexpr (UP || ADD2) {  // low-resolution addend (UP: of the residual; ADD2: of the output): pixel (row wave of the
                                         // half-size tile, column pw/2); chunks c and c + 4 (the tile row below) share it
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int pw = 4 * c + (lane >> 4);
                    x2[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(UP ? xin2 : lo2) +
                        ((size_t)(ty0 / 2 + wave) * (p.W / 2) + ((tx0 + (pw & 15)) >> 1)) * CIN + nh * 128 + (lane & 15) * 8);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int rr = 2 * q + odd;                 // lane pairs exchange one register: see the stem epilogue
                    const int pl = (rr & 3) + 8 * (rr >> 2) + 4 * half;
                    const float va = acc[i][2 * q], vb = acc[i][2 * q + 1];
                    const float g = __shfl_xor(odd ? va : vb, 1, 64);
                    *reinterpret_cast<unsigned*>(slice + pl * OP + (i * 32 + (l31 & ~1)) * 2) = Lp<T>::pack2(odd ? g : va, odd ? vb : g);
                }
            unsigned short* const outs = reinterpret_cast<unsigned short*>(outp);
            u32x4 fin[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int pw = 4 * c + (lane >> 4);
                u32x4 v = *reinterpret_cast<const u32x4*>(slice + pw * OP + (lane & 15) * 16);
                if constexpr (!DS) {
                    u32x4 x4 = {xres[4 * c], xres[4 * c + 1], xres[4 * c + 2], xres[4 * c + 3]};
                    if constexpr (UP) x4 = add_chunk<T>(x4, x2[c & 3]);
                    v = add_chunk<T>(v, x4);
                }
                if constexpr (ADD2) v = add_chunk<T>(v, x2[c & 3]);   // the rounded block output + the low-resolution tensor, rounded again
                fin[c] = v;
                *reinterpret_cast<u32x4*>(outs + ((size_t)(ty0 + 2 * wave + (pw >> 4)) * p.W + (tx0 + (pw & 15))) * CO + nh * 128 + (lane & 15) * 8) = v;
            }
            if (p.pool) {
                // pooled tile row `wave`: horizontal neighbour = lane ^ 16 (pixel +-1), vertical neighbour = chunk c + 4
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
        }
    }
}

}  // namespace hgk
