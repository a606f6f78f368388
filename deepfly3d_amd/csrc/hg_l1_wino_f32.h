// fp32 layer1 tail (64 -> [conv1 elsewhere] -> 3x3 64->64 -> 1x1 64->128 + 1x1 skip convolution 64->128, 2x2 max-pool) with the 3x3 as
// WINOGRAD F(2x2, 3x3): the layer1 sibling of hg_bt_wino_f32.h (read that file's header first: the same one-wave-per-SIMD persistent design, the
// same U / V / position-GEMM structure, the same cost rules).  What differs, because the block has 64 channels:
//   * a tile is 8 x 32 output pixels = 64 patches; wave w = (cout block cb = w & 1 of the two, patch half ph = w >> 1): 32 output channels x 32
//     patches x 16 positions = 256 accumulators, as there.  K = 64 = eight chunks of 8 channels; a V chunk is 32 KB (64 patches), double-buffered;
//   * the input transform has two (patch, channel) items per lane (patch columns tx and tx + 8): 32 packed adds in one clump per chunk;
//   * phase 3: a wave owns two rows of 32 pixels = two 32-pixel MFMA row blocks; K = 64 (W3 on relu(t2)) + 64 (Wd on the raw x): eight 8 KB stages
//     that ALL fit the dead V region -- one ring fill per tile, no wait and no barrier in the K loop; rows of W3 / Wd permuted (row 32 i + l <->
//     channel 4 l + i) so that outputs and pooled outputs move as 16-byte accesses; bias b3 + bd as one float add.
// LDS: V 2 x 32 KB (ring later) | t1 halo 10 x 34 pixels x 256 B = 85 KB (t2 later) | b3 + bd [128], b2 [64].
#pragma once
#include "hg_bt_wino_f32.h"

namespace hgk {

#ifndef L1W_ABL
#define L1W_ABL 0   // development builds: 1 no U loads, 2 no input transform, 4 no chunk barrier, 16 no output transform, 32 no halo DMA for the next tile, 64 no x operand loads, 128 no stores, 256 no phase-3 MFMAs
#endif

constexpr int L1W_TW = 32, L1W_HW = L1W_TW + 2, L1W_HALO = 10 * L1W_HW;   // 340 halo pixels = 85 one-KB DMA pieces
constexpr int L1W_CHUNKS = 8;
constexpr int L1W_U_BYTES = L1W_CHUNKS * 4 * 2 * 4 * 1024;   // [chunk][pass][cout block][column] x 1 KB fragments = 256 KiB
constexpr int L1W_W_BYTES = 4 * BR_STAGE_BYTES;              // W3 (then Wd): four 16-float K slices x 128 permuted rows
constexpr int L1W_STREAM_BYTES = L1W_U_BYTES + 2 * L1W_W_BYTES;
constexpr int L1W_V_BYTES = 32 * 1024;
constexpr int L1W_T1_OFF = 2 * L1W_V_BYTES;
constexpr int L1W_T1_BYTES = L1W_HALO * 256;                 // 87 040
constexpr int L1W_B_OFF = L1W_T1_OFF + L1W_T1_BYTES;
constexpr int L1W_LDS_BYTES = L1W_B_OFF + 512 + 256;
static_assert(L1W_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(8 * 32 * 256 <= L1W_T1_BYTES, "t2 [256 pixels][64 channels] lives in the t1 region");

// W2' [9][64 cout][64 cin] -> U stream: fragment (chunk c, pass e, cout block cb, column j): lane (l31, half) holds U'_{i j}[32 cb + l31][8 c + 4 half + e],
// i = 0..3, with the sign convention of bt_wino_pack_kernel (row / column 2 negated)
__global__ __launch_bounds__(256) void l1_wino_pack_u_kernel(const float* __restrict__ w2, float* __restrict__ ustream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 64 * 64) return;
    const int co = idx >> 6, ci = idx & 63;
    double g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w2[((size_t)(ky * 3 + kx) * 64 + co) * 64 + ci];
    double t[4][3];   // G g
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        t[0][kx] = g[0][kx];
        t[1][kx] = 0.5 * (g[0][kx] + g[1][kx] + g[2][kx]);
        t[2][kx] = 0.5 * (g[0][kx] - g[1][kx] + g[2][kx]);
        t[3][kx] = g[2][kx];
    }
    const int c = ci >> 3, half = (ci >> 2) & 1, e = ci & 3, cb = co >> 5, l31 = co & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double u[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};   // (G g) G^T
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ustream[((size_t)(((c * 4 + e) * 2 + cb) * 4 + j) * 64 + half * 32 + l31) * 4 + i] = (float)(((i == 2) != (j == 2)) ? -u[j] : u[j]);
    }
}
// W [128 cout][64 cin] (W3 or the skip convolution's Wd) -> four stage images (16-float K slice k8; 128 rows x 64 bytes, br_swz), row 32 i + l = channel 4 l + i
__global__ __launch_bounds__(256) void l1_wino_pack_w_kernel(const float* __restrict__ w, unsigned char* __restrict__ stream) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 4 * 512) return;
    const int k8 = idx >> 9, rem = idx & 511, c = rem & 3, r = rem >> 2;
    const int ch = 4 * (r & 31) + (r >> 5);
    *reinterpret_cast<u32x4*>(stream + (size_t)k8 * BR_STAGE_BYTES + br_swz(r, c)) = *reinterpret_cast<const u32x4*>(w + (size_t)ch * 64 + 16 * k8 + 4 * c);
}

// BtRingArgs: in = x [V, H, W, 64], t1in [V, H, W, 64], out (may be null: pooled output only) [V, H, W, 128], pool (may be null) [V, H/2, W/2, 128],
// w2d = U | W3 stages | Wd stages, zeros, b2 [64], b3 [128], bd [128].  H % 8 == 0, W % 32 == 0.
__global__ __launch_bounds__(256, 1) void layer1_wino_f32_kernel(BtRingArgs p) {
    using T = float;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const t1_lds = smem + L1W_T1_OFF;
    float* const b3_lds = reinterpret_cast<float*>(smem + L1W_B_OFF);
    float* const b2_lds = b3_lds + 128;
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned t1_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)t1_lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 1, ph = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles_x = p.W / L1W_TW, tiles_y = p.H / BT_TH;
    const int ntiles = p.V * tiles_y * tiles_x;
    auto tile_of = [&](int vb, int& tx0, int& ty0, int& view) {   // persistent, XCD-aware: as bottleneck_wino_f32_kernel
        const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
        int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
        tx0 = (b % tiles_x) * L1W_TW;
        b /= tiles_x;
        ty0 = (b % tiles_y) * BT_TH;
        view = b / tiles_y;
    };
    // the t1 halo tile (10 x 34 pixels x 64 channels) by LDS-DMA: piece pc = halo pixels 4 pc .. 4 pc + 3, lane -> (pixel, 16-byte slot) fetching the
    // chunk that belongs there (slot ^ (hx & 15)); pixels outside the image from the page of zeros
    unsigned uoff = (unsigned)(lane * 16);   // the one lane-derived register that lives across phase 2 (see bottleneck_wino_f32_kernel)
    asm volatile("" : "+v"(uoff));
    auto t1_issue = [&](int tx0, int ty0, int view) {
        int lane_ = (int)(uoff >> 4);
        asm volatile("" : "+v"(lane_));   // (recomputed per call, lane values and uniform values alike: see bottleneck_wino_f32_kernel)
        int wave_ = wave;
        asm volatile("" : "+s"(wave_));
        const unsigned char* const tin = reinterpret_cast<const unsigned char*>(p.t1in) + (size_t)view * p.H * p.W * 256;
        const unsigned char* const zer = reinterpret_cast<const unsigned char*>(p.zeros);
        const int q = lane_ >> 4, slot = lane_ & 15;
#pragma unroll
        for (int k = 0; k < 22; ++k) {
            const int pc = wave_ + 4 * k;
            if (pc < L1W_HALO / 4) {
                const int hp = 4 * pc + q;
                const int hy = hp / L1W_HW, hx = hp - hy * L1W_HW;
                const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
                const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                const unsigned c16 = (unsigned)((slot ^ (hx & 15)) << 4);
                const unsigned in_view = (unsigned)((y * p.W + x) * 256) + c16;   // (a view's t1 is < 4 GB)
                br_glds_piece64((ok ? tin : zer) + (ok ? in_view : c16), t1_addr + (unsigned)(pc * 1024));
            }
        }
    };
    const unsigned char* const ubase = reinterpret_cast<const unsigned char*>(p.w2d) + (size_t)cb * 4096;
    auto uload = [&](int c, int e, f32x4 (&dst)[4]) { wn_uload4(dst, ubase + (size_t)(c * 4 + e) * 8192, uoff); };

    // input transform: wave w builds patch row ty = w; lane -> (patch columns tx = (lane & 7) and + 8, channel quad (lane >> 3) & 1, channel lane >> 4)
    const int ptx = lane & 7, pkq = (lane >> 3) & 1, pe = lane >> 4;
    unsigned rd[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const int hx = 2 * (ptx + 8 * it) + bb, hp = (2 * wave) * L1W_HW + hx;
            rd[it][bb] = (unsigned)(hp * 256 + ((pkq ^ (hx & 15)) << 4) + 4 * pe);
        }
    // V chunk image [channel e 4][column j 4][quad kq 2][patch half 2][patch 32][row i 4] floats
    const unsigned vwr = (unsigned)(pe * 8192 + pkq * 1024 + (((8 * wave + ptx) ^ (pkq << 3)) << 4));   // (+ 512 for the lane's second item)
    const unsigned vrd = (unsigned)(half * 1024 + ph * 512 + ((l31 ^ (half << 3)) << 4));
    f32x2 tP[2][4], tQ[2][4];
    unsigned ta[2][4];
    auto t_addr = [&](int c) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) ta[it][bb] = (rd[it][bb] ^ (unsigned)(c << 5)) + t1_addr;
    };
    auto t_read = [&](int it, int bb) {
        typedef const __attribute__((address_space(3))) float* lds_f;
        tP[it][bb] = f32x2{*(lds_f)(size_t)ta[it][bb], *(lds_f)(size_t)(ta[it][bb] + L1W_HW * 256)};
        tQ[it][bb] = f32x2{*(lds_f)(size_t)(ta[it][bb] + 2 * L1W_HW * 256), *(lds_f)(size_t)(ta[it][bb] + 3 * L1W_HW * 256)};
    };
    f32x2 vt[2][4], vs[2][4];
    auto t_transform = [&](int c_next_addr) {   // 32 packed adds + eight addresses in one clump
        wn_transform(tP[0], tQ[0], vt[0], vs[0]);
        wn_transform(tP[1], tQ[1], vt[1], vs[1]);
        t_addr(c_next_addr);
    };
    auto v_store = [&](int buf, int j) {   // column j of both items (in phase 2: one column per MFMA group, see bottleneck_wino_f32_kernel)
        unsigned char* const dst = smem + buf * L1W_V_BYTES + vwr;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            *reinterpret_cast<f32x2*>(dst + it * 512 + j * 2048) = vt[it][j];
            *reinterpret_cast<f32x2*>(dst + it * 512 + j * 2048 + 8) = vs[it][j];
        }
    };
    auto t_transform_write = [&](int buf, int c_next_addr) {   // (tile entry: clump, then the stores)
        t_transform(c_next_addr);
#pragma unroll
        for (int j = 0; j < 4; ++j) v_store(buf, j);
    };
    auto ring_issue_all = [&]() {   // W3's four stages -> slots 0 .. 3, Wd's -> 4 .. 7; this wave copies pieces 2 wave, 2 wave + 1 of each
        unsigned wvoff = (unsigned)wave * 2048u + uoff;
        asm volatile("" : "+v"(wvoff));
#pragma unroll
        for (int k = 0; k < 8; ++k)
            br_glds_stage(reinterpret_cast<const unsigned char*>(p.w2d) + L1W_U_BYTES + (size_t)k * BR_STAGE_BYTES, wvoff, ring_addr + (unsigned)(k * BR_STAGE_BYTES + wave * 2048));
    };

    int vb = blockIdx.x;
    int tx0, ty0, view;
    tile_of(vb, tx0, ty0, view);
    t1_issue(tx0, ty0, view);
    if (tid < 128) b3_lds[tid] = p.b3[tid] + p.bd[tid];   // ONE float add, as the direct kernels
    if (tid < 64) b2_lds[tid] = p.b2[tid];
    bool first = true;

#pragma unroll 1
    for (;;) {
        const int vbn = vb + (int)gridDim.x;
        const bool has_next = vbn < ntiles;
        int ntx0 = 0, nty0 = 0, nview = 0;
        if (has_next) tile_of(vbn, ntx0, nty0, nview);
        const size_t ptile = ((size_t)view * p.H + ty0 + 2 * wave) * p.W + tx0;                          // first pixel of this wave's two rows
        const size_t hptile = ((size_t)view * (p.H / 2) + ty0 / 2 + wave) * (p.W / 2) + tx0 / 2;       // ... of its one half-resolution row

        f32x16 acc[16];   // (their first MFMAs take a zero addend)
        f32x4 ufr[4][4];
        uload(0, 0, ufr[0]);
        uload(0, 1, ufr[1]);
        uload(0, 2, ufr[2]);
        if (first) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // the first tile's halo (later tiles': requested a phase 3 ago)
        first = false;
        br_barrier();
        asm volatile("" : "+v"(rd[0][0]), "+v"(rd[0][1]), "+v"(rd[0][2]), "+v"(rd[0][3]), "+v"(rd[1][0]), "+v"(rd[1][1]), "+v"(rd[1][2]), "+v"(rd[1][3]));
        t_addr(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            t_read(0, g);
            t_read(1, g);
        }
        t_transform_write(0, 1);
        br_barrier();

        // ---- phase 2: 8 chunks x 4 passes x 4 columns x 4 rows; chunk c reads V buffer c & 1 and builds V(c + 1) into the other -------------------
        f32x4 vf[2][4];
        // LAST (the peeled final chunk) requests and builds nothing: a fragment load whose destination registers are dead to the compiler would land,
        // asynchronously, in whatever those registers hold by then
        auto chunk = [&](int c, auto br_tag, auto first_tag, auto last_tag) {
            constexpr int BR = decltype(br_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
            const int cn = c + 1, c2 = c + 2 < L1W_CHUNKS ? c + 2 : L1W_CHUNKS - 1;
            const unsigned char* const vb_ = smem + BR * L1W_V_BYTES + vrd;
#pragma unroll
            for (int g = 0; g < 4; ++g) vf[0][g] = *reinterpret_cast<const f32x4*>(vb_ + g * 2048);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __builtin_amdgcn_sched_barrier(0);
                if (!(L1W_ABL & 1)) {
                    if (!LAST || e < 2) wn_uwait<8>(ufr[e]);
                    else if (e == 2) wn_uwait<4>(ufr[e]);
                    else wn_uwait<0>(ufr[e]);
                    if (e == 0) uload(c, 3, ufr[3]);
                    else if (!LAST) uload(cn, e - 1, ufr[e - 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (e == 2 && !LAST && !(L1W_ABL & 2)) {
                    t_transform(c2);   // V(c + 1) from the patches read in pass 0; addresses for the reads of chunk c + 1's pass 0
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (e < 3) vf[(e + 1) & 1][g] = *reinterpret_cast<const f32x4*>(vb_ + (e + 1) * 8192 + g * 2048);
                    if (e == 0 && !LAST && !(L1W_ABL & 2)) {
                        t_read(0, g);
                        t_read(1, g);
                    }
                    if (e == 2 && !LAST && !(L1W_ABL & 2)) v_store(BR ^ 1, g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (FIRST && e == 0)
                            acc[4 * i + g] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[e][g][i], vf[e & 1][g][i], f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
                        else
                            acc[4 * i + g] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[e][g][i], vf[e & 1][g][i], acc[4 * i + g], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!(L1W_ABL & 4)) br_barrier();
        };
        chunk(0, std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{});
#pragma unroll 1
        for (int c = 1; c < L1W_CHUNKS - 1; c += 2) {
            chunk(c, std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
            chunk(c + 1, std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{});
        }
        chunk(L1W_CHUNKS - 1, std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{});

        // ---- phase 3 set-up: lane constants from an opaque copy of the lane index (see bottleneck_wino_f32_kernel), the eight weight stages into the
        //      dead V region, output transform, t2 across -------------------------------------------------------------------------------------------
        int lane3 = (int)(uoff >> 4);
        asm volatile("" : "+v"(lane3));
        const int half3 = lane3 >> 5, l31_3 = lane3 & 31;
        const unsigned char* const wf0 = ring + br_swz(l31_3, half3);
        const unsigned char* const wf1 = ring + br_swz(l31_3, 2 + half3);
        const unsigned lane_full = (unsigned)((4 * half3 * 128 + 4 * l31_3) * 4);   // pixel 4 half of the register's group, channels 4 l31 ..
        const unsigned lane_half = (unsigned)((2 * half3 * 128 + 4 * l31_3) * 4);   // half-resolution pixel 2 half of the quad, channels 4 l31 ..
        ring_issue_all();
        if (L1W_ABL & 16) {
            float keep = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) keep += acc[k][0];
            *reinterpret_cast<float*>(t1_lds + lane3 * 4) = keep;
        } else {
            const int ty = l31_3 >> 3, txl = l31_3 & 7;
            const unsigned wbase = (unsigned)((64 * ty + 16 * ph + 2 * txl) * 256 + ((((8 * cb + half3) ^ (2 * txl) ^ (ty & 1)) & 15) << 4));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 y[2][2];
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b2_lds + 32 * cb + 8 * q + 4 * half3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    float s[2][4];
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
                    for (int bb2 = 0; bb2 < 2; ++bb2)
                        *reinterpret_cast<f32x4*>(t1_lds + (wbase ^ (unsigned)(((2 * q) ^ bb2) << 4)) + (32 * a + bb2) * 256) = y[a][bb2];
            }
        }
        br_wait_vm(0);   // the eight stages (and whatever is older)
        br_barrier();
        f32x16 t2[2][2];   // [pixel block mb][32-channel tile m]: registers 4 g + e <-> channels 32 m + 8 g + 4 half + e of pixel (2 wave + (l31 >> 4), 16 mb + (l31 & 15))
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int py = 2 * wave + (l31_3 >> 4), px = 16 * mb + (l31_3 & 15);
            const unsigned rbase = (unsigned)((py * L1W_TW + px) * 256 + ((half3 ^ (px & 15) ^ (wave & 1)) << 4));
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(t1_lds + (rbase ^ (unsigned)((8 * m + 2 * g) << 4)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) t2[mb][m][4 * g + e] = v[e];
                }
        }
        br_barrier();   // every wave holds its t2: the t1 region may take the next tile's halo
        if (has_next && !(L1W_ABL & 32)) t1_issue(ntx0, nty0, nview);
        // the skip convolution's A operand: x of the lane's pixel, [mb][2 k8 + jj] = channels 16 k8 + 8 jj + 4 half ..
        // (inline assembly, issued HERE: left to hipcc every pair of these loads sat right in front of the MFMAs that read it -- eight exposed round
        // trips per tile)
        f32x4 xop[2][8];
        {
            const unsigned xlane = (unsigned)((((l31_3 >> 4) * p.W + (l31_3 & 15)) * 64 + 4 * half3) * 4);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r4 = 0; r4 < 2; ++r4)
                    if (L1W_ABL & 64) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) xop[mb][4 * r4 + j] = f32x4{1.0f, 0.0f, 0.0f, 0.0f};
                    } else
                    wn_xload4s<32>(*reinterpret_cast<f32x4(*)[4]>(&xop[mb][4 * r4]), reinterpret_cast<const unsigned char*>(p.in) + ptile * 256 + (16 * mb) * 256 + (32 * r4) * 4, xlane);
        }

        // ---- phase 3: out = W3 relu(t2) + Wd x + (b3 + bd): eight resident stages, no wait, no barrier ------------------------------------------------
        f32x16 o[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float bias = b3_lds[4 * l31_3 + i];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[mb][i][r] = bias;
            }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k8 = s & 3, tile = k8 >> 1, q2 = k8 & 1;
            if (s == 4 && !(L1W_ABL & 64)) {   // the x operand (requested 256 MFMAs ago; nothing younger is in flight)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r4 = 0; r4 < 2; ++r4) wn_uwait<0>(*reinterpret_cast<f32x4(*)[4]>(&xop[mb][4 * r4]));
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 wf = *reinterpret_cast<const f32x4*>((jj ? wf1 : wf0) + s * BR_STAGE_BYTES + i * 2048);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        if (L1W_ABL & 256) {
                            o[mb][i][0] += wf[0] + t2[mb][tile][s] + xop[mb][s][0];
                        } else if (s < 4) {
                            mfma_quad<T>(t2[mb][tile][8 * q2 + 4 * jj], t2[mb][tile][8 * q2 + 4 * jj + 1], t2[mb][tile][8 * q2 + 4 * jj + 2], t2[mb][tile][8 * q2 + 4 * jj + 3], wf, o[mb][i]);
                        } else {
                            const f32x4 xa = xop[mb][2 * k8 + jj];
                            mfma_quad<T>(xa[0], xa[1], xa[2], xa[3], wf, o[mb][i]);
                        }
                    }
                }
        }
        // epilogue: register r of the four tiles = channels 4 l31 .. 4 l31 + 3 of pixel (r & 3) + 8 (r >> 2) + 4 half of the block's two rows; by 2x2 quads
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int key = 0; key < 4; ++key) {
                const int r0 = 2 * (key & 1) + 4 * (key >> 1);
                f32x4 ov[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = r0 + (t & 1) + 8 * (t >> 1);
                    ov[t] = f32x4{o[mb][0][r], o[mb][1][r], o[mb][2][r], o[mb][3][r]};
                    const int pl0 = (r & 3) + 8 * (r >> 2);
                    if (p.out && !(L1W_ABL & 128))
                        *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.out) + (ptile + (size_t)(pl0 >> 4) * p.W + 16 * mb + (pl0 & 15)) * 512 + lane_full) = ov[t];
                }
                if (p.pool && (!(L1W_ABL & 128) || ov[0][0] == 12345.678f)) {
                    f32x4 m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(ov[0][e], ov[1][e]), fmaxf(ov[2][e], ov[3][e]));
                    *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.pool) + (hptile + 8 * mb + (key & 1) + 4 * (key >> 1)) * 512 + lane_half) = m;
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
