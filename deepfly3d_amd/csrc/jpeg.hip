// Input front-end, SURVEY.md 8(f) row 1: baseline JPEG -> luma plane on the device, so that the host only reads
// file bytes (the reference decodes with libjpeg inside df2d's DataLoader workers; frames are written by ffmpeg,
// reference df3d/core.py:446-459).  ITU-T T.81 baseline sequential DCT, Huffman, 8 bit; libjpeg's "islow"
// inverse DCT bit for bit (oracle/jpeg_oracle.c is the CPU restatement, itself pinned against libjpeg-turbo).
//
//   jpeg_parse_kernel    one workgroup per file: marker walk out of an LDS window, tables into a per-file
//                        descriptor, then the entropy-coded segment is un-stuffed (FF00 -> FF) cooperatively
//                        into a clean byte stream (tile-wise flag / scan / compact)
//   jpeg_huffman_par_kernel  256 lanes per file, self-synchronising chunks: the fast path
//   jpeg_huffman_kernel  the exact fall-back (restart intervals, invalid or truncated streams, no convergence): one
//                        64-lane wave per file, wave-uniform control flow: the bit buffer lives in scalar
//                        registers, the next 256 stream bytes in one VGPR (v_readlane), 9-bit code look-up
//                        tables in LDS; lane i keeps coefficient i of the current block, so a decoded value is a
//                        compare + select and a finished luma block leaves as one coalesced 128-byte store
//   jpeg_idct_kernel     one thread per 8x8 luma block: de-quantise, 2 x 8 one-dimensional LL&M passes, clamp;
//                        adds the parallel decoder's per-lane DC offsets, writes the per-file status
//
// Only the luma component is reconstructed (the rig's cameras are monochrome; chroma blocks are entropy-decoded
// and dropped).  Integer work: results are bit-exact.  The coefficient buffer between the Huffman kernels and the
// IDCT holds 64 shorts per luma block in zig-zag order.
#include <cstdint>

#include "common.h"

namespace jpg {

enum { ST_OK = 0, ST_TRUNCATED = 1, ST_NOT_JPEG = 2, ST_UNSUPPORTED = 3, ST_CORRUPT = 4, ST_SHAPE = 5 };

struct Meta {
    int status;
    int width, height, ncomp;
    int hmax, vmax;
    int restart_interval;
    int comp_h[4], comp_v[4], comp_tq[4], comp_td[4], comp_ta[4];
    unsigned scan_pos;   // byte offset of the entropy-coded data inside the file
    unsigned clean_len;  // bytes of the un-stuffed stream
    int par_done;        // > 0: the parallel Huffman kernel finished this file (= its synchronisation passes); <= 0: the
                         // sequential kernel decodes it (-1 no convergence, -2 stream ends early, -3 invalid symbols)
    int mcus_x, mcus_y;
    int ybw, ybh;  // luma block grid (MCU padded)
    unsigned short q[4][64];  // natural order
    unsigned char qpresent[4];
    unsigned char dht_present[8];  // [tc*4 + th]
    unsigned char dht_counts[8][16];
    unsigned char dht_vals[8][256];
    // parallel Huffman kernel (par_done > 0): its lane l decoded the DCs of luma blocks [dc_cnt[l-1], dc_cnt[l]) and stored them
    // as running sums that start at zero; dc_sum[l-1] (inclusive scan of the lanes' sums) is what the IDCT adds back
    unsigned dc_cnt[256];
    int dc_sum[256];
};

__constant__ unsigned char ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---------------------------------------------------------------------------------------------------------
// kernel 1: header parse + un-stuffing.  256 threads per file; every thread runs the (uniform) marker walk on an
// LDS window of the file, thread 0 writes the descriptor.
// ---------------------------------------------------------------------------------------------------------
constexpr int WIN = 4096;
constexpr int PT = 256;

struct Window {
    const unsigned char* g;
    unsigned n, base;
    unsigned char* lds;
    __device__ void stage(unsigned pos) {
        __syncthreads();
        base = pos & ~15u;
        for (unsigned i = threadIdx.x * 16; i < WIN; i += PT * 16) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (base + i < n) v = *reinterpret_cast<const uint4*>(g + base + i);  // files are padded to 16 bytes
            *reinterpret_cast<uint4*>(lds + i) = v;
        }
        __syncthreads();
    }
    __device__ unsigned get(unsigned pos) {  // uniform
        if (pos - base >= (unsigned)WIN) stage(pos);
        return lds[pos - base];
    }
    __device__ unsigned get16(unsigned pos) { return (get(pos) << 8) | get(pos + 1); }
};

__global__ __launch_bounds__(PT) void jpeg_parse_kernel(const unsigned char* __restrict__ files, const unsigned* __restrict__ offsets,
                                                        const unsigned* __restrict__ sizes, int expect_w, int expect_h,
                                                        Meta* __restrict__ metas,
                                                        unsigned char* __restrict__ clean) {
    __shared__ __attribute__((aligned(16))) unsigned char win_lds[WIN];
    __shared__ unsigned s_scan[PT];
    __shared__ unsigned s_total;
    const int img = blockIdx.x;
    const unsigned off = offsets[img], n = sizes[img];
    const unsigned char* d = files + off;
    Meta* M = metas + img;
    const int tid = threadIdx.x;
    Window w{d, n, 0xffffffffu, win_lds};
    w.stage(0);

    // descriptor defaults (thread 0 owns all scalar fields)
    if (tid == 0) {
        M->status = ST_OK;
        M->restart_interval = 0;
        M->clean_len = 0;
        M->par_done = 0;
        M->hmax = M->vmax = 1;
        for (int i = 0; i < 4; ++i) M->qpresent[i] = 0;
        for (int i = 0; i < 8; ++i) M->dht_present[i] = 0;
    }
    int status = ST_OK;
    unsigned i = 2, scan_pos = 0;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, have_sof = 0;
    int ch[4] = {1, 1, 1, 1}, cv[4] = {1, 1, 1, 1}, cid[4] = {0, 0, 0, 0};
    if (n < 4 || w.get(0) != 0xFF || w.get(1) != 0xD8) status = ST_NOT_JPEG;
    while (status == ST_OK && scan_pos == 0) {
        if (i + 4 > n) { status = ST_TRUNCATED; break; }
        if (w.get(i) != 0xFF) { status = ST_CORRUPT; break; }
        while (i < n && w.get(i) == 0xFF) ++i;
        if (i >= n) { status = ST_TRUNCATED; break; }
        const unsigned m = w.get(i++);
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) { status = ST_CORRUPT; break; }
        if (i + 2 > n) { status = ST_TRUNCATED; break; }
        const unsigned L = w.get16(i);
        if (L < 2 || i + L > n) { status = ST_TRUNCATED; break; }
        const unsigned s = i + 2, sl = L - 2;
        if (m == 0xDB) {
            unsigned j = 0;
            while (j < sl && status == ST_OK) {
                const unsigned b = w.get(s + j);
                const unsigned pq = b >> 4, tq = b & 15;
                if (tq > 3 || pq > 1) { status = ST_CORRUPT; break; }
                ++j;
                if (j + (pq ? 128 : 64) > sl) { status = ST_CORRUPT; break; }
                for (int k = 0; k < 64; ++k) {
                    const unsigned v = pq ? w.get16(s + j) : w.get(s + j);
                    j += pq ? 2 : 1;
                    if (tid == 0) M->q[tq][ZIGZAG[k]] = (unsigned short)v;
                }
                if (tid == 0) M->qpresent[tq] = 1;
            }
        } else if (m == 0xC4) {
            unsigned j = 0;
            while (j < sl && status == ST_OK) {
                if (j + 17 > sl) { status = ST_CORRUPT; break; }
                const unsigned b = w.get(s + j);
                const unsigned tc = b >> 4, th = b & 15;
                if (tc > 1 || th > 3) { status = ST_CORRUPT; break; }
                unsigned nv = 0;
                for (int k = 0; k < 16; ++k) {
                    const unsigned c = w.get(s + j + 1 + k);
                    nv += c;
                    if (tid == 0) M->dht_counts[tc * 4 + th][k] = (unsigned char)c;
                }
                if (nv > 256 || j + 17 + nv > sl) { status = ST_CORRUPT; break; }
                for (unsigned k = 0; k < nv; ++k) {
                    const unsigned v = w.get(s + j + 17 + k);
                    if (tid == 0) M->dht_vals[tc * 4 + th][k] = (unsigned char)v;
                }
                if (tid == 0) M->dht_present[tc * 4 + th] = 1;
                j += 17 + nv;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6) { status = ST_CORRUPT; break; }
            if (w.get(s) != 8) { status = ST_UNSUPPORTED; break; }
            height = (int)w.get16(s + 1);
            width = (int)w.get16(s + 3);
            ncomp = (int)w.get(s + 5);
            if (ncomp < 1 || ncomp > 4) { status = ST_UNSUPPORTED; break; }
            if (sl < 6 + 3u * ncomp || width == 0 || height == 0) { status = ST_CORRUPT; break; }
            for (int c = 0; c < ncomp; ++c) {
                cid[c] = (int)w.get(s + 6 + 3 * c);
                const unsigned hv = w.get(s + 7 + 3 * c);
                const unsigned tq = w.get(s + 8 + 3 * c);
                ch[c] = hv >> 4;
                cv[c] = hv & 15;
                if (ch[c] < 1 || ch[c] > 4 || cv[c] < 1 || cv[c] > 4 || tq > 3) status = ST_CORRUPT;
                hmax = max(hmax, ch[c]);
                vmax = max(vmax, cv[c]);
                if (tid == 0) M->comp_tq[c] = (int)tq;
            }
            have_sof = 1;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            status = ST_UNSUPPORTED;
        } else if (m == 0xDD) {
            if (sl < 2) { status = ST_CORRUPT; break; }
            const unsigned ri = w.get16(s);
            if (tid == 0) M->restart_interval = (int)ri;
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1) { status = ST_CORRUPT; break; }
            const int ns = (int)w.get(s);
            if (ns != ncomp) { status = ST_UNSUPPORTED; break; }
            if (sl < 1 + 2u * ns + 3) { status = ST_CORRUPT; break; }
            for (int k = 0; k < ns; ++k) {
                const int id = (int)w.get(s + 1 + 2 * k);
                const unsigned t = w.get(s + 2 + 2 * k);
                if (id != cid[k]) status = ST_UNSUPPORTED;
                if ((t >> 4) > 3 || (t & 15) > 3) status = ST_CORRUPT;
                if (tid == 0) {
                    M->comp_td[k] = (int)(t >> 4);
                    M->comp_ta[k] = (int)(t & 15);
                }
            }
            if (status == ST_OK) scan_pos = i + L;
            break;
        }
        i += L;
    }
    if (status == ST_OK && ((expect_w > 0 && expect_w != width) || (expect_h > 0 && expect_h != height))) status = ST_SHAPE;
    if (ncomp == 1) ch[0] = cv[0] = hmax = vmax = 1;
    const int mcus_x = status == ST_OK ? (width + 8 * hmax - 1) / (8 * hmax) : 0;
    const int mcus_y = status == ST_OK ? (height + 8 * vmax - 1) / (8 * vmax) : 0;
    if (tid == 0) {
        M->width = width;
        M->height = height;
        M->ncomp = ncomp;
        M->hmax = hmax;
        M->vmax = vmax;
        for (int c = 0; c < 4; ++c) {
            M->comp_h[c] = ch[c];
            M->comp_v[c] = cv[c];
        }
        M->scan_pos = scan_pos;
        M->mcus_x = mcus_x;
        M->mcus_y = mcus_y;
        M->ybw = mcus_x * ch[0];
        M->ybh = mcus_y * cv[0];
        M->status = status;
    }
    if (status != ST_OK) return;

    // ---- un-stuff [scan_pos, n) into clean + off: drop every 0x00 that follows a 0xFF -----------------
    unsigned char* out = clean + off;
    unsigned produced = 0;
    const unsigned start = scan_pos & ~15u;
    for (unsigned tile = start; tile < n; tile += PT * 16) {
        const unsigned p0 = tile + tid * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        unsigned prev = 0;
        if (p0 < n) {
            v = *reinterpret_cast<const uint4*>(d + p0);
            if (p0 > 0) prev = d[p0 - 1];
        }
        unsigned char by[16];
        *reinterpret_cast<uint4*>(by) = v;
        unsigned keep = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned pos = p0 + k;
            const bool ok = pos >= scan_pos && pos < n && !(by[k] == 0 && prev == 0xFF);
            keep |= (ok ? 1u : 0u) << k;
            prev = by[k];
        }
        const unsigned cnt = __popc(keep);
        // exclusive scan of cnt over the 256 threads
        s_scan[tid] = cnt;
        __syncthreads();
        for (int o = 1; o < PT; o <<= 1) {
            const unsigned a = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += a;
            __syncthreads();
        }
        unsigned dst = produced + s_scan[tid] - cnt;
        if (tid == PT - 1) s_total = s_scan[tid];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (keep & (1u << k)) out[dst++] = by[k];
        }
        __syncthreads();
        produced += s_total;
        __syncthreads();
    }
    // zero padding behind the stream so that the decoder's window loads read defined bytes
    for (unsigned p = produced + tid; p < n && p < produced + 1024; p += PT) out[p] = 0;
    if (tid == 0) M->clean_len = produced;
}

// ---------------------------------------------------------------------------------------------------------
// Huffman tables in LDS (shared by both decode kernels)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned rfl(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

template <int LB, int NTAB = 8>
struct HuffLds {
    unsigned short lut[NTAB][1 << LB];  // (len << 8) | symbol, 0 = code longer than LB bits (or invalid)
    unsigned limit[NTAB][18];           // (maxcode[l] + 1) << (16 - l): left-justified 16-bit compare; [17] = 65536
    int valoff[NTAB][17];               // valptr[l] - mincode[l]
    unsigned char vals[NTAB][256];
};

// Table slot (tc * 4 + th) -> row of a HuffLds: the identity, or the compact numbering of the tables a scan refers to
struct SlotMap {
    int row[8];
};

// Build every present table that has a row; all NT threads of the workgroup call this.  Returns non-zero for an invalid
// table or when a table a component refers to is missing.
template <int LB, int NT, int NTAB>
__device__ int build_tables(const Meta* M, HuffLds<LB, NTAB>& T, const SlotMap& map, int tid) {
    int bad = 0;
    const int lane = tid & 63;
    // the descriptor's small fields through one load per wave each (lane l holds entry l), not one dependent load per entry
    const unsigned present = lane < 8 ? M->dht_present[lane] : 0u;
    for (int slot = 0; slot < 8; ++slot) {
        const int t = map.row[slot];
        if (t < 0 || !__builtin_amdgcn_readlane((int)present, slot)) continue;
        const unsigned counts = lane < 16 ? M->dht_counts[slot][lane] : 0u;
        for (int i = tid; i < (1 << LB); i += NT) T.lut[t][i] = 0;
        for (int i = tid; i < 256; i += NT) T.vals[t][i] = M->dht_vals[slot][i];
        __syncthreads();
        int code = 0, k = 0;
#pragma unroll
        for (int l = 1; l <= 16; ++l) {
            const int c = __builtin_amdgcn_readlane((int)counts, l - 1);
            if (tid == 0) T.valoff[t][l] = k - code;
            // an over-subscribed length (untrusted DHT) would index past the 2^LB-entry table: flag it BEFORE the fill
            const bool fits = code + c <= (1 << l) && k + c <= 256;
            if (!fits) bad = 1;
            if (l <= LB && fits) {  // symbols k .. k+c-1 have the codes code .. code+c-1 of length l
                const int span = 1 << (LB - l);
                for (int e = tid; e < c * span; e += NT) {
                    const int j = e >> (LB - l);
                    T.lut[t][((code + j) << (LB - l)) + (e - j * span)] = (unsigned short)((l << 8) | T.vals[t][(k + j) & 255]);
                }
            }
            k += c;
            code += c;
            if (code > (1 << l)) bad = 1;
            if (tid == 0) T.limit[t][l] = (unsigned)code << (16 - l);
            code <<= 1;
        }
        if (tid == 0) T.limit[t][17] = 0x10000u;
        if (k > 256) bad = 1;
        __syncthreads();
    }
    const int ncomp = (int)rfl((unsigned)M->ncomp);
    for (int c = 0; c < ncomp; ++c) {
        if (!rfl(M->qpresent[M->comp_tq[c]]) || !rfl(M->dht_present[M->comp_td[c]]) || !rfl(M->dht_present[4 + M->comp_ta[c]])) bad = 1;
    }
    return bad;
}

// ---------------------------------------------------------------------------------------------------------
// kernel 2a: parallel Huffman decode by self-synchronisation, 256 lanes per file.
//
// The un-stuffed stream is cut into 256 equal chunks.  Pass 0: every lane decodes its chunk from the chunk's
// first bit as if a block started there; JPEG Huffman streams re-synchronise within a few symbols, so the lane's
// END state (bit position, coefficient index, block-in-MCU) is almost always right although its beginning is
// garbage.  Pass j >= 1: every lane restarts from its predecessor's end state of pass j-1; when no start state
// changes any more the decode equals the sequential one (lane 0 is right by construction, each end state is a
// function of the start state).  A last pass writes the luma coefficients: the block index of a lane's first block
// is the prefix sum of the lanes' block counts; DC differences are written and integrated afterwards.
// Files with restart intervals, files that do not converge within MAX_PASSES (every pass fixes at least one more lane), truncated or invalid streams are
// left to the sequential kernel (2b), which is the exact fall-back: M->par_done tells it what is finished.
// ---------------------------------------------------------------------------------------------------------
constexpr int PLB = 11;  // look-up bits of the parallel decoder
constexpr int PNT = 256;
static_assert(PNT == 256, "Meta::dc_cnt / dc_sum hold one entry per lane");
constexpr int MAX_PASSES = 32;  // a pass costs ~0.25 ms, the sequential fall-back ~40 ms per 70 KB file
constexpr unsigned MIN_CHUNK = 2048;

constexpr int PTAB = 4;  // rows of the parallel decoder's tables: the tables one scan refers to (more: sequential kernel)
using HuffPar = HuffLds<PLB, PTAB>;

struct ParCtx {
    const unsigned* words;    // un-stuffed stream as dwords (global memory, raw byte order)
    unsigned nwords;
    int nbm, nb0;             // blocks per MCU, luma blocks per MCU (they come first)
    int td[4], ta[4];         // table ROW (of HuffPar) per component
    unsigned long long blk;   // 4 bits per block of the MCU: bits 0-1 = DC table row, bits 2-3 = AC table row
    unsigned eobpack;         // byte r = length of the EOB code fused into DC row r's entries (0: none)
    unsigned total_y;
};

// Stream access without memory latency on the per-symbol dependency chain: the wave walks its lanes' chunks in lockstep,
// one 64-bit unit (dwords q, q+1 of the lane's own chunk) per outer step.  c0 c1 c2 hold dwords q .. q+2 (big-endian bit
// order; c2 serves windows that start in q+1), n0 n1 the raw dwords q+3, q+4, loaded one outer step before they are needed.
// Inside a step every lane decodes symbols until its bit position leaves the unit; the step ends when the slowest lane has.
__device__ __forceinline__ unsigned stream_dword(const ParCtx& cx, unsigned i) { return cx.words[min(i, cx.nwords - 1u)]; }  // tail: zero padding

// Look-up entries of the parallel decoder, re-packed from build_tables' (len << 8 | symbol) into what one decode step needs:
//   bits 0-4  bits the step consumes (code + value bits [+ a fused EOB code])      1 .. 31
//   bits 5-8  number of value bits (DC: min(symbol, 15), > 11 is invalid)
//   bits 9-15 what the step adds to the coefficient index k: run + 1, or 64 = the block ends here (EOB)
// DC rows: when the component's EOB code follows the difference bits inside the look-up window the entry swallows it
// (consumes both, ends the block).  A flat block (DC difference + EOB: most blocks of a dark camera frame, and the lanes that
// walk them set the pace of their wave) then costs one step instead of two.  A DC table shared by components with different
// AC tables is not fused.
__device__ __forceinline__ unsigned pack_entry(unsigned adv, unsigned size, unsigned kd) { return adv | (size << 5) | (kd << 9); }

__device__ void pack_entries(HuffPar& T, ParCtx& cx, int ncomp, int rows, unsigned dcrows, int tid, unsigned* s_min) {
    unsigned eobl[PTAB], eobc[PTAB];
    for (int a = 0; a < PTAB; ++a) {  // the EOB code of every AC row (symbol 0x00), while the rows are still unpacked
        eobl[a] = eobc[a] = 0;
        if (a >= rows || ((dcrows >> a) & 1u)) continue;
        if (tid == 0) *s_min = 0xffffffffu;
        __syncthreads();
        for (int j = tid; j < (1 << PLB); j += PNT) {
            const unsigned e = T.lut[a][j];
            if (e != 0 && (e & 255u) == 0) atomicMin(s_min, (unsigned)j);
        }
        __syncthreads();
        const unsigned jmin = *s_min;
        if (jmin != 0xffffffffu) {
            eobl[a] = T.lut[a][jmin] >> 8;
            eobc[a] = jmin >> (PLB - eobl[a]);
        }
        __syncthreads();
    }
    cx.eobpack = 0;
    for (int r = 0; r < PTAB; ++r) {
        if (r >= rows) continue;
        const bool dc = (dcrows >> r) & 1u;
        unsigned el = 0, ec = 0;
        if (dc) {
            int a = -1;
            bool same = true;
            for (int c = 0; c < ncomp; ++c) {
                if (cx.td[c] != r) continue;
                if (a >= 0 && cx.ta[c] != a) same = false;
                a = cx.ta[c];
            }
            for (int i = 0; i < PTAB; ++i) {
                if (same && i == a) {
                    el = eobl[i];
                    ec = eobc[i];
                }
            }
            cx.eobpack |= el << (8 * r);
        }
        for (int i = tid; i < (1 << PLB); i += PNT) {
            const unsigned e = T.lut[r][i];
            if (e == 0) continue;
            const unsigned len = e >> 8, sym = e & 255u;
            unsigned adv, size, kd;
            if (dc) {
                size = min(sym, 15u);
                adv = len + size;
                kd = 1;
                if (el && sym <= 11u && adv + el <= (unsigned)PLB && (((unsigned)i >> (PLB - adv - el)) & ((1u << el) - 1u)) == ec) {
                    adv += el;
                    kd = 64;
                }
            } else {
                size = sym & 15u;
                adv = len + size;
                kd = sym == 0 ? 64u : (sym >> 4) + 1u;
            }
            T.lut[r][i] = (unsigned short)pack_entry(adv, size, kd);
        }
    }
    __syncthreads();
}

// One chunk: decode from the state (p, k, b) up to bit `end`.  `run` = false: the lane keeps its state untouched.  Lanes that
// have left the current unit take part in the steps of the others with a zero entry (no advance), so the step itself is
// branch-free but for the long codes.
template <bool WRITE>
__device__ void decode_chunk(const ParCtx& cx, const HuffPar& T, bool run, unsigned& p, int& k, int& b, unsigned end,
                             unsigned& ycount, int& err, unsigned ybase, short* __restrict__ cbase, int& dcsum, unsigned& ndc) {
    if (!run) end = 0;
    if (run) {
        ycount = 0;
        err = 0;
    }
    unsigned kk = (unsigned)k, b4 = (unsigned)b * 4u;
    const unsigned nbm4 = (unsigned)cx.nbm * 4u, luma4 = (unsigned)cx.nb0 * 4u;
    const unsigned long long blk = cx.blk;
    unsigned q = p >> 5;
    unsigned c0 = __builtin_bswap32(stream_dword(cx, q)), c1 = __builtin_bswap32(stream_dword(cx, q + 1)), c2 = __builtin_bswap32(stream_dword(cx, q + 2));
    unsigned n0 = stream_dword(cx, q + 3), n1 = stream_dword(cx, q + 4);
    const unsigned short* lut = &T.lut[0][0];
    while (__any(p < end)) {
        const unsigned lim = min(end, (q + 2u) << 5);
        while (true) {
            const bool active = p < lim;
            if (!__any(active)) break;
            const bool first = (p >> 5) == q;
            const unsigned long long pair = ((unsigned long long)(first ? c0 : c1) << 32) | (first ? c1 : c2);
            const unsigned win = (unsigned)((pair << (p & 31u)) >> 32);  // the 32 bits starting at bit p
            const bool isdc = kk == 0;
            const unsigned t = ((unsigned)(blk >> b4) >> (isdc ? 0u : 2u)) & 3u;
            unsigned e = lut[(t << PLB) + (win >> (32 - PLB))];
            bool invalid = false;
            if (e == 0 && active) {  // longer than PLB bits: canonical search over the 5 remaining lengths, loads issued together
                const unsigned p16 = win >> 16;
                const unsigned l12 = T.limit[t][12], l13 = T.limit[t][13], l14 = T.limit[t][14], l15 = T.limit[t][15], l16 = T.limit[t][16];
                const int l = 12 + (p16 >= l12) + (p16 >= l13) + (p16 >= l14) + (p16 >= l15) + (p16 >= l16);
                if (l <= 16) {
                    const unsigned v = T.vals[t][(T.valoff[t][l] + (int)(p16 >> (16 - l))) & 255];
                    const unsigned size = isdc ? min(v, 15u) : (v & 15u);
                    e = pack_entry((unsigned)l + size, size, isdc ? 1u : (v == 0 ? 64u : (v >> 4) + 1u));
                } else {
                    e = pack_entry(16u, 0u, isdc ? 1u : 64u);  // invalid code: keep moving
                    invalid = true;
                }
            }
            if (!active) e = 0;
            const unsigned adv = e & 31u;
            const unsigned knext = kk + (e >> 9);
            if (WRITE) {  // (an inactive lane carries e = 0: no value bits, and `inside` below is false)
                const unsigned size = (e >> 5) & 15u;
                const bool fused = isdc && (e >> 9) == 64u;
                const unsigned vend = adv - (fused ? (cx.eobpack >> (8u * t)) & 255u : 0u);  // offset just behind the value bits
                const unsigned bits = __builtin_amdgcn_ubfe(win, 32u - vend, size);
                // JPEG's EXTEND without a branch: values below 2^(size-1) stand for bits - (2^size - 1)
                const unsigned half = (1u << size) >> 1;
                const int val = (int)bits - (((int)(bits - half) >> 31) & (int)((1u << size) - 1u));
                const unsigned j = ybase + ycount;            // luma blocks finished before this symbol
                const bool inside = active && j < cx.total_y;  // (behind the image: marker bytes and padding)
                const bool luma = b4 < luma4;
                const unsigned ci = isdc ? 0u : knext - 1u;  // the coefficient this symbol sets (DC, or size > 0)
                if (inside && (invalid || (isdc ? size > 11u : (size > 0u && ci > 63u)))) err = 1;
                // a luma DC leaves as the lane's running sum of differences (the IDCT adds the sum of the lanes before it);
                // ndc counts them
                const bool ydc = inside && isdc && luma;
                dcsum += ydc ? val : 0;
                ndc += ydc ? 1u : 0u;
                if (inside && luma && (isdc || size > 0u) && ci <= 63u) cbase[(long long)j * 64 + ci] = (short)(isdc ? dcsum : val);
            }
            const bool done = knext >= 64u;
            ycount += (done && b4 < luma4) ? 1u : 0u;
            const unsigned bn = b4 + 4u == nbm4 ? 0u : b4 + 4u;
            b4 = done ? bn : b4;
            kk = done ? 0u : knext;
            p += adv;  // < 32 bits: the position ends inside dword q + 2 at most
        }
        q += 2;
        c0 = c2;
        c1 = __builtin_bswap32(n0);
        c2 = __builtin_bswap32(n1);
        n0 = stream_dword(cx, q + 3);
        n1 = stream_dword(cx, q + 4);
    }
    k = (int)kk;
    b = (int)(b4 >> 2);
}

__global__ __launch_bounds__(PNT) void jpeg_huffman_par_kernel(Meta* __restrict__ metas, const unsigned* __restrict__ offsets,
                                                               const unsigned char* __restrict__ clean, short* __restrict__ coef,
                                                               long long coef_stride) {
    __shared__ HuffPar T;
    __shared__ unsigned s_p[PNT + 1];
    __shared__ unsigned s_kb[PNT + 1];
    __shared__ unsigned s_cnt[PNT];
    __shared__ int s_dc[PNT];
    __shared__ int s_flag;
    __shared__ unsigned s_ep[PNT], s_ekb[PNT];  // end state per chunk
    __shared__ unsigned short s_list[PNT];
    __shared__ unsigned s_wcnt[PNT / 64];
    const int img = blockIdx.x, tid = threadIdx.x;
    Meta* M = metas + img;
    if (tid == 0) M->par_done = 0;
    if (rfl((unsigned)M->status) != ST_OK) return;
    if (rfl((unsigned)M->restart_interval) != 0) return;  // restart markers sit inside the stream: sequential kernel
    ParCtx cx;
    const int ncomp = (int)rfl((unsigned)M->ncomp);
    int nb[4] = {0, 0, 0, 0};
    SlotMap map;
    for (int i = 0; i < 8; ++i) map.row[i] = -1;
    int rows = 0;
    unsigned dcrows = 0;  // bit r: row r holds a DC table
    for (int c = 0; c < 4; ++c) {
        if (c < ncomp) nb[c] = (int)rfl((unsigned)M->comp_h[c]) * (int)rfl((unsigned)M->comp_v[c]);
        const int sd = (int)rfl((unsigned)M->comp_td[c]) & 3, sa = 4 + ((int)rfl((unsigned)M->comp_ta[c]) & 3);
        if (c < ncomp) {
            for (int i = 0; i < 8; ++i) {  // (static indexing keeps the map in scalar registers)
                if ((i == sd || i == sa) && map.row[i] < 0) {
                    if (i < 4 && rows < 32) dcrows |= 1u << rows;
                    map.row[i] = rows++;
                }
            }
        }
        cx.td[c] = cx.ta[c] = 0;
        for (int i = 0; i < 8; ++i) {
            if (i == sd) cx.td[c] = map.row[i];
            if (i == sa) cx.ta[c] = map.row[i];
        }
    }
    if (rows > PTAB) return;                                     // more tables than rows: sequential kernel
    if (build_tables<PLB, PNT, PTAB>(M, T, map, tid)) return;   // the sequential kernel reports the error

    const unsigned clean_len = rfl(M->clean_len);
    cx.words = reinterpret_cast<const unsigned*>(clean + rfl(offsets[img]));
    cx.nwords = (clean_len + 3) / 4 + 8;
    cx.nbm = nb[0] + nb[1] + nb[2] + nb[3];
    if (cx.nbm > 16) return;  // beyond the standard's 10 blocks per MCU: sequential kernel
    cx.nb0 = nb[0];
    pack_entries(T, cx, ncomp, rows, dcrows, tid, &s_wcnt[0]);
    cx.blk = 0;
    for (int i = 0; i < 16; ++i) {
        const int c = (i >= nb[0]) + (i >= nb[0] + nb[1]) + (i >= nb[0] + nb[1] + nb[2]);
        const int tdc = c == 0 ? cx.td[0] : (c == 1 ? cx.td[1] : (c == 2 ? cx.td[2] : cx.td[3]));
        const int tac = c == 0 ? cx.ta[0] : (c == 1 ? cx.ta[1] : (c == 2 ? cx.ta[2] : cx.ta[3]));
        cx.blk |= (unsigned long long)((tdc & 3) | ((tac & 3) << 2)) << (4 * i);
    }
    const unsigned mcus = rfl((unsigned)M->mcus_x) * rfl((unsigned)M->mcus_y);
    cx.total_y = mcus * (unsigned)nb[0];
    const unsigned total_bits = clean_len * 8u;
    // chunks of at least MIN_CHUNK bits: 4:2:0 streams need ~1 000 bits to re-synchronise (the block-in-MCU index
    // locks in last), shorter chunks would only add passes; small files simply use fewer lanes
    unsigned chunk = ((total_bits + PNT - 1) / PNT + 31) & ~31u;
    if (chunk < MIN_CHUNK) chunk = MIN_CHUNK;
    const unsigned nominal = (unsigned)tid * chunk;
    const unsigned end = min(nominal + chunk, total_bits);
    short* cbase = coef + (long long)img * coef_stride;

    // start states of pass 0: speculative (a block starts at the chunk's first bit; chunk 0 is exact).  Pass 1 then runs every
    // chunk from its predecessor's speculative end -- a state that had a whole chunk to synchronise -- and from pass 2 on only
    // the chunks whose predecessor still moved are decoded again (a run-in chunk inside pass 0 would reach the same states one
    // pass earlier, but by decoding every chunk three times instead of twice)
    s_p[tid] = min(nominal, total_bits);
    s_kb[tid] = 0;
    __syncthreads();

    unsigned p = 0, ycount = 0;
    int k = 0, b = 0, err = 0;
    unsigned used_p = 0, used_kb = 0;  // the start state chunk tid was last decoded from
    bool converged = false;
    int passes = 0;
    const int wave = tid >> 6, lane = tid & 63;
    for (int pass = 0; pass < MAX_PASSES; ++pass) {
        passes = pass + 1;
        // a chunk whose start state is the one it was decoded from in the previous pass would only repeat itself: the
        // chunks that do need decoding are compacted (in order) onto the first lanes, so that late passes, which repair a
        // handful of chunks, occupy one wave instead of four
        const unsigned sp = s_p[tid], skb = s_kb[tid];
        const bool redo = pass == 0 || sp != used_p || skb != used_kb;
        used_p = sp;
        used_kb = skb;
        const unsigned long long m = __ballot(redo);
        if (lane == 0) s_wcnt[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned before = 0, n = 0;
        for (int w = 0; w < PNT / 64; ++w) {
            const unsigned c = s_wcnt[w];
            before += w < wave ? c : 0u;
            n += c;
        }
        if (redo) s_list[before + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tid;
        __syncthreads();
        const bool work = (unsigned)tid < n;
        const unsigned c = work ? s_list[tid] : 0u;
        const unsigned ckb = s_kb[c];
        p = s_p[c];
        k = (int)(ckb & 255u);
        b = (int)(ckb >> 8);
        int dc_unused = 0;
        unsigned ndc_unused = 0;
        decode_chunk<false>(cx, T, work, p, k, b, min(c * chunk + chunk, total_bits), ycount, err, 0u, cbase, dc_unused, ndc_unused);
        if (work) {
            s_ep[c] = p;
            s_ekb[c] = (unsigned)k | ((unsigned)b << 8);
            s_cnt[c] = ycount;
        }
        __syncthreads();
        // chunk tid's end state is chunk tid+1's next start state
        int changed = 0;
        if (tid + 1 < PNT) {
            const unsigned np = s_ep[tid], nkb = s_ekb[tid];
            // chunks that lie behind the stream hold nothing: their start state is irrelevant (and would otherwise
            // ripple through the idle chunks one per pass)
            changed = (nominal + chunk < total_bits) && ((s_p[tid + 1] != np) || (s_kb[tid + 1] != nkb));
            s_p[tid + 1] = np;
            s_kb[tid + 1] = nkb;
        }
        if (!__syncthreads_or(changed)) {
            converged = pass > 0 || PNT == 1;
            if (pass > 0) break;
        }
    }
    if (!converged) {
        if (tid == 0) M->par_done = -1;
        return;
    }
    // the counts of the last pass belong to the converged start states: exclusive scan -> first block per lane
    for (int o = 1; o < PNT; o <<= 1) {
        const unsigned a = tid >= o ? s_cnt[tid - o] : 0;
        __syncthreads();
        s_cnt[tid] += a;
        __syncthreads();
    }
    const unsigned all_y = s_cnt[PNT - 1];
    if (all_y < cx.total_y) {
        if (tid == 0) M->par_done = -2;
        return;
    }  // the stream ends early: sequential kernel (it feeds zeros like libjpeg's callers expect)
    ycount = s_cnt[tid] - (tid ? s_cnt[tid - 1] : 0u);
    const unsigned ybase = s_cnt[tid] - ycount;
    p = s_p[tid];
    k = (int)(s_kb[tid] & 255u);
    b = (int)(s_kb[tid] >> 8);
    int dcsum = 0;
    unsigned ndc = 0;
    decode_chunk<true>(cx, T, true, p, k, b, end, ycount, err, ybase, cbase, dcsum, ndc);
    if (tid == 0) s_flag = 0;
    __threadfence_block();
    __syncthreads();
    if (err) s_flag = 1;
    __syncthreads();
    if (s_flag) {
        if (tid == 0) M->par_done = -3;
        return;
    }  // invalid symbols: let the sequential kernel classify the file
    // integrating the DC differences over the luma blocks in decode order: every lane wrote running sums that start at zero;
    // what is missing is the sum of the lanes before it.  Lane l decoded the DCs of blocks [cnt(l-1), cnt(l)) (inclusive
    // scans of the DC counts), so the IDCT kernel, which reads every block anyway, finds a block's lane by bisection and adds
    // that lane's offset (a separate pass over the DC terms costs as much memory time as the whole IDCT).
    s_cnt[tid] = ndc;
    s_dc[tid] = dcsum;
    __syncthreads();
    for (int o = 1; o < PNT; o <<= 1) {
        const unsigned a = tid >= o ? s_cnt[tid - o] : 0u;
        const int d = tid >= o ? s_dc[tid - o] : 0;
        __syncthreads();
        s_cnt[tid] += a;
        s_dc[tid] += d;
        __syncthreads();
    }
    M->dc_cnt[tid] = s_cnt[tid];
    M->dc_sum[tid] = s_dc[tid];
    if (tid == 0) M->par_done = passes;  // > 0: done here (the value is the number of synchronisation passes)
}

// ---------------------------------------------------------------------------------------------------------
// kernel 2b: sequential Huffman decode, one wave per file (exact fall-back; wave-uniform control flow: the bit
// buffer lives in scalar registers, the next 256 stream bytes in one VGPR read with v_readlane; lane i keeps
// coefficient i of the current block, so a decoded value is a compare + select and a finished luma block leaves as
// one coalesced 128-byte store)
// ---------------------------------------------------------------------------------------------------------
constexpr int LUT_BITS = 9;

struct BitReader {
    const unsigned* words;  // un-stuffed stream as aligned dwords
    unsigned nwords;        // number of defined dwords (including padding)
    unsigned long long buf; // left-justified
    int cnt;
    int widx;          // next dword inside `win`
    unsigned wbase;    // dword index of win lane 0
    unsigned win, nxt; // per lane: dword wbase + lane, wbase + 64 + lane
    int lane;

    __device__ __forceinline__ unsigned load(unsigned idx) const { return idx < nwords ? words[idx] : 0u; }
    __device__ void init(const unsigned char* stream, unsigned len_bytes, int ln) {
        words = reinterpret_cast<const unsigned*>(stream);
        nwords = (len_bytes + 3) / 4 + 8;
        lane = ln;
        wbase = 0;
        win = load(lane);
        nxt = load(64 + lane);
        widx = 0;
        buf = 0;
        cnt = 0;
    }
    __device__ __forceinline__ void refill() {  // after this cnt >= 33
        if (cnt <= 32) {
            unsigned w = (unsigned)__builtin_amdgcn_readlane((int)win, widx);
            w = __builtin_bswap32(w);
            buf |= (unsigned long long)w << (32 - cnt);
            cnt += 32;
            if (++widx == 64) {
                widx = 0;
                wbase += 64;
                win = nxt;
                nxt = load(wbase + 64 + lane);
            }
        }
    }
    __device__ __forceinline__ unsigned peek(int n) const { return (unsigned)(buf >> (64 - n)); }
    __device__ __forceinline__ void skip(int n) {
        buf <<= n;
        cnt -= n;
    }
    __device__ __forceinline__ void align_byte() { skip(cnt & 7); }
};

__device__ __forceinline__ int decode_symbol(BitReader& br, const HuffLds<LUT_BITS>& T, int t) {
    br.refill();
    const unsigned e = rfl(T.lut[t][br.peek(LUT_BITS)]);
    if (e) {
        br.skip((int)(e >> 8));
        return (int)(e & 255u);
    }
    const unsigned p16 = br.peek(16);
    int l = LUT_BITS + 1;
    while (l <= 16 && p16 >= rfl(T.limit[t][l])) ++l;
    if (l > 16) return -1;
    const int code = (int)(p16 >> (16 - l));
    br.skip(l);
    return (int)rfl(T.vals[t][(rfl((unsigned)T.valoff[t][l]) + code) & 255]);
}

__device__ __forceinline__ int receive_extend(BitReader& br, int s) {  // s in 1..16, cnt >= s guaranteed by caller
    const int v = (int)br.peek(s);
    br.skip(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

__global__ __launch_bounds__(64) void jpeg_huffman_kernel(Meta* __restrict__ metas, const unsigned* __restrict__ offsets,
                                                          const unsigned char* __restrict__ clean, short* __restrict__ coef,
                                                          long long coef_stride) {
    __shared__ HuffLds<LUT_BITS> T;
    const int img = blockIdx.x, lane = threadIdx.x;
    Meta* M = metas + img;
    if (rfl((unsigned)M->status) != ST_OK) return;
    if ((int)rfl((unsigned)M->par_done) > 0) return;  // the parallel kernel finished this file
    const int ncomp = (int)rfl((unsigned)M->ncomp);
    SlotMap ident;
    for (int i = 0; i < 8; ++i) ident.row[i] = i;
    if (build_tables<LUT_BITS, 64, 8>(M, T, ident, lane)) {
        if (lane == 0) M->status = ST_CORRUPT;
        return;
    }

    // ---- scan ----------------------------------------------------------------------------------------
    BitReader br;
    br.init(clean + rfl(offsets[img]), rfl(M->clean_len), lane);
    const int mcus_x = (int)rfl((unsigned)M->mcus_x), mcus_y = (int)rfl((unsigned)M->mcus_y);
    const int ri = (int)rfl((unsigned)M->restart_interval);
    int ch[4], cv[4], td[4], ta[4], pred[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; ++c) {
        ch[c] = (int)rfl((unsigned)M->comp_h[c]);
        cv[c] = (int)rfl((unsigned)M->comp_v[c]);
        td[c] = (int)rfl((unsigned)M->comp_td[c]) & 3;
        ta[c] = 4 + ((int)rfl((unsigned)M->comp_ta[c]) & 3);
    }
    // lane i keeps coefficient i of the current block in zig-zag order (the order of the coefficient buffer)
    const int kz = lane;
    short* cbase = coef + (long long)img * coef_stride;
    int restart_left = ri, next_rst = 0, status = ST_OK;
    long long yblock = 0;  // luma blocks in decode order

    for (int my = 0; my < mcus_y && status == ST_OK; ++my) {
        for (int mx = 0; mx < mcus_x && status == ST_OK; ++mx) {
            if (ri && restart_left == 0) {
                br.align_byte();
                br.refill();
                int guard = 0;
                while (br.peek(8) == 0xFF && guard++ < 64) {  // marker prefix and fill bytes
                    br.skip(8);
                    br.refill();
                }
                if ((int)br.peek(8) != 0xD0 + next_rst) {
                    status = ST_CORRUPT;
                    break;
                }
                br.skip(8);
                next_rst = (next_rst + 1) & 7;
                restart_left = ri;
                pred[0] = pred[1] = pred[2] = pred[3] = 0;
            }
            for (int c = 0; c < ncomp && status == ST_OK; ++c) {
                const int nblk = ch[c] * cv[c];
                for (int blk = 0; blk < nblk && status == ST_OK; ++blk) {
                    int mine = 0;
                    int s = decode_symbol(br, T, td[c]);
                    if (s < 0 || s > 11) {
                        status = ST_CORRUPT;
                        break;
                    }
                    if (s) {
                        br.refill();
                        pred[c] += receive_extend(br, s);
                    }
                    if (kz == 0) mine = pred[c];
                    int k = 1;
                    while (k < 64) {
                        s = decode_symbol(br, T, ta[c]);
                        if (s < 0) {
                            status = ST_CORRUPT;
                            break;
                        }
                        const int r = s >> 4, sz = s & 15;
                        if (sz == 0) {
                            if (r != 15) break;
                            k += 16;
                            continue;
                        }
                        k += r;
                        if (k > 63) {
                            status = ST_CORRUPT;
                            break;
                        }
                        br.refill();
                        const int val = receive_extend(br, sz);
                        if (kz == k) mine = val;
                        ++k;
                    }
                    if (c == 0) {
                        cbase[yblock * 64 + lane] = (short)mine;
                        ++yblock;
                    }
                }
            }
            if (ri) --restart_left;
        }
    }
    if (status != ST_OK && lane == 0) M->status = status;
}

// ---------------------------------------------------------------------------------------------------------
// kernel 3: de-quantise + islow inverse DCT, one thread per luma block
// ---------------------------------------------------------------------------------------------------------
template <int SHIFT>
__device__ __forceinline__ void idct_1d(const int (&in)[8], int (&out)[8]) {
    const int rnd = 1 << (SHIFT - 1);
    int z1, z2, z3, z4, z5, t0, t1, t2, t3;
    z2 = in[2];
    z3 = in[6];
    z1 = (z2 + z3) * 4433;
    t2 = z1 + z3 * (-15137);
    t3 = z1 + z2 * 6270;
    t0 = (in[0] + in[4]) * 8192;
    t1 = (in[0] - in[4]) * 8192;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = in[7];
    t1 = in[5];
    t2 = in[3];
    t3 = in[1];
    z1 = t0 + t3;
    z2 = t1 + t2;
    z3 = t0 + t2;
    z4 = t1 + t3;
    z5 = (z3 + z4) * 9633;
    t0 *= 2446;
    t1 *= 16819;
    t2 *= 25172;
    t3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 *= -16069;
    z4 *= -3196;
    z3 += z5;
    z4 += z5;
    t0 += z1 + z3;
    t1 += z2 + z4;
    t2 += z2 + z3;
    t3 += z1 + z4;
    out[0] = (t10 + t3 + rnd) >> SHIFT;
    out[7] = (t10 - t3 + rnd) >> SHIFT;
    out[1] = (t11 + t2 + rnd) >> SHIFT;
    out[6] = (t11 - t2 + rnd) >> SHIFT;
    out[2] = (t12 + t1 + rnd) >> SHIFT;
    out[5] = (t12 - t1 + rnd) >> SHIFT;
    out[3] = (t13 + t0 + rnd) >> SHIFT;
    out[4] = (t13 - t0 + rnd) >> SHIFT;
}

// position of natural-order coefficient n in the zig-zag sequence
__device__ __forceinline__ constexpr int zigzag_index(int n) {
    constexpr unsigned char inv[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                       41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                       46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
    return inv[n];
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const Meta* __restrict__ metas, const short* __restrict__ coef,
                                                        long long coef_stride, int n, int width, int height,
                                                        unsigned char* __restrict__ luma, int* __restrict__ status_out,
                                                        int* __restrict__ path_out) {
    const int img = blockIdx.y;
    const Meta* M = metas + img;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the file's verdict leaves with the last kernel of the chain
        status_out[img] = M->status;
        if (path_out) path_out[img] = M->par_done;
    }
    if (M->status != ST_OK) return;
    const int ybw = M->ybw, ybh = M->ybh;
    const int blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= ybw * ybh) return;
    const int bx = blk % ybw, by = blk / ybw;
    if (bx * 8 >= width || by * 8 >= height) return;
    const int h0 = M->comp_h[0], v0 = M->comp_v[0];
    const int j = ((by / v0) * M->mcus_x + bx / h0) * (h0 * v0) + (by % v0) * h0 + bx % h0;  // decode order: MCU by MCU
    const short* c = coef + (long long)img * coef_stride + (long long)j * 64;
    const unsigned short* q = M->q[M->comp_tq[0]];
    int dc_offset = 0;
    if (M->par_done > 0) {
        int lo = 0, hi = 255;  // smallest l with dc_cnt[l] > j
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int mid = (lo + hi) >> 1;
            const bool right = M->dc_cnt[mid] <= (unsigned)j;
            lo = right ? mid + 1 : lo;
            hi = right ? hi : mid;
        }
        dc_offset = lo ? M->dc_sum[lo - 1] : 0;
    }
    // the coefficient buffer is in zig-zag order (where the non-zero terms of a block cluster: the Huffman kernels' sparse
    // stores touch one or two 32-byte sectors per block instead of four); un-zig-zag = static register picks
    unsigned w[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 raw = *reinterpret_cast<const uint4*>(c + i * 8);
        w[4 * i] = raw.x;
        w[4 * i + 1] = raw.y;
        w[4 * i + 2] = raw.z;
        w[4 * i + 3] = raw.w;
    }
    int ws[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int z = zigzag_index(r * 8 + cc);
            const unsigned half = (z & 1) ? w[z >> 1] >> 16 : w[z >> 1] & 0xffffu;
            ws[r][cc] = (int)(short)half * (int)q[r * 8 + cc];
        }
    }
    ws[0][0] = (int)(short)((w[0] & 0xffffu) + (unsigned)dc_offset) * (int)q[0];
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = ws[r][col];
        idct_1d<11>(in, out);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][col] = out[r];
    }
    unsigned char* dst = luma + ((long long)img * height + by * 8) * width + bx * 8;
    const bool full = bx * 8 + 8 <= width && (width & 7) == 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int out[8];
        idct_1d<18>(ws[r], out);
        if (by * 8 + r >= height) break;
        unsigned px[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) px[x] = (unsigned)min(max(out[x] + 128, 0), 255);
        if (full) {
            uint2 pk;
            pk.x = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
            pk.y = px[4] | (px[5] << 8) | (px[6] << 16) | (px[7] << 24);
            *reinterpret_cast<uint2*>(dst + (long long)r * width) = pk;
        } else {
            for (int x = 0; x < 8; ++x)
                if (bx * 8 + x < width) dst[(long long)r * width + x] = (unsigned char)px[x];
        }
    }
}

inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }
inline long long coef_stride_for(int width, int height) {
    // worst case sampling 4x4: MCU 32x32 pixels
    const long long bw = (width + 31) / 32 * 4, bh = (height + 31) / 32 * 4;
    return bw * bh * 64;
}

}  // namespace jpg

extern "C" {

size_t df3d_jpeg_work_bytes(int n, int width, int height, size_t total_file_bytes) {
    if (n < 0 || width <= 0 || height <= 0) return 0;
    long long b = jpg::align_up((long long)n * sizeof(jpg::Meta), 256);
    b += jpg::align_up((long long)total_file_bytes + 4096, 256);
    b += (long long)n * jpg::coef_stride_for(width, height) * 2;
    return (size_t)b;
}

int df3d_jpeg_decode_luma(const unsigned char* files_dev, const unsigned* offsets_dev, const unsigned* sizes_dev, int n,
                          size_t total_file_bytes, unsigned max_file_bytes, int width, int height, unsigned char* luma_dev, int* status_dev, int* path_dev,
                          void* work_dev, size_t work_bytes, int flags, void* stream) {
    DF3D_CHECK_ARG(n >= 0 && width > 0 && height > 0, "bad shape");
    if (n == 0) return DF3D_OK;
    DF3D_CHECK_ARG(files_dev && offsets_dev && sizes_dev && luma_dev && status_dev && work_dev, "null pointer");
    DF3D_CHECK_ARG(((uintptr_t)files_dev & 15) == 0, "files_dev must be 16-byte aligned (and every file offset a multiple of 16)");
    DF3D_CHECK_ARG(work_bytes >= df3d_jpeg_work_bytes(n, width, height, total_file_bytes), "work buffer too small (df3d_jpeg_work_bytes)");
    (void)max_file_bytes;  // (see the header: a hint no decoder needs any more)
    hipStream_t s = df3d::as_stream(stream);
    char* w = static_cast<char*>(work_dev);
    jpg::Meta* metas = reinterpret_cast<jpg::Meta*>(w);
    w += jpg::align_up((long long)n * sizeof(jpg::Meta), 256);
    unsigned char* clean = reinterpret_cast<unsigned char*>(w);
    w += jpg::align_up((long long)total_file_bytes + 4096, 256);
    short* coef = reinterpret_cast<short*>(w);
    const long long cs = jpg::coef_stride_for(width, height);
    hipLaunchKernelGGL(jpg::jpeg_parse_kernel, dim3(n), dim3(jpg::PT), 0, s, files_dev, offsets_dev, sizes_dev, width, height, metas, clean);
    if (!(flags & 1)) {
        // the parallel decoder writes only non-zero coefficients
        DF3D_HIP(hipMemsetAsync(coef, 0, (size_t)n * cs * sizeof(short), s));
        hipLaunchKernelGGL(jpg::jpeg_huffman_par_kernel, dim3(n), dim3(jpg::PNT), 0, s, metas, offsets_dev, clean, coef, cs);
    }
    hipLaunchKernelGGL(jpg::jpeg_huffman_kernel, dim3(n), dim3(64), 0, s, metas, offsets_dev, clean, coef, cs);
    const int blocks = (int)(cs / 64);
    hipLaunchKernelGGL(jpg::jpeg_idct_kernel, dim3((blocks + 255) / 256, n), dim3(256), 0, s, metas, coef, cs, n, width, height, luma_dev, status_dev,
                       path_dev);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
}

}  // extern "C"
