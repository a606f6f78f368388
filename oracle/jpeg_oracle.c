/* ORACLE (test infrastructure only -- never linked into or called by the product path).
 *
 * CPU restatement of the JPEG decode the reference's input front-end performs through libjpeg (inside
 * cv2.imread / Pillow under df2d's DataLoader; call site reference df3d/core.py:177-185, frames written by
 * ffmpeg in df3d/core.py:446-459): baseline sequential DCT, Huffman coding, 8-bit samples (ITU-T T.81 Annex F),
 * LUMA PLANE ONLY, with libjpeg's default "islow" inverse DCT (Loeffler-Ligtenberg-Moshovitz, 13-bit fixed
 * point: CONST_BITS 13, PASS1_BITS 2, descale-with-rounding, clamp of sample+128 to [0, 255]).
 *
 * The algorithm lives in a third-party dependency that is not in /root/reference (libjpeg[-turbo], whatever
 * version cv2/Pillow bundle; this image: Pillow 12.2 + libjpeg-turbo, JPEG_LIB_VERSION 62).  It is pinned
 * against that very library: tests/test_oracle_golden.py compares this decoder with Pillow bit for bit on the
 * reference's own test JPEGs (tests/golden/images) and on Pillow-encoded grayscale / 4:2:0 / 4:4:4 / restart-
 * interval files.  For chroma-neutral files (the monochrome cameras of the rig, which is all the reference ever
 * sees) the luma plane IS the gray image; for coloured files it equals libjpeg's own grayscale output
 * (out_color_space = JCS_GRAYSCALE), not the RGB -> L round trip.
 *
 * Build: gcc -O2 -shared -fPIC (oracle/build_c.py).  Entry points: jpeg_oracle_info, jpeg_oracle_decode_luma.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum {
    JO_OK = 0,
    JO_TRUNCATED = 1,   /* ran out of bytes */
    JO_NOT_JPEG = 2,    /* no SOI */
    JO_UNSUPPORTED = 3, /* progressive / arithmetic / lossless / 12-bit / multi-scan / >4 components */
    JO_CORRUPT = 4,     /* bad table, bad Huffman code, bad marker */
    JO_SHAPE = 5        /* caller's expected width/height differ */
};

typedef struct {
    int present;
    int maxcode[18]; /* largest code of length l, -1 if none */
    int valptr[17];
    int mincode[17];
    uint8_t huffval[256];
} huff_t;

typedef struct {
    const uint8_t* p;
    size_t n, pos;
    uint32_t buf;
    int cnt;
    int hit_marker;
} bits_t;

typedef struct {
    int id, h, v, tq, td, ta, pred;
} comp_t;

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int build_huff(huff_t* h, const uint8_t* counts, const uint8_t* vals, int nvals) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        if (counts[l - 1]) {
            k += counts[l - 1];
            code += counts[l - 1];
            h->maxcode[l] = code - 1;
            if (code > (1 << l)) return JO_CORRUPT;
        } else {
            h->maxcode[l] = -1;
        }
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
    if (k != nvals || k > 256) return JO_CORRUPT;
    memcpy(h->huffval, vals, (size_t)nvals);
    h->present = 1;
    return JO_OK;
}

/* next bit of the entropy-coded segment; FF00 -> FF, a marker ends the data (zeros are fed after it) */
static int get_bit(bits_t* b) {
    if (b->cnt == 0) {
        uint32_t c = 0;
        if (!b->hit_marker && b->pos < b->n) {
            c = b->p[b->pos];
            if (c == 0xFF) {
                uint32_t d = b->pos + 1 < b->n ? b->p[b->pos + 1] : 0xD9;
                if (d == 0) {
                    b->pos += 2;
                } else {
                    b->hit_marker = 1;
                    c = 0;
                }
            } else {
                b->pos += 1;
            }
        }
        b->buf = c;
        b->cnt = 8;
    }
    b->cnt--;
    return (int)((b->buf >> b->cnt) & 1u);
}

static int get_bits(bits_t* b, int n) {
    int v = 0;
    while (n--) v = (v << 1) | get_bit(b);
    return v;
}

static int decode_sym(bits_t* b, const huff_t* h, int* sym) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | get_bit(b);
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) {
            *sym = h->huffval[h->valptr[l] + code - h->mincode[l]];
            return JO_OK;
        }
    }
    return JO_CORRUPT;
}

static int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

/* libjpeg "islow" 8x8 inverse DCT on dequantised coefficients (natural order) -> 64 samples */
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
static void idct_1d(const int32_t in[8], int32_t out[8], int shift) {
    int32_t z1, z2, z3, z4, z5, t0, t1, t2, t3, t10, t11, t12, t13;
    z2 = in[2];
    z3 = in[6];
    z1 = (z2 + z3) * 4433;
    t2 = z1 + z3 * (-15137);
    t3 = z1 + z2 * 6270;
    t0 = (in[0] + in[4]) * 8192;
    t1 = (in[0] - in[4]) * 8192;
    t10 = t0 + t3;
    t13 = t0 - t3;
    t11 = t1 + t2;
    t12 = t1 - t2;
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
    out[0] = DESCALE(t10 + t3, shift);
    out[7] = DESCALE(t10 - t3, shift);
    out[1] = DESCALE(t11 + t2, shift);
    out[6] = DESCALE(t11 - t2, shift);
    out[2] = DESCALE(t12 + t1, shift);
    out[5] = DESCALE(t12 - t1, shift);
    out[3] = DESCALE(t13 + t0, shift);
    out[4] = DESCALE(t13 - t0, shift);
}

static void idct_islow(const int32_t coef[64], uint8_t out[64]) {
    int32_t ws[64], col[8], res[8];
    for (int c = 0; c < 8; ++c) {
        for (int r = 0; r < 8; ++r) col[r] = coef[r * 8 + c];
        idct_1d(col, res, 13 - 2);
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = res[r];
    }
    for (int r = 0; r < 8; ++r) {
        idct_1d(&ws[r * 8], res, 13 + 2 + 3);
        for (int c = 0; c < 8; ++c) {
            int32_t v = res[c] + 128;
            out[r * 8 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

typedef struct {
    int width, height, ncomp, restart_interval, hmax, vmax;
    comp_t comp[4];
    uint16_t q[4][64]; /* natural order */
    int qpresent[4];
    huff_t dc[4], ac[4];
    size_t scan_pos;
} hdr_t;

static int parse_header(const uint8_t* d, size_t n, hdr_t* H) {
    memset(H, 0, sizeof(*H));
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return JO_NOT_JPEG;
    size_t i = 2;
    int have_sof = 0;
    for (;;) {
        if (i + 4 > n) return JO_TRUNCATED;
        if (d[i] != 0xFF) return JO_CORRUPT;
        while (i < n && d[i] == 0xFF) ++i; /* fill bytes */
        if (i >= n) return JO_TRUNCATED;
        int m = d[i++];
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) return JO_CORRUPT;
        if (i + 2 > n) return JO_TRUNCATED;
        size_t L = ((size_t)d[i] << 8) | d[i + 1];
        if (L < 2 || i + L > n) return JO_TRUNCATED;
        const uint8_t* s = d + i + 2;
        size_t sl = L - 2;
        if (m == 0xDB) {
            size_t j = 0;
            while (j < sl) {
                int pq = s[j] >> 4, tq = s[j] & 15;
                if (tq > 3 || pq > 1) return JO_CORRUPT;
                ++j;
                if (j + (pq ? 128 : 64) > sl) return JO_CORRUPT;
                for (int k = 0; k < 64; ++k) {
                    H->q[tq][ZIGZAG[k]] = pq ? (uint16_t)((s[j] << 8) | s[j + 1]) : s[j];
                    j += pq ? 2 : 1;
                }
                H->qpresent[tq] = 1;
            }
        } else if (m == 0xC4) {
            size_t j = 0;
            while (j < sl) {
                if (j + 17 > sl) return JO_CORRUPT;
                int tc = s[j] >> 4, th = s[j] & 15, nv = 0;
                if (tc > 1 || th > 3) return JO_CORRUPT;
                for (int k = 0; k < 16; ++k) nv += s[j + 1 + k];
                if (nv > 256 || j + 17 + (size_t)nv > sl) return JO_CORRUPT;
                int rc = build_huff(tc ? &H->ac[th] : &H->dc[th], s + j + 1, s + j + 17, nv);
                if (rc) return rc;
                j += 17 + (size_t)nv;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6) return JO_CORRUPT;
            if (s[0] != 8) return JO_UNSUPPORTED;
            H->height = (s[1] << 8) | s[2];
            H->width = (s[3] << 8) | s[4];
            H->ncomp = s[5];
            if (H->ncomp < 1 || H->ncomp > 4) return JO_UNSUPPORTED;
            if (sl < 6 + 3 * (size_t)H->ncomp || H->width == 0 || H->height == 0) return JO_CORRUPT;
            for (int c = 0; c < H->ncomp; ++c) {
                comp_t* C = &H->comp[c];
                C->id = s[6 + 3 * c];
                C->h = s[7 + 3 * c] >> 4;
                C->v = s[7 + 3 * c] & 15;
                C->tq = s[8 + 3 * c];
                if (C->h < 1 || C->h > 4 || C->v < 1 || C->v > 4 || C->tq > 3) return JO_CORRUPT;
                if (C->h > H->hmax) H->hmax = C->h;
                if (C->v > H->vmax) H->vmax = C->v;
            }
            have_sof = 1;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return JO_UNSUPPORTED; /* progressive, lossless, arithmetic */
        } else if (m == 0xDD) {
            if (sl < 2) return JO_CORRUPT;
            H->restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1) return JO_CORRUPT;
            int ns = s[0];
            if (ns != H->ncomp) return JO_UNSUPPORTED; /* multi-scan files */
            if (sl < 1 + 2 * (size_t)ns + 3) return JO_CORRUPT;
            for (int k = 0; k < ns; ++k) {
                int id = s[1 + 2 * k], found = 0;
                if (H->comp[k].id == id) {
                    H->comp[k].td = s[2 + 2 * k] >> 4;
                    H->comp[k].ta = s[2 + 2 * k] & 15;
                    found = 1;
                }
                if (!found) return JO_UNSUPPORTED; /* scan order differs from frame order */
                if (H->comp[k].td > 3 || H->comp[k].ta > 3) return JO_CORRUPT;
            }
            H->scan_pos = i + L;
            return JO_OK;
        }
        i += L;
    }
}

int jpeg_oracle_info(const uint8_t* data, size_t len, int* width, int* height, int* ncomp) {
    hdr_t H;
    int rc = parse_header(data, len, &H);
    if (rc) return rc;
    *width = H.width;
    *height = H.height;
    *ncomp = H.ncomp;
    return JO_OK;
}

/* out: height x width luma samples.  coef_out (optional): the luma component's quantised coefficients,
 * [block row][block col][64] natural order, over the MCU-padded block grid; stats (optional): [0] = number of
 * Huffman symbols decoded, [1] = luma blocks per row of coef_out, [2] = luma block rows. */
int jpeg_oracle_decode_luma(const uint8_t* data, size_t len, int expect_w, int expect_h, uint8_t* out, int16_t* coef_out,
                            long long* stats) {
    hdr_t H;
    int rc = parse_header(data, len, &H);
    if (rc) return rc;
    if ((expect_w > 0 && expect_w != H.width) || (expect_h > 0 && expect_h != H.height)) return JO_SHAPE;
    for (int c = 0; c < H.ncomp; ++c) {
        if (!H.qpresent[H.comp[c].tq] || !H.dc[H.comp[c].td].present || !H.ac[H.comp[c].ta].present) return JO_CORRUPT;
    }
    if (H.ncomp == 1) H.comp[0].h = H.comp[0].v = H.hmax = H.vmax = 1; /* single component: MCU = one block */
    const int mcu_w = 8 * H.hmax, mcu_h = 8 * H.vmax;
    const int mcus_x = (H.width + mcu_w - 1) / mcu_w, mcus_y = (H.height + mcu_h - 1) / mcu_h;
    const int ybw = mcus_x * H.comp[0].h, ybh = mcus_y * H.comp[0].v;
    bits_t b = {data + H.scan_pos, len - H.scan_pos, 0, 0, 0, 0};
    long long nsym = 0;
    int restart_left = H.restart_interval, next_rst = 0;
    for (int my = 0; my < mcus_y; ++my) {
        for (int mx = 0; mx < mcus_x; ++mx) {
            if (H.restart_interval && restart_left == 0) {
                /* byte-align, expect RSTn, reset predictors */
                b.cnt = 0;
                if (!b.hit_marker) { /* remaining padding bits were consumed with cnt = 0; now at the marker */
                    if (b.pos + 1 >= b.n || b.p[b.pos] != 0xFF) return JO_CORRUPT;
                }
                size_t q = b.pos;
                while (q < b.n && b.p[q] == 0xFF) ++q;
                if (q >= b.n || b.p[q] != 0xD0 + next_rst) return JO_CORRUPT;
                b.pos = q + 1;
                b.hit_marker = 0;
                next_rst = (next_rst + 1) & 7;
                restart_left = H.restart_interval;
                for (int c = 0; c < H.ncomp; ++c) H.comp[c].pred = 0;
            }
            for (int c = 0; c < H.ncomp; ++c) {
                comp_t* C = &H.comp[c];
                for (int v = 0; v < C->v; ++v) {
                    for (int h = 0; h < C->h; ++h) {
                        int32_t blk[64];
                        int16_t raw[64];
                        memset(blk, 0, sizeof(blk));
                        memset(raw, 0, sizeof(raw));
                        int s;
                        if ((rc = decode_sym(&b, &H.dc[C->td], &s))) return rc;
                        ++nsym;
                        if (s > 11) return JO_CORRUPT;
                        C->pred += extend(get_bits(&b, s), s);
                        raw[0] = (int16_t)C->pred;
                        for (int k = 1; k < 64;) {
                            if ((rc = decode_sym(&b, &H.ac[C->ta], &s))) return rc;
                            ++nsym;
                            int r = s >> 4, sz = s & 15;
                            if (sz == 0) {
                                if (r != 15) break; /* EOB */
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) return JO_CORRUPT;
                            raw[ZIGZAG[k]] = (int16_t)extend(get_bits(&b, sz), sz);
                            ++k;
                        }
                        if (c != 0) continue;
                        const int bx = mx * C->h + h, by = my * C->v + v;
                        if (coef_out) memcpy(coef_out + ((size_t)by * ybw + bx) * 64, raw, sizeof(raw));
                        for (int k = 0; k < 64; ++k) blk[k] = raw[k] * (int32_t)H.q[C->tq][k];
                        uint8_t px[64];
                        idct_islow(blk, px);
                        for (int r = 0; r < 8; ++r) {
                            const int y = by * 8 + r;
                            if (y >= H.height) break;
                            for (int x8 = 0; x8 < 8; ++x8) {
                                const int x = bx * 8 + x8;
                                if (x < H.width) out[(size_t)y * H.width + x] = px[r * 8 + x8];
                            }
                        }
                    }
                }
            }
            if (H.restart_interval) --restart_left;
        }
    }
    if (stats) {
        stats[0] = nsym;
        stats[1] = ybw;
        stats[2] = ybh;
    }
    return JO_OK;
}
