/*
 * mjx_synth.c -- deterministic synthetic baseline-JPEG generator.
 *
 * The reference cannot create inputs (it is a decoder only) and the GPU box has no dataset, so the
 * bench and the property tests need an encoder.  It emits exactly the marker set the reference's
 * parser accepts (jpeg/mod.rs:157-181): SOI, APP0, DQT, SOF0, DHT, SOS, EOI -- no DRI, no APPn>0,
 * 8-bit DQT, Annex-K Huffman tables (none of which has a 1-bit code, huffman.rs:211 `2..17`).
 *
 * Not on the decode hot path; plain C, host only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- T.81 Annex K tables ---------------------------------------------------------------- */
static const uint8_t K_QT_LUMA[64] = {
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t K_QT_CHROMA[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

static const uint8_t K_DC_LUMA_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t K_DC_CHROMA_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t K_DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t K_AC_LUMA_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t K_AC_LUMA_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t K_AC_CHROMA_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t K_AC_CHROMA_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

static const uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ---- pixel content: plane waves + noise --------------------------------------- */
static inline uint64_t xs64(uint64_t *s)
{
    uint64_t x = *s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1DULL;
}

/* R,G,B = 127 + A*(three plane waves, periods 23..130 px) + ~N(0, sigma^2) (Irwin-Hall of 4 uniforms), clipped. */
void mjxs_fill_rgb(uint8_t *rgb, int w, int h, uint64_t seed, float noise_sigma)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;
    if (!s) s = 1;
    for (int k = 0; k < 8; k++) xs64(&s);
    float kx[3][3], ky[3][3], ph[3][3], amp[3][3];
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) {
            float period = 23.0f + (float)(xs64(&s) % 10700) * 0.01f;          /* 23 .. 130 px */
            float theta = (float)(xs64(&s) % 62832) * 1e-4f;
            kx[c][k] = 6.2831853f / period * cosf(theta);
            ky[c][k] = 6.2831853f / period * sinf(theta);
            ph[c][k] = (float)(xs64(&s) % 62832) * 1e-4f;
            amp[c][k] = 20.0f + (float)(xs64(&s) % 2000) * 0.01f;              /* 20 .. 40 */
        }
    float *sx = (float *)malloc(sizeof(float) * (size_t)w * 18), *cx = sx + (size_t)w * 9;
    for (int x = 0; x < w; x++)
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 3; k++) {
                sx[(size_t)x * 9 + c * 3 + k] = sinf(kx[c][k] * (float)x + ph[c][k]);
                cx[(size_t)x * 9 + c * 3 + k] = cosf(kx[c][k] * (float)x + ph[c][k]);
            }
    const float nscale = noise_sigma * 1.7320508f / 65536.0f;   /* sum of 4 U(0,65536): var = 4*65536^2/12 */
    for (int y = 0; y < h; y++) {
        float sy[9], cy[9];
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 3; k++) {
                sy[c * 3 + k] = sinf(ky[c][k] * (float)y);
                cy[c * 3 + k] = cosf(ky[c][k] * (float)y);
            }
        uint8_t *row = rgb + (size_t)y * w * 3;
        for (int x = 0; x < w; x++) {
            const float *sxx = sx + (size_t)x * 9, *cxx = cx + (size_t)x * 9;
            for (int c = 0; c < 3; c++) {
                float v = 127.0f;
                for (int k = 0; k < 3; k++) v += amp[c][k] * (sxx[c * 3 + k] * cy[c * 3 + k] + cxx[c * 3 + k] * sy[c * 3 + k]);
                uint64_t r = xs64(&s);
                float u = (float)((r & 0xffff) + ((r >> 16) & 0xffff) + ((r >> 32) & 0xffff) + (r >> 48)) - 131070.0f;
                v += u * nscale;
                row[x * 3 + c] = (uint8_t)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v + 0.5f));
            }
        }
    }
    free(sx);
}

/* ---- encoder ----------------------------------------------------------------------------- */
typedef struct { uint16_t code[256]; uint8_t len[256]; } henc;

static void henc_build(henc *t, const uint8_t bits[16], const uint8_t *vals)
{
    memset(t, 0, sizeof *t);
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++, k++) { t->code[vals[k]] = (uint16_t)code++; t->len[vals[k]] = (uint8_t)l; }
        code <<= 1;
    }
}

typedef struct { uint8_t *p; size_t n, cap; uint64_t acc; int nacc; int overflow; } bitw;

static inline void bw_byte(bitw *b, uint8_t v)
{
    if (b->n + 2 > b->cap) { b->overflow = 1; return; }
    b->p[b->n++] = v;
    if (v == 0xff) b->p[b->n++] = 0x00;
}
static inline void bw_put(bitw *b, unsigned v, int n)
{
    b->acc = (b->acc << n) | (v & ((1u << n) - 1));
    b->nacc += n;
    while (b->nacc >= 8) { bw_byte(b, (uint8_t)(b->acc >> (b->nacc - 8))); b->nacc -= 8; }
}
static inline void bw_flush(bitw *b)
{
    if (b->nacc > 0) bw_put(b, 0x7f, 8 - b->nacc);     /* pad with 1-bits */
}

static float DCTM[8][8];   /* DCTM[u][x] = 0.5*alpha(u)*cos((2x+1)u*pi/16) */
static int dct_ready = 0;
static void dct_init(void)
{
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++)
            DCTM[u][x] = (float)(0.5 * (u == 0 ? 0.70710678118654752 : 1.0) * cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
    dct_ready = 1;
}

static void fdct_quant(const float in[64], const uint16_t q[64], int16_t out_zz[64])
{
    float tmp[64], res[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int x = 0; x < 8; x++) s += DCTM[u][x] * in[y * 8 + x];
            tmp[y * 8 + u] = s;
        }
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int y = 0; y < 8; y++) s += DCTM[v][y] * tmp[y * 8 + u];
            res[v * 8 + u] = s;
        }
    for (int k = 0; k < 64; k++) {
        float r = res[ZZ[k]] / (float)q[ZZ[k]];
        int iv = (int)(r < 0 ? r - 0.5f : r + 0.5f);
        if (iv > 1023) iv = 1023;
        if (iv < -1023) iv = -1023;
        out_zz[k] = (int16_t)iv;
    }
}

static inline int nbits_of(int v)
{
    int a = v < 0 ? -v : v, n = 0;
    while (a) { n++; a >>= 1; }
    return n;
}

static void encode_block(bitw *b, const int16_t zz[64], int *pred, const henc *dc, const henc *ac)
{
    int diff = zz[0] - *pred;
    *pred = zz[0];
    int s = nbits_of(diff);
    bw_put(b, dc->code[s], dc->len[s]);
    if (s) bw_put(b, (unsigned)(diff < 0 ? diff - 1 : diff), s);
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = zz[k];
        if (v == 0) { run++; continue; }
        while (run > 15) { bw_put(b, ac->code[0xf0], ac->len[0xf0]); run -= 16; }
        s = nbits_of(v);
        int sym = (run << 4) | s;
        bw_put(b, ac->code[sym], ac->len[sym]);
        bw_put(b, (unsigned)(v < 0 ? v - 1 : v), s);
        run = 0;
    }
    if (run) bw_put(b, ac->code[0x00], ac->len[0x00]);
}

static void put16(uint8_t **p, unsigned v) { *(*p)++ = (uint8_t)(v >> 8); *(*p)++ = (uint8_t)v; }

/* subsampling: 0 = 4:4:4, 1 = 4:2:2 (Y 2x1), 2 = 4:2:0 (Y 2x2), 3 = greyscale, 4 = 4:4:0 (Y 1x2).
 * flags: MJXS_DQT16 = write the quantisation tables with 16-bit precision (Pq = 1, jpeg/mod.rs:245-256) and do not
 *        clamp their values at 255 (low qualities then give values up to 99 * 50 = 4950, like libjpeg without force_baseline).
 * Returns bytes written, or 0 if `cap` is too small. */
enum { MJXS_DQT16 = 1 };
size_t mjxs_encode_ex(const uint8_t *rgb, int w, int h, int subsampling, int quality, int flags, uint8_t *out, size_t cap)
{
    if (!dct_ready) dct_init();
    if (w < 1 || h < 1 || w > 65535 || h > 65535 || cap < 1024) return 0;
    const int ncomp = subsampling == 3 ? 1 : 3;
    const int hy = (subsampling == 1 || subsampling == 2) ? 2 : 1;
    const int vy = (subsampling == 2 || subsampling == 4) ? 2 : 1;
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint16_t qt[2][64];
    for (int k = 0; k < 64; k++) {
        int a = (K_QT_LUMA[k] * scale + 50) / 100, c = (K_QT_CHROMA[k] * scale + 50) / 100;
        const int top = (flags & MJXS_DQT16) ? 65535 : 255;
        qt[0][k] = (uint16_t)(a < 1 ? 1 : (a > top ? top : a));
        qt[1][k] = (uint16_t)(c < 1 ? 1 : (c > top ? top : c));
    }
    henc hdc[2], hac[2];
    henc_build(&hdc[0], K_DC_LUMA_BITS, K_DC_VALS);
    henc_build(&hdc[1], K_DC_CHROMA_BITS, K_DC_VALS);
    henc_build(&hac[0], K_AC_LUMA_BITS, K_AC_LUMA_VALS);
    henc_build(&hac[1], K_AC_CHROMA_BITS, K_AC_CHROMA_VALS);

    uint8_t *p = out;
    *p++ = 0xff; *p++ = 0xd8;
    /* APP0 JFIF */
    *p++ = 0xff; *p++ = 0xe0; put16(&p, 16);
    memcpy(p, "JFIF\0", 5); p += 5;
    *p++ = 1; *p++ = 1; *p++ = 0; put16(&p, 1); put16(&p, 1); *p++ = 0; *p++ = 0;
    /* DQT (zig-zag order; 8 bit, or 16 bit big-endian with Pq = 1 in the high nibble) */
    if (flags & MJXS_DQT16) {
        *p++ = 0xff; *p++ = 0xdb; put16(&p, 2 + 129 * (ncomp == 1 ? 1 : 2));
        for (int t = 0; t < (ncomp == 1 ? 1 : 2); t++) { *p++ = (uint8_t)(0x10 | t); for (int k = 0; k < 64; k++) put16(&p, qt[t][ZZ[k]]); }
    } else {
        *p++ = 0xff; *p++ = 0xdb; put16(&p, 2 + 65 * (ncomp == 1 ? 1 : 2));
        for (int t = 0; t < (ncomp == 1 ? 1 : 2); t++) { *p++ = (uint8_t)t; for (int k = 0; k < 64; k++) *p++ = (uint8_t)qt[t][ZZ[k]]; }
    }
    /* SOF0 */
    *p++ = 0xff; *p++ = 0xc0; put16(&p, 8 + 3 * ncomp); *p++ = 8; put16(&p, (unsigned)h); put16(&p, (unsigned)w); *p++ = (uint8_t)ncomp;
    *p++ = 1; *p++ = (uint8_t)((ncomp == 1 ? 0x11 : (hy << 4) | vy)); *p++ = 0;
    if (ncomp == 3) { *p++ = 2; *p++ = 0x11; *p++ = 1; *p++ = 3; *p++ = 0x11; *p++ = 1; }
    /* DHT: one segment */
    {
        const uint8_t *bits[4] = {K_DC_LUMA_BITS, K_AC_LUMA_BITS, K_DC_CHROMA_BITS, K_AC_CHROMA_BITS};
        const uint8_t *vals[4] = {K_DC_VALS, K_AC_LUMA_VALS, K_DC_VALS, K_AC_CHROMA_VALS};
        const int nvals[4] = {12, 162, 12, 162};
        const uint8_t tc[4] = {0x00, 0x10, 0x01, 0x11};
        int nt = ncomp == 1 ? 2 : 4, total = 2;
        for (int t = 0; t < nt; t++) total += 17 + nvals[t];
        *p++ = 0xff; *p++ = 0xc4; put16(&p, (unsigned)total);
        for (int t = 0; t < nt; t++) { *p++ = tc[t]; memcpy(p, bits[t], 16); p += 16; memcpy(p, vals[t], (size_t)nvals[t]); p += nvals[t]; }
    }
    /* SOS */
    *p++ = 0xff; *p++ = 0xda; put16(&p, 6 + 2 * ncomp); *p++ = (uint8_t)ncomp;
    *p++ = 1; *p++ = 0x00;
    if (ncomp == 3) { *p++ = 2; *p++ = 0x11; *p++ = 3; *p++ = 0x11; }
    *p++ = 0; *p++ = 63; *p++ = 0;

    /* planes, padded to whole MCUs by edge replication */
    const int mw = 8 * (ncomp == 1 ? 1 : hy), mh = 8 * (ncomp == 1 ? 1 : vy);
    const int mcux = (w + mw - 1) / mw, mcuy = (h + mh - 1) / mh;
    const int pw = mcux * mw, phh = mcuy * mh;
    float *Y = (float *)malloc(sizeof(float) * (size_t)pw * phh * 3);
    if (!Y) return 0;
    float *Cb = Y + (size_t)pw * phh, *Cr = Cb + (size_t)pw * phh;
    for (int y = 0; y < phh; y++) {
        const uint8_t *row = rgb + (size_t)(y < h ? y : h - 1) * w * 3;
        for (int x = 0; x < pw; x++) {
            const uint8_t *px = row + (size_t)(x < w ? x : w - 1) * 3;
            float r = px[0], g = px[1], b = px[2];
            Y[(size_t)y * pw + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
            Cb[(size_t)y * pw + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
            Cr[(size_t)y * pw + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
        }
    }
    bitw bw = {p, 0, cap - (size_t)(p - out) - 2, 0, 0, 0};
    int pred[3] = {0, 0, 0};
    const int hyy = ncomp == 1 ? 1 : hy, vyy = ncomp == 1 ? 1 : vy;
    for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
            float blk[64];
            int16_t zz[64];
            for (int bv = 0; bv < vyy; bv++)
                for (int bh = 0; bh < hyy; bh++) {
                    int x0 = mx * mw + bh * 8, y0 = my * mh + bv * 8;
                    for (int yy = 0; yy < 8; yy++)
                        for (int xx = 0; xx < 8; xx++) blk[yy * 8 + xx] = Y[(size_t)(y0 + yy) * pw + x0 + xx];
                    fdct_quant(blk, qt[0], zz);
                    encode_block(&bw, zz, &pred[0], &hdc[0], &hac[0]);
                }
            if (ncomp == 3) {
                float *planes[2] = {Cb, Cr};
                for (int c = 0; c < 2; c++) {
                    int x0 = mx * mw, y0 = my * mh;
                    for (int yy = 0; yy < 8; yy++)
                        for (int xx = 0; xx < 8; xx++) {
                            float s = 0;
                            for (int dy = 0; dy < vy; dy++)
                                for (int dx = 0; dx < hy; dx++) s += planes[c][(size_t)(y0 + yy * vy + dy) * pw + x0 + xx * hy + dx];
                            blk[yy * 8 + xx] = s / (float)(hy * vy);
                        }
                    fdct_quant(blk, qt[1], zz);
                    encode_block(&bw, zz, &pred[1 + c], &hdc[1], &hac[1]);
                }
            }
        }
    bw_flush(&bw);
    free(Y);
    if (bw.overflow) return 0;
    p += bw.n;
    *p++ = 0xff; *p++ = 0xd9;
    return (size_t)(p - out);
}

size_t mjxs_encode(const uint8_t *rgb, int w, int h, int subsampling, int quality, uint8_t *out, size_t cap)
{
    return mjxs_encode_ex(rgb, w, h, subsampling, quality, 0, out, cap);
}

/* generate content for `seed` and encode it; returns bytes written (0 = cap too small) */
size_t mjxs_synth_jpeg_ex(int w, int h, int subsampling, int quality, int flags, uint64_t seed, float noise_sigma,
                          uint8_t *out, size_t cap)
{
    uint8_t *rgb = (uint8_t *)malloc((size_t)w * h * 3);
    if (!rgb) return 0;
    mjxs_fill_rgb(rgb, w, h, seed, noise_sigma);
    size_t n = mjxs_encode_ex(rgb, w, h, subsampling, quality, flags, out, cap);
    free(rgb);
    return n;
}

size_t mjxs_synth_jpeg(int w, int h, int subsampling, int quality, uint64_t seed, float noise_sigma,
                       uint8_t *out, size_t cap)
{
    return mjxs_synth_jpeg_ex(w, h, subsampling, quality, 0, seed, noise_sigma, out, cap);
}

#ifdef __cplusplus
}
#endif
