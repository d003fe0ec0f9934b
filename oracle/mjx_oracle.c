/*
 * mjx_oracle.c -- CPU restatement of martinhath/jpeg-rust's decode path.
 *
 * TEST INFRASTRUCTURE ONLY (see mjx_oracle.h).  Build with
 *     gcc -O2 -ffp-contract=off -fno-fast-math
 * so every f32 operation rounds exactly like the reference's safe-Rust f32 code.
 *
 * Reference line numbers refer to /root/reference/src/{transform.rs, jpeg/mod.rs,
 * jpeg/decoder.rs, jpeg/huffman.rs}.  Rust panics (index out of bounds, unwrap on None,
 * assert!, arithmetic overflow in the debug profile the reference's Makefile runs) are
 * mapped to ORC_ERR_REF_PANIC instead of crashing.
 */
#include "mjx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * panic plumbing: a per-decode jmp_buf so deep helpers can "panic" like the Rust code does.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    jmp_buf jb;
    int code;
    char msg[160];
    /* allocations to release on panic */
    void *allocs[64];
    int nallocs;
} orc_env;

static void orc_panic(orc_env *env, int code, const char *msg)
{
    env->code = code;
    snprintf(env->msg, sizeof env->msg, "%s", msg);
    longjmp(env->jb, 1);
}

static void *orc_alloc(orc_env *env, size_t n)
{
    void *p = calloc(n ? n : 1, 1);
    if (!p) orc_panic(env, ORC_ERR_NOMEM, "out of memory");
    if (env->nallocs >= 64) { free(p); orc_panic(env, ORC_ERR_NOMEM, "too many allocations"); }
    env->allocs[env->nallocs++] = p;
    return p;
}

#define PANIC_IF(cond, text) do { if (cond) orc_panic(env, ORC_ERR_REF_PANIC, text); } while (0)

/* ------------------------------------------------------------------------------------------
 * huffman.rs
 * ------------------------------------------------------------------------------------------ */

/* huffman.rs:13-21 HuffmanCode, 24-27 HuffmanTable */
typedef struct { uint8_t length; uint16_t code; uint8_t value; } orc_hcode;
typedef struct {
    int present;
    int ncodes;
    orc_hcode codes[256];
    /* acceleration for faithful_huff=0: index of the first code of each length, count */
    int first[17], count[17];
} orc_htable;

/* huffman.rs:37-58 from_size_data_tables + 80-98 make_code_table (T.81 Figure C.2) */
static void orc_htable_build(orc_env *env, orc_htable *t, const uint8_t size_data[16],
                             const uint8_t *data_table, int ndata)
{
    uint8_t lengths[16 * 255 + 1];
    int nlen = 0;
    for (int i = 0; i < 16; i++)                 /* :39-41 */
        for (int k = 0; k < size_data[i]; k++) lengths[nlen++] = (uint8_t)(i + 1);

    /* make_code_table :80-98 */
    uint16_t codes[16 * 255 + 1];
    int ncode = 0;
    PANIC_IF(nlen == 0, "make_code_table: sizes[0] on empty table");   /* :85 sizes[0] */
    {
        uint16_t code = 0;
        uint8_t current_size = lengths[0];
        for (int i = 0; i < nlen; i++) {
            uint8_t size = lengths[i];
            while (size > current_size) { code = (uint16_t)(code << 1); current_size++; }  /* :87-90 */
            codes[ncode++] = code;                                                           /* :91 */
            if (current_size > 16 || code == 0xffff) break;                                  /* :92-94 */
            code++;                                                                          /* :95 */
        }
    }
    /* :46-56 zip(data_table, code_lengths, code_table): length = the shortest of the three */
    int n = ndata;
    if (nlen < n) n = nlen;
    if (ncode < n) n = ncode;
    if (n > 256) n = 256;
    memset(t, 0, sizeof *t);
    t->present = 1;
    t->ncodes = n;
    for (int i = 0; i < n; i++) {
        t->codes[i].length = lengths[i];
        t->codes[i].code = codes[i];
        t->codes[i].value = data_table[i];
    }
    for (int l = 0; l <= 16; l++) { t->first[l] = 0; t->count[l] = 0; }
    /* codes_of_length :60-76: the first contiguous run of codes with that length */
    for (int l = 1; l <= 16; l++) {
        int a = 0;
        while (a < n && t->codes[a].length != l) a++;
        int b = a;
        while (b < n && t->codes[b].length == l) b++;
        if (a < n) { t->first[l] = a; t->count[l] = b - a; }
    }
}

/* huffman.rs:109-121 HuffmanDecoder */
typedef struct {
    const uint8_t *data;
    size_t len;
    size_t next_index;
    size_t bits_read;
    uint32_t current;
    uint64_t total_bits;   /* bookkeeping only: bits consumed so far */
} orc_hdec;

/* huffman.rs:124-135 */
static void orc_hdec_new(orc_env *env, orc_hdec *d, const uint8_t *data, size_t len, int allow_short)
{
    /* NOT reference behaviour (allow_short: STANDARD layout, not strict_ref): a scan shorter than the four bytes the
       reference preloads -- a flat 8x8 grey picture has one byte of entropy data -- is read with the 0xAA padding the
       reference itself uses past the end of the data (:236-246) */
    PANIC_IF(len < 4 && !(allow_short && len >= 1), "HuffmanDecoder::new: data[0..4] out of bounds");
    d->data = data;
    d->len = len;
    d->current = 0;
    for (size_t k = 0; k < 4; k++) d->current = (d->current << 8) | (k < len ? data[k] : 0xaau);
    d->next_index = 4;
    d->bits_read = 0;
    d->total_bits = 0;
}

/* huffman.rs:231-254 shift_and_fix_current */
static void orc_shift_and_fix(orc_hdec *d, size_t n)
{
    if (n == 0) return;
    d->current = (n >= 32) ? 0 : (d->current << n);
    d->bits_read += n;
    d->total_bits += n;
    while (d->bits_read >= 8) {
        d->bits_read -= 8;
        uint32_t next_num = (d->next_index >= d->len) ? 0xaau : d->data[d->next_index];  /* :236-246 */
        d->current |= next_num << d->bits_read;
        d->next_index++;
    }
}

/* huffman.rs:198-208 read_n_bits */
static uint16_t orc_read_n_bits(orc_env *env, orc_hdec *d, size_t n)
{
    if (n == 0) return 0;
    PANIC_IF(n > 16, "Should not read more than 16 bits at a time!");      /* :202 */
    uint16_t mask = (uint16_t)(0xffff0000u >> n);                           /* BIT_MASKS[n] :5-6 */
    uint16_t current_16 = (uint16_t)(d->current >> 16);
    uint16_t number = (uint16_t)((current_16 & mask) >> (16 - n));
    orc_shift_and_fix(d, n);
    return number;
}

/* huffman.rs:211-227 next_code: lengths 2..=16 ascending; returns -1 for None */
static int orc_next_code(orc_hdec *d, const orc_htable *t, int faithful)
{
    for (int len = (faithful & 2) ? 1 : 2; len < 17; len++) {       /* bit 1 of `faithful`: ext_1bit (not reference) */
        uint16_t mask = (uint16_t)(0xffff0000u >> len);
        uint16_t current_16 = (uint16_t)(d->current >> 16);
        uint16_t bits = (uint16_t)((current_16 & mask) >> (16 - len));
        int a, b;
        if (faithful & 1) {
            /* codes_of_length :60-76 re-scans the whole vector every call */
            a = 0;
            while (a < t->ncodes && t->codes[a].length != len) a++;
            b = a;
            while (b < t->ncodes && t->codes[b].length == len) b++;
            if (a >= t->ncodes) { a = 0; b = 0; }
        } else {
            a = t->first[len];
            b = a + t->count[len];
        }
        for (int i = a; i < b; i++) {                 /* :220 linear find */
            if (t->codes[i].code == bits) {
                orc_shift_and_fix(d, (size_t)len);    /* :222 */
                return t->codes[i].value;
            }
        }
    }
    return -1;
}

/* huffman.rs:256-268 value_correction (T.81 Table F.2 / EXTEND) */
static int16_t orc_value_correction(uint16_t val, size_t len)
{
    if (len == 0) return 0;
    int16_t v = (int16_t)val;
    int16_t base = (int16_t)(1u << (len - 1));
    if (v < base) return (int16_t)(-2 * base + 1 + v);
    return v;
}

/* huffman.rs:146-195 next_block */
static void orc_next_block(orc_env *env, orc_hdec *d, const orc_htable *ac, const orc_htable *dc,
                           int faithful, int16_t block[64])
{
    int num_bits = orc_next_code(d, dc, faithful);
    PANIC_IF(num_bits < 0, "DC lookup fail (unwrap on None)");                      /* :151-156 */
    int16_t dc_coef = orc_value_correction(orc_read_n_bits(env, d, (size_t)num_bits), (size_t)num_bits);
    int len = 0;
    block[len++] = dc_coef;                                                          /* :159 */
    while (len < 64) {                                                               /* :161 */
        int next_code = orc_next_code(d, ac, faithful);
        PANIC_IF(next_code < 0, "ILLEGAL STATE! (AC lookup fail)");                  /* :162 */
        if (next_code == 0x00) {                                                     /* :164-169 */
            while (len < 64) block[len++] = 0;
            break;
        }
        if (next_code == 0xf0) {                                                     /* :170-175 */
            int to_push = 64 - len < 16 ? 64 - len : 16;
            for (int i = 0; i < to_push; i++) block[len++] = 0;
            continue;
        }
        int prepending_zeroes = (next_code & 0xf0) >> 4;                             /* :183 */
        size_t nb = (size_t)(next_code & 0xf);                                       /* :184 */
        uint16_t num = orc_read_n_bits(env, d, nb);
        int16_t number = orc_value_correction(num, nb);
        int room = 64 - len - 1;
        int zeroes_to_push = prepending_zeroes < room ? prepending_zeroes : room;    /* :187 */
        for (int i = 0; i < zeroes_to_push; i++) block[len++] = 0;
        block[len++] = number;                                                       /* :189 */
    }
    PANIC_IF(len != 64, "assert!(block.len() == 64)");                               /* :192 */
}

/* ------------------------------------------------------------------------------------------
 * transform.rs
 * ------------------------------------------------------------------------------------------ */

static float orc_cos_tab[8][8];
static pthread_once_t orc_cos_once = PTHREAD_ONCE_INIT;

/* the f32 argument of transform.rs:78-79: ((2*x+1) * u * Pi / 16).cos() */
static float orc_cos_arg(int x, int u)
{
    const float Pi = 3.14159265358979323846f;        /* transform.rs:16 PI as f32 */
    float xf = (float)x, uf = (float)u;
    float a = 2.0f * xf;
    a = a + 1.0f;
    a = a * uf;
    a = a * Pi;
    a = a / 16.0f;
    return a;
}

static void orc_cos_init(void)
{
    for (int x = 0; x < 8; x++)
        for (int u = 0; u < 8; u++) orc_cos_tab[x][u] = cosf(orc_cos_arg(x, u));
}

/* transform.rs:55-87 discrete_cosine_transform_inverse, d = 8.
 * Accumulation order: y, x outer; v outer / u inner; product left-associated
 * alpha(u)*alpha(v)*f_uv*cos_x*cos_y; sum/4 at the end.  alpha(0) = 1/sqrt(2) in f32. */
void orc_idct_ref(const float in[64], float out[64], int faithful_cos)
{
    const float a0 = 1.0f / sqrtf(2.0f);             /* :57-59 1f32 / 2f32.sqrt() */
    pthread_once(&orc_cos_once, orc_cos_init);
    for (int y = 0; y < 8; y++) {
        for (int x = 0; x < 8; x++) {
            float sum = 0.0f;
            for (int v = 0; v < 8; v++) {
                for (int u = 0; u < 8; u++) {
                    float au = (u == 0) ? a0 : 1.0f;
                    float av = (v == 0) ? a0 : 1.0f;
                    float f_uv = in[v * 8 + u];
                    float cx, cy;
                    if (faithful_cos) {
                        cx = cosf(orc_cos_arg(x, u));
                        cy = cosf(orc_cos_arg(y, v));
                    } else {
                        cx = orc_cos_tab[x][u];
                        cy = orc_cos_tab[y][v];
                    }
                    float p = au * av;
                    p = p * f_uv;
                    p = p * cx;
                    p = p * cy;
                    sum = sum + p;                     /* :77-79 */
                }
            }
            out[y * 8 + x] = sum / 4.0f;               /* :82 */
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * decoder.rs helpers
 * ------------------------------------------------------------------------------------------ */

/* decoder.rs:404-407 */
static const uint8_t ORC_ZIGZAG[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27,
    20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58,
    59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* decoder.rs:382-390 */
uint8_t orc_f32_to_u8(float n)
{
    if (n < 0.0f) return 0;
    if (n > 255.0f) return 255;
    if (n != n) return 0;      /* NaN: `as u8` saturating cast gives 0 */
    return (uint8_t)n;         /* truncation toward zero */
}

/* decoder.rs:392-402 */
void orc_ycbcr_to_rgb(float y, float cb, float cr, uint8_t rgb[3])
{
    const float c_red = 0.299f, c_green = 0.587f, c_blue = 0.114f;
    float kr = 2.0f * c_red;   kr = 2.0f - kr;       /* (2.0 - 2.0 * c_red) */
    float kb = 2.0f * c_blue;  kb = 2.0f - kb;
    float r = cr * kr;  r = r + y;
    float b = cb * kb;  b = b + y;
    float t1 = c_blue * b;
    float t2 = c_red * r;
    float g = y - t1;
    g = g - t2;
    g = g / c_green;
    rgb[0] = orc_f32_to_u8(r + 128.0f);
    rgb[1] = orc_f32_to_u8(g + 128.0f);
    rgb[2] = orc_f32_to_u8(b + 128.0f);
}

/* ------------------------------------------------------------------------------------------
 * parse state (mod.rs:58-87 JPEGImage fields that matter) + decoder fields (decoder.rs:19-52)
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t id, h, v, tq; } orc_framecomp;        /* mod.rs:104-113 */
typedef struct { uint8_t id, dc_sel, ac_sel; } orc_scancomp;   /* mod.rs:132-139 */

typedef struct {
    uint8_t component, dc_table_id, ac_table_id, quantization_id, h, v;   /* decoder.rs:39-52 */
} orc_compfields;

typedef struct {
    orc_htable ac[4], dc[4];
    int qt_present[4];
    uint16_t qt[4][64];
    int have_frame;
    int ncomp_frame;
    orc_framecomp fcomp[256];
    int width, height;
    int restart_interval;      /* MCUs per restart interval (ext_dri), 0 = none */
    /* ext_multiscan: one entry per scan read so far (one component, or two interleaved) */
    struct {
        int ncomp;
        int comp[2];                   /* indices into fcomp, scan order */
        orc_htable dc[2], ac[2];       /* the tables the scan selected, as they were when its SOS was read */
        const uint8_t *data;           /* its entropy-coded segment, FF00 compacted, RSTn left in */
        size_t len;
        int restart_interval;
    } parts[3];
    int nparts;
} orc_parse;

/* decoder.rs:259-288 get_indices.  usize arithmetic: underflow panics. */
static void orc_get_indices(orc_env *env, long x, long y, long max_x, long x_factor, long y_factor,
                            long max_x_factor, long max_y_factor, long *ox, long *oy)
{
    if (max_y_factor > 1 && y_factor == 1) {
        if (max_x_factor > 1 && x_factor == 1) {
            int is_upper = (y & 1) == 0;
            if (is_upper) {
                int move_down = ((x / 2) & 1) == 1;
                if (move_down) { *ox = x / 2 - 1 + (x & 1); *oy = y + 1; return; }
                *ox = x / 2 + (x & 1); *oy = y; return;
            } else {
                int move_up = y > 0 && ((x / 2) & 1) == 0;
                if (move_up) {
                    PANIC_IF(max_x / 2 + x / 2 < 1, "get_indices: usize underflow");
                    *ox = max_x / 2 + x / 2 - 1 + (x & 1); *oy = y; return;
                }
                PANIC_IF(y < 1, "get_indices: usize underflow");
                *ox = max_x / 2 + x / 2 + (x & 1); *oy = y - 1; return;
            }
        } else {
            if ((y & 1) == 0) { *ox = x / 2; *oy = y + (x & 1); return; }
            PANIC_IF(y < (x & 1), "get_indices: usize underflow");
            *ox = x / 2 + max_x / 2; *oy = y - (x & 1); return;
        }
    }
    *ox = x; *oy = y;
}

/* decoder.rs:347-379 fill_block_in_array */
static void orc_fill_block(orc_env *env, const float *block, float *target, size_t target_len,
                           size_t x_scale, size_t y_scale, size_t x, size_t y, size_t stride)
{
    for (size_t line_number = 0; line_number < 8; line_number++) {
        size_t start_x = x * 8 * x_scale;
        if (stride < start_x) continue;                                 /* :360-362 return from closure */
        size_t start_i = y * 8 * y_scale * stride + line_number * stride + start_x;
        for (size_t ind = 0; ind < 8 * x_scale; ind++) {
            float n = block[line_number * 8 + ind / x_scale];           /* repeat(n).take(x_scale) :356 */
            size_t i = ind + start_i;
            for (size_t j = 0; j < y_scale; j++) {
                if (i + j * stride < target_len) {                      /* :370 guard */
                    size_t idx = i + j * stride * 8;                    /* :371 index (disagrees with guard) */
                    PANIC_IF(idx >= target_len, "fill_block_in_array: index out of bounds");
                    target[idx] = n;
                }
            }
        }
    }
}

/* decoder.rs:162-343 decode() */
static void orc_jpeg_decode(orc_env *env, const orc_opts *opts, const orc_parse *ps,
                            const orc_compfields *cf, int ncomp, const uint8_t *data, size_t data_len,
                            orc_image *out)
{
    const size_t W = (size_t)ps->width, H = (size_t)ps->height;
    const size_t nbx = (W + 7) / 8, nby = (H + 7) / 8;            /* :164-165 */
    const size_t num_blocks = nbx * nby;

    size_t hmax = 1, vmax = 1;                                     /* :175-185 (unwrap_or(1)) */
    if (ncomp > 0) { hmax = 0; vmax = 0; }
    for (int c = 0; c < ncomp; c++) {
        if (cf[c].h > hmax) hmax = cf[c].h;
        if (cf[c].v > vmax) vmax = cf[c].v;
    }
    PANIC_IF(ncomp != 1 && ncomp != 3, "panic!(\"asd\") / unsupported component count");  /* :328-330 */

    int hs[3], vs[3];
    for (int c = 0; c < ncomp; c++) { hs[c] = cf[c].h; vs[c] = cf[c].v; }
    if (opts->layout == ORC_LAYOUT_STD && ncomp == 1) { hs[0] = vs[0] = 1; hmax = vmax = 1; }
    PANIC_IF(hmax == 0 || vmax == 0, "zero sampling factor");

    size_t skip_factor = vmax * hmax;                              /* :191 */
    size_t num_read;                                               /* :192 (Q2) */
    const size_t mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    if (opts->layout == ORC_LAYOUT_REF) num_read = (num_blocks + skip_factor - 1) / skip_factor;
    else num_read = mcux * mcuy;
    out->mcus_read = num_read;

    orc_hdec hd;
    memset(&hd, 0, sizeof hd);
    if (!ps->nparts) orc_hdec_new(env, &hd, data, data_len, opts->layout == ORC_LAYOUT_STD && !opts->strict_ref);   /* :189 */

    /* Step 1 :195-215 */
    float prev_dc[3] = {0, 0, 0};
    size_t per_mcu[3], fill[3] = {0, 0, 0};
    for (int c = 0; c < ncomp; c++) {
        per_mcu[c] = (size_t)hs[c] * (size_t)vs[c];
        out->nblocks[c] = num_read * per_mcu[c];
        out->coef[c] = (int16_t *)calloc(out->nblocks[c] * 64 + 1, sizeof(int16_t));
        if (!out->coef[c]) orc_panic(env, ORC_ERR_NOMEM, "out of memory");
        out->hs[c] = hs[c];
        out->vs[c] = vs[c];
    }
    if (ps->nparts) {
        /* ext_multiscan: every scan carries the blocks of one component in raster order over the component's own block
           grid; each is put where the interleaved order would have it (MCU padding blocks stay zero) */
        for (int q = 0; q < ps->nparts; q++) {
            const int pn = ps->parts[q].ncomp;
            orc_hdec pd;
            orc_hdec_new(env, &pd, ps->parts[q].data, ps->parts[q].len, 1);
            const size_t ri = (size_t)ps->parts[q].restart_interval;
            float pred[2] = {0, 0};
            size_t units, gw;                  /* restart units (blocks / MCUs) of the scan; width of its grid */
            if (pn == 1) {                     /* non-interleaved: raster order over the component's own block grid */
                const int c = ps->parts[q].comp[0];
                gw = ((W * (size_t)hs[c] + hmax - 1) / hmax + 7) / 8;
                units = gw * (((H * (size_t)vs[c] + vmax - 1) / vmax + 7) / 8);
            } else {                           /* interleaved subset: MCUs of the scan's own components */
                size_t hm = 1, vm = 1;
                for (int k = 0; k < pn; k++) {
                    if ((size_t)hs[ps->parts[q].comp[k]] > hm) hm = (size_t)hs[ps->parts[q].comp[k]];
                    if ((size_t)vs[ps->parts[q].comp[k]] > vm) vm = (size_t)vs[ps->parts[q].comp[k]];
                }
                const size_t sw = (W * hm + hmax - 1) / hmax, sh = (H * vm + vmax - 1) / vmax;
                gw = (sw + 8 * hm - 1) / (8 * hm);
                units = gw * ((sh + 8 * vm - 1) / (8 * vm));
            }
            for (size_t u = 0; u < units; u++) {
                for (int k = 0; k < pn; k++) {
                    const int c = ps->parts[q].comp[k];
                    const size_t nh = pn == 1 ? 1 : (size_t)hs[c], nv = pn == 1 ? 1 : (size_t)vs[c];
                    for (size_t yy = 0; yy < nv; yy++)
                        for (size_t xx = 0; xx < nh; xx++) {
                            int16_t blk[64];
                            orc_next_block(env, &pd, &ps->parts[q].ac[k], &ps->parts[q].dc[k], (opts->faithful_huff ? 1 : 0) | (opts->ext_1bit ? 2 : 0), blk);
                            pred[k] += (float)blk[0];
                            const size_t bx = (u % gw) * nh + xx, by = (u / gw) * nv + yy;   /* block position in the component */
                            const size_t m = (by / (size_t)vs[c]) * mcux + bx / (size_t)hs[c];
                            const size_t kk = (by % (size_t)vs[c]) * (size_t)hs[c] + bx % (size_t)hs[c];
                            if (bx / (size_t)hs[c] < mcux && m < num_read) {
                                int16_t *dst = out->coef[c] + (m * per_mcu[c] + kk) * 64;
                                memcpy(dst, blk, sizeof blk);
                                dst[0] = (int16_t)pred[k];
                            }
                        }
                }
                if (ri > 0 && (u + 1) % ri == 0 && u + 1 < units) {
                    orc_shift_and_fix(&pd, (size_t)((8 - pd.total_bits % 8) % 8));
                    const uint32_t mk = pd.current >> 16;
                    if ((mk & 0xfff8u) != 0xffd0u) orc_panic(env, ORC_ERR_UNSUPPORTED, "restart marker expected");
                    orc_shift_and_fix(&pd, 16);
                    pred[0] = pred[1] = 0;
                }
            }
            hd.total_bits += pd.total_bits;
        }
    }
    for (size_t m = 0; m < (ps->nparts ? 0 : num_read); m++) {
        for (int c = 0; c < ncomp; c++) {
            PANIC_IF(cf[c].ac_table_id > 3 || !ps->ac[cf[c].ac_table_id].present, "ac_table unwrap on None"); /* :154-156 */
            PANIC_IF(cf[c].dc_table_id > 3 || !ps->dc[cf[c].dc_table_id].present, "dc_table unwrap on None"); /* :158-160 */
            const orc_htable *ac = &ps->ac[cf[c].ac_table_id];
            const orc_htable *dc = &ps->dc[cf[c].dc_table_id];
            for (size_t k = 0; k < per_mcu[c]; k++) {
                int16_t blk[64];
                orc_next_block(env, &hd, ac, dc, (opts->faithful_huff ? 1 : 0) | (opts->ext_1bit ? 2 : 0), blk);   /* :202 */
                float dcv = (float)blk[0] + prev_dc[c];                        /* :208-210 (f32, exact ints) */
                prev_dc[c] = dcv;
                int16_t *dst = out->coef[c] + (fill[c]++) * 64;
                memcpy(dst, blk, sizeof blk);
                dst[0] = (int16_t)dcv;
            }
        }
        if (ps->restart_interval > 0 && (m + 1) % (size_t)ps->restart_interval == 0 && m + 1 < num_read) {
            /* ext_dri: byte-align, expect RSTn (the markers stay in the copied buffer: only FF00 pairs were compacted) */
            orc_shift_and_fix(&hd, (size_t)((8 - hd.total_bits % 8) % 8));
            const uint32_t mk = hd.current >> 16;
            if ((mk & 0xfff8u) != 0xffd0u) orc_panic(env, ORC_ERR_UNSUPPORTED, "restart marker expected");
            orc_shift_and_fix(&hd, 16);
            prev_dc[0] = prev_dc[1] = prev_dc[2] = 0;
        }
    }
    out->bits_used = (size_t)hd.total_bits;

    /* Step 2 :218-315 */
    const size_t num_pixels = W * H;
    float *plane[3] = {0, 0, 0};
    for (int c = 0; c < ncomp; c++) {
        PANIC_IF(cf[c].quantization_id > 3 || !ps->qt_present[cf[c].quantization_id],
                 "Did not find quantization table");                                    /* :222-225 */
        const uint16_t *qt = ps->qt[cf[c].quantization_id];
        const size_t nb = out->nblocks[c];
        float *cblocks = (float *)orc_alloc(env, nb * 64 * sizeof(float));
        for (size_t b = 0; b < nb; b++) {
            const int16_t *zz = out->coef[c] + b * 64;
            float nat[64];
            for (int k = 0; k < 64; k++) nat[ORC_ZIGZAG[k]] = (float)zz[k] * (float)qt[k];  /* :230-232, 425-437 */
            orc_idct_ref(nat, cblocks + b * 64, opts->faithful_cos);                         /* :234 */
        }
        plane[c] = (float *)orc_alloc(env, (num_pixels ? num_pixels : 1) * sizeof(float));  /* :252-255 zeros */

        if (opts->layout == ORC_LAYOUT_REF) {
            /* :239-250 JPEG A.1.1 via f32 */
            float x_i = ceilf((float)W * ((float)cf[c].h / (float)hmax));
            float y_i = ceilf((float)H * ((float)cf[c].v / (float)vmax));
            float xf_f = ceilf((float)W / x_i), yf_f = ceilf((float)H / y_i);
            /* `as usize` on NaN/inf: NaN -> 0 (then division by zero panics at :290) */
            PANIC_IF(!(xf_f >= 1.0f) || !(yf_f >= 1.0f), "attempt to divide by zero (x_factor/y_factor)");
            size_t x_factor = (size_t)xf_f, y_factor = (size_t)yf_f;
            size_t stride = W;
            size_t block_i = 0;
            for (size_t y = 0; y < nby / y_factor; y++) {                       /* :290 */
                for (size_t x = 0; x < nbx / x_factor; x++) {                   /* :291 */
                    long bx, by;
                    orc_get_indices(env, (long)x, (long)y, (long)nbx, (long)x_factor, (long)y_factor,
                                    (long)hmax, (long)vmax, &bx, &by);
                    PANIC_IF(block_i >= nb, "component_blocks[block_i] out of bounds");  /* :303 */
                    orc_fill_block(env, cblocks + block_i * 64, plane[c], num_pixels, x_factor, y_factor,
                                   (size_t)bx, (size_t)by, stride);
                    block_i++;
                }
            }
        } else {
            /* STANDARD layout: block (bh,bv) of MCU (mx,my) sits at component block coords
             * (mx*h+bh, my*v+bv); box replication by hmax/h x vmax/v; clipped to W x H. */
            size_t xs = hmax / (size_t)hs[c], ys = vmax / (size_t)vs[c];
            PANIC_IF(xs * (size_t)hs[c] != hmax || ys * (size_t)vs[c] != vmax, "non-integer subsampling ratio");
            size_t b = 0;
            for (size_t my = 0; my < mcuy; my++)
                for (size_t mx = 0; mx < mcux; mx++)
                    for (size_t bv = 0; bv < (size_t)vs[c]; bv++)
                        for (size_t bh = 0; bh < (size_t)hs[c]; bh++, b++) {
                            const float *blk = cblocks + b * 64;
                            size_t px0 = (mx * hs[c] + bh) * 8 * xs, py0 = (my * vs[c] + bv) * 8 * ys;
                            for (size_t yy = 0; yy < 8 * ys; yy++) {
                                size_t py = py0 + yy;
                                if (py >= H) break;
                                for (size_t xx = 0; xx < 8 * xs; xx++) {
                                    size_t px = px0 + xx;
                                    if (px >= W) break;
                                    plane[c][py * W + px] = blk[(yy / ys) * 8 + xx / xs];
                                }
                            }
                        }
        }
    }

    /* Step 3 :317-331 */
    out->rgb = (uint8_t *)calloc(num_pixels * 3 + 1, 1);
    if (!out->rgb) orc_panic(env, ORC_ERR_NOMEM, "out of memory");
    if (ncomp == 1) {
        for (size_t i = 0; i < num_pixels; i++) {
            uint8_t u = orc_f32_to_u8(plane[0][i] + 128.0f);
            out->rgb[3 * i] = out->rgb[3 * i + 1] = out->rgb[3 * i + 2] = u;
        }
    } else {
        for (size_t i = 0; i < num_pixels; i++)
            orc_ycbcr_to_rgb(plane[0][i], plane[1][i], plane[2][i], out->rgb + 3 * i);
    }
}

/* ------------------------------------------------------------------------------------------
 * mod.rs:202-465 JPEGImage::parse
 * ------------------------------------------------------------------------------------------ */
#define VEC(i) (((size_t)(i) < len) ? vec[(size_t)(i)] : (orc_panic(env, ORC_ERR_REF_PANIC, "index out of bounds in parse"), (uint8_t)0))

/* ext_multiscan: every component has its scan -> decode the picture (components in frame order) */
static void orc_decode_parts(orc_env *env, const orc_opts *opts, orc_parse *ps, orc_image *out)
{
    int covered = 0;
    for (int q = 0; q < ps->nparts; q++) covered += ps->parts[q].ncomp;
    if (covered != ps->ncomp_frame || covered != 3) orc_panic(env, ORC_ERR_UNSUPPORTED, "multi-scan file: a component without a scan");
    orc_compfields ordered[3];
    for (int c = 0; c < 3; c++) {
        ordered[c].component = ps->fcomp[c].id;
        ordered[c].h = ps->fcomp[c].h;
        ordered[c].v = ps->fcomp[c].v;
        ordered[c].quantization_id = ps->fcomp[c].tq;
        ordered[c].dc_table_id = ordered[c].ac_table_id = 0;
    }
    out->width = ps->width; out->height = ps->height; out->ncomp = 3;
    orc_jpeg_decode(env, opts, ps, ordered, 3, NULL, 0, out);
}

static void orc_parse_and_decode(orc_env *env, const uint8_t *vec, size_t len, const orc_opts *opts,
                                 orc_image *out)
{
    orc_parse *ps = (orc_parse *)orc_alloc(env, sizeof(orc_parse));
    size_t i = 0;
    while (i < len) {
        /* bytes_to_marker :157-181 */
        uint8_t b0 = VEC(i);
        uint8_t n = VEC(i + 1);
        if (b0 == 0xff && n == 0) n = VEC(i + 2);          /* :161-164 */
        int known = 0;
        if (b0 == 0xff) {
            switch (n) {
            case 0xc0: case 0xc4: case 0xd8: case 0xd9: case 0xda: case 0xdb: case 0xdd:
            case 0xe0: case 0xec: case 0xee: case 0xfe: known = 1; break;
            default: known = 0;
            }
        }
        if (!known) {
            if (opts->strict_ref || b0 != 0xff) {
                char m[96];
                snprintf(m, sizeof m, "Unhandled byte marker: %02x %02x (i=%zu/%zu)", b0, VEC(i + 1), i, len);
                orc_panic(env, ORC_ERR_REF_PANIC, m);                              /* :456-462 */
            }
            /* lenient: anything length-prefixed that is not a frame type we cannot decode is skipped */
            if (n == 0xc1 || n == 0xc2 || n == 0xc3 || (n >= 0xc5 && n <= 0xcf && n != 0xc8 && n != 0xcc))
                orc_panic(env, ORC_ERR_UNSUPPORTED, "non-baseline SOF marker");
            if (n == 0x01 || (n >= 0xd0 && n <= 0xd7)) { i += 2; continue; }       /* standalone */
            size_t l = ((size_t)VEC(i + 2) << 8) + VEC(i + 3);
            PANIC_IF(l < 2, "segment length < 2");
            i += 2 + l;
            continue;
        }
        if (n == 0xd9 && ps->nparts) { orc_decode_parts(env, opts, ps, out); return; }   /* ext_multiscan: EOI behind the last scan */
        if (n == 0xd8 || n == 0xd9) { i += 2; continue; }                         /* :209-215 */

        size_t seglen = ((size_t)VEC(i + 2) << 8) + VEC(i + 3);                   /* :219 */
        PANIC_IF(seglen < 2, "attempt to subtract with overflow (length - 2)");
        size_t data_length = seglen - 2;
        i += 4;                                                                   /* :220 */

        switch (n) {
        case 0xfe: /* Comment :223-228 */
            PANIC_IF(i + data_length > len, "comment slice out of bounds");
            break;
        case 0xdb: { /* DQT :229-261 */
            size_t index = i;
            while (index < i + data_length) {
                uint8_t pq = VEC(index);
                uint8_t precision = (pq & 0xf0) >> 4, identifier = pq & 0x0f;
                if (precision == 0) {
                    PANIC_IF(index + 65 > len, "DQT slice out of bounds");
                    PANIC_IF(identifier > 3, "quantization_tables index out of bounds");
                    for (int k = 0; k < 64; k++) ps->qt[identifier][k] = vec[index + 1 + k];
                    ps->qt_present[identifier] = 1;
                    index += 65;
                } else if (precision == 1) {
                    PANIC_IF(index + 129 > len, "DQT slice out of bounds");
                    PANIC_IF(identifier > 3, "quantization_tables index out of bounds");
                    for (int k = 0; k < 64; k++)
                        ps->qt[identifier][k] = (uint16_t)((vec[index + 1 + 2 * k] << 8) | vec[index + 2 + 2 * k]);
                    ps->qt_present[identifier] = 1;
                    index += 129;
                } else {
                    orc_panic(env, ORC_ERR_REF_PANIC, "Unknown precision of quantization table");  /* :258 */
                }
            }
            break;
        }
        case 0xc0: { /* SOF0 :262-298 */
            uint8_t ncomp = VEC(i + 5);
            ps->height = (VEC(i + 1) << 8) + VEC(i + 2);
            ps->width = (VEC(i + 3) << 8) + VEC(i + 4);
            size_t index = i + 6;
            for (int c = 0; c < ncomp; c++) {
                uint8_t id = VEC(index), hv = VEC(index + 1), tq = VEC(index + 2);
                uint8_t h = (hv & 0xf0) >> 4, v = hv & 0x0f;
                PANIC_IF(!(h > 0 && h < 3), "assert!(horizontal_sampling_factor > 0 && < 3)");  /* :275-276 */
                PANIC_IF(!(v > 0 && v < 3), "assert!(vertical_sampling_factor > 0 && < 3)");    /* :277 */
                ps->fcomp[c].id = id; ps->fcomp[c].h = h; ps->fcomp[c].v = v; ps->fcomp[c].tq = tq;
                index += 3;
            }
            ps->ncomp_frame = ncomp;
            ps->have_frame = 1;
            break;
        }
        case 0xc4: { /* DHT :299-336 */
            size_t hi = i, segment_end = i + data_length;
            while (hi < segment_end) {
                uint8_t tc = VEC(hi);
                uint8_t table_class = (tc & 0xf0) >> 4, dest = tc & 0x0f;
                hi += 1;
                PANIC_IF(hi + 16 > len, "DHT size_area out of bounds");
                const uint8_t *size_area = vec + hi;
                hi += 16;
                size_t ncodes = 0;
                for (int k = 0; k < 16; k++) ncodes += size_area[k];
                PANIC_IF(hi + ncodes > len, "DHT data_area out of bounds");
                const uint8_t *data_area = vec + hi;
                hi += ncodes;
                PANIC_IF(dest > 3, "huffman tables index out of bounds");
                orc_htable_build(env, table_class == 0 ? &ps->dc[dest] : &ps->ac[dest], size_area,
                                 data_area, (int)ncodes);                          /* :325-335 DC = 0, AC otherwise */
            }
            break;
        }
        case 0xda: { /* SOS :337-423 */
            uint8_t num_components = VEC(i);
            orc_scancomp sc[256];
            for (int c = 0; c < num_components; c++) {          /* :341-348 */
                sc[c].id = VEC(i + 1);
                sc[c].dc_sel = (VEC(i + 2) & 0xf0) >> 4;
                sc[c].ac_sel = VEC(i + 2) & 0x0f;
                i += 2;
            }
            (void)VEC(i + 3);                                    /* :356-359 reads vec[i+1..i+3] */
            i += 4;                                              /* :362 */

            if (opts->ext_multiscan && opts->layout == ORC_LAYOUT_STD && !opts->strict_ref && ps->have_frame &&
                (num_components < ps->ncomp_frame || ps->nparts)) {
                /* NOT reference behaviour: one scan of several.  Its data runs up to the next marker that is not RSTn. */
                if (num_components < 1 || num_components > 2 || ps->nparts >= 3) orc_panic(env, ORC_ERR_UNSUPPORTED, "multi-scan file: scans of one or two components");
                const int q = ps->nparts;
                ps->parts[q].ncomp = num_components;
                for (int k = 0; k < num_components; k++) {
                    int ci = -1;
                    for (int c = 0; c < ps->ncomp_frame; c++) if (ps->fcomp[c].id == sc[k].id) ci = c;
                    if (ci < 0 || sc[k].dc_sel > 3 || sc[k].ac_sel > 3) orc_panic(env, ORC_ERR_UNSUPPORTED, "multi-scan file: unknown component or table");
                    for (int p = 0; p < ps->nparts; p++)
                        for (int j = 0; j < ps->parts[p].ncomp; j++) if (ps->parts[p].comp[j] == ci) orc_panic(env, ORC_ERR_UNSUPPORTED, "multi-scan file: component coded twice");
                    PANIC_IF(!ps->dc[sc[k].dc_sel].present || !ps->ac[sc[k].ac_sel].present, "table unwrap on None");
                    ps->parts[q].comp[k] = ci;
                    ps->parts[q].dc[k] = ps->dc[sc[k].dc_sel];
                    ps->parts[q].ac[k] = ps->ac[sc[k].ac_sel];
                }
                uint8_t *pd = (uint8_t *)orc_alloc(env, len - (i < len ? i : len) + 8);
                size_t np = 0, k = i;
                while (k < len) {
                    if (vec[k] != 0xff) { pd[np++] = vec[k++]; continue; }
                    if (k + 1 >= len) { pd[np++] = vec[k++]; break; }
                    if (vec[k + 1] == 0x00) { pd[np++] = 0xff; k += 2; continue; }
                    if (vec[k + 1] == 0xff) { k += 1; continue; }                                  /* fill byte */
                    if ((vec[k + 1] & 0xf8) == 0xd0) { pd[np++] = 0xff; pd[np++] = vec[k + 1]; k += 2; continue; }   /* RSTn stays */
                    break;
                }
                ps->nparts++;
                ps->parts[q].data = pd;
                ps->parts[q].len = np;
                ps->parts[q].restart_interval = ps->restart_interval;
                i = k;
                continue;
            }

            /* :371-385 copy data, replace ff00 with ff, to end of file */
            uint8_t *enc = (uint8_t *)orc_alloc(env, len - (i < len ? i : len) + 8);
            size_t nenc = 0;
            {
                size_t k = i;
                while (k < len) {
                    enc[nenc++] = vec[k];
                    if (vec[k] == 0xff) {
                        PANIC_IF(k + 1 >= len, "destuff: vec[i + 1] out of bounds");   /* :377 */
                        if (vec[k + 1] == 0x00) k += 1;
                    }
                    k += 1;
                }
            }
            PANIC_IF(!ps->have_frame, "frame_header unwrap on None");      /* :388 */

            /* decoder.rs:83-111 frame_header() then :113-152 scan_header() */
            orc_compfields cf[256 + 256];
            int ncf = 0;
            for (int c = 0; c < ps->ncomp_frame; c++) {
                int found = -1;
                for (int k = 0; k < ncf; k++) if (cf[k].component == ps->fcomp[c].id) { found = k; break; }
                if (found < 0) { found = ncf++; cf[found].component = ps->fcomp[c].id; cf[found].dc_table_id = 0xff; cf[found].ac_table_id = 0xff; }
                cf[found].h = ps->fcomp[c].h; cf[found].v = ps->fcomp[c].v; cf[found].quantization_id = ps->fcomp[c].tq;
            }
            for (int c = 0; c < num_components; c++) {
                int found = -1;
                for (int k = 0; k < ncf; k++) if (cf[k].component == sc[c].id) { found = k; break; }
                if (found >= 0) { cf[found].ac_table_id = sc[c].ac_sel; cf[found].dc_table_id = sc[c].dc_sel; }
                else {   /* :128-138 (note the swapped selectors in the reference's insert path) */
                    found = ncf++;
                    cf[found].component = sc[c].id; cf[found].h = 0xff; cf[found].v = 0xff; cf[found].quantization_id = 0xff;
                    cf[found].dc_table_id = sc[c].ac_sel; cf[found].ac_table_id = sc[c].dc_sel;
                }
            }
            orc_compfields ordered[256];                         /* :141-150 scan order */
            for (int c = 0; c < num_components; c++) {
                int found = -1;
                for (int k = 0; k < ncf; k++) if (cf[k].component == sc[c].id) { found = k; break; }
                PANIC_IF(found < 0, "scan_header: unwrap on None");
                ordered[c] = cf[found];
            }
            out->width = ps->width; out->height = ps->height; out->ncomp = num_components;
            orc_jpeg_decode(env, opts, ps, ordered, num_components, enc, nenc, out);   /* :415 */
            return;                                              /* :417 */
        }
        case 0xdd: /* DRI :424-428 */
            if (!opts->ext_dri) orc_panic(env, opts->strict_ref ? ORC_ERR_REF_PANIC : ORC_ERR_UNSUPPORTED, "got to restart interval def");
            PANIC_IF(data_length < 2 || i + 2 > len, "DRI segment out of bounds");    /* extension: Lr = 4, Ri */
            ps->restart_interval = (vec[i] << 8) | vec[i + 1];
            break;
        case 0xe0: /* APP0 :429-443: reads fixed absolute offsets vec[7], vec[8], vec[10..14], vec[14], vec[15] */
            PANIC_IF(i + 6 > len, "APP0 identifier out of bounds");
            PANIC_IF(len < 16, "APP0 absolute offsets out of bounds");
            break;
        case 0xec: case 0xee: /* APP12 / APP14 :445-450 */
            if (opts->strict_ref) orc_panic(env, ORC_ERR_REF_PANIC, n == 0xec ? "got ApplicationSegment12" : "got ApplicationSegment14");
            break;  /* lenient: skipped */
        default: break;
        }
        i += data_length;                                         /* :454 */
    }
    if (ps->nparts) { orc_decode_parts(env, opts, ps, out); return; }
    orc_panic(env, ORC_ERR_NO_SCAN, "no SOS segment: image_data() is None");
}

int orc_decode(const uint8_t *jpeg, size_t len, const orc_opts *opts, orc_image *out)
{
    static const orc_opts defaults = {0, ORC_LAYOUT_REF, 0, 0, 0, 0, 0};
    const orc_opts *volatile o = opts ? opts : &defaults;
    memset(out, 0, sizeof *out);
    orc_env *env = (orc_env *)calloc(1, sizeof(orc_env));
    if (!env) return ORC_ERR_NOMEM;
    int rc = ORC_OK;
    if (setjmp(env->jb) == 0) {
        orc_parse_and_decode(env, jpeg, len, o, out);
    } else {
        rc = env->code;
        snprintf(out->msg, sizeof out->msg, "%s", env->msg);
        free(out->rgb); out->rgb = NULL;
        if (rc != ORC_OK) for (int c = 0; c < 3; c++) { free(out->coef[c]); out->coef[c] = NULL; out->nblocks[c] = 0; }
    }
    for (int k = 0; k < env->nallocs; k++) free(env->allocs[k]);
    free(env);
    return rc;
}

void orc_free_image(orc_image *img)
{
    if (!img) return;
    free(img->rgb);
    for (int c = 0; c < 3; c++) free(img->coef[c]);
    memset(img, 0, sizeof *img);
}

/* ------------------------------------------------------------------------------------------
 * multi-threaded driver for the CPU baseline (no reference counterpart: the reference is
 * single-threaded; images are independent, so one image per thread is the fair comparison)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *const *jpegs; const size_t *lens; size_t n; const orc_opts *opts; int *status;
    size_t next; uint64_t pixels; pthread_mutex_t mu;
    uint8_t *const *rgb_out; const size_t *rgb_cap;
} orc_many;

static void *orc_many_worker(void *arg)
{
    orc_many *job = (orc_many *)arg;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        size_t i = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (i >= job->n) break;
        orc_image img;
        int rc = orc_decode(job->jpegs[i], job->lens[i], job->opts, &img);
        if (job->status) job->status[i] = rc;
        if (rc == ORC_OK) {
            const size_t bytes = (size_t)img.width * (size_t)img.height * 3;
            if (job->rgb_out && job->rgb_out[i] && job->rgb_cap && job->rgb_cap[i] >= bytes) memcpy(job->rgb_out[i], img.rgb, bytes);
            pthread_mutex_lock(&job->mu);
            job->pixels += (uint64_t)img.width * (uint64_t)img.height;
            pthread_mutex_unlock(&job->mu);
        }
        orc_free_image(&img);
    }
    return NULL;
}

uint64_t orc_decode_many_rgb(const uint8_t *const *jpegs, const size_t *lens, size_t n, const orc_opts *opts, int nthreads,
                             int *status, uint8_t *const *rgb_out, const size_t *rgb_cap)
{
    orc_many job = {jpegs, lens, n, opts, status, 0, 0, PTHREAD_MUTEX_INITIALIZER, rgb_out, rgb_cap};
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    pthread_t th[1024];
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
        if (pthread_create(&th[started], NULL, orc_many_worker, &job) != 0) break;
        started++;
    }
    if (started == 0) orc_many_worker(&job);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    return job.pixels;
}

uint64_t orc_decode_many(const uint8_t *const *jpegs, const size_t *lens, size_t n,
                         const orc_opts *opts, int nthreads, int *status)
{
    return orc_decode_many_rgb(jpegs, lens, n, opts, nthreads, status, NULL, NULL);
}
