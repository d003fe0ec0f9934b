/*
 * mjx_oracle.h -- CPU restatement of martinhath/jpeg-rust's decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.  The product
 * library (jpeg-rust_amd/libmjx.so) never links or calls into this file.
 *
 * Parity pin: the reference ships no tests or golden vectors,
 * and cannot be compiled here (no rustc/cargo).  The oracle is pinned against
 * the four sample JPEGs the reference holds and the known answers recorded in
 * SURVEY.md s4 (coefficient-stream SHA-256s, bit counts, huff_simple0 pixels);
 * see tests/test_oracle_golden.py.  By the project's rule that makes the parity UNPINNED (no reference-owned
 * vector, no run of the reference).  What stands in for a pin is a second, independent restatement in Python /
 * numpy float32 written from the Rust sources (tests/golden/ref_emul.py): the two agree byte for byte on the
 * sample files (coefficients, bits consumed, every RGB byte of the reference's own layout) and on synthetic,
 * 16-bit-DQT, panicking and semantically corrupt inputs.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference/src).
 */
#ifndef MJX_ORACLE_H
#define MJX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC_OK = 0,
    ORC_ERR_REF_PANIC = 1,      /* input on which the reference would panic (bounds, unwrap, assert) */
    ORC_ERR_UNSUPPORTED = 2,    /* marker / feature outside the reference's parser (strict_ref=0 only reports
                                   what cannot be skipped, e.g. DRI, SOF2) */
    ORC_ERR_NO_SCAN = 3,        /* parse reached EOF without SOS: reference returns image_data() == None */
    ORC_ERR_NOMEM = 4
};

enum { ORC_LAYOUT_REF = 0, ORC_LAYOUT_STD = 1 };

typedef struct {
    int strict_ref;   /* 1: APP12/APP14/unknown markers are errors as in jpeg/mod.rs:166-179,445-462;
                         0: skip unknown length-prefixed segments */
    int layout;       /* ORC_LAYOUT_REF: decoder.rs:239-312 bug-for-bug (Q2-Q5);
                         ORC_LAYOUT_STD: standard MCU count / placement / box upsample / edge clip */
    int faithful_cos; /* 1: evaluate cosf() 8192x per block exactly like transform.rs:77-79 (CPU baseline);
                         0: same arithmetic with the 64 distinct cosf() values tabulated (bit-identical, faster) */
    int faithful_huff;/* 1: linear-search per length like huffman.rs:211-227 / 60-76 (CPU baseline);
                         0: same result via per-length first-code table */
    int ext_1bit;     /* NOT reference behaviour: also try 1-bit codes (the reference starts at length 2, huffman.rs:211,
                         SURVEY Q8, and panics on such tables).  Only used to check the GPU path's superset behaviour. */
    int ext_dri;      /* NOT reference behaviour: accept DRI / RSTn restart intervals (T.81 B.2.4.4, E.2.4) instead of panicking
                         like jpeg/mod.rs:424-428: after every Ri MCUs the bit reader moves to the next byte boundary, skips the
                         RSTn marker and the DC predictors start again from 0.  Used to check the GPU path's SURVEY s8(f)-3 row. */
    int ext_multiscan;/* NOT reference behaviour: decode baseline files whose scans carry one component each (non-interleaved
                         order, T.81 A.2.2) instead of stopping after the first scan like jpeg/mod.rs:415-417.  STANDARD
                         layout only.  Used to check the GPU path's SURVEY s8(f)-4 row; pinned by twins of interleaved files
                         (tests/golden/make_multiscan.py). */
} orc_opts;

typedef struct {
    int width, height, ncomp;
    int hs[3], vs[3];            /* sampling factors in scan order */
    uint8_t *rgb;                /* width*height*3, R,G,B, row-major, unpadded (decoder.rs:317-331) */
    int16_t *coef[3];            /* T0 stream per component (scan order): blocks in decode order,
                                    64 x i16 zig-zag order, after DC prediction, before dequant */
    size_t nblocks[3];
    size_t mcus_read;            /* Q2 count (REF) or standard count (STD) */
    size_t bits_used;            /* total bits consumed by the entropy decoder */
    char msg[160];               /* panic / error text */
} orc_image;

int  orc_decode(const uint8_t *jpeg, size_t len, const orc_opts *opts, orc_image *out);
void orc_free_image(orc_image *img);

/* transform.rs:55-87 on one 8x8 block (natural order in, natural order out). */
void orc_idct_ref(const float in[64], float out[64], int faithful_cos);
/* decoder.rs:392-402 + 382-390 */
void orc_ycbcr_to_rgb(float y, float cb, float cr, uint8_t rgb[3]);
/* decoder.rs:382-390 */
uint8_t orc_f32_to_u8(float n);

/* Decode n independent JPEGs with nthreads worker threads (one image per thread at a time).
 * Used by bench.py's cpu_baseline leg.  status[i] receives the per-image result code.
 * Returns total pixels decoded successfully. */
uint64_t orc_decode_many(const uint8_t *const *jpegs, const size_t *lens, size_t n,
                         const orc_opts *opts, int nthreads, int *status);

/* The same, and image i's RGB (width*height*3 bytes) is copied to rgb_out[i] when that pointer is non-NULL and rgb_cap[i]
 * holds it: bench.py's parity gate compares the pictures the CPU baseline decodes anyway with the GPU's. */
uint64_t orc_decode_many_rgb(const uint8_t *const *jpegs, const size_t *lens, size_t n, const orc_opts *opts, int nthreads,
                             int *status, uint8_t *const *rgb_out, const size_t *rgb_cap);

#ifdef __cplusplus
}
#endif
#endif
