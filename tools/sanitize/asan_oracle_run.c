/* The CPU oracle (oracle/mjx_oracle.c, test infrastructure) under AddressSanitizer / UBSan: every fixture decoded in both layouts
 * with every extension switch, then mutated copies (bytes overwritten anywhere, truncations) -- the reference leans on Rust's
 * bounds checks (src/jpeg/decoder.rs:370-371, src/jpeg/huffman.rs:240-247: a panic, never a wild read); its C restatement must
 * report such inputs as ORC_ERR_REF_PANIC / ORC_ERR_UNSUPPORTED without touching memory it does not own.  Built and run by
 * tests/test_sanitizers.py:
 *   gcc -std=c11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off -Ioracle \
 *       tools/sanitize/asan_oracle_run.c oracle/mjx_oracle.c -lm -lpthread -o /tmp/asan_oracle_run
 *   /tmp/asan_oracle_run <mutations per file> tests/data/... tests/golden/pil/...        */
#include "mjx_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned long long rng_state = 0x9e3779b97f4a7c15ull;
static unsigned long long rng(void)
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

static int decode_once(const unsigned char *data, size_t len, int mode, long *ok)
{
    orc_opts o;
    memset(&o, 0, sizeof o);
    o.layout = mode & 1 ? ORC_LAYOUT_STD : ORC_LAYOUT_REF;
    o.strict_ref = (mode >> 1) & 1;
    o.ext_1bit = (mode >> 2) & 1;
    o.ext_dri = (mode >> 2) & 1;
    o.ext_multiscan = ((mode >> 2) & 1) && (mode & 1);
    /* exact-size heap copy: a read past the end of the file is caught */
    unsigned char *heap = (unsigned char *)malloc(len ? len : 1);
    if (!heap) return 1;
    memcpy(heap, data, len);
    orc_image img;
    memset(&img, 0, sizeof img);
    const int rc = orc_decode(heap, len, &o, &img);
    if (rc == ORC_OK) {
        /* touch what the caller would read: the whole picture and every coefficient */
        unsigned long long acc = 0;
        for (size_t i = 0; i < (size_t)img.width * img.height * 3; i++) acc += img.rgb[i];
        for (int c = 0; c < img.ncomp && c < 3; c++)
            for (size_t i = 0; i < img.nblocks[c] * 64; i++) acc += (unsigned short)img.coef[c][i];
        if (acc == 0xffffffffffffffffull) puts("");
        (*ok)++;
    }
    orc_free_image(&img);
    free(heap);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <mutations per file> files...\n", argv[0]); return 2; }
    const int muts = atoi(argv[1]);
    long runs = 0, ok = 0;
    for (int a = 2; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) continue;
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (n < 4 || n > (1 << 20)) { fclose(f); continue; }
        unsigned char *base = (unsigned char *)malloc((size_t)n), *b = (unsigned char *)malloc((size_t)n);
        if (!base || !b || fread(base, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(base); free(b); continue; }
        fclose(f);
        for (int mode = 0; mode < 8; mode++) { decode_once(base, (size_t)n, mode, &ok); runs++; }
        for (int k = 0; k < muts; k++) {
            memcpy(b, base, (size_t)n);
            size_t len = (size_t)n;
            const int m = 1 + (int)(rng() % 6);
            static const unsigned char pick[] = {0xff, 0xd0, 0xd9, 0xda, 0xc4, 0xc0, 0xdb, 0xdd, 0x00, 0x11, 0x22};
            for (int j = 0; j < m; j++) b[rng() % len] = (rng() & 1) ? pick[rng() % sizeof pick] : (unsigned char)rng();
            if (rng() % 4 == 0) len = 1 + rng() % len;
            decode_once(b, len, (int)(rng() % 8), &ok);
            runs++;
        }
        free(base);
        free(b);
    }
    /* the threaded front (bench.py's cpu_baseline leg): shared nothing, but prove it */
    printf("asan oracle run: %ld decodes, %ld pictures\n", runs, ok);
    return 0;
}
