/*
 * mjx.h -- C ABI of the MI355X-native baseline-JPEG decode path.
 *
 * Drop-in boundary for martinhath/jpeg-rust's decode(&[u8]) -> RGB surface
 * (SURVEY.md s8(b)).  Every entry point cites the reference interface it replaces
 * (paths relative to /root/reference/src).  Plain pointers and sizes only: this is what a
 * Rust `extern "C"` block binds (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Nothing here ever panics/aborts across the ABI: the reference's panics (jpeg/mod.rs Q12)
 * become status codes.  The GPU entry points fail with MJX_ERR_DEVICE when no HIP device or
 * kernel image is available -- there is no CPU fallback.
 */
#ifndef MJX_H
#define MJX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------ */
enum {
    MJX_OK = 0,
    MJX_ERR_TRUNCATED = 1,          /* ran off the end of the file while parsing (ref: slice-index panic) */
    MJX_ERR_UNSUPPORTED_MARKER = 2, /* strict_ref: marker outside jpeg/mod.rs:166-179 (incl. APP12/APP14, :445-450) */
    MJX_ERR_DRI_UNSUPPORTED = 3,    /* jpeg/mod.rs:424-428 panics on DRI */
    MJX_ERR_BAD_HUFFMAN = 4,        /* invalid DHT, or no code matched during decode (huffman.rs:156,162) */
    MJX_ERR_REF_PANIC = 5,          /* other input on which the reference panics (asserts, bad DQT precision ...) */
    MJX_ERR_DEVICE = 6,             /* HIP error / no device / kernels missing */
    MJX_ERR_UNSUPPORTED_FORMAT = 7, /* non-baseline SOF, ncomp not in {1,3} (decoder.rs:328-330), sampling > 2 */
    MJX_ERR_NO_SCAN = 8,            /* EOF without SOS: the reference returns image_data() == None */
    MJX_ERR_INVALID_ARG = 9,
    MJX_ERR_NOMEM = 10,
    MJX_ERR_MISSING_TABLE = 11      /* scan references a DQT/DHT slot that was never defined (decoder.rs:154-160,222) */
};

enum {
    MJX_LAYOUT_STANDARD = 0,  /* standard MCU count, block placement, box chroma replication, edge clipping */
    MJX_LAYOUT_REF_COMPAT = 1 /* bug-for-bug placement of decoder.rs:239-312 */
};

enum mjx_destuff {
    MJX_DESTUFF_AUTO = 0,
    MJX_DESTUFF_DEVICE = 1,
    MJX_DESTUFF_HOST = 2
};

typedef struct mjx_opts {
    uint8_t strict_ref;   /* 1: unknown / APP12 / APP14 markers are errors like the reference; 0: skip them */
    uint8_t layout;       /* MJX_LAYOUT_* */
    uint8_t keep_coefs;   /* 1: keep the whole batch's coefficient stream resident (T0 checks, stage-B-only sweeps) */
    uint8_t device_destuff;/* MJX_DESTUFF_*: who removes the FF00 stuffing of jpeg/mod.rs:371-385.
                             MJX_DESTUFF_DEVICE (1): the host does not touch the entropy-coded bytes -- mjx_parse copies them as
                             they are (desc.scan_is_stuffed = 1), and the FF00 -> FF compaction, the search for RSTn markers
                             and the scan's length are the GPU's at upload (mjx_batch_create, mjx_decode_batch, the pool).
                             MJX_DESTUFF_HOST (2): the byte pass on the host.  MJX_DESTUFF_AUTO (0): the host pass in
                             mjx_parse and mjx_decode; in mjx_decode_batch and the pool the GPU for lists of 64 MB and more
                             (MJX_AUTO_DESTUFF_MB; it is the faster of the two there), the host below.  The picture is the
                             same.  Never on the GPU with strict_ref (the byte pass stays, for the reference's unguarded read
                             behind a last FF) nor for multi-scan files (their scans are cut apart on the host). */
    uint32_t chunk_images;/* images per kernel chunk; 0 = library default */
} mjx_opts;

/* ---- inner seam: what jpeg/mod.rs:388-415 hands to JPEGDecoder -------------------------- */
typedef struct mjx_comp {          /* decoder.rs:39-52 JPEGDecoderComponentFields */
    uint8_t id, h, v, tq, td, ta;
} mjx_comp;

typedef struct mjx_hufftab {       /* the two slices given to HuffmanTable::from_size_data_tables, huffman.rs:37 */
    uint8_t bits[16];
    uint8_t vals[256];
} mjx_hufftab;

/* One scan of a multi-scan baseline file: a single component (non-interleaved order, T.81 A.2.2) or an interleaved
 * subset of the frame's components ("0; 1 2;", the separate-luma-and-chroma script of libjpeg's wizard.txt) -- beyond the
 * reference, which stops after the first SOS (jpeg/mod.rs:415-417).  SURVEY s8(f)-4. */
typedef struct mjx_scan_part {
    const uint8_t *scan;           /* this scan's entropy-coded segment, de-stuffed, RSTn taken out */
    size_t scan_len;
    uint8_t ncomp;                 /* components in this scan: 1 or 2 */
    uint8_t comp[3];               /* their indices into mjx_scan_desc.comp, scan order */
    uint16_t restart_interval;     /* MCUs (single component: blocks) per restart interval as defined when the SOS was read */
    uint32_t n_restart;
    const uint32_t *restart_offsets;
    mjx_hufftab dc[3], ac[3];      /* per component of the scan: the tables it selects, as defined when the SOS was read */
} mjx_scan_part;

typedef struct mjx_scan_desc {
    const uint8_t *scan;           /* bytes after the SOS header to end of file, de-stuffed (jpeg/mod.rs:371-385) unless
                                      scan_is_stuffed */
    size_t scan_len;
    uint16_t width, height;        /* .dimensions() decoder.rs:66 */
    uint8_t ncomp;                 /* 1 or 3 */
    mjx_comp comp[3];              /* scan order (decoder.rs:141-150) */
    uint16_t qt[4][64];            /* zig-zag (file) order as in DQT, jpeg/mod.rs:236-256; .quantization_table() */
    uint8_t qt_present;            /* bit i = slot i defined */
    mjx_hufftab dc[4], ac[4];      /* .huffman_dc_tables() / .huffman_ac_tables() decoder.rs:71-77 */
    uint8_t dc_present, ac_present;
    uint8_t scan_is_stuffed;       /* 1: `scan` still holds FF00 pairs and RSTn markers, scan_len is the stuffed length and
                                      restart_offsets is empty: the device de-stuffs, finds the markers and the length */
    /* Restart intervals (T.81 B.2.4.4) -- beyond the reference, which panics on DRI (jpeg/mod.rs:424-428; strict_ref keeps
       that).  mjx_parse removes the RSTn markers from `scan` and lists where each further interval begins. */
    uint16_t restart_interval;     /* MCUs per interval, 0 = none */
    uint32_t n_restart;            /* entries of restart_offsets */
    const uint32_t *restart_offsets; /* byte offset in `scan` of the first byte of interval 1, 2, ... (interval 0 starts at 0) */
    /* Multi-scan files: n_parts > 0 means `scan` is NULL, `comp` lists the frame's components in frame order and every
       component is carried by exactly one of `parts`; the dc / ac slots above are not used. */
    uint8_t n_parts;
    const mjx_scan_part *parts;
    void *owner_;                  /* internal: storage behind `scan` / `parts` when filled by mjx_parse */
} mjx_scan_desc;

/* ---- outer surface ------------------------------------------------------------------------ */
typedef struct mjx_image {         /* JPEGImage: width() mod.rs:467, height() :471, image_data() :475 */
    uint32_t width, height;
    uint8_t *rgb;                  /* width*height*3 bytes, R,G,B, row-major, unpadded; NULL on error */
} mjx_image;

/* JPEGImage::parse, jpeg/mod.rs:202 -- host-side marker walk only (no GPU): fills the POD the decoder needs
 * exactly as mod.rs:228-362 does, de-stuffs the scan (mod.rs:371-385).  Release with mjx_free_scan. */
int mjx_parse(const uint8_t *jpeg, size_t len, const mjx_opts *opts, mjx_scan_desc *out);
void mjx_free_scan(mjx_scan_desc *desc);

/* Host-only check of a parsed scan: tables present and valid, geometry supported and, for MJX_LAYOUT_REF_COMPAT, not
 * one of the inputs on which the reference's placement code panics (decoder.rs:300-303, 370-371; SURVEY Q5 ->
 * MJX_ERR_REF_PANIC).  The same check mjx_batch_create applies per image. */
int mjx_validate(const mjx_scan_desc *desc, const mjx_opts *opts);

/* JPEGImage::parse + image_data() in one call on device `0` (main.rs:31-36 usage): parse, upload, decode on the
 * GPU, copy the RGB back.  Release with mjx_free_image. */
int mjx_decode(const uint8_t *jpeg, size_t len, const mjx_opts *opts, mjx_image *out);
void mjx_free_image(mjx_image *img);

/* ---- batch / device-resident API (one context per process per GPU) ------------------------ */
typedef struct mjx_ctx mjx_ctx;
typedef struct mjx_batch mjx_batch;

int mjx_ctx_create(int device, mjx_ctx **out);
void mjx_ctx_destroy(mjx_ctx *ctx);
/* 1: record HIP events around every kernel class of mjx_batch_decode (read with mjx_batch_kernel_ms) */
int mjx_ctx_set_profiling(mjx_ctx *ctx, int enable);
/* 1: batches of this context are always cut for throughput (512-byte subsequences).  By default a batch too small to fill the
 * device is cut into shorter subsequences, which shortens its latency; a caller that builds a small batch only to replicate
 * it on the device (mjx_batch_tile keeps the base's cut) switches that off, so that the large batch is cut like one that
 * was created at its size.  Also the first mjx_decode_batch / mjx_batch_create on a fresh context allocates its device
 * block and pinned buffers: expect that call to take several times as long as the ones after it (0.5 s against 50 ms for
 * 2048 4K files). */
int mjx_ctx_set_throughput_plan(mjx_ctx *ctx, int enable);
/* NUMA node of the context's GPU (from the PCI device's numa_node in sysfs), -1 when the platform does not say. */
int mjx_ctx_numa_node(const mjx_ctx *ctx);
/* Processors this process may use: the affinity mask, capped by the cgroup CPU quota. */
unsigned mjx_host_processors(void);

/* JPEGDecoder::new(..).frame_header(..).scan_header(..).dimensions(..) + table setters for n images
 * (decoder.rs:55-152): validates, builds decode tables, packs and uploads the scans; device buffers for the
 * outputs are allocated here.  status[i] (optional) receives the per-image code; images with an error are
 * skipped by decode and produce no output.  Inputs are only borrowed for the duration of the call. */
int mjx_batch_create(mjx_ctx *ctx, const mjx_scan_desc *descs, size_t n, const mjx_opts *opts,
                     mjx_batch **out, int *status);
void mjx_batch_free(mjx_batch *b);

/* Replicate the uploaded images `times`x on the device (image i*n+k is a byte copy of image k): builds the
 * large synthetic batches of BASELINE.json configs 4/5 from n unique images without re-uploading. */
int mjx_batch_tile(mjx_ctx *ctx, const mjx_batch *src, size_t times, mjx_batch **out);

/* JPEGDecoder::decode, decoder.rs:162 -- the hot path.  Enqueues every kernel for the whole batch on the
 * context's stream and returns; inputs and outputs stay in HBM.  stages: MJX_STAGE_* mask (0 = all). */
enum { MJX_STAGE_ENTROPY = 1, MJX_STAGE_PIXELS = 2, MJX_STAGE_ALL = 3 };
int mjx_batch_decode(mjx_batch *b, unsigned stages);
/* Block until the stream is idle; fold device-side error flags into the per-image status. */
int mjx_batch_wait(mjx_batch *b);

size_t mjx_batch_size(const mjx_batch *b);
int mjx_batch_status(const mjx_batch *b, size_t i);
/* geometry of image i: width, height, number of blocks per MCU, MCUs decoded */
int mjx_batch_image_info(const mjx_batch *b, size_t i, uint32_t *width, uint32_t *height, uint32_t *blocks_per_mcu,
                         uint32_t *mcus);
/* device pointer + byte size of image i's packed RGB (valid until mjx_batch_free) */
int mjx_batch_rgb_device(const mjx_batch *b, size_t i, void **dev_ptr, size_t *bytes);
/* copy image i's RGB to host memory (width*height*3 bytes) */
int mjx_batch_copy_rgb(mjx_batch *b, size_t i, uint8_t *host_rgb);
/* T0 stream of image i (needs keep_coefs, or i inside the last decoded chunk): blocks in decode (MCU-interleaved)
 * order, 64 x i16 zig-zag, DC prediction applied, before dequantisation.  `cap_blocks` = capacity of host buffer.
 * A multi-scan picture needs keep_coefs: without it its coefficients only exist scan by scan (stage B reads them from the
 * scans' streams) and the call returns MJX_ERR_INVALID_ARG. */
int mjx_batch_copy_coefs(mjx_batch *b, size_t i, int16_t *host_coefs, size_t cap_blocks, size_t *nblocks);

/* Verification helper (bench.py's parity gate, batch-scale tests): compares the decoded RGB of n pairs of pictures on the
 * device -- picture ia[k] of batch a with picture ib[k] of batch b (same device; a == b is fine) -- without copying them
 * to the host.  Per pair: the largest absolute difference of a byte and the number of differing bytes; a pair whose
 * pictures differ in size, or either of which failed to decode, reports max_abs_diff = 0xffffffff. */
int mjx_batch_compare_rgb(mjx_batch *a, const size_t *ia, mjx_batch *b, const size_t *ib, size_t n,
                          uint32_t *max_abs_diff, uint64_t *n_diff);

/* bytes used to compute roofline figures: sum of de-stuffed entropy bytes and of RGB bytes over valid images */
int mjx_batch_bytes(const mjx_batch *b, uint64_t *scan_bytes, uint64_t *rgb_bytes, uint64_t *coef_bytes,
                    uint64_t *pixels);

/* work units of the batch (valid images): subsequences the entropy stage decodes in parallel (512..640 bytes of scan each,
 * chosen per image), coefficient blocks, and the kernel chunks the batch is processed in (one launch per kernel class and chunk) */
int mjx_batch_geometry(const mjx_batch *b, uint64_t *subsequences, uint64_t *blocks, uint64_t *chunks);

/* Runs of a chunk, since the batch was created, whose synchronisation rounds had not converged when the rest of the entropy stage
 * was enqueued behind them: the chunk's pictures were skipped in that run -- and runs in which a picture left the single-decode
 * path on the device (it is skipped by the rest of that run too).  mjx_batch_wait examines and repairs the LAST decode
 * only, so a caller that enqueues several mjx_batch_decode calls before one wait (a throughput measurement) reads here whether
 * every one of them did the whole work.  Synchronises the batch's streams. */
int mjx_batch_unconverged_runs(const mjx_batch *b, uint64_t *runs);

/* accumulated kernel time (ms) and launch count per kernel class since the last reset (profiling enabled) */
enum {
    MJX_K_GATHER = 0,     /* multi-scan pictures only: component streams -> the picture's stream (k_planar_*) */
    MJX_K_HUFF_SYNC = 1,  /* speculative decode + intra-workgroup synchronisation */
    MJX_K_HUFF_FIX = 2,   /* inter-workgroup synchronisation passes */
    MJX_K_HUFF_SCAN = 3,  /* block-count prefix sums */
    MJX_K_HUFF_WRITE = 4, /* final decode writing coefficients */
    MJX_K_DC_SCAN = 5,    /* DC prediction prefix sums */
    MJX_K_IDCT_COLOR = 6, /* dequant + IDCT + upsample + colour + RGB store */
    MJX_K_UPLOAD = 7,     /* upload time, once per batch, not part of a decode: de-stuffing on the device (opts.device_destuff)
                             and the pass that lays the scans out lane-interleaved (k_scan_interleave) */
    MJX_K_HUFF_EMIT = 8,  /* single decode (pictures of one scan without restart intervals): the first decode, which emits -- instead of
                             MJX_K_HUFF_SYNC and the decode of MJX_K_HUFF_WRITE */
    MJX_K_HUFF_PREFIX = 9,/* ... the prefixes of the subsequences whose entry state was wrong, and block words -> DC differences + tile offsets */
    MJX_K_COUNT = 10
};
int mjx_batch_kernel_ms(mjx_batch *b, double ms[MJX_K_COUNT], uint64_t launches[MJX_K_COUNT], int reset);

/* SURVEY s8(b) convenience form: create + decode + wait; rgb_dev[i] receives device pointers owned by *out. */
int mjx_decode_scans(mjx_ctx *ctx, const mjx_scan_desc *descs, size_t n, const mjx_opts *opts,
                     uint8_t **rgb_dev, int *status, mjx_batch **out);

/* The outer surface for a batch of files: JPEGImage::parse of jpeg/mod.rs:202 for n files at once,
 * pipelined -- the list is cut into groups of compressed data (12 MB first, doubling up to 96 MB, 192 MB for long lists);
 * `threads` host threads (0 = half the processors the process may use -- affinity mask and cgroup quota counted --,
 * between 4 and 32) walk the markers and de-stuff group after group into pinned memory, every
 * group goes up in one DMA transfer on an upload stream of its own, and its kernels start behind that transfer's event,
 * so parsing, transfer and decode of successive groups overlap.  status[i] receives the parse / plan / decode status of
 * file i, rgb_dev[i] a device pointer owned by *out (NULL on failure); mjx_batch_image_info(*out, i, ...) gives the
 * dimensions.  *out is a directory of the groups' batches: every per-picture accessor, mjx_batch_decode / _wait / _bytes /
 * _geometry / _kernel_ms / _compare_rgb work on it, mjx_batch_tile does not.  Calls on one context are serialised (the
 * pinned arena belongs to the context).  Release with mjx_batch_free. */
int mjx_decode_batch(mjx_ctx *ctx, const uint8_t *const *jpegs, const size_t *lens, size_t n, const mjx_opts *opts,
                     unsigned threads, uint8_t **rgb_dev, int *status, mjx_batch **out);

/* ---- multi-GPU front: one context + one host thread + one work queue per device, no collective ----------
 * Pictures are independent (decoder.rs:162-343 touches only `self`), so a list of files shards over the GPUs of a node
 * without any exchange: every slot decodes its share with the pipelined mjx_decode_batch on its own device, outputs stay
 * where they were produced.  Files are dealt to the slots by compressed bytes (largest file first, each to the slot with the
 * fewest bytes so far, the lowest slot on a tie: i mod N for a list of equal files, level queues for a skewed one) or round
 * robin (mjx_pool_set_deal).  `devices` may name a device more than once (two slots on one GPU).  One mjx_pool_decode_batch
 * call at a time per pool.  A slot whose device fails fails its own files (their status is the slot's error, the call
 * returns it); the other slots' results stay valid.
 * Host side: the slots share the host.  With threads_per_device == 0 every slot takes max(2, P / (2 N)) parse threads, P =
 * mjx_host_processors(), so that N slots together stay within the processors the process may use (mjx_decode_batch alone
 * takes P / 2); and a slot's host thread -- with it the parse threads it starts and the pinned arena it allocates -- is bound
 * to the processors of its GPU's NUMA node when the platform names one and the process may run there (MJX_POOL_NUMA=0: no). */
typedef struct mjx_pool mjx_pool;
typedef struct mjx_pool_result mjx_pool_result;
int mjx_pool_create(const int *devices, size_t n_devices, mjx_pool **out);
void mjx_pool_destroy(mjx_pool *pool);
size_t mjx_pool_devices(const mjx_pool *pool);
int mjx_pool_device(const mjx_pool *pool, size_t slot);          /* HIP device of a slot, -1 if out of range */
enum { MJX_POOL_DEAL_BY_BYTES = 0, MJX_POOL_DEAL_ROUND_ROBIN = 1 };
int mjx_pool_set_deal(mjx_pool *pool, int deal);
/* slot_of[i] / rgb_dev[i] / status[i] (each optional, n entries): the slot that decoded file i, its device pointer (owned
 * by *out, NULL on failure) and status.  Release with mjx_pool_result_free. */
int mjx_pool_decode_batch(mjx_pool *pool, const uint8_t *const *jpegs, const size_t *lens, size_t n, const mjx_opts *opts,
                          unsigned threads_per_device, int *slot_of, uint8_t **rgb_dev, int *status, mjx_pool_result **out);
/* file i of the call -> its slot, the slot's batch and the picture's index inside it (for mjx_batch_image_info,
 * mjx_batch_copy_rgb, ...) */
int mjx_pool_result_locate(const mjx_pool_result *r, size_t i, size_t *slot, mjx_batch **batch, size_t *index);
/* host side of the call, per slot: parse threads the slot's mjx_decode_batch ran with (0: the slot had no file) and the NUMA
 * node its host thread was bound to (-1: not bound) */
int mjx_pool_result_host(const mjx_pool_result *r, size_t slot, unsigned *threads, int *numa_node);
/* ... and the wall clock of the slot's own mjx_decode_batch (host parse + uploads + kernels, in milliseconds; 0: no file): which
 * queue the call waited for */
int mjx_pool_result_slot_ms(const mjx_pool_result *r, size_t slot, double *ms);
void mjx_pool_result_free(mjx_pool_result *r);

const char *mjx_strerror(int code);
/* library build info: "mjx <version> gfx950 subseq=<bits> ..." */
const char *mjx_version(void);

#ifdef __cplusplus
}
#endif
#endif
