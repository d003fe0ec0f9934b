// mjx_api.hip -- host side of the C ABI: context, batch planning / upload, chunked launch sequence.
//
// Replaces the JPEGDecoder builder and decode() driver (reference src/jpeg/decoder.rs:55-162) and the
// hand-off in JPEGImage::parse (src/jpeg/mod.rs:388-417).  One context = one HIP device + one stream; a batch owns
// every device buffer of its images; mjx_batch_decode only enqueues kernels (no allocation, no host sync).
#include <hip/hip_runtime.h>
#include <cctype>
#include <sched.h>

#include "mjx.h"
#include "mjx_kernels.h"
#include "mjx_plan.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace mjx;

#define HIPOK(expr)                                   \
    do {                                              \
        if ((expr) != hipSuccess) {                   \
            (void)hipGetLastError();                  \
            return MJX_ERR_DEVICE;                    \
        }                                             \
    } while (0)

// Nothing may unwind through the C ABI (include/mjx.h: the reference's panics become status codes; so do ours): every
// entry point that allocates runs its body inside guarded().
template <class F>
static int guarded(F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return MJX_ERR_NOMEM;
    } catch (const std::length_error &) {
        return MJX_ERR_NOMEM;
    } catch (...) {
        return MJX_ERR_DEVICE;
    }
}

struct mjx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // stage B runs here, the entropy stage on `stream` (MJX_STREAMS=1: everything on `stream`), see run_chunk
    hipStream_t stream3 = nullptr;  // MJX_STREAMS=3: the entropy stage of the odd chunks (they have their own scratch set), beside that of the even ones
    std::vector<std::pair<uint32_t *, size_t>> pinned_cache;   // small pinned blocks (a batch's mirror of its round counts) kept for
                                                               // the next batch: hipHostMalloc + hipHostFree were 0.2 ms of a one-shot decode
    bool dc_one_pass = true;        // DC prediction of the common MCU shapes in one pass, k_dc_scan_t (MJX_DC_ONE_PASS=0: two passes)
    // Batches too small to fill the device (DESIGN s11), see build_batch:
    uint32_t merge_loop_max = 64;   // chunks with at most this many merge workgroups (of the 768 the device holds) run their rounds in
                                    // one launch, k_huff_merge_loop (MJX_MERGE_LOOP=n, 0 = never)
    bool loop_fault = false;        // test knob (MJX_LOOP_FAULT=1): that launch waits for a workgroup that does not exist and must give up
    bool dc_fault = false;          // test knob (MJX_DC_FAULT=1): a workgroup of k_dc_scan_t never publishes its sums; the ones behind it must give up
    uint64_t latency_nsub = 32768;  // batches of at most this many 512-byte subsequences (16 MB of scans: 64 of the 1024 workgroup slots of
                                    // k_huff_spec) are cut into shorter subsequences when they fit the loop kernel (MJX_LATENCY_NSUB, 0 = never),
    bool linear_stream = false;                                 // MJX_STREAM_LINEAR=1: the packed stream for every picture (the layout of multi-scan pictures; A/B, tests)
    uint32_t latency_sub_bits = 512;                            // ... this many bits at least (MJX_LATENCY_SUB_BITS)
    uint64_t medium_nsub = 73728;   // batches of up to this many 512-byte subsequences' worth of scan (36 MB, ~32 4K pictures) that are too large for
                                    // the loop kernel get 256-byte subsequences with launch-per-round merges (MJX_MEDIUM_NSUB, 0 = never):
                                    // 12 / 16 / 24 / 32 4K pictures 1.34 / 1.39 / 1.47 / 1.55 -> 1.08 / 1.12 / 1.20 / 1.33 ms; 64 pictures: no gain
    hipStream_t upload = nullptr;   // H2D of the compressed scans + the upload-time kernels (de-stuffing, interleaving): a stream of
                                    // its own, so that the upload of one group of files overlaps the decode of the group before
    // Device blocks of released batches, kept for the next ones: giving tens of gigabytes back to the driver and asking for
    // them again cost 2.6 s per mjx_decode_batch call of 2048 4K files (and hipFree synchronises the device).  At most
    // `cache_limit` bytes are kept (MJX_CACHE_GB, default 96; 0 = off); mjx_ctx_destroy releases them.
    std::mutex cache_mu;
    std::vector<std::pair<uint8_t *, size_t>> cache;
    size_t cache_bytes = 0, cache_limit = size_t(96) << 30;
    // mjx_batch_create: the host lays the de-stuffed scans out lane-interleaved in these two pinned blocks, alternately, and each
    // goes up as one transfer into the scan pool (build_batch, host_interleave); kept between calls
    uint8_t *stage_pin[2] = {nullptr, nullptr};
    size_t stage_pin_cap[2] = {0, 0};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    bool destuff_direct = true;     // scans de-stuffed on the device (no restart intervals): the compaction writes the lane-interleaved region
                                    // itself (MJX_DESTUFF_DIRECT=0: a linear copy + k_scan_interleave)
    bool host_interleave = true;    // MJX_HOST_INTERLEAVE=0: linear upload + k_scan_interleave, as the groups of mjx_decode_batch do
    // Single decode (round 5, DESIGN s3.1): pictures of one scan without restart intervals, cut into subsequences of at least
    // emit_min_sub_bits bits (the long ones), are decoded ONCE by an emitting pass (k_huff_emit) whose lanes warm up over the last emit_warm_bits
    // bits of the subsequence in front of their own and record a checkpoint every emit_cp_bits bits; MJX_SINGLE_DECODE=0: every
    // picture takes the two-pass kernels (k_huff_spec ... k_huff_write).
    uint8_t *rgb_pin = nullptr;    // pinned block mjx_batch_copy_rgb copies through (8 MB, allocated on first use)
    std::mutex rgb_pin_mu;
    bool single_decode = true;
    bool merge_memo = true;        // two generations per subsequence in the merge rounds (MJX_MERGE_MEMO=0: one, as before round 5)
    bool planar_direct = true;     // multi-scan pictures: stage B reads the scans' streams (MJX_PLANAR_DIRECT=0: always through the gather kernels)
    bool emit_merge_listed = true;  // MJX_EMIT_MERGE_LISTED=0: the first merge round of such pictures runs its head slices in place, as for the others
    // (emit_min_sub_bits = the long subsequences of scans of 0.79 MB and more, mjx_huff.h: with the 4096 .. 5120-bit subsequences of
    // shorter scans the warm-up is half a subsequence -- 4096 x 1080p 15.3-16.4 ms per step at 2048 / 1024 / 512 bits of warm-up
    // against 14.9-15.1 on the two-pass kernels, 2048 x 4K at quality 50 24.1-24.7 against 23.8 --, so those keep the two passes)
    uint32_t emit_cp_bits = 1024, emit_warm_bits = 2048, emit_min_sub_bits = uint32_t(kLongSubseqBits), emit_head = kEmitHeadGroups;
    uint8_t *pin_small = nullptr;   // pinned block for the host mirrors of the groups' small pools (mjx_decode_batch)
    size_t pin_small_cap = 0;
    std::mutex batch_mu;            // mjx_decode_batch: one call at a time per context (the pinned arena is shared state)
    int nstreams = 2;
    bool profiling = false;
    bool upload_kernels_apart = true;   // mjx_decode_batch: upload-time kernels on a decode stream, the upload stream carries transfers only (MJX_UPLOAD_APART=0: all on the upload stream)
    bool group_alt_stream = true;       // ... and every second group's entropy stage on the third stream (MJX_GROUP_ALT=0: all on the first)
    bool throughput_plan = false;   // mjx_ctx_set_throughput_plan: never cut a batch into short subsequences (a base that will be tiled)
    // mjx_decode_batch: pinned arena the files of a call are de-stuffed into; kept between calls (fresh pages cost ~0.35 us
    // per KB to fault in and unmap again -- four times the parsing itself), released with the context
    uint8_t *parse_arena = nullptr;
    size_t parse_arena_cap = 0;
    int fix_passes = 6;            // synchronisation rounds enqueued up front (the last one must re-decode nothing; rounds
                                   // behind an empty one leave at once)
    // Extra dynamic LDS per entropy kernel = occupancy caps for experiments (MJX_SPEC/MERGE/WRITE_LDS_PAD); 0 in production.
    size_t spec_lds_pad = 0, merge_lds_pad = 0, write_lds_pad = 0, idct_lds_pad = 0;
    size_t configured_huff = 0, configured_idct = 0;   // dynamic-LDS limits the kernels were last configured for
};

namespace {

struct ImageInfo {
    int status = MJX_OK;
    uint32_t width = 0, height = 0, bpm = 0, nmcu = 0;
    uint64_t nblocks = 0;
    uint64_t rgb_off = 0, rgb_bytes = 0;
    uint64_t coef_off = 0;         // blocks, inside the per-block arrays of its chunk (or of the batch with keep_coefs)
    uint64_t ent_off = 0, ent_cap = 0;   // region of the compact coefficient stream (entries)
    uint32_t ent_rows = 0, ent_hdr = 0;  // > 0: quad-interleaved (DevImage::ent_rows, ent_hdr)
    bool emit = false;             // single decode: the picture's first decode emits (DevImage::emit)
    uint32_t emit_head = 0;
    uint32_t tile_off = 0, ntiles = 0, tile_blocks = 0;
    uint64_t scan_len = 0;
    uint32_t chunk = 0;
    uint32_t role = 0;             // 0 ordinary picture; multi-scan files: 1 = one scan (internal), 2 = the picture
    uint32_t nparts = 0;           // role 2: scans in front of it
    bool planar = false;           // role 2: read without the gather (no stream of its own: mjx_batch_copy_coefs has nothing to expand)
};

struct Chunk {
    size_t first = 0, count = 0;
    uint32_t nsub = 0;             // subsequences in the chunk
    uint64_t scan_bytes = 0;       // scan bytes in the chunk (what closes it, plan_chunks)
    uint64_t blocks = 0;           // coefficient blocks in the chunk
    uint64_t coef_base = 0;        // first block of the chunk inside the per-block arrays (keep_coefs) or 0
    uint64_t entries = 0, ent_base = 0;   // capacity of the chunk's stream regions; first entry (keep_coefs) or 0
    uint32_t tiles = 0, tile_base = 0;    // tile offsets (+1 sentinel per image)
    uint32_t loop_participants = 0;     // > 0: the merge rounds run as one launch (k_huff_merge_loop) with this many workgroups
    uint32_t lut2_cap = 0;         // entries of the largest second table set (pair parts, the counting passes)
    uint32_t max_wg = 0, merge_wgs = 0, max_tiles = 0, lut_cap = 0, max_tile_blocks = 0, mode_mask = 0, layout_mask = 0, max_segs = 0, bpm_mask = 0, max_restart_segs = 0;
    uint64_t plane_words = 0;      // REF_COMPAT scratch of the chunk
    uint32_t max_pixel_wgs = 0;
    uint32_t min_sub_bits = 0xffffffffu;   // shortest subsequence length among its scans (chunk_fix_passes)
    uint32_t max_nsub = 0;                 // subsequences of its longest scan
    uint32_t wg = 128;                     // lanes of its k_huff_spec / k_huff_write workgroups: the largest its scans were cut for (ImagePlan::wg_lanes)
    int learned_passes = 0;        // rounds a repair in mjx_batch_wait found this chunk to need: later decodes of the batch enqueue them up front
    bool has_gather = false;       // holds multi-scan pictures (k_planar_gather runs)
    bool has_copy = false;         // ... some of which are gathered into a stream of their own (the others are read from their scans' streams)
    bool has_emit = false, has_spec = false;   // holds pictures whose first decode emits (k_huff_emit ...) / pictures of the two-pass path
};

// All device buffers of a batch come out of ONE allocation: the layout code runs twice, first measuring, then handing out
// slices.  (A batch needs ~30 buffers; 30 hipMalloc calls cost more than planning and enqueuing a group of 48 files.)
struct DevArena {
    uint8_t *base = nullptr;
    size_t off = 0;
    bool measuring = true;
    std::vector<void *> *separate = nullptr;     // debugging (MJX_NO_ARENA): one hipMalloc per buffer, as before
    template <class T>
    void take(T **p, size_t bytes)
    {
        bytes = (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
        if (!measuring) {
            if (separate) {
                void *q = nullptr;
                (void)hipMalloc(&q, bytes);
                separate->push_back(q);
                *p = reinterpret_cast<T *>(q);
            } else {
                *p = reinterpret_cast<T *>(base + off);
            }
        }
        off += bytes;
    }
};

struct EventPair {
    hipEvent_t a, b;
    int kind;
};

}   // namespace

struct mjx_batch {
    mjx_ctx *ctx = nullptr;
    mjx_opts opts{};
    std::vector<ImageInfo> info;        // every image the kernels see: the caller's pictures, and the scans of multi-scan
                                        // files as internal one-component pictures in front of theirs
    std::vector<size_t> visible;        // caller's picture i -> index into info / himages
    std::vector<DevImage> himages;
    std::vector<Chunk> chunks;
    DevImage *d_images = nullptr;
    uint8_t *d_scan = nullptr;
    size_t scan_pool_bytes = 0;
    // upload state: the linear staging buffer and the host-side copies of the small pools stay alive until the batch is
    // released (an asynchronous upload reads them after build_batch has returned; hipFree would synchronise the device)
    std::vector<void *> separate_allocs;
    size_t arena_bytes = 0;
    uint8_t *arena = nullptr;           // the one device allocation every d_* pointer below points into (see DevArena)
    bool h_mismatch_owned = true;
    size_t h_mismatch_bytes = 0;        // size of the pinned block behind h_mismatch (when owned)
    uint8_t *d_lin = nullptr;
    void *d_ii = nullptr;
    // scans that are de-stuffed on the device (ImagePlan::stuffed): raw bytes, per-scan descriptors, per-segment counts and bases,
    // restart-marker list.  has_stuffed: the DevImages on the device hold geometry the host copies (himages, h_segs) do not.
    uint8_t *d_raw = nullptr;
    void *d_di = nullptr;
    uint32_t *d_segcount = nullptr, *d_segbase = nullptr, *d_rst = nullptr;
    std::vector<DestuffImg> h_di;
    bool has_stuffed = false;
    size_t lut_pool_entries = 0;        // entries of the decode-table pool (identical tables stored once)
    void *d_meta_end = nullptr;         // [d_images, d_meta_end): the small pools, uploaded in one transfer
    std::vector<unsigned char> h_meta;  // ... from this host block, unless the caller lent pinned memory
    std::vector<LutEntry> h_lut;
    std::vector<float> h_qm;
    std::vector<unsigned char> h_ii;
    hipEvent_t ev_entropy[2] = {nullptr, nullptr}, ev_pixels[2] = {nullptr, nullptr};   // per scratch set, see run_chunk
    bool entropy_recorded[2] = {false, false}, pixels_recorded[2] = {false, false};
    uint32_t *d_loopctl = nullptr;      // 8 control words per chunk for k_huff_merge_loop (zero between launches)
    uint32_t *d_segflag[2] = {nullptr, nullptr};   // per scratch set: "running sums published" flags of k_dc_scan_t, one per (image, segment);
    uint32_t dc_gen[2] = {0, 0};                   // zero at creation, a flag is valid when it equals the set's launch count
    bool dc_two_pass = false;           // k_dc_scan_t gave up once on this batch (a workgroup waited too long for its predecessor): two passes from now on
    hipEvent_t uploaded = nullptr;      // recorded on ctx->upload behind the last upload command; the decode streams wait for it
    bool upload_pending = false;
    hipEvent_t copied = nullptr;        // asynchronous upload: the transfers are done (the upload-time kernels wait for it on a decode stream)
    bool alt_entropy_stream = false;    // mjx_decode_batch, every second group: the entropy stage runs on the context's third stream
    // mjx_decode_batch decodes its files in groups (upload of one group overlaps the decode of the one before): the batch
    // handed to the caller is then only a directory of the groups' batches
    std::vector<mjx_batch *> parts;
    std::vector<std::pair<uint32_t, uint32_t>> part_index;     // caller's picture -> (part, picture inside the part)
    LutEntry *d_lut = nullptr;
    float *d_qm = nullptr;
    uint32_t *d_segs = nullptr;         // restart segments of the unique images: (first subsequence, first bit) pairs
    std::vector<uint32_t> h_segs;       // host copy (mjx_batch_tile rebuilds plans from it)
    SubseqState *d_entry = nullptr, *d_exit = nullptr;
    uint8_t *d_gen = nullptr;           // [chunk subsequences]: which of the two sets of entry / exit / checkpoints is current (Gen2)
    uint32_t *d_blkbase = nullptr;
    uint32_t *d_cps = nullptr;          // [chunk subsequences / 256][kMaxCp][256] checkpoints (two words each)
    EmitSub *d_esub = nullptr;          // per subsequence of the chunk: what the emitting first decode left behind (single decode)
    uint32_t *d_pull = nullptr;         // [chunk images] straggler counts of k_huff_merge (per round)
    uint32_t *d_items = nullptr;        // [max_nsub][6] stragglers handed from k_huff_merge to k_huff_merge_tail
    uint32_t max_nsub = 1, max_chunk_images = 1;
    int32_t *d_segsum = nullptr;
    uint32_t *d_entries = nullptr;      // compact coefficient stream
    uint32_t *d_tile_eoff = nullptr;    // stream offset of every stage-B tile (+ sentinel per image)
    uint32_t *d_ebase = nullptr;        // per subsequence: stream entries before it
    uint32_t *d_img_entries = nullptr;  // per image: entries counted by the synchronisation passes
    uint32_t *d_img_flags = nullptr;    // per image: 1 = scan ends before all MCUs (truncated), set by k_huff_scan
    int32_t *d_dc = nullptr;            // per block: the predicted DC (what stage B reads)
    int16_t *d_dcd = nullptr;           // per block: the DC difference as the write pass decoded it (what the prediction kernels read)
    uint8_t *d_rgb = nullptr;
    size_t rgb_pool_bytes = 0;
    int *d_status = nullptr;
    unsigned long long *d_planes = nullptr;   // REF_COMPAT: f32 planes with write-order keys (chunk scratch)
    uint32_t gen_stride = 0;            // subsequences between the two sets of entry / exit / checkpoints (0: one set, MJX_MERGE_MEMO=0)
    uint32_t *d_unconv = nullptr;       // [chunks]: runs of the chunk whose synchronisation rounds had not converged when the rest of the entropy stage ran
                                        // (its pictures were skipped; mjx_batch_wait repairs the last run only) -- mjx_batch_unconverged_runs
    uint32_t *d_mismatch = nullptr;     // [chunks][kMisWords]: re-decodes of every synchronisation round; [kMaxFix]: set when the one-pass DC prediction gave up
    uint32_t *h_mismatch = nullptr;     // pinned mirror
    size_t huff_lds = 0, huff_lds2 = 0, idct_lds = 0;      // tables + HuffImage in LDS: the plain set (write pass), the set with pair parts (counting passes)
    bool decoded_entropy = false;
    unsigned last_stages = MJX_STAGE_ALL;   // what the last mjx_batch_decode was asked for: a repair in mjx_batch_wait runs the same stages
    int last_chunk_resident = -1;
    bool resident_second = false;   // ... and it lives in the second scratch set
    // second set of per-chunk scratch for the chunks that run on ctx->stream2 (null = single stream)
    struct Alt {
        SubseqState *d_entry = nullptr, *d_exit = nullptr;
        uint8_t *d_gen = nullptr;
        uint32_t *d_blkbase = nullptr, *d_ebase = nullptr, *d_cps = nullptr, *d_pull = nullptr, *d_items = nullptr;
        EmitSub *d_esub = nullptr;
        int32_t *d_segsum = nullptr, *d_dc = nullptr;
        int16_t *d_dcd = nullptr;
        unsigned long long *d_planes = nullptr;
        uint32_t *d_entries = nullptr, *d_tile_eoff = nullptr;
    } alt;
    bool dual = false;
    uint64_t scan_bytes = 0, rgb_bytes = 0, coef_bytes = 0, pixels = 0;
    // profiling
    std::vector<EventPair> events;
    std::vector<EventPair> event_pool;
    double ms[MJX_K_COUNT] = {0};
    uint64_t launches[MJX_K_COUNT] = {0};
};

namespace {

constexpr int kMaxFix = 16;
constexpr int kMisWords = kMaxFix + 4;     // words per chunk in d_mismatch / h_mismatch: the rounds' counts, then [kMaxFix] "k_dc_scan_t gave up"
constexpr uint32_t kLoopRounds = 48;      // rounds k_huff_merge_loop runs at most (noise at quality 99-100 needs 12-15 with 512-byte subsequences)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// One device block of at least `bytes`: a cached one of fitting size (at most 1.5x what is asked for), or a new allocation.
int arena_get(mjx_ctx *ctx, size_t bytes, uint8_t **out, size_t *got)
{
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        size_t best = ctx->cache.size();
        for (size_t k = 0; k < ctx->cache.size(); k++)
            if (ctx->cache[k].second >= bytes && ctx->cache[k].second <= bytes + bytes / 2 + (size_t(1) << 20) &&
                (best == ctx->cache.size() || ctx->cache[k].second < ctx->cache[best].second))
                best = k;
        if (best != ctx->cache.size()) {
            *out = ctx->cache[best].first;
            *got = ctx->cache[best].second;
            ctx->cache_bytes -= ctx->cache[best].second;
            ctx->cache.erase(ctx->cache.begin() + long(best));
            return MJX_OK;
        }
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        {   // out of memory with blocks in the cache: give them back and try once more
            std::lock_guard<std::mutex> lk(ctx->cache_mu);
            for (auto &blk : ctx->cache) (void)hipFree(blk.first);
            ctx->cache.clear();
            ctx->cache_bytes = 0;
        }
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return MJX_ERR_NOMEM; }
    }
    *out = static_cast<uint8_t *>(p);
    *got = bytes;
    return MJX_OK;
}
void arena_put(mjx_ctx *ctx, uint8_t *p, size_t bytes)
{
    if (!p) return;
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        if (ctx->cache_bytes + bytes <= ctx->cache_limit) {
            ctx->cache.emplace_back(p, bytes);
            ctx->cache_bytes += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

// Small pinned blocks (whole pages), recycled through the context.
uint32_t *pinned_get(mjx_ctx *ctx, size_t bytes, size_t *got)
{
    bytes = (bytes + 4095) & ~size_t(4095);
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        for (size_t k = 0; k < ctx->pinned_cache.size(); k++)
            if (ctx->pinned_cache[k].second >= bytes && ctx->pinned_cache[k].second <= 4 * bytes) {
                uint32_t *p = ctx->pinned_cache[k].first;
                *got = ctx->pinned_cache[k].second;
                ctx->pinned_cache.erase(ctx->pinned_cache.begin() + long(k));
                return p;
            }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *got = bytes;
    return static_cast<uint32_t *>(p);
}
void pinned_put(mjx_ctx *ctx, uint32_t *p, size_t bytes)
{
    if (!p) return;
    if (ctx && bytes <= (size_t(1) << 20)) {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        if (ctx->pinned_cache.size() < 32) { ctx->pinned_cache.emplace_back(p, bytes); return; }
    }
    (void)hipHostFree(p);
}

void release(mjx_batch *b)
{
    if (!b) return;
    for (mjx_batch *part : b->parts) release(part);
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    if (b->uploaded) (void)hipEventDestroy(b->uploaded);
    if (b->copied) (void)hipEventDestroy(b->copied);
    for (int k = 0; k < 2; k++) {
        if (b->ev_entropy[k]) (void)hipEventDestroy(b->ev_entropy[k]);
        if (b->ev_pixels[k]) (void)hipEventDestroy(b->ev_pixels[k]);
    }
    for (auto &e : b->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : b->event_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    arena_put(b->ctx, b->arena, b->arena_bytes);            // every other device pointer of the batch is a slice of it
    for (void *q : b->separate_allocs) (void)hipFree(q);
    if (b->h_mismatch && b->h_mismatch_owned) pinned_put(b->ctx, b->h_mismatch, b->h_mismatch_bytes);
    delete b;
}

void fill_dev_image(const ImagePlan &p, DevImage &d);

// A multi-scan picture (role 2, plans[kp]; its scans are the nparts plans in front of it): can stage B read its tiles straight
// from the scans' streams (DevImage::planar)?  Not with keep_coefs (mjx_batch_copy_coefs expands the gathered stream), not when a
// tile touches more than two MCU rows or more segments than the kernel holds, not when an interleaved scan's MCU grid is not the
// picture's -- those go through the gather kernels as before.
bool planar_ok(const mjx_ctx *ctx, bool keep_coefs, const std::vector<ImagePlan> &plans, size_t kp, uint32_t *tile_mcus_out)
{
    const ImagePlan &pic = plans[kp];
    if (!ctx->planar_direct || keep_coefs || pic.role != 2 || pic.status != MJX_OK || kp < pic.nparts || pic.ncomp != 3) return false;
    DevImage pd;
    fill_dev_image(pic, pd);
    const uint32_t T = pd.tile_mcus;
    uint32_t kinds = 0;
    for (uint32_t j = 0; j < pic.nparts; j++) {
        const ImagePlan &sp = plans[kp - pic.nparts + j];
        if (sp.status != MJX_OK || sp.role != 1 || sp.part_idx != j) return false;
        // (the kernel finds a block's place by shifts: a segment's blocks per MCU of the picture are a power of two and land in
        // consecutive slots of the MCU)
        auto pow2 = [](uint32_t x) { return x && !(x & (x - 1)); };
        if (sp.ncomp == 1) {
            for (uint32_t c = 0; c < 3; c++)
                if (pic.src_part[c] == j) {
                    if (!pow2(pic.h[c])) return false;
                    kinds += pic.v[c];
                }
        } else {
            if (sp.mcux != pic.mcux || sp.mcuy != pic.mcuy || !pow2(sp.bpm)) return false;
            int prev = -1;
            for (uint32_t q = 0; q < sp.ncomp; q++)
                for (uint32_t c = 0; c < 3; c++)
                    if (pic.src_part[c] == j && pic.src_comp[c] == q) {
                        if (prev >= 0 && int(c) != prev + 1) return false;
                        prev = int(c);
                    }
            kinds += 1;
        }
    }
    const uint32_t pieces = (pic.mcux + T - 2) / pic.mcux + 1;          // MCU rows a tile of T consecutive MCUs can touch
    if (tile_mcus_out) *tile_mcus_out = T;
    return (T & (T - 1)) == 0 && kinds >= 1 && kinds <= kPlanarKinds && pieces <= 2 && pieces * kinds <= kPlanarSegs;
}

// Fills the DevImage of image i from its plan (offsets are assigned by the caller).
void fill_dev_image(const ImagePlan &p, DevImage &d)
{
    std::memset(&d, 0, sizeof d);
    d.himg = p.himg;
    d.width = p.width; d.height = p.height; d.mcux = p.mcux; d.mcuy = p.mcuy; d.nmcu = p.nmcu;
    d.ncomp = p.ncomp; d.bpm = p.bpm; d.hmax = p.hmax; d.vmax = p.vmax;
    d.valid = 1;
    d.nseg = p.nseg;
    d.restart_mcus = p.restart_mcus;
    d.mode = (p.ncomp == 3 && p.h[0] == 2 && p.v[0] == 2 && p.h[1] == 1 && p.v[1] == 1 && p.h[2] == 1 && p.v[2] == 1) ? 1 : 0;
    uint32_t t = (d.mode == 1 && p.layout != MJX_LAYOUT_REF_COMPAT) ? tile_mcus_420() : tile_mcus(p.bpm, p.hmax), l2 = 0;
    while ((1u << (l2 + 1)) <= t) l2++;
    if (!(d.mode == 1 && p.layout != MJX_LAYOUT_REF_COMPAT)) t = 1u << l2;      // (the generic pixel phase needs a power of two)
    d.log2_tile = l2;
    d.tile_mcus = t;
    d.tile_blocks = t * p.bpm;
    if (p.layout == MJX_LAYOUT_REF_COMPAT) {
        d.mode = 2;
        for (uint32_t c = 0; c < 3; c++) { d.ref_xf[c] = uint8_t(p.ref_xf[c]); d.ref_yf[c] = uint8_t(p.ref_yf[c]); }
        d.nbx = p.nbx;
        d.nby = p.nby;
    }
    d.role = p.role;
    d.wg_lanes = p.wg_lanes;
    if (p.role == 1) {                 // a scan of a multi-scan file: no stage B; one tile offset per block for the gather (build_batch: or segment cuts, seg_S)
        d.mode = 7;
        d.log2_tile = 0;
        d.tile_mcus = 1;
        d.tile_blocks = 1;
        d.nparts = p.nparts;
        d.part_idx = p.part_idx;
    } else if (p.role == 2) {
        d.nparts = p.nparts;
        for (uint32_t c = 0; c < 3; c++) {
            d.src_back[c] = p.nparts - p.src_part[c];
            d.src_comp[c] = p.src_comp[c];
            d.cbw[c] = p.cbw[c];
            d.cbh[c] = p.cbh[c];
        }
    }
    std::memcpy(d.blk_comp, p.blk_comp, sizeof d.blk_comp);
    std::memcpy(d.blk_bx, p.blk_bx, sizeof d.blk_bx);
    std::memcpy(d.blk_by, p.blk_by, sizeof d.blk_by);
    uint32_t first = 0;
    for (uint32_t c = 0; c < 3; c++) {
        d.ch[c] = uint8_t(c < p.ncomp ? p.h[c] : 1);
        d.cv[c] = uint8_t(c < p.ncomp ? p.v[c] : 1);
        d.cfirst[c] = uint8_t(first);
        if (c < p.ncomp) first += p.h[c] * p.v[c];
    }
}

// Splits the batch into chunks and assigns chunk-relative offsets (subsequence arrays, coefficient blocks).
void plan_chunks(mjx_batch *b)
{
    const size_t n = b->info.size();
    const bool keep = b->opts.keep_coefs != 0;
    // A chunk should give every kernel many rounds of workgroups per CU (the last, partly filled round of a launch is what a
    // kernel loses: with two write workgroups per CU a launch of 2278 workgroups is 4.45 rounds): by default it is closed after
    // 1.5 GiB of scan (MJX_CHUNK_SCAN_MB; ~1600 4K images -- where the 24 GiB of stream capacity close it first --, all of 4096
    // 1080p images; measured with everything overlapped, 2048 4K pictures per step: 1 / 1.25 / 1.5 / 2 GiB 26.9 / 26.5 / 26.2 /
    // 26.2 ms at quality 75, 42.0 / 41.0 / 42.1 / 42.1 at quality 90, 23.4 / 24.0 / 24.1 / 24.1 at quality 50);
    // opts.chunk_images fixes the image count instead.
    const size_t per_chunk = std::min<size_t>(b->opts.chunk_images ? b->opts.chunk_images : 65535, 65535);
    uint64_t scan_target = b->opts.chunk_images ? ~uint64_t(0) : (uint64_t(3) << 29);
    if (const char *e = std::getenv("MJX_CHUNK_SCAN_MB")) { const long v = std::atol(e); if (v > 0 && !b->opts.chunk_images) scan_target = uint64_t(v) << 20; }
    // ... but never the whole of a large batch in one chunk: the second chunk's entropy kernels run beside the first one's (their
    // own stream) and fill what those leave idle -- the merge rounds' chains, the last round of every launch.  A batch that would
    // fit is cut 3 : 1 (2048 4K pictures at quality 50: 24.2 ms in one chunk, 23.1 as 1536 + 512; 4096 1080p pictures: 15.0 ms
    // in one, 14.5 as 3072 + 1024, 14.7 as two halves).
    if (!b->opts.chunk_images) {
        uint64_t total_scan = 0;
        for (size_t k = 0; k < n; k++) total_scan += b->info[k].scan_len;
        if (total_scan >= (uint64_t(256) << 20)) scan_target = std::min(scan_target, total_scan - total_scan / 4);      // (128 4K pictures, 138 MB: 2.34 ms in one chunk, 2.43 in two)
    }
    const uint64_t kMaxChunkEntries = (uint64_t(32) << 30) / 4;          // 32 GiB of stream capacity per chunk (24 before the columns got their head room, round 5)
    b->chunks.clear();
    uint64_t coef_running = 0, ent_running = 0;
    uint32_t tile_running = 0;
    size_t i = 0;
    while (i < n) {
        Chunk c;
        c.first = i;
        c.coef_base = keep ? coef_running : 0;
        c.ent_base = keep ? ent_running : 0;
        c.tile_base = keep ? tile_running : 0;
        for (;;) {
            if (i >= n) break;
            const ImageInfo &inf = b->info[i];
            // (the scans of a multi-scan file and their picture stay in one chunk: only a group's first image may open one)
            const bool head = inf.role == 0 || (inf.role == 1 && (i == 0 || b->info[i - 1].role != 1));
            if (head && c.count >= per_chunk) break;
            if (head && c.count > 0 && (c.entries + inf.ent_cap > kMaxChunkEntries || c.scan_bytes >= scan_target || c.nsub >= (uint64_t(1) << 22))) break;
            DevImage &d = b->himages[i];
            if (inf.status == MJX_OK) {
                d.sub_off = c.nsub;
                d.coef_off = c.coef_base + c.blocks;
                d.ent_off = c.ent_base + c.entries;
                d.tile_off = c.tile_base + c.tiles;
                b->info[i].coef_off = d.coef_off;
                b->info[i].ent_off = d.ent_off;
                b->info[i].tile_off = d.tile_off;
                c.entries += inf.ent_cap;
                c.tiles += inf.ntiles + 1;
                c.nsub += d.himg.nsub;
                if (d.himg.nsub > 1) c.min_sub_bits = std::min(c.min_sub_bits, d.himg.sub_bits);
                c.max_nsub = std::max(c.max_nsub, d.himg.nsub);
                if (d.role != 2) c.wg = std::max<uint32_t>(c.wg, d.wg_lanes ? d.wg_lanes : uint32_t(kHuffWg));
                c.scan_bytes += inf.scan_len;
                c.blocks += (inf.nblocks + 7) & ~uint64_t(7);          // regions of DC differences start on 32-byte sectors
                // (c.max_wg: set when the chunk is closed, from its longest scan and its workgroup size)
                if (d.himg.nsub > 1) c.merge_wgs = std::max<uint32_t>(c.merge_wgs, (d.himg.nsub - 1 + kMergeWg - 1) / kMergeWg);
                if (d.himg.nsub > 1) c.loop_participants += (d.himg.nsub - 1 + kMergeWg - 1) / kMergeWg;      // workgroups x with x * kMergeWg + 1 < nsub
                const uint32_t T = d.tile_mcus;
                if (d.role != 1) {
                    c.max_tiles = std::max<uint32_t>(c.max_tiles, (d.nmcu + T - 1) / T);
                    c.max_tile_blocks = std::max<uint32_t>(c.max_tile_blocks, T * d.bpm);
                }
                if (d.role == 2) { c.has_gather = true; if (!d.planar) c.has_copy = true; }
                if (d.emit) c.has_emit = true; else if (d.role != 2) c.has_spec = true;
                c.lut_cap = std::max<uint32_t>(c.lut_cap, d.lut_n);
                c.lut2_cap = std::max<uint32_t>(c.lut2_cap, d.lut2_n);
                c.mode_mask |= 1u << d.mode;
                if (d.role != 1) c.layout_mask |= d.planar ? 4u : d.ent_rows ? 2u : 1u;
                c.bpm_mask |= 1u << d.bpm;
                if (d.nseg > 1) c.max_restart_segs = std::max(c.max_restart_segs, d.nseg);
                if (d.mode == 2) {
                    d.plane_off = c.plane_words;
                    c.plane_words += uint64_t(d.width) * d.height * d.ncomp;
                    c.max_pixel_wgs = std::max<uint32_t>(c.max_pixel_wgs, uint32_t((uint64_t(d.width) * d.height + 255) / 256));
                }
                c.max_segs = std::max<uint32_t>(c.max_segs, (d.nmcu + kDcSegMcus - 1) / kDcSegMcus);
            }
            b->info[i].chunk = uint32_t(b->chunks.size());
            c.count++;
            i++;
        }
        // one launch for all the merge rounds only when its workgroups are certain to be resident together (they wait for one
        // another): a twelfth of the device's 3 x 256 slots, so that a dozen such launches (other contexts, other processes) still
        // fit side by side; a launch that cannot get its workgroups together gives up by itself (kLoopGaveUp)
        if (b->has_stuffed) c.wg = uint32_t(kHuffWg);      // (stuffed scans: lengths known on the device only.  k_huff_emit and its followers take the chunk's size since round 6)
        c.max_wg = (c.max_nsub + c.wg - 1) / c.wg;
        if (c.loop_participants > b->ctx->merge_loop_max) c.loop_participants = 0;
        if (b->has_stuffed) c.loop_participants = 0;    // (its workgroups are counted from nsub, which only the device knows exactly for those scans)
        coef_running += c.blocks;
        ent_running += c.entries;
        tile_running += c.tiles;
        b->chunks.push_back(c);
    }
}

int allocate_work_buffers(mjx_batch *b, DevArena &ar)
{
    uint32_t max_nsub = 1;
    uint64_t max_blocks = 1, total_blocks = 0, max_entries = 4, total_entries = 0;
    uint32_t max_tiles_arr = 1, total_tiles_arr = 0;
    uint32_t lut_cap = 8, lut2_cap = 8, max_tile_blocks = 1;
    size_t max_segsum = 1;
    for (const Chunk &c : b->chunks) {
        lut2_cap = std::max(lut2_cap, c.lut2_cap);
        max_segsum = std::max<size_t>(max_segsum, size_t(c.max_segs) * c.count);
        max_nsub = std::max(max_nsub, c.nsub);
        max_blocks = std::max(max_blocks, c.blocks);
        total_blocks += c.blocks;
        max_entries = std::max(max_entries, c.entries);
        total_entries += c.entries;
        max_tiles_arr = std::max(max_tiles_arr, c.tiles);
        total_tiles_arr += c.tiles;
        lut_cap = std::max(lut_cap, c.lut_cap);
        max_tile_blocks = std::max(max_tile_blocks, c.max_tile_blocks);
    }
    const uint64_t coef_blocks = b->opts.keep_coefs ? std::max<uint64_t>(total_blocks, 1) : max_blocks;
    // (two sets of entry / exit / checkpoints per subsequence, the second gen_stride behind the first: Gen2, mjx_kernels.hip)
    b->gen_stride = b->ctx->merge_memo ? uint32_t((size_t(max_nsub) + 255) / 256 * 256) : 0u;
    const size_t sets_nsub = size_t(max_nsub) + 2 * size_t(b->gen_stride);       // (a third set: what a re-decode records while it runs)
    ar.take(&b->d_entry, sets_nsub * sizeof(SubseqState));
    ar.take(&b->d_exit, sets_nsub * sizeof(SubseqState));
    ar.take(&b->d_gen, sets_nsub + 16);
    ar.take(&b->d_blkbase, size_t(max_nsub) * sizeof(uint32_t));
    ar.take(&b->d_cps, (sets_nsub + 256) / 256 * 256 * kMaxCp * 2 * sizeof(uint32_t));
    ar.take(&b->d_esub, size_t(max_nsub) * sizeof(EmitSub));
    size_t max_imgs = 1;
    for (const Chunk &c : b->chunks) max_imgs = std::max(max_imgs, c.count);
    b->max_nsub = max_nsub;
    b->max_chunk_images = uint32_t(max_imgs);
    ar.take(&b->d_pull, max_imgs * kMaxFix * sizeof(uint32_t));
    ar.take(&b->d_items, size_t(max_nsub) * 6 * sizeof(uint32_t));
    ar.take(&b->d_segsum, max_segsum * 3 * sizeof(int32_t));
    {
        uint64_t max_planes = 0;
        for (const Chunk &c : b->chunks) max_planes = std::max(max_planes, c.plane_words);
        if (max_planes) ar.take(&b->d_planes, size_t(max_planes) * 8);
    }
    ar.take(&b->d_entries, size_t(b->opts.keep_coefs ? std::max<uint64_t>(total_entries, 4) : max_entries) * 4 + 64);
    ar.take(&b->d_tile_eoff, size_t(b->opts.keep_coefs ? std::max<uint32_t>(total_tiles_arr, 1) : max_tiles_arr) * 4 + 16);
    ar.take(&b->d_ebase, size_t(max_nsub) * sizeof(uint32_t));
    ar.take(&b->d_dc, size_t(coef_blocks) * sizeof(int32_t) + 64);
    ar.take(&b->d_dcd, size_t(coef_blocks) * sizeof(int16_t) + 64);
    ar.take(&b->d_rgb, std::max<size_t>(b->rgb_pool_bytes, 16));
    // (d_img_entries, d_img_flags, d_status: laid out by build_batch inside the block of small pools, whose upload clears them)
    const size_t mm = std::max<size_t>(b->chunks.size(), 1) * kMisWords * sizeof(uint32_t);
    ar.take(&b->d_mismatch, mm);
    ar.take(&b->d_unconv, std::max<size_t>(b->chunks.size(), 1) * sizeof(uint32_t));
    if (!ar.measuring && !b->h_mismatch) {
        b->h_mismatch = pinned_get(b->ctx, mm, &b->h_mismatch_bytes);
        if (!b->h_mismatch) return MJX_ERR_NOMEM;
        std::memset(b->h_mismatch, 0, mm);
    }
    if (b->ctx->nstreams == 2 && b->chunks.size() > 1) {          // second scratch set for the chunks on stream2
        mjx_batch::Alt &a = b->alt;
        b->dual = true;
        ar.take(&a.d_entry, sets_nsub * sizeof(SubseqState));
        ar.take(&a.d_exit, sets_nsub * sizeof(SubseqState));
        ar.take(&a.d_gen, sets_nsub + 16);
        ar.take(&a.d_blkbase, size_t(max_nsub) * sizeof(uint32_t));
        ar.take(&a.d_ebase, size_t(max_nsub) * sizeof(uint32_t));
        ar.take(&a.d_cps, (sets_nsub + 256) / 256 * 256 * kMaxCp * 2 * sizeof(uint32_t));
        ar.take(&a.d_esub, size_t(max_nsub) * sizeof(EmitSub));
        ar.take(&a.d_pull, max_imgs * kMaxFix * sizeof(uint32_t));
        ar.take(&a.d_items, size_t(max_nsub) * 6 * sizeof(uint32_t));
        ar.take(&a.d_segsum, max_segsum * 3 * sizeof(int32_t));
        uint64_t max_planes = 0;
        for (const Chunk &c : b->chunks) max_planes = std::max(max_planes, c.plane_words);
        if (max_planes) ar.take(&a.d_planes, size_t(max_planes) * 8);
        if (b->opts.keep_coefs) {
            a.d_entries = b->d_entries; a.d_tile_eoff = b->d_tile_eoff; a.d_dc = b->d_dc; a.d_dcd = b->d_dcd;
        } else {
            ar.take(&a.d_entries, size_t(max_entries) * 4 + 64);
            ar.take(&a.d_tile_eoff, size_t(max_tiles_arr) * 4 + 16);
            ar.take(&a.d_dc, size_t(coef_blocks) * sizeof(int32_t) + 64);
            ar.take(&a.d_dcd, size_t(coef_blocks) * sizeof(int16_t) + 64);
        }
    }
    if (ar.measuring) return MJX_OK;
    HIPOK(hipMemsetAsync(b->d_unconv, 0, std::max<size_t>(b->chunks.size(), 1) * sizeof(uint32_t), b->ctx->upload));
    if (const char *e = std::getenv("MJX_POISON")) {
        // debugging aid: scratch that the kernels must write before they read it is filled with a byte pattern, so that a
        // read of stale memory (fresh allocations are usually zero, recycled ones hold the previous batch) shows at once
        const int v = std::atoi(e) & 0xff;
        const char *only = std::getenv("MJX_POISON_ONLY");               // one buffer (index below) instead of all
        const int sel = only ? std::atoi(only) : -1;
        const size_t cps_bytes = (size_t(max_nsub) + 256) / 256 * 256 * kMaxCp * 2 * sizeof(uint32_t);
        struct { void *p; size_t n; } bufs[] = {
            {b->d_entry, size_t(max_nsub) * sizeof(SubseqState)},                 // 0
            {b->d_exit, size_t(max_nsub) * sizeof(SubseqState)},                  // 1
            {b->d_blkbase, size_t(max_nsub) * sizeof(uint32_t)},                  // 2
            {b->d_ebase, size_t(max_nsub) * sizeof(uint32_t)},                    // 3
            {b->d_cps, cps_bytes},                                                // 4
            {b->d_items, size_t(max_nsub) * 6 * sizeof(uint32_t)},                // 5
            {b->d_segsum, max_segsum * 3 * sizeof(int32_t)},                      // 6
            {b->d_entries, size_t(b->opts.keep_coefs ? std::max<uint64_t>(total_entries, 4) : max_entries) * 4 + 64},      // 7
            {b->d_tile_eoff, size_t(b->opts.keep_coefs ? std::max<uint32_t>(total_tiles_arr, 1) : max_tiles_arr) * 4 + 16},   // 8
            {b->d_dc, size_t(coef_blocks) * sizeof(int32_t) + 64},                // 9
            {b->d_dcd, size_t(coef_blocks) * sizeof(int16_t) + 64},               // 10
            {b->d_esub, size_t(max_nsub) * sizeof(EmitSub)},                      // 11
        };
        for (int k = 0; k < int(sizeof bufs / sizeof bufs[0]); k++)
            if (sel < 0 || sel == k) HIPOK(hipMemsetAsync(bufs[k].p, v, bufs[k].n, b->ctx->upload));
    }
    b->huff_lds = huff_lds_bytes(lut_cap);
    b->huff_lds2 = huff_lds_bytes(lut2_cap);
    b->idct_lds = idct_lds_bytes(max_tile_blocks);
    // the largest dynamic LDS any entropy launch asks for, from the launchers' own expressions (mjx_kernels.hip): the write pass
    // on the plain table set, the counting passes on the set with pair parts
    const size_t lds_write = b->huff_lds + huff_window_bytes() + huff_stage_bytes();
    const size_t lds_count = std::max(b->huff_lds2 + std::max(huff_window_bytes(), huff_merge_bytes()), b->huff_lds + huff_prefix_bytes());
    if (std::max(lds_write, lds_count) > 160 * 1024 || b->idct_lds > 160 * 1024) return MJX_ERR_UNSUPPORTED_FORMAT;
    {
        const size_t pad = std::max(b->ctx->spec_lds_pad, std::max(b->ctx->merge_lds_pad, b->ctx->write_lds_pad));
        const size_t want_huff = std::max(lds_write, lds_count) + pad;
        const size_t want_idct = b->idct_lds + b->ctx->idct_lds_pad;
        if (want_huff > b->ctx->configured_huff || want_idct > b->ctx->configured_idct) {
            if (configure_kernels(std::max(want_huff, b->ctx->configured_huff), std::max(want_idct, b->ctx->configured_idct)) != 0) { (void)hipGetLastError(); return MJX_ERR_DEVICE; }
            b->ctx->configured_huff = std::max(want_huff, b->ctx->configured_huff);
            b->ctx->configured_idct = std::max(want_idct, b->ctx->configured_idct);
        }
    }
    return MJX_OK;
}

void prof_begin(mjx_batch *b, int kind, hipStream_t st)
{
    if (!b->ctx->profiling) return;
    EventPair e;
    if (!b->event_pool.empty()) { e = b->event_pool.back(); b->event_pool.pop_back(); }
    else { (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); }
    e.kind = kind;
    (void)hipEventRecord(e.a, st);
    b->events.push_back(e);
}
void prof_end(mjx_batch *b, hipStream_t st)
{
    if (!b->ctx->profiling) return;
    (void)hipEventRecord(b->events.back().b, st);
}

// Synchronisation rounds enqueued up front for a chunk: the context's number (MJX_FIX_PASSES, 6), and ten where the chunk holds
// scans cut shorter than 2048 bits -- small pictures in a large batch, cut so that they fill a workgroup (replan_subsequences):
// the chains of subsequences that do not synchronise are as long in bits, hence longer in subsequences (measured: 16384 x 512x512
// needs nine rounds, 8192 x 1024x768 eight, 32768 x 256x256 seven; a round with nothing to do is two launches that leave at
// once).  Fewer than a chunk needs is not an error -- mjx_batch_wait runs the rest -- but costs that wait a second pass.
int chunk_fix_passes(const mjx_batch *b, size_t ci)
{
    const int base = std::min(b->ctx->fix_passes, kMaxFix);
    // (large chunks only: a small batch pays for every launch, and its wait is behind one decode anyway)
    // A chunk that a wait had to repair remembers what it needed: the same pictures need the same rounds the next time.
    const int rule = (b->chunks[ci].min_sub_bits < 2048u && b->chunks[ci].nsub >= 65536u) ? std::max(base, std::min(10, kMaxFix)) : base;
    return std::max(rule, std::min(b->chunks[ci].learned_passes, kMaxFix));
}

// Enqueue one chunk.  `fix_passes` inter-workgroup passes are launched; the mismatch count of the last one is copied
// to the pinned mirror and examined in mjx_batch_wait.  `phases` selects which parts of the entropy stage run
// (the repair path of mjx_batch_wait continues fix passes without restarting the speculative decode).
enum { PH_SYNC = 1, PH_FIX = 2, PH_TAIL = 4, PH_ENTROPY_ALL = 7 };
int run_chunk(mjx_batch *b, size_t ci, unsigned stages, int fix_passes, unsigned phases = PH_ENTROPY_ALL, bool force_first = false)
{
    const Chunk &c = b->chunks[ci];
    if (c.count == 0 || c.nsub == 0) return MJX_OK;
    // Two streams (the default): the entropy stage of every chunk runs on `stream`, stage B on `stream2`, chained by events --
    // so the pixel kernel of chunk k (HBM stores, LDS) always shares the device with the entropy kernels of chunk k+1
    // (instruction issue, latency) instead of two kernels of one kind meeting by chance.  Odd chunks use the second set of
    // scratch buffers; the entropy stage of chunk k+2 waits for stage B of chunk k, which reads the set it writes.
    const bool second = b->dual && (ci & 1) && !force_first;
    const int set = second ? 1 : 0;
    hipStream_t st = (second && b->ctx->stream3) ? b->ctx->stream3 : b->ctx->stream;
    // (mjx_decode_batch: the groups of a list are batches of one chunk each; every second one takes the third stream for its
    // entropy stage, so that two groups' latency-bound kernel chains run side by side like the chunks of one batch do)
    if (b->alt_entropy_stream && b->ctx->stream3 && !force_first && !b->dual) st = b->ctx->stream3;
    hipStream_t sp = (b->ctx->stream2 && !force_first) ? b->ctx->stream2 : b->ctx->stream;
    if (sp != st && !b->ev_entropy[0]) {
        for (int k = 0; k < 2; k++) {
            HIPOK(hipEventCreateWithFlags(&b->ev_entropy[k], hipEventDisableTiming));
            HIPOK(hipEventCreateWithFlags(&b->ev_pixels[k], hipEventDisableTiming));
        }
    }
    if (sp != st && (stages & MJX_STAGE_ENTROPY) && b->pixels_recorded[set]) HIPOK(hipStreamWaitEvent(st, b->ev_pixels[set], 0));
#define SCR(x) (second ? b->alt.x : b->x)
    const DevImage *imgs = b->d_images + c.first;
    const uint32_t nimg = uint32_t(c.count);
    int32_t *dcb = SCR(d_dc);              // image offsets already include the chunk base
    fix_passes = std::min(fix_passes, kMaxFix);
    // (two generations per subsequence in the merge rounds -- Gen2, mjx_kernels.hip -- for the pictures of the two-pass path; a chunk
    // of nothing but pictures whose first decode emits runs exactly as before)
    const uint32_t gen_stride = c.has_spec ? b->gen_stride : 0u;
    if ((stages & MJX_STAGE_ENTROPY) && (phases & PH_SYNC)) {
        // (every subsequence starts with its first decode in the first set and nothing in the second: k_huff_spec clears its byte)
        if (c.has_spec) {
            prof_begin(b, MJX_K_HUFF_SYNC, st);
            launch_huff_spec(st, c.max_wg, nimg, b->huff_lds2, b->ctx->spec_lds_pad, imgs, b->d_scan, b->d_lut, SCR(d_entry), SCR(d_exit), SCR(d_cps), b->d_segs, c.wg, gen_stride ? SCR(d_gen) : nullptr);
            prof_end(b, st);
        }
        if (c.has_emit) {       // single decode: these pictures' first decode emits (LDS as the write pass: plain tables, windows, rings)
            prof_begin(b, MJX_K_HUFF_EMIT, st);
            launch_huff_emit(st, c.max_wg, nimg, b->huff_lds, b->ctx->write_lds_pad, imgs, b->d_scan, b->d_lut, SCR(d_entry), SCR(d_exit), SCR(d_cps), SCR(d_esub), SCR(d_entries), c.wg);
            prof_end(b, st);
        }
    }
    if ((stages & MJX_STAGE_ENTROPY) && (phases & PH_FIX)) {
        HIPOK(hipMemsetAsync(b->d_mismatch + ci * kMisWords, 0, kMisWords * sizeof(uint32_t), st));
        if (c.merge_wgs > 0 && c.loop_participants > 0 && fix_passes > 0) {
            // (the count of the last round run lands where the last enqueued round's would: mjx_batch_wait and k_huff_scan look there)
            prof_begin(b, MJX_K_HUFF_FIX, st);
            launch_huff_merge_loop(st, c.merge_wgs, nimg, b->huff_lds2, b->ctx->merge_lds_pad, imgs, b->d_scan, b->d_lut, SCR(d_entry), SCR(d_exit), SCR(d_cps),
                                   b->d_mismatch + ci * kMisWords + fix_passes - 1, b->d_segs, b->d_loopctl + ci * 8,
                                   c.loop_participants + (b->ctx->loop_fault ? 1u : 0u), kLoopRounds, b->ctx->loop_fault ? 1u << 12 : 1u << 22, SCR(d_esub), SCR(d_gen), gen_stride);
            prof_end(b, st);
        } else if (c.merge_wgs > 0) {
            HIPOK(hipMemsetAsync(SCR(d_pull), 0, size_t(nimg) * std::max(fix_passes, 1) * sizeof(uint32_t), st));      // the straggler counts of every round
            for (int k = 0; k < fix_passes; k++) {
                prof_begin(b, MJX_K_HUFF_FIX, st);
                launch_huff_merge(st, c.merge_wgs, nimg, b->huff_lds2, b->ctx->merge_lds_pad, imgs, b->d_scan, b->d_lut, SCR(d_entry), SCR(d_exit), SCR(d_cps),
                                  b->d_mismatch + ci * kMisWords + k, SCR(d_items), SCR(d_pull) + size_t(k) * nimg, b->d_segs,
                                  k > 0 ? b->d_mismatch + ci * kMisWords + k - 1 : nullptr,
                                  // (a chunk of pictures whose lanes warmed up: a fifth of the subsequences re-decode, not all of them --
                                  // the first round, too, only lists its items and the straggler kernel decodes them packed)
                                  k == 0 && (phases & PH_SYNC) && !(c.has_emit && !c.has_spec && b->ctx->emit_merge_listed), SCR(d_esub),
                                  c.max_nsub ? c.max_nsub - 1 : 0u, SCR(d_gen), gen_stride);
                prof_end(b, st);
            }
        }
    }
    if ((stages & MJX_STAGE_ENTROPY) && (phases & PH_TAIL)) {
        prof_begin(b, MJX_K_HUFF_SCAN, st);
        // the round whose count decides whether the synchronisation has converged: the last one enqueued above, or --
        // tail-only call of the repair path -- the last of a full set of rounds
        const uint32_t *verdict = nullptr;
        if (c.merge_wgs > 0) {
            const int last = (phases & PH_FIX) ? fix_passes - 1 : kMaxFix - 1;
            if (last >= 0) verdict = b->d_mismatch + ci * kMisWords + last;
        }
        // (the merge rounds are over: their straggler lists and counts are free, the scan lists the prefix pass's subsequences there)
        launch_huff_scan(st, nimg, imgs, SCR(d_exit), SCR(d_blkbase), SCR(d_ebase), b->d_img_entries, b->d_img_flags, b->d_segs, verdict,
                         SCR(d_esub), SCR(d_items), SCR(d_pull), b->d_mismatch + ci * kMisWords + kMaxFix + 1, b->d_unconv + ci,
                         SCR(d_entry), SCR(d_gen), gen_stride);
        prof_end(b, st);
        if (c.has_spec) {
            prof_begin(b, MJX_K_HUFF_WRITE, st);
            launch_huff_write(st, c.max_wg, nimg, b->huff_lds, b->ctx->write_lds_pad, imgs, b->d_scan, b->d_lut, SCR(d_entry), SCR(d_blkbase), SCR(d_ebase),
                                  SCR(d_entries), SCR(d_tile_eoff), SCR(d_dcd), b->d_status, b->d_img_flags, b->d_segs, SCR(d_exit), SCR(d_cps), c.wg);
            prof_end(b, st);
        }
        if (c.has_emit) {       // the prefixes of the lanes whose entry was wrong; block words -> DC differences + tile offsets
            prof_begin(b, MJX_K_HUFF_PREFIX, st);
            launch_huff_prefix(st, c.max_wg, nimg, b->huff_lds, imgs, b->d_scan, b->d_lut, SCR(d_entry), SCR(d_exit), SCR(d_cps), SCR(d_esub), SCR(d_blkbase),
                               SCR(d_entries), b->d_status, b->d_img_flags, b->d_mismatch + ci * kMisWords + kMaxFix + 1, SCR(d_dcd), SCR(d_tile_eoff), SCR(d_items), SCR(d_pull), b->d_unconv + ci, c.wg);
            prof_end(b, st);
        }
        prof_begin(b, MJX_K_DC_SCAN, st);
        launch_dc_scan(st, c.max_segs, nimg, imgs, SCR(d_dcd), dcb, SCR(d_segsum), b->d_img_flags, c.bpm_mask, c.max_restart_segs,
                       // (a repair run -- force_first -- takes the two-pass kernels, which wait for nobody: nothing looks at the
                       // "gave up" word after it, round-3 review)
                       (b->ctx->dc_one_pass && !b->dc_two_pass && !force_first) ? b->d_segflag[set] : nullptr, ++b->dc_gen[set],
                       b->d_mismatch + ci * kMisWords + kMaxFix, b->ctx->dc_fault ? 1u << 10 : 1u << 20, b->ctx->dc_fault);
        prof_end(b, st);
        if (c.has_gather) {
            prof_begin(b, MJX_K_GATHER, st);
            launch_planar_gather(st, c.max_tiles, nimg, imgs, SCR(d_entries), SCR(d_tile_eoff), dcb, b->d_img_flags, c.has_copy);
            prof_end(b, st);
        }
    }
    // what the host looks at in mjx_batch_wait: the rounds' counts and the "DC prediction gave up" word
    if ((stages & MJX_STAGE_ENTROPY) && (phases & (PH_FIX | PH_TAIL)))
        HIPOK(hipMemcpyAsync(b->h_mismatch + ci * kMisWords, b->d_mismatch + ci * kMisWords, kMisWords * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (sp != st && (stages & MJX_STAGE_ENTROPY)) {
        HIPOK(hipEventRecord(b->ev_entropy[set], st));
        b->entropy_recorded[set] = true;
    }
    if (stages & MJX_STAGE_PIXELS) {
        if (sp != st && b->entropy_recorded[set]) HIPOK(hipStreamWaitEvent(sp, b->ev_entropy[set], 0));
        prof_begin(b, MJX_K_IDCT_COLOR, sp);
        if (c.plane_words) HIPOK(hipMemsetAsync(SCR(d_planes), 0, size_t(c.plane_words) * 8, sp));
        // (dense: over ~1400 bytes of scan per stage-B tile -- more than the 2048 stream entries the kernel's default form prefetches)
        launch_idct_color(sp, c.max_tiles, nimg, b->idct_lds + b->ctx->idct_lds_pad, imgs, SCR(d_entries), SCR(d_tile_eoff), dcb, b->d_qm, b->d_rgb, c.mode_mask, SCR(d_planes), b->d_img_flags,
                          c.scan_bytes > uint64_t(c.tiles) * (1400u * tile_mcus_420() / 32u), c.layout_mask, b->ctx->idct_lds_pad);
        if (c.plane_words) launch_ref_color(sp, c.max_pixel_wgs, nimg, imgs, SCR(d_planes), b->d_rgb, b->d_img_flags);
        prof_end(b, sp);
        if (sp != st) {
            HIPOK(hipEventRecord(b->ev_pixels[set], sp));
            b->pixels_recorded[set] = true;
        }
    }
    HIPOK(hipGetLastError());
    return MJX_OK;
#undef SCR
}

int sync_streams(const mjx_batch *b)
{
    HIPOK(hipStreamSynchronize(b->ctx->stream));
    if (b->ctx->stream2) HIPOK(hipStreamSynchronize(b->ctx->stream2));
    if (b->ctx->stream3) HIPOK(hipStreamSynchronize(b->ctx->stream3));
    return MJX_OK;
}

void collect_events(mjx_batch *b)
{
    for (auto &e : b->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            b->ms[e.kind] += ms;
            b->launches[e.kind]++;
        }
        b->event_pool.push_back(e);
    }
    b->events.clear();
}

// Pinned host memory lent to build_batch for the host mirrors of its small pools (mjx_decode_batch: an asynchronous copy from
// pageable memory is staged by the runtime and waits for the stream's earlier transfers -- the host thread would stall behind
// the previous group's DMA).
constexpr uint64_t kLongBatchSubs = uint64_t(1) << 19;      // short subsequences' worth of scan (256 MiB) below which a batch keeps them (build_batch)

unsigned usable_processors();

// The lane-interleaved region of one scan (LaneBits in mjx_kernels.hip: piece k of subsequence s at (k * cols + s) * 16, the
// pieces behind a subsequence's own continuing with the bytes that follow it in the scan, 0xAA past the end of the scan --
// huffman.rs:236-246 -- and in the padding columns), written by the host from the linear de-stuffed scan the caller handed over
// (jpeg/mod.rs:371-385 keeps it as one Vec<u8>; the decoder borrows it, decoder.rs:55).  What k_scan_interleave does on the
// device for the scans that are de-stuffed there; for scans that arrive de-stuffed the layout is part of packing them for the
// upload, and the device starts from the pool as it will read it.  Columns [s0, s1).
void host_interleave_columns(const ImagePlan &p, uint32_t nsub, uint32_t cols, uint32_t rows, uint8_t *dst, uint32_t s0, uint32_t s1)
{
    const uint8_t *src = p.scan;
    const size_t len = p.scan_len;
    const uint32_t sub_bytes = p.himg.sub_bits / 8u;
    uint32_t g = 0;                                                            // segment of subsequence s (restart intervals)
    for (uint32_t s = s0; s < s1; s++) {
        if (s >= nsub) continue;                                               // (padding column: stays 0xAA)
        size_t at;
        if (p.nseg <= 1 || p.seg.size() < 4) at = size_t(s) * sub_bytes;
        else {
            while (g + 1 < p.nseg && p.seg[2 * (g + 1)] <= s) g++;
            while (g > 0 && p.seg[2 * g] > s) g--;
            at = size_t(p.seg[2 * g + 1] >> 3) + size_t(s - p.seg[2 * g]) * sub_bytes;
        }
        uint8_t *col = dst + size_t(s) * 16u;
        for (uint32_t k = 0; k < rows; k++, at += 16) {
            uint8_t *d = col + size_t(k) * cols * 16u;
            if (at + 16 <= len) std::memcpy(d, src + at, 16);
            else if (at < len) std::memcpy(d, src + at, len - at);             // (the rest of the piece stays 0xAA)
        }
    }
}

struct PinnedBump {
    uint8_t *base = nullptr;
    size_t cap = 0, used = 0;
};

// Builds a batch from plans.  Scan bytes come from the plans' host pointers -- de-stuffed, or (ImagePlan::stuffed) as they
// stand in the file: those are compacted on the device (k_destuff_*), which also finds their length and restart markers
// and writes the geometry that depends on them into the DevImages -- or (src != nullptr) are copied on the device from
// `src`'s pool, `times` repetitions of its images.
int build_batch(mjx_ctx *ctx, const std::vector<ImagePlan> &plans_in, const mjx_opts &opts, const mjx_batch *src,
                size_t times, mjx_batch **out, int *status, bool async_upload = false,
                uint32_t *pinned_words = nullptr, size_t pinned_cap = 0, PinnedBump *pin = nullptr, bool latency_plan = true,
                bool alt_entropy_stream = false)
{
    // A batch too small to fill the device (one picture, a handful) is bound by the serial chain of a lane -- ~810 symbols
    // of a 512-byte subsequence per decode pass, 0.29 us each -- not by throughput: its scans are cut into shorter subsequences
    // (more lanes, shorter chains).  Short subsequences synchronise over several of them, i.e. over more rounds than are worth
    // enqueuing one by one, so the cut is tied to k_huff_merge_loop (all rounds in one launch, as many as it takes): the shortest
    // of 64 / 128 / 256 bytes with which the batch still fits that kernel's residency limit; 64 bytes only for tiny batches (the
    // rounds get longer with the number of subsequences: one 4K picture 0.77 ms with 128 bytes, 0.83 with 64; lena.jpeg 0.43 / 0.37).
    // One 512x512 picture, 512 -> 64 bytes: k_huff_spec 235 -> 41 us, k_huff_write 376 -> 61 us, merge rounds 163 -> 137 us.
    std::vector<ImagePlan> replanned;
    const std::vector<ImagePlan> *use = &plans_in;
    bool any_stuffed = false;
    for (const ImagePlan &p : plans_in) any_stuffed = any_stuffed || (p.status == MJX_OK && p.stuffed);
    if (!src && latency_plan && !any_stuffed && !ctx->throughput_plan && ctx->latency_nsub > 0 && ctx->merge_loop_max > 0) {       // (not for the groups of a pipelined list: they overlap, throughput counts)
        uint64_t total = 0;
        for (const ImagePlan &p : plans_in) if (p.status == MJX_OK) total += (uint64_t(p.himg.total_bits) + uint64_t(kSubseqBits) - 1) / uint64_t(kSubseqBits);    // in short subsequences, whatever length the picture got
        if (total > 0 && total <= ctx->latency_nsub) {
            for (uint32_t bits = std::max<uint32_t>(ctx->latency_sub_bits, total <= 1024 ? 512u : 1024u); bits < uint32_t(kSubseqBits); bits *= 2) {
                std::vector<ImagePlan> cut = plans_in;
                uint64_t wgs = 0;
                for (ImagePlan &p : cut) {
                    if (p.status != MJX_OK) continue;
                    replan_subsequences(p, bits);
                    if (p.himg.nsub > 1) wgs += (p.himg.nsub - 1 + kMergeWg - 1) / kMergeWg;
                }
                if (wgs <= ctx->merge_loop_max) { replanned.swap(cut); use = &replanned; break; }
            }
        }
        // too many merge workgroups for the loop kernel, but still far from filling the device (up to ~64 4K pictures): half-length
        // subsequences with the enqueued rounds (content that needs more than six of them pays one trip to the host)
        if (use == &plans_in && total > 0 && total <= ctx->medium_nsub) {
            replanned = plans_in;
            for (ImagePlan &p : replanned) if (p.status == MJX_OK) replan_subsequences(p, uint32_t(kSubseqBits) / 2);
            use = &replanned;
        }
        // not enough pictures to fill the device with long subsequences (half the lanes per picture): the short ones, as before
        // round 4 -- 32 4K pictures 1.15 instead of 1.77 ms, 128 pictures 2.3 instead of 2.7; from ~256 pictures on the long ones win
        if (use == &plans_in && total > 0 && total < kLongBatchSubs) {
            bool any_long = false;
            for (const ImagePlan &p : plans_in) any_long = any_long || (p.status == MJX_OK && p.himg.sub_bits > uint32_t(kSubseqBits) * 5 / 4);
            if (any_long) {
                replanned = plans_in;
                for (ImagePlan &p : replanned) if (p.status == MJX_OK && p.himg.sub_bits > uint32_t(kSubseqBits) * 5 / 4) replan_subsequences(p, uint32_t(kSubseqBits), false);
                use = &replanned;
            }
        }
    }
    // Scans that took the long subsequences because they fill ONE 256-lane workgroup with them (replan_subsequences, round 6) keep
    // them only in a batch of nothing else: a chunk runs its entropy kernels at its pictures' largest workgroup size, and next to
    // 512-lane pictures such a scan would sit in half an idle workgroup (and the chunk would launch both kernel families: 2048 x
    // 1080p 4:2:2, where two thirds of the pictures fit and a third does not, 390 -> 306 Gpixels/s).
    {
        bool any_fit = false, any_other = false;
        for (const ImagePlan &p : *use) {
            if (p.status != MJX_OK || p.role == 2) continue;
            const bool fit = p.wg_lanes == 256u && p.himg.sub_bits >= uint32_t(kLongSubseqBits);
            any_fit = any_fit || fit;
            any_other = any_other || !fit;
        }
        if (any_fit && any_other) {
            if (use == &plans_in) { replanned = plans_in; use = &replanned; }
            for (ImagePlan &p : replanned)
                if (p.status == MJX_OK && p.role != 2 && p.wg_lanes == 256u && p.himg.sub_bits >= uint32_t(kLongSubseqBits)) replan_subsequences(p, uint32_t(kSubseqBits), false);
        }
    }
    const std::vector<ImagePlan> &plans = *use;
    mjx_batch *b = new (std::nothrow) mjx_batch;
    if (!b) return MJX_ERR_NOMEM;
    struct Owner {                      // releases the half-built batch on every early exit, exceptions included -- behind the upload
        mjx_batch *b;                   // stream: copies into the block may still be queued, and release() hands the block to the next batch
        ~Owner() { if (b) { if (b->ctx && b->ctx->upload) (void)hipStreamSynchronize(b->ctx->upload); release(b); } }
    } owner{b};
    b->ctx = ctx;
    b->opts = opts;
    const size_t nu = plans.size(), n = nu * times;
    b->info.resize(n);
    b->himages.resize(n);
    // pools for the unique images
    // scan_off: the image's lane-interleaved region in the pool (what the kernels read, see LaneBits in mjx_kernels.hip);
    // lin_off: its linear de-stuffed scan in the staging buffer the region is built from at upload
    std::vector<uint64_t> scan_off(nu, 0), lin_off(nu, 0), raw_off(nu, 0);
    std::vector<uint32_t> lut_off(nu, 0), lut_n(nu, 0), seg_off(nu, 0);
    size_t scan_pool = 0, lin_pool = 0, lut_pool = 0, raw_pool = 0;
    auto layout_nsub = [](const ImagePlan &p) { return p.nsub_layout ? p.nsub_layout : p.himg.nsub; };
    b->has_stuffed = any_stuffed && !src;
    std::vector<char> lut_first(nu, 1);                     // 0: the image shares an earlier image's tables
    std::unordered_multimap<uint64_t, size_t> lut_seen;
    b->h_segs.clear();
    for (size_t k = 0; k < nu; k++) {
        if (plans[k].status != MJX_OK) continue;
        scan_off[k] = scan_pool;
        scan_pool += align_up(size_t(scan_region_bytes(layout_nsub(plans[k]), plans[k].himg.sub_bits)), 256);
        lin_off[k] = lin_pool;
        lin_pool += align_up(plans[k].scan_len, 16) + 16;
        if (plans[k].stuffed && !src) {                     // raw bytes for the device-side compaction: whole 64-byte pieces + one behind
            raw_off[k] = raw_pool;
            raw_pool += align_up(plans[k].scan_len, 64) + 64;
        }
        if (src) {                          // replicated on the device: the source batch's table pool, its offsets
            lut_off[k] = src->himages[k].lut_off;
            lut_n[k] = uint32_t(plans[k].lut.size());
            lut_pool = src->lut_pool_entries;
        } else
        {   // identical decode tables (every file written with the Annex-K tables, for one) are stored once: the table pool
            // of a batch then stays in L2 for the staging loads of every workgroup, and the upload shrinks by 12 KB per file
            const std::vector<LutEntry> &l = plans[k].lut;
            uint64_t hsh = 1469598103934665603ull;
            for (LutEntry e : l) hsh = (hsh ^ e) * 1099511628211ull;
            bool found = false;
            auto range = lut_seen.equal_range(hsh);
            for (auto it = range.first; it != range.second; ++it) {
                const size_t j = it->second;
                if (plans[j].lut.size() == l.size() && std::memcmp(plans[j].lut.data(), l.data(), l.size() * sizeof(LutEntry)) == 0) {
                    lut_off[k] = lut_off[j];
                    lut_first[k] = 0;
                    found = true;
                    break;
                }
            }
            lut_n[k] = uint32_t(l.size());
            if (!found) {
                lut_seen.emplace(hsh, k);
                lut_off[k] = uint32_t(lut_pool);
                lut_pool += l.size();
            }
        }
        seg_off[k] = uint32_t(b->h_segs.size() / 2);
        b->h_segs.insert(b->h_segs.end(), plans[k].seg.begin(), plans[k].seg.end());
    }
    scan_pool = align_up(std::max<size_t>(scan_pool, 16), 256);
    b->scan_pool_bytes = scan_pool * times;
    uint64_t rgb_pool = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t k = i % nu, rep = i / nu;
        const ImagePlan &p = plans[k];
        ImageInfo &inf = b->info[i];
        inf.status = p.status;
        DevImage &d = b->himages[i];
        std::memset(&d, 0, sizeof d);
        d.status_idx = uint32_t(i);
        if (p.status != MJX_OK) continue;
        fill_dev_image(p, d);
        d.status_idx = uint32_t(i);
        d.scan_off = rep * scan_pool + scan_off[k];
        d.scan_cols = scan_region_cols(layout_nsub(p));
        d.lut_off = lut_off[k];
        d.lut_n = p.lut_plain_n;
        d.lut2_off = lut_off[k] + p.lut_plain_n;
        d.lut2_n = lut_n[k] - p.lut_plain_n;
        d.qm_off = uint32_t(k * 192);
        d.seg_off = seg_off[k];
        inf.width = p.width; inf.height = p.height; inf.bpm = p.bpm; inf.nmcu = p.nmcu;
        inf.nblocks = uint64_t(p.nmcu) * p.bpm;
        inf.tile_blocks = d.tile_blocks;
        inf.ntiles = uint32_t((inf.nblocks + d.tile_blocks - 1) / d.tile_blocks);
        // every stream entry consumes at least 2 bits of scan (1-bit code + 1 value bit) and a block holds at most 63
        // every subsequence's run of entries is rounded up to a whole 32-byte group (null entries): see k_huff_write
        const uint64_t group_pad = uint64_t(p.himg.nsub) * stream_group_entries() + 2 * stream_group_entries();
        inf.ent_cap = (std::min<uint64_t>(uint64_t(p.scan_len) * 4, inf.nblocks * 63) + group_pad) / stream_group_entries() * stream_group_entries();   // regions start on whole store groups
        // with restart intervals the lanes also fill what the synchronisation passes counted after a segment's last block
        // (garbage, up to one entry per two bits of scan)
        if (p.restart_mcus) inf.ent_cap = (uint64_t(p.scan_len) * 4 + 64 + group_pad) / stream_group_entries() * stream_group_entries();
        // A picture of one scan gets the quad-interleaved stream: one column of fixed capacity per subsequence (mjx_kernels.h,
        // stream_phys); tile offsets are 32-bit virtual indices into the columns.
        const bool will_emit = ctx->single_decode && p.role == 0 && !ctx->linear_stream && p.nseg <= 1 && p.restart_mcus == 0 && p.himg.sub_bits >= ctx->emit_min_sub_bits;
        if (p.role == 0 && !ctx->linear_stream) {
            const uint32_t rows = stream_rows_for(p.himg.sub_bits, will_emit);
            const uint64_t cap = stream_quad_entries(layout_nsub(p), rows);
            if (rows < 65536u && cap < 0xffffffffull) {
                d.ent_rows = rows;
                d.ent_hdr = stream_hdr_entries(layout_nsub(p));
                inf.ent_cap = cap + d.ent_hdr;
            }
        }
        // single decode: the picture's first decode emits (one scan, no restart intervals, quad-interleaved stream, subsequences long
        // enough that the warm-up is a small share of them -- the short cuts of small batches keep the two-pass kernels)
        if (will_emit && d.ent_rows) {
            d.emit = 1;
            d.emit_head = ctx->emit_head;
            d.himg.cp_bits = p.himg.sub_bits >= 2 * ctx->emit_cp_bits ? ctx->emit_cp_bits : uint32_t(kCpBits);
            d.himg.warm_bits = std::min(ctx->emit_warm_bits, p.himg.sub_bits / 32u * 32u);
        }
        if (p.role == 1 && k + (p.nparts - p.part_idx) < nu) {       // a scan of a multi-scan file: segments instead of an offset per block?
            const size_t kp = k + (p.nparts - p.part_idx);
            uint32_t T = 0;
            if (planar_ok(ctx, b->opts.keep_coefs != 0, plans, kp, &T)) {
                const ImagePlan &pic = plans[kp];
                d.seg_T = T;
                d.seg_mcux = pic.mcux;
                d.seg_S = planar_row_slots(pic.mcux, T);
                d.seg_hs = d.seg_vs = 1;
                if (p.ncomp == 1)
                    for (uint32_t c = 0; c < 3; c++)
                        if (pic.src_part[c] == p.part_idx) { d.seg_hs = pic.h[c]; d.seg_vs = pic.v[c]; }
                inf.ntiles = p.mcuy * d.seg_S;
            }
        }
        if (p.role == 2 && planar_ok(ctx, b->opts.keep_coefs != 0, plans, k, nullptr)) {
            d.planar = 1;
            uint32_t nk = 0, first[3] = {0, 0, 0};
            for (uint32_t c = 1; c < 3; c++) first[c] = first[c - 1] + p.h[c - 1] * p.v[c - 1];
            for (uint32_t j = 0; j < p.nparts; j++) {
                const ImagePlan &sp = plans[k - p.nparts + j];
                if (sp.ncomp == 1) {
                    for (uint32_t c = 0; c < 3; c++) {
                        if (p.src_part[c] != j) continue;
                        for (uint32_t v = 0; v < p.v[c]; v++, nk++) {
                            d.pk_back[nk] = uint8_t(p.nparts - j);
                            d.pk_v[nk] = uint8_t(v); d.pk_vs[nk] = uint8_t(p.v[c]); d.pk_hs[nk] = uint8_t(p.h[c]); d.pk_u[nk] = uint8_t(p.h[c]);
                            for (uint32_t w = 0; w < p.h[c]; w++) d.pk_map[nk][w] = uint8_t(first[c] + v * p.h[c] + w);
                        }
                    }
                } else {
                    d.pk_back[nk] = uint8_t(p.nparts - j);
                    d.pk_v[nk] = 0; d.pk_vs[nk] = 1; d.pk_hs[nk] = 1; d.pk_u[nk] = uint8_t(sp.bpm);
                    uint32_t at = 0;
                    for (uint32_t q = 0; q < sp.ncomp; q++)
                        for (uint32_t c = 0; c < 3; c++)
                            if (p.src_part[c] == j && p.src_comp[c] == q)
                                for (uint32_t o = 0; o < p.h[c] * p.v[c]; o++) d.pk_map[nk][at++] = uint8_t(first[c] + o);
                    nk++;
                }
            }
            d.pk_n = uint8_t(nk);
            inf.planar = true;
        }
        inf.emit = d.emit != 0;
        inf.emit_head = d.emit_head;
        inf.ent_rows = d.ent_rows;
        inf.ent_hdr = d.ent_hdr;
        inf.role = p.role;
        if (p.role == 2) {                     // gathered from the three scans in front of it
            if (i < p.nparts) return MJX_ERR_INVALID_ARG;
            inf.nparts = p.nparts;
            inf.ent_cap = 8;
            for (uint32_t k = 1; k <= p.nparts && !inf.planar; k++) inf.ent_cap += b->info[i - k].ent_cap;
        }
        inf.ent_cap = (inf.ent_cap + 31) / 32 * 32;        // regions start on whole 128-byte lines (rows of the quad-interleaved stream)
        d.ent_cap = uint32_t(std::min<uint64_t>(inf.ent_cap, 0xffffffffu));
        inf.scan_len = p.scan_len;
        inf.rgb_off = rgb_pool;
        inf.rgb_bytes = p.role == 1 ? 0 : uint64_t(p.width) * p.height * 3;     // (a scan of a multi-scan file has no picture)
        d.rgb_off = rgb_pool;
        rgb_pool += align_up(inf.rgb_bytes, 256);
        b->scan_bytes += p.scan_len;
        b->rgb_bytes += inf.rgb_bytes;
        if (p.role != 1) {
            b->coef_bytes += inf.nblocks * 128;
            b->pixels += uint64_t(p.width) * p.height;
        }
    }
    for (size_t i = 0; i < n; i++)
        if (b->info[i].role != 1) b->visible.push_back(i);
    b->rgb_pool_bytes = rgb_pool;
    b->lut_pool_entries = lut_pool;
    plan_chunks(b);
    if (pinned_words && std::max<size_t>(b->chunks.size(), 1) * kMisWords <= pinned_cap) {    // lent by the caller (mjx_decode_batch)
        b->h_mismatch = pinned_words;
        b->h_mismatch_owned = false;
        std::memset(b->h_mismatch, 0, std::max<size_t>(b->chunks.size(), 1) * kMisWords * sizeof(uint32_t));
    }

    int rc = MJX_OK;
    auto dev = [&]() -> int {
        hipStream_t up = ctx->upload;
        HIPOK(hipSetDevice(ctx->device));
        // Scans that lie close together in host memory in upload order (the pinned arena mjx_decode_batch parses into) go up
        // as ONE transfer, gaps included; the de-stuffed ones into the linear staging buffer, the stuffed ones into the raw one.
        struct Span { bool one = false; const uint8_t *p0 = nullptr; size_t bytes = 0; };
        auto find_span = [&](bool stuffed, size_t align) {
            Span sp;
            const uint8_t *prev_end = nullptr;
            size_t payload = 0, count = 0;
            bool ordered = true;
            for (size_t k = 0; k < nu && ordered; k++) {
                const ImagePlan &p = plans[k];
                if (p.status != MJX_OK || p.stuffed != stuffed || p.scan_len == 0 || !p.scan) continue;
                if (!sp.p0) sp.p0 = p.scan;
                if (prev_end && p.scan < prev_end) ordered = false;
                if ((size_t(p.scan - sp.p0) & (align - 1)) != 0) ordered = false;        // (staging offsets stay aligned)
                prev_end = p.scan + p.scan_len;
                payload += p.scan_len;
                count++;
            }
            // (only inside the context's own pinned arena: the gaps between the scans are read as well)
            const bool in_arena = sp.p0 && ctx->parse_arena && sp.p0 >= ctx->parse_arena && prev_end <= ctx->parse_arena + ctx->parse_arena_cap;
            if (ordered && in_arena && count > 1 && size_t(prev_end - sp.p0) <= payload + payload / 4 + 4096 + 128 * count) {
                sp.one = true;
                sp.bytes = size_t(prev_end - sp.p0);
            }
            return sp;
        };
        // Scans that arrive de-stuffed through mjx_batch_create are laid out lane-interleaved by the host while it packs them for
        // the upload (host_interleave_columns): the device starts from the pool as its kernels read it, no upload-time kernel
        // runs.  The groups of mjx_decode_batch (async_upload) keep the linear transfer + k_scan_interleave: there the host's
        // threads are busy parsing the next group and the kernel hides behind the transfers.
        const bool host_il = !src && !async_upload && ctx->host_interleave;
        const Span lin_span = host_il ? Span{} : find_span(false, 16), raw_span = any_stuffed ? find_span(true, 64) : Span{};
        const bool one_copy = lin_span.one;
        if (lin_span.one) {
            // (the stuffed scans' compacted copies follow the span in the linear buffer)
            size_t at = align_up(lin_span.bytes, 16) + 16;
            for (size_t k = 0; k < nu; k++) {
                if (plans[k].status != MJX_OK) continue;
                if (!plans[k].stuffed) { if (plans[k].scan_len && plans[k].scan) lin_off[k] = size_t(plans[k].scan - lin_span.p0); }
                else { lin_off[k] = at; at += align_up(plans[k].scan_len, 16) + 16; }
            }
            lin_pool = at;
        }
        if (raw_span.one) {
            raw_pool = align_up(raw_span.bytes, 64) + 128;
            for (size_t k = 0; k < nu; k++)
                if (plans[k].status == MJX_OK && plans[k].stuffed) raw_off[k] = size_t(plans[k].scan - raw_span.p0);
        }
        // stuffed scans: what the device-side compaction needs per scan (k_destuff_*)
        std::vector<DestuffImg> di;
        uint32_t destuff_segs = 0, destuff_max_seg = 0, rst_words = 0;
        bool destuff_restarts = false, any_direct = false;
        // the images that have a scan of their own to interleave
        std::vector<InterleaveImg> ii;
        uint32_t max_pieces = 0;
        if (!src)
            for (size_t k = 0; k < nu; k++) {
                const ImagePlan &p = plans[k];
                if (p.status != MJX_OK || p.himg.nsub == 0) continue;
                if (p.stuffed) {                                                                // (its length: written by k_destuff_prefix)
                    DestuffImg x{};
                    x.raw_off = raw_off[k];
                    x.raw_len = p.scan_len;
                    x.out_off = lin_off[k];
                    x.seg0 = destuff_segs;
                    x.nseg = uint32_t((p.scan_len + kDestuffSeg - 1) / kDestuffSeg);
                    x.image = uint32_t(k);
                    x.ii_index = uint32_t(ii.size());
                    x.restarts = p.restart_mcus ? 1u : 0u;
                    x.rst0 = rst_words;
                    x.rst_cap = p.restart_mcus ? p.nseg + 8u : 0u;
                    // (round 5) without restart intervals the compaction writes the lane-interleaved region itself
                    x.direct = (ctx->destuff_direct && !p.restart_mcus && p.nseg <= 1) ? 1u : 0u;
                    any_direct = any_direct || x.direct;
                    rst_words += x.rst_cap;
                    destuff_segs += x.nseg;
                    destuff_max_seg = std::max(destuff_max_seg, x.nseg);
                    destuff_restarts = destuff_restarts || x.restarts;
                    di.push_back(x);
                    if (x.direct) continue;                                                      // (no linear copy to lay out)
                }
                if (host_il && !p.stuffed) continue;                                            // (laid out by the host, below)
                ii.push_back(InterleaveImg{0, uint32_t(p.scan_len), uint32_t(k)});              // (lin_off is filled in below)
                max_pieces = std::max<uint32_t>(max_pieces, scan_region_cols(layout_nsub(p)) * scan_region_rows(p.himg.sub_bits));
            }
        size_t segflag_words = 6;                                       // (as max_segsum in allocate_work_buffers)
        for (const Chunk &c : b->chunks) segflag_words = std::max<size_t>(segflag_words, size_t(c.max_segs) * c.count * 6);
        {
            DevArena ar;
            auto layout = [&]() -> int {
                // the small pools first, back to back: they go up in one transfer from one host block (meta_*)
                ar.take(&b->d_images, std::max<size_t>(n, 1) * sizeof(DevImage));
                ar.take(&b->d_lut, std::max<size_t>(lut_pool, 8) * sizeof(LutEntry));
                ar.take(&b->d_qm, std::max<size_t>(nu, 1) * 192 * sizeof(float));
                ar.take(&b->d_segs, std::max<size_t>(b->h_segs.size(), 2) * sizeof(uint32_t));
                ar.take(&b->d_ii, std::max<size_t>(ii.size(), 1) * sizeof(InterleaveImg));
                // per-image words the kernels expect to be zero: inside the block, so that its one transfer clears them (three
                // hipMemsetAsync calls were six fill kernels in front of every group's transfer in mjx_decode_batch)
                ar.take(&b->d_img_entries, std::max<size_t>(n, 1) * sizeof(uint32_t));
                ar.take(&b->d_img_flags, std::max<size_t>(n, 1) * sizeof(uint32_t));
                ar.take(&b->d_status, std::max<size_t>(n, 1) * sizeof(int));
                ar.take(&b->d_loopctl, std::max<size_t>(b->chunks.size(), 1) * 8 * sizeof(uint32_t));
                ar.take(&b->d_segflag[0], 2 * segflag_words * sizeof(uint32_t));      // (segflag_words counts 32-bit words: 3 x 64 bit per segment)
                b->d_segflag[1] = b->d_segflag[0] + segflag_words;
                ar.take(&b->d_meta_end, 16);
                ar.take(&b->d_scan, b->scan_pool_bytes + 256);
                if (!src) ar.take(&b->d_lin, lin_pool + 256);
                if (!di.empty()) {
                    ar.take(&b->d_raw, raw_pool + 256);
                    ar.take(&b->d_di, di.size() * sizeof(DestuffImg));
                    ar.take(&b->d_segcount, size_t(destuff_segs) * 2 * sizeof(uint32_t) + 16);
                    ar.take(&b->d_segbase, size_t(destuff_segs) * 2 * sizeof(uint32_t) + 16);
                    ar.take(&b->d_rst, size_t(rst_words) * sizeof(uint32_t) + 16);
                }
                return allocate_work_buffers(b, ar);
            };
            { const int rcl = layout(); if (rcl != MJX_OK) return rcl; }
            if (std::getenv("MJX_NO_ARENA")) ar.separate = &b->separate_allocs;
            else { const int rcg = arena_get(ctx, ar.off, &b->arena, &b->arena_bytes); if (rcg != MJX_OK) return rcg; }
            ar.base = b->arena;
            ar.off = 0;
            ar.measuring = false;
            { const int rcl = layout(); if (rcl != MJX_OK) return rcl; }
        }
        const bool meta_block = b->arena != nullptr;                 // (MJX_NO_ARENA: separate buffers, separate copies)
        const size_t meta_bytes = meta_block ? size_t(reinterpret_cast<uint8_t *>(b->d_meta_end) - reinterpret_cast<uint8_t *>(b->d_images)) : 0;
        uint8_t *meta = nullptr;                                     // host mirror of [d_images, d_meta_end)
        if (meta_block) {
            if (pin && pin->used + meta_bytes <= pin->cap) {          // pinned (lent by mjx_decode_batch): a truly asynchronous copy
                meta = pin->base + pin->used;
                pin->used += (meta_bytes + 255) & ~size_t(255);
            } else {
                b->h_meta.resize(meta_bytes);
                meta = b->h_meta.data();
            }
        }
        auto mirror = [&](const void *dev) { return meta + (reinterpret_cast<const uint8_t *>(dev) - reinterpret_cast<const uint8_t *>(b->d_images)); };
        const size_t zero_words = std::max<size_t>(n, 1);
        if (meta_block) {
            std::memset(mirror(b->d_img_entries), 0, zero_words * sizeof(uint32_t));
            std::memset(mirror(b->d_img_flags), 0, zero_words * sizeof(uint32_t));
            std::memset(mirror(b->d_status), 0, zero_words * sizeof(int));
            std::memset(mirror(b->d_loopctl), 0, std::max<size_t>(b->chunks.size(), 1) * 8 * sizeof(uint32_t));
            std::memset(mirror(b->d_segflag[0]), 0, 2 * segflag_words * sizeof(uint32_t));
            std::memcpy(mirror(b->d_images), b->himages.data(), n * sizeof(DevImage));
            if (!b->h_segs.empty()) std::memcpy(mirror(b->d_segs), b->h_segs.data(), b->h_segs.size() * sizeof(uint32_t));
        } else {
            HIPOK(hipMemsetAsync(b->d_img_entries, 0, zero_words * sizeof(uint32_t), up));
            HIPOK(hipMemsetAsync(b->d_img_flags, 0, zero_words * sizeof(uint32_t), up));
            HIPOK(hipMemsetAsync(b->d_status, 0, zero_words * sizeof(int), up));
            HIPOK(hipMemsetAsync(b->d_loopctl, 0, std::max<size_t>(b->chunks.size(), 1) * 8 * sizeof(uint32_t), up));
            HIPOK(hipMemsetAsync(b->d_segflag[0], 0, 2 * segflag_words * sizeof(uint32_t), up));
            HIPOK(hipMemcpyAsync(b->d_images, b->himages.data(), n * sizeof(DevImage), hipMemcpyHostToDevice, up));
            if (!b->h_segs.empty()) HIPOK(hipMemcpyAsync(b->d_segs, b->h_segs.data(), b->h_segs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, up));
        }
        if (src) {
            if (src->scan_pool_bytes != scan_pool) return MJX_ERR_INVALID_ARG;
            if (meta_block) HIPOK(hipMemcpyAsync(b->d_images, meta, meta_bytes, hipMemcpyHostToDevice, up));
            // (on the upload stream, which is synchronised below: a device-to-device hipMemcpy on the null stream may return
            // before the copy has happened, and the decode streams are not ordered behind the null stream -- a decode kernel
            // that starts on a table pool of zeros never leaves its first symbol)
            for (size_t rep = 0; rep < times; rep++)
                HIPOK(hipMemcpyAsync(b->d_scan + rep * scan_pool, src->d_scan, scan_pool, hipMemcpyDeviceToDevice, up));
            HIPOK(hipMemcpyAsync(b->d_lut, src->d_lut, std::max<size_t>(lut_pool, 8) * sizeof(LutEntry), hipMemcpyDeviceToDevice, up));
            HIPOK(hipMemcpyAsync(b->d_qm, src->d_qm, std::max<size_t>(nu, 1) * 192 * sizeof(float), hipMemcpyDeviceToDevice, up));
        } else {
            const bool timing = std::getenv("MJX_TIMING") != nullptr;
            auto now = [] { return std::chrono::steady_clock::now(); };
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) { return std::chrono::duration<double, std::milli>(b2 - a).count(); };
            const auto t0 = now();
            // Asynchronous upload (mjx_decode_batch: one batch per group of files): the upload stream carries nothing but the
            // transfers, so that the DMA engine goes from one group's bytes straight to the next group's -- with the de-stuffing
            // and interleave kernels between them it idled 0.15 ms per group (a kernel, a small copy and two engine hand-overs).
            // The kernels go to the decode stream the group's entropy stage runs on, behind an event that marks the copies.
            b->alt_entropy_stream = async_upload && alt_entropy_stream && ctx->stream3 && ctx->group_alt_stream;
            const bool apart = async_upload && ctx->upload_kernels_apart;
            hipStream_t ks = apart ? (b->alt_entropy_stream ? ctx->stream3 : ctx->stream) : up;
            auto kernels_behind_copies = [&]() -> int {
                if (!apart || b->copied) return MJX_OK;
                HIPOK(hipEventCreateWithFlags(&b->copied, hipEventDisableTiming));
                HIPOK(hipEventRecord(b->copied, up));
                HIPOK(hipStreamWaitEvent(ks, b->copied, 0));
                return MJX_OK;
            };
            LutEntry *h_lut;
            float *h_qm;
            if (meta_block) {
                h_lut = reinterpret_cast<LutEntry *>(mirror(b->d_lut));
                h_qm = reinterpret_cast<float *>(mirror(b->d_qm));
            } else {
                b->h_lut.assign(std::max<size_t>(lut_pool, 8), 0);
                b->h_qm.assign(std::max<size_t>(nu, 1) * 192, 0.f);
                h_lut = b->h_lut.data();
                h_qm = b->h_qm.data();
            }
            for (size_t k = 0; k < nu; k++) {
                const ImagePlan &p = plans[k];
                if (p.status != MJX_OK) continue;
                if (lut_first[k]) std::memcpy(h_lut + lut_off[k], p.lut.data(), p.lut.size() * sizeof(LutEntry));
                std::memcpy(h_qm + k * 192, p.qmult, sizeof p.qmult);
            }
            if (!meta_block) {
                HIPOK(hipMemcpyAsync(b->d_lut, b->h_lut.data(), b->h_lut.size() * sizeof(LutEntry), hipMemcpyHostToDevice, up));
                HIPOK(hipMemcpyAsync(b->d_qm, b->h_qm.data(), b->h_qm.size() * sizeof(float), hipMemcpyHostToDevice, up));
            }
            const auto t1 = now();
            // Every scan goes up straight from the caller's buffer into a linear staging buffer (a host-side staging copy of
            // the whole pool -- first-touch page faults included -- cost six times the transfer itself); k_scan_interleave then
            // builds the lane-interleaved regions the kernels read, padding with the 0xAA the reference reads past the end of
            // a scan (huffman.rs:236-246).  When the scans lie close together in host memory in upload order (the pinned
            // arena mjx_decode_batch de-stuffs into) they go up as ONE transfer, gaps included: 512 separate 1 MB copies reached
            // 36 GB/s, one copy runs at the link's rate.
            // the small pools first: the de-stuffing kernels write the geometry they find into the DevImages
            for (InterleaveImg &x : ii) x.lin_off = lin_off[x.image];
            if (meta_block) {
                if (!ii.empty()) std::memcpy(mirror(b->d_ii), ii.data(), ii.size() * sizeof(InterleaveImg));
                HIPOK(hipMemcpyAsync(b->d_images, meta, meta_bytes, hipMemcpyHostToDevice, up));       // every small pool at once
            } else if (!ii.empty()) {
                b->h_ii.assign(reinterpret_cast<const unsigned char *>(ii.data()), reinterpret_cast<const unsigned char *>(ii.data() + ii.size()));
                HIPOK(hipMemcpyAsync(b->d_ii, b->h_ii.data(), b->h_ii.size(), hipMemcpyHostToDevice, up));
            }
            if (any_direct) HIPOK(hipMemsetAsync(b->d_scan, 0xaa, b->scan_pool_bytes, up));      // (what the direct compaction does not write: past a scan's end, padding columns)
            if (host_il) {
                // groups of scans whose regions fill a staging block: the host's threads write block A while block B is on the link
                const size_t kStage = size_t(64) << 20;
                size_t k0 = 0;
                int which = 0;
                const unsigned nthr = std::max(1u, std::min(16u, usable_processors()));
                while (k0 < nu) {
                    size_t k1 = k0, first = nu, last = nu;
                    size_t span_lo = 0, span_hi = 0;
                    for (; k1 < nu; k1++) {
                        const ImagePlan &p = plans[k1];
                        // (a scan that is de-stuffed on the device has its pool region between its neighbours': a span that went on
                        // across it would overwrite that region -- 0xAA from the fill above, which k_destuff_scatter relies on past the
                        // scan's end -- with whatever the staging block holds there.  The span ends in front of it.  Round-5 advisor.)
                        if (p.status == MJX_OK && p.stuffed && p.himg.nsub != 0 && first != nu) break;
                        if (p.status != MJX_OK || p.stuffed || p.himg.nsub == 0) continue;
                        const size_t lo = scan_off[k1], hi = lo + size_t(scan_region_bytes(layout_nsub(p), p.himg.sub_bits));
                        if (first != nu && hi - span_lo > kStage) break;
                        if (first == nu) { first = k1; span_lo = lo; }
                        last = k1;
                        span_hi = hi;
                    }
                    if (first == nu) break;
                    const size_t bytes = span_hi - span_lo;
                    if (ctx->stage_pin_cap[which] < bytes) {
                        if (ctx->stage_done[which]) HIPOK(hipEventSynchronize(ctx->stage_done[which]));
                        if (ctx->stage_pin[which]) (void)hipHostFree(ctx->stage_pin[which]);
                        ctx->stage_pin[which] = nullptr;
                        ctx->stage_pin_cap[which] = 0;
                        void *hp = nullptr;
                        if (hipHostMalloc(&hp, std::max(bytes, kStage), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return MJX_ERR_NOMEM; }
                        ctx->stage_pin[which] = static_cast<uint8_t *>(hp);
                        ctx->stage_pin_cap[which] = std::max(bytes, kStage);
                    }
                    if (!ctx->stage_done[which]) HIPOK(hipEventCreateWithFlags(&ctx->stage_done[which], hipEventDisableTiming));
                    else HIPOK(hipEventSynchronize(ctx->stage_done[which]));           // the block's previous transfer has left it
                    uint8_t *blk = ctx->stage_pin[which];
                    // work items: (scan, range of columns); the threads take them in turn
                    struct Item { size_t k; uint32_t s0, s1; };
                    std::vector<Item> items;
                    for (size_t k = first; k <= last; k++) {
                        const ImagePlan &p = plans[k];
                        if (p.status != MJX_OK || p.stuffed || p.himg.nsub == 0) continue;
                        const uint32_t cols = scan_region_cols(layout_nsub(p));
                        for (uint32_t c0 = 0; c0 < cols; c0 += 256) items.push_back(Item{k, c0, std::min(cols, c0 + 256)});
                    }
                    std::atomic<size_t> next{0};
                    auto work = [&] {
                        for (;;) {
                            const size_t j = next.fetch_add(1);
                            if (j >= items.size()) return;
                            const ImagePlan &p = plans[items[j].k];
                            const uint32_t cols = scan_region_cols(layout_nsub(p)), rows = scan_region_rows(p.himg.sub_bits);
                            uint8_t *dst = blk + (scan_off[items[j].k] - span_lo);
                            for (uint32_t k = 0; k < rows; k++)                      // 0xAA first: past the end of the scan, padding columns
                                std::memset(dst + (size_t(k) * cols + items[j].s0) * 16u, 0xaa, size_t(items[j].s1 - items[j].s0) * 16u);
                            host_interleave_columns(p, p.himg.nsub, cols, rows, dst, items[j].s0, items[j].s1);
                        }
                    };
                    {
                        // (a picture or a handful: one thread -- starting and joining helpers costs more than laying out 100 KB)
                        const unsigned t = bytes < (size_t(4) << 20) ? 1u : unsigned(std::min<size_t>(nthr, items.size()));
                        std::vector<std::thread> pool;
                        for (unsigned q = 1; q < t; q++) pool.emplace_back(work);
                        work();
                        for (std::thread &th : pool) th.join();
                    }
                    // (alignment gaps between the regions of a block go up as they are: nobody reads them)
                    HIPOK(hipMemcpyAsync(b->d_scan + span_lo, blk, bytes, hipMemcpyHostToDevice, up));
                    HIPOK(hipEventRecord(ctx->stage_done[which], up));
                    which ^= 1;
                    k0 = last + 1;
                }
            } else if (one_copy) {
                HIPOK(hipMemcpyAsync(b->d_lin, lin_span.p0, lin_span.bytes, hipMemcpyHostToDevice, up));
            } else {
                for (size_t k = 0; k < nu; k++) {
                    const ImagePlan &p = plans[k];
                    if (p.status != MJX_OK || p.stuffed || p.scan_len == 0 || !p.scan) continue;
                    HIPOK(hipMemcpyAsync(b->d_lin + lin_off[k], p.scan, p.scan_len, hipMemcpyHostToDevice, up));
                }
            }
            if (!di.empty()) {
                if (raw_span.one) {
                    HIPOK(hipMemcpyAsync(b->d_raw, raw_span.p0, raw_span.bytes, hipMemcpyHostToDevice, up));
                } else {
                    for (size_t k = 0; k < nu; k++) {
                        const ImagePlan &p = plans[k];
                        if (p.status != MJX_OK || !p.stuffed || p.scan_len == 0) continue;
                        HIPOK(hipMemcpyAsync(b->d_raw + raw_off[k], p.scan, p.scan_len, hipMemcpyHostToDevice, up));
                    }
                }
                b->h_di = di;                                                  // (kept alive behind the asynchronous copy)
                HIPOK(hipMemcpyAsync(b->d_di, b->h_di.data(), b->h_di.size() * sizeof(DestuffImg), hipMemcpyHostToDevice, up));
                { const int rck = kernels_behind_copies(); if (rck != MJX_OK) return rck; }
                prof_begin(b, MJX_K_UPLOAD, ks);
                launch_destuff(ks, destuff_max_seg, uint32_t(di.size()), destuff_restarts, static_cast<const DestuffImg *>(b->d_di), b->d_raw,
                               b->d_segcount, b->d_segbase, b->d_lin, b->d_rst, b->d_images, static_cast<InterleaveImg *>(b->d_ii), b->d_segs,
                               b->d_img_flags, b->d_scan);
                prof_end(b, ks);
                HIPOK(hipGetLastError());
            }
            if (timing && !async_upload) {
                HIPOK(hipStreamSynchronize(up));
                std::fprintf(stderr, "[mjx] staging %.2f ms, H2D of %.1f MB (%s) %.2f ms\n", ms(t0, t1), (lin_pool + raw_pool) / 1e6, one_copy ? "one transfer" : "per scan", ms(t1, now()));
            }
            const auto t2 = now();
            {   // linear -> lane-interleaved
                if (!ii.empty()) {
                    const InterleaveImg *d_ii = static_cast<const InterleaveImg *>(b->d_ii);
                    { const int rck = kernels_behind_copies(); if (rck != MJX_OK) return rck; }
                    prof_begin(b, MJX_K_UPLOAD, ks);
                    for (size_t at = 0; at < ii.size(); at += 32768)
                        launch_scan_interleave(ks, max_pieces, uint32_t(std::min<size_t>(32768, ii.size() - at)), d_ii + at, b->d_images, b->d_lin, b->d_scan, b->d_segs);
                    prof_end(b, ks);
                    HIPOK(hipGetLastError());
                }
                if (timing && !async_upload) {
                    HIPOK(hipStreamSynchronize(up));
                    std::fprintf(stderr, "[mjx] interleave %.2f ms\n", ms(t2, now()));
                }
            }
        }
        if (async_upload) {
            // the caller goes on (parsing and uploading the next group of files) while the DMA engine works; the decode
            // streams wait for this event before their first kernel (run_chunk)
            HIPOK(hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming));
            HIPOK(hipEventRecord(b->uploaded, b->copied ? (b->alt_entropy_stream ? ctx->stream3 : ctx->stream) : up));
            b->upload_pending = true;
        } else {
            HIPOK(hipStreamSynchronize(up));

        }
        return MJX_OK;
    };
    rc = dev();
    if (rc != MJX_OK) return rc;
    if (status) for (size_t i = 0; i < b->visible.size(); i++) status[i] = b->info[b->visible[i]].status;
    owner.b = nullptr;
    *out = b;
    return MJX_OK;
}

std::mutex g_default_mu;
mjx_ctx *g_default_ctx = nullptr;

}   // namespace

// ---- context -------------------------------------------------------------------------------------
extern "C" int mjx_ctx_create(int device, mjx_ctx **out)
{
    return guarded([&]() -> int {
    if (!out) return MJX_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return MJX_ERR_DEVICE; }
    if (device < 0 || device >= ndev) return MJX_ERR_INVALID_ARG;
    HIPOK(hipSetDevice(device));
    mjx_ctx *c = new (std::nothrow) mjx_ctx;
    if (!c) return MJX_ERR_NOMEM;
    c->device = device;
    {
        // MJX_HIGH_PRIO = entropy | pixels | none: which of the two decode streams is created with the high priority
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char *hp = std::getenv("MJX_HIGH_PRIO");
        const bool ent_high = hp && std::strcmp(hp, "entropy") == 0 && hi != lo;
        const hipError_t e1 = ent_high ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi)
                                       : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e1 != hipSuccess) { (void)hipGetLastError(); delete c; return MJX_ERR_DEVICE; }
    }
    if (hipStreamCreateWithFlags(&c->upload, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return MJX_ERR_DEVICE; }
    if (const char *e = std::getenv("MJX_CACHE_GB")) c->cache_limit = size_t(std::max(0L, std::atol(e))) << 30;
    c->nstreams = 2;
    bool third = true;              // MJX_STREAMS=1: one stream, 2: entropy stage + stage B, 3 (default): the odd chunks' entropy stage on a stream of its own
    if (const char *e = std::getenv("MJX_STREAMS")) { c->nstreams = std::atoi(e) == 1 ? 1 : 2; third = std::atoi(e) >= 3; }
    if (const char *e = std::getenv("MJX_LATENCY_NSUB")) c->latency_nsub = uint64_t(std::max(0L, std::atol(e)));
    if (const char *e = std::getenv("MJX_MEDIUM_NSUB")) c->medium_nsub = uint64_t(std::max(0L, std::atol(e)));
    if (const char *e = std::getenv("MJX_STREAM_LINEAR")) c->linear_stream = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_HOST_INTERLEAVE")) c->host_interleave = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_DESTUFF_DIRECT")) c->destuff_direct = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_SINGLE_DECODE")) c->single_decode = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_PLANAR_DIRECT")) c->planar_direct = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_MERGE_MEMO")) c->merge_memo = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_EMIT_MERGE_LISTED")) c->emit_merge_listed = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_EMIT_CP_BITS")) c->emit_cp_bits = uint32_t(std::max(long(kCpBits), std::atol(e))) / uint32_t(kCpBits) * uint32_t(kCpBits);
    if (const char *e = std::getenv("MJX_EMIT_WARM_BITS")) c->emit_warm_bits = uint32_t(std::max(0L, std::atol(e))) / 32u * 32u;
    if (const char *e = std::getenv("MJX_EMIT_MIN_SUB_BITS")) c->emit_min_sub_bits = uint32_t(std::max(long(kCpBits), std::atol(e)));
    if (const char *e = std::getenv("MJX_EMIT_HEAD")) c->emit_head = uint32_t(std::max(0L, std::min(long(kEmitHeadGroups), std::atol(e))));      // (tests: no head room = every prefix that grows falls back)
    if (const char *e = std::getenv("MJX_LATENCY_SUB_BITS")) c->latency_sub_bits = uint32_t(std::max(512L, std::min(long(kSubseqBits), std::atol(e))));
    if (const char *e = std::getenv("MJX_DC_ONE_PASS")) c->dc_one_pass = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_UPLOAD_APART")) c->upload_kernels_apart = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_GROUP_ALT")) c->group_alt_stream = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_LOOP_FAULT")) c->loop_fault = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_DC_FAULT")) c->dc_fault = std::atoi(e) != 0;
    if (const char *e = std::getenv("MJX_MERGE_LOOP")) c->merge_loop_max = uint32_t(std::min(192L, std::max(0L, std::atol(e))));
    if (c->nstreams == 2) {
        // With two streams stage B's gets the higher priority (MJX_PIXEL_PRIORITY=0: equal): its workgroups are placed first when
        // a CU frees resources, so the pixel kernel keeps close to its stand-alone pace and the entropy kernels fill what it
        // leaves.  With three (two chunks' entropy stages side by side: the write pass waits for HBM, the synchronisation
        // passes for instruction issue) equal priorities measured better: 30.1 against 30.5 ms per step, 31.3 with two streams.
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char *pe = std::getenv("MJX_PIXEL_PRIORITY"), *hp = std::getenv("MJX_HIGH_PRIO");
        const bool prio = hp ? std::strcmp(hp, "pixels") == 0 : (pe ? std::atoi(pe) != 0 : !third);
        const hipError_t e2 = (prio && hi != lo) ? hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi)
                                                 : hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
        if (e2 != hipSuccess) { (void)hipGetLastError(); c->stream2 = nullptr; c->nstreams = 1; }
        if (third && c->stream2 && hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->stream3 = nullptr; }
    }
    if (const char *e = std::getenv("MJX_SPEC_LDS_PAD")) c->spec_lds_pad = size_t(std::atoi(e));
    if (const char *e = std::getenv("MJX_MERGE_LDS_PAD")) c->merge_lds_pad = size_t(std::atoi(e));
    if (const char *e = std::getenv("MJX_WRITE_LDS_PAD")) c->write_lds_pad = size_t(std::atoi(e));
    if (const char *e = std::getenv("MJX_IDCT_LDS_PAD")) c->idct_lds_pad = size_t(std::atoi(e));
    if (const char *e = std::getenv("MJX_FIX_PASSES")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= kMaxFix) c->fix_passes = v;
    }
    *out = c;
    return MJX_OK;
    });
}

extern "C" void mjx_ctx_destroy(mjx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->stream3) { (void)hipStreamSynchronize(ctx->stream3); (void)hipStreamDestroy(ctx->stream3); }
    if (ctx->upload) { (void)hipStreamSynchronize(ctx->upload); (void)hipStreamDestroy(ctx->upload); }
    if (ctx->parse_arena) (void)hipHostFree(ctx->parse_arena);
    if (ctx->pin_small) (void)hipHostFree(ctx->pin_small);
    if (ctx->rgb_pin) (void)hipHostFree(ctx->rgb_pin);
    for (int k = 0; k < 2; k++) {
        if (ctx->stage_pin[k]) (void)hipHostFree(ctx->stage_pin[k]);
        if (ctx->stage_done[k]) (void)hipEventDestroy(ctx->stage_done[k]);
    }
    for (auto &blk : ctx->cache) (void)hipFree(blk.first);
    for (auto &blk : ctx->pinned_cache) (void)hipHostFree(blk.first);
    delete ctx;
}

extern "C" int mjx_ctx_set_profiling(mjx_ctx *ctx, int enable)
{
    if (!ctx) return MJX_ERR_INVALID_ARG;
    ctx->profiling = enable != 0;
    return MJX_OK;
}

extern "C" int mjx_ctx_set_throughput_plan(mjx_ctx *ctx, int enable)
{
    if (!ctx) return MJX_ERR_INVALID_ARG;
    ctx->throughput_plan = enable != 0;
    return MJX_OK;
}

// NUMA node of the context's GPU: /sys/bus/pci/devices/<domain:bus:device.function>/numa_node (-1 there: the platform does
// not say, e.g. a single-node host or a VM).
extern "C" int mjx_ctx_numa_node(const mjx_ctx *ctx)
{
    if (!ctx) return -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, int(sizeof bus) - 1, ctx->device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; c++) *c = char(std::tolower(static_cast<unsigned char>(*c)));
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    int node = -1;
    if (FILE *f = std::fopen(path.c_str(), "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    return node;
}

extern "C" unsigned mjx_host_processors(void) { return usable_processors(); }

// ---- batch ---------------------------------------------------------------------------------------
extern "C" int mjx_batch_create(mjx_ctx *ctx, const mjx_scan_desc *descs, size_t n, const mjx_opts *opts,
                                mjx_batch **out, int *status)
{
    return guarded([&]() -> int {
    if (!ctx || !out || (!descs && n)) return MJX_ERR_INVALID_ARG;
    *out = nullptr;
    mjx_opts o{};
    if (opts) o = *opts;
    const mjx_scan_desc *dd = descs;
    std::vector<ImagePlan> plans;
    std::vector<size_t> plan_of(n);                    // input i -> its picture's plan (multi-scan files add plans in front)
    plans.reserve(n);
    const auto tp0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; i++) {
        plan_input(dd[i], o, plans);
        plan_of[i] = plans.size() - 1;
    }
    if (std::getenv("MJX_TIMING"))
        std::fprintf(stderr, "[mjx] planning %zu inputs %.2f ms\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
    (void)plan_of;
    return build_batch(ctx, plans, o, nullptr, 1, out, status);
    });
}

// Host-only: would this scan decode?  Runs the same planning step as mjx_batch_create (tables, geometry, and for
// MJX_LAYOUT_REF_COMPAT the inputs on which the reference panics) without touching the GPU.
extern "C" int mjx_validate(const mjx_scan_desc *desc, const mjx_opts *opts)
{
    return guarded([&]() -> int {
    if (!desc) return MJX_ERR_INVALID_ARG;
    mjx_opts o{};
    if (opts) o = *opts;
    std::vector<ImagePlan> plans;                      // (a multi-scan file: every scan is checked, the picture's plan is last)
    plan_input(*desc, o, plans);
    return plans.back().status;
    });
}

extern "C" int mjx_batch_tile(mjx_ctx *ctx, const mjx_batch *src, size_t times, mjx_batch **out)
{
    return guarded([&]() -> int {
    if (!ctx || !src || !out || times == 0 || !src->parts.empty()) return MJX_ERR_INVALID_ARG;
    *out = nullptr;
    // rebuild light-weight plans from the source batch's device images (geometry only; tables stay on the device)
    const size_t nu = src->info.size();
    std::vector<ImagePlan> plans(nu);
    // scans that were de-stuffed on the device: their length, subsequence count and segment table were written there
    std::vector<DevImage> dev_images;
    std::vector<uint32_t> dev_segs;
    if (src->has_stuffed) {
        HIPOK(hipSetDevice(src->ctx->device));
        HIPOK(hipStreamSynchronize(src->ctx->upload));
        dev_images.resize(nu);
        dev_segs.resize(src->h_segs.size());
        HIPOK(hipMemcpy(dev_images.data(), src->d_images, nu * sizeof(DevImage), hipMemcpyDeviceToHost));
        if (!dev_segs.empty()) HIPOK(hipMemcpy(dev_segs.data(), src->d_segs, dev_segs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    const std::vector<uint32_t> &segs_now = src->has_stuffed ? dev_segs : src->h_segs;
    for (size_t k = 0; k < nu; k++) {
        ImagePlan &p = plans[k];
        const DevImage &d = src->has_stuffed ? dev_images[k] : src->himages[k];
        p.status = src->info[k].status;
        if (p.status != MJX_OK) continue;
        p.width = d.width; p.height = d.height; p.ncomp = d.ncomp; p.bpm = d.bpm; p.hmax = d.hmax; p.vmax = d.vmax;
        p.mcux = d.mcux; p.mcuy = d.mcuy; p.nmcu = d.nmcu;
        for (uint32_t c = 0; c < 3; c++) { p.h[c] = d.ch[c]; p.v[c] = d.cv[c]; }
        std::memcpy(p.blk_comp, d.blk_comp, sizeof p.blk_comp);
        std::memcpy(p.blk_bx, d.blk_bx, sizeof p.blk_bx);
        std::memcpy(p.blk_by, d.blk_by, sizeof p.blk_by);
        p.himg = d.himg;
        p.layout = src->opts.layout;
        for (uint32_t c = 0; c < 3; c++) { p.ref_xf[c] = d.ref_xf[c]; p.ref_yf[c] = d.ref_yf[c]; }
        p.nbx = d.nbx;
        p.nby = d.nby;
        p.lut.assign(size_t(d.lut_n) + d.lut2_n, 0);      // sizes only: the pool is copied device-to-device
        p.lut_plain_n = d.lut_n;
        p.nseg = d.nseg;
        p.restart_mcus = d.restart_mcus;
        p.role = d.role;
        p.nparts = d.nparts;
        p.part_idx = d.part_idx;
        p.wg_lanes = d.wg_lanes ? d.wg_lanes : uint32_t(kHuffWg);
        for (uint32_t c = 0; c < 3; c++) {
            p.cbw[c] = d.cbw[c];
            p.cbh[c] = d.cbh[c];
            p.src_comp[c] = d.src_comp[c];
            p.src_part[c] = d.nparts - d.src_back[c];
        }
        p.seg.assign(segs_now.begin() + size_t(d.seg_off) * 2, segs_now.begin() + size_t(d.seg_off) * 2 + 2 * (size_t(d.nseg) + 1));
        p.nsub_layout = src->himages[k].scan_cols;        // (the region the source's pool was laid out for)
        p.scan = nullptr;
        p.scan_len = src->info[k].scan_len;
    }
    return build_batch(ctx, plans, src->opts, src, times, out, nullptr);
    });
}

extern "C" void mjx_batch_free(mjx_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)sync_streams(b);
    if (b->ctx->upload) (void)hipStreamSynchronize(b->ctx->upload);
    release(b);
}

extern "C" int mjx_batch_decode(mjx_batch *b, unsigned stages)
{
    return guarded([&]() -> int {
    if (!b) return MJX_ERR_INVALID_ARG;
    if (!b->parts.empty()) {
        for (mjx_batch *part : b->parts) { const int rc = mjx_batch_decode(part, stages); if (rc != MJX_OK) return rc; }
        return MJX_OK;
    }
    if (stages == 0) stages = MJX_STAGE_ALL;
    HIPOK(hipSetDevice(b->ctx->device));
    if ((stages & MJX_STAGE_PIXELS) && !(stages & MJX_STAGE_ENTROPY)) {
        // stage-B-only sweep: the coefficients of every chunk must still be resident
        if (!b->decoded_entropy || (!b->opts.keep_coefs && b->chunks.size() > 1)) return MJX_ERR_INVALID_ARG;
    }
    if (b->upload_pending) {                     // asynchronous upload (mjx_decode_batch): kernels start behind it
        HIPOK(hipStreamWaitEvent(b->ctx->stream, b->uploaded, 0));
        if (b->ctx->stream2) HIPOK(hipStreamWaitEvent(b->ctx->stream2, b->uploaded, 0));
        if (b->ctx->stream3) HIPOK(hipStreamWaitEvent(b->ctx->stream3, b->uploaded, 0));
        b->upload_pending = false;
    }
    for (size_t ci = 0; ci < b->chunks.size(); ci++) {
        const int rc = run_chunk(b, ci, stages, chunk_fix_passes(b, ci));
        if (rc != MJX_OK) return rc;
    }
    b->last_stages = stages;
    if (stages & MJX_STAGE_ENTROPY) {
        b->decoded_entropy = true;
        b->last_chunk_resident = int(b->chunks.size()) - 1;
        b->resident_second = b->dual && ((b->chunks.size() - 1) & 1);
    }
    return MJX_OK;
    });
}

extern "C" int mjx_batch_wait(mjx_batch *b)
{
    return guarded([&]() -> int {
    if (!b) return MJX_ERR_INVALID_ARG;
    if (!b->parts.empty()) {
        for (mjx_batch *part : b->parts) { const int rc = mjx_batch_wait(part); if (rc != MJX_OK) return rc; }
        return MJX_OK;
    }
    HIPOK(hipSetDevice(b->ctx->device));
    { const int rc0 = sync_streams(b); if (rc0 != MJX_OK) return rc0; }
    collect_events(b);
    // Fixed-point check: the last inter-workgroup pass of every chunk must have found nothing to repair.  If it did
    // (pathological stream that stays unsynchronised across a whole 32 KiB workgroup span), redo that chunk with
    // more passes until a pass counts zero.
    if (b->decoded_entropy) {
        bool first_set_rewritten = false;
        int attempts = 0;
        for (size_t ci = 0; ci < b->chunks.size(); ci++) {
            const Chunk &c = b->chunks[ci];
            // The one-pass DC prediction hands running sums from workgroup to workgroup; a workgroup that waited too long for
            // its predecessor (one that was never dispatched in front of it: nothing promises the dispatch order) gave up and
            // said so here.  The chunk's DC values are then part differences, part predictions: it is decoded again -- on the
            // first scratch set, with the two-pass kernels, which wait for nobody -- and so is everything this batch decodes later.
            if (b->h_mismatch[ci * kMisWords + kMaxFix] != 0) {
                if (std::getenv("MJX_TIMING")) std::fprintf(stderr, "[mjx] chunk %zu: one-pass DC prediction gave up, decoding the chunk again with two passes\n", ci);
                b->dc_two_pass = true;
                first_set_rewritten = true;
                HIPOK(hipMemsetAsync(b->d_status + c.first, 0, c.count * sizeof(int), b->ctx->stream));
                const int rcd = run_chunk(b, ci, b->last_stages | MJX_STAGE_ENTROPY, chunk_fix_passes(b, ci), PH_ENTROPY_ALL, true);
                if (rcd != MJX_OK) return rcd;
                HIPOK(hipStreamSynchronize(b->ctx->stream));
                collect_events(b);
                b->last_chunk_resident = int(ci);
                b->resident_second = false;
            }
            // Single decode: a lane's prefix found no head room in front of its first decode's entries (k_huff_prefix; or the counts did
            // not fit together).  The chunk's pictures are handed to the two-pass kernels, now and in later decodes of this batch.
            bool fell_back = false;
            auto emit_fallback = [&]() -> int {
                if (b->h_mismatch[ci * kMisWords + kMaxFix + 1] == 0) return MJX_OK;
                if (std::getenv("MJX_EXP_NO_FALLBACK")) return MJX_OK;      // (measurement builds that emit garbage: time the kernels, keep the path)
                fell_back = true;
                if (std::getenv("MJX_TIMING")) std::fprintf(stderr, "[mjx] chunk %zu: single decode fell back (reason bits %u, see k_huff_prefix), decoding the chunk again with the two-pass kernels\n", ci, b->h_mismatch[ci * kMisWords + kMaxFix + 1]);
                Chunk &cc = b->chunks[ci];
                // the pictures the kernels gave up on (flag 2: k_huff_scan -- more items than its list holds --, k_huff_prefix) leave the
                // single-decode path, here and on the device (patched in place, not uploaded again: the DevImages of scans that were
                // de-stuffed on the device hold geometry only the device knows); the chunk is decoded again, the others as before
                std::vector<uint32_t> fl(cc.count);
                HIPOK(hipMemcpy(fl.data(), b->d_img_flags + cc.first, cc.count * sizeof(uint32_t), hipMemcpyDeviceToHost));
                bool left = false;
                for (size_t i = cc.first; i < cc.first + cc.count; i++) {
                    DevImage &d = b->himages[i];
                    if (!d.emit) continue;
                    if (fl[i - cc.first] != 2u) { left = true; continue; }
                    d.emit = 0;
                    d.himg.cp_bits = uint32_t(kCpBits);
                    d.himg.warm_bits = 0;
                    b->info[i].emit = false;
                }
                cc.has_emit = left;
                cc.has_spec = true;
                first_set_rewritten = true;
                launch_emit_off(b->ctx->stream, b->d_images + cc.first, uint32_t(cc.count), b->d_img_flags);
                HIPOK(hipMemsetAsync(b->d_status + cc.first, 0, cc.count * sizeof(int), b->ctx->stream));
                HIPOK(hipMemsetAsync(b->d_img_flags + cc.first, 0, cc.count * sizeof(uint32_t), b->ctx->stream));
                const int rcf = run_chunk(b, ci, b->last_stages | MJX_STAGE_ENTROPY, chunk_fix_passes(b, ci), PH_ENTROPY_ALL, true);
                if (rcf != MJX_OK) return rcf;
                HIPOK(hipStreamSynchronize(b->ctx->stream));
                collect_events(b);
                b->last_chunk_resident = int(ci);
                b->resident_second = false;
                return MJX_OK;
            };
            { const int rcf = emit_fallback(); if (rcf != MJX_OK) return rcf; }
            // (the chunk has just been decoded again, with some of its pictures on the other path: it is looked at again from the top --
            // that decode may need more rounds than were enqueued, or may have lost further pictures; every fall-back takes at least
            // one picture off the single-decode path, so this ends)
            if (fell_back && ++attempts < 64) { ci--; continue; }
            if (c.merge_wgs == 0) continue;
            const int passes = chunk_fix_passes(b, ci);
            if (b->h_mismatch[ci * kMisWords + passes - 1] == 0) continue;
            if (std::getenv("MJX_TIMING")) std::fprintf(stderr, "[mjx] chunk %zu unconverged after %d rounds (%u re-decodes in the last): repairing\n", ci, passes, b->h_mismatch[ci * kMisWords + passes - 1]);
            // k_huff_merge_loop could not get its workgroups resident together and gave up (count = all ones): its control words
            // are in an unknown state; the chunk goes on with one launch per round, now and in later decodes
            auto loop_gave_up = [&](uint32_t count) -> int {
                if (count != 0xffffffffu || !b->chunks[ci].loop_participants) return MJX_OK;
                b->chunks[ci].loop_participants = 0;
                HIPOK(hipMemsetAsync(b->d_loopctl + ci * 8, 0, 8 * sizeof(uint32_t), b->ctx->stream));
                return MJX_OK;
            };
            { const int rcg = loop_gave_up(b->h_mismatch[ci * kMisWords + passes - 1]); if (rcg != MJX_OK) return rcg; }
            // repair: keep running fix passes -- each one extends the verified prefix -- until one finds nothing.  If a later
            // chunk has reused this chunk's state arrays (or the chunk ran on the second set, which the repair does not
            // use), its synchronisation starts again from the speculative decode; otherwise -- a batch of one chunk, the
            // usual case of a small batch with short subsequences -- the rounds simply continue where they stopped.
            const bool second_set = b->dual && (ci & 1);
            const bool intact = !second_set && !first_set_rewritten && (b->dual ? ci + 2 >= b->chunks.size() : ci + 1 >= b->chunks.size());
            if (!intact) first_set_rewritten = true;             // (a restarted repair runs on the first set, whatever the chunk)
            HIPOK(hipMemsetAsync(b->d_status + c.first, 0, c.count * sizeof(int), b->ctx->stream));
            int rc = MJX_OK;
            if (!intact) {
                rc = run_chunk(b, ci, MJX_STAGE_ENTROPY, 0, PH_SYNC, true);
                if (rc != MJX_OK) return rc;
            }
            const int more = intact ? passes : kMaxFix;
            int rounds_run = intact ? passes : 0;
            for (;;) {
                rc = run_chunk(b, ci, MJX_STAGE_ENTROPY, more, PH_FIX, true);
                if (rc != MJX_OK) return rc;
                HIPOK(hipStreamSynchronize(b->ctx->stream));
                if (std::getenv("MJX_TIMING")) std::fprintf(stderr, "[mjx]   repair rounds: %u re-decodes left\n", b->h_mismatch[ci * kMisWords + more - 1]);
                if (b->h_mismatch[ci * kMisWords + more - 1] == 0) {
                    // what the chunk needed: the rounds before this set + the first of this set that found nothing (+ 1: that round must run)
                    int first_zero = more - 1;
                    while (first_zero > 0 && b->h_mismatch[ci * kMisWords + first_zero - 1] == 0) first_zero--;
                    b->chunks[ci].learned_passes = std::min(kMaxFix, std::max(b->chunks[ci].learned_passes, rounds_run + first_zero + 1));
                    break;
                }
                rounds_run += more;
                { const int rcg = loop_gave_up(b->h_mismatch[ci * kMisWords + more - 1]); if (rcg != MJX_OK) return rcg; }
            }
            rc = run_chunk(b, ci, b->last_stages | MJX_STAGE_ENTROPY, 0, PH_TAIL, true);
            if (rc != MJX_OK) return rc;
            HIPOK(hipStreamSynchronize(b->ctx->stream));
            collect_events(b);
            b->last_chunk_resident = int(ci);
            b->resident_second = false;
            { const int rcf = emit_fallback(); if (rcf != MJX_OK) return rcf; }      // (the prefix pass ran for the first time just now)
            if (fell_back && ++attempts < 64) { ci--; continue; }
        }
    }
    std::vector<int> dev(b->info.size());
    std::vector<uint32_t> flags(b->info.size());
    const uint8_t *f0 = reinterpret_cast<const uint8_t *>(b->d_img_flags), *s0 = reinterpret_cast<const uint8_t *>(b->d_status);
    if (!dev.empty() && b->arena && s0 > f0 && size_t(s0 - f0) + dev.size() * sizeof(int) <= (size_t(1) << 20)) {
        // both arrays lie side by side in the block of small pools: one transfer (a synchronous copy costs ~15 us whatever its size)
        std::vector<uint8_t> both(size_t(s0 - f0) + dev.size() * sizeof(int));
        HIPOK(hipMemcpy(both.data(), f0, both.size(), hipMemcpyDeviceToHost));
        std::memcpy(flags.data(), both.data(), flags.size() * sizeof(uint32_t));
        std::memcpy(dev.data(), both.data() + (s0 - f0), dev.size() * sizeof(int));
    } else {
        if (!dev.empty()) HIPOK(hipMemcpy(dev.data(), b->d_status, dev.size() * sizeof(int), hipMemcpyDeviceToHost));
        if (!flags.empty()) HIPOK(hipMemcpy(flags.data(), b->d_img_flags, flags.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < dev.size(); i++) {
        if (b->info[i].status != MJX_OK) continue;
        if (flags[i]) b->info[i].status = MJX_ERR_TRUNCATED;           // scan ended before the last MCU
        else if (dev[i]) b->info[i].status = MJX_ERR_BAD_HUFFMAN;       // no code matched (huffman.rs:156/162)
    }
    for (size_t i = 0; i < dev.size(); i++) {                           // a multi-scan picture takes its scans' failures
        if (b->info[i].role != 2 || (b->info[i].status != MJX_OK && b->info[i].status != MJX_ERR_TRUNCATED)) continue;
        for (size_t k = 1; k <= b->info[i].nparts && k <= i; k++)
            if (b->info[i - k].status != MJX_OK) b->info[i].status = b->info[i - k].status;
    }
    return MJX_OK;
    });
}

namespace {
// caller's picture index -> internal image (multi-scan files keep their scans as internal images)
inline bool visible_index(const mjx_batch *b, size_t i, size_t &ii)
{
    if (!b || i >= b->visible.size()) return false;
    ii = b->visible[i];
    return true;
}
// A batch returned by mjx_decode_batch is a directory of the batches its groups of files were decoded in: picture i of the
// caller -> (the group's batch, the picture's index inside it).  Other batches route to themselves.
inline bool route(const mjx_batch *b, size_t i, mjx_batch *&pb, size_t &pi)
{
    if (!b) return false;
    if (b->parts.empty()) { pb = const_cast<mjx_batch *>(b); pi = i; return true; }
    if (i >= b->part_index.size()) return false;
    pb = b->parts[b->part_index[i].first];
    pi = b->part_index[i].second;
    return true;
}
}   // namespace

extern "C" size_t mjx_batch_size(const mjx_batch *b) { return !b ? 0 : (b->parts.empty() ? b->visible.size() : b->part_index.size()); }

extern "C" int mjx_batch_status(const mjx_batch *b, size_t iu)
{
    if (b && !b->parts.empty()) { mjx_batch *pb; size_t pi; return route(b, iu, pb, pi) ? mjx_batch_status(pb, pi) : MJX_ERR_INVALID_ARG; }
    size_t i;
    if (!visible_index(b, iu, i)) return MJX_ERR_INVALID_ARG;
    return b->info[i].status;
}

extern "C" int mjx_batch_image_info(const mjx_batch *b, size_t iu, uint32_t *width, uint32_t *height,
                                    uint32_t *blocks_per_mcu, uint32_t *mcus)
{
    if (b && !b->parts.empty()) { mjx_batch *pb; size_t pi; return route(b, iu, pb, pi) ? mjx_batch_image_info(pb, pi, width, height, blocks_per_mcu, mcus) : MJX_ERR_INVALID_ARG; }
    size_t i;
    if (!visible_index(b, iu, i)) return MJX_ERR_INVALID_ARG;
    const ImageInfo &inf = b->info[i];
    if (width) *width = inf.width;
    if (height) *height = inf.height;
    if (blocks_per_mcu) *blocks_per_mcu = inf.bpm;
    if (mcus) *mcus = inf.nmcu;
    return inf.status;
}

extern "C" int mjx_batch_rgb_device(const mjx_batch *b, size_t iu, void **dev_ptr, size_t *bytes)
{
    if (b && !b->parts.empty()) { mjx_batch *pb; size_t pi; return route(b, iu, pb, pi) ? mjx_batch_rgb_device(pb, pi, dev_ptr, bytes) : MJX_ERR_INVALID_ARG; }
    size_t i;
    if (!visible_index(b, iu, i) || !dev_ptr) return MJX_ERR_INVALID_ARG;
    const ImageInfo &inf = b->info[i];
    if (inf.status != MJX_OK) { *dev_ptr = nullptr; if (bytes) *bytes = 0; return inf.status; }
    *dev_ptr = b->d_rgb + inf.rgb_off;
    if (bytes) *bytes = size_t(inf.rgb_bytes);
    return MJX_OK;
}

// Device -> host memory the library does not own, through a pinned block of the context, in pieces: a copy straight into the
// caller's memory makes the runtime register those pages with the driver for the transfer, and when the caller frees them -- a
// picture of a megabyte and more is an mmap of its own -- the unmapping of a registered range evicts the process's queues for a
// moment: every later call then waited 20-30 ms (in steps of ten) on its first synchronisation.  Seen on 2x2-chroma.jpeg (1.3 MB of
// RGB) through mjx_decode, not on lena.jpeg (0.8 MB: under the runtime's own staging limit); round 5, tools/probes/oneshot_2x2.py.
static int copy_from_device(mjx_ctx *ctx, void *host, const void *dev, size_t total)
{
    if (total < (size_t(512) << 10)) {                        // (small: the runtime stages these itself)
        HIPOK(hipMemcpy(host, dev, total, hipMemcpyDeviceToHost));
        return MJX_OK;
    }
    std::lock_guard<std::mutex> lk(ctx->rgb_pin_mu);
    const size_t kPiece = size_t(8) << 20;
    if (!ctx->rgb_pin) {
        void *hp = nullptr;
        if (hipHostMalloc(&hp, kPiece, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); hp = nullptr; }
        ctx->rgb_pin = static_cast<uint8_t *>(hp);
    }
    if (!ctx->rgb_pin) {
        HIPOK(hipMemcpy(host, dev, total, hipMemcpyDeviceToHost));
        return MJX_OK;
    }
    for (size_t at = 0; at < total; at += kPiece) {
        const size_t n = std::min(kPiece, total - at);
        HIPOK(hipMemcpy(ctx->rgb_pin, static_cast<const uint8_t *>(dev) + at, n, hipMemcpyDeviceToHost));
        std::memcpy(static_cast<uint8_t *>(host) + at, ctx->rgb_pin, n);
    }
    return MJX_OK;
}

extern "C" int mjx_batch_copy_rgb(mjx_batch *b, size_t iu, uint8_t *host_rgb)
{
    if (b && !b->parts.empty()) { mjx_batch *pb; size_t pi; return route(b, iu, pb, pi) ? mjx_batch_copy_rgb(pb, pi, host_rgb) : MJX_ERR_INVALID_ARG; }
    return guarded([&]() -> int {
    size_t i;
    if (!visible_index(b, iu, i) || !host_rgb) return MJX_ERR_INVALID_ARG;
    const ImageInfo &inf = b->info[i];
    if (inf.status != MJX_OK) return inf.status;
    HIPOK(hipSetDevice(b->ctx->device));
    { const int rcs = sync_streams(b); if (rcs != MJX_OK) return rcs; }
    return copy_from_device(b->ctx, host_rgb, b->d_rgb + inf.rgb_off, size_t(inf.rgb_bytes));
    });
}

extern "C" int mjx_batch_copy_coefs(mjx_batch *b, size_t iu, int16_t *host_coefs, size_t cap_blocks, size_t *nblocks)
{
    if (b && !b->parts.empty()) { mjx_batch *pb; size_t pi; return route(b, iu, pb, pi) ? mjx_batch_copy_coefs(pb, pi, host_coefs, cap_blocks, nblocks) : MJX_ERR_INVALID_ARG; }
    return guarded([&]() -> int {
    size_t i;
    if (!visible_index(b, iu, i) || !host_coefs) return MJX_ERR_INVALID_ARG;
    const ImageInfo &inf = b->info[i];
    if (inf.status != MJX_OK) return inf.status;
    if (nblocks) *nblocks = size_t(inf.nblocks);
    if (cap_blocks < inf.nblocks) return MJX_ERR_INVALID_ARG;
    if (!b->decoded_entropy) return MJX_ERR_INVALID_ARG;
    if (!b->opts.keep_coefs && int(inf.chunk) != b->last_chunk_resident) return MJX_ERR_INVALID_ARG;
    if (inf.planar) return MJX_ERR_INVALID_ARG;        // (a multi-scan picture of a batch without keep_coefs: its coefficients only exist scan by scan)
    HIPOK(hipSetDevice(b->ctx->device));
    { const int rcs = sync_streams(b); if (rcs != MJX_OK) return rcs; }
    // expand the compact stream (entries + tile offsets + predicted DCs) into dense zig-zag blocks on the host
    std::vector<uint32_t> eoff(size_t(inf.ntiles) + 1);
    const bool sec = b->resident_second && !b->opts.keep_coefs;
    const uint32_t *tile_eoff = sec ? b->alt.d_tile_eoff : b->d_tile_eoff, *entries = sec ? b->alt.d_entries : b->d_entries;
    const int32_t *dcs = sec ? b->alt.d_dc : b->d_dc;
    { const int rcc = copy_from_device(b->ctx, eoff.data(), tile_eoff + inf.tile_off, eoff.size() * 4); if (rcc != MJX_OK) return rcc; }
    std::vector<int32_t> dc(size_t(inf.nblocks));
    { const int rcc = copy_from_device(b->ctx, dc.data(), dcs + inf.coef_off, dc.size() * sizeof(int32_t)); if (rcc != MJX_OK) return rcc; }
    std::memset(host_coefs, 0, size_t(inf.nblocks) * 128);
    auto place = [&](uint32_t first, uint32_t e) {
        const uint64_t blk = first + (((e >> 22) - first) & 0xffu);
        if (blk < inf.nblocks) host_coefs[blk * 64 + ((e >> 16) & 63)] = int16_t(e & 0xffff);
    };
    if (inf.ent_rows) {
        // quad-interleaved: a tile runs from (subsequence, entry) of its own offset to that of the next one, through the whole
        // runs (the 16-bit group counts at the head of the region) of the subsequences between
        std::vector<uint32_t> ent(size_t(inf.ent_cap));
        { const int rcc = copy_from_device(b->ctx, ent.data(), entries + inf.ent_off, ent.size() * 4); if (rcc != MJX_OK) return rcc; }
        const uint32_t *runs = ent.data();                   // one run word per subsequence: first group, end group, label offset (mjx_kernels.h)
        const uint32_t *col = ent.data() + inf.ent_hdr;
        const uint32_t cap = inf.ent_rows * 8u;
        const uint64_t ncols = (inf.ent_cap - inf.ent_hdr) / cap;
        for (uint32_t t = 0; t < inf.ntiles; t++) {
            const uint32_t first = t * inf.tile_blocks;
            const uint32_t s0 = eoff[t] / cap, j0 = eoff[t] % cap, s1 = eoff[t + 1] / cap, j1 = eoff[t + 1] % cap;
            if (s0 > s1 || s1 >= ncols) return MJX_ERR_DEVICE;
            for (uint32_t s = s0; s <= s1; s++) {
                const uint32_t lo = s == s0 ? j0 : run_first(runs[s]) * 8u, hi = s == s1 ? j1 : std::min<uint32_t>(run_end(runs[s]) * 8u, cap);
                const uint32_t label = run_label(runs[s]) << 22;
                for (uint32_t j = lo; j < hi; j++) {
                    const uint32_t e = col[stream_phys(s, j, inf.ent_rows)];
                    if ((e >> 16) & 63u) place(first, e + label);      // (position 0: a null entry)
                }
            }
        }
    } else {
        const uint32_t nent = eoff.back();
        if (nent > inf.ent_cap) return MJX_ERR_DEVICE;
        std::vector<uint32_t> ent(size_t(nent) + 1);
        if (nent) { const int rcc = copy_from_device(b->ctx, ent.data(), entries + inf.ent_off, size_t(nent) * 4); if (rcc != MJX_OK) return rcc; }
        for (uint32_t t = 0; t < inf.ntiles; t++) {
            const uint32_t first = t * inf.tile_blocks;
            for (uint32_t j = eoff[t]; j < eoff[t + 1] && j < nent; j++) place(first, ent[j]);
        }
    }
    for (size_t k = 0; k < dc.size(); k++) host_coefs[k * 64] = int16_t(dc[k]);
    return MJX_OK;
    });
}

extern "C" int mjx_batch_compare_rgb(mjx_batch *a, const size_t *ia, mjx_batch *b, const size_t *ib, size_t n,
                                     uint32_t *max_abs_diff, uint64_t *n_diff)
{
    return guarded([&]() -> int {
        if (!a || !b || (n && (!ia || !ib || !max_abs_diff))) return MJX_ERR_INVALID_ARG;
        if (a->ctx->device != b->ctx->device) return MJX_ERR_INVALID_ARG;
        if (n == 0) return MJX_OK;
        if (n > 0x7fffffffu / 64) return MJX_ERR_INVALID_ARG;
        HIPOK(hipSetDevice(a->ctx->device));
        HIPOK(hipDeviceSynchronize());                       // (every stream of either batch, parts included)
        std::vector<RgbPair> pairs(n);
        std::vector<char> bad(n, 0);
        uint64_t max_bytes = 0;
        for (size_t k = 0; k < n; k++) {
            mjx_batch *pa, *pb;
            size_t ka, kb, xa, xb;
            if (!route(a, ia[k], pa, ka) || !route(b, ib[k], pb, kb)) return MJX_ERR_INVALID_ARG;
            if (!visible_index(pa, ka, xa) || !visible_index(pb, kb, xb)) return MJX_ERR_INVALID_ARG;
            const ImageInfo &fa = pa->info[xa], &fb = pb->info[xb];
            pairs[k] = RgbPair{pa->d_rgb + fa.rgb_off, pb->d_rgb + fb.rgb_off, fa.rgb_bytes};
            if (fa.status != MJX_OK || fb.status != MJX_OK || fa.width != fb.width || fa.height != fb.height) {
                bad[k] = 1;
                pairs[k].bytes = 0;
            }
            max_bytes = std::max<uint64_t>(max_bytes, pairs[k].bytes);
        }
        RgbPair *d_pairs = nullptr;
        uint32_t *d_max = nullptr;
        unsigned long long *d_cnt = nullptr;
        std::vector<uint32_t> hmax(n, 0);
        std::vector<unsigned long long> hcnt(n, 0);
        int rc = MJX_OK;
        if (hipMalloc(&d_pairs, n * sizeof(RgbPair)) != hipSuccess || hipMalloc(&d_max, n * 4) != hipSuccess || hipMalloc(&d_cnt, n * 8) != hipSuccess)
            rc = MJX_ERR_DEVICE;
        if (rc == MJX_OK && (hipMemcpy(d_pairs, pairs.data(), n * sizeof(RgbPair), hipMemcpyHostToDevice) != hipSuccess ||
                             hipMemset(d_max, 0, n * 4) != hipSuccess || hipMemset(d_cnt, 0, n * 8) != hipSuccess))
            rc = MJX_ERR_DEVICE;
        if (rc == MJX_OK) {
            launch_rgb_compare(a->ctx->stream, uint32_t(n), max_bytes, d_pairs, d_max, d_cnt);
            if (hipStreamSynchronize(a->ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess ||
                hipMemcpy(hmax.data(), d_max, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(hcnt.data(), d_cnt, n * 8, hipMemcpyDeviceToHost) != hipSuccess)
                rc = MJX_ERR_DEVICE;
        }
        (void)hipFree(d_pairs); (void)hipFree(d_max); (void)hipFree(d_cnt);
        if (rc != MJX_OK) { (void)hipGetLastError(); return rc; }
        for (size_t k = 0; k < n; k++) {
            max_abs_diff[k] = bad[k] ? 0xffffffffu : hmax[k];
            if (n_diff) n_diff[k] = bad[k] ? 0 : uint64_t(hcnt[k]);
        }
        return MJX_OK;
    });
}

extern "C" int mjx_batch_bytes(const mjx_batch *b, uint64_t *scan_bytes, uint64_t *rgb_bytes, uint64_t *coef_bytes,
                               uint64_t *pixels)
{
    return guarded([&]() -> int {
    if (!b) return MJX_ERR_INVALID_ARG;
    if (!b->parts.empty()) {
        uint64_t acc[4] = {0, 0, 0, 0};
        for (const mjx_batch *part : b->parts) {
            uint64_t v[4] = {0, 0, 0, 0};
            const int rc = mjx_batch_bytes(part, &v[0], &v[1], coef_bytes ? &v[2] : nullptr, &v[3]);
            if (rc != MJX_OK) return rc;
            for (int k = 0; k < 4; k++) acc[k] += v[k];
        }
        if (scan_bytes) *scan_bytes = acc[0];
        if (rgb_bytes) *rgb_bytes = acc[1];
        if (coef_bytes) *coef_bytes = acc[2];
        if (pixels) *pixels = acc[3];
        return MJX_OK;
    }
    if (scan_bytes) *scan_bytes = b->scan_bytes;
    if (rgb_bytes) *rgb_bytes = b->rgb_bytes;
    if (pixels) *pixels = b->pixels;
    if (coef_bytes) {
        // intermediate coefficient representation: 4 bytes per stream entry + 2 bytes of DC per block
        uint64_t total = 0;
        if (b->decoded_entropy) {
            HIPOK(hipSetDevice(b->ctx->device));
            { const int rcs = sync_streams(b); if (rcs != MJX_OK) return rcs; }
            std::vector<uint32_t> cnt(b->info.size());
            if (!cnt.empty()) HIPOK(hipMemcpy(cnt.data(), b->d_img_entries, cnt.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < cnt.size(); i++)
                if (b->info[i].status == MJX_OK) total += uint64_t(cnt[i]) * 4 + b->info[i].nblocks * 4;
        }
        *coef_bytes = total;
    }
    return MJX_OK;
    });
}

extern "C" int mjx_batch_unconverged_runs(const mjx_batch *b, uint64_t *runs)
{
    if (!b || !runs) return MJX_ERR_INVALID_ARG;
    *runs = 0;
    if (!b->parts.empty()) {
        for (const mjx_batch *part : b->parts) {
            uint64_t v = 0;
            const int rc = mjx_batch_unconverged_runs(part, &v);
            if (rc != MJX_OK) return rc;
            *runs += v;
        }
        return MJX_OK;
    }
    return guarded([&]() -> int {
    HIPOK(hipSetDevice(b->ctx->device));
    { const int rcs = sync_streams(b); if (rcs != MJX_OK) return rcs; }
    std::vector<uint32_t> v(std::max<size_t>(b->chunks.size(), 1));
    HIPOK(hipMemcpy(v.data(), b->d_unconv, v.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t x : v) *runs += x;
    return MJX_OK;
    });
}

extern "C" int mjx_batch_geometry(const mjx_batch *b, uint64_t *subsequences, uint64_t *blocks, uint64_t *chunks)
{
    if (!b) return MJX_ERR_INVALID_ARG;
    if (!b->parts.empty()) {
        uint64_t acc[3] = {0, 0, 0};
        for (const mjx_batch *part : b->parts) {
            uint64_t v[3];
            (void)mjx_batch_geometry(part, &v[0], &v[1], &v[2]);
            for (int k = 0; k < 3; k++) acc[k] += v[k];
        }
        if (subsequences) *subsequences = acc[0];
        if (blocks) *blocks = acc[1];
        if (chunks) *chunks = acc[2];
        return MJX_OK;
    }
    uint64_t ns = 0, nb = 0;
    // scans de-stuffed on the device: the host planned with the stuffed length as an upper bound, the exact number of
    // subsequences is in the device's copy of the images (k_destuff_prefix / k_restart_geometry)
    std::vector<DevImage> dev_images;
    // (the upload-time kernels run on the upload stream or -- groups of a pipelined list, upload_kernels_apart -- on a decode
    // stream behind the copies; b->uploaded is recorded behind them on whichever stream ran them)
    const bool upload_done = b->has_stuffed && !b->himages.empty() && hipSetDevice(b->ctx->device) == hipSuccess &&
                             (b->uploaded ? hipEventSynchronize(b->uploaded) : hipStreamSynchronize(b->ctx->upload)) == hipSuccess;
    if (upload_done) {
        dev_images.resize(b->himages.size());
        if (hipMemcpy(dev_images.data(), b->d_images, dev_images.size() * sizeof(DevImage), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); dev_images.clear(); }
    }
    for (size_t i = 0; i < b->info.size(); i++) {
        if (b->info[i].status != MJX_OK) continue;
        ns += dev_images.empty() ? b->himages[i].himg.nsub : dev_images[i].himg.nsub;
        if (b->info[i].role != 1) nb += b->info[i].nblocks;
    }
    if (subsequences) *subsequences = ns;
    if (blocks) *blocks = nb;
    if (chunks) *chunks = b->chunks.size();
    return MJX_OK;
}

extern "C" int mjx_batch_kernel_ms(mjx_batch *b, double ms[MJX_K_COUNT], uint64_t launches[MJX_K_COUNT], int reset)
{
    return guarded([&]() -> int {
    if (!b) return MJX_ERR_INVALID_ARG;
    if (!b->parts.empty()) {
        double acc[MJX_K_COUNT] = {0};
        uint64_t cnt[MJX_K_COUNT] = {0};
        for (mjx_batch *part : b->parts) {
            double m[MJX_K_COUNT];
            uint64_t l[MJX_K_COUNT];
            const int rc = mjx_batch_kernel_ms(part, m, l, reset);
            if (rc != MJX_OK) return rc;
            for (int k = 0; k < MJX_K_COUNT; k++) { acc[k] += m[k]; cnt[k] += l[k]; }
        }
        for (int k = 0; k < MJX_K_COUNT; k++) { if (ms) ms[k] = acc[k]; if (launches) launches[k] = cnt[k]; }
        return MJX_OK;
    }
    HIPOK(hipSetDevice(b->ctx->device));
    { const int rcs = sync_streams(b); if (rcs != MJX_OK) return rcs; }
    collect_events(b);
    for (int k = 0; k < MJX_K_COUNT; k++) {
        if (ms) ms[k] = b->ms[k];
        if (launches) launches[k] = b->launches[k];
        if (reset) { b->ms[k] = 0; b->launches[k] = 0; }
    }
    return MJX_OK;
    });
}

extern "C" int mjx_decode_scans(mjx_ctx *ctx, const mjx_scan_desc *descs, size_t n, const mjx_opts *opts,
                                uint8_t **rgb_dev, int *status, mjx_batch **out)
{
    return guarded([&]() -> int {
    if (!out) return MJX_ERR_INVALID_ARG;
    int rc = mjx_batch_create(ctx, descs, n, opts, out, nullptr);
    if (rc != MJX_OK) return rc;
    rc = mjx_batch_decode(*out, MJX_STAGE_ALL);
    if (rc == MJX_OK) rc = mjx_batch_wait(*out);
    if (rc != MJX_OK) { mjx_batch_free(*out); *out = nullptr; return rc; }
    for (size_t i = 0; i < n; i++) {
        if (status) status[i] = mjx_batch_status(*out, i);
        if (rgb_dev) {
            void *p = nullptr;
            (void)mjx_batch_rgb_device(*out, i, &p, nullptr);
            rgb_dev[i] = static_cast<uint8_t *>(p);
        }
    }
    return MJX_OK;
    });
}

namespace {
// processors this process may use: the hardware's count, cut down to the affinity mask and to the cgroup's CPU quota
unsigned usable_processors()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1u, unsigned(CPU_COUNT(&set))));
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {              // cgroup v2: "<quota> <period>" or "max <period>"
        long long quota = 0, period = 0;
        if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
            n = std::min(n, std::max(1u, unsigned((quota + period - 1) / period)));
        std::fclose(f);
    } else {
        long long quota = -1, period = 0;                                     // cgroup v1
        if (FILE *q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(q, "%lld", &quota) != 1) quota = -1; std::fclose(q); }
        if (FILE *pf = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(pf, "%lld", &period) != 1) period = 0; std::fclose(pf); }
        if (quota > 0 && period > 0) n = std::min(n, std::max(1u, unsigned((quota + period - 1) / period)));
    }
    return n;
}
}   // namespace

// The outer surface for a list of files, pipelined ("double-buffered chunks: H2D of chunk k+1 overlaps the
// kernels of chunk k"; the seam it replaces is jpeg/mod.rs:371-417, parse -> de-stuff -> decode of one file).  The files are
// cut into groups of compressed data (12 MB first, doubling up to 96 MB: small groups give the device work early, large ones keep
// the per-group launch overhead low).  Host threads walk the markers and de-stuff group after group into
// the context's pinned arena; as soon as a group is parsed the calling thread plans it, enqueues its upload (one DMA
// transfer per group, on the upload stream) and its kernels (on the decode streams, behind the upload's event), and
// turns to the next group -- so parsing of group g+2, the transfer of group g+1 and the kernels of group g run at the
// same time, and the call takes about as long as the slowest of the three (on PCIe Gen5: the transfer).
extern "C" int mjx_decode_batch(mjx_ctx *ctx, const uint8_t *const *jpegs, const size_t *lens, size_t n, const mjx_opts *opts,
                                unsigned threads, uint8_t **rgb_dev, int *status, mjx_batch **out)
{
    return guarded([&]() -> int {
    if (!ctx || !out || ((!jpegs || !lens) && n)) return MJX_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> serial(ctx->batch_mu);        // the pinned arena is shared state of the context
    mjx_opts o{};
    if (opts) o = *opts;
    const bool timing = std::getenv("MJX_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    // parse threads: half the processors the process may use (cgroup quota included), the rest is for the calling thread
    // and the runtime's own threads -- with a quota of 16 processors 8 threads were faster than 16, 24 or 32
    unsigned nt = threads ? threads : std::max(4u, usable_processors() / 2);
    nt = std::max(1u, std::min(nt, 32u));
    nt = unsigned(std::min<size_t>(nt, std::max<size_t>(n, 1)));
    if (std::getenv("MJX_TIMING")) std::fprintf(stderr, "[mjx] decode_batch on device %d: %zu files, %u parse threads\n", ctx->device, n, nt);
    // groups: small ones first so that the device has work early (12 MB, doubling), then group_bytes each.  Consecutive
    // transfers start 0.3 ms apart whatever stream they are on (measured with a copy stream of their own, and with three),
    // so long lists get larger groups: 192 MB from 1.5 GB on, 96 MB below.  Smaller groups again at the end of the list
    // (so that less is left to decode when the last transfer has landed) were tried: no gain.  Kept coefficients (and lists
    // of up to eight files) take the whole list as one group (one batch, as mjx_batch_create would build it); lists de-stuffed on
    // the device are cut into groups like the others.
    size_t total_bytes = 0;
    for (size_t i = 0; i < n; i++) total_bytes += lens[i];
    if (o.device_destuff == MJX_DESTUFF_AUTO) {
        // who de-stuffs: on a list of this size the GPU (the host threads then only walk the markers and copy: 13.9 against
        // 15.3 ms per 512 4K files, 45.3 against 46.6 per 2048); a few pictures are faster without the three upload kernels
        size_t auto_mb = 64;
        if (const char *e = std::getenv("MJX_AUTO_DESTUFF_MB")) auto_mb = size_t(std::max(0L, std::atol(e)));
        o.device_destuff = (!o.strict_ref && total_bytes >= (auto_mb << 20)) ? MJX_DESTUFF_DEVICE : MJX_DESTUFF_HOST;
    }
    size_t group_bytes = size_t(total_bytes >= (size_t(1536) << 20) ? 192 : 96) << 20;
    if (const char *e = std::getenv("MJX_GROUP_MB")) { const long v = std::atol(e); if (v > 0) group_bytes = size_t(v) << 20; }
    const bool single = o.keep_coefs || n <= 8;
    // Group sizes: a ramp up from `first_bytes` (times `grow` per group: the device has work early, and the host -- which must
    // have a group parsed and planned before its transfer can start -- keeps ahead of the DMA engine while the groups are
    // small), then `group_bytes`.  A ramp down at the end of the list (MJX_GROUP_TAPER=1: halving, so that less is left to
    // decode when the last transfer has landed) was measured again in round 4, with the other knobs (first group 6 / 8 / 12 MB,
    // growth 1.5 / 1.6 / 2, groups of 64 / 96 / 128 MB): 13.5 .. 14.8 ms per 512 4K files whatever the schedule -- what the
    // call waits for is the transfer (10.5 ms beside the kernels), the first group's parse and the last group's kernel chain.
    size_t first_bytes = std::min(group_bytes / 8, size_t(12) << 20);
    double grow = 2.0;
    bool taper = false;
    if (const char *e = std::getenv("MJX_GROUP_FIRST_MB")) { const double v = std::atof(e); if (v > 0) first_bytes = size_t(v * 1048576.0); }
    if (const char *e = std::getenv("MJX_GROUP_GROW")) { const double v = std::atof(e); if (v >= 1.0) grow = v; }
    // (the ramp below must end: a growth of 1.0 or a first group that rounds down to nothing would push sizes for ever)
    grow = std::max(grow, 1.05);
    first_bytes = std::max(first_bytes, size_t(4096));        // (and the ramp itself is capped at 64 steps)
    if (const char *e = std::getenv("MJX_GROUP_TAPER")) taper = std::atoi(e) != 0;
    std::vector<size_t> gfirst{0};
    if (!single) {
        std::vector<size_t> sizes, up, down;                // target bytes per group, front to back
        for (double t = double(first_bytes); size_t(t) < group_bytes && up.size() < 64; t *= grow) up.push_back(size_t(t));
        if (taper) for (size_t t = group_bytes / 2; t >= first_bytes && t > 0; t /= 2) down.push_back(t);
        size_t up_sum = 0, down_sum = 0;
        for (size_t t : up) up_sum += t;
        for (size_t t : down) down_sum += t;
        if (total_bytes < up_sum + down_sum + group_bytes / 2) {       // a short list: the ramp up, as far as the bytes go
            size_t left = total_bytes;
            for (size_t k = 0; left > 0; k++) {
                const size_t want = k < up.size() ? up[k] : group_bytes;
                sizes.push_back(std::min(want, left));
                left -= sizes.back();
            }
        } else {
            sizes = up;
            const size_t mid = total_bytes - up_sum - down_sum;
            const size_t k = std::max<size_t>(1, (mid + group_bytes / 2) / group_bytes);   // equal groups of about group_bytes in the middle
            for (size_t j = 0; j < k; j++) sizes.push_back(mid / k + (j + 1 == k ? mid % k : 0));
            for (size_t t : down) sizes.push_back(t);
        }
        size_t acc = 0, gi = 0;
        for (size_t i = 0; i < n; i++) {
            acc += lens[i];
            if (gi < sizes.size() && acc >= sizes[gi] && i + 1 < n) {
                gfirst.push_back(i + 1);
                acc = 0;
                gi++;
            }
        }
    }
    gfirst.push_back(n);
    const size_t ngroups = gfirst.size() - 1;
    // one arena for all the de-stuffed scans (a slice of len + 64 bytes per file; what does not fit -- files with many
    // restart markers, multi-scan files -- is allocated by the parser): hundreds of megabyte-sized malloc / free pairs cost
    // several times the parsing.  Pinned: the scans are uploaded straight from it by DMA.
    std::vector<size_t> slice(n + 1, 0);
    for (size_t i = 0; i < n; i++) slice[i + 1] = slice[i] + ((lens[i] + 64 + 63) & ~size_t(63));
    if (ctx->parse_arena_cap < slice[n]) {
        HIPOK(hipSetDevice(ctx->device));
        if (ctx->parse_arena) (void)hipHostFree(ctx->parse_arena);
        ctx->parse_arena = nullptr;
        ctx->parse_arena_cap = 0;
        void *pa = nullptr;
        if (hipHostMalloc(&pa, slice[n] + slice[n] / 4 + 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return MJX_ERR_NOMEM; }
        ctx->parse_arena = static_cast<uint8_t *>(pa);
        ctx->parse_arena_cap = slice[n] + slice[n] / 4 + 64;
    }
    uint8_t *const arena = ctx->parse_arena;
    PinnedBump pin;
    {
        const size_t want = std::min<size_t>(size_t(256) << 20, n * (size_t(20) << 10) + (size_t(1) << 20));
        if (ctx->pin_small_cap < want) {
            if (ctx->pin_small) (void)hipHostFree(ctx->pin_small);
            ctx->pin_small = nullptr;
            ctx->pin_small_cap = 0;
            void *pp = nullptr;
            if (hipHostMalloc(&pp, want, hipHostMallocDefault) == hipSuccess) { ctx->pin_small = static_cast<uint8_t *>(pp); ctx->pin_small_cap = want; }
            else (void)hipGetLastError();
        }
        pin.base = ctx->pin_small;
        pin.cap = ctx->pin_small_cap;
    }
    std::vector<mjx_scan_desc> descs(n);
    std::vector<int> prc(n, MJX_OK);
    std::vector<std::vector<ImagePlan>> file_plans(n);
    // host side: files are parsed in list order by a pool of threads; a group is ready when all its files are
    std::atomic<size_t> next{0};
    std::vector<std::atomic<size_t>> done(ngroups);
    for (auto &d : done) d.store(0);
    std::vector<uint32_t> group_of(n);
    for (size_t g = 0; g < ngroups; g++)
        for (size_t i = gfirst[g]; i < gfirst[g + 1]; i++) group_of[i] = uint32_t(g);
    std::mutex mu;
    std::condition_variable cv;
    auto work = [&] {
        for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            int rc;
            try {
                rc = mjx::parse_into(jpegs[i], lens[i], &o, &descs[i], arena + slice[i], slice[i + 1] - slice[i]);
            } catch (...) {
                rc = MJX_ERR_NOMEM;
            }
            prc[i] = rc;
            if (rc != MJX_OK) std::memset(&descs[i], 0, sizeof descs[i]);
            try {
                plan_input(descs[i], o, file_plans[i]);          // geometry + decode tables, also on the worker
            } catch (...) {
                file_plans[i].clear();
            }
            const size_t g = group_of[i];
            if (done[g].fetch_add(1) + 1 == gfirst[g + 1] - gfirst[g]) {
                std::lock_guard<std::mutex> lk(mu);
                cv.notify_all();
            }
        }
    };
    std::vector<std::thread> pool;
    struct Joiner {                                            // (the workers borrow this frame: never leave it before they are done)
        std::vector<std::thread> &p;
        std::atomic<size_t> &next;
        size_t n;
        ~Joiner() { next.store(n); for (std::thread &t : p) if (t.joinable()) t.join(); }
    } joiner{pool, next, n};
    for (unsigned t = 0; t < nt; t++) pool.emplace_back(work);

    mjx_batch *dir = new (std::nothrow) mjx_batch;
    if (!dir) return MJX_ERR_NOMEM;
    struct Owner { mjx_batch *b; ~Owner() { if (b) { (void)hipDeviceSynchronize(); release(b); } } } owner{dir};
    dir->ctx = ctx;
    dir->opts = o;
    dir->part_index.resize(n);
    dir->h_mismatch = pinned_get(ctx, ngroups * 8 * kMisWords * sizeof(uint32_t), &dir->h_mismatch_bytes);
    if (!dir->h_mismatch) return MJX_ERR_NOMEM;
    int rc = MJX_OK;
    for (size_t g = 0; g < ngroups && rc == MJX_OK; g++) {
        const size_t f0 = gfirst[g], cnt = gfirst[g + 1] - f0;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done[g].load() == cnt; });
        }
        const double t_parsed = since();
        mjx_batch *part = nullptr;
        std::vector<ImagePlan> plans;
        plans.reserve(cnt);
        for (size_t i = 0; i < cnt; i++) {
            if (file_plans[f0 + i].empty()) { rc = MJX_ERR_NOMEM; break; }
            for (ImagePlan &pl : file_plans[f0 + i]) plans.push_back(std::move(pl));
        }
        if (rc != MJX_OK) break;
        constexpr size_t kPinnedPerPart = 8 * kMisWords;
        rc = build_batch(ctx, plans, o, nullptr, 1, &part, nullptr, true, dir->h_mismatch + g * kPinnedPerPart, kPinnedPerPart, &pin, ngroups == 1, (g & 1) != 0);
        if (rc != MJX_OK) break;
        dir->parts.push_back(part);
        if (part->visible.size() != cnt) { rc = MJX_ERR_DEVICE; break; }
        for (size_t i = 0; i < cnt; i++) dir->part_index[f0 + i] = {uint32_t(g), uint32_t(i)};
        const double t_built = since();
        rc = mjx_batch_decode(part, MJX_STAGE_ALL);
        if (timing) std::fprintf(stderr, "[mjx] group %zu: %zu files, parsed at %.2f ms, uploaded+planned at %.2f, enqueued at %.2f\n", g, cnt, t_parsed, t_built, since());
    }
    joiner.next.store(n);
    for (std::thread &t : pool) if (t.joinable()) t.join();
    for (size_t i = 0; i < n; i++) mjx_free_scan(&descs[i]);
    if (rc == MJX_OK) rc = mjx_batch_wait(dir);
    if (timing) std::fprintf(stderr, "[mjx] decode_batch: %zu files in %zu groups, %u threads, %.2f ms\n", n, ngroups, nt, since());
    if (rc != MJX_OK) return rc;
    for (size_t i = 0; i < n; i++) {
        const int si = prc[i] != MJX_OK ? prc[i] : mjx_batch_status(dir, i);
        if (status) status[i] = si;
        if (rgb_dev) {
            void *p = nullptr;
            if (si == MJX_OK) (void)mjx_batch_rgb_device(dir, i, &p, nullptr);
            rgb_dev[i] = static_cast<uint8_t *>(p);
        }
    }
    owner.b = nullptr;
    *out = dir;
    return MJX_OK;
    });
}

// ---- one-shot surface ------------------------------------------------------------------------------
extern "C" int mjx_decode(const uint8_t *jpeg, size_t len, const mjx_opts *opts, mjx_image *out)
{
    return guarded([&]() -> int {
    if (!out) return MJX_ERR_INVALID_ARG;
    out->width = out->height = 0;
    out->rgb = nullptr;
    mjx_scan_desc d;
    int rc = mjx_parse(jpeg, len, opts, &d);
    if (rc != MJX_OK) return rc;
    mjx_ctx *ctx = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_default_mu);
        if (!g_default_ctx) {
            rc = mjx_ctx_create(0, &g_default_ctx);
            if (rc != MJX_OK) { mjx_free_scan(&d); return rc; }
        }
        ctx = g_default_ctx;
    }
    std::lock_guard<std::mutex> lk(g_default_mu);
    mjx_batch *b = nullptr;
    int st = MJX_OK;
    rc = mjx_batch_create(ctx, &d, 1, opts, &b, &st);
    mjx_free_scan(&d);
    if (rc != MJX_OK) return rc;
    if (st != MJX_OK) { mjx_batch_free(b); return st; }
    rc = mjx_batch_decode(b, MJX_STAGE_ALL);
    if (rc == MJX_OK) rc = mjx_batch_wait(b);
    if (rc == MJX_OK) rc = mjx_batch_status(b, 0);
    if (rc == MJX_OK) {
        const ImageInfo &inf = b->info[b->visible[0]];          // (a multi-scan file keeps its scans in front of the picture)
        out->width = inf.width;
        out->height = inf.height;
        out->rgb = static_cast<uint8_t *>(std::malloc(size_t(inf.rgb_bytes) + 1));
        if (!out->rgb) rc = MJX_ERR_NOMEM;
        else rc = mjx_batch_copy_rgb(b, 0, out->rgb);
        if (rc != MJX_OK) { std::free(out->rgb); out->rgb = nullptr; }
    }
    mjx_batch_free(b);
    return rc;
    });
}

extern "C" void mjx_free_image(mjx_image *img)
{
    if (!img) return;
    std::free(img->rgb);
    img->rgb = nullptr;
}

extern "C" const char *mjx_version(void)
{
    static char v[160];
    static std::once_flag once;
    std::call_once(once, [] {
        std::snprintf(v, sizeof v, "mjx 0.2 gfx950 subseq_bits=%d..%d checkpoint_bits=%d lut_primary_bits=%d huff_wg_lanes=%d merge_wg_lanes=%d",
                      kSubseqBits, kSubseqBits * 5 / 4, kCpBits, kLutPrimaryBits, kHuffWg, kMergeWg);
    });
    return v;
}
