// mjx_kernels.hip -- gfx950 kernels of the baseline-JPEG decode path.
//
//   stage A (entropy, replaces HuffmanDecoder::next_block + the MCU loop, reference
//            src/jpeg/huffman.rs:146-254 and src/jpeg/decoder.rs:195-215):
//     k_huff_spec   speculative decode of every 512-byte subsequence, recording exit states and checkpoints
//     k_huff_merge (+ k_huff_merge_tail)  one synchronisation round: re-decode the subsequences whose entry changed
//                   until they merge with their previous decode
//     k_huff_scan   per-image exclusive scan of completed-block / stream-entry counts; fences unconverged chunks
//     k_huff_write  final decode from the synchronised entry states into the compact coefficient stream
//     k_dc_sums / k_dc_apply (+ _t<BPM>, k_dc_restart)   DC prediction (decoder.rs:208-210) as a per-component prefix sum
//     k_planar_count / _offsets / _copy   multi-scan files only: component streams -> the picture's stream in MCU order
//     k_destuff_count / _scatter          optional: the FF00 -> FF compaction of jpeg/mod.rs:371-385 at upload
//   stage B (pixels, replaces decoder.rs:227-235, 239-331 and src/transform.rs:55-87):
//     k_idct_color  dequant + un-zigzag + 8x8 float IDCT + chroma replication + YCbCr->RGB + packed store
//     k_ref_color   REF_COMPAT layout only: the reference's plane-wise colour step
//
// Without restart markers (the reference panics on DRI, jpeg/mod.rs:424-428; files that have them are cut at the markers
// as well) intra-image parallelism comes from the self-synchronisation of Huffman codes: a lane that starts decoding
// at an arbitrary bit with a guessed state converges to the true symbol boundaries; entry states are iterated to a
// fixed point.
#include <type_traits>
#include <hip/hip_runtime.h>

#include "mjx_kernels.h"

#include <algorithm>

namespace mjx {

// ------------------------------------------------------------------------------------------------
// stage A building blocks
// ------------------------------------------------------------------------------------------------
// The image's de-stuffed scan in HBM, **lane-interleaved**: the scan is cut into the subsequences the lanes decode, every
// subsequence into 16-byte pieces, and piece k of subsequence s lies at (k * scan_cols + s) * 16 -- row k holds piece k of
// all subsequences side by side.  The 64 lanes of a wave advance through their subsequences at nearly the same pace, so
// their window refills read the same row: 1 KiB of consecutive bytes, eight lines each used by eight lanes.  (With the
// scan stored linearly a lane's 16-byte reads were 512 bytes apart from its neighbours': every lane pulled its own
// 128-byte line through L2 four or five times, once per refill -- FETCH_SIZE was 6.3x the scan -- and waited for HBM
// each time.)  A column carries kLookPieces more pieces than the subsequence has: a copy of the bytes that follow it
// in the stream (0xAA past the end of the scan, huffman.rs:236-246), because a lane decodes the symbol that straddles
// its end and stages up to 48 bytes of look-ahead.  Built once per upload by k_scan_interleave.
// Inside the kernels all bit positions are relative to the lane's subsequence; the states kept in HBM stay relative to the scan.
struct LaneBits {
    const unsigned char *base;      // wave-uniform: the image's region of the pool
    uint32_t col;                   // byte offset of the lane's column inside a row: subsequence index * 16
    uint32_t row_stride;            // bytes per row: scan_cols * 16
    // (cached loads: a 128-byte line holds the pieces of eight lanes, which ask for them at different times -- streamed past L2
    // these loads cost the write pass 0.35 ms per step; what is streamed are the pictures and stage B's reads of the stream)
    __device__ __forceinline__ uint4 piece(uint32_t k) const { return *reinterpret_cast<const uint4 *>(base + (col + k * row_stride)); }
};

// Checkpoints live in HBM in blocks of 256 consecutive subsequences; inside a block they are row-major by checkpoint
// (row k = the two words of checkpoint k -- state | blocks, stream entries -- of the block's 256 subsequences, one 8-byte
// pair each, rows 2 KiB apart), so the lanes of a wave touch adjacent pairs of a row and a workgroup's checkpoints stay
// within one region.  In merge rounds the previous decode's state word is requested one boundary ahead (~45 symbols),
// so the comparison at the boundary does not expose a round trip.  Addresses are `uniform base + 32-bit lane offset` (a
// chunk's checkpoints stay far below 4 GiB: its stream capacity is capped in plan_chunks).
constexpr uint32_t kCpRow = 256, kCpRowBytes = kCpRow * 8;
__device__ __forceinline__ uint32_t cps_byte_off(uint32_t idx)
{
    return (idx / kCpRow) * (kMaxCp * kCpRowBytes) + (idx % kCpRow) * 8;
}

#ifndef MJX_GRID_WG_FAST
#define MJX_GRID_WG_FAST 0      // 1: the layout of rounds 1-3 (a picture's workgroups fast), for A/B
#endif
// The grid of the entropy kernels that walk (picture, workgroup of the picture): the PICTURE is the fast dimension.  Workgroups go
// to the eight XCDs round robin in launch order; with the picture's workgroups fast and a grid of 4 x pictures, the fourth
// workgroup of every picture -- the short one, or an empty one when another picture of the chunk needs four -- always met the
// same two XCDs, which then idled: the write pass of 2048 4K pictures took 9.3 instead of 7.2 ms at 640-byte subsequences
// (three full workgroups + an empty slot per picture), and which subsequence lengths were "good" was an artefact of that.
// Round 5: ... and within every whole run of eight pictures the order is rotated by a hash of the run's index.  A chunk whose
// pictures alternate with a period of 2, 4 or 8 -- the [Y, Cb, Cr, picture] images of three-scan files are the case that showed
// it -- put all the long pictures' second and third workgroups onto the same two XCDs (1024 three-scan 4K files, one stream:
// counting pass 3.9 ms, write pass 13.0; the same bits as luma + interleaved chroma, three images per file: 1.65 and 6.6).
// (picture_of_slot: for every kernel whose fast grid dimension is the picture)
#ifndef MJX_GRID_ROTATE
#define MJX_GRID_ROTATE 1
#endif
__device__ __forceinline__ uint32_t picture_of_slot(uint32_t x, uint32_t n)
{
    if (!MJX_GRID_ROTATE || x >= (n & ~7u)) return x;                       // (the last, partial run keeps its order)
    return (x & ~7u) | ((x + (((x >> 3) * 0x9E3779B1u) >> 29)) & 7u);
}
__device__ __forceinline__ uint32_t entropy_grid_image() { return MJX_GRID_WG_FAST ? blockIdx.y : picture_of_slot(blockIdx.x, gridDim.x); }
__device__ __forceinline__ uint32_t entropy_grid_wg() { return MJX_GRID_WG_FAST ? blockIdx.x : blockIdx.y; }
inline dim3 entropy_grid(uint32_t max_wg, uint32_t nimg) { return MJX_GRID_WG_FAST ? dim3(max_wg, nimg) : dim3(nimg, max_wg); }

struct GlobalCps {
    unsigned char *base;    // the chunk's checkpoint array (wave-uniform)
    uint32_t off;           // cps_byte_off(subsequence index in the chunk)
    uint32_t next;          // prefetched state word of the previous decode
    __device__ __forceinline__ uint2 &pair(uint32_t k) const { return *reinterpret_cast<uint2 *>(base + (off + k * kCpRowBytes)); }
    __device__ __forceinline__ uint32_t &word(uint32_t k, uint32_t j) const { return *reinterpret_cast<uint32_t *>(base + (off + k * kCpRowBytes + 4 * j)); }
    __device__ __forceinline__ void prime() { next = word(0, 0); }
    __device__ __forceinline__ uint32_t get(uint32_t k)
    {
        const uint32_t v = next;
        if (k + 1 < uint32_t(kMaxCp)) next = word(k + 1, 0);
        return v;
    }
    __device__ __forceinline__ uint32_t get_m(uint32_t k) const { return word(k, 1); }
    __device__ __forceinline__ uint32_t get_w(uint32_t k) const { return word(k, 0); }
    __device__ __forceinline__ CpPair get_pair(uint32_t k) const { const uint2 v = pair(k); return CpPair{v.x, v.y}; }
    __device__ __forceinline__ void set(uint32_t k, uint32_t v, uint32_t m) const { pair(k) = make_uint2(v, m); }
};

struct __attribute__((packed, aligned(4))) Entry4 { uint32_t a, b, c, d; };

// A lane's contiguous run of output dwords, staged in a ring of 2 * GROUP dwords in LDS and written to HBM as whole
// aligned groups of GROUP dwords (16-byte stores back to back).  The lane's region fills slowly (a 128-byte line of
// stream entries per ~45 symbols, ~20 us), longer than a dirty line survives in L2, so what reaches HBM is what each
// store instruction carried: dword stores cost a 32-byte sector each (and an instruction per symbol step), grouped
// stores write every sector once.  flush_groups() is called by the whole wave every kFlushEvery symbols
// (wave-uniform branch); the caller guarantees that at most GROUP dwords are pushed between two calls, so the
// ring never overflows.  A run that does not start on a group boundary first goes dword-wise up to the next one.
#ifndef MJX_FLUSH_EVERY
#define MJX_FLUSH_EVERY 8
#endif
constexpr uint32_t kFlushEvery = MJX_FLUSH_EVERY;
template <uint32_t GROUP, bool ALIGNED = false>
struct LaneRing {
    static constexpr uint32_t kRing = 2 * GROUP;
    uint32_t *ring;         // the lane's kRing dwords of LDS
    uint32_t *out;          // the image's region (aligned to GROUP dwords) -- or, ALIGNED with gstride != GROUP, the lane's column
    uint32_t off;           // next index
    uint32_t flushed;       // indices below this are in HBM
    uint32_t gstride = GROUP;   // ALIGNED: dwords from one group of the run to the next (GROUP: the run is contiguous; 4 * GROUP: a
                                // column of a quad-interleaved stream, see stream_phys)
    __device__ __forceinline__ void begin(uint32_t *lds, uint32_t *region, uint32_t first)
    {
        ring = lds;
        out = region;
        off = flushed = first;
    }
    __device__ __forceinline__ void push(uint32_t v)
    {
        ring[off & (kRing - 1)] = v;
        off++;
    }
    __device__ __forceinline__ void store_group()
    {
        if constexpr (GROUP >= 4) {
            static_assert(GROUP == 4 || GROUP == 8, "one or two 16-byte stores");
            const uint4 *src = reinterpret_cast<const uint4 *>(ring + (flushed & (kRing - 1)));
#ifdef MJX_EXP_STORE_SMALL              // (measurement builds only: the same store instructions, into 8 KB per image -- they stay in L2)
            uint4 *dst = reinterpret_cast<uint4 *>(out + (flushed & 0x7f8u));
#else
            uint4 *dst = reinterpret_cast<uint4 *>(ALIGNED ? out + size_t(flushed / GROUP) * gstride : out + flushed);
#endif
            const uint4 v0 = src[0], v1 = src[GROUP / 4 - 1];
#ifndef MJX_EXP_NOSTORE                 // (measurement builds only: what the write pass costs without its global stores)
            dst[0] = v0;
            if constexpr (GROUP == 8) dst[1] = v1;
#else
            asm volatile("" :: "v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w), "v"(dst));
#endif
        } else if constexpr (GROUP == 2) {
            *reinterpret_cast<uint2 *>(out + flushed) = *reinterpret_cast<const uint2 *>(ring + (flushed & (kRing - 1)));
        } else {
            out[flushed] = ring[flushed & (kRing - 1)];
        }
        flushed += GROUP;
    }
    __device__ __forceinline__ void flush_groups()
    {
        if constexpr (ALIGNED) {
            // the run starts on a group boundary and at most GROUP dwords arrive between two calls: one group at most
            if (__builtin_amdgcn_ballot_w64(flushed + GROUP <= off)) {
                if (flushed + GROUP <= off) store_group();
            }
        } else {
            while (__builtin_amdgcn_ballot_w64((flushed & (GROUP - 1)) != 0 && flushed < off)) {
                if ((flushed & (GROUP - 1)) != 0 && flushed < off) {
                    out[flushed] = ring[flushed & (kRing - 1)];
                    flushed++;
                }
            }
            while (__builtin_amdgcn_ballot_w64((flushed & (GROUP - 1)) == 0 && flushed + GROUP <= off)) {
                if ((flushed & (GROUP - 1)) == 0 && flushed + GROUP <= off) store_group();
            }
        }
    }
    __device__ __forceinline__ void flush_all()
    {
        flush_groups();
        for (uint32_t i = flushed; i < off; i++) out[i] = ring[i & (kRing - 1)];
        flushed = off;
    }
};

// The DC differences leave as 16-bit words (round 3; int32 before): what EXTEND returns is an i16 in the reference as well
// (huffman.rs:256-268).  A lane's run in the per-block array fills an eighth as fast as its run of stream entries -- a
// 16-byte group per ~20 symbols, a 128-byte line per 160 -- so practically every group reaches HBM as a partial write of its
// own: with no DC output at all the write pass takes 8.7 instead of 10.7 ms per 2048 pictures, with every second difference
// dropped 9.9.  Halfwords halve the groups for the same 32 bytes of ring per lane; the prediction kernels read them and write
// the int32 predictions stage B reads into a second array (same indexing), so a repair run finds the differences intact.
struct DcRing16 {
    static constexpr uint32_t kRing = 16, kGroup = 8;     // halfwords: 32 bytes of LDS per lane, 16-byte groups
    uint16_t *ring;
    int16_t *out;           // the image's region (aligned to kGroup halfwords)
    uint32_t off, flushed;  // block indices: next, first not yet in HBM
    __device__ __forceinline__ void begin(uint32_t *lds, int16_t *region, uint32_t first)
    {
        ring = reinterpret_cast<uint16_t *>(lds);
        out = region;
        off = flushed = first;
    }
    __device__ __forceinline__ void push(int v)
    {
        ring[off & (kRing - 1)] = uint16_t(v);
        off++;
    }
    __device__ __forceinline__ void flush_groups()
    {
        while (__builtin_amdgcn_ballot_w64((flushed & (kGroup - 1)) != 0 && flushed < off)) {      // up to the first group boundary
            if ((flushed & (kGroup - 1)) != 0 && flushed < off) {
                out[flushed] = int16_t(ring[flushed & (kRing - 1)]);
                flushed++;
            }
        }
        while (__builtin_amdgcn_ballot_w64((flushed & (kGroup - 1)) == 0 && flushed + kGroup <= off)) {
            if ((flushed & (kGroup - 1)) == 0 && flushed + kGroup <= off) {
                *reinterpret_cast<uint4 *>(out + flushed) = *reinterpret_cast<const uint4 *>(ring + (flushed & (kRing - 1)));
                flushed += kGroup;
            }
        }
    }
    __device__ __forceinline__ void flush_all()
    {
        flush_groups();
        for (uint32_t i = flushed; i < off; i++) out[i] = int16_t(ring[i & (kRing - 1)]);
        flushed = off;
    }
};

// Sink of the write pass: the compact coefficient stream (see coef_entry), DC differences (one per block), tile
// offsets.  A symbol adds at most one stream entry, a block takes at least two symbols: the rings below hold.
#ifndef MJX_AC_GROUP
#define MJX_AC_GROUP 8
#endif
#ifndef MJX_DC_GROUP
#define MJX_DC_GROUP 4
#endif
constexpr uint32_t kAcGroup = MJX_AC_GROUP;                                          // 32-byte sectors of entries
#ifndef MJX_RING_PAD
#define MJX_RING_PAD 0
#endif
constexpr uint32_t kAcRingStride = 2 * kAcGroup + MJX_RING_PAD;                       // dwords between two lanes' rings (a multiple of 4: 16-byte reads)
static_assert(kAcGroup >= kFlushEvery && DcRing16::kGroup - 1 + kFlushEvery <= DcRing16::kRing, "ring capacity between two flushes (a block takes two symbols at least; the DC ring is flushed every second time)");
__device__ __forceinline__ uint32_t stream_run(uint32_t m) { return (m + kAcGroup - 1) & ~(kAcGroup - 1); }
struct StreamSink {
    LaneRing<kAcGroup, true> ac_ring;   // index = entry index in the image's stream region; runs are whole groups
    DcRing16 dc_ring;               // index = block index in the image
    uint32_t *tile_eoff;    // the image's tile offsets (+ sentinel)
    uint32_t tile_virt;     // what a tile offset adds to the ring's index: 0 (linear stream: the index is the offset), or the
                            // virtual index of the lane's column (quad-interleaved: subsequence * column capacity)
    int *status;
    uint32_t blk_bits;      // the current block's index, placed as in coef_entry (bits above the field: don't care),
                            // + 63 << 16: minus the scaled r = position
    uint32_t next_tile_blk, tile_idx, tile_blocks, total_blocks, ntiles;
    const DevImage *seg_im; // a scan of a multi-scan picture that is read without the gather: the "tiles" are its segments (planar_cut)
    __device__ __forceinline__ void next_cut(uint32_t from_blk)          // the first cut at or after block from_blk
    {
        const DevImage &im = *seg_im;
        const PlanarCut c = planar_cut((from_blk + im.bpm - 1) / im.bpm, im.mcux, im.seg_S, im.seg_T, im.seg_mcux, im.seg_hs, im.seg_vs);
        tile_idx = c.slot;
        next_tile_blk = c.mcu * im.bpm;
    }
    static __device__ __forceinline__ uint32_t block_bits(uint32_t blk) { return ((blk & 0xffu) << 22) + (63u << kRShift); }
    __device__ __forceinline__ void dc(uint32_t b, int v)
    {
#if !defined(MJX_EXP_DCSTREAM) && !defined(MJX_EXP_NODC)
        dc_ring.push(v);                            // (b == dc_ring.off - 1: a lane's blocks are consecutive)
#endif
        if (b == next_tile_blk) {           // first block of a stage-B tile: remember where its entries start
            tile_eoff[tile_idx] = tile_virt + ac_ring.off;
            if (seg_im) {
                next_cut(b + 1);
            } else {
                tile_idx++;
                next_tile_blk += tile_blocks;
            }
        }
    }
    __device__ __forceinline__ void ac(uint32_t, uint32_t r_scaled, int v)
    {
        ac_ring.push((uint32_t(v) & 0xffffu) | (blk_bits - r_scaled));      // coef_entry: position = 63 - r
    }
    __device__ __forceinline__ void flush_groups()          // wave-uniform call
    {
        ac_ring.flush_groups();
        dc_ring.flush_groups();
    }
    __device__ __forceinline__ void flush_step(uint32_t it)  // wave-uniform call, every kFlushEvery symbols
    {
        ac_ring.flush_groups();
        // the DC ring every second time: a block takes two symbols at least, so at most kFlushEvery differences arrive in
        // 2 * kFlushEvery symbols on top of the < kGroup that wait for their group (9.83 -> 9.74 ms per 2048 pictures)
#if !defined(MJX_EXP_DCSTREAM) && !defined(MJX_EXP_NODC)
        if (it % (2 * kFlushEvery) == 0) dc_ring.flush_groups();
#endif
    }
    __device__ __forceinline__ void flush_entries() { ac_ring.flush_groups(); }                             // (the two halves of flush_step: diagnostic builds stamp between them)
    __device__ __forceinline__ void flush_dc(uint32_t it) { if (it % (2 * kFlushEvery) == 0) dc_ring.flush_groups(); }
    __device__ __forceinline__ void block_done(uint32_t next_blk)
    {
        blk_bits += 1u << 22;
        if (next_blk == total_blocks) tile_eoff[ntiles] = tile_virt + ac_ring.off;
    }
    __device__ __forceinline__ void flush()
    {
        ac_ring.flush_groups();             // (the caller padded the run to a whole group)
        dc_ring.flush_all();
    }
    __device__ __forceinline__ void bad_code(uint32_t, uint32_t) const { atomicOr(status, 1); }
    __device__ __forceinline__ void tick() const {}
};

// Decode tables + per-image constants into LDS (dynamic LDS: the tables, then HuffImage).  The dynamic LDS of the
// entropy kernels is declared with the alignment of a primary table (kLutAlign), which lut_slot relies on.
constexpr uint32_t kLutAlign = kLutPrimarySize * sizeof(LutEntry);
// PAIR: the image's second table set (AC tables with a pair part), for the passes that take two symbols per step.
template <bool PAIR = false>
__device__ __forceinline__ void stage_tables(const DevImage &im, const LutEntry *lut_pool, unsigned char *smem,
                                             const HuffImage *&himg, const LutEntry *&lut)
{
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t n = PAIR ? im.lut2_n : im.lut_n;
    LutEntry *l = reinterpret_cast<LutEntry *>(smem);
    HuffImage *h = reinterpret_cast<HuffImage *>(smem + size_t(n) * sizeof(LutEntry));
    for (uint32_t i = tid; i < sizeof(HuffImage) / 4; i += nthr)
        reinterpret_cast<uint32_t *>(h)[i] = reinterpret_cast<const uint32_t *>(&im.himg)[i];
    const uint4 *lsrc = reinterpret_cast<const uint4 *>(lut_pool + (PAIR ? im.lut2_off : im.lut_off));
    uint4 *ldst = reinterpret_cast<uint4 *>(l);
    for (uint32_t g = tid; g < n / 4; g += nthr) ldst[g] = lsrc[g];
    __syncthreads();
    // table offsets -> absolute LDS addresses (see lut_at); the tables come first in the workgroup's LDS, below 64 KiB
    if (tid < uint32_t(kMaxBlocksPerMcu))
        h->btab[tid].tabs = (PAIR ? h->tabs_pair[tid] : h->btab[tid].tabs) + uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)(l))) * 0x10001u;
    __syncthreads();
    himg = h;
    lut = l;
}

// Where subsequence s of an image lies.  Without restart intervals the scan is one segment cut every sub_bits bits.
// With them every interval is a segment of its own: segs[g] = (first subsequence, first bit) of
// segment g, subsequences do not straddle segments, and a segment's first subsequence starts in the known state
// (block 0 of an MCU, DC code next) -- it needs no synchronisation.
struct SubLoc {
    uint32_t start, end;        // bits: the subsequence owns the symbols that start in (start, end] (and at start, if first)
    uint32_t seg;               // segment index
    uint32_t seg_sub0;          // first subsequence of the segment
};
__device__ __forceinline__ SubLoc locate_sub(const DevImage &im, const HuffImage &h, const uint32_t *segs, uint32_t s)
{
    SubLoc l;
    if (im.nseg <= 1) {                                                    // wave-uniform
        l.seg = 0;
        l.seg_sub0 = 0;
        l.start = s * h.sub_bits;
        const uint32_t e = l.start + h.sub_bits;
        l.end = e < h.total_bits ? e : h.total_bits;
        return l;
    }
    const uint2 *sg = reinterpret_cast<const uint2 *>(segs) + im.seg_off;
    uint32_t lo = 0, hi = im.nseg;                                         // largest g with sg[g].x <= s
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sg[mid].x <= s) lo = mid; else hi = mid;
    }
    const uint2 a = sg[lo], nx = sg[lo + 1];
    l.seg = lo;
    l.seg_sub0 = a.x;
    l.start = a.y + (s - a.x) * h.sub_bits;
    const uint32_t e = l.start + h.sub_bits;
    l.end = e < nx.y ? e : nx.y;
    return l;
}

// Per-lane window of the bitstream in LDS.  A lane that read its bits straight from HBM would wait for an L2 round
// trip on almost every symbol: some lane of the wave always needs a refill, loads retire in order per wave, and the
// freshly requested dword cannot be moved up a register queue before it has landed.  (In the write pass such loads
// would also queue behind the scattered stores.)  Instead every lane owns kWinDwords big-endian dwords in LDS; the
// wave restages all its windows together (wave-uniform branch, 16-byte loads) whenever one lane is about to run
// out, about every 40 symbols, and the per-symbol refills only touch LDS.
#ifndef MJX_WIN_DWORDS
#define MJX_WIN_DWORDS 8
#endif
constexpr int kWinDwords = MJX_WIN_DWORDS;
#ifndef MJX_WIN_PAD
#define MJX_WIN_PAD 1
#endif
constexpr int kWinStride = kWinDwords + MJX_WIN_PAD;      // odd stride: lanes spread over all banks
#ifndef MJX_WIN_PREFETCH
#define MJX_WIN_PREFETCH 1
#endif
static_assert(MJX_WIN_DWORDS == 8 || !MJX_WIN_PREFETCH, "the prefetching restage moves windows of two 16-byte pieces");
struct LdsWindow {
    const unsigned char *lds;    // lane's window
    uint32_t wbase;              // stream byte offset of the window's first dword (as of the last fill the caller noted)
    uint32_t rp;                 // read pointer (LDS address): the dword after w1 (stream offset wn - 4)
    __device__ __forceinline__ uint32_t be32(uint32_t byte_off) const { return *reinterpret_cast<const uint32_t *>(lds + (byte_off - wbase)); }
    __device__ __forceinline__ uint32_t ahead(uint32_t) const { return *(const volatile __attribute__((address_space(3))) uint32_t *)(uintptr_t(rp)); }   // (volatile: one plain read per step, not folded into the restage branch)
    __device__ __forceinline__ void advance() { rp += 4; }
};
// wbase: byte offset inside the lane's subsequence, a multiple of 16
__device__ __forceinline__ void window_fill(uint32_t *lds, const LaneBits &g, uint32_t wbase)
{
#pragma unroll
    for (int q = 0; q < kWinDwords / 4; q++) {
        const uint4 v = g.piece((wbase >> 4) + q);
        lds[4 * q] = __builtin_bswap32(v.x);
        lds[4 * q + 1] = __builtin_bswap32(v.y);
        lds[4 * q + 2] = __builtin_bswap32(v.z);
        lds[4 * q + 3] = __builtin_bswap32(v.w);
    }
}

#if defined(MJX_STAMP) || defined(MJX_STAMP_B)
// Diagnostic build: shader-clock stamps between the parts of a wave step of the write pass (MJX_STAMP) or of a stage-B tile
// (MJX_STAMP_B), summed per part over the launch.
// (The stamp drains the LDS counter -- cdna guide s7 -- so the build's run time means nothing; its shares do.)
__device__ unsigned long long g_stamp_acc[8];
struct WaveStamp {
    uint32_t last, acc[8];
    __device__ __forceinline__ uint32_t now() { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return uint32_t(t); }
    __device__ __forceinline__ void begin() { for (int k = 0; k < 8; k++) acc[k] = 0; last = now(); }
    __device__ __forceinline__ void at(int k) { __builtin_amdgcn_sched_barrier(0); const uint32_t t = now(); acc[k] += t - last; last = t; __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void end()
    {
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_stamp_acc[k], (unsigned long long)acc[k]);
    }
};
extern "C" int mjx_debug_stamps(unsigned long long out[8], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_acc), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_acc), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#endif

// Decode loop over LDS windows: a plain per-lane loop.  A lane that is done (left its subsequence, merged with its
// previous decode, or ran past the last block in the write pass) drops out of the exec mask, so its state stays put in
// its registers; the window restage and the ring flush are uniform over the lanes still active.  (These kernels are
// bound by vector-instruction issue -- ~2 cycles per wave instruction on a SIMD-32 -- so the form of the loop
// matters: an `if (running)` body inside a wave-uniform loop made the compiler copy the lane state in and out of
// temporaries, 20 of 57 vector instructions per step.)
//   WRITE  emit coefficients through `sink`      CP  0: none, 1: record checkpoints, 2: record + merge (see mjx_huff.h)
//   PAIRSTEP  two symbols per step where the table's pair part allows (counting passes on the second table set)
template <bool WRITE, int CP, bool PAIRSTEP = !WRITE, class Sink, class CpStore>
__device__ __forceinline__ SubseqState wave_decode(bool live, SubseqState entry, uint32_t end_bit, uint32_t blk,
                                                   uint32_t blk_limit, const LaneBits &g, uint32_t *my_win, const LutEntry *lut,
                                                   const HuffImage &h, Sink &sink, CpStore &cps, uint32_t sub_start,
                                                   SubseqState old_exit)
{
    LaneState st;
    LaneEvents ev;
    const unsigned char *win_lds = reinterpret_cast<const unsigned char *>(my_win);
    const uint32_t win_addr = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)(my_win)));
    LdsWindow win;
    {
        const uint32_t first = 4u * ((entry.p + 31u) >> 5);                     // byte offset of the lane's w1
        win = LdsWindow{win_lds, (first ? first - 4u : 0u) & ~15u, 0u};          // (pieces are 16 bytes)
        window_fill(my_win, g, win.wbase);
        lane_begin(st, win, h, entry);
        win.rp = win_addr + (st.wn - 4u - win.wbase);
    }
    events_begin<CP>(ev, sub_start, end_bit, h.cp_bits);
    const uint32_t win_end = win_addr + 4u * kWinDwords;
    const uint32_t total_blocks = blk_limit;                               // (write pass: first block the lane must not write)
    bool running = live && entry.p <= end_bit && !(WRITE && blk >= total_blocks);
    const bool started = running;
    uint32_t it = 1;
#ifdef MJX_STAMP
    WaveStamp sp;
    if (WRITE) sp.begin();
#endif
#if MJX_WIN_PREFETCH
    // Round 6: the pieces a restage will write are requested right behind the restage before it, into registers -- a restage used to be
    // load, wait for L2 / HBM, write, with the whole wave parked (a fifth of an emitting wave's cycles in the stamps).  Between two
    // restages a lane moves on by less than a window, so what it will need is two of the three pieces behind its window's first one:
    // (wb + 1, wb + 2) or (wb + 2, wb + 3); a lane still inside its first piece keeps its window.
    uint32_t wb = win.wbase >> 4;                                              // first piece of the lane's window
    const uint32_t last_piece = (h.sub_bits >> 7) + kLookPieces - 1u;          // (the column's last row: scan_region_rows)
    uint4 pa = make_uint4(0, 0, 0, 0), pb = pa, pc = pa;
    if (running) {
        pa = g.piece(min(wb + 1u, last_piece));
        pb = g.piece(min(wb + 2u, last_piece));
        pc = g.piece(min(wb + 3u, last_piece));
    }
#endif
    while (running) {                                                          // per-lane loop: finished lanes are masked off
#if MJX_WIN_PREFETCH
        if (__builtin_amdgcn_ballot_w64(win.rp >= win_end)) {                  // uniform over the active lanes: restage
            const uint32_t q1 = (st.wn - 4u) >> 4;                             // (w0, w1 are in registers; wn - 4 is read next)
            if (q1 != wb) {
                const bool two = q1 != wb + 1u;
                const uint4 lo = two ? pb : pa, hi = two ? pc : pb;
                my_win[0] = __builtin_bswap32(lo.x); my_win[1] = __builtin_bswap32(lo.y); my_win[2] = __builtin_bswap32(lo.z); my_win[3] = __builtin_bswap32(lo.w);
                my_win[4] = __builtin_bswap32(hi.x); my_win[5] = __builtin_bswap32(hi.y); my_win[6] = __builtin_bswap32(hi.z); my_win[7] = __builtin_bswap32(hi.w);
                win.rp = win_addr + ((st.wn - 4u) & 15u);
                wb = q1;
            }
            pa = g.piece(min(wb + 1u, last_piece));                           // ... and the next restage's pieces
            pb = g.piece(min(wb + 2u, last_piece));
            pc = g.piece(min(wb + 3u, last_piece));
        }
#else
        if (__builtin_amdgcn_ballot_w64(win.rp >= win_end)) {                  // uniform over the active lanes: restage
            window_fill(my_win, g, (st.wn - 4u) & ~15u);                       // (w0, w1 are in registers; wn - 4 is read next)
            win.rp = win_addr + ((st.wn - 4u) & 15u);
        }
#endif
#ifdef MJX_STAMP
        if (WRITE) { sp.at(0); (void)symbol_step<WRITE, PAIRSTEP>(st, win, lut, h, blk, sink, sp); sp.at(5); }
        else
#endif
        (void)symbol_step<WRITE, PAIRSTEP>(st, win, lut, h, blk, sink);
        // Events (checkpoints, the end of the subsequence) are due when wn -- it only moves when the lane takes a new
        // dword -- has reached a boundary.  Both are tested on wn alone, as two flat conditions: these loops are bound
        // by scalar-instruction issue (one per SIMD turn), and exec-mask bookkeeping for nested "crossed -> event ->
        // done" conditions cost more than the vector work it guarded.
        static_assert(CP == 0 || CP == 1, "the kernels record checkpoints or not; merging re-decodes go slice-wise (merge_slice)");
        if (CP == 1 && st.wn >= ev.next_wn && st.wn < ev.end_wn) checkpoint_record(st, ev, cps);
        running = st.wn < ev.end_wn && !(WRITE && blk >= total_blocks);
#ifdef MJX_STAMP
        if constexpr (WRITE) {
            if (it % kFlushEvery == 0) { sink.flush_entries(); sp.at(6); sink.flush_dc(it); }
            it++;
            sp.at(7);
        }
#else
        if (WRITE) {
            if (it % kFlushEvery == 0) sink.flush_step(it);
            it++;
        }
#endif
    }
#ifdef MJX_STAMP
    if (WRITE) sp.end();
#endif
    if (!started) return make_state(entry.p, entry.z, entry.c);
    if (CP) checkpoint_fixup(cps, ev.k, st.n, lane_m(st));
    return lane_exit(st, ev, h, old_exit);
}

// k_huff_spec: every lane decodes its subsequence from the guess "a block starts exactly here", recording its exit
// state and a checkpoint every 256 bits.  Lanes do the same amount of work (+-3 %), so the plain per-lane loop
// (exec-masked by the compiler) is efficient.
extern "C" __global__ __launch_bounds__(kHuffWg) void k_huff_spec(const DevImage *images, const uint8_t *scan_pool,
                                                                   const LutEntry *lut_pool, SubseqState *g_entry,
                                                                   SubseqState *g_exit, uint32_t *g_cps, uint32_t win_off,
                                                                   const uint32_t *segs, uint8_t *g_gen)
{
    extern __shared__ __attribute__((aligned(kLutAlign))) unsigned char smem[];   // tables, HuffImage, windows
    uint32_t *s_win = reinterpret_cast<uint32_t *>(smem + win_off);
    const uint32_t img = entropy_grid_image(), wgi = entropy_grid_wg();
    const DevImage &im = images[img];
    // (blockDim.x lanes: 512, or -- round 5 -- 256 / 128 for chunks whose scans are all that short: the LDS of these kernels is sized
    // per lane, and a scan of a hundred subsequences in a workgroup of 512 holds four times the LDS it uses)
    if (!im.valid || im.emit || wgi * blockDim.x >= im.himg.nsub) return;      // (emit: the picture's first decode is k_huff_emit's)
    const HuffImage *h;
    const LutEntry *lut;
    stage_tables<true>(im, lut_pool, smem, h, lut);
    const uint32_t s = wgi * blockDim.x + threadIdx.x;
    const bool live = s < h->nsub;
    const LaneBits bits{scan_pool + im.scan_off, (live ? s : 0u) * 16u, im.scan_cols * 16u};
    const SubLoc loc = locate_sub(im, *h, segs, live ? s : 0u);
    const SubseqState e = make_state(0, 0, 0);                              // (positions relative to the subsequence)
    NullSink sink;
    GlobalCps cps{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(im.sub_off + (live ? s : 0u)), 0};
    SubseqState x = wave_decode<false, 1>(live, e, live ? loc.end - loc.start : 0u, 0, 0xffffffffu, bits, s_win + threadIdx.x * kWinStride, lut, *h, sink, cps, 0u, e);
    if (!live) return;
    x.p += loc.start;
    g_entry[im.sub_off + s] = make_state(loc.start, 0, 0);
    g_exit[im.sub_off + s] = x;
    if (g_gen) g_gen[im.sub_off + s] = 0;                                   // (Gen2: this decode lies in the first set, the second holds nothing)
}

// k_huff_merge: one synchronisation round.  Subsequence s must start where s-1 ended: if entry[s] differs from
// exit[s-1], s is re-decoded from the corrected entry until it meets the path recorded by the previous decode of s
// (checkpoint match: median ~100 of ~800 symbols) or reaches the end.  A round that re-decodes nothing proves the
// fixed point entry[s] == exit[s-1] for all s, which (entry[0] being the true start) is the true decode; `mismatches`
// counts the re-decoded items of this round.  Exits are read while other lanes may be rewriting them (8-byte aligned
// accesses): a stale read only defers the repair to the next round, and the zero-count round is race-free by definition.
//
// Re-decodes differ in length by an order of magnitude, so they advance in slices of one checkpoint interval (256
// bits, ~50 symbols, the same work for every lane) and between slices the workgroup packs its unfinished items into
// its lowest lanes through LDS; waves left without items skip the slice.  An item between slices is just
// (s, p, z, c, n, m, k): the lane that picks it up stages the 48 bytes of stream the slice can touch in its LDS
// window and rebuilds the decoder registers from p, z, c.
// The slowest few per cent of the items need all sixteen slices; left in place they would hold their workgroup's
// LDS (and with it the CU's occupancy) for a single wave's worth of work.  So k_huff_merge runs kHeadSlices slices and
// appends what is still unfinished to a per-image list in HBM; k_huff_merge_tail picks the lists up with small
// workgroups that hold nothing but the tables and a window per lane (five of 256 lanes per CU) and runs every item to its end.
constexpr int kMergeWin = 16, kMergeStride = kMergeWin + 1, kItemDwords = 6;    // four pieces: a slice touches < 56 bytes from a 16-byte boundary
#ifndef MJX_HEAD_SLICES
#define MJX_HEAD_SLICES 6
#endif
constexpr int kHeadSlices = MJX_HEAD_SLICES;
#ifndef MJX_LATER_HEAD_SLICES
#define MJX_LATER_HEAD_SLICES 0
#endif
struct MergeItem { uint32_t s, p, zc, n, m, k; };
static_assert(kCpBits == 256, "a merge slice is one interval of kCpBits = 256 bits (window of four pieces); a picture's checkpoints lie every HuffImage::cp_bits, a multiple of it");

// One slice of one item: decode from (p, z, c) to the next checkpoint boundary (or the end of the subsequence).
// Returns true when the item is finished (its exit and checkpoints are final), false when `it` holds the progress.
// `depth` (when the item finishes): kEmitAll if the re-decode left the subsequence without meeting the previous decode's path,
// else 1 + the index of the checkpoint where it met it.
// Two generations (round 5, Gen2): every subsequence keeps the decode before the last as well -- entry, exit, checkpoints, in a second
// set of the chunk's arrays `stride` subsequences behind the first; gen[idx] bit 0 = the set that is current, bit 1 = the other set
// holds a decode.  A re-decode records its checkpoints in a scratch set (the third, 2 x stride behind the first) and compares with
// BOTH recorded paths.  When it is done (merge_finish): it met the current path -- the usual case -- and its few checkpoints in
// front of the meeting point replace that path's, in place; or it met the older path: the same there, and the sets change roles;
// or it met neither and becomes the current decode in the older one's place.  Why: where content does not synchronise (runs of identical flat MCUs) a wrong state is handed
// from lane to lane, each lane re-decodes from it without ever meeting its own -- true -- first decode and used to overwrite that
// decode's checkpoints; the truth, one round behind, then had to decode every subsequence in full again.  With the older decode
// kept it meets it at the first checkpoint, and a lane whose entry merely RETURNS to the one its older decode started from takes
// that decode back without decoding (round_begin).  An item carries the set it writes and whether the other one is valid in zc
// (bits 16, 17).  stride == 0: one generation, as before (pictures whose first decode emitted: k_huff_prefix needs that decode's
// own checkpoints).
struct Gen2 { uint8_t *gen; uint32_t stride; };
__device__ __forceinline__ bool merge_slice(MergeItem &it, const DevImage &im, const HuffImage &h, const LutEntry *lut,
                                            const unsigned char *region, uint32_t *my_win, const SubseqState *g_exit,
                                            uint32_t *g_cps, SubseqState &x, const uint32_t *segs, uint32_t &depth, Gen2 g2, uint32_t &met)
{
    const SubLoc loc = locate_sub(im, h, segs, it.s);
    const uint32_t sub_start = loc.start, end_bit = loc.end - loc.start;       // (the lane works relative to its subsequence)
    const uint32_t gs = im.emit ? 0u : g2.stride, oset = gs ? (it.zc >> 16) & 1u : 0u, cset = gs ? oset ^ 1u : 0u;
    const bool other_valid = gs && ((it.zc >> 17) & 1u);
    const uint32_t idx_c = im.sub_off + it.s + cset * gs, idx_o = im.sub_off + it.s + oset * gs;
    const GlobalCps cps{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(idx_c), 0};          // the current decode's checkpoints
    const GlobalCps cpo{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(idx_o), 0};          // the older decode's
    const GlobalCps cpw{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(im.sub_off + it.s + 2u * gs), 0};   // where this decode records (gs == 0: in place)
    bool fin = false;
    depth = kEmitAll;
    met = 0;
    x = make_state(it.p, it.zc & 0xffu, (it.zc >> 8) & 0xffu, it.n, it.m);
    const uint32_t pl = it.p - sub_start;
    if (it.p >= sub_start && pl <= end_bit) {                      // (else nothing starts inside s)
        // it.k counts slices (intervals of kCpBits); a checkpoint lies at the end of slice it.k when that boundary is a multiple of the
        // picture's cp_bits (every slice for the pictures of the two-pass path, every fourth or so for those whose first decode emits)
        const uint32_t per = h.cp_bits / uint32_t(kCpBits);
        const bool at_cp = (it.k + 1u) % per == 0u;
        const uint32_t ck = (it.k + 1u) / per - 1u;                // index of that checkpoint
        const uint32_t old_word = (at_cp && ck < uint32_t(kMaxCp)) ? cps.get_w(ck) : 0u;   // requested early
        const uint32_t old_word2 = (at_cp && ck < uint32_t(kMaxCp) && other_valid) ? cpo.get_w(ck) : 0u;
        const uint32_t wi1 = (pl + 31u) >> 5, wbase = (wi1 ? 4u * wi1 - 4u : 0u) & ~15u;
        const LaneBits bits{region, it.s * 16u, im.scan_cols * 16u};
#pragma unroll
        for (int q = 0; q < kMergeWin / 4; q++) {
            const uint4 v = bits.piece((wbase >> 4) + q);
            my_win[4 * q] = __builtin_bswap32(v.x);
            my_win[4 * q + 1] = __builtin_bswap32(v.y);
            my_win[4 * q + 2] = __builtin_bswap32(v.z);
            my_win[4 * q + 3] = __builtin_bswap32(v.w);
        }
        LdsWindow win{reinterpret_cast<const unsigned char *>(my_win), wbase, 0u};
        LaneState st;
        lane_begin(st, win, h, make_state(pl, it.zc & 0xffu, (it.zc >> 8) & 0xffu));
        win.rp = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)(my_win))) + (st.wn - 4u - wbase);
        st.n = it.n;
        lane_add_m(st, it.m);
        const uint32_t end_wn = wn_after(end_bit);
        uint32_t blk = 0;
        NullSink sink;
        uint32_t stop_wn = wn_after((it.k + 1) * kCpBits);
        stop_wn = stop_wn < end_wn ? stop_wn : end_wn;
        while (st.wn < stop_wn) (void)symbol_step<false, true>(st, win, lut, h, blk, sink);
        if (st.wn >= end_wn) {                                     // left the subsequence without merging
            fin = true;
            x = make_state(lane_pos(st) + sub_start, lane_z(st), lane_c(st, h), st.n, lane_m(st));
        } else {
            const uint32_t state = cp_state_word(st);
            if (at_cp && (old_word & kCpStateMask) == state) {     // met the previous decode's path
                const SubseqState old_exit = g_exit[idx_c];
                fin = true;
                depth = ck + 1u;
                met = 1;
                x = make_state(old_exit.p, old_exit.z, old_exit.c, st.n + ((old_word >> 16) & 0x7fffu), lane_m(st) + cps.get_m(ck));
            } else if (at_cp && other_valid && (old_word2 & kCpStateMask) == state) {     // ... the path of the decode before it
                const SubseqState old_exit = g_exit[idx_o];
                fin = true;
                depth = ck + 1u;
                met = 2;
                x = make_state(old_exit.p, old_exit.z, old_exit.c, st.n + ((old_word2 >> 16) & 0x7fffu), lane_m(st) + cpo.get_m(ck));
            } else {
                if (at_cp) cpw.set(ck, state | (st.n << 16), lane_m(st));
                it.k++;
            }
        }
        if (!fin) {
            it.p = lane_pos(st) + sub_start;
            it.zc = lane_z(st) | (lane_c(st, h) << 8) | (it.zc & 0x30000u);
            it.n = st.n;
            it.m = lane_m(st);
        }
    } else {
        fin = true;
    }
    return fin;
}
// A finished item: its exit, and the counts of the checkpoints it recorded turned from "so far" into "to the end".  Pictures whose
// first decode emitted keep, per subsequence, the deepest point at which any of its re-decodes met the recorded path: from there on
// the first decode's entries are the true ones (k_huff_prefix writes what lies in front of it).
__device__ __forceinline__ void merge_finish(const MergeItem &it, const DevImage &im, const SubseqState &x,
                                             SubseqState *g_entry, SubseqState *g_exit, uint32_t *g_cps, EmitSub *g_esub, uint32_t depth,
                                             Gen2 g2, uint32_t met)
{
    const uint32_t gs = im.emit ? 0u : g2.stride, idx = im.sub_off + it.s;
    const uint32_t per = im.himg.cp_bits / uint32_t(kCpBits), nrec = it.k / per;      // checkpoints this decode recorded
    if (!gs) {
        const GlobalCps cps{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(idx), 0};
        checkpoint_fixup(cps, nrec, x.n, x.m);
        g_exit[idx] = x;
    } else {
        // where the decode's path now lives: with the path it met (in front of the meeting point its own checkpoints replace that
        // path's), or -- it met neither -- in the older decode's place
        const uint32_t oset = (it.zc >> 16) & 1u, cset = oset ^ 1u, dset = met == 1 ? cset : oset;
        const GlobalCps src{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(idx + 2u * gs), 0};
        const GlobalCps dst{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(idx + dset * gs), 0};
        for (uint32_t j = 0; j < nrec; j++) {                     // (counts so far -> counts to the exit, as checkpoint_fixup)
            const CpPair v = src.get_pair(j);
            dst.set(j, (v.w & kCpStateMask) | ((x.n - ((v.w >> 16) & 0x7fffu)) << 16), x.m - v.m);
        }
        g_entry[idx + dset * gs] = g_entry[idx + 2u * gs];         // (the state this decode started from: round_begin left it there)
        g_exit[idx + dset * gs] = x;
        if (met != 1) g2.gen[idx] = uint8_t(oset | 2u);            // the set that held the older decode is the current one now
    }
    if (im.emit) {
        uint32_t &kf = g_esub[im.sub_off + it.s].kfix;             // (one lane per subsequence and round: no race)
        kf = kf > depth ? kf : depth;
    }
}

// What a lane does at the start of a round: subsequence it.s must start where it.s - 1 ended.  Returns true when it has to be decoded
// again (the item is set up: its new entry is written, into the set this decode will write); `changed` also when the lane took its
// older decode back instead (its entry returned to the one that decode started from: no decoding, but its exit has changed).
// (cooperative: every thread of the merge workgroup calls it, `live` = the thread has a subsequence.  Remembered decodes are taken
// back in a SWEEP: when a lane takes its older decode back its exit changes, and the lane behind it -- in the same workgroup --
// may find that very exit to be what ITS older decode started from, and so on; the lanes' current exits stand in LDS (`s_x`, in the
// window area, unused at this point) and the workgroup iterates until no lane changes hands any more -- lane k is settled after k
// iterations at the latest, a picture of photographs after one.  The predecessor of the workgroup's first lane is read once.)
__device__ __forceinline__ bool round_begin(MergeItem &it, bool live, const DevImage &im, SubseqState *g_entry, const SubseqState *g_exit,
                                            const uint32_t *segs, Gen2 g2, bool &changed, unsigned long long *s_x, uint32_t *s_any)
{
    const uint32_t idx = im.sub_off + it.s, gs = im.emit ? 0u : g2.stride, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    changed = false;
    if (!gs) {                                                      // one generation: every lane for itself (uniform: im.emit is the picture's)
        if (!live) return false;
        const SubseqState prev = g_exit[idx - 1];
        bool active = !same_entry(prev, g_entry[idx]);
        if (im.nseg > 1 && locate_sub(im, im.himg, segs, it.s).seg_sub0 == it.s) active = false;   // a segment's first: entry known
        it.p = prev.p;
        it.zc = prev.z | (uint32_t(prev.c) << 8);
        if (active) g_entry[idx] = make_state(prev.p, prev.z, prev.c);
        changed = active;
        return active;
    }
    auto head = [](const SubseqState &st) { return (unsigned long long)st.p | ((unsigned long long)st.z << 32) | ((unsigned long long)st.c << 40); };
    const uint32_t g = live ? g2.gen[idx] : 0u;
    uint32_t cur = g & 1u;
    const bool valid2 = (g & 2u) != 0;
    unsigned long long e[2] = {0, 0}, x[2] = {0, 0}, prev_glob = 0;
    bool seg_first = false;
    if (live) {
        e[cur] = head(g_entry[idx + cur * gs]);
        x[cur] = head(g_exit[idx + cur * gs]);
        if (valid2) { e[cur ^ 1u] = head(g_entry[idx + (cur ^ 1u) * gs]); x[cur ^ 1u] = head(g_exit[idx + (cur ^ 1u) * gs]); }
        seg_first = im.nseg > 1 && locate_sub(im, im.himg, segs, it.s).seg_sub0 == it.s;
        if (tid == 0) prev_glob = head(g_exit[idx - 1 + (g2.gen[idx - 1] & 1u) * gs]);
    }
    s_x[tid] = x[cur];
    __syncthreads();
    bool flipped = false;
    for (uint32_t iter = 0; iter < uint32_t(kMergeWg); iter++) {
        const unsigned long long prev = tid ? s_x[tid - 1] : prev_glob;
        const bool flip = live && !seg_first && valid2 && prev != e[cur] && prev == e[cur ^ 1u];
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(flip);
        if (lane == 0) s_any[wave] = mm ? 1u : 0u;
        __syncthreads();                                            // (also: every lane has read its neighbour's exit of this iteration)
        uint32_t any = 0;
        for (uint32_t w = 0; w < uint32_t(kMergeWg) / 64u; w++) any |= s_any[w];
        if (!any) break;
        if (flip) { cur ^= 1u; s_x[tid] = x[cur]; flipped = true; }
        __syncthreads();
    }
    const unsigned long long prev = tid ? s_x[tid - 1] : prev_glob;
    const bool active = live && !seg_first && prev != e[cur];
    __syncthreads();                                                // (s_x lies in the window area: the slices may write there from here on)
    if (flipped) g2.gen[idx] = uint8_t(cur | 2u);                  // the older decode started from exactly this state: it is the current one again
    it.p = uint32_t(prev);
    it.zc = uint32_t(prev >> 32) & 0xffffu;                         // z | c << 8
    if (active) {
        it.zc |= ((cur ^ 1u) << 16) | (valid2 ? 1u << 17 : 0u);     // (bit 16: the set that is not current; bit 17: it holds a decode)
        g_entry[idx + 2u * gs] = make_state(uint32_t(prev), uint32_t(prev >> 32) & 0xffu, uint32_t(prev >> 40) & 0xffu);   // (for merge_finish)
    }
    changed = active || flipped;
    return active;
}

extern "C" __global__ __launch_bounds__(kMergeWg) void k_huff_merge(const DevImage *images, const uint8_t *scan_pool,
                                                                const LutEntry *lut_pool, SubseqState *g_entry,
                                                                SubseqState *g_exit, uint32_t *g_cps, uint32_t *mismatches,
                                                                uint32_t win_off, uint32_t *g_items, uint32_t *g_item_count,
                                                                const uint32_t *segs, const uint32_t *prev_mismatches,
                                                                uint32_t head_slices, EmitSub *g_esub, uint8_t *g_gen, uint32_t gen_stride)
{
    const Gen2 g2{g_gen, gen_stride};
    // A round behind one that re-decoded nothing has nothing to do either (the fixed point is reached): it leaves at
    // once, its own count stays zero, and so does every later round's.  That makes spare rounds nearly free (a launch),
    // so enough of them are enqueued for streams that synchronise slowly (noisy pictures at quality >= 95 need 5..15).
    if (prev_mismatches && *prev_mismatches == 0) return;
    extern __shared__ __attribute__((aligned(kLutAlign))) unsigned char smem[];   // tables, HuffImage, windows (= item exchange), wave counts
    uint32_t *s_win = reinterpret_cast<uint32_t *>(smem + win_off);
    uint32_t *s_cnt = s_win + kMergeWg * kMergeStride;                     // (no static LDS: it would be padded to the dynamic part's alignment)
    const uint32_t img = entropy_grid_image(), wgi = entropy_grid_wg();
    const DevImage &im = images[img];
    if (!im.valid || wgi * kMergeWg + 1 >= im.himg.nsub) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    MergeItem it{wgi * kMergeWg + tid + 1, 0, 0, 0, 0, 0};
    bool active = false, changed = false;
    active = round_begin(it, it.s < im.himg.nsub, im, g_entry, g_exit, segs, g2, changed, reinterpret_cast<unsigned long long *>(s_win), s_cnt);
    {   // nothing to repair in this workgroup?  (not __syncthreads_or: its static LDS word would be padded to kLutAlign)
        // (what is counted is what CHANGED -- lanes that decode again and lanes that took their older decode back: the fixed point
        // is a round in which nothing changes)
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(changed);   // (all lanes vote: outside the branch)
        const unsigned long long ma = __builtin_amdgcn_ballot_w64(active);
        if (lane == 0) s_cnt[wave] = uint32_t(__popcll(mm)) | (uint32_t(__popcll(ma)) << 16);
        __syncthreads();
        uint32_t any = 0;
        for (uint32_t w = 0; w < kMergeWg / 64; w++) any += s_cnt[w];
        if (!any) return;
        if (tid == 0) atomicAdd(mismatches, any & 0xffffu);                // one atomic per workgroup, not per wave: all of them hit one word
        __syncthreads();                                                   // (s_cnt is written again in the slice loop)
        if (!(any >> 16)) return;                                          // nothing to decode here
    }
    auto hand_over = [&](unsigned long long mask, uint32_t rank) {      // the wave's unfinished items -> k_huff_merge_tail
        uint32_t base = 0;
        if (lane == 0 && mask) base = atomicAdd(g_item_count + img, uint32_t(__popcll(mask)));
        base = __shfl(base, 0);
        if (active) {
            uint32_t *slot = g_items + (size_t(im.sub_off) + base + rank) * kItemDwords;
            slot[0] = it.s; slot[1] = it.p; slot[2] = it.zc; slot[3] = it.n; slot[4] = it.m; slot[5] = it.k;
        }
    };
    // Rounds behind the first re-decode a few per cent of the subsequences: a workgroup would stage its tables and sit through
    // its slices for a handful of lanes (the second round took 0.27 ms per 1024 pictures, over half of the first one's, for a
    // twentieth of its items).  With head_slices == 0 the round only finds its items; k_huff_merge_tail, whose
    // workgroups take them packed, does all the decoding.
    if (head_slices == 0) {
        const unsigned long long mask = __ballot(active);
        hand_over(mask, __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)));
        return;
    }
    const HuffImage *h;
    const LutEntry *lut;
    stage_tables<true>(im, lut_pool, smem, h, lut);
    const unsigned char *bytes = scan_pool + im.scan_off;          // the image's (lane-interleaved) region
    uint32_t *my_win = s_win + tid * kMergeStride;
    for (uint32_t slice = 0;; slice++) {
        if (active) {
            SubseqState x;
            uint32_t depth, met;
            if (merge_slice(it, im, *h, lut, bytes, my_win, g_exit, g_cps, x, segs, depth, g2, met)) {
                merge_finish(it, im, x, g_entry, g_exit, g_cps, g_esub, depth, g2, met);
                active = false;
            }
        }
        const unsigned long long mask = __ballot(active);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
        if (slice + 1 == head_slices) {                                    // hand the stragglers to k_huff_merge_tail
            hand_over(mask, rank);
            break;
        }
        // pack the unfinished items into the lowest lanes
        if (lane == 0) s_cnt[wave] = uint32_t(__popcll(mask));
        __syncthreads();                                                   // (also: every lane is done with its window)
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kMergeWg / 64; w++) {
            const uint32_t c = s_cnt[w];
            before += w < wave ? c : 0u;
            total += c;
        }
        if (total == 0) break;
        if (active) {
            uint32_t *slot = s_win + (before + rank) * kItemDwords;
            slot[0] = it.s; slot[1] = it.p; slot[2] = it.zc; slot[3] = it.n; slot[4] = it.m; slot[5] = it.k;
        }
        __syncthreads();
        active = tid < total;
        if (active) {
            const uint32_t *slot = s_win + tid * kItemDwords;
            it.s = slot[0]; it.p = slot[1]; it.zc = slot[2]; it.n = slot[3]; it.m = slot[4]; it.k = slot[5];
        }
        __syncthreads();                                                   // the exchange area becomes windows again
    }
}

#ifndef MJX_TAIL_WG
#define MJX_TAIL_WG 256
#endif
static_assert(MJX_TAIL_WG <= MJX_MERGE_WG && MJX_MERGE_WG % MJX_TAIL_WG == 0, "the straggler kernel's grid is the merge grid times kMergeWg / kTailWg");
constexpr uint32_t kTailWg = MJX_TAIL_WG;      // lanes per workgroup (measured 64 / 128 / 256: 2.50 / 2.30 / 2.20 ms of merge rounds per 2048 pictures)
extern "C" __global__ __launch_bounds__(kTailWg) void k_huff_merge_tail(const DevImage *images, const uint8_t *scan_pool,
                                                                    const LutEntry *lut_pool, SubseqState *g_entry, SubseqState *g_exit,
                                                                    uint32_t *g_cps, uint32_t win_off,
                                                                    const uint32_t *g_items, const uint32_t *g_item_count,
                                                                    const uint32_t *segs, EmitSub *g_esub, uint8_t *g_gen, uint32_t gen_stride)
{
    const Gen2 g2{g_gen, gen_stride};
    extern __shared__ __attribute__((aligned(kLutAlign))) unsigned char smem[];   // tables, HuffImage, a window per lane
    uint32_t *s_win = reinterpret_cast<uint32_t *>(smem + win_off);
    // Workgroup (image, group): the image is the fast grid dimension, so that the groups that have items -- the first
    // few of every image -- are consecutive workgroup ids and spread over all XCDs and CUs; with the group as the fast
    // dimension they recur with the period of the grid and land on a fraction of the CUs.
    const uint32_t img = picture_of_slot(blockIdx.x, gridDim.x), group = blockIdx.y;
    const DevImage &im = images[img];
    if (!im.valid) return;
    const uint32_t count = g_item_count[img];
    if (group * kTailWg >= count) return;
    const HuffImage *h;
    const LutEntry *lut;
    stage_tables<true>(im, lut_pool, smem, h, lut);
    const uint32_t j = group * kTailWg + threadIdx.x;
    if (j >= count) return;
    const uint32_t *slot = g_items + (size_t(im.sub_off) + j) * kItemDwords;
    MergeItem it{slot[0], slot[1], slot[2], slot[3], slot[4], slot[5]};
    const unsigned char *bytes = scan_pool + im.scan_off;
    uint32_t *my_win = s_win + threadIdx.x * kMergeStride;
    // every lane runs its item to the end first; the read-modify-write of the recorded checkpoints then happens once
    // for the whole wave instead of after every slice for the lanes that happen to finish there
    SubseqState x;
    uint32_t depth, met;
    while (!merge_slice(it, im, *h, lut, bytes, my_win, g_exit, g_cps, x, segs, depth, g2, met)) {}
    merge_finish(it, im, x, g_entry, g_exit, g_cps, g_esub, depth, g2, met);
}

// ---- the merge rounds of a small batch in one launch ------------------------------------------------------------------
// A batch that leaves the device mostly empty (one picture, a handful) pays for its rounds in launches: six rounds enqueued =
// twelve kernels and six memsets in a row, most of them for nothing once the states have settled, and a picture that needs a
// seventh round pays a trip to the host.  When all merge workgroups of a chunk fit on the device at once (launch_huff_merge_loop
// checks), one launch runs the rounds in a loop with a device-wide barrier between them, until a round re-decodes nothing
// (the same proof of the fixed point as above) or `max_rounds` are spent.  Stragglers are not handed to a second kernel:
// occupancy is no concern here, a workgroup runs all its items to the end.
//   ctl[0] barrier arrivals (monotonic), ctl[1] workgroups that have left, ctl[2..4] re-decodes of round r in ctl[2 + r % 3],
//   ctl[5] set when a workgroup has given up waiting; the last workgroup to leave zeroes ctl for the next launch.
//   `verdict` receives the count of the last round run, or kLoopGaveUp.
// The workgroups wait for one another, so all of them must be resident at once.  The host only uses the kernel for chunks far
// below the device's capacity (launch_huff_merge_loop), but it cannot see what else the device is running (other processes,
// other contexts' loops): a workgroup that has waited a few seconds at a barrier gives up, raises ctl[5], and everybody leaves --
// the states are a valid intermediate result at any time, and mjx_batch_wait continues with the launch-per-round kernels.
constexpr uint32_t kLoopGaveUp = 0xffffffffu;
__device__ __forceinline__ bool grid_barrier(uint32_t *ctl, uint32_t target, uint32_t *s_flag, uint32_t spin_limit)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                                   // release what the workgroup has written
        atomicAdd(ctl, 1u);
        uint32_t spins = 0, gave_up = 0;
        while (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if ((++spins & 1023u) == 0 && (spins >= spin_limit || __hip_atomic_load(ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(ctl + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = 1;
                break;
            }
        }
        *s_flag = gave_up;
        __threadfence();                                                   // acquire what the others have written (one fence per
    }                                                                      // workgroup: its waves share the CU's L1 and the XCD's L2)
    __syncthreads();
    return *s_flag == 0;
}

extern "C" __global__ __launch_bounds__(kMergeWg) void k_huff_merge_loop(const DevImage *images, const uint8_t *scan_pool,
                                                                     const LutEntry *lut_pool, SubseqState *g_entry,
                                                                     SubseqState *g_exit, uint32_t *g_cps, uint32_t *verdict,
                                                                     uint32_t win_off, const uint32_t *segs, uint32_t *ctl,
                                                                     uint32_t participants, uint32_t max_rounds, uint32_t spin_limit,
                                                                     EmitSub *g_esub, uint8_t *g_gen, uint32_t gen_stride)
{
    const Gen2 g2{g_gen, gen_stride};
    extern __shared__ __attribute__((aligned(kLutAlign))) unsigned char smem[];   // tables, HuffImage, windows (= item exchange), wave counts
    uint32_t *s_win = reinterpret_cast<uint32_t *>(smem + win_off);
    uint32_t *s_cnt = s_win + kMergeWg * kMergeStride;
    const DevImage &im = images[blockIdx.y];
    if (!im.valid || blockIdx.x * kMergeWg + 1 >= im.himg.nsub) return;    // (not among the participants: the host counts the same way)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const HuffImage *h;
    const LutEntry *lut;
    stage_tables<true>(im, lut_pool, smem, h, lut);
    const unsigned char *bytes = scan_pool + im.scan_off;
    uint32_t *my_win = s_win + tid * kMergeStride;
    uint32_t last = 0;
    bool gave_up = __hip_atomic_load(ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;       // (dispatched after the others had given up)
    for (uint32_t round = 0; round < max_rounds && !gave_up; round++) {
        uint32_t *count = ctl + 2 + round % 3;
        if (tid == 0) ctl[2 + (round + 1) % 3] = 0;                        // the next round's count (nobody reads or adds to it in this round)
        MergeItem it{blockIdx.x * kMergeWg + tid + 1, 0, 0, 0, 0, 0};
        bool active = false, changed = false;
        active = round_begin(it, it.s < im.himg.nsub, im, g_entry, g_exit, segs, g2, changed, reinterpret_cast<unsigned long long *>(s_win), s_cnt);
        {
            const unsigned long long mm = __ballot(changed);
            if (lane == 0 && mm) atomicAdd(count, uint32_t(__popcll(mm)));
        }
        for (;;) {
            if (active) {
                SubseqState x;
                uint32_t depth, met;
                if (merge_slice(it, im, *h, lut, bytes, my_win, g_exit, g_cps, x, segs, depth, g2, met)) {
                    merge_finish(it, im, x, g_entry, g_exit, g_cps, g_esub, depth, g2, met);
                    active = false;
                }
            }
            const unsigned long long mask = __ballot(active);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
            if (lane == 0) s_cnt[wave] = uint32_t(__popcll(mask));
            __syncthreads();                                               // (also: every lane is done with its window)
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < kMergeWg / 64; w++) {
                const uint32_t c = s_cnt[w];
                before += w < wave ? c : 0u;
                total += c;
            }
            if (total == 0) break;
            if (active) {
                uint32_t *slot = s_win + (before + rank) * kItemDwords;
                slot[0] = it.s; slot[1] = it.p; slot[2] = it.zc; slot[3] = it.n; slot[4] = it.m; slot[5] = it.k;
            }
            __syncthreads();
            active = tid < total;
            if (active) {
                const uint32_t *slot = s_win + tid * kItemDwords;
                it.s = slot[0]; it.p = slot[1]; it.zc = slot[2]; it.n = slot[3]; it.m = slot[4]; it.k = slot[5];
            }
            __syncthreads();                                               // the exchange area becomes windows again
        }
        if (!grid_barrier(ctl, (round + 1) * participants, s_cnt + kMergeWg / 64, spin_limit)) { gave_up = true; break; }
        last = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last == 0) break;
    }
    // leave: the last workgroup out publishes the verdict and clears the control words for the next launch
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        if (gave_up) *verdict = kLoopGaveUp;
        if (atomicAdd(ctl + 1, 1u) + 1 == participants) {
            if (!gave_up && !__hip_atomic_load(ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) *verdict = last;
            ctl[0] = 0; ctl[2] = 0; ctl[3] = 0; ctl[4] = 0; ctl[5] = 0;
            __threadfence();
            ctl[1] = 0;
        }
    }
}

// Workgroup-wide exclusive scan helper (256 lanes): returns the exclusive prefix of v, total in *total.
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t *s_tmp, uint32_t *total)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= uint32_t(d)) incl += o;
    }
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWgLanes / 64; w++) {
        const uint32_t c = s_tmp[w];
        base += (w < wave) ? c : 0;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// ---- device-side de-stuffing and marker scan ------------------------------------------------------------------
// jpeg/mod.rs:371-385 copies the bytes after the SOS header and drops the 00 of every FF 00 pair.  Whether byte i is
// dropped depends only on bytes i-1 and i (a dropped byte is 00, so it never starts a pair itself): keep(i) =
// !(b[i] == 00 && b[i-1] == FF).  With restart intervals (beyond the reference) the RSTn markers leave
// the stream as well, and that is local too: an FF is never the second byte of a pair the host parser consumes (those
// are 00 or Dn), so every FF is looked at as a first byte -- FF followed by D0..D7 is a marker, both bytes go, and the
// number of bytes kept in front of it is where the next interval begins.  So the copy is a stream compaction and the
// marker list a second one:
//   k_destuff_count     bytes kept and markers found per 16 KiB segment of every scan
//   k_destuff_prefix    one workgroup per scan: exclusive sums of both over its segments; the scan's length is known now,
//                       and with it the picture's geometry -- total_bits, the number of subsequences -- which is written into
//                       the DevImage the decode kernels read (the host planned with the stuffed length as an upper bound and
//                       never learns the exact one: nothing waits for the device)
//   k_destuff_scatter   every lane writes the kept bytes of its 64-byte piece at segment base + workgroup-scan offset,
//                       and the offsets of its markers into the scan's marker list
//   k_restart_geometry  scans with restart intervals: the list becomes the picture's segment table (see locate_sub)
// All of it runs once per upload, before any decode, on the upload stream.
// keep flags of the 64 bytes [i0, i0+64) of `raw` as a bit mask (+ the bytes themselves in q[0..3]); returns the
// number of kept bytes.  *rst_out: bit j set = a marker FF Dn begins at byte j (restarts only).  The raw staging buffer is
// 64-byte aligned per image and padded by 64 bytes (build_batch), so the four 16-byte loads of a piece and the byte behind
// it stay inside the image's own region.
__device__ __forceinline__ uint32_t destuff_keep_mask(const uint8_t *raw, uint64_t i0, uint64_t raw_len, uint4 q[4],
                                                      uint64_t *mask_out, bool restarts, uint64_t *rst_out)
{
    uint64_t mask = 0, rst = 0;
    if (i0 < raw_len) {
        const uint4 *src = reinterpret_cast<const uint4 *>(raw + i0);
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = src[k];
        uint32_t prev = i0 > 0 ? raw[i0 - 1] : 0u;
        const uint32_t n = uint32_t(min(uint64_t(64), raw_len - i0));
        const uint32_t behind = i0 + 64 < raw_len ? raw[i0 + 64] : 0u;       // (a marker may straddle two pieces)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t b = (w[j >> 2] >> ((j & 3) * 8)) & 0xffu;
                const uint32_t jj = uint32_t(k * 16 + j);
                const uint32_t next = j < 15 ? (w[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 0xffu
                                             : (k < 3 ? (k == 0 ? q[1].x : k == 1 ? q[2].x : q[3].x) & 0xffu : behind);
                bool keep = !(b == 0x00u && prev == 0xffu);
                if (restarts) {
                    const bool marker = b == 0xffu && (next & 0xf8u) == 0xd0u && jj + 1 < n + (i0 + 64 < raw_len ? 1u : 0u);
                    if (marker) { keep = false; if (jj < n) rst |= 1ull << jj; }
                    if ((b & 0xf8u) == 0xd0u && prev == 0xffu) keep = false;
                }
                if (keep && jj < n) mask |= 1ull << jj;
                prev = b;
            }
        }
    }
    *mask_out = mask;
    *rst_out = rst;
    return uint32_t(__popcll(mask));
}

extern "C" __global__ __launch_bounds__(256) void k_destuff_count(const DestuffImg *imgs, const uint8_t *raw,
                                                                   uint2 *segcount)
{
    __shared__ uint32_t s_tmp[4];
    const DestuffImg im = imgs[blockIdx.y];
    if (blockIdx.x >= im.nseg) return;
    uint64_t mask, rst;
    uint4 q[4];
    const uint32_t cnt = destuff_keep_mask(raw + im.raw_off, uint64_t(blockIdx.x) * kDestuffSeg + threadIdx.x * 64ull, im.raw_len, q, &mask,
                                           im.restarts != 0, &rst);
    uint32_t total, total_rst;
    (void)wg_exclusive_scan(cnt, s_tmp, &total);
    (void)wg_exclusive_scan(uint32_t(__popcll(rst)), s_tmp, &total_rst);
    if (threadIdx.x == 0) segcount[im.seg0 + blockIdx.x] = make_uint2(total, total_rst);
}

// One workgroup per scan: segment bases, the scan's de-stuffed length, the picture's geometry.
extern "C" __global__ __launch_bounds__(256) void k_destuff_prefix(const DestuffImg *imgs, const uint2 *segcount, uint2 *segbase,
                                                                    DevImage *images, InterleaveImg *ii, uint32_t *img_flags)
{
    __shared__ uint32_t s_tmp[4];
    const DestuffImg im = imgs[blockIdx.x];
    uint32_t run = 0, run_rst = 0;
    for (uint32_t g0 = 0; g0 < im.nseg; g0 += 256) {
        const uint32_t g = g0 + threadIdx.x;
        const uint2 c = g < im.nseg ? segcount[im.seg0 + g] : make_uint2(0u, 0u);
        uint32_t total, total_rst;
        const uint32_t ex = wg_exclusive_scan(c.x, s_tmp, &total);
        const uint32_t exr = wg_exclusive_scan(c.y, s_tmp, &total_rst);
        if (g < im.nseg) segbase[im.seg0 + g] = make_uint2(run + ex, run_rst + exr);
        run += total;
        run_rst += total_rst;
    }
    if (threadIdx.x == 0) {
        DevImage &d = images[im.image];
        if (!im.direct) ii[im.ii_index].lin_len = run;
        d.himg.total_bits = run * 8u;
        d.n_rst_found = run_rst;
        if (d.nseg <= 1) d.himg.nsub = (run * 8u + d.himg.sub_bits - 1) / d.himg.sub_bits;      // (else: k_restart_geometry)
        if (run == 0) { img_flags[d.status_idx] = 1u; d.upload_short = 1u; }   // nothing but stuffing: reported as truncated
    }
}

// The workgroup compacts its 16 KiB segment into LDS (each lane drops its kept bytes at its scan offset), then writes
// the compacted bytes out in lane order, so a wave stores 64 consecutive bytes per instruction.
// DestuffImg::direct (round 5): the kept bytes go straight to their places in the picture's lane-interleaved region -- byte r of
// subsequence s at piece (r >> 4) of column s, and the first kLookPieces pieces of a subsequence once more behind the column in
// front of it (LaneBits) -- instead of into a linear copy that k_scan_interleave would read again; the region was filled with
// 0xAA (huffman.rs:236-246) before the launch.
extern "C" __global__ __launch_bounds__(256) void k_destuff_scatter(const DestuffImg *imgs, const uint8_t *raw,
                                                                     const uint2 *segbase, uint8_t *pool, uint32_t *rst_off,
                                                                     const DevImage *images, uint8_t *scan_pool)
{
    __shared__ uint32_t s_tmp[4];
    __shared__ uint8_t s_out[kDestuffSeg];
    const DestuffImg im = imgs[blockIdx.y];
    if (blockIdx.x >= im.nseg) return;
    const uint64_t i0 = uint64_t(blockIdx.x) * kDestuffSeg + threadIdx.x * 64ull;
    uint64_t mask, rst;
    uint4 q[4];
    const uint32_t cnt = destuff_keep_mask(raw + im.raw_off, i0, im.raw_len, q, &mask, im.restarts != 0, &rst);
    const uint2 base = segbase[im.seg0 + blockIdx.x];
    uint32_t total, total_rst;
    uint32_t o = wg_exclusive_scan(cnt, s_tmp, &total);
    if (im.restarts) {
        uint32_t r = base.y + wg_exclusive_scan(uint32_t(__popcll(rst)), s_tmp, &total_rst);
        for (uint64_t m = rst; m; m &= m - 1) {                             // where the interval behind each marker begins
            const uint32_t j = uint32_t(__ffsll((long long)m)) - 1u;
            if (r < im.rst_cap) rst_off[im.rst0 + r] = base.x + o + uint32_t(__popcll(mask & ((1ull << j) - 1ull)));
            r++;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((mask >> (k * 16 + j)) & 1) s_out[o++] = uint8_t(w[j >> 2] >> ((j & 3) * 8));
    }
    __syncthreads();
    if (im.direct) {
        const DevImage &d = images[im.image];
        const uint32_t sub_bytes = d.himg.sub_bits >> 3, cols = d.scan_cols, own_rows = d.himg.sub_bits >> 7;
        uint8_t *region = scan_pool + d.scan_off;
        for (uint32_t i = threadIdx.x; i < total; i += 256) {
            const uint32_t g = base.x + i, s = g / sub_bytes, r = g - s * sub_bytes;
            const uint8_t v = s_out[i];
            region[(size_t(r >> 4) * cols + s) * 16u + (r & 15u)] = v;
            // ... and into the look-ahead rows of the columns in front: byte r of subsequence s is byte (j - 1) * sub_bytes + r behind the
            // end of subsequence s - j.  (Subsequences of 64 bytes and more: j = 1 only.  Shorter ones -- MJX_FIT_SHORT below 512 bits --
            // need their look-ahead from two or more followers; the single copy left those bytes 0xAA.  Round-5 advisor.)
            for (uint32_t j = 1u, at = r; j <= s && at < kLookPieces * 16u; j++, at += sub_bytes)
                region[(size_t(own_rows + (at >> 4)) * cols + (s - j)) * 16u + (at & 15u)] = v;
        }
        return;
    }
    uint8_t *dst = pool + im.out_off + base.x;
    for (uint32_t i = threadIdx.x; i < total; i += 256) dst[i] = s_out[i];
}

// Restart intervals of a scan that was de-stuffed on the device: the marker list -> the picture's segment table, as
// plan_image builds it from mjx_scan_desc.restart_offsets on the host -- segs[g] = (first subsequence, first bit) of
// segment g, a segment of L bits takes ceil(L / sub_bits) subsequences (one when it is empty), and a sentinel (number
// of subsequences, total bits).  Fewer markers than the picture's MCU count needs: truncated.  One workgroup per picture.
extern "C" __global__ __launch_bounds__(256) void k_restart_geometry(const DestuffImg *imgs, DevImage *images, const uint32_t *rst_off,
                                                                      uint32_t *segs, uint32_t *img_flags)
{
    __shared__ uint32_t s_tmp[4];
    const DestuffImg im = imgs[blockIdx.x];
    if (!im.restarts) return;
    DevImage &d = images[im.image];
    const uint32_t nseg = d.nseg, total_bits = d.himg.total_bits, sub_bits = d.himg.sub_bits;
    const uint32_t found = min(d.n_rst_found, im.rst_cap);
    uint2 *sg = reinterpret_cast<uint2 *>(segs) + d.seg_off;
    auto bit0 = [&](uint32_t g) { return g == 0 ? 0u : (g - 1 < found ? min(rst_off[im.rst0 + g - 1] * 8u, total_bits) : total_bits); };
    uint32_t run = 0;
    for (uint32_t g0 = 0; g0 < nseg; g0 += 256) {
        const uint32_t g = g0 + threadIdx.x;
        uint32_t subs = 0, b0 = 0;
        if (g < nseg) {
            b0 = bit0(g);
            const uint32_t b1 = g + 1 == nseg ? total_bits : bit0(g + 1);
            const uint32_t len = b1 > b0 ? b1 - b0 : 0u;
            subs = len ? (len + sub_bits - 1) / sub_bits : 1u;
        }
        uint32_t total;
        const uint32_t ex = wg_exclusive_scan(subs, s_tmp, &total);
        if (g < nseg) sg[g] = make_uint2(run + ex, b0);
        run += total;
    }
    if (threadIdx.x == 0) {
        sg[nseg] = make_uint2(run, total_bits);
        d.himg.nsub = run;
        if (found + 1 < nseg) { img_flags[d.status_idx] = 1u; d.upload_short = 1u; }   // fewer RSTn markers than intervals
    }
}

// ---- lane-interleaved scan pool (see LaneBits) --------------------------------------------------------------------
// One lane per 16-byte piece of the region: piece (row k, column s) = bytes [start_s + 16 k, + 16) of the linear
// de-stuffed scan, where start_s is the first byte of subsequence s; 0xAA where the scan has ended (huffman.rs:236-246)
// and in the padding columns.  Writes are fully coalesced; the reads of a wave are 64 pieces a subsequence apart, each
// line is read by eight waves of the launch (L2 absorbs that).  Runs once per upload, before any decode.
extern "C" __global__ __launch_bounds__(256) void k_scan_interleave(const InterleaveImg *imgs, const DevImage *images,
                                                                     const uint8_t *linear, uint8_t *pool, const uint32_t *segs)
{
    const InterleaveImg ii = imgs[blockIdx.y];
    const DevImage &im = images[ii.image];
    const uint32_t cols = im.scan_cols, rows = scan_region_rows(im.himg.sub_bits);
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= cols * rows) return;
    const uint32_t k = q / cols, s = q % cols;
    uint32_t w[4] = {0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau};
    if (s < im.himg.nsub) {
        const SubLoc loc = locate_sub(im, im.himg, segs, s);
        const uint64_t at = uint64_t(loc.start >> 3) + 16u * k;              // (subsequences start on byte boundaries)
        const uint8_t *src = linear + ii.lin_off;
        if (at + 16 <= ii.lin_len && ((ii.lin_off + at) & 3u) == 0) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(src + at);
            w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; w[3] = p[3];
        } else if (at < ii.lin_len) {
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) {
                const uint32_t b = at + j < ii.lin_len ? src[at + j] : 0xaau;
                w[j >> 2] = (w[j >> 2] & ~(0xffu << ((j & 3) * 8))) | (b << ((j & 3) * 8));
            }
        }
    }
    *reinterpret_cast<uint4 *>(pool + im.scan_off + uint64_t(q) * 16u) = make_uint4(w[0], w[1], w[2], w[3]);
}

// blkbase[s] / ebase[s] = blocks completed / stream entries produced before subsequence s (one workgroup per image).
// A subsequence's run of stream entries is rounded up to whole store groups (the write pass fills up with null entries).
// Pictures whose first decode emits: what k_huff_prefix decodes again is listed here -- one item per checkpoint interval of the
// prefix of every subsequence whose entry was wrong: (subsequence, interval).  The merge rounds have left the true path's state
// and counts at every checkpoint in front of the merge point, so the intervals of one prefix are independent pieces of work of the
// same length (~200 symbols at 1024 bits): no lane of the prefix pass waits for a neighbour that decodes a whole subsequence.
constexpr uint32_t kItemShift = 20;           // item = subsequence | interval << kItemShift
__device__ __forceinline__ uint32_t prefix_intervals(const DevImage &im, uint32_t s, uint32_t kfix)
{
    if (kfix != kEmitAll) return kfix;
    const uint32_t L = im.himg.sub_bits, start = s * L, len = min(start + L, im.himg.total_bits) - start;
    return max(1u, (len + im.himg.cp_bits - 1u) / im.himg.cp_bits);
}
extern "C" __global__ __launch_bounds__(256) void k_huff_scan(const DevImage *images, SubseqState *g_exit,
                                                               uint32_t *g_blkbase, uint32_t *g_ebase,
                                                               uint32_t *img_entries, uint32_t *img_flags,
                                                               const uint32_t *segs, const uint32_t *verdict,
                                                               const EmitSub *g_esub, uint32_t *g_items, uint32_t *g_item_count,
                                                               uint32_t *fallback, uint32_t *unconverged, SubseqState *g_entry,
                                                               uint8_t *g_gen, uint32_t gen_stride)
{
    __shared__ uint32_t s_tmp[4];
    const DevImage &im = images[blockIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0 && verdict && *verdict != 0) atomicAdd(unconverged, 1u);      // (one count per run of the chunk)
    if (!im.valid || im.role == 2) return;          // (role 2: a multi-scan picture, flagged by k_planar_gather)
    // `verdict` = what the last enqueued synchronisation round of this chunk re-decoded.  Non-zero: the entries are not
    // yet the fixed point (the host finds out at mjx_batch_wait and runs more rounds), block and entry counts of
    // neighbouring subsequences do not fit together, and the kernels behind this one would write a stream with holes
    // -- tile offsets nobody wrote -- and read it.  The chunk's images are flagged instead; every later kernel skips
    // flagged images, and the repair run clears the flag.
    if (verdict && *verdict != 0) {
        if (threadIdx.x == 0) {
            img_entries[im.status_idx] = 0;
            img_flags[im.status_idx] = 2u;
        }
        return;
    }
    const uint32_t nsub = im.himg.nsub, tid = threadIdx.x;
    const uint32_t per = (nsub + kWgLanes - 1) / kWgLanes;
    const uint32_t a = min(nsub, tid * per), b = min(nsub, a + per);
    if (gen_stride && !im.emit) {
        // the rounds are over: where a subsequence's current decode lies in the second set (Gen2), its entry and exit move to the
        // first -- everything behind this kernel reads the first set only
        // (the bytes stay as they are: the next decode of the chunk clears them all, and no round runs between this fold and then --
        // the fold itself may run again; four bytes in flight at a time: the thread's subsequences are few and the loads dependent
        // on nothing)
        for (uint32_t s0 = a; s0 < b; s0 += 4) {
            uint32_t gq[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) gq[q] = s0 + q < b ? g_gen[im.sub_off + s0 + q] : 0u;
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t idx = im.sub_off + s0 + q;
                if (gq[q] & 1u) {
                    g_entry[idx] = g_entry[idx + gen_stride];
                    g_exit[idx] = g_exit[idx + gen_stride];
                }
            }
        }
    }
    uint32_t sum_n = 0, sum_m = 0;
    for (uint32_t s = a; s < b; s++) { sum_n += g_exit[im.sub_off + s].n; sum_m += stream_run(g_exit[im.sub_off + s].m); }
    uint32_t total_n, total_m;
    uint32_t run_n = wg_exclusive_scan(sum_n, s_tmp, &total_n);
    uint32_t run_m = wg_exclusive_scan(sum_m, s_tmp, &total_m);
    for (uint32_t s = a; s < b; s++) {
        g_blkbase[im.sub_off + s] = run_n;
        g_ebase[im.sub_off + s] = run_m;
        run_n += g_exit[im.sub_off + s].n;
        run_m += stream_run(g_exit[im.sub_off + s].m);
    }
    // The scan ends before every MCU is decoded (truncated file): the reference would go on decoding its 0xAA padding
    // (huffman.rs:236-246); here the image is reported as truncated and skipped by the later kernels.  With restart
    // intervals every segment must hold its own blocks (restart_mcus MCUs, the last one what is left): a short segment
    // would leave tiles without offsets.
    __shared__ uint32_t s_short;
    if (tid == 0) s_short = (im.nseg <= 1 && total_n < im.himg.total_blocks) ? 1u : 0u;
    __threadfence_block();
    __syncthreads();                                                       // g_blkbase of this image is complete
    if (im.nseg > 1) {
        const uint2 *sg = reinterpret_cast<const uint2 *>(segs) + im.seg_off;
        const uint32_t seg_blocks = im.restart_mcus * im.bpm;
        for (uint32_t g = tid; g < im.nseg; g += kWgLanes) {
            const uint32_t s0 = sg[g].x, s1 = sg[g + 1].x;
            const uint32_t n0 = g_blkbase[im.sub_off + s0], n1 = s1 < nsub ? g_blkbase[im.sub_off + s1] : total_n;
            const uint32_t want = min(seg_blocks, im.himg.total_blocks - g * seg_blocks);
            if (n1 - n0 < want) atomicOr(&s_short, 1u);
        }
        __syncthreads();
    }
    if (tid == 0) {
        img_entries[im.status_idx] = total_m;      // upper bound of the entries the write pass produces
        img_flags[im.status_idx] = s_short | im.upload_short;     // (a repair run comes through here again: the upload-time diagnosis stays)
    }
    if (im.emit) {
        uint32_t cnt = 0;
        for (uint32_t s = a; s < b; s++) {
            const uint32_t kf = g_esub[im.sub_off + s].kfix;
            if (kf) cnt += prefix_intervals(im, s, kf);
        }
        uint32_t total;
        uint32_t at = wg_exclusive_scan(cnt, s_tmp, &total);
        // the list has room for kItemDwords items per subsequence (the merge rounds' straggler list); a picture that needs more -- most
        // of its lanes decode most of their subsequence again: noise at quality 99 -- is better off with the two-pass kernels anyway
        if (total > nsub * uint32_t(kItemDwords) || nsub >= (1u << kItemShift)) {
            // (the first picture of this run that falls back also counts the run: its pictures are skipped by everything behind, and a
            // caller who enqueues several decodes before one wait learns of it from mjx_batch_unconverged_runs -- round-5 advisor)
            if (tid == 0) { img_flags[im.status_idx] = 2u; if (atomicOr(fallback, 64u) == 0u) atomicAdd(unconverged, 1u); g_item_count[blockIdx.x] = 0; }
            return;
        }
        uint32_t *items = g_items + size_t(im.sub_off) * kItemDwords;
        for (uint32_t s = a; s < b; s++) {
            const uint32_t kf = g_esub[im.sub_off + s].kfix;
            if (!kf) continue;
            const uint32_t n = prefix_intervals(im, s, kf);
            for (uint32_t k = 0; k < n; k++) items[at++] = s | (k << kItemShift);
        }
        if (tid == 0) g_item_count[blockIdx.x] = total;
    }
}

extern "C" __global__ __launch_bounds__(kHuffWg) void k_huff_write(const DevImage *images, const uint8_t *scan_pool,
                                                                const LutEntry *lut_pool, const SubseqState *g_entry,
                                                                const uint32_t *g_blkbase, const uint32_t *g_ebase,
                                                                uint32_t *entries, uint32_t *tile_eoff, int16_t *dcdiff,
                                                                int *status, const uint32_t *img_flags, uint32_t win_off,
                                                                const uint32_t *segs, const SubseqState *g_exit, uint32_t *g_cps)
{
    extern __shared__ __attribute__((aligned(kLutAlign))) unsigned char smem[];   // tables, HuffImage, windows, rings
    uint32_t *s_win = reinterpret_cast<uint32_t *>(smem + win_off);
    const uint32_t img = entropy_grid_image(), wgi = entropy_grid_wg();
    const DevImage &im = images[img];
    if (!im.valid || im.emit || wgi * blockDim.x >= im.himg.nsub || img_flags[im.status_idx]) return;      // (blockDim.x: see k_huff_spec)
    const HuffImage *h;
    const LutEntry *lut;
    stage_tables(im, lut_pool, smem, h, lut);
    const uint32_t s = wgi * blockDim.x + threadIdx.x;
    const bool live = s < h->nsub;
    const LaneBits gbits{scan_pool + im.scan_off, (live ? s : 0u) * 16u, im.scan_cols * 16u};
    SubseqState e = make_state(0, 0, 0);
    uint32_t blk = 0, end_bit = 0, ebase = 0, blk_limit = h->total_blocks, pad_to = 0;
    bool live_entry = false;
    if (live) {
        const SubLoc loc = locate_sub(im, *h, segs, s);
        e = g_entry[im.sub_off + s];
        live_entry = e.p >= loc.start;                                     // (always, once the rounds have converged)
        e.p -= loc.start;                                                  // the lane works relative to its subsequence
        blk = g_blkbase[im.sub_off + s];
        ebase = g_ebase[im.sub_off + s];
        end_bit = loc.end - loc.start;
        if (im.nseg > 1) {
            // blocks are counted from the segment's first one (restart_mcus MCUs per segment); the lane stops at the
            // segment's last block
            const uint32_t seg_blocks = im.restart_mcus * im.bpm, seg_first = loc.seg * seg_blocks;
            blk = seg_first + (blk - g_blkbase[im.sub_off + loc.seg_sub0]);
            blk_limit = min(seg_first + seg_blocks, h->total_blocks);
        }
        // the lane's run in the stream is what the synchronisation passes counted for it, rounded up to whole store
        // groups: what it does not produce (group padding, blocks past the last one, garbage after a segment's last
        // block) is filled with null entries (position 0, skipped by stage B), so the stream has no holes and every
        // store is a whole aligned group
        pad_to = ebase + stream_run(g_exit[im.sub_off + s].m);
        pad_to = pad_to < im.ent_cap ? pad_to : im.ent_cap;
        if (im.ent_rows) pad_to = min(stream_run(g_exit[im.sub_off + s].m), im.ent_rows * 8u);     // quad-interleaved: the lane's column starts at 0
    }
    StreamSink sink;
    sink.tile_eoff = tile_eoff + im.tile_off;
    sink.status = status + im.status_idx;
    const uint32_t first_start = blk + (e.z ? 1u : 0u);                    // first block whose DC this lane decodes
    {
        uint32_t *rings = s_win + blockDim.x * kWinStride;
        if (im.ent_rows) {          // quad-interleaved stream: the lane's column (stream_phys(s, 0)), groups four groups apart
            const uint32_t sl = live ? s : 0u;
            sink.ac_ring.begin(rings + threadIdx.x * kAcRingStride, entries + im.ent_off + im.ent_hdr + stream_phys(sl, 0, im.ent_rows), 0u);
            if (live) (entries + im.ent_off)[s] = run_word(0u, pad_to >> 3, 0u);      // the run in groups, for stage B (labels = block indices: no offset)
            sink.ac_ring.gstride = kAcGroup * kStreamQuad;
            sink.tile_virt = sl * (im.ent_rows * 8u);
        } else {
            sink.ac_ring.begin(rings + threadIdx.x * kAcRingStride, entries + im.ent_off, ebase);
            sink.tile_virt = 0;
        }
        rings += blockDim.x * kAcRingStride;
        sink.dc_ring.begin(rings + threadIdx.x * (DcRing16::kRing / 2), dcdiff + im.coef_off, first_start);
    }
    sink.blk_bits = StreamSink::block_bits(blk);
    sink.tile_blocks = im.tile_blocks;
    sink.total_blocks = h->total_blocks;
    sink.ntiles = (h->total_blocks + im.tile_blocks - 1) / im.tile_blocks;
    sink.seg_im = im.seg_S ? &im : nullptr;
    if (im.seg_S) {
        sink.ntiles = im.mcuy * im.seg_S;                                  // (the sentinel's slot)
        sink.next_cut(first_start);
    } else {
        sink.tile_idx = (first_start + im.tile_blocks - 1) / im.tile_blocks;
        sink.next_tile_blk = sink.tile_idx * im.tile_blocks;
    }
#ifdef MJX_EXP_WRITE_CP      // (measurement build: what an emitting pass pays for recording checkpoints as the counting pass does)
    GlobalCps cpw{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(im.sub_off + (live ? s : 0u)), 0};
    wave_decode<true, 1>(live && live_entry, e, end_bit, blk, blk_limit, gbits, s_win + threadIdx.x * kWinStride, lut, *h, sink, cpw, 0, e);
#else
    (void)g_cps;
    NoCheckpoints nocp;
    wave_decode<true, 0>(live && live_entry, e, end_bit, blk, blk_limit, gbits, s_win + threadIdx.x * kWinStride, lut, *h, sink, nocp, 0, e);
#endif
    sink.flush_groups();                                                   // (the rings hold one flush period, no more)
    for (uint32_t it = 1; __builtin_amdgcn_ballot_w64(sink.ac_ring.off < pad_to); it++) {
        if (sink.ac_ring.off < pad_to) sink.ac_ring.push(0u);              // null entry
        if (it % kFlushEvery == 0) sink.flush_groups();
    }
    sink.flush();
}

// ---- single decode (round 5): the first decode emits -----------------------------------------------------------------------
// The reference decodes every symbol once (huffman.rs:146-195).  The two-pass path above decodes every symbol twice -- k_huff_spec
// to find where the subsequences begin, k_huff_write to emit from there -- because an emitting lane needs to know which block of
// the picture it is in, and that is only known once all the subsequences before it have been counted.  Here the first decode
// emits without knowing:
//   k_huff_emit    A lane first WARMS UP: it decodes the last warm_bits bits of the subsequence in front of its own from the guess
//                  "a block starts here", counting only; Huffman codes self-synchronise, so at its own subsequence's start it
//                  is on the true path with probability ~0.8 (1024 bits) .. 0.94 (2048).  Then it decodes its subsequence ONCE,
//                  emitting: entries into its column from index H on, labelled with the number of blocks the lane has completed
//                  (coef_entry's block field); one word per block -- {DC difference, column index where its entries begin} -- into
//                  the top of the column, downwards (block_word_index), from index Hb + label on; checkpoints every cp_bits.
//   merge rounds   as before (counting): the true entry state of every subsequence; per subsequence the deepest checkpoint at which
//                  a re-decode met the recorded path (EmitSub::kfix) -- from there on what k_huff_emit wrote is the true decode.
//   k_huff_scan    blocks completed before every subsequence.
//   k_huff_prefix  the lanes whose entry was wrong (a fifth at 1024 bits of warm-up) decode from the true entry up to that
//                  checkpoint, emitting RIGHT-ALIGNED against the first valid entry / block word of the first decode, with labels
//                  that continue into that decode's: the subsequence's run in the column stays one contiguous run.  Every
//                  subsequence's run word (first group, end group, label offset) goes to the head of the stream region.
//   k_block_gather block words -> dcdiff[] (what the DC prediction kernels read) and the tile offsets stage B starts from.
// Stage B adds the run's label offset to its entries' labels; nothing else changes for it.
struct BlkRing {
    static constexpr uint32_t kRing = 8, kGroup = 4;      // words: 32 bytes of LDS per lane, 16-byte groups
    uint32_t *ring;
    uint32_t *col;          // the lane's column: word j of it lies at col[(j >> 3) * 8 * kStreamQuad + (j & 7)]
    uint32_t quads;         // column capacity / 4 (block word group g lies at column index 4 * (quads - 1 - g))
    uint32_t off, flushed;  // block word indices: next, first not yet in HBM
    // first: index of the first word that will be pushed.  The run starts on the group boundary at or below it: a word in front of
    // `first` in that group (the slot of a block whose DC symbol lay in the subsequence before) leaves with the group as whatever the
    // ring holds -- nobody reads it --, and the flush needs no word-by-word lead-in to the first boundary.
    __device__ __forceinline__ void begin(uint32_t *lds, uint32_t *column, uint32_t rows, uint32_t first)
    {
        ring = lds;
        col = column;
        quads = rows * 2u;
        off = first;
        flushed = first & ~(kGroup - 1);
    }
    __device__ __forceinline__ uint32_t *group_at(uint32_t i) const
    {
        const uint32_t t = quads - 1u - (i >> 2);
        return col + (t >> 1) * (8u * kStreamQuad) + (t & 1u) * 4u;
    }
    __device__ __forceinline__ void push(uint32_t v)
    {
        ring[off & (kRing - 1)] = v;
        off++;
    }
    // (called every kFlushEvery symbols: at most kFlushEvery / 2 words have arrived on top of the < kGroup that waited -- one group at most)
    __device__ __forceinline__ void flush_groups()
    {
        if (__builtin_amdgcn_ballot_w64(flushed + kGroup <= off)) {
            if (flushed + kGroup <= off) {
                *reinterpret_cast<uint4 *>(group_at(flushed)) = *reinterpret_cast<const uint4 *>(ring + (flushed & (kRing - 1)));
                flushed += kGroup;
            }
        }
    }
    __device__ __forceinline__ void flush_all()
    {
        flush_groups();
        flush_groups();
        for (uint32_t i = flushed; i < off; i++) group_at(i)[i & 3u] = ring[i & (kRing - 1)];
        flushed = off;
    }
};
static_assert(BlkRing::kGroup - 1 + kFlushEvery / 2 < 2 * BlkRing::kGroup && 2 * BlkRing::kGroup <= BlkRing::kRing, "block ring: a block takes two symbols at least, the ring is flushed every kFlushEvery symbols");

struct EmitSink {
    LaneRing<kAcGroup, true> ac_ring;   // index = entry index in the lane's column
    BlkRing blk_ring;                   // index = head room + label of the block
    uint32_t blk_bits;                  // the current block's label, placed as in coef_entry, + 63 << 16 (see StreamSink)
    uint32_t bad_pos, bad_lbl;
#if defined(MJX_EXP_EMIT_NOBLK)        // (measurement build, garbage out: the emitting pass without its block words)
    __device__ __forceinline__ void dc(uint32_t, int v) { asm volatile("" :: "v"(v)); }
#else
    __device__ __forceinline__ void dc(uint32_t, int v) { blk_ring.push((uint32_t(v) & 0xffffu) | (ac_ring.off << 16)); }
#endif
    __device__ __forceinline__ void ac(uint32_t, uint32_t r_scaled, int v) { ac_ring.push((uint32_t(v) & 0xffffu) | (blk_bits - r_scaled)); }
    __device__ __forceinline__ void block_done(uint32_t) { blk_bits += 1u << 22; }
    __device__ __forceinline__ void bad_code(uint32_t b, uint32_t pos) { if (bad_pos == 0xffffffffu) { bad_pos = pos; bad_lbl = b; } }
    __device__ __forceinline__ void tick() const {}
    __device__ __forceinline__ void flush_groups() { ac_ring.flush_groups(); blk_ring.flush_groups(); }
#if defined(MJX_EXP_EMIT_NOBLK)
    __device__ __forceinline__ void flush_step(uint32_t) { ac_ring.flush_groups(); }
#else
    __device__ __forceinline__ void flush_step(uint32_t) { ac_ring.flush_groups(); blk_ring.flush_groups(); }
#endif
    __device__ __forceinline__ void flush_entries() { ac_ring.flush_groups(); }
    __device__ __forceinline__ void flush_dc(uint32_t) { blk_ring.flush_groups(); }
    __device__ __forceinline__ void flush() { ac_ring.flush_groups(); blk_ring.flush_all(); }
};

extern "C" __global__ __launch_bounds__(kHuffWg) void k_huff_emit(const DevImage *images, const uint8_t *scan_pool,
                                                               const LutEntry *lut_pool, SubseqState *g_entry, SubseqState *g_exit,
                                                               uint32_t *g_cps, EmitSub *g_esub, uint32_t *entries, uint32_t win_off)
{
    extern __shared__ __attribute__((aligned(kLutAlign))) unsigned char smem[];   // tables, HuffImage, windows, rings
    uint32_t *s_win = reinterpret_cast<uint32_t *>(smem + win_off);
    // (round 6: a raised wave priority for this latency-bound pass beside stage B -- s_setprio 1 / 3 -- measured 25.57 / 25.43 against
    // 25.63 ms per step, five passes of 20 steps each on one box, spread 0.25: nothing)
    const uint32_t img = entropy_grid_image(), wgi = entropy_grid_wg();
    const DevImage &im = images[img];
    // (blockDim.x lanes: 512, or -- round 6 -- 256 / 128 for a chunk whose scans are all that short: see k_huff_spec)
    if (!im.valid || !im.emit || wgi * blockDim.x >= im.himg.nsub) return;
    const HuffImage *h;
    const LutEntry *lut;
    stage_tables(im, lut_pool, smem, h, lut);
    const uint32_t s = wgi * blockDim.x + threadIdx.x;
    const bool live = s < h->nsub;
    const uint32_t sl = live ? s : 0u, L = h->sub_bits;
    const uint32_t start = sl * L, end = min(start + L, h->total_bits);
    uint32_t *my_win = s_win + threadIdx.x * kWinStride;
    // ---- warm-up over the end of the subsequence in front (its column of the pool), counting only
    SubseqState e0 = make_state(0, 0, 0);
    const uint32_t W = h->warm_bits;
    if (W) {                                                                        // (uniform over the launch)
        const bool warm = live && s > 0;
        const LaneBits prev{scan_pool + im.scan_off, (warm ? s - 1u : 0u) * 16u, im.scan_cols * 16u};
        NullSink ns;
        NoCheckpoints nocp;
        const SubseqState w0 = make_state(L - W, 0, 0);
        const SubseqState x = wave_decode<false, 0, false>(warm, w0, L, 0, 0xffffffffu, prev, my_win, lut, *h, ns, nocp, 0u, w0);
        if (warm) e0 = make_state(x.p - L, x.z, x.c);
    }
    // ---- the subsequence itself, once, emitting
    const LaneBits gbits{scan_pool + im.scan_off, sl * 16u, im.scan_cols * 16u};
    const uint32_t H = im.emit_head * 8u, Hb = im.emit_head * 4u;
    uint32_t *column = entries + im.ent_off + im.ent_hdr + stream_phys(sl, 0, im.ent_rows);
    EmitSink sink;
    {
        uint32_t *rings = s_win + blockDim.x * kWinStride;
        sink.ac_ring.begin(rings + threadIdx.x * kAcRingStride, column, H);
        sink.ac_ring.gstride = kAcGroup * kStreamQuad;
        rings += blockDim.x * kAcRingStride;
        sink.blk_ring.begin(rings + threadIdx.x * BlkRing::kRing, column, im.ent_rows, Hb + (e0.z ? 1u : 0u));
    }
    sink.blk_bits = StreamSink::block_bits(0);
    sink.bad_pos = sink.bad_lbl = 0xffffffffu;
    GlobalCps cps{reinterpret_cast<unsigned char *>(g_cps), cps_byte_off(im.sub_off + sl), 0};
#if defined(MJX_EXP_EMIT_NOCP)         // (measurement build, garbage out: the emitting pass without recording checkpoints)
    NoCheckpoints nocp2;
    (void)cps;
    SubseqState x = wave_decode<true, 0, false>(live, e0, end - start, 0, 0xffffffffu, gbits, my_win, lut, *h, sink, nocp2, 0u, e0);
#else
    SubseqState x = wave_decode<true, 1, false>(live, e0, end - start, 0, 0xffffffffu, gbits, my_win, lut, *h, sink, cps, 0u, e0);
#endif
    sink.flush_groups();                                                   // (the rings hold one flush period, no more)
    const uint32_t pad_to = live ? min(stream_run(sink.ac_ring.off), im.ent_rows * 8u) : 0u;
    for (uint32_t it = 1; __builtin_amdgcn_ballot_w64(sink.ac_ring.off < pad_to); it++) {
        if (sink.ac_ring.off < pad_to) sink.ac_ring.push(0u);              // null entries up to the group boundary
        if (it % kFlushEvery == 0) sink.flush_groups();
    }
    sink.flush();
    if (!live) return;
    g_entry[im.sub_off + s] = make_state(e0.p + start, e0.z, e0.c);
    x.p += start;
    g_exit[im.sub_off + s] = x;
    EmitSub es;
    es.d0n = x.n; es.d0m = x.m; es.kfix = 0; es.bad = sink.bad_pos; es.bad_lbl = sink.bad_lbl; es.lbl = 0; es.pad_[0] = es.pad_[1] = 0;
    g_esub[im.sub_off + s] = es;
}

// Fall-back: the pictures of a chunk that the single-decode kernels gave up on (flag 2) go to the two-pass kernels.  On the device, field by field: the DevImages of scans that were
// de-stuffed there hold geometry only the device knows (k_destuff_prefix, k_restart_geometry) -- the host's copies must not
// overwrite them.
extern "C" __global__ void k_emit_off(DevImage *images, uint32_t nimg, const uint32_t *img_flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nimg || !images[i].emit || img_flags[images[i].status_idx] != 2u) return;      // (2: the picture k_huff_scan / k_huff_prefix gave up on)
    images[i].emit = 0;
    images[i].himg.cp_bits = uint32_t(kCpBits);
    images[i].himg.warm_bits = 0;
}

// Direct sink of k_huff_prefix: few lanes, short runs -- dword stores.
struct PrefixSink {
    uint32_t *column;
    uint32_t rows, eoff, e_hi;   // next entry index; the prefix must end exactly at e_hi
    uint32_t blk_bits;           // label of the current block as in coef_entry + 63 << 16
    uint32_t bword0;             // block word index of label 0 of this decode
    uint32_t blk0, total_blocks; // the decode's first block in the picture
    int *status;
    bool over;
    __device__ __forceinline__ uint32_t *at(uint32_t j) const { return column + (j >> 3) * (8u * kStreamQuad) + (j & 7u); }
    __device__ __forceinline__ void dc(uint32_t b, int v) { *at(block_word_index(bword0 + b, rows)) = (uint32_t(v) & 0xffffu) | (eoff << 16); }
    __device__ __forceinline__ void ac(uint32_t, uint32_t r_scaled, int v)
    {
        if (eoff < e_hi) *at(eoff) = (uint32_t(v) & 0xffffu) | (blk_bits - r_scaled);
        else over = true;
        eoff++;
    }
    __device__ __forceinline__ void block_done(uint32_t) { blk_bits += 1u << 22; }
    __device__ __forceinline__ void bad_code(uint32_t b, uint32_t) const { if (blk0 + b < total_blocks) atomicOr(status, 1); }
    __device__ __forceinline__ void tick() const {}
};

constexpr uint32_t kPrefixWg = 64;           // one wave: it holds the tables and a window per lane, and leaves when its own items are done
extern "C" __global__ __launch_bounds__(kPrefixWg) void k_huff_prefix(const DevImage *images, const uint8_t *scan_pool,
                                                                    const LutEntry *lut_pool, const SubseqState *g_entry,
                                                                    const SubseqState *g_exit, const uint32_t *g_cps, EmitSub *g_esub,
                                                                    const uint32_t *g_blkbase, uint32_t *entries, int *status,
                                                                    uint32_t *img_flags, uint32_t *fallback, uint32_t win_off,
                                                                    const uint32_t *g_items, const uint32_t *g_item_count, uint32_t *unconverged)
{
    extern __shared__ __attribute__((aligned(kLutAlign))) unsigned char smem[];   // tables, HuffImage, a window per lane
    uint32_t *s_win = reinterpret_cast<uint32_t *>(smem + win_off);
    const uint32_t img = picture_of_slot(blockIdx.x, gridDim.x), group = blockIdx.y;
    const DevImage &im = images[img];
    if (!im.valid || !im.emit || img_flags[im.status_idx]) return;
    const uint32_t count = g_item_count[img];
    if (group * kPrefixWg >= count) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t L = im.himg.sub_bits, total_blocks = im.himg.total_blocks, cpb = im.himg.cp_bits;
    const uint32_t H = im.emit_head * 8u, Hb = im.emit_head * 4u, rows = im.ent_rows;
    uint32_t *runs = entries + im.ent_off;
    uint32_t *columns = entries + im.ent_off + im.ent_hdr;
    const bool need = group * kPrefixWg + tid < count;
    const uint32_t item = need ? (g_items + size_t(im.sub_off) * kItemDwords)[group * kPrefixWg + tid] : 0u;
    const uint32_t s = item & ((1u << kItemShift) - 1u), k = item >> kItemShift;
    const HuffImage *h;
    const LutEntry *lut;
    stage_tables(im, lut_pool, smem, h, lut);                              // (barriers inside: every lane of the workgroup comes here)
    if (!need) return;
    const EmitSub es = g_esub[im.sub_off + s];
    const SubseqState e = g_entry[im.sub_off + s], x = g_exit[im.sub_off + s];
    const uint32_t B = g_blkbase[im.sub_off + s];
    const uint32_t sub_start = s * L, end_rel = min(sub_start + L, h->total_bits) - sub_start;
    const GlobalCps cps{reinterpret_cast<unsigned char *>(const_cast<uint32_t *>(g_cps)), cps_byte_off(im.sub_off + s), 0};
    // the subsequence's prefix: where it ends (the merge point) and what the first decode had counted there
    const uint32_t nint = prefix_intervals(im, s, es.kfix);
    uint32_t n0r = 0, m0r = 0, recK = 0;
    if (es.kfix != kEmitAll) {
        const CpPair cp = cps.get_pair(es.kfix - 1u);                      // the first decode's record there: still its own (see merge_finish)
        recK = cp.w;
        n0r = (cp.w >> 16) & 0x7fffu;
        m0r = cp.m;
    }
    const uint32_t n0K = es.d0n - n0r, j0K = es.d0m - m0r, n_p = x.n - n0r, m_p = x.m - m0r;
    // head room: the prefix ends where the valid part of the first decode begins
    // (why a picture fell back, for MJX_TIMING: 1 no head room, 2 counts of the first decode / the merge rounds do not fit together,
    // 4 a record of the true path is missing, 8 the item did not end in the recorded state, 16 ... with the recorded counts,
    // 32 it ran past the prefix's end; k_huff_scan: 64 more items than the list holds)
    uint32_t why = 0;
    if (m_p > H + j0K || n_p > Hb + n0K) why |= 1u;
    if (x.n < n0r || x.m < m0r || es.d0n < n0r || es.d0m < m0r || k >= nint) why |= 2u;
    bool failed = why != 0;
    const uint32_t off_e = H + j0K - m_p;
    const int32_t lbl = int32_t(n0K) - int32_t(n_p);
    uint32_t *column = columns + stream_phys(s, 0, rows);
    // this item: the true path from checkpoint k - 1 (the entry for k == 0) to checkpoint k (the merge point / the end for the last one)
    uint32_t p, zc, n_start = 0, m_start = 0;
    if (k == 0) {
        p = e.p - sub_start;
        zc = e.z | (uint32_t(e.c) << 8);
        if (e.p < sub_start) { failed = true; why |= 2u; }
    } else {
        const CpPair cp = cps.get_pair(k - 1u);                            // recorded by the merge rounds on the true path, counts to the end of the subsequence
        const uint32_t nbn = (cp.w >> 12) & 0xfu, c1 = (nbn ? nbn : h->bpm) - 1u;
        p = 8u * (wn_after(k * cpb) - 8u) - (cp.w & 31u);
        zc = (64u - ((cp.w >> 5) & 0x7fu)) | (((c1 ? c1 : h->bpm) - 1u) << 8);
        n_start = x.n - ((cp.w >> 16) & 0x7fffu);
        m_start = x.m - cp.m;
        if (!(cp.w & kCpValid) || ((cp.w >> 16) & 0x7fffu) > x.n || cp.m > x.m) { failed = true; why |= 4u; }
    }
    const bool last = k + 1u == nint;
    const uint32_t stop_rel = last ? (es.kfix == kEmitAll ? end_rel : es.kfix * cpb) : (k + 1u) * cpb;
    if (!failed) {
        PrefixSink sink;
        sink.column = column;
        sink.rows = rows;
        sink.eoff = off_e + m_start;
        sink.e_hi = H + j0K;
        sink.blk_bits = StreamSink::block_bits(uint32_t(lbl) + n_start);
        sink.bword0 = uint32_t(int32_t(Hb) + lbl);
        sink.blk0 = B;
        sink.total_blocks = total_blocks;
        sink.status = status + im.status_idx;
        sink.over = false;
        if (k == 0) for (uint32_t j = off_e & ~7u; j < off_e; j++) *sink.at(j) = 0u;   // null entries in front of the run's first entry
        uint32_t n = n_start;
        uint32_t matched_word = 0;
        if (p <= end_rel) {
            const uint32_t end_wn = wn_after(end_rel), last_wn = min(wn_after(stop_rel), end_wn);
            uint32_t *my_win = s_win + tid * kMergeStride;
            const LaneBits bits{scan_pool + im.scan_off, s * 16u, im.scan_cols * 16u};
            for (uint32_t q = p / uint32_t(kCpBits);; q++) {                // slices of kCpBits bits, a window of four pieces each (as merge_slice)
                const uint32_t wi1 = (p + 31u) >> 5, wbase = (wi1 ? 4u * wi1 - 4u : 0u) & ~15u;
#pragma unroll
                for (int i = 0; i < kMergeWin / 4; i++) {
                    const uint4 v = bits.piece((wbase >> 4) + i);
                    my_win[4 * i] = __builtin_bswap32(v.x);
                    my_win[4 * i + 1] = __builtin_bswap32(v.y);
                    my_win[4 * i + 2] = __builtin_bswap32(v.z);
                    my_win[4 * i + 3] = __builtin_bswap32(v.w);
                }
                LdsWindow win{reinterpret_cast<const unsigned char *>(my_win), wbase, 0u};
                LaneState st;
                lane_begin(st, win, *h, make_state(p, zc & 0xffu, zc >> 8));
                win.rp = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)(my_win))) + (st.wn - 4u - wbase);
                st.n = n;
                uint32_t blk = n;
                const uint32_t stop_wn = min(wn_after((q + 1u) * uint32_t(kCpBits)), last_wn);
                while (st.wn < stop_wn) (void)symbol_step<true, false>(st, win, lut, *h, blk, sink);
                n = st.n;
                p = lane_pos(st);
                zc = lane_z(st) | (lane_c(st, *h) << 8);
                if (st.wn >= last_wn) {
                    matched_word = cp_state_word(st);
                    break;
                }
            }
        }
        // the item must end where the records say the true path is: at an inner checkpoint the merge rounds' record, at the merge
        // point the first decode's -- state and counts
        if (last && es.kfix == kEmitAll) {
            if (sink.eoff != H + j0K || n != n_p) why |= 16u;
        } else {
            const CpPair want = last ? CpPair{recK, m0r} : cps.get_pair(k);
            if ((want.w & kCpStateMask) != matched_word) why |= 8u;
            if (n != x.n - ((want.w >> 16) & 0x7fffu) || sink.eoff != off_e + (x.m - want.m)) why |= 16u;
        }
        if (sink.over) why |= 32u;
        failed = why != 0;
    }
    if (failed) {
        // no room in front of the first decode's entries (or an inconsistency): the picture is decoded by the two-pass kernels instead
        img_flags[im.status_idx] = 2u;
        if (atomicOr(fallback, why ? why : 128u) == 0u) atomicAdd(unconverged, 1u);      // (see k_huff_scan)
        return;
    }
    if (k != 0) return;
    // (the subsequence's first item also leaves its run word, its label offset, and the verdict on what the first decode could not decode)
    runs[s] = run_word(off_e >> 3, (H + es.d0m + 7u) >> 3, (B - uint32_t(lbl)) & 0xffu);
    g_esub[im.sub_off + s].lbl = lbl;
    if (es.bad != 0xffffffffu && es.kfix != kEmitAll) {
        // an invalid bit pattern the first decode met: it counts if it lies on the part of that decode that stays (at or behind the merge point)
        const uint32_t pK = 8u * (wn_after(es.kfix * cpb) - 8u) - (recK & 31u);
        if (es.bad >= pK && int64_t(B) + int64_t(es.bad_lbl) - lbl < int64_t(total_blocks)) atomicOr(status + im.status_idx, 1);
    }
}

// The block words of every subsequence's run -> DC differences in the picture's block order, and the offsets of the stage-B tiles
// whose first block starts in it.  A workgroup takes kGatherSubs consecutive subsequences: their records are staged in LDS with
// coalesced loads (a wave that fetched its own subsequence's records one dependent load after the other spent its time waiting:
// 1.6 ms per 2048 4K pictures), then every wave walks its share -- one block word per lane and step.
constexpr uint32_t kGatherSubs = 64;
extern "C" __global__ __launch_bounds__(256) void k_block_gather(const DevImage *images, const SubseqState *g_entry, const SubseqState *g_exit,
                                                                  const EmitSub *g_esub, const uint32_t *g_blkbase, uint32_t *entries,
                                                                  int16_t *dcdiff, uint32_t *tile_eoff, const uint32_t *img_flags, int *status)
{
    __shared__ uint32_t s_B[kGatherSubs], s_n[kGatherSubs], s_d0m[kGatherSubs];
    __shared__ int32_t s_lbl[kGatherSubs];
    __shared__ uint8_t s_fs[kGatherSubs + 1], s_exz[kGatherSubs];
    const uint32_t img = picture_of_slot(blockIdx.x, gridDim.x);
    const DevImage &im = images[img];
    const uint32_t s0 = blockIdx.y * kGatherSubs, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (!im.valid || !im.emit || s0 >= im.himg.nsub || img_flags[im.status_idx]) return;
    const uint32_t nsub = im.himg.nsub, total_blocks = im.himg.total_blocks, rows = im.ent_rows, cap = rows * 8u;
    const uint32_t Hb = im.emit_head * 4u, H = im.emit_head * 8u;
    if (tid <= kGatherSubs && s0 + tid < nsub) s_fs[tid] = g_entry[im.sub_off + s0 + tid].z ? 1 : 0;
    if (tid < kGatherSubs && s0 + tid < nsub) {
        const uint32_t s = s0 + tid;
        const SubseqState ex = g_exit[im.sub_off + s];
        const EmitSub es = g_esub[im.sub_off + s];
        const uint32_t B0 = g_blkbase[im.sub_off + s];
        s_B[tid] = B0;
        s_n[tid] = ex.n;
        s_exz[tid] = ex.z ? 1 : 0;
        s_lbl[tid] = es.lbl;
        s_d0m[tid] = es.d0m;
        if (es.kfix == 0) {
            // the lane's entry was right: its run is what k_huff_emit wrote, its labels count from the first block of the subsequence
            // (the others' run words: k_huff_prefix)
            (entries + im.ent_off)[s] = run_word(H >> 3, (H + es.d0m + 7u) >> 3, B0 & 0xffu);
            if (es.bad != 0xffffffffu && B0 + es.bad_lbl < total_blocks) atomicOr(status + im.status_idx, 1);
        }
    }
    __syncthreads();
    int16_t *dc = dcdiff + im.coef_off;
    uint32_t *eoff = tile_eoff + im.tile_off;
    const uint32_t tb = im.tile_blocks, ntiles = (total_blocks + tb - 1) / tb;
    const uint32_t tb_inv = uint32_t(0xffffffffu / tb) + 1u;                 // a / tb == mulhi(a, tb_inv) while a * tb < 2^32
    const bool exact_inv = uint64_t(total_blocks + 1u) * tb < (uint64_t(1) << 32) && tb > 1u;
    constexpr uint32_t per_wave = kGatherSubs / 4;
    // Software pipeline over the wave's subsequences: the block words of subsequence i + 1 are requested before those of
    // subsequence i are stored, so that a wave does not sit out a memory round trip per subsequence.  A lane takes a whole 16-byte
    // group of four block words (a subsequence of the bench holds ~210 blocks = 53 groups: one load instruction per subsequence
    // and wave; longer ones take further rounds in place); words of the group that lie outside the subsequence's own range --
    // in front of its first block, behind its last -- are skipped.
    struct Sub { uint32_t s, B0, i0, i_end, shift; const uint32_t *column; bool live; };
    auto describe = [&](uint32_t i) {
        Sub d;
        const uint32_t t = wave * per_wave + i;
        d.s = s0 + t;
        d.live = i < per_wave && d.s < nsub;
        const uint32_t tt = d.live ? t : 0u;
        d.B0 = s_B[tt];
        const uint32_t B1 = d.B0 + s_n[tt], fs1 = d.s + 1 < nsub ? s_fs[tt + 1] : s_exz[tt];
        const uint32_t a_end = d.live ? min(B1 + fs1, total_blocks + 1u) : 0u, a0 = d.B0 + s_fs[tt];
        d.shift = uint32_t(int32_t(Hb) + s_lbl[tt]);                    // block word index = block - B0 + shift
        d.i0 = a0 - d.B0 + d.shift;
        d.i_end = a_end > a0 ? a_end - d.B0 + d.shift : d.i0;
        d.column = entries + im.ent_off + im.ent_hdr + stream_phys(d.live ? d.s : 0u, 0, rows);
        // the scan ends with the picture's last block: the last tile ends where the last subsequence's run does
        if (d.live && d.s == nsub - 1 && lane == 0 && B1 + fs1 <= total_blocks) eoff[ntiles] = d.s * cap + H + s_d0m[tt];
        return d;
    };
    const uint32_t quads = rows * 2u;
    auto fetch = [&](const Sub &d, uint32_t g) {                          // group g of the column's block words (see BlkRing::group_at)
        uint4 w = make_uint4(0, 0, 0, 0);
        if (4u * g < d.i_end && g < quads) {
            const uint32_t t = quads - 1u - g;
            w = *reinterpret_cast<const uint4 *>(d.column + (t >> 1) * (8u * kStreamQuad) + (t & 1u) * 4u);
        }
        return w;
    };
    auto place = [&](const Sub &d, uint32_t g, const uint4 &w4) {
        const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t i = 4u * g + q;
            if (i < d.i0 || i >= d.i_end) continue;
            const uint32_t a = d.B0 + i - d.shift;
            if (a < total_blocks) dc[a] = int16_t(w[q] & 0xffffu);
            const uint32_t tl = exact_inv ? __umulhi(a, tb_inv) : a / tb;      // (exact while block index x tile size < 2^32)
            if (a == total_blocks) eoff[ntiles] = d.s * cap + (w[q] >> 16);
            else if (tl * tb == a) eoff[tl] = d.s * cap + (w[q] >> 16);
        }
    };
    Sub cur = describe(0);
    uint4 wc = fetch(cur, (cur.i0 >> 2) + lane);
    for (uint32_t i = 0; i < per_wave; i++) {
        if (!cur.live) break;
        const Sub nxt = describe(i + 1);
        const uint4 wn = fetch(nxt, (nxt.i0 >> 2) + lane);
        place(cur, (cur.i0 >> 2) + lane, wc);
        for (uint32_t g = (cur.i0 >> 2) + lane + 64u; 4u * g < cur.i_end; g += 64u) place(cur, g, fetch(cur, g));      // (more than 256 blocks)
        cur = nxt;
        wc = wn;
    }
}

// Multi-scan pictures (beyond the reference, which stops after the first scan).  Every scan went through
// the entropy stage as a one-component picture of its own: stream entries and DC values in the component's raster
// order over its own block grid, one tile offset per block.  The picture's stream in MCU order is gathered from them (pictures of
// keep_coefs batches and of geometries the direct form does not take; the others are read where they lie, DevImage::planar):
//   k_planar_count    one workgroup per tile of the picture: a lane per block slot looks up its block's run in the
//                     component stream (blocks that exist only as MCU padding have none); the tile's total
//   k_planar_offsets  one workgroup per picture: exclusive scan of the totals -> tile offsets
//   k_planar_copy     one workgroup per tile: a workgroup scan places the runs; entries are copied with the block field
//                     rewritten, DC values land in MCU order
// The result is what k_huff_write + DC prediction leave behind for an interleaved picture, so stage B runs unchanged.
struct PlanarSlot { uint32_t s0, cnt, c, rb; bool inside, real; };
__device__ __forceinline__ PlanarSlot planar_slot(const DevImage *images, const DevImage &im, uint32_t img, uint32_t tile,
                                                  const uint32_t *tile_eoff)
{
    const uint32_t T = im.tile_mcus, bpm = im.bpm, tid = threadIdx.x;
    const uint32_t m = tile * T + tid / bpm, k = tid % bpm;
    PlanarSlot p{0, 0, 0, 0, false, false};
    p.inside = tid < im.tile_blocks && m < im.nmcu;
    if (p.inside) {
        p.c = im.blk_comp[k];
        const uint32_t bx = (m % im.mcux) * im.ch[p.c] + im.blk_bx[k], by = (m / im.mcux) * im.cv[p.c] + im.blk_by[k];
        const DevImage &sim = images[img - im.src_back[p.c]];
        if (sim.ncomp == 1) {                           // non-interleaved scan: raster order over the component's own grid
            p.real = bx < im.cbw[p.c] && by < im.cbh[p.c];
            p.rb = by * im.cbw[p.c] + bx;
        } else {                                        // interleaved subset: that scan's own MCU order
            const uint32_t h = im.ch[p.c], v = im.cv[p.c], sm = (by / v) * sim.mcux + bx / h;
            p.real = bx / h < sim.mcux && sm < sim.nmcu;
            p.rb = sm * sim.bpm + sim.cfirst[im.src_comp[p.c]] + (by % v) * h + bx % h;
        }
        if (p.real) {
            const uint32_t *se = tile_eoff + sim.tile_off;                               // one offset per block (+ sentinel)
            p.s0 = se[p.rb];
            p.cnt = se[p.rb + 1] - p.s0;
        }
    }
    return p;
}
__device__ __forceinline__ bool planar_flags(const DevImage *images, const DevImage &im, uint32_t img, const uint32_t *img_flags,
                                             uint32_t &bad)
{
    bad = img_flags[images[img - im.src_back[0]].status_idx] | img_flags[images[img - im.src_back[1]].status_idx] |
          img_flags[images[img - im.src_back[2]].status_idx];
    return bad != 0;
}

extern "C" __global__ __launch_bounds__(256) void k_planar_count(const DevImage *images, uint32_t *tile_eoff,
                                                                  const uint32_t *img_flags)
{
    __shared__ uint32_t s_tmp[4];
    const uint32_t img = blockIdx.y, tile = blockIdx.x;
    const DevImage &im = images[img];
    const uint32_t T = im.tile_mcus;
    uint32_t bad;
    if (!im.valid || im.role != 2 || im.planar || tile >= (im.nmcu + T - 1) / T || planar_flags(images, im, img, img_flags, bad)) return;
    const PlanarSlot p = planar_slot(images, im, img, tile, tile_eoff);
    uint32_t total;
    (void)wg_exclusive_scan(p.cnt, s_tmp, &total);
    if (threadIdx.x == 0) (tile_eoff + im.tile_off)[tile + 1] = total;
}

extern "C" __global__ __launch_bounds__(256) void k_planar_offsets(const DevImage *images, uint32_t *tile_eoff,
                                                                    uint32_t *img_flags)
{
    __shared__ uint32_t s_tmp[4];
    const uint32_t img = blockIdx.x, tid = threadIdx.x;
    const DevImage &im = images[img];
    if (!im.valid || im.role != 2) return;
    uint32_t bad;
    if (planar_flags(images, im, img, img_flags, bad)) {          // a scan is short (or its chunk unconverged): no picture
        if (tid == 0) img_flags[im.status_idx] = bad;
        return;
    }
    if (im.planar) {                                              // read without the gather: stage B only needs the verdict
        if (tid == 0) img_flags[im.status_idx] = 0;
        return;
    }
    const uint32_t T = im.tile_mcus, ntiles = (im.nmcu + T - 1) / T;
    uint32_t *eoff = tile_eoff + im.tile_off;
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < ntiles; t0 += kWgLanes) {           // eoff[t + 1] holds tile t's count -> the offset behind it
        const uint32_t t = t0 + tid;
        const uint32_t v = t < ntiles ? eoff[t + 1] : 0u;
        uint32_t total;
        const uint32_t ex = wg_exclusive_scan(v, s_tmp, &total);
        if (t < ntiles) eoff[t + 1] = run + ex + v;
        run += total;
    }
    if (tid == 0) {
        eoff[0] = 0;
        img_flags[im.status_idx] = 0;
    }
}

extern "C" __global__ __launch_bounds__(256) void k_planar_copy(const DevImage *images, uint32_t *entries,
                                                                 const uint32_t *tile_eoff, int32_t *dcbuf,
                                                                 const uint32_t *img_flags)
{
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_at[257];          // where each block slot's run starts inside the tile's output (+ total)
    __shared__ uint64_t s_src[256];         // ... and where it comes from (index into the entry pool: 64-bit, a chunk's
                                            // stream regions pass 2^32 entries after a few hundred multi-scan 4K pictures)
    const uint32_t img = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const DevImage &im = images[img];
    const uint32_t T = im.tile_mcus;
    if (!im.valid || im.role != 2 || im.planar || tile >= (im.nmcu + T - 1) / T || img_flags[im.status_idx]) return;
    const PlanarSlot p = planar_slot(images, im, img, tile, tile_eoff);
    uint32_t total;
    const uint32_t at = wg_exclusive_scan(p.cnt, s_tmp, &total);
    s_at[tid] = at;
    if (tid == 0) s_at[256] = total;
    if (p.inside) {
        const DevImage &sim = images[img - im.src_back[p.c]];
        (dcbuf + im.coef_off)[tile * im.tile_blocks + tid] = p.real ? (dcbuf + sim.coef_off)[p.rb] : 0;
        s_src[tid] = sim.ent_off + p.s0;
    }
    __syncthreads();
    // the tile's entries, consecutive lanes on consecutive outputs: entry i belongs to the last slot that starts at or
    // before i (binary search over the 256 starts; empty slots share their successor's start and are never chosen)
    uint32_t *dst = entries + im.ent_off + (tile_eoff + im.tile_off)[tile];
    for (uint32_t i = tid; i < total; i += 256) {
        uint32_t lo = 0, hi = 256;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_at[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t e = entries[s_src[lo] + (i - s_at[lo])];
        dst[i] = (e & 0x003fffffu) | (((tile * im.tile_blocks + lo) & 0xffu) << 22);
    }
}

// DC prediction (decoder.rs:173, 208-210: running sum per component, never reset) as a two-level prefix sum over
// `dcd`, which holds per-block differences (16-bit words, see DcRing16) in decode order; the predictions go to dcbuf
// (int32, same indexing), which stage B reads.  An image is cut into segments of kDcSegMcus MCUs:
//   k_dc_sums   one workgroup per segment: per-component sum of the segment's differences -> segsum
//   k_dc_apply  one workgroup per segment: carry-in = sums of the preceding segments; then MCU chunks of 256 with
//               coalesced loads into LDS, one lane per MCU, a workgroup scan, coalesced write-back of absolute DCs.
__device__ __forceinline__ void wg_reduce3(int32_t v[3], int32_t (*s_w)[3], int32_t out[3])
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[c] += __shfl_xor(v[c], d);
    if (lane == 0) { s_w[wave][0] = v[0]; s_w[wave][1] = v[1]; s_w[wave][2] = v[2]; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) out[c] = s_w[0][c] + s_w[1][c] + s_w[2][c] + s_w[3][c];
    __syncthreads();
}

constexpr uint32_t kDcFastShapes = (1u << 1) | (1u << 3) | (1u << 4) | (1u << 6);     // blocks per MCU with a fast kernel
extern "C" __global__ __launch_bounds__(256) void k_dc_sums(const DevImage *images, const int16_t *dcd,
                                                             int32_t *segsum, uint32_t max_segs, const uint32_t *img_flags)
{
    __shared__ int32_t s_w[4][3];
    const DevImage &im = images[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kDcSegMcus;
    if (!im.valid || m0 >= im.nmcu || img_flags[im.status_idx] || ((kDcFastShapes >> im.bpm) & 1u) || im.nseg > 1) return;
    const uint32_t bpm = im.bpm, tid = threadIdx.x;
    const uint32_t nv = (min(uint32_t(kDcSegMcus), im.nmcu - m0)) * bpm;
    const int16_t *dc = dcd + im.coef_off + size_t(m0) * bpm;
    int32_t sum[3] = {0, 0, 0};
    for (uint32_t i = tid; i < nv; i += 256) {
        const int32_t v = dc[i];
        const uint32_t c = im.blk_comp[i % bpm];
        sum[0] += c == 0 ? v : 0;
        sum[1] += c == 1 ? v : 0;
        sum[2] += c == 2 ? v : 0;
    }
    int32_t tot[3];
    wg_reduce3(sum, s_w, tot);
    if (tid < 3) segsum[(size_t(blockIdx.y) * max_segs + blockIdx.x) * 3 + tid] = tot[tid];
}

extern "C" __global__ __launch_bounds__(256) void k_dc_apply(const DevImage *images, const int16_t *dcd, int32_t *dcbuf,
                                                              const int32_t *segsum, uint32_t max_segs,
                                                              const uint32_t *img_flags)
{
    __shared__ int32_t s_dc[256 * kMaxBlocksPerMcu];
    __shared__ int32_t s_wsum[4][3];
    const DevImage &im = images[blockIdx.y];
    const uint32_t seg0 = blockIdx.x * kDcSegMcus;
    if (!im.valid || seg0 >= im.nmcu || img_flags[im.status_idx] || ((kDcFastShapes >> im.bpm) & 1u) || im.nseg > 1) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, bpm = im.bpm;
    const uint32_t seg1 = min(im.nmcu, seg0 + kDcSegMcus);
    int32_t *dc = dcbuf + im.coef_off;
    const int16_t *dd = dcd + im.coef_off;
    int32_t carry[3] = {0, 0, 0};
    for (uint32_t sgi = 0; sgi < blockIdx.x; sgi++) {
        const int32_t *p = segsum + (size_t(blockIdx.y) * max_segs + sgi) * 3;
        carry[0] += p[0]; carry[1] += p[1]; carry[2] += p[2];
    }
    uint32_t comp[kMaxBlocksPerMcu];
#pragma unroll
    for (uint32_t j = 0; j < kMaxBlocksPerMcu; j++) comp[j] = im.blk_comp[j];
    for (uint32_t m0 = seg0; m0 < seg1; m0 += 256) {
        const uint32_t nm = min(256u, seg1 - m0), nv = nm * bpm;
        for (uint32_t i = tid; i < nv; i += 256) s_dc[i] = dd[size_t(m0) * bpm + i];
        __syncthreads();
        int32_t sum[3] = {0, 0, 0};
        if (tid < nm) {
#pragma unroll
            for (uint32_t j = 0; j < kMaxBlocksPerMcu; j++) {
                if (j < bpm) {
                    const int32_t v = s_dc[tid * bpm + j];
                    sum[0] += comp[j] == 0 ? v : 0;
                    sum[1] += comp[j] == 1 ? v : 0;
                    sum[2] += comp[j] == 2 ? v : 0;
                }
            }
        }
        int32_t incl[3] = {sum[0], sum[1], sum[2]};
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int32_t o = __shfl_up(incl[c], d);
                if (lane >= uint32_t(d)) incl[c] += o;
            }
        }
        if (lane == 63) { s_wsum[wave][0] = incl[0]; s_wsum[wave][1] = incl[1]; s_wsum[wave][2] = incl[2]; }
        __syncthreads();
        int32_t base[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            base[c] = carry[c];
            int32_t total = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                const int32_t v = s_wsum[w][c];
                base[c] += w < wave ? v : 0;
                total += v;
            }
            base[c] += incl[c] - sum[c];
            carry[c] += total;
        }
        if (tid < nm) {
#pragma unroll
            for (uint32_t j = 0; j < kMaxBlocksPerMcu; j++) {
                if (j < bpm) {
                    const uint32_t c = comp[j];
                    const int32_t r = (c == 0 ? base[0] : (c == 1 ? base[1] : base[2])) + s_dc[tid * bpm + j];
                    s_dc[tid * bpm + j] = r;
                    base[0] = c == 0 ? r : base[0];
                    base[1] = c == 1 ? r : base[1];
                    base[2] = c == 2 ? r : base[2];
                }
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < nv; i += 256) dc[size_t(m0) * bpm + i] = s_dc[i];
        __syncthreads();
    }
}

// Fast forms for the common MCU shapes (BPM blocks per MCU known at compile time): a lane owns kDcLaneMcus
// consecutive MCUs of the segment, keeps their differences in registers (16-byte loads, all in flight at once),
// and the workgroup needs a single scan.  Images with another MCU shape take the generic kernels above.
// (Staging the segment through LDS for fully coalesced accesses was measured 2.4x slower.)
constexpr int kDcLaneMcus = kDcSegMcus / 256;
struct __attribute__((packed, aligned(4))) Int4 { int32_t a, b, c, d; };

// The lane's differences (16-bit words: 16-byte loads of eight, the lane's run starts on a multiple of eight blocks) into v;
// p = where its predictions go.
template <int BPM>
__device__ __forceinline__ uint32_t dc_lane_load(const DevImage &im, const int16_t *dcd, int32_t *dcbuf, uint32_t seg0, uint32_t seg1,
                                                 int32_t (&v)[kDcLaneMcus * BPM], int32_t *&p)
{
    static_assert(kDcLaneMcus % 8 == 0, "a lane's run is whole 16-byte pieces of halfwords");
    const uint32_t m_first = seg0 + threadIdx.x * kDcLaneMcus;
    const uint32_t nvalid = m_first < seg1 ? min(uint32_t(kDcLaneMcus), seg1 - m_first) : 0u;
    p = dcbuf + im.coef_off + size_t(m_first) * BPM;
    const int16_t *d = dcd + im.coef_off + size_t(m_first) * BPM;
    if (nvalid == kDcLaneMcus) {
#pragma unroll
        for (int q = 0; q < kDcLaneMcus * BPM / 8; q++) {
            const uint4 x = reinterpret_cast<const uint4 *>(d)[q];
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[8 * q + 2 * j] = int32_t(int16_t(w[j] & 0xffffu));
                v[8 * q + 2 * j + 1] = int32_t(w[j]) >> 16;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kDcLaneMcus * BPM; i++) v[i] = uint32_t(i) < nvalid * BPM ? int32_t(d[i]) : 0;
    }
    return nvalid;
}

template <int BPM>
__global__ __launch_bounds__(256) void k_dc_sums_t(const DevImage *images, const int16_t *dcd, int32_t *segsum,
                                                   uint32_t max_segs, const uint32_t *img_flags)
{
    static_assert((kDcLaneMcus * BPM) % 4 == 0, "16-byte pieces");
    __shared__ int32_t s_w[4][3];
    const DevImage &im = images[blockIdx.y];
    const uint32_t seg0 = blockIdx.x * kDcSegMcus;
    if (!im.valid || im.bpm != BPM || seg0 >= im.nmcu || img_flags[im.status_idx] || im.nseg > 1) return;
    const uint32_t seg1 = min(im.nmcu, seg0 + kDcSegMcus);
    int32_t v[kDcLaneMcus * BPM];
    int32_t *p;
    (void)dc_lane_load<BPM>(im, dcd, nullptr, seg0, seg1, v, p);
    int32_t sum[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < BPM; j++) {
        const uint32_t c = im.blk_comp[j];
        int32_t t = 0;
#pragma unroll
        for (int m = 0; m < kDcLaneMcus; m++) t += v[m * BPM + j];
        sum[0] += c == 0 ? t : 0;
        sum[1] += c == 1 ? t : 0;
        sum[2] += c == 2 ? t : 0;
    }
    int32_t tot[3];
    wg_reduce3(sum, s_w, tot);
    if (threadIdx.x < 3) segsum[(size_t(blockIdx.y) * max_segs + blockIdx.x) * 3 + threadIdx.x] = tot[threadIdx.x];
}

template <int BPM>
__global__ __launch_bounds__(256) void k_dc_apply_t(const DevImage *images, const int16_t *dcd, int32_t *dcbuf, const int32_t *segsum,
                                                    uint32_t max_segs, const uint32_t *img_flags)
{
    __shared__ int32_t s_wsum[4][3];
    const DevImage &im = images[blockIdx.y];
    const uint32_t seg0 = blockIdx.x * kDcSegMcus;
    if (!im.valid || im.bpm != BPM || seg0 >= im.nmcu || img_flags[im.status_idx] || im.nseg > 1) return;
    const uint32_t seg1 = min(im.nmcu, seg0 + kDcSegMcus);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t v[kDcLaneMcus * BPM];
    int32_t *p;
    const uint32_t nvalid = dc_lane_load<BPM>(im, dcd, dcbuf, seg0, seg1, v, p);
    int32_t carry[3] = {0, 0, 0};
    for (uint32_t sgi = 0; sgi < blockIdx.x; sgi++) {
        const int32_t *q = segsum + (size_t(blockIdx.y) * max_segs + sgi) * 3;
        carry[0] += q[0]; carry[1] += q[1]; carry[2] += q[2];
    }
    uint32_t comp[BPM];
#pragma unroll
    for (int j = 0; j < BPM; j++) comp[j] = im.blk_comp[j];
    int32_t sum[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < BPM; j++) {
        int32_t t = 0;
#pragma unroll
        for (int m = 0; m < kDcLaneMcus; m++) t += v[m * BPM + j];
        sum[0] += comp[j] == 0 ? t : 0;
        sum[1] += comp[j] == 1 ? t : 0;
        sum[2] += comp[j] == 2 ? t : 0;
    }
    int32_t incl[3] = {sum[0], sum[1], sum[2]};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int32_t o = __shfl_up(incl[c], d);
            if (lane >= uint32_t(d)) incl[c] += o;
        }
    }
    if (lane == 63) { s_wsum[wave][0] = incl[0]; s_wsum[wave][1] = incl[1]; s_wsum[wave][2] = incl[2]; }
    __syncthreads();
    int32_t base[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        base[c] = carry[c] + incl[c] - sum[c];
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) base[c] += w < wave ? s_wsum[w][c] : 0;
    }
#pragma unroll
    for (int m = 0; m < kDcLaneMcus; m++)
#pragma unroll
        for (int j = 0; j < BPM; j++) {
            const uint32_t c = comp[j];
            const int32_t r = (c == 0 ? base[0] : (c == 1 ? base[1] : base[2])) + v[m * BPM + j];
            v[m * BPM + j] = r;
            base[0] = c == 0 ? r : base[0];
            base[1] = c == 1 ? r : base[1];
            base[2] = c == 2 ? r : base[2];
        }
    if (nvalid == kDcLaneMcus) {
#pragma unroll
        for (int q = 0; q < kDcLaneMcus * BPM / 4; q++)
            reinterpret_cast<Int4 *>(p)[q] = Int4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    } else {
#pragma unroll
        for (int i = 0; i < kDcLaneMcus * BPM; i++)
            if (uint32_t(i) < nvalid * BPM) p[i] = v[i];
    }
}

// One pass instead of k_dc_sums_t + k_dc_apply_t (the differences are read once, not twice: 1.6 instead of 2.4 GB of traffic
// per 1024 4K pictures): the workgroup of segment x keeps its differences in registers, publishes the running sums up to and
// including its segment -- what segment x - 1 published plus its own totals -- and then writes the predictions.  Workgroups
// are dispatched in the order of their linear id, so the workgroup a segment waits for has always been started before it
// and waits only for still earlier ones: no residency condition, unlike k_huff_merge_loop.  The image is the fast grid
// dimension: segment x of all images, then segment x + 1 of all images -- with the segment as the fast dimension the
// sixteen workgroups of a picture start together and wait for one another in a chain (5.2 instead of 1.0 ms per step).
//   segflag[(image, segment)][component] = gen << 32 | running sum: valid once its upper half equals `gen` (the words are zero
//   when the batch is created and `gen` grows with every launch on this set of buffers).
// Nothing promises the dispatch order, so the wait is bounded: after `spin_limit` polls a workgroup gives up -- it sets *fail,
// publishes a poisoned word (upper half gen | kDcPoison) so that the segments behind it give up at once instead of waiting
// their own limit, and leaves without writing.  The host sees *fail in mjx_batch_wait and decodes the chunk again with the
// two-pass kernels (k_dc_sums_t / k_dc_apply_t), which wait for nobody.  `fault` (test knob): segment 0 of the chunk's first
// image never publishes.
constexpr uint32_t kDcPoison = 0x80000000u;
template <int BPM>
__global__ __launch_bounds__(256) void k_dc_scan_t(const DevImage *images, const int16_t *dcd, int32_t *dcbuf, uint32_t *segsum,
                                                   uint32_t max_segs, const uint32_t *img_flags, uint32_t gen, uint32_t *fail,
                                                   uint32_t spin_limit, uint32_t fault)
{
    __shared__ int32_t s_wsum[4][3];
    __shared__ int32_t s_carry[3];
    __shared__ uint32_t s_gave_up;
    if (threadIdx.x == 0) s_gave_up = 0;
    const uint32_t img = picture_of_slot(blockIdx.x, gridDim.x), seg = blockIdx.y;         // (the image is the fast dimension, see above)
    const DevImage &im = images[img];
    const uint32_t seg0 = seg * kDcSegMcus;
    if (!im.valid || im.bpm != BPM || seg0 >= im.nmcu || img_flags[im.status_idx] || im.nseg > 1) return;
    const uint32_t seg1 = min(im.nmcu, seg0 + kDcSegMcus);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t v[kDcLaneMcus * BPM];
    int32_t *p;
    const uint32_t nvalid = dc_lane_load<BPM>(im, dcd, dcbuf, seg0, seg1, v, p);
    uint32_t comp[BPM];
#pragma unroll
    for (int j = 0; j < BPM; j++) comp[j] = im.blk_comp[j];
    int32_t sum[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < BPM; j++) {
        int32_t t = 0;
#pragma unroll
        for (int m = 0; m < kDcLaneMcus; m++) t += v[m * BPM + j];
        sum[0] += comp[j] == 0 ? t : 0;
        sum[1] += comp[j] == 1 ? t : 0;
        sum[2] += comp[j] == 2 ? t : 0;
    }
    int32_t incl[3] = {sum[0], sum[1], sum[2]};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int32_t o = __shfl_up(incl[c], d);
            if (lane >= uint32_t(d)) incl[c] += o;
        }
    }
    if (lane == 63) { s_wsum[wave][0] = incl[0]; s_wsum[wave][1] = incl[1]; s_wsum[wave][2] = incl[2]; }
    __syncthreads();
    if (threadIdx.x < 3) {                                      // carry in from the segment before, carry out to the one behind
        // One 64-bit word per component, `gen` in its upper half: a word validates itself, so plain device-scope atomics do
        // (performed at the memory side, coherent across the XCDs) -- release / acquire fences write back and invalidate the
        // whole L2 of the XCD, 16 384 times per launch: 3.6 instead of 1.0 ms per step.
        const uint32_t c = threadIdx.x;
        unsigned long long *words = reinterpret_cast<unsigned long long *>(segsum);
        const size_t at = (size_t(img) * max_segs + seg) * 3 + c;
        int32_t carry = 0;
        bool gave_up = false;
        if (seg > 0) {
            unsigned long long w;
            uint32_t polls = 0;
            while (uint32_t((w = atomicAdd(words + at - 3, 0ull)) >> 32) != gen) {
                if (uint32_t(w >> 32) == (gen | kDcPoison) || ++polls >= spin_limit) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            carry = int32_t(uint32_t(w));
        }
        s_carry[c] = carry;
        const int32_t out = carry + s_wsum[0][c] + s_wsum[1][c] + s_wsum[2][c] + s_wsum[3][c];
        if (gave_up) {
            s_gave_up = 1;
            atomicOr(fail, 1u);
            (void)atomicExch(words + at, static_cast<unsigned long long>(gen | kDcPoison) << 32);
        } else if (!(fault && img == 0 && seg == 0)) {
            (void)atomicExch(words + at, (static_cast<unsigned long long>(gen) << 32) | uint32_t(out));
        }
    }
    __syncthreads();
    if (s_gave_up) return;                                      // (the segment keeps its differences; the chunk is decoded again)
    int32_t base[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        base[c] = s_carry[c] + incl[c] - sum[c];
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) base[c] += w < wave ? s_wsum[w][c] : 0;
    }
#pragma unroll
    for (int m = 0; m < kDcLaneMcus; m++)
#pragma unroll
        for (int j = 0; j < BPM; j++) {
            const uint32_t c = comp[j];
            const int32_t r = (c == 0 ? base[0] : (c == 1 ? base[1] : base[2])) + v[m * BPM + j];
            v[m * BPM + j] = r;
            base[0] = c == 0 ? r : base[0];
            base[1] = c == 1 ? r : base[1];
            base[2] = c == 2 ? r : base[2];
        }
    if (nvalid == kDcLaneMcus) {
#pragma unroll
        for (int q = 0; q < kDcLaneMcus * BPM / 4; q++)
            reinterpret_cast<Int4 *>(p)[q] = Int4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    } else {
#pragma unroll
        for (int i = 0; i < kDcLaneMcus * BPM; i++)
            if (uint32_t(i) < nvalid * BPM) p[i] = v[i];
    }
}

// Restart intervals: the DC predictors start again at 0 in every interval (T.81 E.2.4), so the
// prediction is independent per interval: one lane walks the blocks of one interval.
extern "C" __global__ __launch_bounds__(256) void k_dc_restart(const DevImage *images, const int16_t *dcd, int32_t *dcbuf,
                                                                const uint32_t *img_flags)
{
    const DevImage &im = images[blockIdx.y];
    if (!im.valid || im.nseg <= 1 || img_flags[im.status_idx]) return;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= im.nseg) return;
    const uint32_t seg_blocks = im.restart_mcus * im.bpm, b0 = g * seg_blocks;
    const uint32_t b1 = min(b0 + seg_blocks, im.himg.total_blocks);
    int32_t *dc = dcbuf + im.coef_off;
    const int16_t *dd = dcd + im.coef_off;
    int32_t p0 = 0, p1 = 0, p2 = 0;
    uint32_t j = 0;
    for (uint32_t b = b0; b < b1; b += 8) {                                // eight loads in flight, then the serial part
        int32_t v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) v[q] = b + q < b1 ? int32_t(dd[b + q]) : 0;
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t c = im.blk_comp[j];
            const int32_t r = (c == 0 ? p0 : (c == 1 ? p1 : p2)) + v[q];
            if (b + q < b1) dc[b + q] = r;
            p0 = c == 0 ? r : p0;
            p1 = c == 1 ? r : p1;
            p2 = c == 2 ? r : p2;
            j = j + 1 == im.bpm ? 0 : j + 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stage B
// ------------------------------------------------------------------------------------------------
constexpr int kPixStride = 68;       // floats per IDCT output block (64 + 4)

// 8-point inverse DCT, Arai-Agui-Nakajima factorisation on inputs pre-scaled by the AAN factors (folded into the
// dequantisation multipliers on the host, mjx_plan.cpp).  5 multiplies, 29 additions.  Same transform as the
// reference's direct-form DCT-III (transform.rs:55-87) up to float rounding.
// (T = float, or a pair of floats: two transforms side by side in packed instructions, see idct_row_inplace)
typedef float float_pair __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ void idct8(T &i0, T &i1, T &i2, T &i3, T &i4, T &i5, T &i6, T &i7)
{
    const T t10 = i0 + i4, t11 = i0 - i4;
    const T t13 = i2 + i6, t12 = (i2 - i6) * 1.414213562f - t13;
    const T e0 = t10 + t13, e3 = t10 - t13, e1 = t11 + t12, e2 = t11 - t12;
    const T z13 = i5 + i3, z10 = i5 - i3, z11 = i1 + i7, z12 = i1 - i7;
    const T o7 = z11 + z13;
    const T u11 = (z11 - z13) * 1.414213562f;
    const T z5 = (z10 + z12) * 1.847759065f;
    const T u10 = 1.082392200f * z12 - z5;
    const T u12 = -2.613125930f * z10 + z5;
    const T o6 = u12 - o7, o5 = u11 - o6, o4 = u10 + o5;
    i0 = e0 + o7; i7 = e0 - o7;
    i1 = e1 + o6; i6 = e1 - o6;
    i2 = e2 + o5; i5 = e2 - o5;
    i4 = e3 + o4; i3 = e3 - o4;
}

// decoder.rs:382-390 f32_to_u8 (clamp to [0,255], truncate toward zero) fused with the byte packing:
// floor() then v_cvt_pk_u8_f32, which saturates to [0,255] and inserts the byte (it rounds to nearest, hence the
// floor; negative inputs saturate to 0 either way).  Checked against clamp+truncate in tools/probes/cvt_probe.hip.
__device__ __forceinline__ uint32_t pack_u8(float n, uint32_t byte, uint32_t word)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(n), byte, word);
}

// decoder.rs:392-402 y_cb_cr_to_rgb:  r = cr (2 - 2 c_red) + y,  b = cb (2 - 2 c_blue) + y,
// g = (y - c_blue b - c_red r) / c_green, then + 128, clamp, truncate.  Substituting r and b, g is y minus fixed
// multiples of cb and cr (c_red + c_green + c_blue = 1), so a chroma sample contributes three terms that are shared by
// every pixel it covers (four in 4:2:0), and a pixel costs three additions.  The + 128 is not added here: the
// luminance samples arrive with it (it is put on the DC coefficient before the inverse DCT, k_idct_color phase 1).
// The reference rounds its longer chain of f32 operations differently; the results agree within the 1-LSB bound of
// the parity tests (a sample must lie within ~1e-4 of an integer for the truncation to differ).
struct Rgb { float r, g, b; };
struct ChromaTerms { float r, g, b; };
__device__ __forceinline__ ChromaTerms chroma_terms(float cb, float cr)
{
    const float c_red = 0.299f, c_green = 0.587f, c_blue = 0.114f;
    const float kr = 2.0f - 2.0f * c_red, kb = 2.0f - 2.0f * c_blue;
    ChromaTerms t;
    t.r = cr * kr;
    t.b = cb * kb;
    t.g = __builtin_fmaf(cb, -(c_blue * kb / c_green), cr * -(c_red * kr / c_green));
    return t;
}
__device__ __forceinline__ Rgb ycc_to_rgb(float y128, const ChromaTerms &t)
{
    return Rgb{y128 + t.r, y128 + t.g, y128 + t.b};
}

struct __attribute__((packed, aligned(4))) Rgb4 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(1))) Rgb4u { uint32_t a, b, c; };

// 4 pixels -> 12 bytes R,G,B,R,G,B,...   decoder.rs:382-390 f32_to_u8 = clamp to [0,255] + truncate.  v_cvt_pk_u8_f32
// saturates and inserts the byte; it rounds in the current f32 rounding mode, so the twelve conversions run with
// MODE.FP_ROUND = toward-zero (checked against clamp+truncate on the GPU: tools/probes/cvt_rtz_probe.hip) and the mode
// is back to nearest-even before the block ends.  One opaque asm block: no float arithmetic can move inside it.
__device__ __forceinline__ Rgb4 pack4(const Rgb p[4])
{
    Rgb4 o;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "s_nop 1\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %11, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %12, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %13, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
                 "s_nop 1"
                 : "=&v"(o.a), "=&v"(o.b), "=&v"(o.c)
                 : "v"(p[0].r), "v"(p[0].g), "v"(p[0].b), "v"(p[1].r),      // dword a: R0 G0 B0 R1
                   "v"(p[1].g), "v"(p[1].b), "v"(p[2].r), "v"(p[2].g),      // dword b: G1 B1 R2 G2
                   "v"(p[2].b), "v"(p[3].r), "v"(p[3].g), "v"(p[3].b));     // dword c: B2 R3 G3 B3
    return o;
}

// The pictures leave with non-temporal stores (round 3): 25 MB per picture that nothing reads again soon would push the
// write pass's half-filled lines out of L2 when the two kernels share the device (30.1 -> 29.65 ms per step with the default
// streams; 16.1 -> 16.0 ms for stage B alone).  The compiler fuses the three dword stores into one global_store_dwordx3 nt.
__device__ __forceinline__ void store_rgb4(uint8_t *dst, const Rgb4 &v)
{
#if defined(MJX_EXP_NO_RGB_STORE)      // (measurement build: stage B without its picture stores -- the in-CU work alone)
    asm volatile("" :: "v"(v.a), "v"(v.b), "v"(v.c), "v"(dst));
    return;
#endif
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    __builtin_nontemporal_store(v.a, d);
    __builtin_nontemporal_store(v.b, d + 1);
    __builtin_nontemporal_store(v.c, d + 2);
}

__device__ __forceinline__ void store4(uint8_t *dst, const Rgb4 &v, bool aligned, uint32_t npix)
{
    if (npix >= 4) {
        if (aligned) store_rgb4(dst, v);
        else *reinterpret_cast<Rgb4u *>(dst) = Rgb4u{v.a, v.b, v.c};
    } else {
        const uint32_t w[3] = {v.a, v.b, v.c};
        for (uint32_t k = 0; k < npix * 3; k++) dst[k] = uint8_t(w[k >> 2] >> ((k & 3) * 8));
    }
}

// ---- stage B pipeline pieces ---------------------------------------------------------------------
// A workgroup walks kTilesPerWg consecutive tiles of one image.  While it transforms tile t it already holds the loads
// of tile t+1 in flight (stream offsets, up to kPrefetch entries per lane, the lane's DC), so the HBM round trips of a
// tile overlap the arithmetic of the previous one instead of sitting on the workgroup's critical path.
#ifndef MJX_PREFETCH
#define MJX_PREFETCH 8
#endif
#ifndef MJX_PREFETCH_DENSE
#define MJX_PREFETCH_DENSE 12
#endif
#ifndef MJX_TILES_PER_WG
#define MJX_TILES_PER_WG 16
#endif
constexpr int kPrefetch = MJX_PREFETCH;         // stream entries per lane held in registers (2048 per tile; the rest is read when the tile is scattered)
constexpr int kPrefetchDense = MJX_PREFETCH_DENSE;   // ... in the 4:2:0 kernel's form for dense streams (3072 per tile: quality 90 and up), chosen per chunk by the host
static_assert(MJX_PREFETCH == 8 && MJX_PREFETCH_DENSE > 8, "scatter batches of 4, 6, 8 and the full depth");
constexpr int kTilesPerWg = MJX_TILES_PER_WG;
// The 4:2:0 kernel has two forms.  MJX_WIDE420 = 1 (round 6, MODE 3 below): 16 lanes per MCU, tiles of 16 MCUs = 25 KB of LDS, the
// inverse DCT split into a column pass (two lanes per block) and a row pass fused with the colour step (a lane per 8 x 2 pixels) --
// twice the waves per LDS byte of the other form.  MJX_WIDE420 = 0: 8 lanes per MCU, tiles of 32 MCUs = 52 KB, a lane per block
// for the whole transform (rounds 1-5, MODE 1).
#ifndef MJX_WIDE420
#define MJX_WIDE420 0
#endif
#ifndef MJX_TILE420
#define MJX_TILE420 (MJX_WIDE420 ? 16 : 32)
#endif
constexpr uint32_t kTile420 = MJX_TILE420;        // MCUs per tile of the 4:2:0 kernel
constexpr uint32_t kLanes420 = kTile420 * (MJX_WIDE420 ? 16 : 8);
constexpr int kMode420 = MJX_WIDE420 ? 3 : 1;     // the kernel form (template MODE) that takes the pictures of DevImage::mode 1
static_assert(!MJX_WIDE420 || kTile420 == 16, "the wide form's lane maps are written for 256 lanes = 16 MCUs");
// MODE 3: an MCU's six blocks lie in three SUPER-ROWS of kWRow floats -- (Y00, Y01), (Y10, Y11), (Cb, Cr) --, 128 values + 4 of padding
// (an odd number of 16-byte slots: the 16 lanes of a ds_read_b128 group, one MCU apart, fall into 16 different slots).  Inside a
// super-row, lane q (0..3) of the column pass owns the eight floats  m * 32 + 8 q .. + 7  of every row pair m (0..3):
//   coefficients   Y, block side sd = q >> 1, columns c = 4 (q & 1) .. + 3:   (r >> 1) * 32 + 8 q + (r & 1) * 4 + (c & 3)
//                  chroma, columns c = 2 q, 2 q + 1, both components:          (r >> 1) * 32 + 8 q + (r & 1) * 4 + (c & 1) * 2 + comp
//   after the column pass (written back in place)
//                  Y:       (r >> 1) * 32 + sd * 16 + c * 2 + (r & 1)      -- (row 2m, row 2m+1) of a column side by side: the row pass's pairs
//                  chroma:  unchanged                                       -- (Cb, Cr) of a column side by side
constexpr uint32_t kWRow = 132, kWMcu = 3 * kWRow;
constexpr uint32_t wide_tile_bytes() { return kTile420 * kWMcu * 4; }

template <int PF>
struct TileFetch {
    uint32_t e0, e1;                 // the tile's slice of the compact stream
    uint32_t ent[PF];
    int32_t dc;
};

template <uint32_t LANES, int PF>
__device__ __forceinline__ void tile_fetch(const uint32_t *__restrict__ src, const uint32_t *__restrict__ eoff,
                                           const int32_t *__restrict__ dc, uint32_t tile, uint32_t tile_blocks,
                                           uint32_t total_blocks, TileFetch<PF> &f)
{
    const uint32_t tid = threadIdx.x;
    f.e0 = eoff[0];                  // (eoff: the workgroup's copy of its tiles' offsets in LDS, see k_idct_color)
    f.e1 = eoff[1];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        const uint32_t i = f.e0 + tid + LANES * k;
        f.ent[k] = i < f.e1 ? __builtin_nontemporal_load(src + i) : 0u;      // (read once: streamed past L2, like the pictures on their way out)
    }
    const uint32_t blk = tile * tile_blocks + tid;
    f.dc = (tid < tile_blocks && blk < total_blocks) ? __builtin_nontemporal_load(dc + blk) : 0;
}

// A tile of the quad-interleaved stream (mjx_kernels.h: stream_phys, quad_prepare, quad_cell): lane i of the workgroup takes
// the tile's i-th store group (+ 256 per further round) -- all eight entries of the 32-byte group, two 16-byte loads -- and
// masks what belongs to the neighbouring tiles (k_lo / k_hi).  QuadFetch<R>: R rounds are prefetched; a tile of the bench
// content has ~190 groups, at quality 90 ~380.  (Half a group per lane, so that a wave's load touches 32 rows instead of 64 twice
// over: 16.1 instead of 15.8 ms per 2048 pictures in stage B.  The same groups read from consecutive addresses --
// -DMJX_EXP_QUAD_CONTIG2, garbage out -- 14.9: the spread over rows shared with the neighbours costs 0.9 ms)
template <int R>
struct QuadFetch {
    uint32_t ncells;                                    // the tile's groups
    uint32_t ent[R][8];
    uint32_t k_lo[R], k_hi[R];
    uint32_t lab[R];                                    // label offset of the group's subsequence (run_label)
    int32_t dc;
};
// group `o` of the tile: loads it and says which of its entries are the tile's; returns the tile's groups
__device__ __forceinline__ uint32_t quad_load(const uint32_t *__restrict__ src, const QuadView &q, uint32_t k, uint32_t o,
                                              uint32_t *ent, uint32_t &k_lo, uint32_t &k_hi, uint32_t &lab)
{
    QuadCell cell;
    const uint32_t total = quad_cell(q, k, o, cell);
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (cell.phys != 0xffffffffu) {
        const uint4 *p = reinterpret_cast<const uint4 *>(src + cell.phys);
        a = p[0];
        b = p[1];
    }
    ent[0] = a.x; ent[1] = a.y; ent[2] = a.z; ent[3] = a.w;
    ent[4] = b.x; ent[5] = b.y; ent[6] = b.z; ent[7] = b.w;
    k_lo = cell.k_lo;
    k_hi = cell.k_hi;
    lab = cell.label;
    return total;
}
template <uint32_t LANES, int R>
__device__ __forceinline__ void tile_fetch_quad(const uint32_t *__restrict__ src, const QuadView &q, uint32_t k,
                                                const int32_t *__restrict__ dc, uint32_t tile, uint32_t tile_blocks,
                                                uint32_t total_blocks, QuadFetch<R> &f)
{
    static_assert(kAcGroup == 8, "the stream's store groups are 32 bytes");
    const uint32_t tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; r++) f.ncells = quad_load(src, q, k, tid + LANES * r, f.ent[r], f.k_lo[r], f.k_hi[r], f.lab[r]);
    const uint32_t blk = tile * tile_blocks + tid;
    f.dc = (tid < tile_blocks && blk < total_blocks) ? __builtin_nontemporal_load(dc + blk) : 0;
}
// (what pins a tile's prefetched words: an empty asm that "uses" them, so the wait for their loads is placed there)
template <int PF>
__device__ __forceinline__ void settle(TileFetch<PF> &f)
{
#pragma unroll
    for (int k = 0; k < PF; k++) asm volatile("" : "+v"(f.ent[k]));
    asm volatile("" : "+v"(f.dc), "+v"(f.e0), "+v"(f.e1));
}
template <int R>
__device__ __forceinline__ void settle(QuadFetch<R> &f)
{
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("" : "+v"(f.ent[r][k]));
        asm volatile("" : "+v"(f.k_lo[r]), "+v"(f.k_hi[r]), "+v"(f.lab[r]));
    }
    asm volatile("" : "+v"(f.dc), "+v"(f.ncells));
}
__device__ __forceinline__ void quad_mask(uint32_t *ent, uint32_t k_lo, uint32_t k_hi)
{
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) ent[k] = (k >= k_lo && k < k_hi) ? ent[k] : 0u;
}

// One stream entry -> one float in the tile: find the block from the entry's block byte, multiply by the
// dequantisation x IDCT-prescale factor of its zig-zag position, store at the natural-order position
// (un-zigzag, decoder.rs:230-232, 425-437).
//
// The tile's entries are scattered in batches of kPrefetch per lane, without a branch (round 4).  Entry by entry behind its own
// `if (valid)`, each one paid two LDS round trips of its own -- the block's component and the position's natural index, then
// the multiplier -- before its store: sixteen exposed round trips per lane and tile.  Now the component of a 4:2:0 block comes
// from arithmetic on its index (other layouts: one byte read, batched as well), so the multiplier's and the natural position's
// addresses are known at once, all reads of a batch are in flight together, and an entry that must not land (a null entry --
// position 0 --, a block of another tile) is sent to a word of row padding instead of being branched around.  (The tables stay
// two small arrays: a single array of {multiplier, offset} pairs would add 0.7 KB of static LDS, and with the allocation
// granule that is the third workgroup per CU -- measured: 17.6 instead of 15.5 ms per 2048 pictures.)
// where a lane sends the entries that must not land: a padding word (floats 64..67 of a block row are never read) of a row of
// its own -- rows 0..127 exist in every tile allocation (tile_mcus), and the 32 lanes of an LDS store group hit 32 banks
template <int MODE>
__device__ __forceinline__ uint32_t dump_bytes()
{
    // (MODE 3: the four padding words of super-rows 0..7 -- 32 lanes, 32 banks)
    if (MODE == 3) return (threadIdx.x & 7u) * (kWRow * 4u) + (128u + ((threadIdx.x >> 3) & 3u)) * 4u;
    return (threadIdx.x & 127u) * uint32_t(kPixStride * 4) + (64u + ((threadIdx.x >> 3) & 3u)) * 4u;
}
// MODE 3: where block b of the tile (MCU order, Y Y Y Y Cb Cr) keeps its coefficients -- bytes into the tile, to which the position's
// table entry is added (s_nat: 64 entries for luminance, 64 for chrominance)
__device__ __forceinline__ uint32_t wide_block_bytes(uint32_t b)
{
    const uint32_t t = (b * 171u) >> 10, j = b - 6u * t;
    const uint32_t in_mcu = j < 4u ? (j >> 1) * (kWRow * 4u) + (j & 1u) * 64u : 2u * kWRow * 4u + (j & 1u) * 4u;
    return t * (kWMcu * 4u) + in_mcu;
}

template <int MODE>
__device__ __forceinline__ uint32_t comp_of_block(uint32_t b, const uint8_t *s_comp)
{
    if (MODE == 1 || MODE == 3) {                           // Y Y Y Y Cb Cr: block b of the tile, b < 256
        const uint32_t j = b - 6u * ((b * 171u) >> 10);     // (171 / 1024: exact quotient by 6 below 512)
        return j < 4u ? 0u : j - 3u;
    }
    return s_comp[b];
}

template <int MODE, int N>
__device__ __forceinline__ void scatter_at(const uint32_t *ent, const uint32_t *b, uint32_t nblk, float *tile_f,
                                           const float *s_qm, const uint8_t *s_nat, const uint8_t *s_comp);
template <int MODE, int N>
__device__ __forceinline__ void scatter_batch(const uint32_t *ent, uint32_t first_lo, uint32_t nblk, float *tile_f,
                                              const float *s_qm, const uint8_t *s_nat, const uint8_t *s_comp)
{
    uint32_t b[N];
#pragma unroll
    for (int k = 0; k < N; k++) b[k] = ((ent[k] >> 22) - first_lo) & 0xffu;
    scatter_at<MODE, N>(ent, b, nblk, tile_f, s_qm, s_nat, s_comp);
}
// (b[k]: the block slot of entry k in the tile; anything >= nblk is dropped)
template <int MODE, int N>
__device__ __forceinline__ void scatter_at(const uint32_t *ent, const uint32_t *b, uint32_t nblk, float *tile_f,
                                           const float *s_qm, const uint8_t *s_nat, const uint8_t *s_comp)
{
    uint32_t pos[N], comp[N], nat[N];
    float qm[N];
#pragma unroll
    for (int k = 0; k < N; k++) pos[k] = (ent[k] >> 16) & 63u;
#pragma unroll
    for (int k = 0; k < N; k++) comp[k] = comp_of_block<MODE>((MODE == 1 || MODE == 3) ? b[k] : (b[k] < nblk ? b[k] : 0u), s_comp);
#pragma unroll
    for (int k = 0; k < N; k++) {
        qm[k] = s_qm[comp[k] * 64u + pos[k]];
        nat[k] = s_nat[MODE == 3 ? pos[k] + (comp[k] ? 64u : 0u) : pos[k]];
    }
    unsigned char *base = reinterpret_cast<unsigned char *>(tile_f);
    const uint32_t dump = dump_bytes<MODE>();
#pragma unroll
    for (int k = 0; k < N; k++) {
        const bool ok = b[k] < nblk && pos[k] != 0;           // pos == 0 marks a null entry (the write pass fills up its runs with them)
        const uint32_t at = ok ? (MODE == 3 ? wide_block_bytes(b[k]) : b[k] * uint32_t(kPixStride * 4)) + nat[k] * 4u : dump;
#if defined(MJX_EXP_NO_SCATTER_STORE)  // (measurement build, garbage out: the scatter phase without its LDS stores)
        asm volatile("" :: "v"(at), "v"(float(int32_t(int16_t(ent[k] & 0xffffu))) * qm[k]));
#else
        *reinterpret_cast<float *>(base + at) = float(int32_t(int16_t(ent[k] & 0xffffu))) * qm[k];
#endif
    }
}

// ---- multi-scan pictures read without the gather (round 5; DevImage::planar, mjx_kernels.h) ---------------------------------------
// A tile's entries lie in up to kPlanarSegs segments of the scans' linear streams (pieces x kinds).  A ROUND of the scatter is one
// segment's next 256 entries, one per lane, so which segment a lane's entry belongs to -- hence how its block label maps into the
// tile -- is uniform over the workgroup: one packed word per round.  The workgroup keeps three tiles' worth of records in LDS -- the
// tile being scattered, the one whose entries are being fetched, the one being prepared --: eight lanes (one per segment) look a
// tile's segments up in the scans' tables two tiles ahead (the two loads are issued at the top of a tile iteration and used at its
// settle point, a whole inverse DCT later) and list its rounds.
struct alignas(4) PlanarKindL {      // per kind of segment, prepared once per workgroup
    uint32_t ent_rel;                // the scan's stream region, relative to the file's first scan's
    uint32_t tbl_off;                // its segment table in tile_eoff
    uint32_t mcux, mcuy;             // the scan's MCU grid
    uint16_t S;
    uint8_t bpm, hs, vs, v, log2u, mapbase;
};
struct alignas(4) PlanarRound {
    uint32_t src;                    // first entry of the round, relative to the file's first scan's region
    uint32_t how;                    // low byte of the scan's index of the segment's first block [7:0] | block slot in the tile of the piece's first
                                     // MCU [15:8] | slot inside the MCU of the segment's first block per MCU [23:16] | log2(blocks per MCU) [31:24]
    uint32_t cnt;                    // entries of the round (<= 256; 0: no such round)
};
struct alignas(4) PlanarTile {
    uint32_t start[kPlanarSegs];     // per segment: entries into the scan's region,
    uint16_t len[kPlanarSegs];       // ... how many,
    uint32_t how[kPlanarSegs];       // ... PlanarRound::how
    PlanarRound rnd[8];              // the prefetched rounds
    uint8_t rest_seg, rest_rnd, pad_[2];     // where the rounds beyond them begin (rest_seg == kPlanarSegs: nowhere)
};
struct PlanarProbe { uint32_t ia, ie, how; bool any; };
// Segment j of the tile whose first MCU is m0 = row r0, column a0 of the picture: where its two table words are, and how its blocks
// map into the tile.  (LOG2T: the tile is a power of two MCUs.)
__device__ __forceinline__ PlanarProbe planar_probe(const PlanarKindL *kinds, uint32_t nk, uint32_t j, uint32_t m0, uint32_t r0, uint32_t a0,
                                                    uint32_t log2T, uint32_t nmcu, uint32_t mcux, uint32_t bpm)
{
    PlanarProbe p{0, 0, 0, false};
    const uint32_t piece = j >= nk ? 1u : 0u, kind = piece ? j - nk : j;       // (two pieces at most: the host sends other pictures through the gather)
    if (m0 >= nmcu || kind >= nk) return p;
    const uint32_t T = 1u << log2T, nm = min(T, nmcu - m0);
    const uint32_t r = r0 + piece, a = piece ? 0u : a0, ms = r * mcux + a;      // the piece's first MCU
    if (ms >= m0 + nm) return p;
    const uint32_t b = min(mcux, a + (m0 + nm - ms));
    const PlanarKindL &K = kinds[kind];
    const uint32_t Rs = r * K.vs + K.v, Ca = a * K.hs;
    if (Rs >= K.mcuy || Ca >= K.mcux) return p;
    const uint32_t ia = Rs * K.S + ((ms >> log2T) - ((r * mcux) >> log2T));
    p.ia = K.tbl_off + ia;
    p.ie = K.tbl_off + (b * K.hs >= K.mcux ? (Rs + 1) * K.S : ia + 1);
    p.how = (((Rs * K.mcux + Ca) * K.bpm) & 0xffu) | (((ms - m0) * bpm) << 8) | (uint32_t(K.mapbase) << 16) | (uint32_t(K.log2u) << 24);
    p.any = true;
    return p;
}
// Lanes 0 .. kPlanarSegs-1 of one wave, lane j with segment j's probe and table words: the tile's segment records and its rounds.
// (Lock step: what the lanes store to LDS here is read back by their neighbours further down without a barrier.)
template <uint32_t LANES>
__device__ __forceinline__ void planar_list(PlanarTile &t, const PlanarKindL *kinds, uint32_t nk, uint32_t j, const PlanarProbe &p,
                                            uint32_t start, uint32_t end)
{
    const uint32_t len = p.any && end > start ? min(end - start, 0xffffu) : 0u;
    t.start[j] = start;
    t.len[j] = uint16_t(len);
    t.how[j] = p.how;
    t.rnd[j].cnt = 0;                                              // (kPlanarSegs == the prefetched rounds: a lane clears the record of its number)
    if (j == 0) t.rest_seg = uint8_t(kPlanarSegs);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t before = 0;
    for (uint32_t i = 0; i < kPlanarSegs; i++) before += i < j ? (uint32_t(t.len[i]) + LANES - 1) / LANES : 0u;
    const uint32_t ent_rel = kinds[j >= nk ? j - nk : j].ent_rel;
    for (uint32_t r = 0; r * LANES < len; r++) {
        const uint32_t k = before + r;
        if (k < 8) t.rnd[k] = PlanarRound{ent_rel + start + r * LANES, p.how, min(LANES, len - r * LANES)};
        else if (k == 8) { t.rest_seg = uint8_t(j); t.rest_rnd = uint8_t(r); }
    }
}
// what a lane needs to find its block's DC value in the scan that carries its component (the same for every tile: a tile is whole MCUs)
struct PlanarDc {
    uint64_t base;                   // the scan's first block in dcbuf
    uint32_t ky, kx, k0;             // block index in the scan = my * ky + mx * kx + k0   (mx, my: the MCU's column and row in the picture)
    uint32_t rx, r0x, lx, ry, r0y, ly;       // the block exists in the scan if mx * rx + r0x < lx and my * ry + r0y < ly
    uint32_t mt;                     // the lane's MCU inside the tile
    bool lane;                       // the lane has a block slot at all
};
__device__ __forceinline__ PlanarDc planar_dc_prepare(const DevImage *images, const DevImage &im, uint32_t img, uint32_t bpm, uint32_t tile_blocks)
{
    PlanarDc d{};
    const uint32_t tid = threadIdx.x;
    d.lane = tid < tile_blocks;
    d.mt = tid / bpm;
    const uint32_t k = tid - d.mt * bpm, c = im.blk_comp[k];
    const DevImage &sim = images[img - im.src_back[c]];
    d.base = sim.coef_off;
    if (sim.ncomp == 1) {            // raster order over the component's own grid
        d.ky = im.cv[c] * im.cbw[c]; d.kx = im.ch[c]; d.k0 = im.blk_by[k] * im.cbw[c] + im.blk_bx[k];
        d.rx = im.ch[c]; d.r0x = im.blk_bx[k]; d.lx = im.cbw[c];
        d.ry = im.cv[c]; d.r0y = im.blk_by[k]; d.ly = im.cbh[c];
    } else {                         // interleaved subset: that scan's MCU order (its grid is the picture's: the host checks)
        d.ky = sim.mcux * sim.bpm; d.kx = sim.bpm; d.k0 = sim.cfirst[im.src_comp[c]] + im.blk_by[k] * im.ch[c] + im.blk_bx[k];
        d.rx = 1; d.r0x = 0; d.lx = sim.mcux;
        d.ry = 1; d.r0y = 0; d.ly = sim.mcuy;
    }
    return d;
}
template <int PF>
struct PlanarFetch {
    uint32_t ent[PF];
    int32_t dc;
};
// (src: the entry pool from the file's first scan's region on; mx, my: column and row in the picture of the tile's first MCU)
template <uint32_t LANES, int PF>
__device__ __forceinline__ void tile_fetch_planar(const uint32_t *__restrict__ src, const PlanarTile &t, const int32_t *__restrict__ dcbuf,
                                                  const PlanarDc &d, uint32_t m0, uint32_t r0, uint32_t a0, uint32_t nmcu, uint32_t mcux,
                                                  PlanarFetch<PF> &f)
{
    const uint32_t tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < PF; k++) {
        const uint32_t at = __builtin_amdgcn_readfirstlane(t.rnd[k].src), cnt = __builtin_amdgcn_readfirstlane(t.rnd[k].cnt);
        f.ent[k] = tid < cnt ? __builtin_nontemporal_load(src + at + tid) : 0u;
    }
    // the lane's MCU: d.mt MCUs behind the tile's first (a tile spans two MCU rows at most)
    uint32_t mx = a0 + d.mt, my = r0;
    if (mx >= mcux) { mx -= mcux; my++; }
    const bool real = d.lane && m0 + d.mt < nmcu && mx * d.rx + d.r0x < d.lx && my * d.ry + d.r0y < d.ly;
    f.dc = real ? __builtin_nontemporal_load(dcbuf + d.base + (my * d.ky + mx * d.kx + d.k0)) : 0;
}
template <int PF>
__device__ __forceinline__ void settle(PlanarFetch<PF> &f)
{
#pragma unroll
    for (int k = 0; k < PF; k++) asm volatile("" : "+v"(f.ent[k]));
    asm volatile("" : "+v"(f.dc));
}
// block slot in the tile of an entry of a round (how: PlanarRound::how, uniform)
__device__ __forceinline__ uint32_t planar_block(uint32_t e, uint32_t how, uint32_t bpm)
{
    const uint32_t jb = ((e >> 22) - how) & 0xffu;               // the block inside the segment
    const uint32_t l = how >> 24, fm = jb >> l;                   // its MCU of the picture, counted from the piece's first
    return min(((how >> 8) & 0xffu) + ((how >> 16) & 0xffu) + fm * bpm + (jb - (fm << l)), 255u);
}
template <int MODE, int N>
__device__ __forceinline__ void scatter_planar(const uint32_t *ent, const PlanarTile &t, uint32_t bpm, uint32_t nblk, float *tile_f,
                                               const float *s_qm, const uint8_t *s_nat, const uint8_t *s_comp)
{
    uint32_t bs[N];
#pragma unroll
    for (int k = 0; k < N; k++) bs[k] = planar_block(ent[k], __builtin_amdgcn_readfirstlane(t.rnd[k].how), bpm);
    scatter_at<MODE, N>(ent, bs, nblk, tile_f, s_qm, s_nat, s_comp);
}

// One lane = one 8x8 block: 16 x ds_read_b128 of its row, 8 row + 8 column transforms in registers, back to the row.
// The transforms run two at a time in packed fp32 instructions (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32: two floats of a
// register pair per instruction, at the rate of one plain instruction -- stage B is bound by vector-instruction issue, and the
// inverse DCT was two thirds of its vector instructions): the row transforms on pairs of neighbouring rows, the column
// transforms on pairs of neighbouring columns, with one transposition of the 2x2 sub-blocks between them.  For the first pass
// the coefficients must lie in the lane's row as pairs (row 2i, row 2i+1) of one column: idct_slot() is that order, and the
// scatter phase writes it (its table maps the zig-zag position straight to the slot); the second pass leaves the samples in
// plain row-major order for the pixel phase.
__device__ __host__ constexpr uint32_t idct_slot(uint32_t natural) { return (((natural >> 4) * 8u + (natural & 7u)) << 1) | ((natural >> 3) & 1u); }
__device__ __forceinline__ void idct_row_inplace(float *rowf)
{
    float_pair p[4][8];                 // p[i][k] = (row 2i, row 2i+1) of column k
    float4 *row = reinterpret_cast<float4 *>(rowf);
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const float4 t = row[q];
        p[q >> 2][2 * (q & 3)] = float_pair{t.x, t.y};
        p[q >> 2][2 * (q & 3) + 1] = float_pair{t.z, t.w};
    }
#pragma unroll
    for (int i = 0; i < 4; i++) idct8(p[i][0], p[i][1], p[i][2], p[i][3], p[i][4], p[i][5], p[i][6], p[i][7]);
    // (round 6: the second pass in plain instructions on the halves of the first pass's pairs -- no transposition, 272 plain operations
    // instead of 136 packed ones + 64 moves -- measured 7.79 / 7.80 against 7.67 / 7.68 ms per 4096 x 1080p: not kept.  A packed fp32
    // instruction takes 4 cycles of the SIMD, a plain one 2 with two or more waves ready and 4 for a lone wave: tools/probes/op_cost_probe.hip.)
    float_pair c[8][4];                 // c[r][j] = (column 2j, column 2j+1) of row r
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            c[2 * i][j] = __builtin_shufflevector(p[i][2 * j], p[i][2 * j + 1], 0, 2);
            c[2 * i + 1][j] = __builtin_shufflevector(p[i][2 * j], p[i][2 * j + 1], 1, 3);
        }
#pragma unroll
    for (int j = 0; j < 4; j++) idct8(c[0][j], c[1][j], c[2][j], c[3][j], c[4][j], c[5][j], c[6][j], c[7][j]);
#if defined(MJX_EXP_NO_WRITEBACK)      // (measurement build, garbage out: what the 16 ds_write_b128 of a block's samples cost -- the results stay "used")
#pragma unroll
    for (int q = 0; q < 16; q++) asm volatile("" :: "v"(c[q >> 1][2 * (q & 1)].x), "v"(c[q >> 1][2 * (q & 1)].y), "v"(c[q >> 1][2 * (q & 1) + 1].x), "v"(c[q >> 1][2 * (q & 1) + 1].y));
#else
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const float_pair a = c[q >> 1][2 * (q & 1)], b = c[q >> 1][2 * (q & 1) + 1];
        row[q] = make_float4(a.x, a.y, b.x, b.y);
    }
#endif
}

// Reads 4 horizontally adjacent samples of one component for the pixel strip starting at MCU-local (x, y);
// replicates when the component is subsampled (box replication: the reference never interpolates).
// Sampling ratios are 1 or 2 (jpeg/mod.rs:275-277), so the divisions are shifts.
// What the generic pixel phase needs of the picture's sampling layout, read ONCE per workgroup into registers: read through the
// descriptor inside the row loop, every field was fetched again after every store to the picture (a byte pointer may alias
// anything) -- the pixel phase of a 4:2:2 tile took 30.7 k cycles per wave where the 4:2:0 form takes 9.1 k for twice the pixels
// (round 5, tools/stamp_stage_b.py).
struct GenShape {
    uint32_t xsh[3], ysh[3];         // log2 of the replication of component c's samples
    uint32_t first[3], ch[3];        // its first block inside the MCU, its blocks per MCU row
    uint32_t ncomp, bpm, hmax, vmax, log2_tile, width, height, mcux;
};
__device__ __forceinline__ GenShape gen_shape(const DevImage &im)
{
    GenShape g;
    for (uint32_t c = 0; c < 3; c++) {
        g.xsh[c] = im.hmax > im.ch[c] ? 1 : 0;
        g.ysh[c] = im.vmax > im.cv[c] ? 1 : 0;
        g.first[c] = im.cfirst[c];
        g.ch[c] = im.ch[c];
    }
    g.ncomp = im.ncomp; g.bpm = im.bpm; g.hmax = im.hmax; g.vmax = im.vmax; g.log2_tile = im.log2_tile;
    g.width = im.width; g.height = im.height; g.mcux = im.mcux;
    return g;
}
__device__ __forceinline__ void load4(const float *tile, const GenShape &g, uint32_t mcu_blk0, uint32_t c,
                                      uint32_t x, uint32_t y, float out[4])
{
    const uint32_t xs = x >> g.xsh[c], ys = y >> g.ysh[c];
    const uint32_t blk = mcu_blk0 + g.first[c] + (ys >> 3) * g.ch[c] + (xs >> 3);
    const float *p = tile + blk * kPixStride + (ys & 7) * 8 + (xs & 7);
    if (!g.xsh[c]) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
        const float2 v = *reinterpret_cast<const float2 *>(p);
        out[0] = v.x; out[1] = v.x; out[2] = v.y; out[3] = v.y;
    }
}

#ifndef MJX_PIX_XCHG
#define MJX_PIX_XCHG 1
#endif
#ifndef MJX_PIX_PKADD
#define MJX_PIX_PKADD 1
#endif
#ifndef MJX_FETCH_EARLY
#define MJX_FETCH_EARLY 1
#endif
// (y.x + t.x, y.y + t.x) and (y.x + t.y, y.y + t.y): one v_pk_add_f32 each, the second operand's half chosen by op_sel / op_sel_hi
__device__ __forceinline__ float_pair pk_add_lo(float_pair y, float_pair t)
{
    float_pair d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(y), "v"(t));
    return d;
}
__device__ __forceinline__ float_pair pk_add_hi(float_pair y, float_pair t)
{
    float_pair d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(y), "v"(t));
    return d;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Phase 3 for 4:2:0 (Y 2x2, Cb 1x1, Cr 1x1): lane -> (MCU t, 4-pixel strip sx) fixed; each step handles a 4x2 pixel
// patch that shares one pair of chroma samples per component (box replication).  INTERIOR: every pixel of the tile
// lies inside the image and rows are 4-byte aligned, so all eight stores are unconditional 12-byte stores.
//   XCHG    the tile holds all its MCUs: every lane reads every sample it owns, as an exchange with zero, whether its pixels lie inside
//           the picture or not -- the tile is clean for the next one (no zero-fill pass)
//   BOUNDS  pixels may lie outside the picture (the last column of MCUs, the last row) or rows are not 4-byte aligned: stores are
//           predicated per pixel; without it (an INTERIOR tile) all eight stores are unconditional 12-byte stores
// (round 5: the exchange form used to need a tile inside one MCU row and fully inside the picture; a tile that wraps into the next
// row -- one in 7.5 at 4K, one in 3.75 at 1080p -- or touches the picture's edge took plain reads, bounds everywhere, and cost
// the tile behind it a zero-fill pass)
template <bool XCHG, bool BOUNDS>
__device__ __forceinline__ void pixels_420(uint32_t width, uint32_t height, uint32_t mcux, const float *tile, uint32_t m0,
                                           uint32_t nm, uint8_t *out_img, bool aligned)
{
    constexpr bool INTERIOR = XCHG && !BOUNDS;
    const uint32_t tid = threadIdx.x;
    const uint32_t q = tid % (kTile420 * 4), t = q >> 2, sx = q & 3;
    if (!XCHG && t >= nm) return;
    const uint32_t m = m0 + t;
    const uint32_t mx = m % mcux, my = m / mcux;
    const uint32_t px = mx * 16 + sx * 4;
    if (!XCHG && px >= width) return;
    const uint32_t npix = INTERIOR ? 4u : (px < width ? min(4u, width - px) : 0u);
    const float *ybase = tile + (t * 6 + (sx >> 1)) * kPixStride + (sx & 1) * 4;
    const float *cbase = tile + (t * 6 + 4) * kPixStride + sx * 2;
    uint8_t *col = out_img + (size_t(my) * 16 * width + px) * 3;
    // all the samples the lane converts in this tile are requested first (four row pairs: 48 registers, free here -- the
    // transform's registers are dead), so that the LDS round trips of the four steps overlap instead of each step waiting
    // for its own
    f32x4 ya[4], yb[4];
    f32x2 cb[4], cr[4];
#if defined(MJX_EXP_PLAIN_READS)       // (measurement build, garbage out: plain reads where the exchanges with zero are, no zero-fill pass either)
    if (false) {
#else
    if (XCHG && MJX_PIX_XCHG) {
#endif
        // An interior tile: every sample of the tile is read exactly once in this phase, by exactly one lane -- so the read is an
        // exchange with zero (ds_wrxchg), and the tile is clean for the next one's coefficients without a zero-fill pass
        // (52 KB of LDS stores and a barrier per tile).  The compiler does not track LDS operations inside asm statements, so
        // the sixteen exchanges and the wait for them are ONE statement: its outputs are defined when it ends, whatever the
        // register allocator or the scheduler place around it (round 4 had a statement per exchange and hand-counted staged
        // waits between them -- correct only as long as no copy of a result was placed in front of its wait).
        const f32x2 zero = {0.0f, 0.0f};
        uint32_t ay[4], ac[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t rp = tid / (kTile420 * 4) + 2 * j;
            const float *yp = ybase + (rp >> 2) * 2 * kPixStride + ((rp * 2) & 7) * 8;
            ay[j] = uint32_t(uintptr_t((const __attribute__((address_space(3))) float *)(yp)));
            ac[j] = uint32_t(uintptr_t((const __attribute__((address_space(3))) float *)(cbase + rp * 8)));
        }
        asm volatile(
            "ds_wrxchg2_rtn_b64 %0, %16, %24, %24 offset1:1\n\t"
            "ds_wrxchg2_rtn_b64 %1, %16, %24, %24 offset0:4 offset1:5\n\t"
            "ds_wrxchg_rtn_b64 %2, %20, %24\n\t"
            "ds_wrxchg_rtn_b64 %3, %20, %24 offset:%25\n\t"
            "ds_wrxchg2_rtn_b64 %4, %17, %24, %24 offset1:1\n\t"
            "ds_wrxchg2_rtn_b64 %5, %17, %24, %24 offset0:4 offset1:5\n\t"
            "ds_wrxchg_rtn_b64 %6, %21, %24\n\t"
            "ds_wrxchg_rtn_b64 %7, %21, %24 offset:%25\n\t"
            "ds_wrxchg2_rtn_b64 %8, %18, %24, %24 offset1:1\n\t"
            "ds_wrxchg2_rtn_b64 %9, %18, %24, %24 offset0:4 offset1:5\n\t"
            "ds_wrxchg_rtn_b64 %10, %22, %24\n\t"
            "ds_wrxchg_rtn_b64 %11, %22, %24 offset:%25\n\t"
            "ds_wrxchg2_rtn_b64 %12, %19, %24, %24 offset1:1\n\t"
            "ds_wrxchg2_rtn_b64 %13, %19, %24, %24 offset0:4 offset1:5\n\t"
            "ds_wrxchg_rtn_b64 %14, %23, %24\n\t"
            "ds_wrxchg_rtn_b64 %15, %23, %24 offset:%25\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(ya[0]), "=&v"(yb[0]), "=&v"(cb[0]), "=&v"(cr[0]), "=&v"(ya[1]), "=&v"(yb[1]), "=&v"(cb[1]), "=&v"(cr[1]),
              "=&v"(ya[2]), "=&v"(yb[2]), "=&v"(cb[2]), "=&v"(cr[2]), "=&v"(ya[3]), "=&v"(yb[3]), "=&v"(cb[3]), "=&v"(cr[3])
            : "v"(ay[0]), "v"(ay[1]), "v"(ay[2]), "v"(ay[3]), "v"(ac[0]), "v"(ac[1]), "v"(ac[2]), "v"(ac[3]), "v"(zero), "n"(kPixStride * 4)
            : "memory");
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t rp = tid / (kTile420 * 4) + 2 * j;                 // row pair 0..7 inside the MCU
            const float *yp = ybase + (rp >> 2) * 2 * kPixStride + ((rp * 2) & 7) * 8;
            ya[j] = *reinterpret_cast<const f32x4 *>(yp);
            yb[j] = *reinterpret_cast<const f32x4 *>(yp + 8);
            cb[j] = *reinterpret_cast<const f32x2 *>(cbase + rp * 8);
            cr[j] = *reinterpret_cast<const f32x2 *>(cbase + kPixStride + rp * 8);
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t rp = tid / (kTile420 * 4) + 2 * j;
        const uint32_t py = my * 16 + rp * 2;
        if (!INTERIOR && (py >= height || npix == 0)) break;
        const ChromaTerms c0 = chroma_terms(cb[j].x, cr[j].x), c1 = chroma_terms(cb[j].y, cr[j].y);
        Rgb p[4];
#if MJX_PIX_PKADD
        // two pixels per addition (v_pk_add_f32, the chroma term broadcast to both halves by op_sel): 12 packed instead of 24 plain
        // additions per lane and step.  A packed instruction takes 4 cycles of the SIMD whatever the occupancy, a plain one 2 -- when
        // two or more of the SIMD's waves have one ready, 4 for a lone wave (tools/probes/op_cost_probe.hip): never slower, and
        // twice as fast whenever the other workgroups of the CU wait for memory.
        const float_pair t0rg = {c0.r, c0.g}, t0b = {c0.b, c0.b}, t1rg = {c1.r, c1.g}, t1b = {c1.b, c1.b};
        auto rows = [&](const f32x4 &y, Rgb *q) {
            const float_pair y01 = __builtin_shufflevector(y, y, 0, 1), y23 = __builtin_shufflevector(y, y, 2, 3);
            const float_pair r01 = pk_add_lo(y01, t0rg), g01 = pk_add_hi(y01, t0rg), b01 = pk_add_lo(y01, t0b);
            const float_pair r23 = pk_add_lo(y23, t1rg), g23 = pk_add_hi(y23, t1rg), b23 = pk_add_lo(y23, t1b);
            q[0] = Rgb{r01.x, g01.x, b01.x}; q[1] = Rgb{r01.y, g01.y, b01.y};
            q[2] = Rgb{r23.x, g23.x, b23.x}; q[3] = Rgb{r23.y, g23.y, b23.y};
        };
        rows(ya[j], p);
#else
        p[0] = ycc_to_rgb(ya[j].x, c0); p[1] = ycc_to_rgb(ya[j].y, c0);
        p[2] = ycc_to_rgb(ya[j].z, c1); p[3] = ycc_to_rgb(ya[j].w, c1);
#endif
        uint8_t *dst = col + size_t(rp) * 2 * width * 3;
        if (INTERIOR) store_rgb4(dst, pack4(p));
        else store4(dst, pack4(p), aligned, npix);
        if (INTERIOR || py + 1 < height) {
#if MJX_PIX_PKADD
            rows(yb[j], p);
#else
            p[0] = ycc_to_rgb(yb[j].x, c0); p[1] = ycc_to_rgb(yb[j].y, c0);
            p[2] = ycc_to_rgb(yb[j].z, c1); p[3] = ycc_to_rgb(yb[j].w, c1);
#endif
            if (INTERIOR) store_rgb4(dst + size_t(width) * 3, pack4(p));
            else store4(dst + size_t(width) * 3, pack4(p), aligned, npix);
        }
    }
}

// Phase 3 for any sampling layout: 4-pixel strips; lane -> (MCU t, strip sx) is fixed, rows advance by 256/R per step.
// INTERIOR: every MCU of the tile lies in one MCU row and fully inside the picture, rows are 4-byte aligned -- no bounds, plain 12-byte stores
template <bool INTERIOR, bool COLOUR>
__device__ __forceinline__ void pixels_generic_t(const GenShape &g, const float *tile, uint32_t m0, uint32_t nm,
                                                 uint8_t *out_img, bool aligned)
{
    const uint32_t tid = threadIdx.x, bpm = g.bpm;
    const uint32_t lstrips = g.hmax == 2 ? 2u : 1u;          // log2 of the 4-pixel strips per MCU row (2*hmax)
    const uint32_t R = (1u << g.log2_tile) << lstrips;       // strips per pixel row of the tile (power of two <= 256)
    const uint32_t q = tid & (R - 1);
    const uint32_t t = q >> lstrips, sx = q & ((1u << lstrips) - 1);
    const uint32_t rows = 8 * g.vmax, lR = g.log2_tile + lstrips, rstep = 256u >> lR;
    if (t >= nm) return;
    const uint32_t m = m0 + t;
    const uint32_t my = m / g.mcux, mx = m - my * g.mcux;
    const uint32_t px = mx * 8 * g.hmax + sx * 4;
    if (!INTERIOR && px >= g.width) return;
    const uint32_t npix = INTERIOR ? 4u : min(4u, g.width - px);
    uint8_t *dst = out_img + (size_t(my * rows + (tid >> lR)) * g.width + px) * 3;
    const size_t dstep = size_t(rstep) * g.width * 3;
    for (uint32_t r = tid >> lR; r < rows; r += rstep, dst += dstep) {
        if (!INTERIOR && my * rows + r >= g.height) break;
        float yv[4], cbv[4], crv[4];
        load4(tile, g, t * bpm, 0, sx * 4, r, yv);
        Rgb p[4];
        if (COLOUR) {
            load4(tile, g, t * bpm, 1, sx * 4, r, cbv);
            load4(tile, g, t * bpm, 2, sx * 4, r, crv);
#pragma unroll
            for (int k = 0; k < 4; k++) p[k] = ycc_to_rgb(yv[k], chroma_terms(cbv[k], crv[k]));
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) p[k].r = p[k].g = p[k].b = yv[k];                          // decoder.rs:318-325 (+128 is in the samples)
        }
        if (INTERIOR) store_rgb4(dst, pack4(p));
        else store4(dst, pack4(p), aligned, npix);
    }
}
__device__ __forceinline__ void pixels_generic(const GenShape &g, const float *tile, uint32_t m0, uint32_t nm,
                                               uint8_t *out_img, bool aligned)
{
    const uint32_t T = 1u << g.log2_tile, my0 = m0 / g.mcux, mx0 = m0 - my0 * g.mcux;
    const bool interior = aligned && nm == T && mx0 + T <= g.mcux && (mx0 + T) * 8 * g.hmax <= g.width && (my0 + 1) * 8 * g.vmax <= g.height;
    if (g.ncomp == 3) {
        if (interior) pixels_generic_t<true, true>(g, tile, m0, nm, out_img, aligned);
        else pixels_generic_t<false, true>(g, tile, m0, nm, out_img, aligned);
    } else {
        if (interior) pixels_generic_t<true, false>(g, tile, m0, nm, out_img, aligned);
        else pixels_generic_t<false, false>(g, tile, m0, nm, out_img, aligned);
    }
}

// ---- REF_COMPAT placement (MODE 2) ---------------------------------------------------------------------------
// Reproduces decoder.rs:259-312 + fill_block_in_array (:347-379) bug for bug: the component's blocks
// are taken in decode order as a raster counter, get_indices maps the counter to a block position with formulas
// fitted to one image, lines are replicated horizontally but *tiled* vertically, nothing is clipped at the right
// edge, and later writes overwrite earlier ones.  The overwrite order is what makes this hard to parallelise; here
// every store carries its position in the reference's loop order as the high half of a 64-bit word and lands with
// atomicMax, so the surviving value is exactly the reference's last write.  Inputs on which the reference would
// index out of bounds are rejected on the host (mjx_plan.cpp), so every index below is in range.
__device__ __forceinline__ bool dev_get_indices(uint32_t x, uint32_t y, uint32_t max_x, uint32_t xf, uint32_t yf,
                                                uint32_t hmax, uint32_t vmax, uint32_t &ox, uint32_t &oy)
{
    if (vmax > 1 && yf == 1) {
        if (hmax > 1 && xf == 1) {
            if ((y & 1) == 0) {
                if (((x / 2) & 1) == 1) { ox = x / 2 - 1 + (x & 1); oy = y + 1; }
                else { ox = x / 2 + (x & 1); oy = y; }
                return true;
            }
            if (((x / 2) & 1) == 0) { ox = max_x / 2 + x / 2 - 1 + (x & 1); oy = y; return true; }
            ox = max_x / 2 + x / 2 + (x & 1); oy = y - 1;
            return true;
        }
        if ((y & 1) == 0) { ox = x / 2; oy = y + (x & 1); return true; }
        ox = x / 2 + max_x / 2; oy = y - (x & 1);
        return true;
    }
    ox = x; oy = y;
    return true;
}

__device__ __forceinline__ void place_ref(const DevImage &im, const float *tile, uint32_t tile_first_blk, uint32_t nblk,
                                          unsigned long long *planes)
{
    const uint32_t tid = threadIdx.x;
    if (tid >= nblk) return;
    const uint32_t b = tile_first_blk + tid, bpm = im.bpm;
    const uint32_t m = b / bpm, j = b % bpm, c = im.blk_comp[j];
    const uint32_t hv = uint32_t(im.ch[c]) * im.cv[c];
    const uint32_t block_i = m * hv + (j - im.cfirst[c]);
    const uint32_t xs = im.ref_xf[c], ys = im.ref_yf[c];
    const uint32_t cols = im.nbx / xs, rows = im.nby / ys;
    if (block_i >= cols * rows) return;                                    // decoder.rs:290-291 never reaches it
    const uint32_t x = block_i % cols, y = block_i / cols;
    uint32_t bx, by;
    dev_get_indices(x, y, im.nbx, xs, ys, im.hmax, im.vmax, bx, by);
    const size_t W = im.width, len = W * im.height;
    const size_t start_x = size_t(bx) * 8 * xs;
    if (W < start_x) return;                                               // :360
    unsigned long long *plane = planes + im.plane_off + size_t(c) * len;
    const float *blk = tile + tid * kPixStride;
    for (uint32_t line = 0; line < 8; line++) {
        const size_t start_i = size_t(by) * 8 * ys * W + size_t(line) * W + start_x;
        for (uint32_t ind = 0; ind < 8 * xs; ind++) {
            const float n = blk[line * 8 + ind / xs];                      // repeat(n).take(x_scale) :356
            const size_t i = start_i + ind;
            for (uint32_t jj = 0; jj < ys; jj++) {
                if (i + size_t(jj) * W < len) {                            // :370 guard; :371 index
                    const unsigned long long order = ((size_t(block_i) * 8 + line) * (8 * xs) + ind) * ys + jj + 1;
                    atomicMax(plane + i + size_t(jj) * W * 8, (order << 32) | __float_as_uint(n));
                }
            }
        }
    }
}

// decoder.rs:317-331 on the placed planes: one lane per pixel.
extern "C" __global__ __launch_bounds__(256) void k_ref_color(const DevImage *images, const unsigned long long *planes,
                                                               uint8_t *rgb, const uint32_t *img_flags)
{
    const DevImage &im = images[blockIdx.y];
    if (!im.valid || im.mode != 2 || img_flags[im.status_idx]) return;
    const size_t len = size_t(im.width) * im.height;
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= len) return;
    const unsigned long long *pl = planes + im.plane_off;
    const float y = __uint_as_float(uint32_t(pl[i]));
    uint32_t word = 0;
    if (im.ncomp == 3) {
        const float cb = __uint_as_float(uint32_t(pl[len + i])), cr = __uint_as_float(uint32_t(pl[2 * len + i]));
        const Rgb p = ycc_to_rgb(y + 128.0f, chroma_terms(cb, cr));
        word = pack_u8(p.b, 2, pack_u8(p.g, 1, pack_u8(p.r, 0, 0)));
    } else {
        const float v = y + 128.0f;
        word = pack_u8(v, 2, pack_u8(v, 1, pack_u8(v, 0, 0)));
    }
    uint8_t *dst = rgb + im.rgb_off + i * 3;
    dst[0] = uint8_t(word); dst[1] = uint8_t(word >> 8); dst[2] = uint8_t(word >> 16);
}

// ---- MODE 3: the wide 4:2:0 form (round 6; layout: kWRow above) ----------------------------------------------------------------
// Phase 2, the column pass: a lane owns four columns of a luminance block, or two columns of both chrominance blocks of an MCU --
// eight rows x four floats, one ds_read_b128 per row, two packed 8-point transforms down the columns (the two columns of a pair, or
// the Cb and the Cr column, side by side in packed fp32 instructions), eight ds_write_b128 back to where the rows came from.
// Lanes 0 .. 8T-1 are luminance (MCU t = lane % T, then super-row and q), lanes 8T .. 12T-1 chrominance: the kind is uniform over a
// wave, and the 16 lanes of a ds_read_b128 group are 16 MCUs of one q -- 16 different 16-byte slots (kWMcu / 4 = 99 is odd).
__device__ __forceinline__ void wide_column_pass(float *tile_f, uint32_t nm)
{
    constexpr uint32_t T = kTile420;
    const uint32_t tid = threadIdx.x;
    const uint32_t t = tid % T, g = tid / T;                 // g: 0..7 luminance (super-row g >> 2, q = g & 3), 8..11 chrominance (q = g - 8)
    if (g >= 12u || t >= nm) return;
    const bool chroma = g >= 8u;
    float4 *p = reinterpret_cast<float4 *>(tile_f + t * kWMcu + (chroma ? 2u : g >> 2) * kWRow + 8u * (g & 3u));
    float_pair a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float4 v = p[(r >> 1) * 8 + (r & 1)];
        a[r] = float_pair{v.x, v.y};
        b[r] = float_pair{v.z, v.w};
    }
    idct8(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
    idct8(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7]);
    if (chroma) {
#pragma unroll
        for (int r = 0; r < 8; r++) p[(r >> 1) * 8 + (r & 1)] = make_float4(a[r].x, a[r].y, b[r].x, b[r].y);
    } else {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            p[m * 8] = make_float4(a[2 * m].x, a[2 * m + 1].x, a[2 * m].y, a[2 * m + 1].y);
            p[m * 8 + 1] = make_float4(b[2 * m].x, b[2 * m + 1].x, b[2 * m].y, b[2 * m + 1].y);
        }
    }
}

__device__ __forceinline__ void store_rgb4_plain(uint8_t *dst, const Rgb4 &v)
{
#if defined(MJX_EXP_NO_RGB_STORE)
    asm volatile("" :: "v"(v.a), "v"(v.b), "v"(v.c), "v"(dst));
    return;
#endif
    *reinterpret_cast<Rgb4 *>(dst) = v;
}

// Phase 3, row pass + colour: lane -> (row pair k of the MCU, side sd, MCU t) = 8 x 2 pixels.  It takes the two luminance rows as
// eight (row 2m, row 2m+1) pairs -- 64 contiguous bytes -- and row k of both chrominance blocks as eight (Cb, Cr) pairs, runs one
// packed 8-point transform along each, keeps the four chrominance samples over its pixels (columns 4 sd .. 4 sd + 3; the other side's
// lane computes the same transform for the other four) and converts: 2 x 24 bytes, two 12-byte stores per row.  A wave's 64 lanes
// write the 8-pixel halves of 16 MCUs of two row pairs: 768 contiguous bytes per row from each pair of store instructions (plain
// stores: L2 merges the two halves of a line; as streaming stores the same shape took 19.8 instead of 9.1 ms per 51 GB,
// tools/probes/rgb_store_probe.hip).
//   XCHG   the tile holds all its MCUs: the luminance rows are read as an exchange with zero and the chrominance rows -- read by both
//          sides' lanes -- are cleared by them, two pieces each, behind the reads: the tile is clean for the next one
//   BOUNDS pixels may lie outside the picture, or rows are not 4-byte aligned: stores predicated per pixel
template <bool XCHG, bool BOUNDS>
__device__ __forceinline__ void pixels_wide(uint32_t width, uint32_t height, uint32_t mcux, float *tile, uint32_t m0, uint32_t nm,
                                            uint8_t *out_img, bool aligned)
{
    constexpr uint32_t T = kTile420;
    constexpr bool INTERIOR = XCHG && !BOUNDS;
    const uint32_t tid = threadIdx.x;
    const uint32_t t = tid % T, sd = (tid / T) & 1u, k = tid / (2u * T);
    if (!XCHG && t >= nm) return;
    const uint32_t m = m0 + t;
    const uint32_t mx = m % mcux, my = m / mcux;
    float *yb = tile + t * kWMcu + (k >> 2) * kWRow + (k & 3u) * 32u + sd * 16u;
    float *cb = tile + t * kWMcu + 2u * kWRow + (k >> 1) * 32u + (k & 1u) * 4u;
    float_pair y[8], c[8];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 v = *reinterpret_cast<const float4 *>(cb + 8 * q);
        c[2 * q] = float_pair{v.x, v.y};
        c[2 * q + 1] = float_pair{v.z, v.w};
    }
    if (XCHG && MJX_PIX_XCHG) {
        // (one statement with its wait: the compiler does not track LDS operations inside asm, see pixels_420)
        const f32x2 zero = {0.0f, 0.0f};
        const uint32_t ay = uint32_t(uintptr_t((const __attribute__((address_space(3))) float *)(yb)));
        f32x4 y0, y1, y2, y3;
        asm volatile(
            "ds_wrxchg2_rtn_b64 %0, %4, %5, %5 offset1:1\n\t"
            "ds_wrxchg2_rtn_b64 %1, %4, %5, %5 offset0:2 offset1:3\n\t"
            "ds_wrxchg2_rtn_b64 %2, %4, %5, %5 offset0:4 offset1:5\n\t"
            "ds_wrxchg2_rtn_b64 %3, %4, %5, %5 offset0:6 offset1:7\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
            : "v"(ay), "v"(zero)
            : "memory");
        y[0] = float_pair{y0.x, y0.y}; y[1] = float_pair{y0.z, y0.w};
        y[2] = float_pair{y1.x, y1.y}; y[3] = float_pair{y1.z, y1.w};
        y[4] = float_pair{y2.x, y2.y}; y[5] = float_pair{y2.z, y2.w};
        y[6] = float_pair{y3.x, y3.y}; y[7] = float_pair{y3.z, y3.w};
        // the chrominance row: this side's lane clears pieces 2 sd and 2 sd + 1 (every lane of the wave has read by now: the reads
        // above are earlier instructions of the same wave, and a wave's LDS instructions execute in order)
        *reinterpret_cast<float4 *>(cb + 16u * sd) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(cb + 16u * sd + 8u) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4 *>(yb + 4 * q);
            y[2 * q] = float_pair{v.x, v.y};
            y[2 * q + 1] = float_pair{v.z, v.w};
        }
    }
    idct8(y[0], y[1], y[2], y[3], y[4], y[5], y[6], y[7]);          // y[x] = (pixel x of row 2k, of row 2k + 1)
    idct8(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);          // c[x] = (Cb, Cr) sample x of chrominance row k
    const uint32_t px = mx * 16u + sd * 8u, py = my * 16u + k * 2u;
    uint8_t *dst = out_img + (size_t(py) * width + px) * 3;
    const uint32_t npix = INTERIOR ? 8u : (px < width ? min(8u, width - px) : 0u);
    if (!INTERIOR && (py >= height || npix == 0)) return;
#pragma unroll
    for (int h = 0; h < 2; h++) {                                    // four pixels x two rows at a time
        Rgb r0[4], r1[4];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float_pair cc = sd ? c[4 + 2 * h + i] : c[2 * h + i];
            const ChromaTerms ct = chroma_terms(cc.x, cc.y);
            const float_pair ya = y[4 * h + 2 * i], yb2 = y[4 * h + 2 * i + 1];
            r0[2 * i] = ycc_to_rgb(ya.x, ct); r0[2 * i + 1] = ycc_to_rgb(yb2.x, ct);
            r1[2 * i] = ycc_to_rgb(ya.y, ct); r1[2 * i + 1] = ycc_to_rgb(yb2.y, ct);
        }
        const uint32_t left = npix > 4u * h ? npix - 4u * h : 0u;
        if (INTERIOR) {
            store_rgb4_plain(dst + 12 * h, pack4(r0));
            store_rgb4_plain(dst + size_t(width) * 3 + 12 * h, pack4(r1));
        } else if (left) {
            store4(dst + 12 * h, pack4(r0), aligned, left);
            if (py + 1 < height) store4(dst + size_t(width) * 3 + 12 * h, pack4(r1), aligned, left);
        }
    }
}

// MODE 0: any sampling layout.  MODE 1: Y 2x2 + Cb 1x1 + Cr 1x1 (4:2:0, 6 blocks per MCU, tile = 32 MCUs).
// MODE 3: the same pictures (DevImage::mode 1) in the wide form (above): 16 lanes per MCU, tile = 16 MCUs.
// MODE 2: any sampling layout, REF_COMPAT placement into the f32 plane scratch (k_ref_color finishes the image).
//   phase 0  zero the tile's sample rows in LDS
//   phase 1  scatter the tile's slice of the compact coefficient stream into them, entry-parallel (lane i holds
//            entries i, i+256, ... prefetched during the previous tile), DC values from dcbuf (prediction-summed);
//            then issue the loads of the next tile
//   phase 2  one lane = one 8x8 block: float AAN inverse DCT in registers (transform.rs:55-87 up to rounding)
//   phase 3  chroma replication + YCbCr->RGB + packed stores
//   SRC  where the entries come from: 0 the picture's linear stream, 1 its quad-interleaved stream, 2 the linear streams of the scans of
//        a multi-scan picture (planar, above)
template <int MODE, int PF, int SRC>
__global__ __launch_bounds__(256) void k_idct_color(const DevImage *__restrict__ images,
                                                     const uint32_t *__restrict__ entries,
                                                     const uint32_t *__restrict__ tile_eoff,
                                                     const int32_t *__restrict__ dcbuf, const float *__restrict__ qmult,
                                                     uint8_t *__restrict__ rgb, unsigned long long *__restrict__ planes,
                                                     const uint32_t *__restrict__ img_flags, uint32_t tiles_per_wg)
{
    // (its own symbol: dynamic LDS arrays of one name share their alignment, and the entropy kernels ask for 2 KiB)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_px[];
    __shared__ float s_qm[3 * 64];
    __shared__ uint8_t s_nat[MODE == 3 ? 128 : 64];     // (MODE 3: the luminance table, then the chrominance one)
    __shared__ uint8_t s_comp[(MODE == 1 || MODE == 3) ? 4 : 256];     // (4:2:0: the component comes from arithmetic on the block index)
    // per tile of the workgroup (+ sentinel).  Linear stream: s_eoff = the tile's first entry.  Quad-interleaved stream (QUAD):
    // s_eoff = the subsequence and s_at = the entry in its column where the tile starts; s_cum: see quad_prepare
    constexpr bool QUAD = SRC == 1, PLANAR = SRC == 2;
    __shared__ uint32_t s_eoff[PLANAR ? 1 : kTilesPerWg + 1];
    __shared__ uint16_t s_at[QUAD ? kTilesPerWg + 1 : 1];
    __shared__ QuadCum s_cum[QUAD ? kTilesPerWg : 1];
    __shared__ PlanarKindL s_kind[PLANAR ? kPlanarKinds : 1];
    __shared__ PlanarTile s_ptile[PLANAR ? 3 : 1];
    const DevImage &im = images[blockIdx.y];
    if (!im.valid || im.mode != uint32_t(MODE == 3 ? 1 : MODE) || (im.planar ? 2 : im.ent_rows != 0 ? 1 : 0) != SRC || img_flags[im.status_idx]) return;
    // everything the tile loop needs from the descriptor, read once (uniform -> scalar registers)
    constexpr bool M420 = MODE == 1 || MODE == 3;
    constexpr uint32_t LANES = M420 ? kLanes420 : 256u;
    const uint32_t T = M420 ? kTile420 : (1u << im.log2_tile);
    const uint32_t bpm = M420 ? 6u : im.bpm;
    const uint32_t nmcu = im.nmcu, width = im.width, height = im.height, mcux = im.mcux;
    const uint32_t total_blocks = im.himg.total_blocks;
    const uint32_t tile_blocks = T * bpm;
    const uint32_t ntiles = (nmcu + T - 1) / T;
    const uint32_t tile0 = blockIdx.x * tiles_per_wg;             // (tiles_per_wg <= kTilesPerWg: few for small launches, see launch_idct_color)
    if (tile0 >= ntiles) return;
    const uint32_t tile1 = min(ntiles, tile0 + tiles_per_wg);
    const uint32_t tid = threadIdx.x;
    const uint32_t *__restrict__ src = entries + im.ent_off + im.ent_hdr;
    const uint32_t *__restrict__ eoff = tile_eoff + im.tile_off;
    const int32_t *__restrict__ dcs = dcbuf + im.coef_off;
    uint8_t *__restrict__ out_img = rgb + im.rgb_off;
    const bool aligned = ((width * 3u) & 3u) == 0 && (im.rgb_off & 3u) == 0;
    // The stream offsets of all the workgroup's tiles are fetched once: a tile's entries can then be requested without
    // first waiting for its offsets (two dependent round trips per tile were what paced the tile loop).
    // (Quad-interleaved stream: an offset is subsequence * column capacity + entry index in the column; split here, once.)
    if (!PLANAR && tid <= tiles_per_wg) {
        uint32_t v = eoff[min(tile0 + tid, ntiles)];
        if constexpr (QUAD) {
            const uint32_t cap = im.ent_rows * 8u, sub = v / cap;
            s_at[tid] = uint16_t(v - sub * cap);
            v = sub;
        }
        s_eoff[tid] = v;
    }
    const uint32_t nk = PLANAR ? im.pk_n : 0u, log2T = im.log2_tile;
    PlanarDc pdc{};
    const uint32_t *__restrict__ psrc = entries;             // planar: the entry pool from the file's first scan's region on
    uint32_t pr0 = 0, pa0 = 0;                                // ... row and column in the picture of the current tile's first MCU
    if constexpr (PLANAR) {
        const uint64_t base_off = images[blockIdx.y - im.nparts].ent_off;
        psrc = entries + base_off;
        pr0 = (tile0 * T) / mcux;
        pa0 = tile0 * T - pr0 * mcux;
        // the kinds of segments (one lane each), then the first two tiles' segments (the tile loop prepares two tiles ahead)
        if (tid < nk) {
            const DevImage &sim = images[blockIdx.y - im.pk_back[tid]];
            PlanarKindL K{};
            K.ent_rel = uint32_t(sim.ent_off - base_off);
            K.tbl_off = sim.tile_off;
            K.mcux = sim.mcux; K.mcuy = sim.mcuy;
            K.S = uint16_t(sim.seg_S);
            K.bpm = uint8_t(sim.bpm); K.hs = im.pk_hs[tid]; K.vs = im.pk_vs[tid]; K.v = im.pk_v[tid];
            K.log2u = uint8_t(31 - __builtin_clz(uint32_t(im.pk_u[tid]) | 1u)); K.mapbase = im.pk_map[tid][0];
            s_kind[tid] = K;
        }
        pdc = planar_dc_prepare(images, im, blockIdx.y, bpm, tile_blocks);
        __syncthreads();
        for (uint32_t which = 0; which < 2; which++) {          // (wave 0: see planar_list)
            if (tid < kPlanarSegs) {
                uint32_t a1 = pa0 + which * T, r1 = pr0;
                while (a1 >= mcux) { a1 -= mcux; r1++; }                     // (T = mcux + 1 from the row's last column: two rows on)
                const PlanarProbe pp = planar_probe(s_kind, nk, tid, (tile0 + which) * T, r1, a1, log2T, nmcu, mcux, bpm);
                const uint32_t st0 = pp.any ? tile_eoff[pp.ia] : 0u, en0 = pp.any ? tile_eoff[pp.ie] : 0u;
                planar_list<LANES>(s_ptile[(tile0 + which) % 3u], s_kind, nk, tid, pp, st0, en0);
            }
        }
    }
    __syncthreads();
    const QuadView qv{s_eoff, s_at, s_cum, entries + im.ent_off, im.ent_rows, im.himg.nsub};
    if constexpr (QUAD) {
        if (tid < tile1 - tile0) s_cum[tid] = quad_prepare(qv, tid);
        __syncthreads();
    }
    constexpr int QR = PF / 8;       // quad-interleaved stream: rounds that are prefetched
    static_assert(!QUAD || PF % 8 == 0, "whole groups per lane and round");
    static_assert(!PLANAR || PF == 8, "the planar form prefetches eight rounds");
    typename std::conditional<QUAD, QuadFetch<QR>, typename std::conditional<PLANAR, PlanarFetch<PF>, TileFetch<PF>>::type>::type cur;
    if constexpr (QUAD) tile_fetch_quad<LANES>(src, qv, 0, dcs, tile0, tile_blocks, total_blocks, cur);
    else if constexpr (PLANAR) tile_fetch_planar<LANES, PF>(psrc, s_ptile[tile0 % 3u], dcbuf, pdc, tile0 * T, pr0, pa0, nmcu, mcux, cur);
    else tile_fetch<LANES, PF>(src, s_eoff, dcs, tile0, tile_blocks, total_blocks, cur);
    for (uint32_t i = tid; i < 192; i += LANES) s_qm[i] = qmult[im.qm_off + i];
    if (tid < 64) {
        constexpr uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        if (MODE == 3) {    // ... -> the float inside the block's quarter of its super-row (kWRow above), luminance and chrominance
            const uint32_t r = ZZ[tid] >> 3, c = ZZ[tid] & 7u;
            s_nat[tid] = uint8_t((r >> 1) * 32u + (c >> 2) * 8u + (r & 1u) * 4u + (c & 3u));
            s_nat[64 + tid] = uint8_t((r >> 1) * 32u + (c >> 1) * 8u + (r & 1u) * 4u + (c & 1u) * 2u);
        } else
        s_nat[tid] = uint8_t(idct_slot(ZZ[tid]));      // zig-zag position -> where the inverse DCT expects the coefficient
    }
    if (!M420 && tid < tile_blocks) s_comp[tid] = im.blk_comp[tid % bpm];
    // the lane's own block slot (tid) has the same component in every tile (a tile is whole MCUs): its DC multiplier and level shift
    const uint32_t my_comp = im.blk_comp[tid % bpm];
    float my_dc_qm = qmult[im.qm_off + my_comp * 64];
    float my_dc_add = (MODE != 2 && my_comp == 0) ? 128.0f : 0.0f;
    float *tile_f = reinterpret_cast<float *>(smem_px);
    GenShape gshape{};
    if (MODE == 0) gshape = gen_shape(im);
    // (the first tile's words are settled before the loop, so that on no path into a tile iteration a load is pending
    // on them: see the settle point behind phase 2)
    settle(cur);
    // (... and the lane's DC multiplier: first used inside the loop, its load would be waited for there -- in every iteration, with
    // a vmcnt(0) that also drains the pixel stores of the tile before: the wait-count pass merges the loop's entry state into its
    // back edge.  Round 6: found in the ISA; it was a store drain per tile and wave right behind the pixel phase.)
    asm volatile("" : "+v"(my_dc_qm), "+v"(my_dc_add));
    if (!M420) asm volatile("" :: "v"(gshape.width));

#ifdef MJX_STAMP_B
    WaveStamp sp;
    sp.begin();
#define MJX_SB(k) sp.at(k)
#else
#define MJX_SB(k)
#endif
    bool clean = false;          // the tile's sample rows are zero already (the pixel phase of an interior 4:2:0 tile clears what it reads)
    for (uint32_t tile = tile0; tile < tile1; tile++) {
        const uint32_t m0 = tile * T;
        const uint32_t nm = min(T, nmcu - m0), nblk = nm * bpm;
        if (!clean) {   // phase 0
            float4 *z = reinterpret_cast<float4 *>(smem_px);
            const uint32_t nq = MODE == 3 ? nm * (kWMcu / 4) : nblk * (kPixStride / 4);
            for (uint32_t i = tid; i < nq; i += LANES) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
        }
        MJX_SB(0);
        // (planar: the two table words of the segment this lane prepares, two tiles ahead -- used at the settle point below)
        PlanarProbe ahead{0, 0, 0, false};
        uint32_t ahead_st = 0, ahead_en = 0;
        uint32_t pr1 = pr0, pa1 = pa0 + T;                        // the next tile's first MCU
        if constexpr (PLANAR) {
            while (pa1 >= mcux) { pa1 -= mcux; pr1++; }
            if (tid < kPlanarSegs && tile + 2 < tile1) {
                uint32_t a2 = pa1 + T, r2 = pr1;
                while (a2 >= mcux) { a2 -= mcux; r2++; }
                ahead = planar_probe(s_kind, nk, tid, (tile + 2) * T, r2, a2, log2T, nmcu, mcux, bpm);
                if (ahead.any) {
                    ahead_st = tile_eoff[ahead.ia];
                    ahead_en = tile_eoff[ahead.ie];
                }
            }
        }
        // The next tile's loads go out BEFORE this tile's entries are scattered (round 6; MJX_FETCH_EARLY=0: behind the scatter, as in
        // rounds 2-5): what finds a lane's group -- quad_cell's reads of the workgroup's tables in LDS -- is then not queued behind the
        // scatter's LDS stores, and the loads have the scatter phase on top of the inverse DCT to land in (13 more registers: cur and
        // nxt are live together; still three waves per SIMD).
        auto nxt = cur;
        if (MJX_FETCH_EARLY && !PLANAR && tile + 1 < tile1) {
            if constexpr (QUAD) tile_fetch_quad<LANES>(src, qv, tile + 1 - tile0, dcs, tile + 1, tile_blocks, total_blocks, nxt);
            else if constexpr (!PLANAR) tile_fetch<LANES, PF>(src, s_eoff + (tile + 1 - tile0), dcs, tile + 1, tile_blocks, total_blocks, nxt);
        }
        {   // phase 1
            const uint32_t first_lo = (tile * tile_blocks) & 0xffu;
            const uint32_t wave0 = tid & ~63u;
            if constexpr (PLANAR) {
                // as many rounds as the tile has -- a uniform choice between a few batch sizes, as for the linear stream below
                const PlanarTile &pt = s_ptile[tile % 3u];
                if (__builtin_amdgcn_readfirstlane(pt.rnd[4].cnt) == 0) scatter_planar<MODE, 4>(cur.ent, pt, bpm, nblk, tile_f, s_qm, s_nat, s_comp);
                else if (__builtin_amdgcn_readfirstlane(pt.rnd[6].cnt) == 0) scatter_planar<MODE, 6>(cur.ent, pt, bpm, nblk, tile_f, s_qm, s_nat, s_comp);
                else scatter_planar<MODE, 8>(cur.ent, pt, bpm, nblk, tile_f, s_qm, s_nat, s_comp);
                // what the tile has beyond the prefetched rounds (dense streams)
                for (uint32_t j = pt.rest_seg, r0 = pt.rest_rnd; j < kPlanarSegs; j++, r0 = 0) {
                    const uint32_t rel = s_kind[j >= nk ? j - nk : j].ent_rel + pt.start[j], len = pt.len[j], how = pt.how[j];
                    for (uint32_t i0 = r0 * LANES; i0 < len; i0 += LANES) {
                        uint32_t e1[1], b1[1];
                        e1[0] = i0 + tid < len ? psrc[rel + i0 + tid] : 0u;
                        b1[0] = planar_block(e1[0], how, bpm);
                        scatter_at<MODE, 1>(e1, b1, nblk, tile_f, s_qm, s_nat, s_comp);
                    }
                }
            } else if constexpr (QUAD) {
                // the prefetched groups of the tile (see quad_load); what a tile has beyond them takes further rounds
#pragma unroll
                for (int r = 0; r < QR; r++) {
                    if (cur.ncells > wave0 + LANES * r) {                                    // uniform over the wave
                        quad_mask(cur.ent[r], cur.k_lo[r], cur.k_hi[r]);
                        // (the labels of the group's entries + its subsequence's label offset = the blocks' indices in the picture, mod 256)
                        scatter_batch<MODE, 8>(cur.ent[r], first_lo - cur.lab[r], nblk, tile_f, s_qm, s_nat, s_comp);
                    }
                }
                for (uint32_t h0 = LANES * QR; h0 < cur.ncells; h0 += LANES) {
                    uint32_t more[8], k_lo, k_hi, lab;
                    quad_load(src, qv, tile - tile0, h0 + tid, more, k_lo, k_hi, lab);
                    quad_mask(more, k_lo, k_hi);
                    scatter_batch<MODE, 8>(more, first_lo - lab, nblk, tile_f, s_qm, s_nat, s_comp);
                }
            } else {
                // The prefetched words the wave really has (a tile of the bench content holds ~1500 entries, 5.9 per lane; at quality
                // 90 twice that): the batch is cut to that many slots -- a wave-uniform choice between a few batch sizes, so the
                // reads of a batch stay in flight together and empty slots cost nothing.  (Words behind the tile's last entry were
                // fetched as 0: null entries.)
                const uint32_t have = cur.e1 - cur.e0;
                const uint32_t slots = have > wave0 ? (have - wave0 + LANES - 1) / LANES : 0u;      // uniform over the wave
                if (slots <= 4) scatter_batch<MODE, 4>(cur.ent, first_lo, nblk, tile_f, s_qm, s_nat, s_comp);
                else if (slots <= 6) scatter_batch<MODE, 6>(cur.ent, first_lo, nblk, tile_f, s_qm, s_nat, s_comp);
                else if (slots <= 8 || PF == 8) scatter_batch<MODE, 8>(cur.ent, first_lo, nblk, tile_f, s_qm, s_nat, s_comp);
                else if (PF > 8) scatter_batch<MODE, PF>(cur.ent, first_lo, nblk, tile_f, s_qm, s_nat, s_comp);
                for (uint32_t i0 = cur.e0 + LANES * PF; i0 < cur.e1; i0 += LANES * 4) {     // what a dense tile has beyond the prefetched words
                    uint32_t more[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t i = i0 + tid + LANES * k;
                        more[k] = i < cur.e1 ? src[i] : 0u;
                    }
                    scatter_batch<MODE, 4>(more, first_lo, nblk, tile_f, s_qm, s_nat, s_comp);
                }
            }
            // DC; luminance blocks also take the + 128 of decoder.rs:318-330 here (a constant on the DC term of the
            // prescaled transform is the same constant on all 64 samples); REF_COMPAT adds it per pixel in k_ref_color,
            // where samples no block covers must come out as 0 + 128
            MJX_SB(1);
            if (MODE == 3) {
                if (tid < nblk) *reinterpret_cast<float *>(smem_px + wide_block_bytes(tid)) = float(cur.dc) * my_dc_qm + my_dc_add;
            } else
            if (tid < nblk) tile_f[tid * kPixStride] = float(cur.dc) * my_dc_qm + my_dc_add;
        }
        if (!(MJX_FETCH_EARLY && !PLANAR) && tile + 1 < tile1) {
            if constexpr (QUAD) tile_fetch_quad<LANES>(src, qv, tile + 1 - tile0, dcs, tile + 1, tile_blocks, total_blocks, nxt);
            else if constexpr (PLANAR) tile_fetch_planar<LANES, PF>(psrc, s_ptile[(tile + 1) % 3u], dcbuf, pdc, (tile + 1) * T, pr1, pa1, nmcu, mcux, nxt);
            else tile_fetch<LANES, PF>(src, s_eoff + (tile + 1 - tile0), dcs, tile + 1, tile_blocks, total_blocks, nxt);
        }
        MJX_SB(2);
        __syncthreads();
        MJX_SB(3);
        if (MODE == 3) wide_column_pass(tile_f, nm);                      // phase 2
        else
        if (tid < nblk) idct_row_inplace(tile_f + tid * kPixStride);
        MJX_SB(4);
        __syncthreads();
        MJX_SB(3);
        // The next tile's prefetched words are made to land here, a whole IDCT phase after their loads were issued and
        // before this tile's pixel stores go out.  Left to the next iteration, the wait for them is a vmcnt(0) that
        // sits behind those stores -- a full store drain per tile.
        settle(nxt);
        if constexpr (PLANAR) {
            // the segments of the tile after the next one (its records were the previous tile's: nobody reads them any more)
            if (tid < kPlanarSegs && tile + 2 < tile1) planar_list<LANES>(s_ptile[(tile + 2) % 3u], s_kind, nk, tid, ahead, ahead_st, ahead_en);
            pr0 = pr1;
            pa0 = pa1;
        }
        MJX_SB(5);
        if (MODE == 3) {                                                  // phase 3
            const uint32_t mx0 = m0 % mcux, my1 = (m0 + T - 1) / mcux;
            const bool whole = nm == T;
            const bool interior = whole && aligned && (my1 + 1) * 16 <= height && (mcux * 16 <= width || mx0 + T < mcux);
            if (interior) pixels_wide<true, false>(width, height, mcux, tile_f, m0, nm, out_img, aligned);
            else if (whole) pixels_wide<true, true>(width, height, mcux, tile_f, m0, nm, out_img, aligned);
            else pixels_wide<false, true>(width, height, mcux, tile_f, m0, nm, out_img, aligned);
            clean = whole && MJX_PIX_XCHG;
        } else if (MODE == 1) {
            // interior tile: all 32 MCUs in one MCU row, fully inside the image, rows 4-byte aligned
            // whole tile: all its 32 MCUs exist; interior: ... and every one of them lies fully inside the picture (the tile may wrap
            // into the next MCU row), rows 4-byte aligned
            const uint32_t mx0 = m0 % mcux, my1 = (m0 + T - 1) / mcux;
            const bool whole = nm == T;
            const bool interior = whole && aligned && (my1 + 1) * 16 <= height && (mcux * 16 <= width || mx0 + T < mcux);
            if (interior) pixels_420<true, false>(width, height, mcux, tile_f, m0, nm, out_img, aligned);
            else if (whole) pixels_420<true, true>(width, height, mcux, tile_f, m0, nm, out_img, aligned);
            else pixels_420<false, true>(width, height, mcux, tile_f, m0, nm, out_img, aligned);
            clean = whole && MJX_PIX_XCHG;
        } else if (MODE == 2) {
            place_ref(im, tile_f, tile * tile_blocks, nblk, planes);
        } else {
            pixels_generic(gshape, tile_f, m0, nm, out_img, aligned);
        }
        MJX_SB(6);
        __syncthreads();
        MJX_SB(7);
        cur = nxt;
    }
#ifdef MJX_STAMP_B
    sp.end();
#endif
#undef MJX_SB
}

// ---- verification helper: byte-wise comparison of decoded pictures on the device ------------------------------
// (mjx_batch_compare_rgb: the parity gate of bench.py and the batch-scale tests compare tens of gigabytes of output
// without copying them to the host.)  One workgroup per 16 KiB of a pair; per pair the largest absolute byte
// difference and the number of differing bytes.
extern "C" __global__ __launch_bounds__(256) void k_rgb_compare(const RgbPair *pairs, uint32_t *maxdiff,
                                                                 unsigned long long *ndiff)
{
    const RgbPair p = pairs[blockIdx.y];
    const uint64_t base = uint64_t(blockIdx.x) * 16384u;
    if (base >= p.bytes) return;
    const uint8_t *a = p.a, *b = p.b;
    uint32_t mx = 0, cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint64_t i = base + (uint64_t(k) * 256u + threadIdx.x) * 16u;
        if (i + 16 <= p.bytes) {
            const uint4 va = *reinterpret_cast<const uint4 *>(a + i), vb = *reinterpret_cast<const uint4 *>(b + i);
            const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int d = int((wa[w] >> (8 * j)) & 0xffu) - int((wb[w] >> (8 * j)) & 0xffu);
                    const uint32_t ad = uint32_t(d < 0 ? -d : d);
                    mx = ad > mx ? ad : mx;
                    cnt += ad != 0;
                }
        } else {
            for (uint64_t j = i; j < p.bytes && j < i + 16; j++) {
                const int d = int(a[j]) - int(b[j]);
                const uint32_t ad = uint32_t(d < 0 ? -d : d);
                mx = ad > mx ? ad : mx;
                cnt += ad != 0;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(mx, d);
        mx = o > mx ? o : mx;
        cnt += __shfl_xor(cnt, d);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicMax(maxdiff + blockIdx.y, mx);
        atomicAdd(ndiff + blockIdx.y, (unsigned long long)cnt);
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers (declared in mjx_kernels.h)
// ------------------------------------------------------------------------------------------------
size_t huff_lds_bytes(uint32_t lut_cap_entries) { return (sizeof(HuffImage) + size_t(lut_cap_entries) * sizeof(LutEntry) + 15) / 16 * 16; }
size_t huff_window_bytes(uint32_t lanes) { return size_t(lanes) * kWinStride * 4; }
size_t huff_window_bytes() { return huff_window_bytes(uint32_t(kHuffWg)); }
size_t huff_merge_bytes() { return size_t(kMergeWg) * kMergeStride * 4 + (kMergeWg / 64 + 1) * 4; }
size_t huff_stage_bytes(uint32_t lanes) { return size_t(lanes) * (kAcRingStride * 4 + DcRing16::kRing * 2); }    // the write pass's rings
size_t huff_stage_bytes() { return huff_stage_bytes(uint32_t(kHuffWg)); }

uint32_t tile_mcus_420() { return kTile420; }
uint32_t stream_group_entries() { return kAcGroup; }
size_t idct_lds_bytes(uint32_t max_tile_blocks) { return size_t(max_tile_blocks) * kPixStride * 4; }

int configure_kernels(size_t huff_lds, size_t idct_lds)
{
    hipError_t e = hipSuccess;
    if (huff_lds > 64 * 1024) {   // tables + windows + occupancy padding of the largest entropy launch
        const int cap = int(huff_lds);
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_huff_spec), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_huff_merge), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_huff_merge_loop), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_huff_write), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_huff_emit), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_huff_prefix), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    }
    if (e == hipSuccess && idct_lds > 64 * 1024) {
        const void *fns[] = {reinterpret_cast<const void *>(k_idct_color<0, kPrefetch, 0>), reinterpret_cast<const void *>(k_idct_color<0, 8, 1>),
                             reinterpret_cast<const void *>(k_idct_color<kMode420, kPrefetch, 0>), reinterpret_cast<const void *>(k_idct_color<kMode420, kPrefetchDense, 0>),
                             reinterpret_cast<const void *>(k_idct_color<kMode420, 8, 1>), reinterpret_cast<const void *>(k_idct_color<kMode420, 16, 1>),
                             reinterpret_cast<const void *>(k_idct_color<2, kPrefetch, 0>), reinterpret_cast<const void *>(k_idct_color<2, 8, 1>),
                             reinterpret_cast<const void *>(k_idct_color<0, 8, 2>), reinterpret_cast<const void *>(k_idct_color<kMode420, 8, 2>)};
        for (const void *f : fns)
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(idct_lds));
    }
    return e == hipSuccess ? 0 : int(e);
}

void launch_destuff(hipStream_t st, uint32_t max_seg, uint32_t nimg, bool any_restarts, const DestuffImg *imgs, const uint8_t *raw,
                    uint32_t *segcount, uint32_t *segbase, uint8_t *pool, uint32_t *rst_off, DevImage *images, InterleaveImg *ii,
                    uint32_t *segs, uint32_t *img_flags, uint8_t *scan_pool)
{
    if (nimg == 0) return;
    for (uint32_t at = 0; at < nimg; at += 32768) {                          // (a grid's y dimension holds 65 535 images)
        const uint32_t cnt = std::min<uint32_t>(32768, nimg - at);
        hipLaunchKernelGGL(k_destuff_count, dim3(max_seg, cnt), dim3(256), 0, st, imgs + at, raw, reinterpret_cast<uint2 *>(segcount));
    }
    hipLaunchKernelGGL(k_destuff_prefix, dim3(nimg), dim3(256), 0, st, imgs, reinterpret_cast<const uint2 *>(segcount), reinterpret_cast<uint2 *>(segbase),
                       images, ii, img_flags);
    for (uint32_t at = 0; at < nimg; at += 32768) {
        const uint32_t cnt = std::min<uint32_t>(32768, nimg - at);
        hipLaunchKernelGGL(k_destuff_scatter, dim3(max_seg, cnt), dim3(256), 0, st, imgs + at, raw, reinterpret_cast<const uint2 *>(segbase), pool, rst_off, images, scan_pool);
    }
    if (any_restarts) hipLaunchKernelGGL(k_restart_geometry, dim3(nimg), dim3(256), 0, st, imgs, images, rst_off, segs, img_flags);
}

void launch_scan_interleave(hipStream_t st, uint32_t max_pieces, uint32_t nimg, const InterleaveImg *imgs, const DevImage *images,
                            const uint8_t *linear, uint8_t *pool, const uint32_t *segs)
{
    if (max_pieces == 0 || nimg == 0) return;
    hipLaunchKernelGGL(k_scan_interleave, dim3((max_pieces + 255) / 256, nimg), dim3(256), 0, st, imgs, images, linear, pool, segs);
}

void launch_huff_spec(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                      const uint8_t *scan_pool, const LutEntry *lut_pool, SubseqState *entry, SubseqState *exit_,
                      uint32_t *cps, const uint32_t *segs, uint32_t lanes, uint8_t *gen)
{
    // (lanes: 512, or 256 / 128 for a chunk of short scans -- max_wg counts workgroups of that size; the occupancy pad belongs to the full size)
    const size_t lds = tables_lds + huff_window_bytes(lanes) + (lanes == uint32_t(kHuffWg) ? pad_lds : 0);
    hipLaunchKernelGGL(k_huff_spec, entropy_grid(max_wg, nimg), dim3(lanes), lds, st, images, scan_pool, lut_pool, entry, exit_, cps, uint32_t(tables_lds), segs, gen);
}

void launch_huff_merge(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                       const uint8_t *scan_pool, const LutEntry *lut_pool, SubseqState *entry, SubseqState *exit_,
                       uint32_t *cps, uint32_t *mismatches, uint32_t *items, uint32_t *item_count, const uint32_t *segs,
                       const uint32_t *prev_mismatches, bool first_round, EmitSub *esub, uint32_t max_items, uint8_t *gen, uint32_t gen_stride)
{
    // (item_count: this round's straggler counts, one per image, zeroed by the caller -- one memset for all the rounds of a chunk)
    const size_t lds = tables_lds + huff_merge_bytes() + pad_lds;
    hipLaunchKernelGGL(k_huff_merge, entropy_grid(max_wg, nimg), dim3(kMergeWg), lds, st, images, scan_pool, lut_pool, entry, exit_, cps, mismatches, uint32_t(tables_lds), items, item_count, segs, prev_mismatches,
                       first_round ? uint32_t(kHeadSlices) : uint32_t(MJX_LATER_HEAD_SLICES), esub, gen, gen_stride);
    const size_t tail_lds = tables_lds + size_t(kTailWg) * kMergeStride * 4;
    // (max_items: the most items a picture of the chunk can list = its subsequences - 1.  A chunk of small pictures pays for these
    // rounds in workgroup launches -- 32768 x 256x256: a million per step --: no second group of 256 for pictures that cannot fill one)
    const uint32_t groups = std::max<uint32_t>(1u, std::min<uint32_t>(max_wg * (kMergeWg / kTailWg), (max_items + kTailWg - 1) / kTailWg));
    hipLaunchKernelGGL(k_huff_merge_tail, dim3(nimg, groups), dim3(kTailWg), tail_lds, st, images, scan_pool, lut_pool, entry, exit_, cps, uint32_t(tables_lds), items, item_count, segs, esub, gen, gen_stride);
}

void launch_huff_merge_loop(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                            const uint8_t *scan_pool, const LutEntry *lut_pool, SubseqState *entry, SubseqState *exit_,
                            uint32_t *cps, uint32_t *verdict, const uint32_t *segs, uint32_t *ctl, uint32_t participants, uint32_t max_rounds,
                            uint32_t spin_limit, EmitSub *esub, uint8_t *gen, uint32_t gen_stride)
{
    const size_t lds = tables_lds + huff_merge_bytes() + pad_lds;
    hipLaunchKernelGGL(k_huff_merge_loop, dim3(max_wg, nimg), dim3(kMergeWg), lds, st, images, scan_pool, lut_pool, entry, exit_, cps, verdict, uint32_t(tables_lds), segs, ctl, participants, max_rounds, spin_limit, esub, gen, gen_stride);
}

void launch_emit_off(hipStream_t st, DevImage *images, uint32_t nimg, const uint32_t *img_flags)
{
    if (nimg) hipLaunchKernelGGL(k_emit_off, dim3((nimg + 255) / 256), dim3(256), 0, st, images, nimg, img_flags);
}

size_t huff_prefix_bytes() { return size_t(kPrefixWg) * kMergeStride * 4; }     // k_huff_prefix: a window per lane

void launch_huff_emit(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                      const uint8_t *scan_pool, const LutEntry *lut_pool, SubseqState *entry, SubseqState *exit_, uint32_t *cps,
                      EmitSub *esub, uint32_t *entries, uint32_t lanes)
{
    const size_t lds = tables_lds + huff_window_bytes(lanes) + huff_stage_bytes(lanes) + (lanes == uint32_t(kHuffWg) ? pad_lds : 0);
    hipLaunchKernelGGL(k_huff_emit, entropy_grid(max_wg, nimg), dim3(lanes), lds, st, images, scan_pool, lut_pool, entry, exit_, cps, esub, entries, uint32_t(tables_lds));
}

void launch_huff_prefix(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, const DevImage *images,
                        const uint8_t *scan_pool, const LutEntry *lut_pool, const SubseqState *entry, const SubseqState *exit_,
                        const uint32_t *cps, EmitSub *esub, const uint32_t *blkbase, uint32_t *entries, int *status, uint32_t *img_flags,
                        uint32_t *fallback, int16_t *dcdiff, uint32_t *tile_eoff, const uint32_t *items, const uint32_t *item_count, uint32_t *unconverged,
                        uint32_t lanes)
{
    // (the picture is the fast grid dimension, as in the other entropy kernels; a wave per 64 listed items -- the grid is sized for
    // the list's capacity, the waves beyond a picture's count leave at once)
    const uint32_t groups = max_wg * (lanes / kPrefixWg) * kItemDwords;     // (the list's capacity; a picture of the bench has ~270 items: five waves)
    hipLaunchKernelGGL(k_huff_prefix, dim3(nimg, groups), dim3(kPrefixWg), tables_lds + huff_prefix_bytes(), st, images, scan_pool, lut_pool, entry, exit_, cps, esub,
                       blkbase, entries, status, img_flags, fallback, uint32_t(tables_lds), items, item_count, unconverged);
    hipLaunchKernelGGL(k_block_gather, dim3(nimg, (max_wg * lanes + kGatherSubs - 1) / kGatherSubs), dim3(256), 0, st, images, entry, exit_, esub, blkbase,
                       const_cast<uint32_t *>(entries), dcdiff, tile_eoff, img_flags, status);
}

void launch_huff_scan(hipStream_t st, uint32_t nimg, const DevImage *images, SubseqState *exit_, uint32_t *blkbase,
                      uint32_t *ebase, uint32_t *img_entries, uint32_t *img_flags, const uint32_t *segs, const uint32_t *verdict,
                      const EmitSub *esub, uint32_t *items, uint32_t *item_count, uint32_t *fallback, uint32_t *unconverged,
                      SubseqState *entry, uint8_t *gen, uint32_t gen_stride)
{
    hipLaunchKernelGGL(k_huff_scan, dim3(nimg), dim3(kWgLanes), 0, st, images, exit_, blkbase, ebase, img_entries, img_flags, segs, verdict, esub, items, item_count, fallback, unconverged, entry, gen, gen_stride);
}

void launch_huff_write(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                       const uint8_t *scan_pool, const LutEntry *lut_pool, const SubseqState *entry,
                       const uint32_t *blkbase, const uint32_t *ebase, uint32_t *entries, uint32_t *tile_eoff,
                       int16_t *dcdiff, int *status, const uint32_t *img_flags, const uint32_t *segs, const SubseqState *exit_, uint32_t *cps,
                       uint32_t lanes)
{
    const size_t lds = tables_lds + huff_window_bytes(lanes) + huff_stage_bytes(lanes) + (lanes == uint32_t(kHuffWg) ? pad_lds : 0);
    hipLaunchKernelGGL(k_huff_write, entropy_grid(max_wg, nimg), dim3(lanes), lds, st, images, scan_pool, lut_pool, entry, blkbase, ebase, entries, tile_eoff, dcdiff, status, img_flags, uint32_t(tables_lds), segs, exit_, cps);
}

void launch_dc_scan(hipStream_t st, uint32_t max_segs, uint32_t nimg, const DevImage *images, const int16_t *dcd, int32_t *dcbuf,
                    int32_t *segsum, const uint32_t *img_flags, uint32_t bpm_mask, uint32_t max_restart_segs,
                    uint32_t *segflag, uint32_t gen, uint32_t *fail, uint32_t spin_limit, bool fault)
{
    const dim3 grid(max_segs, nimg), wg(256);
    if (segflag) {          // the common MCU shapes in one pass (k_dc_scan_t); segflag == nullptr: two passes as before
        const dim3 grid(nimg, max_segs);
        if (bpm_mask & (1u << 1)) hipLaunchKernelGGL(k_dc_scan_t<1>, grid, wg, 0, st, images, dcd, dcbuf, segflag, max_segs, img_flags, gen, fail, spin_limit, fault ? 1u : 0u);
        if (bpm_mask & (1u << 3)) hipLaunchKernelGGL(k_dc_scan_t<3>, grid, wg, 0, st, images, dcd, dcbuf, segflag, max_segs, img_flags, gen, fail, spin_limit, fault ? 1u : 0u);
        if (bpm_mask & (1u << 4)) hipLaunchKernelGGL(k_dc_scan_t<4>, grid, wg, 0, st, images, dcd, dcbuf, segflag, max_segs, img_flags, gen, fail, spin_limit, fault ? 1u : 0u);
        if (bpm_mask & (1u << 6)) hipLaunchKernelGGL(k_dc_scan_t<6>, grid, wg, 0, st, images, dcd, dcbuf, segflag, max_segs, img_flags, gen, fail, spin_limit, fault ? 1u : 0u);
        bpm_mask &= ~kDcFastShapes;
    }
#define MJX_DC_PASS(KERNEL, ...)                                                                                           \
    if (bpm_mask & (1u << 1)) hipLaunchKernelGGL(KERNEL##_t<1>, grid, wg, 0, st, images, __VA_ARGS__, segsum, max_segs, img_flags); \
    if (bpm_mask & (1u << 3)) hipLaunchKernelGGL(KERNEL##_t<3>, grid, wg, 0, st, images, __VA_ARGS__, segsum, max_segs, img_flags); \
    if (bpm_mask & (1u << 4)) hipLaunchKernelGGL(KERNEL##_t<4>, grid, wg, 0, st, images, __VA_ARGS__, segsum, max_segs, img_flags); \
    if (bpm_mask & (1u << 6)) hipLaunchKernelGGL(KERNEL##_t<6>, grid, wg, 0, st, images, __VA_ARGS__, segsum, max_segs, img_flags); \
    if (bpm_mask & ~kDcFastShapes) hipLaunchKernelGGL(KERNEL, grid, wg, 0, st, images, __VA_ARGS__, segsum, max_segs, img_flags);
    MJX_DC_PASS(k_dc_sums, dcd)
    MJX_DC_PASS(k_dc_apply, dcd, dcbuf)
#undef MJX_DC_PASS
    if (max_restart_segs) hipLaunchKernelGGL(k_dc_restart, dim3((max_restart_segs + 255) / 256, nimg), wg, 0, st, images, dcd, dcbuf, img_flags);
}

void launch_idct_color(hipStream_t st, uint32_t max_tiles, uint32_t nimg, size_t lds, const DevImage *images,
                       const uint32_t *entries, const uint32_t *tile_eoff, const int32_t *dcbuf, const float *qmult,
                       uint8_t *rgb, uint32_t mode_mask, unsigned long long *planes, const uint32_t *img_flags, bool dense,
                       uint32_t layout_mask, size_t lds_pad)
{
    // A workgroup walks up to kTilesPerWg consecutive tiles of its image (offsets fetched once, the next tile's loads in
    // flight during this tile's arithmetic) -- when the launch has tiles to spare: with fewer than a few rounds of 3
    // workgroups x 256 CUs the tiles are spread instead (one picture of 512x512 is 43 tiles: 3 workgroups took 195 us,
    // 43 take 20).
    const uint64_t total = uint64_t(max_tiles) * nimg;
    const uint32_t tpw = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(kTilesPerWg), total / 3072)));
    const uint32_t gx = (max_tiles + tpw - 1) / tpw;
    // (layout_mask: bit 0 = the chunk has pictures with a linear stream, bit 1 = with a quad-interleaved one, bit 2 = multi-scan
    // pictures read straight from their scans' streams; a kernel form leaves the other kinds' pictures alone)
    // (the 4:2:0 form's tile has its own size: the wide form's 25 KB must not be rounded up to the other pictures' tiles)
    const size_t lds420 = MJX_WIDE420 ? wide_tile_bytes() + lds_pad : lds;
#define MJX_IDCT(M, P, Q) hipLaunchKernelGGL((k_idct_color<M, P, Q>), dim3(gx, nimg), dim3((M == 1 || M == 3) ? kLanes420 : 256u), (M == 3 ? lds420 : lds), st, images, entries, tile_eoff, dcbuf, qmult, rgb, planes, img_flags, tpw)
    if (mode_mask & 1u) {
        if (layout_mask & 1u) MJX_IDCT(0, kPrefetch, 0);
        if (layout_mask & 2u) MJX_IDCT(0, 8, 1);
        if (layout_mask & 4u) MJX_IDCT(0, 8, 2);
    }
    if (mode_mask & 2u) {
        // (linear streams that are dense -- more than ~2048 entries per tile: quality 90 and up -- take the form that prefetches
        // twelve words per lane instead of eight: 18.2 -> 17.2 ms per 2048 4K pictures at quality 90; at quality 75 the four
        // extra loads per lane and tile cost 0.15 ms)
        if (layout_mask & 1u) {
            if (dense) MJX_IDCT(kMode420, kPrefetchDense, 0);
            else MJX_IDCT(kMode420, kPrefetch, 0);
        }
        // (quad-interleaved streams: one round of 256 groups prefetched, two for dense streams -- 20.2 -> ... ms at quality 90)
        if (layout_mask & 2u) {
            if (dense) MJX_IDCT(kMode420, 16, 1);
            else MJX_IDCT(kMode420, 8, 1);
        }
        if (layout_mask & 4u) MJX_IDCT(kMode420, 8, 2);
    }
    if (mode_mask & 4u) {
        if (layout_mask & 1u) MJX_IDCT(2, kPrefetch, 0);
        if (layout_mask & 2u) MJX_IDCT(2, 8, 1);
    }
#undef MJX_IDCT
}

void launch_planar_gather(hipStream_t st, uint32_t max_tiles, uint32_t nimg, const DevImage *images, uint32_t *entries,
                          uint32_t *tile_eoff, int32_t *dcbuf, uint32_t *img_flags, bool copy)
{
    // (copy == false: every multi-scan picture of the chunk is read without the gather -- k_planar_offsets only hands the scans' verdicts on)
    if (copy) hipLaunchKernelGGL(k_planar_count, dim3(max_tiles, nimg), dim3(256), 0, st, images, tile_eoff, img_flags);
    hipLaunchKernelGGL(k_planar_offsets, dim3(nimg), dim3(256), 0, st, images, tile_eoff, img_flags);
    if (copy) hipLaunchKernelGGL(k_planar_copy, dim3(max_tiles, nimg), dim3(256), 0, st, images, entries, tile_eoff, dcbuf, img_flags);
}

void launch_rgb_compare(hipStream_t st, uint32_t npairs, uint64_t max_bytes, const RgbPair *pairs, uint32_t *maxdiff,
                        unsigned long long *ndiff)
{
    const uint32_t gx = uint32_t((max_bytes + 16383) / 16384);
    if (gx == 0 || npairs == 0) return;
    for (uint32_t p0 = 0; p0 < npairs; p0 += 65535u) {                     // (a grid's y dimension holds 65 535 pairs)
        const uint32_t np = std::min(65535u, npairs - p0);
        hipLaunchKernelGGL(k_rgb_compare, dim3(gx, np), dim3(256), 0, st, pairs + p0, maxdiff + p0, ndiff + p0);
    }
}

void launch_ref_color(hipStream_t st, uint32_t max_pixel_wgs, uint32_t nimg, const DevImage *images,
                      const unsigned long long *planes, uint8_t *rgb, const uint32_t *img_flags)
{
    hipLaunchKernelGGL(k_ref_color, dim3(max_pixel_wgs, nimg), dim3(256), 0, st, images, planes, rgb, img_flags);
}

}   // namespace mjx
