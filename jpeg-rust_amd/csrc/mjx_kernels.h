// mjx_kernels.h -- device-side data layout shared by the kernel file and the host API.
#ifndef MJX_KERNELS_H
#define MJX_KERNELS_H

#include "mjx_huff.h"

namespace mjx {

constexpr int kWgLanes = 256;                                   // lanes of the scan / prefix-sum workgroups
constexpr int kDcSegMcus = 2048;                                // MCUs per DC-prediction segment (k_dc_sums / k_dc_apply)
constexpr uint32_t kPlanarKinds = 4, kPlanarSegs = 8;           // multi-scan pictures read without the gather: segment kinds per piece, segments per tile

// One image of a chunk, as the kernels see it (HBM, read-only during decode).
struct DevImage {
    HuffImage himg;
    uint64_t scan_off;      // bytes into the scan pool (256-byte aligned): the image's lane-interleaved region, see LaneBits
    uint64_t coef_off;      // blocks into the per-block arrays (dcbuf)
    uint64_t ent_off;       // entries into the compact coefficient stream pool (start of the image's region)
    uint64_t rgb_off;       // bytes into the RGB pool
    uint32_t scan_cols;     // columns of the region = subsequences rounded up to 8 (a row = scan_cols 16-byte pieces);
                            // rows = sub_bits / 128 + kLookPieces
    uint32_t lut_off;       // entries into the decode-table pool (multiple of 4): the plain table set (write pass)
    uint32_t lut_n;         // entries (multiple of 4)
    uint32_t lut2_off, lut2_n;   // the set with pair parts (counting passes: k_huff_spec, merge rounds), HuffImage::tabs_pair
    uint32_t sub_off;       // index of subsequence 0 in the per-subsequence arrays
    uint32_t qm_off;        // floats into the dequant-multiplier pool (3 x 64 per image)
    uint32_t width, height, mcux, mcuy, nmcu;
    uint32_t ncomp, bpm, hmax, vmax;
    uint32_t valid;         // 0: skip (error at plan time)
    uint32_t status_idx;    // index into the batch-wide device status array
    uint32_t log2_tile;     // stage-B tile = 1 << log2_tile MCUs (modes 0 and 2; mode 1: floor(log2(tile_mcus)), unused)
    uint32_t tile_mcus;     // MCUs per stage-B tile (mode 1: tile_mcus_420(), which need not be a power of two)
    uint32_t mode;          // stage-B specialisation: 0 generic, 1 = 4:2:0 (Y 2x2, Cb 1x1, Cr 1x1), 2 = REF_COMPAT placement
    uint32_t tile_off;      // index of the image's first tile offset in the tile_eoff array
    uint32_t tile_blocks;   // blocks per stage-B tile = tile_mcus * bpm  (<= 256)
    uint8_t blk_comp[kMaxBlocksPerMcu], blk_bx[kMaxBlocksPerMcu], blk_by[kMaxBlocksPerMcu];
    uint8_t ch[4], cv[4];   // sampling factors per component
    uint8_t cfirst[4];      // first block position of each component inside the MCU
    // REF_COMPAT layout only (mode 2): decoder.rs:239-250 replication factors, block grid, f32 plane scratch
    uint8_t ref_xf[4], ref_yf[4];
    uint32_t nbx, nby;
    uint64_t plane_off;     // 64-bit words into the plane scratch (ncomp planes of width*height words)
    // restart intervals: nseg segments of restart_mcus MCUs; (first subsequence, first bit) pairs + sentinel at seg_off
    uint32_t seg_off;       // index (in pairs) into the segment pool
    uint32_t nseg;          // 1 = no restart interval
    uint32_t restart_mcus;
    uint32_t ent_cap;       // entries the image's stream region holds
    uint32_t ent_rows;      // 0: the stream is linear, the lanes' runs packed behind one another (multi-scan pictures and their scans).
                            // > 0: quad-interleaved columns (see stream_phys): rows of 32 bytes every lane's column holds
    uint32_t ent_hdr;       // quad-interleaved: entries (4-byte words) at the head of the region that hold the subsequences' run
                            // lengths in groups, 16 bits each (stream_hdr_entries); the columns follow
    uint32_t n_rst_found;   // scans de-stuffed on the device: RSTn markers found (k_destuff_prefix)
    uint32_t upload_short;  // ... and their diagnosis at upload (a scan of nothing but stuffing, fewer RSTn markers than intervals): the
                            // picture is truncated whatever the block counts of a later decode say (k_huff_scan ORs it in)
    // multi-scan files: role 1 = one scan as a picture of its own (one component in raster order, or two
    // interleaved; entropy stage only; tile = one block, so tile_eoff holds an offset per block -- or, seg_S != 0, where the picture's
    // tile segments begin, see below), role 2 = the picture
    // (no scan; the k_planar_* kernels build its stream from the role-1 images: component c comes from the image
    // src_back[c] places before it in the image array, as that image's src_comp[c]-th component)
    uint32_t role;
    uint32_t src_back[3], src_comp[3];
    uint32_t wg_lanes;          // lanes of the entropy workgroups this scan was cut for (ImagePlan::wg_lanes; the chunk runs at its pictures' largest)
    uint32_t nparts;            // (role 1 as well: the scans of its file, of which it is the part_idx-th)
    uint32_t part_idx;
    uint32_t cbw[3], cbh[3];    // role 2: block grid of each component's own scan
    // Multi-scan pictures WITHOUT the gather (round 5, `planar`): stage B reads a tile's entries straight out of the scans' streams.
    // A tile of the picture is, in every scan, a few runs of consecutive blocks -- per MCU row the tile touches ("piece") and per
    // block row of the component inside the MCU row: a SEGMENT.  The write pass of a scan (role 1, seg_S != 0) records where every
    // segment's entries begin instead of an offset per block: a cut lies at every scan-MCU-row start and wherever a tile of the
    // picture begins (planar_cut); the table has seg_S slots per scan-MCU row (slot k = the k-th tile boundary inside the row).
    //   role 1: seg_S, seg_T (the picture's MCUs per tile), seg_mcux (the picture's MCUs per row), seg_hs / seg_vs (scan MCUs per
    //           picture MCU: the component's sampling factors for a one-component scan, 1 for an interleaved subset)
    //   role 2: planar = 1; pk_n kinds of segments per piece, per kind: the scan (pk_back images before the picture), its block
    //           row inside the MCU row (pk_v of pk_vs), scan MCUs per picture MCU (pk_hs), blocks of the segment per picture MCU
    //           (pk_u) and where these land in the picture's MCU (pk_map)
    uint32_t planar;
    uint32_t seg_S, seg_T, seg_mcux, seg_hs, seg_vs;
    uint8_t pk_n, pk_back[kPlanarKinds], pk_v[kPlanarKinds], pk_vs[kPlanarKinds], pk_hs[kPlanarKinds], pk_u[kPlanarKinds];
    uint8_t pk_map[kPlanarKinds][kMaxBlocksPerMcu];
    // single decode (round 5): 1 = the picture's first decode emits (k_huff_emit, then k_huff_prefix + k_block_gather instead of
    // k_huff_spec ... k_huff_write): pictures of one scan without restart intervals, quad-interleaved stream
    uint32_t emit;
    uint32_t emit_head;         // groups of head room in front of the first decode's entries (kEmitHeadGroups; tests shrink it)
};

// Where the segments of a scan begin (DevImage::seg_S): the first cut at or after scan MCU q0, as the scan MCU it lies at and its
// slot in the scan's table.  A cut lies at the start of every scan-MCU row and wherever a tile of the picture (seg_T picture MCUs
// in raster order) begins; a cut past the component's last real column is the next row's start (T.81 A.2.2: a one-component scan
// has no MCU padding blocks).  q0 at or past the scan's last MCU gives the sentinel slot, mcuy * seg_S.
struct PlanarCut { uint32_t mcu, slot; };
MJX_HD PlanarCut planar_cut(uint32_t q0, uint32_t scan_mcux, uint32_t S, uint32_t T, uint32_t pic_mcux, uint32_t hs, uint32_t vs)
{
    const uint32_t Rs = q0 / scan_mcux, Cs = q0 - Rs * scan_mcux;
    if (Cs == 0) return PlanarCut{q0, Rs * S};
    const uint32_t base = (Rs / vs) * pic_mcux;                  // the picture's first MCU in this MCU row
    const uint32_t t = (base + (Cs + hs - 1) / hs + T - 1) / T;  // the tile that begins at or after this column
    const uint32_t Cs2 = (t * T - base) * hs;
    if (Cs2 >= scan_mcux) return PlanarCut{(Rs + 1) * scan_mcux, (Rs + 1) * S};
    return PlanarCut{Rs * scan_mcux + Cs2, Rs * S + (t - base / T)};
}
// slots per scan-MCU row: the row's start + the tile boundaries inside a row of the picture
MJX_HD uint32_t planar_row_slots(uint32_t pic_mcux, uint32_t T) { return (pic_mcux + T - 1) / T + 1; }

// ---- compact coefficient stream, quad-interleaved (round 4) -----------------------------------------------------------
// A write-pass lane appends 32-byte groups (8 entries) to a stream of its own.  Packed one lane's run behind the other (the linear
// layout), every lane keeps a 128-byte line open over four flushes and a wave's flush touches 64 lines 3 KB apart: three fifths
// of the write pass went into these stores.  Here every subsequence owns a **column** of fixed capacity -- an
// entry takes at least 2 bits of scan, so sub_bits / 2 + 1 entries at most -- and four neighbouring columns are interleaved
// group by group: row r of quad q is one 128-byte line holding group r of subsequences 4q .. 4q + 3.  The four lanes flush
// within a step or two of each other, so the line is whole before it leaves L2 (k_huff_write 9.6 -> 8.05 ms per 2048 4K pictures
// for the stores alone), and a stage-B tile -- two or three lanes' worth of entries -- still reads consecutive lines.
//   entry j of subsequence s lies at  stream_phys(s, j, rows) = ((s >> 2) * rows + (j >> 3)) * 32 + (s & 3) * 8 + (j & 7)
//   a tile's start is recorded as the virtual index  s * (rows * 8) + j  (tile_eoff stays one 32-bit word per tile)
#ifndef MJX_STREAM_QUAD
#define MJX_STREAM_QUAD 4
#endif
constexpr uint32_t kStreamQuad = MJX_STREAM_QUAD;      // columns interleaved per row (a power of two; 4: a row is one 128-byte line)
// Single decode (round 5, k_huff_emit): a column also holds one word per block -- {DC difference, column index where the block's
// entries begin} -- filled from its top downwards (block word i at column index cap - 4 - 4 * (i >> 2) + (i & 3): 16-byte groups).
// A block takes at least two bits of scan and every entry two more, so entries and block words together are at most
// sub_bits / 2 + 2 words and never meet.  kEmitHeadGroups rows of head room on top of that: the first decode of a subsequence starts
// its entries at index kEmitHeadGroups * 8 (its block words at kEmitHeadWords), so that a prefix that is decoded again
// from the true entry state can be written RIGHT-ALIGNED against the part of the first decode that stays valid, whatever the
// difference between the two prefixes' counts (photographic content: <= 24 entries, <= 33 blocks; flat content, whose blocks take
// a handful of bits, reaches a hundred and more blocks per 1024 bits of prefix; more than the head room = the picture falls back
// to the two-pass kernels).
constexpr uint32_t kEmitHeadGroups = 64;                       // entries: 512 of head room
constexpr uint32_t kEmitHeadWords = kEmitHeadGroups * 4u;      // block words of head room: 256 (DevImage::emit_head scales both)
constexpr uint32_t kEmitExtraRows = kEmitHeadGroups + kEmitHeadWords / 8u + 2u;
// (rows of a column: what the entries of a subsequence can need, and -- only for pictures whose first decode emits -- the head room and
// the block words on top; a picture that cannot take the single-decode path does not pay for them: at 1024-bit subsequences the
// column would be 164 rows instead of 66.  Round-5 advisor.)
MJX_HD uint32_t stream_rows_base(uint32_t sub_bits) { return (sub_bits / 2u + 1u + 7u) / 8u + 1u; }
MJX_HD uint32_t stream_rows_for(uint32_t sub_bits, bool emit = true) { return stream_rows_base(sub_bits) + (emit ? kEmitExtraRows : 0u); }
MJX_HD uint64_t stream_quad_entries(uint32_t nsub, uint32_t rows) { return uint64_t((nsub + kStreamQuad - 1) / kStreamQuad) * rows * 8u * kStreamQuad; }
// Head of a picture's stream region: one 32-bit word per subsequence -- the run of its entries in the column, in store groups,
// and what is added to its entries' block labels:   first group [11:0] | end group [23:12] | label offset [31:24]
// (k_huff_write: first group 0, offset 0 -- its entries carry the block's index in the picture; k_huff_emit / k_huff_prefix: the
// labels count the blocks the lane has completed, and the offset is the low byte of what turns that into the block's index).
MJX_HD uint32_t stream_hdr_entries(uint32_t nsub) { return (nsub * 4u + 127u) / 128u * 32u; }      // whole 128-byte lines
MJX_HD constexpr uint32_t run_word(uint32_t g0, uint32_t g1, uint32_t label_off) { return (g0 & 0xfffu) | ((g1 & 0xfffu) << 12) | (label_off << 24); }
MJX_HD constexpr uint32_t run_first(uint32_t w) { return w & 0xfffu; }
MJX_HD constexpr uint32_t run_end(uint32_t w) { return (w >> 12) & 0xfffu; }
MJX_HD constexpr uint32_t run_label(uint32_t w) { return w >> 24; }
// column index of block word i (see above)
MJX_HD uint32_t block_word_index(uint32_t i, uint32_t rows) { return rows * 8u - 4u - 4u * (i >> 2) + (i & 3u); }
// per-subsequence record of the emitting decode (chunk array, index sub_off + s)
struct alignas(16) EmitSub {
    uint32_t d0n, d0m;      // blocks completed / entries produced by the first decode (k_huff_emit)
    uint32_t kfix;          // 0: its entry state was right; k + 1: the true path meets it at checkpoint k; kEmitAll: nowhere inside
    uint32_t bad;           // bit position (inside the subsequence) of the first symbol of the first decode that no code matches, or 0xffffffff
    uint32_t bad_lbl;       // ... and the label of its block
    int32_t lbl;            // label of the first decode's block at the merge point minus the true path's (k_huff_prefix)
    uint32_t pad_[2];
};
constexpr uint32_t kEmitAll = 0xffffu;
MJX_HD uint64_t stream_phys(uint32_t s, uint32_t j, uint32_t rows) { return (uint64_t(s / kStreamQuad) * rows + (j >> 3)) * (8u * kStreamQuad) + (s % kStreamQuad) * 8u + (j & 7u); }

// ---- reading a tile out of the quad-interleaved stream (stage B; also the host-side expansion and the CPU emulation of the tests) ----
// A tile's entries are store groups in the columns of a few neighbouring subsequences: from the tile's own start (subsequence
// s0, entry j0 -- its tile offset, split by the caller) to the end of s0's run, the whole runs of the subsequences between, and
// the head of s1's column up to the next tile's start.  The groups are numbered in that order; quad_cell() finds group o.  So
// that a lane of stage B finds its group without a loop over memory, quad_prepare() packs, once per tile, the cumulative group
// counts of the tile's first kQuadSegs subsequences into 16 bytes; tiles that span more (beyond quality ~97) take the loop over
// the run lengths at the head of the stream region.  (With a dependent LDS read per subsequence in the fetch -- which sits
// between the scatter phase and the barrier in front of the inverse DCT -- stage B took 0.3 ms more per 2048 pictures.)
#if defined(__HIP_DEVICE_COMPILE__)
#define MJX_UNROLL _Pragma("unroll")
#else
#define MJX_UNROLL
#endif
constexpr uint32_t kQuadSegs = 8;
// What the workgroup prepares per tile (32 bytes of LDS, two 16-byte reads): one word per subsequence of the tile, in order --
//   cumulative groups of the tile after this subsequence [13:0] | first group of the subsequence's run [23:14] | label offset [31:24]
// (the cumulative count of the last slot = the tile's groups; all ones there: more than kQuadSegs subsequences, the loop over the
// run words).  Round 5: the first group and the label offset are new -- a run no longer starts at row 0 of its column and its
// entries no longer carry the block's index in the picture (see run_word); packed here, once per tile, a lane still finds its
// group without a dependent read.
struct alignas(16) QuadCum { uint32_t w[kQuadSegs]; };
constexpr uint32_t kQuadCumMask = 0x3fffu, kQuadMany = 0x3fffu;
static_assert((kMaxSubseqBits / 2 + 8) / 8 + 1 + kEmitExtraRows < 1024, "a run's first group must fit ten bits (stream_rows_for(kMaxSubseqBits) < 1024)");
struct QuadView {
    const uint32_t *s_sub;               // LDS: subsequence of the workgroup's tile starts ...
    const uint16_t *s_at;                // ... and entry index in its column
    const QuadCum *s_cum;                 // LDS, per tile: see QuadCum
    const uint32_t *runs;                // the runs of all the picture's subsequences (run_word): the head of its stream region
    uint32_t rows, nsub;
};
struct QuadCell { uint32_t phys, k_lo, k_hi, label; };     // phys = 0xffffffff: no group for this lane; label: run_label of the group's subsequence
// What lane t of the workgroup prepares for tile t.
MJX_HD QuadCum quad_prepare(const QuadView &q, uint32_t k)
{
    const uint32_t s0 = q.s_sub[k], s1 = q.s_sub[k + 1], j0 = q.s_at[k], j1 = q.s_at[k + 1];
    QuadCum out;
MJX_UNROLL
    for (uint32_t i = 0; i < kQuadSegs; i++) out.w[i] = 0;
    if (s1 < s0 || s1 >= q.nsub) return out;                      // (offsets of a picture that did not decode: no entries, no reads)
    if (s1 - s0 >= kQuadSegs) { out.w[kQuadSegs - 1] = kQuadMany; return out; }
    uint32_t c = 0, run[kQuadSegs];
MJX_UNROLL
    for (uint32_t i = 0; i < kQuadSegs; i++) run[i] = q.runs[s0 + i < s1 ? s0 + i : s1];       // (all in flight together)
MJX_UNROLL
    for (uint32_t i = 0; i < kQuadSegs; i++) {
        const uint32_t s = s0 + i;
        if (s <= s1) {
            const uint32_t gs = i == 0 ? j0 >> 3 : run_first(run[i]);
            const uint32_t ge = s == s1 ? (j1 + 7u) >> 3 : run_end(run[i]);
            c += ge > gs ? ge - gs : 0u;
        }
        out.w[i] = (c < kQuadCumMask - 1u ? c : kQuadCumMask - 1u) | ((run_first(run[i]) & 0x3ffu) << 14) | (run_label(run[i]) << 24);
    }
    return out;
}
// Group o of tile k (of the workgroup): where it lies and which of its entries are the tile's.  Returns the tile's groups.
MJX_HD uint32_t quad_cell(const QuadView &q, uint32_t k, uint32_t o, QuadCell &cell)
{
    const uint32_t s0 = q.s_sub[k], s1 = q.s_sub[k + 1], j0 = q.s_at[k], j1 = q.s_at[k + 1];
    const QuadCum cw = q.s_cum[k];
    cell.phys = 0xffffffffu;
    cell.k_lo = 0;
    cell.k_hi = 8;
    cell.label = 0;
    uint32_t total, s = s0, first = j0 >> 3, before = 0;              // the group's subsequence, that one's first group, groups before it
    if ((cw.w[kQuadSegs - 1] & kQuadCumMask) != kQuadMany) {
        total = cw.w[kQuadSegs - 1] & kQuadCumMask;
        uint32_t sel = cw.w[0];
        bool own = true;                                                   // the group lies in the tile's first subsequence: it starts at j0
MJX_UNROLL
        for (uint32_t i = 0; i + 1 < kQuadSegs; i++) {
            const uint32_t ci = cw.w[i] & kQuadCumMask;
            const bool behind = o >= ci;                                   // (the counts stay at the total behind the tile's last subsequence)
            s += behind ? 1u : 0u;
            before = behind ? ci : before;
            sel = behind ? cw.w[i + 1] : sel;
            own = own && !behind;
        }
        first = own ? first : (sel >> 14) & 0x3ffu;
        cell.label = sel >> 24;
    } else {
        total = 0;
        bool found = false;
        for (uint32_t t = s0; t <= s1; t++) {
            const uint32_t rw = q.runs[t];
            const uint32_t gs = t == s0 ? j0 >> 3 : run_first(rw);
            const uint32_t ge = t == s1 ? (j1 + 7u) >> 3 : run_end(rw);
            const uint32_t cnt = ge > gs ? ge - gs : 0u;
            if (!found && o < total + cnt) { s = t; first = gs; before = total; found = true; cell.label = run_label(rw); }
            total += cnt;
        }
    }
    if (o < total) {
        const uint32_t at = (first + o - before) * 8u;
        cell.phys = uint32_t(stream_phys(s, at, q.rows));
        // (measurement builds only, garbage out: the tile's groups read from consecutive addresses; ... from the tile's own
        // first row on, so that tiles read different addresses; from three neighbouring columns in turn, what numbering the
        // groups across the columns would touch at best)
#if defined(MJX_EXP_QUAD_CONTIG)
        cell.phys = uint32_t(stream_phys(s0 & ~3u, 0, q.rows)) + o * 8u;
#elif defined(MJX_EXP_QUAD_CONTIG2)
        cell.phys = uint32_t(stream_phys(s0 & ~3u, (j0 >> 3) * 8u, q.rows)) + o * 8u;
#elif defined(MJX_EXP_QUAD_ZIP)
        cell.phys = uint32_t(stream_phys((s0 & ~3u) + o % 3u, (o / 3u) * 8u, q.rows));
#endif
        cell.k_lo = (o == 0u) ? j0 & 7u : 0u;
        cell.k_hi = (o == total - 1u && (j1 & 7u) != 0u) ? j1 & 7u : 8u;
    }
    return total;
}


// Lane-interleaved scan pool: pieces of 16 bytes, kLookPieces of look-ahead behind every subsequence (see LaneBits).
constexpr uint32_t kLookPieces = 4;
MJX_HD uint32_t scan_region_cols(uint32_t nsub) { return nsub ? (nsub + 7u) & ~7u : 0u; }
MJX_HD uint32_t scan_region_rows(uint32_t sub_bits) { return sub_bits / 128u + kLookPieces; }
MJX_HD uint64_t scan_region_bytes(uint32_t nsub, uint32_t sub_bits) { return uint64_t(scan_region_cols(nsub)) * scan_region_rows(sub_bits) * 16u; }
// One image of an upload for k_scan_interleave: where its linear de-stuffed scan lies in the staging buffer.
struct InterleaveImg {
    uint64_t lin_off;       // bytes into the linear staging buffer
    uint32_t lin_len;       // de-stuffed scan bytes
    uint32_t image;         // index into the DevImage array
};

// Device-side de-stuffing and marker scan (jpeg/mod.rs:371-385 on the GPU): one scan of an upload.
constexpr int kDestuffSeg = 16384;       // raw bytes per workgroup
struct DestuffImg {
    uint64_t raw_off, raw_len;           // stuffed bytes in the raw staging buffer (64-byte aligned, 64 bytes of padding behind)
    uint64_t out_off;                    // where the de-stuffed scan goes in the linear staging buffer
    uint32_t seg0, nseg;                 // this scan's slots in the per-segment count / base arrays (pairs: bytes kept, markers)
    uint32_t image;                      // its DevImage
    uint32_t ii_index;                   // its InterleaveImg (receives the de-stuffed length)
    uint32_t restarts;                   // 1: the picture has restart intervals -- RSTn markers leave the stream and are listed
    uint32_t rst0, rst_cap;              // the scan's slots in the marker list
    uint32_t direct;                     // 1: k_destuff_scatter writes the lane-interleaved region itself (scans without restart intervals,
                                         // round 5: no linear copy, no k_scan_interleave; the region was filled with 0xAA before); ii_index unused
    uint32_t pad_;
};

// mjx_batch_compare_rgb: one pair of pictures (device pointers: the pictures may live in different pools)
struct RgbPair { const uint8_t *a, *b; uint64_t bytes; };

// MCUs per stage-B tile: a power of two so that lane -> (MCU, strip) is a shift, and at most 256 strips per row.  At most
// MJX_GENERIC_TILE_BLOCKS blocks: the 4:2:0 kernel's 192, so that three workgroups fit a CU (round 5: a tile of 256 blocks -- 4:2:2,
// 64 MCUs -- held 69.6 KB of LDS, two workgroups per CU, and every picture of its batch was launched with that).
#ifndef MJX_GENERIC_TILE_BLOCKS
#define MJX_GENERIC_TILE_BLOCKS 192
#endif
inline uint32_t tile_mcus(uint32_t bpm, uint32_t hmax)
{
    uint32_t t = 1;
    while (t * 2 * bpm <= MJX_GENERIC_TILE_BLOCKS) t *= 2;
    const uint32_t cap = 256 / (2 * hmax);
    return t < cap ? t : cap;
}


uint32_t tile_mcus_420();       // MCUs per stage-B tile of the 4:2:0 kernel (mode 1)

// ---- launchers (mjx_kernels.hip); all asynchronous on `st` ---------------------------------------------
#if defined(__HIPCC__) || defined(MJX_WITH_HIP_RUNTIME)
size_t huff_lds_bytes(uint32_t lut_cap_entries);
size_t idct_lds_bytes(uint32_t max_tile_blocks);
int configure_kernels(size_t huff_lds, size_t idct_lds);
size_t huff_window_bytes();     // LDS the windowed entropy kernels need on top of huff_lds_bytes()
size_t huff_stage_bytes();      // ... and the write pass's entry rings
size_t huff_merge_bytes();      // LDS the merge kernels (rounds, straggler kernel, loop kernel) need on top of huff_lds_bytes()
uint32_t stream_group_entries();    // entries per store group of the write pass: a subsequence's run in the stream is rounded up to whole groups
// count -> prefix (+ geometry into `images`) -> scatter (+ marker list) -> segment tables of the pictures with restart intervals;
// segcount / segbase: two words per 16 KiB segment of every scan
void launch_destuff(hipStream_t st, uint32_t max_seg, uint32_t nimg, bool any_restarts, const DestuffImg *imgs, const uint8_t *raw,
                    uint32_t *segcount, uint32_t *segbase, uint8_t *pool, uint32_t *rst_off, DevImage *images, InterleaveImg *ii,
                    uint32_t *segs, uint32_t *img_flags, uint8_t *scan_pool /* the lane-interleaved pool: DestuffImg::direct scans go straight there */);
void launch_scan_interleave(hipStream_t st, uint32_t max_pieces, uint32_t nimg, const InterleaveImg *imgs, const DevImage *images,
                            const uint8_t *linear, uint8_t *pool, const uint32_t *segs);
void launch_huff_spec(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                      const uint8_t *scan_pool, const LutEntry *lut_pool, SubseqState *entry, SubseqState *exit_,
                      uint32_t *cps, const uint32_t *segs, uint32_t lanes /* per workgroup: kHuffWg, or 256 / 128 for a chunk of short scans */,
                      uint8_t *gen /* [chunk subsequences] cleared per subsequence (Gen2), or null */);
void launch_huff_merge(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                       const uint8_t *scan_pool, const LutEntry *lut_pool, SubseqState *entry, SubseqState *exit_,
                       uint32_t *cps, uint32_t *mismatches, uint32_t *items, uint32_t *item_count, const uint32_t *segs,
                       const uint32_t *prev_mismatches /* count of the round before, or null for the first */,
                       bool first_round /* most subsequences re-decode: the workgroups run their first slices in place */,
                       EmitSub *esub /* pictures whose first decode emits: the merge depth is kept here */,
                       uint32_t max_items /* subsequences of the chunk's longest scan - 1 (an upper bound will do) */,
                       uint8_t *gen /* [chunk subsequences]: which of the two sets of entry / exit / checkpoints is current (Gen2, mjx_kernels.hip) */,
                       uint32_t gen_stride /* subsequences between the two sets (a multiple of 256); 0: one set */);
// The merge rounds of a small chunk in one launch (device-wide barrier between rounds); `participants` = the workgroups (x, image)
// with x * merge_wg_lanes() + 1 < nsub(image): all of them must be resident at once (the caller checks against merge_loop_capacity()).
void launch_huff_merge_loop(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                            const uint8_t *scan_pool, const LutEntry *lut_pool, SubseqState *entry, SubseqState *exit_,
                            uint32_t *cps, uint32_t *verdict, const uint32_t *segs, uint32_t *ctl, uint32_t participants, uint32_t max_rounds,
                            uint32_t spin_limit,       // spin_limit: polls of a barrier (~1.5 us each) before a workgroup gives up
                            EmitSub *esub,
                            uint8_t *gen, uint32_t gen_stride /* as launch_huff_merge */);
// single decode (round 5): the first decode of a picture emits (k_huff_emit); after the merge rounds and the scan, k_huff_prefix
// re-decodes what lay in front of the merge point for the lanes whose entry was wrong and k_block_gather makes dcdiff / tile offsets
size_t huff_prefix_bytes();
void launch_emit_off(hipStream_t st, DevImage *images, uint32_t nimg, const uint32_t *img_flags);     // fall-back: the flagged pictures leave the single-decode path (device copies patched in place)
void launch_huff_emit(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                      const uint8_t *scan_pool, const LutEntry *lut_pool, SubseqState *entry, SubseqState *exit_, uint32_t *cps,
                      EmitSub *esub, uint32_t *entries, uint32_t lanes /* as launch_huff_spec */);
void launch_huff_prefix(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, const DevImage *images,
                        const uint8_t *scan_pool, const LutEntry *lut_pool, const SubseqState *entry, const SubseqState *exit_,
                        const uint32_t *cps, EmitSub *esub, const uint32_t *blkbase, uint32_t *entries, int *status, uint32_t *img_flags,
                        uint32_t *fallback /* device word: a picture had no head room for its prefix */, int16_t *dcdiff, uint32_t *tile_eoff,
                        const uint32_t *items, const uint32_t *item_count /* as written by k_huff_scan */,
                        uint32_t *unconverged /* device word, counted up by the first picture of the run that falls back */,
                        uint32_t lanes /* of the chunk's entropy workgroups (max_wg counts those) */);
void launch_huff_scan(hipStream_t st, uint32_t nimg, const DevImage *images, SubseqState *exit_, uint32_t *blkbase,
                      uint32_t *ebase, uint32_t *img_entries, uint32_t *img_flags, const uint32_t *segs,
                      const uint32_t *verdict /* device word: re-decodes of the last synchronisation round, or null */,
                      const EmitSub *esub, uint32_t *items /* [chunk subsequences][6]: per picture the (subsequence, checkpoint interval) pieces k_huff_prefix decodes again */,
                      uint32_t *item_count /* [chunk images] */, uint32_t *fallback /* device word: a picture goes to the two-pass kernels */,
                      uint32_t *unconverged /* device word, counted up when `verdict` is not zero */,
                      SubseqState *entry, uint8_t *gen, uint32_t gen_stride /* the rounds' second set is folded back into the first here */);
void launch_huff_write(hipStream_t st, uint32_t max_wg, uint32_t nimg, size_t tables_lds, size_t pad_lds, const DevImage *images,
                       const uint8_t *scan_pool, const LutEntry *lut_pool, const SubseqState *entry,
                       const uint32_t *blkbase, const uint32_t *ebase, uint32_t *entries, uint32_t *tile_eoff,
                       int16_t *dcdiff, int *status, const uint32_t *img_flags, const uint32_t *segs, const SubseqState *exit_,
                       uint32_t *cps /* measurement builds only */, uint32_t lanes /* as launch_huff_spec */);
void launch_dc_scan(hipStream_t st, uint32_t max_segs, uint32_t nimg, const DevImage *images, const int16_t *dcdiff, int32_t *dcbuf,
                    int32_t *segsum, const uint32_t *img_flags, uint32_t bpm_mask, uint32_t max_restart_segs,
                    uint32_t *segflag = nullptr, uint32_t gen = 0,
                    uint32_t *fail = nullptr /* device word, set when the one-pass kernel gave up waiting */, uint32_t spin_limit = 1u << 20,
                    bool fault = false /* test knob: a workgroup never publishes */);
void launch_idct_color(hipStream_t st, uint32_t max_tiles, uint32_t nimg, size_t lds, const DevImage *images,
                       const uint32_t *entries, const uint32_t *tile_eoff, const int32_t *dcbuf, const float *qmult,
                       uint8_t *rgb, uint32_t mode_mask, unsigned long long *planes, const uint32_t *img_flags,
                       bool dense /* the chunk's linear streams are dense (many entries per tile): deeper prefetch in the 4:2:0 kernel */,
                       uint32_t layout_mask /* bit 0: pictures with a linear stream, bit 1: with a quad-interleaved one */,
                       size_t lds_pad = 0 /* part of `lds` that is occupancy padding (a test knob): the 4:2:0 form sizes its own tile and adds it */);
// multi-scan pictures: component streams (raster order) -> the picture's stream in MCU order, tile offsets, DC values
void launch_planar_gather(hipStream_t st, uint32_t max_tiles, uint32_t nimg, const DevImage *images, uint32_t *entries,
                          uint32_t *tile_eoff, int32_t *dcbuf, uint32_t *img_flags, bool copy);
void launch_rgb_compare(hipStream_t st, uint32_t npairs, uint64_t max_bytes, const RgbPair *pairs, uint32_t *maxdiff,
                        unsigned long long *ndiff);
void launch_ref_color(hipStream_t st, uint32_t max_pixel_wgs, uint32_t nimg, const DevImage *images,
                      const unsigned long long *planes, uint8_t *rgb, const uint32_t *img_flags);
#endif

}   // namespace mjx
#endif
