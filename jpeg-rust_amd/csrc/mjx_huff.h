// mjx_huff.h -- per-lane baseline-JPEG entropy decode, shared by the HIP kernels (device) and
// the CPU emulation harness in tests/emul (host).  No wave/workgroup cooperation lives here:
// one call decodes one *subsequence* of the bitstream (512..640 bytes, fixed per image) from a given entry state.
//
// What it replaces in the reference (src/jpeg/huffman.rs): next_code (211-227) becomes a
// two-level table lookup, read_n_bits + value_correction (198-208, 256-268) become a shift and a
// branch-free EXTEND, next_block's EOB / ZRL / run clamps (164-189) are folded into the table so
// the symbol step is uniform:  pos = min(z + run, 63); coef[pos] = value; z = pos + 1   (kept as r = 64 - z, zinc = run + 1).
//   * EOB (0x00)  -> run = 63, size 0 : jumps to the end of the block (huffman.rs:164-169)
//   * ZRL (0xf0)  -> run = 15, size 0 : min(z+15,63)+1 == min(z+16,64)   (huffman.rs:170-175)
//   * r/s         -> run = r,  size s : min(r, 64-len-1) zeros then the value (huffman.rs:183-189)
//   * DC symbol s -> run = 0,  size s at z == 0 (huffman.rs:151-159)
#ifndef MJX_HUFF_H
#define MJX_HUFF_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MJX_HD __host__ __device__ __forceinline__
#else
#define MJX_HD inline
#endif

namespace mjx {

#ifndef MJX_SUBSEQ_BYTES
#define MJX_SUBSEQ_BYTES 512
#endif
constexpr int kSubseqBytes = MJX_SUBSEQ_BYTES;  // bytes of scan per lane, at least (HuffImage::sub_bits is the image's value)
constexpr int kSubseqBits = kSubseqBytes * 8;
// Scans of at least kLongScanBits (1.5 workgroups' worth of them) are cut into subsequences twice as long: the counting pass
// records and the merge rounds re-decode half as many boundaries, and a picture holds fewer workgroups' LDS for the same work --
// 2048 4K pictures of 0.94 MB take 26.3 instead of 27.7 ms per step with everything overlapped, 28.3 instead of 29.6 on one stream;
// 1080p pictures (0.24 MB, less than one workgroup of long subsequences) would lose 13 %.
constexpr int kLongSubseqBits = 2 * kSubseqBits;
constexpr int kMaxSubseqBits = kLongSubseqBits * 5 / 4;          // (the longest length choose_subseq_bits returns)
#ifndef MJX_CP_BITS
#define MJX_CP_BITS 256
#endif
constexpr int kCpBits = MJX_CP_BITS;           // bits between two checkpoints of a subsequence
#ifndef MJX_HUFF_WG
#define MJX_HUFF_WG 512
#endif
#ifndef MJX_MERGE_WG
#define MJX_MERGE_WG 512
#endif
constexpr int kMergeWg = MJX_MERGE_WG;                          // ... per k_huff_merge workgroup
#ifndef MJX_LONG_SCAN_HALF_WGS
#define MJX_LONG_SCAN_HALF_WGS 3      // scans of at least this many half workgroups' worth of long subsequences are cut into long ones
#endif
constexpr long long kLongScanBits = (long long)(MJX_LONG_SCAN_HALF_WGS) * MJX_HUFF_WG * kLongSubseqBits / 2;
constexpr int kHuffWg = MJX_HUFF_WG;                            // lanes (= subsequences) per k_huff_spec / merge / write workgroup:
                                                                // the decode tables in LDS are shared by kHuffWg / 64 waves

#ifndef MJX_LUT_BITS
#define MJX_LUT_BITS 9
#endif
constexpr int kLutPrimaryBits = MJX_LUT_BITS;
constexpr int kLutPrimarySize = 1 << kLutPrimaryBits;
constexpr int kMaxBlocksPerMcu = 12;           // 3 components x (2x2)

// ---- decode table entry (uint32) ---------------------------------------------------------------
//  direct : 0 [31] | bad [30] | zinc [22:16] | 0 [15] | size [14:11] | cnt [8] | adv [4:0]
//             adv  = code length + size: the bits the symbol consumes (1..27); size = value bits (0..11)
//             zinc = zig-zag positions the symbol advances: run + 1; 64 for EOB (saturates the block); 16 for ZRL
//             cnt  = 1 for an AC symbol with size != 0 (it produces one entry of the compact coefficient stream)
//             bad  = 1 for bit patterns no code matches (huffman.rs:156/162); such an entry consumes one bit and is
//                    only ever reached through a link
//           The fields are placed for the lane update (see LaneState and symbol_step; these kernels are bound by
//           vector-instruction issue, so the count matters):
//             x -= e & kLutXMask                     bits left in the current dword -= adv, stream entries += cnt
//             r  = sat_sub(r, e & kLutZincMask)      r is kept scaled by 2^16
//             value bits = bfe(window, -e, e >> 11)  offset (32 - adv) mod 32 and width size: the low five bits
//  link   : 1 [31] | byte offset of the sub-table from the table's base [19:4] | nbits [3:0]
// Pair part (AC tables of an image's second table set): kLutPrimarySize more entries right behind the primary part.  Entry i
// is the symbol that FOLLOWS the one primary entry i describes, when its code also lies wholly inside the kLutPrimaryBits
// index bits (first symbol's code and value bits, then the second symbol's code) -- else 0.  The passes that only count
// (no coefficient values: k_huff_spec, the merge rounds) then take two symbols per step where they can: a third of all steps
// on photographic content.  Never behind an end-of-block, a link or a bad entry.
typedef uint32_t LutEntry;
constexpr uint32_t kLutLink = 0x80000000u;
constexpr uint32_t kLutXMask = 0x0000011fu, kLutZincMask = 0x007f0000u;
constexpr uint32_t kLutBad = 1u << 30, kLutCnt = 1u << 8;
MJX_HD constexpr LutEntry lut_direct(unsigned len, unsigned run, unsigned size, bool is_ac)
{
    return ((run + 1u) << 16) | (size << 11) | ((is_ac && size) ? kLutCnt : 0u) | (len + size);
}
MJX_HD constexpr LutEntry lut_invalid() { return lut_direct(1, 0, 0, false) | kLutBad; }
MJX_HD constexpr LutEntry lut_link(unsigned offset_entries, unsigned nbits) { return kLutLink | ((offset_entries * 4u) << 4) | nbits; }
MJX_HD constexpr bool lut_is_link(LutEntry e) { return int32_t(e) < 0; }
MJX_HD constexpr unsigned lut_link_offset(LutEntry e) { return ((e >> 4) & 0xffffu) / 4u; }      // entries

// ---- per-subsequence synchronisation state -------------------------------------------------------------------
// The first 8 bytes (what the next subsequence must start from) are read while other lanes may rewrite them, so they
// form one naturally aligned 8-byte word; the counters are only consumed after the rounds have converged.
struct alignas(8) SubseqState {
    uint32_t p;    // bit position (relative to the image's scan) of the first symbol after the boundary
    uint8_t z;     // zig-zag index of the next coefficient (0 = next symbol is a DC code)
    uint8_t c;     // block index inside the MCU (selects the DC/AC table pair)
    uint16_t pad;
    uint32_t n;    // blocks completed inside the subsequence
    uint32_t m;    // non-zero AC coefficients (= entries of the compact coefficient stream) inside the subsequence
};
MJX_HD bool same_entry(const SubseqState &a, const SubseqState &b) { return a.p == b.p && a.z == b.z && a.c == b.c; }
MJX_HD SubseqState make_state(uint32_t p, uint32_t z, uint32_t c, uint32_t n = 0, uint32_t m = 0)
{
    SubseqState s;
    s.p = p; s.z = uint8_t(z); s.c = uint8_t(c); s.pad = 0; s.n = n; s.m = m;
    return s;
}

// Per-image constants the lane needs (lives in LDS on the device).
// The blocks of an MCU cycle through btab: when a block ends the lane fetches the entry of the next one (one 8-byte
// read) and has its table pair and the index of the block after it, with no comparisons.
struct BlockTab {
    uint32_t tabs;     // byte offset of the block's DC table | byte offset of its AC table << 16 (from the image's tables)
    uint32_t next;     // block-in-MCU of the following block
};
struct HuffImage {
    BlockTab btab[kMaxBlocksPerMcu];
    uint32_t tabs_pair[kMaxBlocksPerMcu];   // BlockTab::tabs for the image's second table set (AC tables with a pair part, see
                                         // symbol_step), which the passes that only count use
    uint32_t bpm;                        // blocks per MCU
    uint32_t total_bits;                 // scan_len * 8
    uint32_t total_blocks;               // MCUs to decode * bpm
    uint32_t nsub;                       // ceil(total_bits / sub_bits)
    uint32_t sub_bits;                   // bits per subsequence: a multiple of kCpBits, at most kMaxSubseqBits (kSubseqBits .. 5/4 of it, or kLongSubseqBits .. 5/4 of that; less in small batches)
    uint32_t cp_bits;                    // bits between two checkpoints of a subsequence: kCpBits, or a multiple of it for pictures whose
                                         // first decode emits (round 5, k_huff_emit: there every checkpoint recorded costs the emitting
                                         // pass, and only the few lanes whose entry was wrong ever use them)
    uint32_t warm_bits;                  // k_huff_emit: bits in front of its subsequence over which a lane warms up (a multiple of 32)
    uint32_t pad_[1];
};

// Subsequence length of an image.  512 bytes per lane is the sweet spot, but a workgroup's LDS (tables, windows,
// rings) is allocated for all its lanes: an image of 4.1 workgroups' worth of 512-byte subsequences would hold five
// workgroups' LDS, the fifth for a single wave's work.  When at most 25 % longer subsequences make the image fit into
// one workgroup less, they are chosen instead.
// `base`: the length to start from -- kSubseqBits, or less when a batch is too small to fill the device and shorter
// subsequences (more lanes, shorter serial chains) cut its latency (see replan_subsequences).
MJX_HD uint32_t choose_subseq_bits(uint32_t total_bits, uint32_t base = uint32_t(kSubseqBits))
{
    const uint32_t nsub = (total_bits + base - 1) / base, nwg = nsub / uint32_t(kHuffWg);
    if (nwg == 0 || nsub % uint32_t(kHuffWg) == 0) return base;
    const uint32_t lanes = nwg * uint32_t(kHuffWg);
    const uint32_t bits = ((total_bits + lanes - 1) / lanes + kCpBits - 1) / kCpBits * kCpBits;
    return (bits >= base && bits <= base * 5 / 4) ? bits : base;
}
static_assert(sizeof(HuffImage) % 16 == 0, "the decode tables follow HuffImage in LDS and are staged in 16-byte pieces");

// ---- compact coefficient stream -------------------------------------------------------------------------------
// One 32-bit entry per non-zero AC coefficient, in decode order:  value[15:0] | zig-zag position[21:16] | block[29:22]
// (low 8 bits of the block's index inside the image; bits 30-31 carry nothing and are not looked at).  DC differences
// go to a separate array, one per block.
MJX_HD constexpr uint32_t coef_entry(int val, uint32_t pos, uint32_t blk)
{
    return (uint32_t(val) & 0xffffu) | (pos << 16) | ((blk & 0xffu) << 22);
}

// A sink that discards everything (synchronisation passes).
struct NullSink {
    MJX_HD void dc(uint32_t, int) const {}                       // (block, value)
    MJX_HD void ac(uint32_t, uint32_t, int) const {}             // (block, r after the symbol -- scaled, see LaneState --, value)
    MJX_HD void block_done(uint32_t) const {}
    MJX_HD void bad_code(uint32_t, uint32_t) const {}        // (block, bit position of the symbol no code matches)
    MJX_HD void tick() const {}          // one call per decoded symbol (statistics in the CPU emulation)
    MJX_HD void flush_groups() const {}
    MJX_HD void flush_step(uint32_t) const {}
};

// ---- checkpoints: early merge of a re-decode with the path of the previous decode ----------------------
// While decoding a subsequence the lane records its state at the first symbol after every 256-bit boundary inside
// the subsequence (strictly after: a lane notices a boundary when it moves on to the next dword of the stream).  A
// later re-decode of the same subsequence (its entry changed) compares its own state at each boundary with the
// recorded one: equal (p, z, c) means the two decodes coincide from there on, so the re-decode stops and inherits
// the old exit.  This is the self-synchronisation property used at a finer grain: most re-decodes merge after ~100
// symbols instead of running all ~800.
//   word 0: bit31 valid | n[30:16] | block-in-MCU + 2 (mod blocks per MCU) [15:12] | 64 - z [11:5] | t [4:0]      word 1: m
//   (at a boundary the lane's dword position is the boundary itself, so t stands for p)
//   n, m = blocks / stream entries from the checkpoint to the end of the subsequence (after the decode's fix-up);
//   while a decode is running they temporarily hold the counts from the start to the checkpoint.
constexpr int kMaxCp = kMaxSubseqBits / kCpBits - 1;      // checkpoints of the longest subsequence
constexpr uint32_t kCpValid = 0x80000000u, kCpStateMask = 0x8000ffffu;
struct CpPair { uint32_t w, m; };                                // the two words of a checkpoint (one 8-byte access)
struct NoCheckpoints {
    MJX_HD uint32_t get(uint32_t) const { return 0; }             // word 0 of checkpoint k (may prefetch k+1)
    MJX_HD uint32_t get_m(uint32_t) const { return 0; }           // word 1 of checkpoint k (previous decode)
    MJX_HD CpPair get_pair(uint32_t) const { return CpPair{0, 0}; }   // both words of a checkpoint
    MJX_HD void set(uint32_t, uint32_t, uint32_t) const {}        // both words
};

// Registers of one lane's decoder.  The stream is seen through two big-endian dwords w0 w1; the lane's position is
//     p = 8 * (wn - 8) - t,      t = bits of w0 not yet consumed (0..31),      wn - 8 = byte offset of w1,
// so the next 32 bits of the stream are the funnel shift {w0,w1} >> t (the shifter reads the low five bits of x).
// t lives in the low byte of x and a down-counter of the stream entries in the upper 24 bits, so one subtraction of
// the masked table entry moves both; a low byte above 31 (the borrow went through it) = the lane has moved into w1
// and takes the next dword: x += 32 puts t back into 0..31 and returns the borrow.  That dword (at wn - 4) is read on
// every step together with the table lookup -- the bit source is an LDS window on the device, so the read is cheap
// but not free to wait for -- and is at hand when a refill needs it.  Everything that depends on p alone (end of the
// subsequence, checkpoints) is only looked at on that refill: boundaries are multiples of 32 bits.
constexpr uint32_t kLaneM0 = 0xffffffu;    // the entry counter starts here and counts down
constexpr uint32_t kRShift = 16, kRBlock = 64u << kRShift;
struct LaneState {
    uint32_t x;             // kLaneM0 - m [31:8] | t [7:0]
    uint32_t r;             // coefficients left in the current block (64 - zig-zag index), scaled by 2^16
    uint32_t n;             // blocks completed
    BlockTab nb;            // table entry of the block after the current one, fetched when the current block began
                            // (off the critical path); nb.next = block-in-MCU of the block after that
    uint32_t dcb, acb;      // byte offsets of the current block's DC / AC table (the DC table is used while r == 64)
    uint32_t wn;            // byte offset of the dword after w1, + 4
    uint32_t w0, w1;
};
MJX_HD uint32_t lane_t(const LaneState &st) { return st.x & 31u; }
MJX_HD uint32_t lane_m(const LaneState &st) { return kLaneM0 - (st.x >> 8); }
MJX_HD void lane_add_m(LaneState &st, uint32_t m) { st.x -= m << 8; }
MJX_HD uint32_t lane_pos(const LaneState &st) { return 8u * (st.wn - 8u) - lane_t(st); }
MJX_HD uint32_t lane_z(const LaneState &st) { return 64u - (st.r >> kRShift); }
MJX_HD uint32_t lane_c(const LaneState &st, const HuffImage &img)
{
    const uint32_t c1 = (st.nb.next ? st.nb.next : img.bpm) - 1u;     // the next block
    return (c1 ? c1 : img.bpm) - 1u;                                 // the current one
}
// byte offset of the first dword a lane fetches after it has left the bits below `bit` (bit rounded up to a dword)
MJX_HD uint32_t wn_after(uint32_t bit) { return 4u * ((bit + 31u) >> 5) + 12u; }

MJX_HD uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh)      // low 32 bits of {hi,lo} >> sh, sh in 0..31
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return uint32_t(((uint64_t(hi) << 32) | lo) >> (sh & 31));
#endif
}
MJX_HD uint32_t sat_sub(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;                                                      // (opaque to the compiler: it would turn the test
    asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));   // "difference == 0" into "a <= b" and a copy)
    return d;
#else
    return a > b ? a - b : 0u;
#endif
}
MJX_HD uint32_t bits_field(uint32_t v, uint32_t off, uint32_t width)   // width 0 -> 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return (width & 31u) ? (v >> (off & 31u)) & (0xffffffffu >> (32u - (width & 31u))) : 0u;
#endif
}
MJX_HD int32_t bits_field_signed(uint32_t v, uint32_t off, uint32_t width)   // sign-extended from the field's top bit; width 0 -> 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sbfe(int32_t(v), off, width);
#else
    const uint32_t f = bits_field(v, off, width);
    return (width & 31u) && (f >> ((width & 31u) - 1u)) ? int32_t(f) - int32_t(1u << (width & 31u)) : int32_t(f);
#endif
}
MJX_HD LutEntry lut_at(const LutEntry *lut, uint32_t byte_off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // On the device the table offsets a lane carries (BlockTab::tabs, patched by stage_tables) are absolute LDS
    // addresses, so the lookup address is one shift-add of the code bits.
    (void)lut;
    return *(const __attribute__((address_space(3))) LutEntry *)(byte_off);
#else
    return *reinterpret_cast<const LutEntry *>(reinterpret_cast<const unsigned char *>(lut) + byte_off);
#endif
}

// Slot of the primary table the next kLutPrimaryBits bits select.  The primary tables start on multiples of their
// size (mjx_plan.cpp lays the tables out that way; on the device the tables sit at the start of the workgroup's LDS,
// itself aligned), so the index is OR-ed in.
MJX_HD uint32_t lut_slot(uint32_t base, uint32_t w)
{
    return ((w >> (32 - kLutPrimaryBits - 2)) & uint32_t((kLutPrimarySize - 1) * 4)) | base;
}

//   BitSrc::be32(b)   -> big-endian dword at byte offset b of the image's scan (0xAAAAAAAA past the end,
//                        huffman.rs:236-246)
//   BitSrc::ahead(b)  -> the same for b = wn - 4, the dword a lane reads on every step (a source may keep a read
//                        pointer for it: advance() is called whenever wn moves on by 4)
template <class BitSrc>
MJX_HD void lane_begin(LaneState &st, const BitSrc &bits, const HuffImage &img, SubseqState entry)
{
    const uint32_t wi1 = (entry.p + 31u) >> 5;                   // dwords lying completely below p, rounded up
    st.wn = 4u * wi1 + 8u;
    st.x = (kLaneM0 << 8) | (32u * wi1 - entry.p);
    st.r = (64u - entry.z) << kRShift;
    st.n = 0;
    const BlockTab bt = img.btab[entry.c];
    st.nb = img.btab[bt.next];
    st.acb = bt.tabs >> 16;
    st.dcb = bt.tabs & 0xffffu;
    st.w0 = wi1 ? bits.be32(st.wn - 12u) : 0u;                   // p == 0: all of w0 is "consumed", nothing to load
    st.w1 = bits.be32(st.wn - 8u);
}

// (diagnostic builds only, -DMJX_STAMP: where a wave's cycles go inside the step; the default hook is empty)
struct NoStamp { MJX_HD void at(int) {} };

// One Huffman symbol: table lookup, EXTEND, coefficient placement, state update, window refill.
// Returns true when the lane moved on to the next dword of the stream (the caller then looks at lane_event).
template <bool WRITE, bool PAIR = false, class BitSrc, class Sink, class Stamp>
MJX_HD bool symbol_step(LaneState &st, BitSrc &bits, const LutEntry *lut, const HuffImage &img, uint32_t &blk,
                        Sink &sink, Stamp &sp);
template <bool WRITE, bool PAIR = false, class BitSrc, class Sink>
MJX_HD bool symbol_step(LaneState &st, BitSrc &bits, const LutEntry *lut, const HuffImage &img, uint32_t &blk,
                        Sink &sink)
{
    NoStamp sp;
    return symbol_step<WRITE, PAIR>(st, bits, lut, img, blk, sink, sp);
}
template <bool WRITE, bool PAIR, class BitSrc, class Sink, class Stamp>
MJX_HD bool symbol_step(LaneState &st, BitSrc &bits, const LutEntry *lut, const HuffImage &img, uint32_t &blk,
                        Sink &sink, Stamp &sp)
{
    static_assert(!(WRITE && PAIR), "the pair part carries no value bits: counting passes only");
    const uint32_t w = funnel(st.w0, st.w1, st.x);                                // next 32 bits of the stream
    const uint32_t ahead = bits.ahead(st.wn - 4u);                                // the dword after w1 (see LaneState)
    const bool is_dc = st.r == kRBlock;
    const uint32_t base = is_dc ? st.dcb : st.acb;                                // (tables are aligned: see lut_slot)
    const uint32_t slot = lut_slot(base, w);
    LutEntry e = lut_at(lut, slot);
    LutEntry q = PAIR ? lut_at(lut, slot + uint32_t(kLutPrimarySize * sizeof(LutEntry))) : 0u;     // (one read instruction for both: ds_read2st64_b32)
    sp.at(1);                                                                     // the primary entry is there
    if (lut_is_link(e)) {
        const uint32_t nb = e & 15u;
        e = lut_at(lut, base + bits_field(e, 4, 16) + bits_field(w, 32u - kLutPrimaryBits - nb, nb) * 4u);
        if (WRITE && (e & kLutBad)) sink.bad_code(blk, lane_pos(st));             // (invalid patterns always come this way)
    }
    sp.at(2);                                                                     // ... and the second-level one
    sink.tick();
    const uint32_t r_old = st.r;
    st.r = sat_sub(st.r, e & kLutZincMask);

    if (WRITE) {
        // EXTEND (T.81 F.2, huffman.rs:256-268) without a comparison: g = the value bits sign-extended from their leading
        // bit; flipping everything above them gives the value itself (leading bit 1) or value - 1 (leading bit 0)
        const uint32_t size = e >> 11;                                            // (the low five bits count)
        const int32_t g = bits_field_signed(w, 0u - e, size);
        const int32_t hflip = g ^ int32_t(0xffffffffu << (size & 31u));
        const int32_t val = hflip - (hflip >> 31);
#if defined(MJX_EXP_DCSTREAM)
        if (r_old == kRBlock) sink.dc(blk, val);
        if (e & kLutCnt) sink.ac(blk, st.r, val);
#else
        if (r_old == kRBlock) sink.dc(blk, val);
        else if (e & kLutCnt) sink.ac(blk, st.r, val);
#endif
    }
    sp.at(3);                                                                     // value, output
    st.x -= e & kLutXMask;
    if (PAIR) {
        // The symbol behind it as well -- unless this one was a DC code (its table has no pair part: what was read is not
        // one), has just ended the block (the next symbol is a DC code), or took the lane into the next dword: boundaries
        // (checkpoints, the end of the subsequence) are looked at when a lane moves on to a new dword, and a step that takes
        // two symbols must see them exactly where single steps would.
        q = (is_dc || st.r == 0 || (st.x & 0xe0u) != 0) ? 0u : q;
        st.r = sat_sub(st.r, q & kLutZincMask);
        st.x -= q & kLutXMask;
        if (q) sink.tick();
    }
    if (st.r == 0) {
        st.r = kRBlock;
        st.n++;
        blk++;
        if (WRITE) sink.block_done(blk);
        st.dcb = st.nb.tabs & 0xffffu;
        st.acb = st.nb.tabs >> 16;
#if defined(__HIP_DEVICE_COMPILE__)
        // (the table entry of the block after the next one is not needed before the next block ends: the read goes straight
        // into the lane's registers, last in this region, so that nothing in it waits for the LDS round trip)
        asm volatile("" : "+v"(st.dcb), "+v"(st.acb));
#endif
        st.nb = img.btab[st.nb.next];
    }
    sp.at(4);                                                                     // end of block
    if ((st.x & 0xffu) > 31u) {
        st.x += 32u;                                                              // t += 32, the borrow goes back
        st.w0 = st.w1;
        st.w1 = ahead;
        st.wn += 4;
        bits.advance();
        return true;
    }
    return false;
}

// What a lane looks at when it moves on to a new dword: the end of its subsequence and the checkpoints.
struct LaneEvents {
    uint32_t next_wn;       // st.wn at/after which the next event is due
    uint32_t end_wn;        // ... at/after which the lane has left its subsequence
    uint32_t k;             // index of the next checkpoint
    bool merged;            // the re-decode met the previous decode's path: the exit is the previous exit
    uint32_t cp_bytes;      // bytes of stream between two checkpoints (HuffImage::cp_bits / 8)
};
template <int CP>
MJX_HD void events_begin(LaneEvents &ev, uint32_t sub_start, uint32_t end_bit, uint32_t cp_bits = uint32_t(kCpBits))
{
    ev.end_wn = wn_after(end_bit);
    ev.k = 0;
    ev.merged = false;
    ev.cp_bytes = cp_bits / 8u;
    ev.next_wn = CP ? wn_after(sub_start + cp_bits) : ev.end_wn;
    if (ev.next_wn > ev.end_wn) ev.next_wn = ev.end_wn;
}
// Returns true when the lane is finished.  Otherwise (a checkpoint boundary was crossed): compares with the state the
// previous decode recorded there (CP == 2; equal = the decodes coincide from here on), or records the lane's state
// with the counts so far and moves to the next boundary.
MJX_HD uint32_t cp_state_word(const LaneState &st) { return lane_t(st) | (st.r >> (kRShift - 5)) | (st.nb.next << 12) | kCpValid; }
template <int CP, class CpStore>
MJX_HD bool lane_event(LaneState &st, LaneEvents &ev, const HuffImage &img, CpStore &cps)
{
    if (st.wn < ev.next_wn) return false;
    if (st.wn >= ev.end_wn) return true;
    if (CP) {
        const uint32_t state = cp_state_word(st);
        if (CP == 2) {
            const uint32_t old = cps.get(ev.k);
            if ((old & kCpStateMask) == state) {
                st.n += (old >> 16) & 0x7fffu;
                lane_add_m(st, cps.get_m(ev.k));
                ev.merged = true;
                return true;
            }
        }
        cps.set(ev.k, state | (st.n << 16), lane_m(st));
        ev.k++;
        ev.next_wn += ev.cp_bytes;
        if (ev.next_wn > ev.end_wn) ev.next_wn = ev.end_wn;
    }
    return false;
}
// The recording half of lane_event for decodes that never merge (CP == 1), for callers that test the boundaries
// themselves: the lane has reached ev.next_wn and is still inside its subsequence.
template <class CpStore>
MJX_HD void checkpoint_record(const LaneState &st, LaneEvents &ev, CpStore &cps)
{
    cps.set(ev.k, cp_state_word(st) | (st.n << 16), lane_m(st));
    ev.k++;
    ev.next_wn += ev.cp_bytes;
    if (ev.next_wn > ev.end_wn) ev.next_wn = ev.end_wn;
}
MJX_HD SubseqState lane_exit(const LaneState &st, const LaneEvents &ev, const HuffImage &img, const SubseqState &old_exit)
{
    if (ev.merged) return make_state(old_exit.p, old_exit.z, old_exit.c, st.n, lane_m(st));
    return make_state(lane_pos(st), lane_z(st), lane_c(st, img), st.n, lane_m(st));
}

// counts-so-far -> counts-to-the-end for the checkpoints this decode recorded (four at a time: the loads of a group
// are in flight together)
template <class CpStore>
MJX_HD void checkpoint_fixup(CpStore &cps, uint32_t k, uint32_t n_total, uint32_t m_total)
{
    for (uint32_t j0 = 0; j0 < k; j0 += 4) {
        CpPair v[4];
        for (uint32_t q = 0; q < 4; q++) v[q] = j0 + q < k ? cps.get_pair(j0 + q) : CpPair{0u, 0u};
        for (uint32_t q = 0; q < 4; q++)
            if (j0 + q < k)
                cps.set(j0 + q, (v[q].w & kCpStateMask) | ((n_total - ((v[q].w >> 16) & 0x7fffu)) << 16), m_total - v[q].m);
    }
}

// Decode the symbols of one subsequence: those that start in (sub_start, end_bit] (end_bit rounded up to a dword: the
// last subsequence of a scan may run up to 31 bits into the 0xAA padding the reference would read as well), from
// `entry` (the first of them).
//   lut              -> the image's decode tables
//   WRITE            -> emit coefficients for blocks < img.total_blocks through `sink`, starting at block `blk`
//   CP               -> 0: no checkpoints; 1: record checkpoints in `cps`; 2: record and merge with the previous
//                       decode of this subsequence (`sub_start` = its first bit, `old_exit` = that decode's exit)
template <bool WRITE, int CP, bool PAIR = false, class BitSrc, class Sink, class CpStore>
MJX_HD SubseqState decode_subseq(BitSrc bits, const LutEntry *lut, const HuffImage &img, SubseqState entry,
                                 uint32_t end_bit, uint32_t blk, Sink &sink, CpStore &cps, uint32_t sub_start,
                                 SubseqState old_exit)
{
    if (entry.p > end_bit) return make_state(entry.p, entry.z, entry.c);           // nothing starts inside
    LaneState st;
    LaneEvents ev;
    lane_begin(st, bits, img, entry);
    events_begin<CP>(ev, sub_start, end_bit, img.cp_bits);
    bool running = !(WRITE && blk >= img.total_blocks);
    while (running) {                               // (one back edge: a second one makes the compiler split the loop)
        const bool crossed = symbol_step<WRITE, PAIR>(st, bits, lut, img, blk, sink);
        bool done = WRITE && blk >= img.total_blocks;
        if (crossed) done = lane_event<CP>(st, ev, img, cps) || done;
        running = !done;
    }
    if (CP) checkpoint_fixup(cps, ev.k, st.n, lane_m(st));
    return lane_exit(st, ev, img, old_exit);
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host-side table construction (mjx_lut.cpp) ---------------------------------------------------
// Appends the two-level decode table for one DHT table to `out` (LutEntry words) and returns its size in
// entries, or a negative MJX_ERR_* code.  `is_dc`: symbols are DC size categories (run = 0).
// `pair`: with a pair part behind the primary part (AC tables of the second table set).
int build_decode_table(const uint8_t bits[16], const uint8_t *vals, bool is_dc, LutEntry *out, int cap, bool pair = false);
#endif

}   // namespace mjx
#endif
