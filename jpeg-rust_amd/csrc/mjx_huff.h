// mjx_huff.h -- per-lane baseline-JPEG entropy decode, shared by the HIP kernels (device) and
// the CPU emulation harness in tests/emul (host).  No wave/workgroup cooperation lives here:
// one call decodes one fixed-size *subsequence* of the bitstream from a given entry state.
//
// What it replaces in the reference (src/jpeg/huffman.rs): next_code (211-227) becomes a
// two-level table lookup, read_n_bits + value_correction (198-208, 256-268) become a shift and a
// branch-free EXTEND, next_block's EOB / ZRL / run clamps (164-189) are folded into the table so
// the symbol step is uniform:  pos = min(z + run, 63); coef[pos] = value; z = pos + 1.
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
constexpr int kSubseqBytes = MJX_SUBSEQ_BYTES;  // bytes of scan per lane
constexpr int kSubseqBits = kSubseqBytes * 8;
constexpr int kLutPrimaryBits = 10;
constexpr int kLutPrimarySize = 1 << kLutPrimaryBits;
constexpr int kMaxBlocksPerMcu = 12;           // 3 components x (2x2)

// ---- decode table entry (uint16) ---------------------------------------------------------------
//  direct : bit15 = 0 | size[14:11] | run[10:5] | len[4:0]      (len = total code length, 1..16)
//  link   : bit15 = 1 | sub-table offset / 2 [14:4] (entries, relative to the table base) | nbits[3:0]
//  invalid: 0
constexpr uint16_t kLutLinkBit = 0x8000;
MJX_HD constexpr uint16_t lut_direct(unsigned len, unsigned run, unsigned size)
{
    return uint16_t((size << 11) | (run << 5) | len);
}
MJX_HD constexpr uint16_t lut_link(unsigned offset, unsigned nbits) { return uint16_t(0x8000u | ((offset >> 1) << 4) | nbits); }
MJX_HD constexpr unsigned lut_link_offset(unsigned e) { return ((e >> 4) & 0x7ffu) << 1; }

// ---- per-subsequence synchronisation state -------------------------------------------------------------------
// The first 8 bytes (what the next subsequence must start from) are read while other lanes may rewrite them, so they
// form one naturally aligned 8-byte word; the counters are only consumed after the rounds have converged.
struct alignas(8) SubseqState {
    uint32_t p;    // bit position (relative to the image's scan) of the first symbol at/after the boundary
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
struct HuffImage {
    // The blocks of an MCU are ordered by component, so the table pair of block c follows from two thresholds:
    // c < cfirst1 -> component 0, c < cfirst2 -> component 1, else component 2.  All five words are uniform per image
    // (scalar registers on the device): no memory access when a block ends.
    uint32_t ctab[3];                    // per component: dc table base | ac table base << 16 (entry offsets)
    uint32_t cfirst1, cfirst2;           // first block-in-MCU of components 1 and 2 (= bpm when absent)
    uint32_t bpm;                        // blocks per MCU
    uint32_t total_bits;                 // scan_len * 8
    uint32_t total_blocks;               // MCUs to decode * bpm
    uint32_t nsub;                       // ceil(total_bits / kSubseqBits)
};

// ---- compact coefficient stream -------------------------------------------------------------------------------
// One 32-bit entry per non-zero AC coefficient, in decode order:  value[15:0] | zig-zag position[21:16] | block[29:22]
// (low 8 bits of the block's index inside the image).  DC differences go to a separate array, one per block.
MJX_HD constexpr uint32_t coef_entry(int val, uint32_t pos, uint32_t blk)
{
    return (uint32_t(val) & 0xffffu) | (pos << 16) | ((blk & 0xffu) << 22);
}

// A sink that discards everything (synchronisation passes).
struct NullSink {
    MJX_HD void dc(uint32_t, int) const {}
    MJX_HD void ac(uint32_t, unsigned, int) const {}
    MJX_HD void block_done(uint32_t) const {}
    MJX_HD void bad_code(uint32_t) const {}
    MJX_HD void tick() const {}          // one call per decoded symbol (statistics in the CPU emulation)
};

// ---- checkpoints: early merge of a re-decode with the path of the previous decode ----------------------
// While decoding a subsequence the lane records its state at the first symbol at/after every 256-bit boundary
// inside the subsequence.  A later re-decode of the same subsequence (its entry changed) compares its own state
// at each boundary with the recorded one: equal (p, z, c) means the two decodes coincide from there on, so the
// re-decode stops and inherits the old exit.  This is the self-synchronisation property used at a finer grain:
// most re-decodes merge after a few dozen symbols instead of running all ~180.
//   word 0: bit31 valid | n[30:16] | c[15:12] | z[11:6] | p - boundary [5:0]      word 1: m
//   n, m = blocks / stream entries from the checkpoint to the end of the subsequence (after the decode's fix-up);
//   while a decode is running they temporarily hold the counts from the start to the checkpoint.
constexpr int kCpBits = 256;
constexpr int kNumCp = kSubseqBits / kCpBits - 1;
constexpr uint32_t kCpValid = 0x80000000u, kCpStateMask = 0x8000ffffu;
struct NoCheckpoints {
    MJX_HD uint32_t get(uint32_t) const { return 0; }             // word 0 of checkpoint k (may prefetch k+1)
    MJX_HD uint32_t get_m(uint32_t) const { return 0; }           // word 1 of checkpoint k (previous decode)
    MJX_HD uint32_t get_plain(uint32_t) const { return 0; }       // word 0 of a checkpoint recorded by *this* decode
    MJX_HD uint32_t get_m_plain(uint32_t) const { return 0; }     // word 1 of a checkpoint recorded by *this* decode
    MJX_HD void set(uint32_t, uint32_t, uint32_t) const {}        // both words
};

// Registers of one lane's decoder: position + a three-dword look-ahead window of the big-endian bitstream.
// w2 is fetched one refill early and kept *raw* (BitSrc::raw32); it is only converted (BitSrc::fix, the byte swap of
// a little-endian load) when it moves into w1 at the next refill, so nothing consumes a global-memory load for a
// whole dword of symbols (~6) and its latency stays off the lane's critical path.
struct LaneState {
    uint32_t p, z, c, n, m;   // bit position, zig-zag index, block-in-MCU, blocks completed, stream entries produced
    uint32_t tab;             // table pair of block c (see HuffImage::ctab)
    uint32_t wi, o;           // dword index of w0, bit offset inside it
    uint32_t w0, w1, w2;
};

MJX_HD uint32_t block_tab(const HuffImage &img, uint32_t c)
{
    return c < img.cfirst1 ? img.ctab[0] : (c < img.cfirst2 ? img.ctab[1] : img.ctab[2]);
}

template <class BitSrc>
MJX_HD void lane_begin(LaneState &st, const BitSrc &bits, const HuffImage &img, SubseqState entry)
{
    st.p = entry.p; st.z = entry.z; st.c = entry.c; st.n = 0; st.m = 0;
    st.tab = block_tab(img, st.c);
    st.wi = st.p >> 5; st.o = st.p & 31;
    st.w0 = bits.be32(st.wi); st.w1 = bits.be32(st.wi + 1); st.w2 = bits.raw32(st.wi + 2);
}

// One Huffman symbol: table lookup, EXTEND, coefficient placement, state update, window refill.
template <bool WRITE, class BitSrc, class Sink>
MJX_HD void symbol_step(LaneState &st, const BitSrc &bits, const uint16_t *lut, const HuffImage &img, uint32_t &blk,
                        Sink &sink)
{
    const uint32_t w = st.o ? ((st.w0 << st.o) | (st.w1 >> (32 - st.o))) : st.w0;   // next 32 bits of the stream
    const uint32_t base = st.z ? (st.tab >> 16) : (st.tab & 0xffff);
    uint32_t e = lut[base + (w >> (32 - kLutPrimaryBits))];
    if (e & kLutLinkBit) {
        const uint32_t nb = e & 15, off = lut_link_offset(e);
        e = lut[base + off + ((w << kLutPrimaryBits) >> (32 - nb))];
    }
    uint32_t len = e & 31;
    const uint32_t run = (e >> 5) & 63, size = (e >> 11) & 15;
    sink.tick();
    if (len == 0) {                                                               // no code matches (huffman.rs:156/162)
        if (WRITE) sink.bad_code(blk);
        len = 1;
    }
    uint32_t pos = st.z + run;
    pos = pos > 63 ? 63 : pos;
    if (WRITE) {
        const uint32_t v = w << len;                                              // value bits, left aligned
        const uint32_t vb = (v >> 1) >> (31 - size);                              // size == 0 -> 0
        const int32_t val = int32_t(vb) - int32_t(((1u << size) - 1u) & ((v >> 31) - 1u));   // EXTEND, T.81 F.2
        if (st.z == 0) sink.dc(blk, val);
        else if (size) sink.ac(blk, pos, val);
    }
    st.m += (st.z != 0 && size != 0) ? 1u : 0u;
    st.z = pos + 1;
    if (st.z == 64) {
        st.z = 0;
        st.c = (st.c + 1 == img.bpm) ? 0 : st.c + 1;
        st.tab = block_tab(img, st.c);
        st.n++;
        blk++;
        if (WRITE) sink.block_done(blk);
    }
    const uint32_t adv = len + size;
    st.p += adv;
    st.o += adv;
    if (st.o >= 32) {
        st.o -= 32;
        st.wi++;
        st.w0 = st.w1;
        st.w1 = BitSrc::fix(st.w2);
        st.w2 = bits.raw32(st.wi + 2);
    }
}

// Checkpoint test at a symbol start.  Returns true when the lane's state equals the one recorded by the previous
// decode of this subsequence at the same 256-bit boundary (the decodes coincide from here on); otherwise records the
// state (with the blocks completed so far) and advances to the next boundary.
template <bool COMPARE, class CpStore>
MJX_HD bool checkpoint_merge(const LaneState &st, CpStore &cps, uint32_t &k, uint32_t &cp_bit, uint32_t &n_rest,
                             uint32_t &m_rest)
{
    const uint32_t state = (st.p - cp_bit) | (st.z << 6) | (st.c << 12) | kCpValid;
    if (COMPARE) {
        const uint32_t old = cps.get(k);
        if ((old & kCpStateMask) == state) {
            n_rest = (old >> 16) & 0x7fffu;
            m_rest = cps.get_m(k);
            return true;
        }
    }
    cps.set(k, state | (st.n << 16), st.m);
    k++;
    cp_bit += kCpBits;
    return false;
}

// counts-so-far -> counts-to-the-end for the checkpoints this decode recorded
template <class CpStore>
MJX_HD void checkpoint_fixup(CpStore &cps, uint32_t k, uint32_t n_total, uint32_t m_total)
{
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t wv = cps.get_plain(j);
        cps.set(j, (wv & kCpStateMask) | ((n_total - ((wv >> 16) & 0x7fffu)) << 16), m_total - cps.get_m_plain(j));
    }
}

// Decode from `entry` until the bit position reaches `end_bit`.
//   BitSrc::be32(i)  -> big-endian dword i of the image's scan (0xAAAAAAAA past the end, huffman.rs:236-246);
//                       raw32(i) / fix(raw) = the same in two steps (load, then byte order)
//   lut              -> the image's decode tables
//   WRITE            -> emit coefficients for blocks < img.total_blocks through `sink`, starting at block `blk`
//   CP               -> 0: no checkpoints; 1: record checkpoints in `cps`; 2: record and merge with the previous
//                       decode of this subsequence (`sub_start` = its first bit, `old_exit` = that decode's exit)
template <bool WRITE, int CP, class BitSrc, class Sink, class CpStore>
MJX_HD SubseqState decode_subseq(const BitSrc &bits, const uint16_t *lut, const HuffImage &img, SubseqState entry,
                                 uint32_t end_bit, uint32_t blk, Sink &sink, CpStore &cps, uint32_t sub_start,
                                 SubseqState old_exit)
{
    LaneState st;
    lane_begin(st, bits, img, entry);
    uint32_t cp_bit = sub_start + kCpBits, k = 0;
    while (st.p < end_bit) {
        if (WRITE && blk >= img.total_blocks) break;
        if (CP && st.p >= cp_bit) {
            uint32_t n_rest, m_rest;
            if (checkpoint_merge<CP == 2>(st, cps, k, cp_bit, n_rest, m_rest)) {
                st.n += n_rest;
                st.m += m_rest;
                st.p = old_exit.p; st.z = old_exit.z; st.c = old_exit.c;
                break;
            }
        }
        symbol_step<WRITE>(st, bits, lut, img, blk, sink);
    }
    if (CP) checkpoint_fixup(cps, k, st.n, st.m);
    return make_state(st.p, st.z, st.c, st.n, st.m);
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host-side table construction (mjx_lut.cpp) ---------------------------------------------------
// Appends the two-level decode table for one DHT table to `out` (uint16 entries) and returns its size in
// entries, or a negative MJX_ERR_* code.  `is_dc`: symbols are DC size categories (run = 0).
int build_decode_table(const uint8_t bits[16], const uint8_t *vals, bool is_dc, uint16_t *out, int cap);
#endif

}   // namespace mjx
#endif
