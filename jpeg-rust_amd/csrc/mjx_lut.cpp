// mjx_lut.cpp -- host-side construction of the two-level Huffman decode tables.
//
// Replaces HuffmanTable::from_size_data_tables / make_code_table (reference src/jpeg/huffman.rs:37-98):
// canonical codes are assigned exactly as T.81 Figure C.2 does (code <<= 1 per length step, +1 per symbol);
// instead of a sorted Vec<HuffmanCode> searched linearly per length (huffman.rs:60-76, 211-227) the codes are
// expanded into a 2^9-entry primary table plus small sub-tables for longer codes, with the symbol's
// run / size already decoded and the reference's EOB / ZRL behaviour folded in (see mjx_huff.h).
#include "mjx.h"
#include "mjx_huff.h"

#include <cstring>

namespace mjx {

static LutEntry entry_for_symbol(unsigned len, uint8_t sym, bool is_dc)
{
    if (is_dc) {
        if (sym > 15) return lut_invalid();          // read_n_bits asserts n <= 16 (huffman.rs:202); 16 is unusable
#ifdef MJX_EXP_DCSTREAM                          // (measurement builds only: the DC difference as an entry of the stream)
        return lut_direct(len, 0, sym, false) | kLutCnt;
#else
        return lut_direct(len, 0, sym, false);
#endif
    }
    const unsigned r = sym >> 4, s = sym & 15;
    if (sym == 0x00) return lut_direct(len, 63, 0, true);  // EOB
    return lut_direct(len, r, s, true);              // includes ZRL (r = 15, s = 0) and the degenerate r/0 symbols
}

int build_decode_table(const uint8_t bits[16], const uint8_t *vals, bool is_dc, LutEntry *out, int cap, bool pair)
{
    // canonical code assignment
    struct Code { uint16_t code; uint8_t len, sym; };
    Code codes[256];
    int ncodes = 0;
    unsigned code = 0;
    for (int l = 1; l <= 16; l++) {
        for (int k = 0; k < bits[l - 1]; k++) {
            if (ncodes >= 256) return -MJX_ERR_BAD_HUFFMAN;
            if (code >= (1u << l)) return -MJX_ERR_BAD_HUFFMAN;          // over-subscribed
            codes[ncodes] = Code{uint16_t(code), uint8_t(l), vals[ncodes]};
            ncodes++;
            code++;
        }
        code <<= 1;
    }
    if (ncodes == 0) return -MJX_ERR_BAD_HUFFMAN;
    if (cap < kLutPrimarySize) return -MJX_ERR_NOMEM;
    for (int k = 0; k < kLutPrimarySize; k++) out[k] = lut_invalid();
    int used = kLutPrimarySize;
    if (pair) {                                           // the pair part lies right behind the primary part, filled in at the end
        if (cap < 2 * kLutPrimarySize) return -MJX_ERR_NOMEM;
        for (int k = 0; k < kLutPrimarySize; k++) out[kLutPrimarySize + k] = 0;
        used = 2 * kLutPrimarySize;
    }

    // longest code under each primary prefix that needs a sub-table
    uint8_t maxlen[kLutPrimarySize];
    std::memset(maxlen, 0, sizeof maxlen);
    for (int i = 0; i < ncodes; i++) {
        const Code &c = codes[i];
        if (c.len > kLutPrimaryBits) {
            const unsigned prefix = c.code >> (c.len - kLutPrimaryBits);
            if (c.len > maxlen[prefix]) maxlen[prefix] = c.len;
        }
    }
    for (int i = 0; i < ncodes; i++) {
        const Code &c = codes[i];
        const LutEntry e = entry_for_symbol(c.len, c.sym, is_dc);
        if (c.len <= kLutPrimaryBits) {
            const unsigned first = unsigned(c.code) << (kLutPrimaryBits - c.len), count = 1u << (kLutPrimaryBits - c.len);
            for (unsigned k = 0; k < count; k++) out[first + k] = e;
        } else {
            const unsigned prefix = c.code >> (c.len - kLutPrimaryBits);
            const unsigned nb = maxlen[prefix] - kLutPrimaryBits;        // 1..7
            if (!lut_is_link(out[prefix])) {
                if (used + (1 << nb) > cap) return -MJX_ERR_NOMEM;
                out[prefix] = lut_link(unsigned(used), nb);
                for (int k = 0; k < (1 << nb); k++) out[used + k] = lut_invalid();
                used += 1 << nb;
            }
            const unsigned sub = lut_link_offset(out[prefix]);
            const unsigned rest_len = c.len - kLutPrimaryBits;
            const unsigned rest = c.code & ((1u << rest_len) - 1);
            const unsigned first = rest << (nb - rest_len), count = 1u << (nb - rest_len);
            for (unsigned k = 0; k < count; k++) out[sub + first + k] = e;
        }
    }
    // Bit patterns that match no code: the primary table sends them through the link path (zero index bits) to one
    // shared invalid entry, so that only the rare second-level lookup has to test for bad codes.
    if (used + 1 > cap) return -MJX_ERR_NOMEM;
    const int bad_at = used++;
    out[bad_at] = lut_invalid();
    for (int k = 0; k < kLutPrimarySize; k++)
        if (out[k] == lut_invalid()) out[k] = lut_link(unsigned(bad_at), 0);
    if (pair && !is_dc) {
        for (unsigned i = 0; i < unsigned(kLutPrimarySize); i++) {
            const LutEntry e1 = out[i];
            if (lut_is_link(e1) || (e1 & kLutBad)) continue;
            const unsigned adv1 = e1 & 31u, zinc1 = (e1 >> 16) & 0x7fu;
            if (zinc1 >= 64u || adv1 >= unsigned(kLutPrimaryBits)) continue;     // end of block: the next symbol is a DC code; or no index bit left
            const unsigned rem = unsigned(kLutPrimaryBits) - adv1;
            const LutEntry e2 = out[(i << adv1) & unsigned(kLutPrimarySize - 1)];        // the bits behind the first symbol, zeros behind them
            if (lut_is_link(e2) || (e2 & kLutBad)) continue;
            const unsigned len2 = (e2 & 31u) - ((e2 >> 11) & 15u);
            if (len2 <= rem) out[kLutPrimarySize + i] = e2;                  // its code is all there: the symbol is certain
        }
    }
    return used;
}

}   // namespace mjx
