// mjx_plan.cpp -- see mjx_plan.h.
#include "mjx_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace mjx {

const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

bool ref_get_indices(long x, long y, long max_x, long x_factor, long y_factor, long max_x_factor, long max_y_factor,
                     long *ox, long *oy)
{
    if (max_y_factor > 1 && y_factor == 1) {
        if (max_x_factor > 1 && x_factor == 1) {                 // decoder.rs:262-278, fitted to nbx = 94
            if ((y & 1) == 0) {
                if (((x / 2) & 1) == 1) { *ox = x / 2 - 1 + (x & 1); *oy = y + 1; }
                else { *ox = x / 2 + (x & 1); *oy = y; }
                return true;
            }
            if (y > 0 && ((x / 2) & 1) == 0) {
                if (max_x / 2 + x / 2 < 1) return false;
                *ox = max_x / 2 + x / 2 - 1 + (x & 1); *oy = y;
                return true;
            }
            if (y < 1) return false;
            *ox = max_x / 2 + x / 2 + (x & 1); *oy = y - 1;
            return true;
        }
        if ((y & 1) == 0) { *ox = x / 2; *oy = y + (x & 1); return true; }      // decoder.rs:279-285
        if (y < (x & 1)) return false;
        *ox = x / 2 + max_x / 2; *oy = y - (x & 1);
        return true;
    }
    *ox = x; *oy = y;                                            // decoder.rs:287
    return true;
}

// REF_COMPAT: geometry of decoder.rs:239-312 and the inputs on which that code panics: a block index
// past the component's blocks (:303), usize underflow in get_indices, or a store index past the plane where the
// guard `i + j*stride < len` (:370) disagrees with the index `i + j*stride*8` (:371).
static int plan_ref_layout(ImagePlan &p)
{
    const size_t W = p.width, H = p.height, len = W * H;
    p.nbx = (p.width + 7) / 8;
    p.nby = (p.height + 7) / 8;
    for (uint32_t c = 0; c < p.ncomp; c++) {
        const float x_i = std::ceil(float(p.width) * (float(p.h[c]) / float(p.hmax)));      // decoder.rs:239-246
        const float y_i = std::ceil(float(p.height) * (float(p.v[c]) / float(p.vmax)));
        const float xf = std::ceil(float(p.width) / x_i), yf = std::ceil(float(p.height) / y_i);
        if (!(xf >= 1.0f) || !(yf >= 1.0f)) return MJX_ERR_REF_PANIC;                       // division by zero at :290
        const size_t xs = size_t(xf), ys = size_t(yf);
        p.ref_xf[c] = uint32_t(xs);
        p.ref_yf[c] = uint32_t(ys);
        const size_t cols = p.nbx / xs, rows = p.nby / ys;
        const size_t nblk = size_t(p.nmcu) * p.h[c] * p.v[c];
        if (cols * rows > nblk) return MJX_ERR_REF_PANIC;                                   // component_blocks[block_i]
        for (size_t y = 0; y < rows; y++)
            for (size_t x = 0; x < cols; x++) {
                long bx, by;
                if (!ref_get_indices(long(x), long(y), long(p.nbx), long(xs), long(ys), long(p.hmax), long(p.vmax), &bx, &by))
                    return MJX_ERR_REF_PANIC;
                const size_t start_x = size_t(bx) * 8 * xs;
                if (W < start_x) continue;                                                  // :360 skips the line
                for (size_t line = 0; line < 8; line++) {
                    const size_t a = size_t(by) * 8 * ys * W + line * W + start_x, b = a + 8 * xs - 1;   // i range
                    for (size_t j = 1; j < ys; j++) {
                        // guard passes (i < len - jW) while the index i + 8jW is out of range (i >= len - 8jW)
                        const size_t lo = len > 8 * j * W ? len - 8 * j * W : 0, hi = len > j * W ? len - j * W : 0;
                        if (hi == 0) continue;
                        if (std::max(a, lo) <= std::min(b, hi - 1)) return MJX_ERR_REF_PANIC;
                    }
                }
            }
    }
    return MJX_OK;
}

// Cuts the scan of a planned picture into subsequences of about `base_bits` bits: p.seg holds the first bit of every
// segment (one segment without restart intervals); sets himg.sub_bits, himg.nsub and the first subsequence of every segment.
// (MJX_LONG_FIT=0: scans below kLongScanBits never take the long subsequences, as before round 6 -- for the A/B)
static bool long_fit_enabled()
{
    static const bool on = [] { const char *e = std::getenv("MJX_LONG_FIT"); return !e || std::atoi(e) != 0; }();
    return on;
}

void replan_subsequences(ImagePlan &p, uint32_t base_bits, bool allow_long)
{
    p.wg_lanes = uint32_t(kHuffWg);
    if (p.role == 2 || p.seg.size() < 2 * (size_t(p.nseg) + 1)) return;        // (role 2: no scan of its own)
    // long scans without restart intervals: long subsequences (mjx_huff.h: kLongSubseqBits), unless the caller asks for short ones
    uint32_t long_lanes = 0, long_bits = 0;
    if (allow_long && base_bits == uint32_t(kSubseqBits) && p.nseg == 1 && p.restart_mcus == 0 && (long long)p.himg.total_bits >= kLongScanBits)
        base_bits = uint32_t(kLongSubseqBits);
    else if (allow_long && base_bits == uint32_t(kSubseqBits) && p.role == 0 && p.nseg == 1 && p.restart_mcus == 0 && !p.stuffed && long_fit_enabled()) {
        // Round 6 (BASELINE config 4): a shorter scan whose long subsequences fill ONE workgroup of 256 lanes to seven
        // eighths takes them as well -- a 1080p scan of 0.24 MB is 238 of them in a 256-lane workgroup -- and with them the single decode (k_huff_emit runs at the chunk's workgroup size since this round; at
        // 512 lanes a 1080p picture held half an idle workgroup's LDS, which is why the long cut used to lose 13 % there).
        // (the length may stretch to 5/4 of the long one, as choose_subseq_bits stretches it to save a workgroup)
        // (256 lanes and up: what smaller scans do was settled on batches of small pictures in round 5 and is left alone)
        // ... and 256 lanes only: measured (tools/ab.sh -e MJX_LONG_FIT=0 -e MJX_LONG_FIT=1, same box) 4096 x 1080p 14.36 / 14.32 -> 13.84 / 14.08 ms per
        // step (592 -> 608 Gpixels/s); a 4K scan at quality 50 in ONE 512-lane workgroup of 500 long subsequences loses to its two passes
        // over 1000 short ones (21.5 -> 23.5 ms per 2048 pictures: k_huff_emit 10.5 ms against 2.0 + 5.6), so 512 lanes keep the old rule
        for (uint32_t lanes = 256u; lanes <= 256u && !long_lanes; lanes *= 2u) {
            const uint32_t fit = std::max<uint32_t>(uint32_t(kLongSubseqBits), ((p.himg.total_bits + lanes - 1) / lanes + uint32_t(kCpBits) - 1) / uint32_t(kCpBits) * uint32_t(kCpBits));
            if (fit <= uint32_t(kMaxSubseqBits) && uint64_t(p.himg.total_bits) * 8u >= uint64_t(lanes) * uint32_t(kLongSubseqBits) * 7u) { long_lanes = lanes; long_bits = fit; }
        }
        if (long_lanes) base_bits = uint32_t(kLongSubseqBits);
    }
    if (p.stuffed) {
        // the exact length and the restart offsets are only known on the device: an upper bound of the subsequence count
        // (every segment adds less than one to total / sub_bits) for the host's sizing, the subsequence length itself is final
        base_bits = std::max<uint32_t>(uint32_t(kCpBits), std::min<uint32_t>(base_bits, uint32_t(kLongSubseqBits)) / uint32_t(kCpBits) * uint32_t(kCpBits));
        p.himg.sub_bits = p.nseg == 1 ? choose_subseq_bits(p.himg.total_bits, base_bits) : base_bits;
        p.himg.nsub = (p.himg.total_bits + p.himg.sub_bits - 1) / p.himg.sub_bits + (p.nseg > 1 ? p.nseg : 0u);
        for (uint32_t g = 0; g <= p.nseg; g++) { p.seg[2 * size_t(g)] = g == p.nseg ? p.himg.nsub : 0u; p.seg[2 * size_t(g) + 1] = g == p.nseg ? p.himg.total_bits : 0u; }
        return;
    }
    base_bits = std::max<uint32_t>(uint32_t(kCpBits), std::min<uint32_t>(base_bits, uint32_t(kLongSubseqBits)) / uint32_t(kCpBits) * uint32_t(kCpBits));
    // A scan that does not fill one workgroup at the default length is cut shorter, so that it does: the workgroup holds its LDS for
    // as long as its longest lane decodes, whatever the number of lanes at work (the chroma scans of a three-scan 4K file: 272 and 219
    // subsequences of 4096 bits -- half a workgroup idle for the whole pass; round 5).
    // (MJX_FIT_SHORT=0: off, for the A/B; a value above 1: the shortest cut in bits instead of 1024)
    static const uint32_t fit_floor = [] { const char *e = std::getenv("MJX_FIT_SHORT"); const long v = e ? std::atol(e) : 1; return uint32_t(v <= 1 ? v * 4 * kCpBits : std::max<long>(v, kCpBits)); }();
    const bool fit_short = fit_floor != 0;
    p.wg_lanes = long_lanes ? long_lanes : uint32_t(kHuffWg);
    if (fit_short && base_bits == uint32_t(kSubseqBits) && p.nseg == 1 && p.restart_mcus == 0 && p.himg.total_bits < uint32_t(kSubseqBits) * uint32_t(kHuffWg) / 4u * 3u) {       // (under three quarters of a workgroup)
        // ... of 512 lanes, or -- the LDS of the counting and the write pass is sized per lane -- of 256 / 128, the fewest that
        // hold the scan at the default length: four times the workgroups per CU for the same scan (32768 x 256x256: 130 -> 176
        // Gpixels/s, 16384 x 500x375: 235 -> 284; `-DMJX_HUFF_WG=128` had measured it for the whole library)
        uint32_t lanes = uint32_t(kHuffWg);
        while (lanes > 128u && uint64_t(p.himg.total_bits) <= uint64_t(lanes / 2u) * uint32_t(kSubseqBits)) lanes /= 2u;
        p.wg_lanes = lanes;
        const uint32_t fit = ((p.himg.total_bits + lanes - 1) / lanes + uint32_t(kCpBits) - 1) / uint32_t(kCpBits) * uint32_t(kCpBits);
        base_bits = std::max<uint32_t>(fit_floor / uint32_t(kCpBits) * uint32_t(kCpBits), std::min(base_bits, fit));
    }
    const uint32_t top = base_bits * 5 / 4;
    auto bit0 = [&](uint32_t g) { return p.seg[2 * size_t(g) + 1]; };
    p.himg.sub_bits = long_bits ? long_bits : choose_subseq_bits(p.himg.total_bits, base_bits);
    if (p.nseg == 1 && p.restart_mcus == 0) {
        p.himg.nsub = (p.himg.total_bits + p.himg.sub_bits - 1) / p.himg.sub_bits;
        p.seg[0] = 0;
        p.seg[2] = p.himg.nsub;
        return;
    }
    auto count = [&](uint32_t bits) {
        uint32_t n = 0;
        for (uint32_t g = 0; g < p.nseg; g++) {
            const uint32_t len = bit0(g + 1) - bit0(g);
            n += len ? (len + bits - 1) / bits : 1u;
        }
        return n;
    };
    // the same workgroup-filling rule as choose_subseq_bits, on the segmented count
    uint32_t sub = count(p.himg.sub_bits);
    const uint32_t nwg = sub / uint32_t(kHuffWg);
    if (nwg > 0 && sub % uint32_t(kHuffWg) != 0)
        for (uint32_t bits = p.himg.sub_bits + kCpBits; bits <= top; bits += kCpBits)
            if (count(bits) <= nwg * uint32_t(kHuffWg)) { p.himg.sub_bits = bits; break; }
    sub = 0;
    for (uint32_t g = 0; g <= p.nseg; g++) {
        if (g > 0) {
            const uint32_t len = bit0(g) - bit0(g - 1);
            sub += len ? (len + p.himg.sub_bits - 1) / p.himg.sub_bits : 1u;
        }
        p.seg[2 * size_t(g)] = sub;
    }
    p.himg.nsub = sub;
}

int plan_image(const mjx_scan_desc &d, const mjx_opts &opts, ImagePlan &p, bool scan_part)
{
    p = ImagePlan{};
    auto fail = [&](int code) { p.status = code; return code; };
    const bool gather = d.n_parts != 0;            // the picture of a multi-scan file: geometry only, no scan of its own
    if (gather) {
        if (!d.parts || d.n_parts < 2 || d.n_parts > 3 || d.ncomp != 3) return fail(MJX_ERR_UNSUPPORTED_FORMAT);
        if (opts.layout == MJX_LAYOUT_REF_COMPAT || opts.strict_ref) return fail(MJX_ERR_UNSUPPORTED_FORMAT);   // the reference stops after scan 1
    } else if (!d.scan) {
        return fail(MJX_ERR_INVALID_ARG);
    }
    if (d.ncomp != 1 && d.ncomp != 3 && !(scan_part && d.ncomp == 2)) return fail(MJX_ERR_UNSUPPORTED_FORMAT);     // decoder.rs:328-330
    if (d.width == 0 || d.height == 0) return fail(MJX_ERR_REF_PANIC);             // x_factor division by zero
    // huffman.rs:127-128 preloads data[0..4] and panics on a shorter scan; the bug-compatible modes keep that.  Otherwise a
    // short scan (a flat 8x8 grey picture has one byte of entropy data) is decoded: past its end the lanes read the 0xAA
    // padding the reference itself reads there (huffman.rs:236-246).
    if (!gather && d.scan_len < ((opts.strict_ref || opts.layout == MJX_LAYOUT_REF_COMPAT) ? 4u : 1u)) return fail(MJX_ERR_TRUNCATED);
    if (d.scan_len >= (size_t(1) << 28)) return fail(MJX_ERR_UNSUPPORTED_FORMAT);  // bit positions are 32 bit
    p.width = d.width;
    p.height = d.height;
    p.ncomp = d.ncomp;
    p.layout = opts.layout;
    p.scan = d.scan;
    p.scan_len = d.scan_len;
    p.stuffed = d.scan_is_stuffed != 0 && !gather;
    for (uint32_t c = 0; c < p.ncomp; c++) {
        const mjx_comp &k = d.comp[c];
        if (k.h < 1 || k.h > 2 || k.v < 1 || k.v > 2) return fail(MJX_ERR_UNSUPPORTED_FORMAT);   // mod.rs:275-277
        if (k.tq > 3 || !(d.qt_present & (1u << k.tq))) return fail(MJX_ERR_MISSING_TABLE);      // decoder.rs:222-225
        if (!gather && (k.td > 3 || !(d.dc_present & (1u << k.td)))) return fail(MJX_ERR_MISSING_TABLE);      // decoder.rs:158-160
        if (!gather && (k.ta > 3 || !(d.ac_present & (1u << k.ta)))) return fail(MJX_ERR_MISSING_TABLE);      // decoder.rs:154-156
        p.h[c] = k.h;
        p.v[c] = k.v;
        p.tq[c] = k.tq;
    }
    // A single-component scan is non-interleaved in the standard: one block per MCU whatever SOF0 says.
    // The reference keeps h*v blocks per "MCU" (decoder.rs:200-201), which REF_COMPAT reproduces.
    if (p.ncomp == 1 && p.layout == MJX_LAYOUT_STANDARD) p.h[0] = p.v[0] = 1;
    p.hmax = p.vmax = 1;
    for (uint32_t c = 0; c < p.ncomp; c++) {
        if (p.h[c] > p.hmax) p.hmax = p.h[c];
        if (p.v[c] > p.vmax) p.vmax = p.v[c];
    }
    for (uint32_t c = 0; c < p.ncomp; c++)
        if (p.hmax % p.h[c] || p.vmax % p.v[c]) return fail(MJX_ERR_UNSUPPORTED_FORMAT);
    p.bpm = 0;
    for (uint32_t c = 0; c < p.ncomp; c++)
        for (uint32_t by = 0; by < p.v[c]; by++)
            for (uint32_t bx = 0; bx < p.h[c]; bx++) {
                p.blk_comp[p.bpm] = uint8_t(c);
                p.blk_bx[p.bpm] = uint8_t(bx);
                p.blk_by[p.bpm] = uint8_t(by);
                p.bpm++;
            }
    p.mcux = (p.width + 8 * p.hmax - 1) / (8 * p.hmax);
    p.mcuy = (p.height + 8 * p.vmax - 1) / (8 * p.vmax);
    if (p.layout == MJX_LAYOUT_REF_COMPAT) {
        const uint64_t nb = uint64_t((p.width + 7) / 8) * ((p.height + 7) / 8);    // decoder.rs:164-166
        const uint64_t f = uint64_t(p.hmax) * p.vmax;
        p.nmcu = uint32_t((nb + f - 1) / f);                                       // decoder.rs:191-192 (Q2)
    } else {
        p.nmcu = p.mcux * p.mcuy;
    }
    if (p.layout == MJX_LAYOUT_REF_COMPAT) {
        const int rc = plan_ref_layout(p);
        if (rc != MJX_OK) return fail(rc);
    }

    if (gather) {
        // no scan of its own: the kernels of the entropy stage find no subsequences here; stage B needs the geometry above,
        // the dequantisation multipliers below and the components' own block grids (what the scans actually carry)
        p.role = 2;
        std::memset(&p.himg, 0, sizeof p.himg);
        p.himg.bpm = p.bpm;
        p.himg.total_blocks = p.nmcu * p.bpm;
        p.himg.sub_bits = kSubseqBits;
        p.himg.cp_bits = uint32_t(kCpBits);
        p.nseg = 1;
        p.seg = {0u, 0u, 0u, 0u};
        for (uint32_t c = 0; c < p.ncomp; c++) {
            p.cbw[c] = ((p.width * p.h[c] + p.hmax - 1) / p.hmax + 7) / 8;
            p.cbh[c] = ((p.height * p.v[c] + p.vmax - 1) / p.vmax + 7) / 8;
        }
    }
    int dc_base[4] = {-1, -1, -1, -1}, ac_base[4] = {-1, -1, -1, -1};
    if (!gather) {
    // decode tables: each distinct (class, slot) used by the scan is built once.  Layout: the primary tables first, each
    // on a multiple of its size (lut_slot ORs the index into the base) -- an AC table of the second set with its pair part
    // right behind it --, then the sub-tables; a table's links are relative to its own primary table.
    // Two sets: the plain one for the write pass, and one whose AC tables have a pair part for the passes that only count
    // (mjx_huff.h); p.lut holds the first, then (from lut_plain_n on) the second.
    p.lut.clear();
    static thread_local LutEntry tmp[2 * kLutPrimarySize + 4096];
    int dc_base2[4] = {-1, -1, -1, -1}, ac_base2[4] = {-1, -1, -1, -1};
    auto build_set = [&](bool pair, std::vector<LutEntry> &out, int *dcb, int *acb) -> int {
        std::vector<std::vector<LutEntry>> built;
        std::vector<int> units;                              // primary-sized slots the table takes in front (1, or 2 with a pair part)
        int next_unit = 0;
        auto add_table = [&](const mjx_hufftab &t, bool is_dc) -> int {
            const bool with_pair = pair && !is_dc;
            const int n = build_decode_table(t.bits, t.vals, is_dc, tmp, int(sizeof tmp / sizeof tmp[0]), with_pair);
            if (n < 0) return n;
            built.emplace_back(tmp, tmp + n);
            units.push_back(with_pair ? 2 : 1);
            const int base = next_unit * kLutPrimarySize;
            next_unit += units.back();
            return base;
        };
        // (a slot that holds the same table as one built already shares it: the scans of a multi-scan file arrive with a slot per
        // component, and components that decode alike are what shortens the decoder's block-in-MCU state below)
        auto same_table = [](const mjx_hufftab &x, const mjx_hufftab &y) {
            size_t n = 0;
            for (int i = 0; i < 16; i++) n += x.bits[i];
            return std::memcmp(x.bits, y.bits, sizeof x.bits) == 0 && std::memcmp(x.vals, y.vals, std::min(n, sizeof x.vals)) == 0;
        };
        for (uint32_t c = 0; c < p.ncomp; c++) {
            const mjx_comp &k = d.comp[c];
            for (uint32_t e = 0; e < c && dcb[k.td] < 0; e++)
                if (same_table(d.dc[d.comp[e].td], d.dc[k.td])) dcb[k.td] = dcb[d.comp[e].td];
            if (dcb[k.td] < 0) {
                const int b = add_table(d.dc[k.td], true);
                if (b < 0) return b;
                dcb[k.td] = b;
            }
            for (uint32_t e = 0; e < c && acb[k.ta] < 0; e++)
                if (same_table(d.ac[d.comp[e].ta], d.ac[k.ta])) acb[k.ta] = acb[d.comp[e].ta];
            if (acb[k.ta] < 0) {
                const int b = add_table(d.ac[k.ta], false);
                if (b < 0) return b;
                acb[k.ta] = b;
            }
        }
        size_t subs = size_t(next_unit) * kLutPrimarySize;                            // where the next sub-table region goes
        out.assign(subs, 0);
        size_t own = 0;
        for (size_t k = 0; k < built.size(); k++) {
            const std::vector<LutEntry> &t = built[k];
            const size_t front = size_t(units[k]) * kLutPrimarySize;
            for (size_t i = 0; i < front; i++) {
                LutEntry e = t[i];
                if (i < size_t(kLutPrimarySize) && lut_is_link(e)) e = lut_link(unsigned(lut_link_offset(e) - front + subs - own), e & 15u);
                out[own + i] = e;
            }
            out.insert(out.end(), t.begin() + long(front), t.end());
            subs += t.size() - front;
            own += front;
        }
        // table offsets become 16-bit LDS addresses on the device; four tables of a baseline scan need < 24 KB
        if (out.size() * sizeof(LutEntry) > 0x7fff) return -MJX_ERR_BAD_HUFFMAN;
        while (out.size() % 4) out.push_back(0);                                      // 16-byte granules for staging
        return 0;
    };
    {
        std::vector<LutEntry> second;
        int rc1 = build_set(false, p.lut, dc_base, ac_base);
        if (rc1 == 0) rc1 = build_set(true, second, dc_base2, ac_base2);
        if (rc1 < 0) return fail(-rc1);
        p.lut_plain_n = uint32_t(p.lut.size());
        p.lut.insert(p.lut.end(), second.begin(), second.end());
    }
    std::memset(&p.himg, 0, sizeof p.himg);
    for (uint32_t b = 0; b < p.bpm; b++) {
        const mjx_comp &k = d.comp[p.blk_comp[b]];
        p.himg.btab[b].tabs = uint32_t(dc_base[k.td]) * uint32_t(sizeof(LutEntry)) | (uint32_t(ac_base[k.ta]) * uint32_t(sizeof(LutEntry)) << 16);
        p.himg.btab[b].next = b + 1 == p.bpm ? 0 : b + 1;
        p.himg.tabs_pair[b] = uint32_t(dc_base2[k.td]) * uint32_t(sizeof(LutEntry)) | (uint32_t(ac_base2[k.ta]) * uint32_t(sizeof(LutEntry)) << 16);
    }
    p.himg.bpm = p.bpm;
    {
        // The decoder's "block in MCU" state only selects tables.  Where the MCU's table sequence repeats with a shorter period --
        // Cb and Cr in a scan of their own share both tables: period 1 -- the state counts inside that period: two decodes that
        // agree on the bit position and the zig-zag index but disagree on which of the two chroma blocks they are in would
        // otherwise never be seen to coincide, and every subsequence of such a scan was re-decoded in one serial chain
        // (1024 copies of a 4K luma + chroma file: 41.7 ms of merge rounds; block indices come from counts, not from this state).
        uint32_t per = 1;
        for (; per < p.bpm; per++) {
            if (p.bpm % per) continue;
            bool rep = true;
            for (uint32_t b = per; b < p.bpm && rep; b++)
                rep = p.himg.btab[b].tabs == p.himg.btab[b - per].tabs && p.himg.tabs_pair[b] == p.himg.tabs_pair[b - per];
            if (rep) break;
        }
        for (uint32_t b = 0; b < per; b++) p.himg.btab[b].next = b + 1 == per ? 0 : b + 1;
        p.himg.bpm = per;
    }
    p.himg.cp_bits = uint32_t(kCpBits);            // (build_batch widens it for the pictures whose first decode emits)
    p.himg.warm_bits = 0;
    p.himg.total_bits = uint32_t(p.scan_len * 8);
    p.himg.total_blocks = p.nmcu * p.bpm;
    p.restart_mcus = d.restart_interval;
    p.seg.clear();
    if (p.restart_mcus == 0) {
        p.nseg = 1;
        p.seg = {0u, 0u, 0u, p.himg.total_bits};
    } else {
        // Each restart interval is decoded on its own: its first subsequence starts in a known state, subsequences do
        // not straddle intervals, and the DC predictors start again (T.81 E.2.4).  REF_COMPAT has no meaning here: the
        // reference panics on DRI (jpeg/mod.rs:424-428).
        if (p.layout == MJX_LAYOUT_REF_COMPAT) return fail(MJX_ERR_DRI_UNSUPPORTED);
        p.nseg = (p.nmcu + p.restart_mcus - 1) / p.restart_mcus;
        if (!p.stuffed && d.n_restart + 1 < p.nseg) return fail(MJX_ERR_TRUNCATED);     // fewer RSTn markers than intervals (a stuffed scan: the device counts them)
        for (uint32_t g = 0; g <= p.nseg && p.stuffed; g++) { p.seg.push_back(0u); p.seg.push_back(0u); }      // placeholder, see replan_subsequences
        for (uint32_t g = 0; g <= p.nseg && !p.stuffed; g++) {
            const uint64_t byte0 = g == 0 ? 0 : (g - 1 < d.n_restart ? d.restart_offsets[g - 1] : p.scan_len);
            if (byte0 > p.scan_len) return fail(MJX_ERR_INVALID_ARG);
            const uint32_t bit0 = g == p.nseg ? p.himg.total_bits : uint32_t(byte0 * 8);
            if (g > 0 && bit0 < p.seg[2 * g - 1]) return fail(MJX_ERR_INVALID_ARG);
            p.seg.push_back(0u);
            p.seg.push_back(bit0);
        }
    }
    replan_subsequences(p, uint32_t(kSubseqBits));

    }   // !gather

    // dequantisation x IDCT prescale, zig-zag order (reference: decoder.rs:230-232 multiplies by the raw table;
    // the AAN row/column factors and the 1/8 are folded in here so the kernel does one multiply per coefficient)
    double aan[8];
    for (int k = 0; k < 8; k++) aan[k] = k == 0 ? 1.0 : std::cos(k * 3.14159265358979323846 / 16.0) * std::sqrt(2.0);
    for (uint32_t c = 0; c < p.ncomp; c++)
        for (int k = 0; k < 64; k++) {
            const int nat = kZigZag[k];
            p.qmult[c][k] = float(double(d.qt[p.tq[c]][k]) * aan[nat >> 3] * aan[nat & 7] / 8.0);
        }
    return p.status;
}

void plan_input(const mjx_scan_desc &d, const mjx_opts &opts, std::vector<ImagePlan> &out)
{
    if (d.n_parts == 0) {
        out.emplace_back();
        plan_image(d, opts, out.back());
        return;
    }
    ImagePlan pic;
    plan_image(d, opts, pic);
    const size_t first = out.size();
    int bad = pic.status;
    uint32_t seen = 0;
    for (uint32_t k = 0; k < d.n_parts && bad == MJX_OK; k++) {
        const mjx_scan_part &part = d.parts[k];
        if (part.ncomp < 1 || part.ncomp > 2) { bad = MJX_ERR_UNSUPPORTED_FORMAT; break; }
        // the scan as a picture of its own: one component over the component's own block grid (non-interleaved order), or
        // two interleaved components on the MCU grid their sampling factors give
        mjx_scan_desc sub;
        std::memset(&sub, 0, sizeof sub);
        sub.scan = part.scan;
        sub.scan_len = part.scan_len;
        sub.ncomp = part.ncomp;
        uint32_t hm = 1, vm = 1;
        for (uint32_t q = 0; q < part.ncomp; q++) {
            const uint32_t c = part.comp[q];
            if (c > 2 || ((seen >> c) & 1)) { bad = MJX_ERR_UNSUPPORTED_FORMAT; break; }
            seen |= 1u << c;
            pic.src_part[c] = k;
            pic.src_comp[c] = q;
            hm = std::max(hm, pic.h[c]);
            vm = std::max(vm, pic.v[c]);
            sub.comp[q] = mjx_comp{d.comp[c].id, uint8_t(part.ncomp == 1 ? 1 : pic.h[c]), uint8_t(part.ncomp == 1 ? 1 : pic.v[c]),
                                   d.comp[c].tq, uint8_t(q), uint8_t(q)};
            sub.dc[q] = part.dc[q];
            sub.ac[q] = part.ac[q];
        }
        if (bad != MJX_OK) break;
        sub.width = uint16_t((uint32_t(d.width) * hm + pic.hmax - 1) / pic.hmax);
        sub.height = uint16_t((uint32_t(d.height) * vm + pic.vmax - 1) / pic.vmax);
        std::memcpy(sub.qt, d.qt, sizeof sub.qt);
        sub.qt_present = d.qt_present;
        sub.dc_present = sub.ac_present = uint8_t((1u << part.ncomp) - 1);
        sub.restart_interval = part.restart_interval;
        sub.n_restart = part.n_restart;
        sub.restart_offsets = part.restart_offsets;
        out.emplace_back();
        ImagePlan &pp = out.back();
        plan_image(sub, opts, pp, true);
        pp.role = 1;
        pp.part_idx = k;
        pp.nparts = d.n_parts;
        if (pp.status != MJX_OK) bad = pp.status;
    }
    if (bad == MJX_OK && seen != 7u) bad = MJX_ERR_UNSUPPORTED_FORMAT;
    pic.nparts = d.n_parts;
    if (bad != MJX_OK) {                       // one status for the whole picture
        out.resize(first);
        pic = ImagePlan{};
        pic.status = bad;
    }
    out.push_back(pic);
}

}   // namespace mjx
