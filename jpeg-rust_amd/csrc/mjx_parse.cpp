// mjx_parse.cpp -- host-side JFIF marker walk (no GPU code).
//
// Replaces JPEGImage::parse (reference src/jpeg/mod.rs:202-465) up to the point where the
// reference constructs JPEGDecoder (mod.rs:388-413): it collects DQT/DHT tables, the SOF0 frame
// header and the SOS scan header, re-orders the component fields to scan order as
// JPEGDecoder::scan_header does (src/jpeg/decoder.rs:113-152), and de-stuffs the entropy-coded
// segment to end-of-file (mod.rs:371-385).  The reference's panics become status codes.
#include "mjx.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct ParseError {
    int code;
};

// Bounds-checked view over the file: a read past the end is the reference's slice-index panic.
class ByteView {
public:
    ByteView(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    size_t size() const { return n_; }
    uint8_t at(size_t i) const {
        if (i >= n_) throw ParseError{MJX_ERR_TRUNCATED};
        return p_[i];
    }
    unsigned be16(size_t i) const { return (unsigned(at(i)) << 8) | at(i + 1); }   // mod.rs:9-13 u8s_to_u16
    const uint8_t *span(size_t i, size_t len) const {
        if (len > n_ || i > n_ - len) throw ParseError{MJX_ERR_TRUNCATED};
        return p_ + i;
    }
private:
    const uint8_t *p_;
    size_t n_;
};

struct FrameComp { uint8_t id, h, v, tq; };   // mod.rs:104-113

struct ParserState {
    bool have_frame = false;
    unsigned width = 0, height = 0;
    std::vector<FrameComp> frame;
    unsigned restart_interval = 0;     // DRI (accepted unless strict_ref)
    // multi-scan files (one component per scan): what each scan contributed, assembled into one block at the end
    struct Part {
        std::vector<uint8_t> scan;
        std::vector<uint32_t> rst;
        uint8_t ncomp = 0, comp[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
        unsigned restart_interval = 0;
        mjx_hufftab dc[3], ac[3];
    };
    std::vector<Part> parts;
    // storage lent by the caller for the de-stuffed scan (mjx_parse_into); null: the parser allocates
    uint8_t *lent = nullptr;
    size_t lent_cap = 0;
    mjx_scan_desc *d;
};

// Marker classes of bytes_to_marker, mod.rs:157-181.
enum class Seg { SOI_EOI, DQT, SOF0, DHT, SOS, DRI, APP0, APP12_14, COM, OTHER_SKIPPABLE, OTHER_STANDALONE, OTHER_SOF };

Seg classify(uint8_t m)
{
    switch (m) {
    case 0xd8: case 0xd9: return Seg::SOI_EOI;
    case 0xdb: return Seg::DQT;
    case 0xc0: return Seg::SOF0;
    case 0xc4: return Seg::DHT;
    case 0xda: return Seg::SOS;
    case 0xdd: return Seg::DRI;
    case 0xe0: return Seg::APP0;
    case 0xec: case 0xee: return Seg::APP12_14;
    case 0xfe: return Seg::COM;
    default: break;
    }
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) return Seg::OTHER_STANDALONE;
    if (m >= 0xc1 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) return Seg::OTHER_SOF;
    return Seg::OTHER_SKIPPABLE;
}

void read_dqt(const ByteView &f, size_t pos, size_t len, mjx_scan_desc *d, bool strict)   // mod.rs:229-261
{
    size_t idx = pos;
    while (idx < pos + len) {
        const uint8_t pq = f.at(idx);
        const unsigned precision = pq >> 4, slot = pq & 15;
        if (precision > 1) throw ParseError{strict ? MJX_ERR_REF_PANIC : MJX_ERR_UNSUPPORTED_FORMAT};   // :258
        const size_t bytes = precision ? 128 : 64;
        const uint8_t *src = f.span(idx + 1, bytes);
        if (slot > 3) throw ParseError{MJX_ERR_REF_PANIC};                       // array index panic
        for (int k = 0; k < 64; k++)
            d->qt[slot][k] = precision ? uint16_t((src[2 * k] << 8) | src[2 * k + 1]) : src[k];
        d->qt_present |= uint8_t(1u << slot);
        idx += 1 + bytes;
    }
}

void read_sof0(const ByteView &f, size_t pos, ParserState &st, bool strict)               // mod.rs:262-298
{
    st.height = f.be16(pos + 1);
    st.width = f.be16(pos + 3);
    const unsigned n = f.at(pos + 5);
    st.frame.clear();
    size_t idx = pos + 6;
    for (unsigned c = 0; c < n; c++, idx += 3) {
        FrameComp fc{f.at(idx), uint8_t(f.at(idx + 1) >> 4), uint8_t(f.at(idx + 1) & 15), f.at(idx + 2)};
        if (fc.h < 1 || fc.h > 2 || fc.v < 1 || fc.v > 2)                        // asserts :275-277
            throw ParseError{strict ? MJX_ERR_REF_PANIC : MJX_ERR_UNSUPPORTED_FORMAT};
        st.frame.push_back(fc);
    }
    st.have_frame = true;
}

void read_dht(const ByteView &f, size_t pos, size_t len, mjx_scan_desc *d)                // mod.rs:299-336
{
    size_t idx = pos;
    const size_t end = pos + len;
    while (idx < end) {
        const uint8_t tc = f.at(idx++);
        const unsigned cls = tc >> 4, slot = tc & 15;
        const uint8_t *bits = f.span(idx, 16);
        idx += 16;
        size_t ncodes = 0;
        for (int k = 0; k < 16; k++) ncodes += bits[k];
        const uint8_t *vals = f.span(idx, ncodes);
        idx += ncodes;
        if (slot > 3) throw ParseError{MJX_ERR_REF_PANIC};
        if (ncodes == 0) throw ParseError{MJX_ERR_BAD_HUFFMAN};                  // huffman.rs:85 sizes[0] panics
        if (ncodes > 256) throw ParseError{MJX_ERR_BAD_HUFFMAN};
        mjx_hufftab &t = (cls == 0) ? d->dc[slot] : d->ac[slot];                 // :328 DC = 0, anything else AC
        std::memset(&t, 0, sizeof t);
        std::memcpy(t.bits, bits, 16);
        std::memcpy(t.vals, vals, ncodes);
        if (cls == 0) d->dc_present |= uint8_t(1u << slot); else d->ac_present |= uint8_t(1u << slot);
    }
}

// SOS header (mod.rs:337-362) + component re-ordering (decoder.rs:83-152) + de-stuffing (mod.rs:371-385).
// One scan of a multi-scan file (one component, or two interleaved): its entropy-coded segment runs up to the next marker
// that is not RSTn.  Returns the offset of that marker.
struct ScanCompSel { uint8_t id, td, ta; };
size_t read_part(const ByteView &f, size_t i, const ScanCompSel *sel, unsigned n, ParserState &st)
{
    const mjx_scan_desc *d = st.d;
    ParserState::Part part;
    part.ncomp = uint8_t(n);
    for (unsigned q = 0; q < n; q++) {
        int ci = -1;
        for (size_t c = 0; c < st.frame.size(); c++) if (st.frame[c].id == sel[q].id) ci = int(c);
        if (ci < 0 || sel[q].td > 3 || sel[q].ta > 3) throw ParseError{MJX_ERR_UNSUPPORTED_FORMAT};
        for (const ParserState::Part &p : st.parts)
            for (unsigned k = 0; k < p.ncomp; k++) if (p.comp[k] == ci) throw ParseError{MJX_ERR_UNSUPPORTED_FORMAT};   // twice
        for (unsigned k = 0; k < q; k++) if (part.comp[k] == ci) throw ParseError{MJX_ERR_UNSUPPORTED_FORMAT};
        if (!((d->dc_present >> sel[q].td) & 1) || !((d->ac_present >> sel[q].ta) & 1)) throw ParseError{MJX_ERR_MISSING_TABLE};
        part.comp[q] = uint8_t(ci);
        part.td[q] = sel[q].td;
        part.ta[q] = sel[q].ta;
        part.dc[q] = d->dc[sel[q].td];
        part.ac[q] = d->ac[sel[q].ta];
    }
    part.restart_interval = st.restart_interval;
    const size_t total = f.size();
    size_t k = i;
    while (k < total) {
        const uint8_t b = f.at(k);
        if (b != 0xff) { part.scan.push_back(b); k++; continue; }
        if (k + 1 >= total) { part.scan.push_back(b); k++; break; }
        const uint8_t m = f.at(k + 1);
        if (m == 0x00) { part.scan.push_back(0xff); k += 2; continue; }
        if (m == 0xff) { k++; continue; }                                         // fill byte (T.81 B.1.1.2)
        if ((m & 0xf8) == 0xd0 && part.restart_interval) {                        // RSTn: the next interval starts here
            part.rst.push_back(uint32_t(part.scan.size()));
            k += 2;
            continue;
        }
        break;                                                                    // the next marker segment
    }
    st.parts.push_back(std::move(part));
    return k;
}

// All scans of a multi-scan file are read: one block holds the parts array, the scans and the restart offsets.
void finish_parts(ParserState &st)
{
    mjx_scan_desc *d = st.d;
    const size_t n = st.parts.size();
    size_t covered = 0;
    for (const ParserState::Part &p : st.parts) covered += p.ncomp;
    if (covered != st.frame.size() || st.frame.size() != 3) throw ParseError{MJX_ERR_UNSUPPORTED_FORMAT};   // a component without a scan
    size_t bytes = (n * sizeof(mjx_scan_part) + 15) & ~size_t(15);
    for (const ParserState::Part &p : st.parts) bytes += ((p.scan.size() + 32 + 15) & ~size_t(15)) + ((p.rst.size() * 4 + 15) & ~size_t(15));
    uint8_t *buf = static_cast<uint8_t *>(std::malloc(bytes));
    if (!buf) throw ParseError{MJX_ERR_NOMEM};
    mjx_scan_part *parts = reinterpret_cast<mjx_scan_part *>(buf);
    size_t off = (n * sizeof(mjx_scan_part) + 15) & ~size_t(15);
    d->width = uint16_t(st.width);
    d->height = uint16_t(st.height);
    d->ncomp = uint8_t(st.frame.size());
    for (size_t k = 0; k < n; k++) {
        const ParserState::Part &p = st.parts[k];
        for (unsigned q = 0; q < p.ncomp; q++) {
            const FrameComp &fc = st.frame[p.comp[q]];
            d->comp[p.comp[q]] = mjx_comp{fc.id, fc.h, fc.v, fc.tq, p.td[q], p.ta[q]};
        }
        mjx_scan_part &o = parts[k];
        std::memset(&o, 0, sizeof o);
        uint8_t *sc = buf + off;
        if (!p.scan.empty()) std::memcpy(sc, p.scan.data(), p.scan.size());
        std::memset(sc + p.scan.size(), 0xaa, 32);                                // huffman.rs:236-246
        off += (p.scan.size() + 32 + 15) & ~size_t(15);
        uint32_t *rst = reinterpret_cast<uint32_t *>(buf + off);
        if (!p.rst.empty()) std::memcpy(rst, p.rst.data(), p.rst.size() * 4);
        off += (p.rst.size() * 4 + 15) & ~size_t(15);
        o.scan = sc;
        o.scan_len = p.scan.size();
        o.ncomp = p.ncomp;
        for (unsigned q = 0; q < p.ncomp; q++) { o.comp[q] = p.comp[q]; o.dc[q] = p.dc[q]; o.ac[q] = p.ac[q]; }
        o.restart_interval = uint16_t(p.restart_interval);
        o.n_restart = uint32_t(p.rst.size());
        o.restart_offsets = p.rst.empty() ? nullptr : rst;
    }
    d->scan = nullptr;
    d->scan_len = 0;
    d->n_parts = uint8_t(n);
    d->parts = parts;
    d->owner_ = buf;
}

// Returns 0 when the file's only scan has been read, else (multi-scan file) the offset of the marker behind this scan.
size_t read_sos(const ByteView &f, size_t pos, ParserState &st, bool strict, bool keep_stuffed)
{
    mjx_scan_desc *d = st.d;
    const unsigned n = f.at(pos);
    struct ScanComp { uint8_t id, td, ta; };
    std::vector<ScanComp> sc;
    size_t i = pos;
    for (unsigned c = 0; c < n; c++, i += 2) sc.push_back({f.at(i + 1), uint8_t(f.at(i + 2) >> 4), uint8_t(f.at(i + 2) & 15)});
    (void)f.at(i + 3);                                  // Ss, Se, AhAl are read (mod.rs:356-359) and ignored
    i += 4;
    if (!st.have_frame) throw ParseError{MJX_ERR_REF_PANIC};                     // mod.rs:388 unwrap
    // A scan that carries fewer components than the frame is one of several.  The reference decodes the
    // first one with the frame's sampling factors and stops (mod.rs:415-417); strict_ref keeps that.  Otherwise scans of
    // one component, or of two interleaved ones, are collected (read_part) until every component has been seen.
    if (!strict && (n < st.frame.size() || !st.parts.empty())) {
        if (n < 1 || n > 2 || st.frame.size() != 3) throw ParseError{MJX_ERR_UNSUPPORTED_FORMAT};
        ScanCompSel sel[2];
        for (unsigned q = 0; q < n; q++) sel[q] = ScanCompSel{sc[q].id, sc[q].td, sc[q].ta};
        return read_part(f, i, sel, n, st);
    }
    if (n != 1 && n != 3) throw ParseError{MJX_ERR_UNSUPPORTED_FORMAT};          // decoder.rs:328-330 panic!("asd")

    d->width = uint16_t(st.width);
    d->height = uint16_t(st.height);
    d->ncomp = uint8_t(n);
    for (unsigned c = 0; c < n; c++) {
        const FrameComp *fc = nullptr;
        for (const FrameComp &cand : st.frame) if (cand.id == sc[c].id) { fc = &cand; break; }
        if (!fc) throw ParseError{MJX_ERR_REF_PANIC};   // decoder.rs:128-138 inserts 0xff factors; decode then panics
        d->comp[c] = mjx_comp{sc[c].id, fc->h, fc->v, fc->tq, sc[c].td, sc[c].ta};
    }

    // everything after the SOS header, FF00 -> FF, markers (EOI...) kept verbatim -- except RSTn when a restart
    // interval is defined: those are taken out and the offset of the byte that follows is recorded
    const size_t total = f.size();
    const size_t remain = i < total ? total - i : 0;
    const uint8_t *src = remain ? f.span(i, remain) : nullptr;
    const bool restarts = st.restart_interval != 0;
    if (strict) keep_stuffed = false;                    // (strict_ref reports the reference's unguarded vec[i + 1]: that takes the byte pass)
    // keep_stuffed (opts.device_destuff): no pass over the entropy-coded bytes at all -- they are copied as they are, FF00
    // pairs and RSTn markers included; the device compacts them, lists the markers and works out the scan's length
    size_t max_rst = 0;
    for (size_t k = 0; restarts && !keep_stuffed && k + 1 < remain; k++) max_rst += (src[k] == 0xff && (src[k + 1] & 0xf8) == 0xd0);
    const size_t scan_room = (remain + 32 + 3) & ~size_t(3);
    const size_t need = scan_room + max_rst * sizeof(uint32_t) + 4;
    const bool lent = st.lent && need <= st.lent_cap;
    uint8_t *buf = lent ? st.lent : static_cast<uint8_t *>(std::malloc(need));
    if (!buf) throw ParseError{MJX_ERR_NOMEM};
    uint32_t *rst = reinterpret_cast<uint32_t *>(buf + scan_room);
    uint32_t nrst = 0;
    size_t w = 0;
    if (keep_stuffed) {                                  // the GPU compacts FF00 pairs (mjx_batch_create)
        if (remain) std::memcpy(buf, src, remain);
        w = remain;
        d->scan_is_stuffed = 1;
    }
    for (size_t k = 0; !keep_stuffed && k < remain;) {
        // runs without 0xFF are copied whole (memchr / memcpy: the byte-wise loop was the slowest part of the host side)
        const uint8_t *ff = static_cast<const uint8_t *>(std::memchr(src + k, 0xff, remain - k));
        const size_t run = ff ? size_t(ff - (src + k)) : remain - k;
        if (run) std::memcpy(buf + w, src + k, run);
        w += run;
        k += run;
        if (!ff) break;
        buf[w++] = 0xff;
        if (k + 1 >= remain) {                          // mod.rs:377 reads vec[i + 1] unguarded
            if (strict) { if (!lent) std::free(buf); throw ParseError{MJX_ERR_TRUNCATED}; }
            k += 1;
        } else if (src[k + 1] == 0x00) {
            k += 2;
        } else if (restarts && (src[k + 1] & 0xf8) == 0xd0) {
            w--;                                        // drop FF Dn; the next interval starts at the following byte
            k += 2;
            rst[nrst++] = uint32_t(w);
        } else {
            k += 1;
        }
    }
    std::memset(buf + w, 0xaa, 32);                     // huffman.rs:236-246: bytes past the end read as 0xaa
    d->scan = buf;
    d->scan_len = w;
    d->restart_interval = uint16_t(st.restart_interval);
    d->n_restart = nrst;
    d->restart_offsets = nrst ? rst : nullptr;
    d->owner_ = lent ? nullptr : buf;
    return 0;
}

void walk(const uint8_t *jpeg, size_t len, const mjx_opts &opts, mjx_scan_desc *out, uint8_t *lent, size_t lent_cap)
{
    const ByteView f(jpeg, len);
    const bool strict = opts.strict_ref != 0;
    ParserState st;
    st.d = out;
    st.lent = lent;
    st.lent_cap = lent_cap;
    size_t i = 0;
    while (i < len) {
        // bytes_to_marker, mod.rs:157-181 (including its "FF 00 xx" quirk, :161-164)
        if (f.at(i) != 0xff) throw ParseError{strict ? MJX_ERR_UNSUPPORTED_MARKER : MJX_ERR_UNSUPPORTED_FORMAT};
        uint8_t m = f.at(i + 1);
        if (m == 0) m = f.at(i + 2);
        const Seg seg = classify(m);
        if (seg == Seg::SOI_EOI) {                                                // :209-215
            if (m == 0xd9 && !st.parts.empty()) { finish_parts(st); return; }      // EOI behind the last scan of several
            i += 2;
            continue;
        }
        if (seg == Seg::OTHER_SKIPPABLE || seg == Seg::OTHER_STANDALONE || seg == Seg::OTHER_SOF) {
            if (strict) throw ParseError{MJX_ERR_UNSUPPORTED_MARKER};             // :456-462
            if (seg == Seg::OTHER_SOF) throw ParseError{MJX_ERR_UNSUPPORTED_FORMAT};
            if (seg == Seg::OTHER_STANDALONE) { i += 2; continue; }
        }
        const unsigned seglen = f.be16(i + 2);
        if (seglen < 2) throw ParseError{strict ? MJX_ERR_REF_PANIC : MJX_ERR_TRUNCATED};   // :219 u16 underflow
        const size_t body = seglen - 2;
        i += 4;
        switch (seg) {
        case Seg::DQT: read_dqt(f, i, body, out, strict); break;
        case Seg::SOF0: read_sof0(f, i, st, strict); break;
        case Seg::DHT: read_dht(f, i, body, out); break;
        case Seg::SOS: {
            const size_t next = read_sos(f, i, st, strict, opts.device_destuff == MJX_DESTUFF_DEVICE);
            if (!next) return;                                                    // :415-417 returns after the first scan
            i = next;
            continue;
        }
        case Seg::DRI:                                                             // :424-428 panics; T.81 B.2.4.4 otherwise
            if (strict) throw ParseError{MJX_ERR_DRI_UNSUPPORTED};
            if (body < 2) throw ParseError{MJX_ERR_TRUNCATED};
            st.restart_interval = f.be16(i);
            break;
        case Seg::APP0:                                                            // :429-443 absolute offsets up to vec[15]
            (void)f.span(i, 6);
            if (strict && len < 16) throw ParseError{MJX_ERR_TRUNCATED};
            break;
        case Seg::APP12_14:
            if (strict) throw ParseError{MJX_ERR_UNSUPPORTED_MARKER};             // :445-450
            break;
        case Seg::COM: (void)f.span(i, body); break;                              // :223-228
        default: break;
        }
        i += body;                                                                // :454
    }
    if (!st.parts.empty()) { finish_parts(st); return; }
    throw ParseError{MJX_ERR_NO_SCAN};
}

}   // namespace

namespace mjx {
// mjx_parse with storage for the de-stuffed scan lent by the caller (len + 64 bytes are always enough for a file without
// restart markers; when `cap` is too small the parser allocates as usual).  Used by mjx_decode_batch, which parses many
// files into one arena: hundreds of megabyte-sized malloc / free pairs cost more than the parsing.
int parse_into(const uint8_t *jpeg, size_t len, const mjx_opts *opts, mjx_scan_desc *out, uint8_t *storage, size_t cap);
}

extern "C" int mjx_parse(const uint8_t *jpeg, size_t len, const mjx_opts *opts, mjx_scan_desc *out)
{
    return mjx::parse_into(jpeg, len, opts, out, nullptr, 0);
}

int mjx::parse_into(const uint8_t *jpeg, size_t len, const mjx_opts *opts, mjx_scan_desc *out, uint8_t *storage, size_t cap)
{
    if (!out || (!jpeg && len)) return MJX_ERR_INVALID_ARG;
    mjx_opts o{};
    if (opts) o = *opts;
    std::memset(out, 0, sizeof *out);
    try {
        walk(jpeg, len, o, out, storage, cap);
    } catch (const ParseError &e) {
        std::free(out->owner_);
        std::memset(out, 0, sizeof *out);
        return e.code;
    } catch (...) {
        std::free(out->owner_);
        std::memset(out, 0, sizeof *out);
        return MJX_ERR_NOMEM;
    }
    return MJX_OK;
}

extern "C" void mjx_free_scan(mjx_scan_desc *desc)
{
    if (!desc) return;
    std::free(desc->owner_);
    desc->owner_ = nullptr;
    desc->scan = nullptr;
    desc->scan_len = 0;
}

extern "C" const char *mjx_strerror(int code)
{
    switch (code) {
    case MJX_OK: return "ok";
    case MJX_ERR_TRUNCATED: return "truncated input";
    case MJX_ERR_UNSUPPORTED_MARKER: return "marker not handled by the reference parser";
    case MJX_ERR_DRI_UNSUPPORTED: return "restart intervals are not supported";
    case MJX_ERR_BAD_HUFFMAN: return "invalid Huffman table or code";
    case MJX_ERR_REF_PANIC: return "input on which the reference panics";
    case MJX_ERR_DEVICE: return "HIP device error (no GPU, or kernels unavailable)";
    case MJX_ERR_UNSUPPORTED_FORMAT: return "unsupported JPEG format";
    case MJX_ERR_NO_SCAN: return "no scan found";
    case MJX_ERR_INVALID_ARG: return "invalid argument";
    case MJX_ERR_NOMEM: return "out of memory";
    case MJX_ERR_MISSING_TABLE: return "scan references an undefined table";
    default: return "unknown error";
    }
}
