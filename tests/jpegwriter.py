"""Minimal baseline-JPEG *writer* for tests (test infrastructure, no reference counterpart: the reference only decodes).

Builds files the synthetic generator cannot: pictures from chosen coefficient blocks (the T1 per-block tier),
custom Huffman tables (every run/size symbol, including the degenerate `0x?0` ones of SURVEY Q9), and entropy-coded
segments that are syntactically valid symbol sequences but semantically corrupt (runs past the end of a block), on which
the reference does not panic but clamps (src/jpeg/huffman.rs:170-189).  Only markers the reference parses are written
(src/jpeg/mod.rs:166-179): SOI, DQT, SOF0, DHT, SOS, EOI.
"""
import struct

import numpy as np

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
          28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
          54, 47, 55, 62, 63]       # src/jpeg/decoder.rs:404-407


def huff_codes(bits, vals):
    """T.81 Fig. C.2 (src/jpeg/huffman.rs:80-98): symbol -> (code, length)."""
    codes, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            codes[vals[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return codes


def tables_from_jpeg(data):
    """DHT tables of a file: {(class, slot): (bits[16], vals)}."""
    out, i = {}, 2
    while i + 4 <= len(data) and data[i] == 0xff:
        m = data[i + 1]
        ln = struct.unpack(">H", data[i + 2:i + 4])[0]
        p = data[i + 4:i + 2 + ln]
        if m == 0xc4:
            j = 0
            while j < len(p):
                bits = list(p[j + 1:j + 17])
                nv = sum(bits)
                out[(p[j] >> 4, p[j] & 15)] = (bits, list(p[j + 17:j + 17 + nv]))
                j += 17 + nv
        if m == 0xda:
            break
        i += 2 + ln
    return out


class BitWriter:
    """MSB-first bit packer with FF -> FF00 byte stuffing (undone by src/jpeg/mod.rs:371-385)."""

    def __init__(self):
        self.out, self.acc, self.n, self.bits = bytearray(), 0, 0, 0

    def put(self, code, ln):
        if ln == 0:
            return
        self.acc = (self.acc << ln) | (code & ((1 << ln) - 1))
        self.n += ln
        self.bits += ln
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xff
            self.out.append(b)
            if b == 0xff:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)       # pad with ones (T.81 F.1.2.3)
        return bytes(self.out)


def magnitude(v):
    a = abs(int(v))
    s = a.bit_length()
    return s, (int(v) if v >= 0 else int(v) + (1 << s) - 1)


def encode_blocks(blocks, comp_of_block, dc_codes, ac_codes):
    """Valid entropy coding of `blocks` (int [n, 64], zig-zag order, absolute DC) in the given order; comp_of_block[k] selects
    the predictor and the code tables of block k."""
    w, pred = BitWriter(), {}
    for k, blk in enumerate(blocks):
        c = comp_of_block[k]
        s, bits = magnitude(int(blk[0]) - pred.get(c, 0))
        pred[c] = int(blk[0])
        w.put(*dc_codes[c][s])
        w.put(bits, s)
        run = 0
        last = max([i for i in range(1, 64) if blk[i]], default=0)
        for i in range(1, last + 1):
            if blk[i] == 0:
                run += 1
                continue
            while run > 15:
                w.put(*ac_codes[c][0xf0])
                run -= 16
            s, bits = magnitude(blk[i])
            w.put(*ac_codes[c][(run << 4) | s])
            w.put(bits, s)
            run = 0
        if last < 63:
            w.put(*ac_codes[c][0x00])
    return w.flush()


def write_jpeg(width, height, comps, qts, dhts, entropy, qt16=False):
    """comps: [(id, h, v, tq, td, ta)] in frame = scan order; qts: {slot: 64 values in zig-zag (file) order};
    dhts: {(class, slot): (bits, vals)}; entropy: the stuffed entropy-coded segment."""
    out = bytearray(b"\xff\xd8")
    for slot, q in sorted(qts.items()):
        if qt16:
            out += b"\xff\xdb" + struct.pack(">H", 2 + 129) + bytes([0x10 | slot]) + b"".join(struct.pack(">H", int(v)) for v in q)
        else:
            out += b"\xff\xdb" + struct.pack(">H", 2 + 65) + bytes([slot]) + bytes(int(v) for v in q)
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * len(comps), 8, height, width, len(comps))
    for cid, h, v, tq, _, _ in comps:
        out += bytes([cid, (h << 4) | v, tq])
    for (tc, th), (bits, vals) in sorted(dhts.items()):
        out += b"\xff\xc4" + struct.pack(">H", 2 + 17 + len(vals)) + bytes([(tc << 4) | th]) + bytes(bits) + bytes(vals)
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps))
    for cid, _, _, _, td, ta in comps:
        out += bytes([cid, (td << 4) | ta])
    out += bytes([0, 63, 0]) + entropy + b"\xff\xd9"
    return bytes(out)


def grey_jpeg_from_blocks(blocks_zz, blocks_x, qt_zz, tables, qt16=False):
    """A greyscale picture whose 8x8 blocks (raster order, blocks_x per row) carry exactly the coefficients given
    (int [n, 64] zig-zag order, absolute DC, before dequantisation)."""
    n = len(blocks_zz)
    assert n % blocks_x == 0
    dc = {0: huff_codes(*tables[(0, 0)])}
    ac = {0: huff_codes(*tables[(1, 0)])}
    ent = encode_blocks(blocks_zz, [0] * n, dc, ac)
    return write_jpeg(blocks_x * 8, (n // blocks_x) * 8, [(1, 1, 1, 0, 0, 0)], {0: qt_zz},
                      {(0, 0): tables[(0, 0)], (1, 0): tables[(1, 0)]}, ent, qt16=qt16)


# ---- syntactically valid, semantically corrupt streams ---------------------------------------------------
def full_ac_table():
    """An AC table that holds all 256 run/size symbols -- including the degenerate `0x?0` ones (r zeros then a 0,
    src/jpeg/huffman.rs:176-189) that the Annex-K tables lack: 128 codes of 8 bits, 128 of 9 bits (no 1-bit code).
    Symbol order: a fixed permutation, so that frequent and rare symbols mix over both lengths."""
    rng = np.random.default_rng(256)
    vals = [int(v) for v in rng.permutation(256)]
    bits = [0] * 16
    bits[7], bits[8] = 128, 128
    return bits, vals


def small_dc_table(max_size=8):
    """DC sizes 0..max_size with 4-bit codes (sums of differences stay far inside i16 on small pictures)."""
    bits = [0] * 16
    bits[3] = max_size + 1
    return bits, list(range(max_size + 1))


def random_symbol_stream(rng, nsymbols, dc_tab, ac_tab, p_dc=0.12, max_size=15):
    """Random *valid codes* with random value bits, written with no regard to block structure: DC-table codes are only
    valid where the decoder expects them, so the stream is produced by simulating the decoder's table choice --
    src/jpeg/huffman.rs:146-195 semantics: first symbol of a block from the DC table, then AC symbols until 64 coefficients
    are produced (EOB fills up; ZRL and runs are clamped, Q9).  Returns the stuffed bytes and the number of whole blocks."""
    dcc, acc = huff_codes(*dc_tab), huff_codes(*ac_tab)
    dc_syms, ac_syms = [s for s in dcc if s <= max_size], [s for s in acc if (s & 15) <= max_size]    # (small values: the
    w, blocks, z = BitWriter(), 0, 0                               # samples stay in range and the RGB comparison means something)
    for _ in range(nsymbols):
        if z == 0:
            s = dc_syms[int(rng.integers(len(dc_syms)))]
            w.put(*dcc[s])
            w.put(int(rng.integers(1 << s)) if s else 0, s)
            z = 1
            continue
        if rng.random() < p_dc:
            sym = 0x00                                   # EOB now and then, so that blocks also end regularly
        else:
            sym = ac_syms[int(rng.integers(len(ac_syms)))]
        w.put(*acc[sym])
        r, s = sym >> 4, sym & 15
        if sym == 0x00:
            z = 64
        elif sym == 0xf0:
            z = min(z + 16, 64)
        else:
            w.put(int(rng.integers(1 << s)) if s else 0, s)
            z = min(z + r, 63) + 1
        if z >= 64:
            z = 0
            blocks += 1
    return w.flush(), blocks


# ---- non-interleaved twins of interleaved files, without Pillow (tests/golden/make_multiscan.py is the checked version) ---
def _fast_encode_blocks(blocks, dc_codes, ac_codes):
    """encode_blocks for one component, touching only the non-zero coefficients (a 4K picture has 200 000 blocks)."""
    w, pred = BitWriter(), 0
    nzr, nzc = np.nonzero(blocks[:, 1:])
    starts = np.searchsorted(nzr, np.arange(blocks.shape[0] + 1))
    for k in range(blocks.shape[0]):
        blk = blocks[k]
        s, bits = magnitude(int(blk[0]) - pred)
        pred = int(blk[0])
        w.put(*dc_codes[s])
        w.put(bits, s)
        prev = 0
        cols = nzc[starts[k]:starts[k + 1]] + 1
        for i in cols:
            run = int(i) - prev - 1
            while run > 15:
                w.put(*ac_codes[0xf0])
                run -= 16
            s, bits = magnitude(int(blk[i]))
            w.put(*ac_codes[(run << 4) | s])
            w.put(bits, s)
            prev = int(i)
        if prev < 63:
            w.put(*ac_codes[0x00])
    return w.flush()


def noninterleaved_twin(data, ref):
    """One scan per component, blocks in raster order over the component's own block grid (T.81 A.2.2), from the quantised
    coefficients `ref` (oracle decode of `data`, STANDARD layout).  Same coefficients, same picture."""
    comps, dht, sos_comp, sos_at, i = [], {}, [], None, 2
    while True:
        m = data[i + 1]
        ln = struct.unpack(">H", data[i + 2:i + 4])[0]
        p = data[i + 4:i + 2 + ln]
        if m == 0xc0:
            H, W, n = struct.unpack(">HH", p[1:5]) + (p[5],)
            comps = [(p[6 + 3 * c], p[7 + 3 * c] >> 4, p[7 + 3 * c] & 15) for c in range(n)]
        elif m == 0xda:
            sos_comp = [(p[1 + 2 * c], p[2 + 2 * c] >> 4, p[2 + 2 * c] & 15) for c in range(p[0])]
            sos_at = i
            break
        i += 2 + ln
    for key, (bits, vals) in tables_from_jpeg(data).items():
        dht[key] = huff_codes(bits, vals)
    assert len(comps) == 3 and [c[0] for c in sos_comp] == [c[0] for c in comps]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux = (W + 8 * hmax - 1) // (8 * hmax)
    out = bytearray(data[:sos_at])
    for c, (cid, h, v) in enumerate(comps):
        bw = ((W * h + hmax - 1) // hmax + 7) // 8
        bh = ((H * v + vmax - 1) // vmax + 7) // 8
        mc = ref.coefs[c].reshape(ref.mcus, v, h, 64)
        yy, xx = np.mgrid[0:bh, 0:bw]
        raster = mc[(yy // v) * mcux + (xx // h), yy % v, xx % h].reshape(-1, 64)
        _, td, ta = sos_comp[c]
        out += bytes([0xff, 0xda, 0, 8, 1, cid, (td << 4) | ta, 0, 63, 0])
        out += _fast_encode_blocks(raster, dht[(0, td)], dht[(1, ta)])
    out += b"\xff\xd9"
    return bytes(out)
