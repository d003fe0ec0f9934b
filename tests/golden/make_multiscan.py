"""Writes non-interleaved (one scan per component) twins of interleaved baseline JPEGs: tests/golden/pil/ms_*.jpg.

    python tests/golden/make_multiscan.py

The quantised coefficients of a source file (decoded by the CPU oracle, test infrastructure) are re-encoded with the
source file's own Huffman tables into one scan per component, blocks in raster order over the component's own block
grid (T.81 A.2.2: the MCU padding blocks of the interleaved scan do not exist in a non-interleaved one).  A twin
therefore decodes to exactly the source's picture; Pillow (libjpeg) is used to confirm that before a file is written.
"""
import io, os, struct, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as orc

PIL_DIR = os.path.join(ROOT, "tests", "golden", "pil")


def segments(data):
    """-> list of (marker, payload offset, payload length) up to and including SOS."""
    out, i = [], 2
    while True:
        assert data[i] == 0xff, i
        m = data[i + 1]
        ln = struct.unpack(">H", data[i + 2:i + 4])[0]
        out.append((m, i + 4, ln - 2))
        if m == 0xda:
            return out
        i += 2 + ln


def huff_codes(bits, vals):
    codes, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            codes[vals[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return codes


class BitWriter:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, code, ln):
        self.acc = (self.acc << ln) | code
        self.n += ln
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


def encode_scan(blocks, dc, ac, restart=0, ncomp=1):
    """blocks: int16 [n, 64] zig-zag order, DC absolute; with ncomp > 1 the blocks of an MCU follow each other (one per
    component), dc / ac are lists of tables per component and restart counts MCUs."""
    if ncomp == 1:
        dc, ac = [dc], [ac]
    w, pred = BitWriter(), [0] * ncomp
    chunks, rst = [], 0
    for k, blk in enumerate(blocks):
        c = k % ncomp
        if restart and k and k % (restart * ncomp) == 0:
            chunks.append(w.flush() + bytes([0xff, 0xd0 + (rst & 7)]))
            rst += 1
            w, pred = BitWriter(), [0] * ncomp
        s, bits = magnitude(int(blk[0]) - pred[c])
        pred[c] = int(blk[0])
        w.put(*dc[c][s])
        if s:
            w.put(bits, s)
        run = 0
        last = max([i for i in range(1, 64) if blk[i]], default=0)
        for i in range(1, last + 1):
            if blk[i] == 0:
                run += 1
                continue
            while run > 15:
                w.put(*ac[c][0xf0])
                run -= 16
            s, bits = magnitude(blk[i])
            w.put(*ac[c][(run << 4) | s])
            w.put(bits, s)
            run = 0
        if last < 63:
            w.put(*ac[c][0x00])
    chunks.append(w.flush())
    return b"".join(chunks)


def twin(data, restart=0, chroma_together=False):
    """chroma_together: two scans, "0; 1 2;" (libjpeg wizard.txt: separate scans for luma and chroma) instead of three."""
    ref = orc.decode(data, layout=orc.LAYOUT_STD, ext_1bit=True)
    segs = segments(data)
    comps, dht_dc, dht_ac, sos_comp = [], {}, {}, []
    for m, off, ln in segs:
        p = data[off:off + ln]
        if m == 0xc0:
            H, W, n = struct.unpack(">HH", p[1:5]) + (p[5],)
            comps = [(p[6 + 3 * c], p[7 + 3 * c] >> 4, p[7 + 3 * c] & 15) for c in range(n)]
        elif m == 0xc4:
            i = 0
            while i < len(p):
                tc, th = p[i] >> 4, p[i] & 15
                bits = list(p[i + 1:i + 17])
                nv = sum(bits)
                (dht_ac if tc else dht_dc)[th] = huff_codes(bits, list(p[i + 17:i + 17 + nv]))
                i += 17 + nv
        elif m == 0xda:
            sos_comp = [(p[1 + 2 * c], p[2 + 2 * c] >> 4, p[2 + 2 * c] & 15) for c in range(p[0])]
            sos_at = off - 4
    assert len(sos_comp) == len(comps) == 3 and [c[0] for c in sos_comp] == [c[0] for c in comps]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux = (W + 8 * hmax - 1) // (8 * hmax)
    out = bytearray(data[:sos_at])
    if restart:
        out += bytes([0xff, 0xdd, 0, 4]) + struct.pack(">H", restart)
    rasters = []
    for c, (cid, h, v) in enumerate(comps):
        bw = ((W * h + hmax - 1) // hmax + 7) // 8          # the component's own block grid
        bh = ((H * v + vmax - 1) // vmax + 7) // 8
        mc = ref.coefs[c].reshape(ref.mcus, v, h, 64)        # MCU order: [mcu][by][bx]
        raster = np.empty((bh, bw, 64), np.int16)
        for y in range(bh):
            for x in range(bw):
                raster[y, x] = mc[(y // v) * mcux + (x // h), y % v, x % h]
        rasters.append(raster)
    for c, (cid, h, v) in enumerate(comps):
        _, td, ta = sos_comp[c]
        if chroma_together and c == 1:
            # Cb and Cr interleaved: both 1x1 here, so the scan's MCU is one block of each, over their common block grid
            assert comps[1][1:] == comps[2][1:] == (1, 1) and rasters[1].shape == rasters[2].shape
            _, td2, ta2 = sos_comp[2]
            out += bytes([0xff, 0xda, 0, 10, 2, cid, (td << 4) | ta, comps[2][0], (td2 << 4) | ta2, 0, 63, 0])
            both = np.stack([rasters[1].reshape(-1, 64), rasters[2].reshape(-1, 64)], 1).reshape(-1, 64)
            out += encode_scan(both, [dht_dc[td], dht_dc[td2]], [dht_ac[ta], dht_ac[ta2]], restart, ncomp=2)
            break
        out += bytes([0xff, 0xda, 0, 8, 1, cid, (td << 4) | ta, 0, 63, 0])
        out += encode_scan(rasters[c].reshape(-1, 64), dht_dc[td], dht_ac[ta], restart)
    out += b"\xff\xd9"
    out = bytes(out)
    a = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    b = np.asarray(Image.open(io.BytesIO(out)).convert("RGB"))
    assert np.array_equal(a, b), "libjpeg decodes the twin differently"
    return out


if __name__ == "__main__":
    jobs = [("std_420_big.jpg", "ms_420_big.jpg", 0), ("opt_444_q40.jpg", "ms_444_q40.jpg", 0), ("opt_422_q95.jpg", "ms_422_q95.jpg", 0),
            ("opt_420_q85.jpg", "ms_420_q85_rst.jpg", 7), ("dri_420_r5_plain.jpg", "ms_420_odd.jpg", 0),
            ("std_420_big.jpg", "ms2_420_big.jpg", 0), ("opt_420_q85.jpg", "ms2_420_q85_rst.jpg", 5), ("opt_444_q40.jpg", "ms2_444_q40.jpg", 0)]
    for src, dst, rst in jobs:
        data = open(os.path.join(PIL_DIR, src), "rb").read()
        t = twin(data, rst, chroma_together=dst.startswith("ms2_"))
        open(os.path.join(PIL_DIR, dst), "wb").write(t)
        im = Image.open(io.BytesIO(t))
        print(dst, im.size, len(data), "->", len(t), "bytes")
